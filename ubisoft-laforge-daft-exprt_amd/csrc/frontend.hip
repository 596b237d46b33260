// K17 -- mel / energy front-end of the synthesis path:
//   `mel_spectrogram_HiFi` (extract_features.py:330-359): Hann-windowed STFT (n_fft 1024, hop 256, optional centre /
//   reflect padding), |X| = sqrt(re^2 + im^2 + 1e-9), mel filterbank, log(clamp(., min_clipping));
//   `extract_energy(exp(mel))` (extract_features.py:299-304, generate.py:457): L2 norm over the mel channels.
//
// One workgroup per frame, everything in LDS: the windowed frame (hop-strided, reflect-padded -- no padded copy of the
// waveform, no frame matrix) goes through a radix-4 Stockham FFT (n_fft / 4 threads, one butterfly per thread per stage,
// ping-pong buffers, twiddles from a device table computed in double precision), the 513 magnitudes stay in LDS, the
// mel projection walks each filter's non-zero bin range there (a bin feeds at most two filters), and the frame energy is
// reduced in the same workgroup.  HBM traffic = the algorithmic minimum: 1 KB of waveform in, 324 B out per frame.
// (The first version computed the DFT as a dense fp32 MFMA GEMM: 2.1 MFLOP per frame, 24.5 M frames/s = 33 % of the fp32
// matrix peak -- but ~50x the arithmetic of an FFT for an op whose roofline is HBM.)
#include "dx_common.h"

namespace {

// twiddle[t] = exp(-2 pi i t / n_fft) (cos, sin), window[n] = periodic Hann (torch.hann_window)
__global__ void fe_tables_kernel(float* __restrict__ twiddle, float* __restrict__ window, int n_fft) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= n_fft) return;
  double s, c;
  sincospi(2.0 * (double)n / (double)n_fft, &s, &c);
  twiddle[2 * n] = (float)c;
  twiddle[2 * n + 1] = (float)(-s);
  window[n] = (float)(0.5 - 0.5 * c);
}

struct FeArgs {
  const float* wav; long ldw; const int64_t* n_samples; const float* twiddle; const float* window;
  const float* fb; const int* lo; const int* hi;
  float* mel; float* energy; int64_t* n_frames;
  int T, n_mel, hop, centered; float min_clip;
};

__device__ __forceinline__ int fe_frames(long ns, int n_fft, int hop, int centered) {
  if (centered) return ns >= 1 ? (int)(1 + ns / hop) : 0;        // torch.stft, center=True
  return ns >= n_fft ? (int)(1 + (ns - n_fft) / hop) : 0;
}

struct cplx { float re, im; };
__device__ __forceinline__ cplx cmul(cplx a, cplx b) { return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }

// grid (T, B); NFFT / 4 threads
template <int NFFT>
__global__ __launch_bounds__(NFFT / 4) void fe_fft_mel_kernel(FeArgs a) {
  constexpr int NT = NFFT / 4, NB = NFFT / 2 + 1;
  __shared__ float bufr[2][NFFT], bufi[2][NFFT];
  __shared__ float mag[NB + 3];
  __shared__ float red[NT / 64 > 0 ? NT / 64 : 1];
  const int j = threadIdx.x, f = blockIdx.x, b = blockIdx.y;
  const long ns = (long)a.n_samples[b];
  const int nfr = fe_frames(ns, NFFT, a.hop, a.centered);
  if (f == 0 && j == 0) a.n_frames[b] = nfr;
  if (f >= nfr) {                                                 // frames past the utterance: zeros
    for (int m = j; m < a.n_mel; m += NT) a.mel[((long)b * a.n_mel + m) * a.T + f] = 0.f;
    if (j == 0) a.energy[(long)b * a.T + f] = 0.f;
    return;
  }
  const float* wav = a.wav + (long)b * a.ldw;
  const long start = (long)f * a.hop - (a.centered ? NFFT / 2 : 0);
#pragma unroll
  for (int r = 0; r < 4; ++r) {                                   // windowed frame, reflect padding as torch's pad_mode='reflect'
    const int n = j + r * NT;
    long i = start + n;
    if (i < 0) i = -i;
    if (i >= ns) i = 2 * (ns - 1) - i;
    bufr[0][n] = (i >= 0 && i < ns) ? wav[i] * a.window[n] : 0.f;
    bufi[0][n] = 0.f;
  }
  __syncthreads();
  int cur = 0;
#pragma unroll
  for (int Ns = 1; Ns < NFFT; Ns *= 4) {                          // radix-4 Stockham stages
    const int k = j & (Ns - 1);
    cplx v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = {bufr[cur][j + r * NT], bufi[cur][j + r * NT]};
    if (Ns > 1) {
      const int t = k * (NFFT / (4 * Ns));                        // angle = -2 pi k / (4 Ns)
#pragma unroll
      for (int r = 1; r < 4; ++r) {
        const cplx w = {a.twiddle[2 * (r * t)], a.twiddle[2 * (r * t) + 1]};
        v[r] = cmul(v[r], w);
      }
    }
    const cplx s02 = {v[0].re + v[2].re, v[0].im + v[2].im}, d02 = {v[0].re - v[2].re, v[0].im - v[2].im};
    const cplx s13 = {v[1].re + v[3].re, v[1].im + v[3].im}, d13 = {v[1].re - v[3].re, v[1].im - v[3].im};
    const cplx y0 = {s02.re + s13.re, s02.im + s13.im}, y2 = {s02.re - s13.re, s02.im - s13.im};
    const cplx y1 = {d02.re + d13.im, d02.im - d13.re};           // d02 - i d13
    const cplx y3 = {d02.re - d13.im, d02.im + d13.re};           // d02 + i d13
    const int o = (j - k) * 4 + k;                                // (j / Ns) * 4 Ns + k
    bufr[cur ^ 1][o] = y0.re; bufi[cur ^ 1][o] = y0.im;
    bufr[cur ^ 1][o + Ns] = y1.re; bufi[cur ^ 1][o + Ns] = y1.im;
    bufr[cur ^ 1][o + 2 * Ns] = y2.re; bufi[cur ^ 1][o + 2 * Ns] = y2.im;
    bufr[cur ^ 1][o + 3 * Ns] = y3.re; bufi[cur ^ 1][o + 3 * Ns] = y3.im;
    cur ^= 1;
    __syncthreads();
  }
  for (int k = j; k < NB; k += NT) {
    const float re = bufr[cur][k], im = bufi[cur][k];
    mag[k] = sqrtf(re * re + im * im + 1e-9f);
  }
  __syncthreads();
  float e2 = 0.f;
  for (int m = j; m < a.n_mel; m += NT) {
    const float* w = a.fb + (long)m * NB;
    float s = 0.f;
    for (int k = a.lo[m]; k < a.hi[m]; ++k) s = fmaf(w[k], mag[k], s);
    const float c = fmaxf(s, a.min_clip);
    a.mel[((long)b * a.n_mel + m) * a.T + f] = logf(c);
    e2 += c * c;
  }
  e2 = dx_wave_sum(e2);
  if ((j & 63) == 0) red[j >> 6] = e2;
  __syncthreads();
  if (j == 0) {
    float t = 0.f;
    for (int w = 0; w < NT / 64; ++w) t += red[w];
    a.energy[(long)b * a.T + f] = sqrtf(t);
  }
}

}  // namespace

extern "C" int dx_mel_tables(float* twiddle, float* window, int n_fft, void* stream) {
  DX_REQUIRE(twiddle && window && (n_fft == 256 || n_fft == 1024 || n_fft == 4096), DX_ERR_ARG,
             "dx_mel_tables: n_fft=%d (256, 1024 or 4096)", n_fft);
  hipLaunchKernelGGL(fe_tables_kernel, dim3(dx_cdiv(n_fft, 256)), dim3(256), 0, (hipStream_t)stream, twiddle, window, n_fft);
  DX_LAUNCH_CHECK();
  return DX_OK;
}

extern "C" int dx_mel_spectrogram(const float* wav, long ldw, const int64_t* n_samples, const float* twiddle, const float* window,
                                  const float* fb, const int* fb_lo, const int* fb_hi, float* mel, float* energy,
                                  int64_t* n_frames, int B, int T, int n_fft, int hop, int n_mel, int centered, float min_clip,
                                  void* stream) {
  DX_REQUIRE(wav && n_samples && twiddle && window && fb && fb_lo && fb_hi && mel && energy && n_frames, DX_ERR_ARG,
             "dx_mel_spectrogram: null pointer");
  DX_REQUIRE(B > 0 && T > 0 && hop > 0 && n_mel > 0, DX_ERR_SHAPE, "dx_mel_spectrogram: bad shape B=%d T=%d hop=%d n_mel=%d", B, T, hop, n_mel);
  FeArgs a{wav, ldw, n_samples, twiddle, window, fb, fb_lo, fb_hi, mel, energy, n_frames, T, n_mel, hop, centered, min_clip};
  hipStream_t s = (hipStream_t)stream;
  dim3 grid(T, B);
  if (n_fft == 1024) hipLaunchKernelGGL(fe_fft_mel_kernel<1024>, grid, dim3(256), 0, s, a);
  else if (n_fft == 256) hipLaunchKernelGGL(fe_fft_mel_kernel<256>, grid, dim3(64), 0, s, a);
  else if (n_fft == 4096) hipLaunchKernelGGL(fe_fft_mel_kernel<4096>, grid, dim3(1024), 0, s, a);
  else { dx_set_error("dx_mel_spectrogram: n_fft=%d unsupported (256, 1024, 4096: radix-4 stages)", n_fft); return DX_ERR_UNSUPPORTED; }
  DX_LAUNCH_CHECK();
  return DX_OK;
}
