// K17 -- mel / energy front-end of the synthesis path:
//   `mel_spectrogram_HiFi` (extract_features.py:330-359): Hann-windowed STFT (n_fft 1024, hop 256, optional centre /
//   reflect padding), |X| = sqrt(re^2 + im^2 + 1e-9), mel filterbank, log(clamp(., min_clipping));
//   `extract_energy(exp(mel))` (extract_features.py:299-304, generate.py:457): L2 norm over the mel channels.
//
// The STFT is a dense contraction frames x n_fft x (2 * bins): it runs on the exact-fp32 MFMA
// (v_mfma_f32_32x32x2_f32, fp32 products, fp32 accumulation) against a precomputed [column][n] DFT basis whose columns
// interleave cos / -sin, so a bin's real and imaginary parts land in neighbouring lanes of the accumulator tile.
// Framing (hop-strided, reflect-padded windows) happens in the operand loader: no padded copy of the waveform, no
// (frames, n_fft) matrix in memory.  The mel projection is sparse (a bin feeds at most two filters): a second small
// kernel walks each filter's non-zero bin range and also produces the frame energies.
#include "dx_common.h"

namespace {

constexpr int FE_BM = 64, FE_BN = 128, FE_BK = 32, FE_LD = FE_BK + 4, FE_THREADS = 256;

// basis[c][n] for column c = 2 k (cos) / 2 k + 1 (-sin), n < n_fft; window[n] = periodic Hann (torch.hann_window)
__global__ void fe_tables_kernel(float* __restrict__ basis, float* __restrict__ window, int n_fft, int ncols_pad) {
  const long total = (long)ncols_pad * n_fft;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i / n_fft), n = (int)(i - (long)c * n_fft), k = c >> 1;
    float v = 0.f;
    if (k <= n_fft / 2) {
      const long kn = ((long)k * n) % n_fft;                    // exact phase reduction
      double s, co;
      sincospi(2.0 * (double)kn / (double)n_fft, &s, &co);
      v = (c & 1) ? (float)(-s) : (float)co;
    }
    basis[i] = v;
    if (c == 0) {
      double s, co;
      sincospi(2.0 * (double)n / (double)n_fft, &s, &co);
      window[n] = (float)(0.5 - 0.5 * co);
    }
  }
}

struct StftArgs {
  const float* wav; long ldw; const int64_t* n_samples; const float* basis; const float* window;
  float* mag; int T, nb_pad, n_fft, hop, centered, ncols;
};

__device__ __forceinline__ int fe_frames(long ns, int n_fft, int hop, int centered) {
  if (centered) return ns >= 1 ? (int)(1 + ns / hop) : 0;        // torch.stft, center=True (reflect padding needs >= 2 samples)
  return ns >= n_fft ? (int)(1 + (ns - n_fft) / hop) : 0;
}

// grid (ceil(T / 64), ceil(ncols / 128), B): 64 frames x 128 basis columns (= 64 bins) per workgroup
__global__ __launch_bounds__(FE_THREADS) void fe_stft_kernel(StftArgs a) {
  __shared__ __attribute__((aligned(16))) float As[FE_BM * FE_LD];
  __shared__ __attribute__((aligned(16))) float Bs[FE_BN * FE_LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, g = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  const int b = blockIdx.z, f0 = blockIdx.x * FE_BM, c0 = blockIdx.y * FE_BN;
  const long ns = (long)a.n_samples[b];
  const int nfr = fe_frames(ns, a.n_fft, a.hop, a.centered);
  if (f0 >= nfr) return;
  const float* wav = a.wav + (long)b * a.ldw;
  const int shift = a.centered ? a.n_fft / 2 : 0;
  f32x16 acc[2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  for (int k0 = 0; k0 < a.n_fft; k0 += FE_BK) {
    {   // A: 64 frames x 32 samples, windowed; thread = (frame, 8 consecutive samples)
      const int fr = tid >> 2, kk = (tid & 3) * 8;
      const int f = f0 + fr;
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        long i = (long)f * a.hop + k0 + kk + e - shift;
        if (i < 0) i = -i;                                        // reflect (no edge repeat), as torch's pad_mode='reflect'
        if (i >= ns) i = 2 * (ns - 1) - i;
        v[e] = (f < nfr && i >= 0 && i < ns) ? wav[i] * a.window[k0 + kk + e] : 0.f;
      }
      *reinterpret_cast<f32x4*>(&As[fr * FE_LD + kk]) = f32x4{v[0], v[1], v[2], v[3]};
      *reinterpret_cast<f32x4*>(&As[fr * FE_LD + kk + 4]) = f32x4{v[4], v[5], v[6], v[7]};
    }
    {   // B: 128 columns x 32 samples of the basis; thread = (column, 16 consecutive samples)
      const int cl = tid >> 1, kk = (tid & 1) * 16;
      const float* src = a.basis + (long)(c0 + cl) * a.n_fft + k0 + kk;
#pragma unroll
      for (int q = 0; q < 4; ++q)
        *reinterpret_cast<f32x4*>(&Bs[cl * FE_LD + kk + 4 * q]) = (c0 + cl < a.ncols) ? *reinterpret_cast<const f32x4*>(src + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < FE_BK / 16; ++ks) {
      const f32x8 af = dx_load8<float, float>(&As[(wm * 32 + l31) * FE_LD + ks * 16 + g * 8]);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const f32x8 bf = dx_load8<float, float>(&Bs[(wn * 64 + j * 32 + l31) * FE_LD + ks * 16 + g * 8]);
        dx_mma(acc[j], af, bf);
      }
    }
    __syncthreads();
  }
  // lane = column: even lanes hold re, their right neighbours im of the same bin
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = c0 + wn * 64 + j * 32 + l31, bin = col >> 1;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float re = acc[j][r], im = __shfl_xor(re, 1, 64);
      const int f = f0 + wm * 32 + dx_acc_row(r, g);
      if (!(l31 & 1) && f < nfr && bin <= a.n_fft / 2) a.mag[((long)b * a.T + f) * a.nb_pad + bin] = sqrtf(re * re + im * im + 1e-9f);
    }
  }
}

struct MelArgs {
  const float* mag; const float* fb; const int* lo; const int* hi; const int64_t* n_samples;
  float* mel; float* energy; int64_t* n_frames;
  int T, nb_pad, nbins, n_mel, n_fft, hop, centered; float min_clip;
};

// grid (T, B), 128 threads: thread m < n_mel = one mel channel of one frame; frames past the utterance: zeros
__global__ __launch_bounds__(128) void fe_mel_kernel(MelArgs a) {
  __shared__ float red[2];
  const int f = blockIdx.x, b = blockIdx.y, m = threadIdx.x;
  const int nfr = fe_frames((long)a.n_samples[b], a.n_fft, a.hop, a.centered);
  if (f == 0 && m == 0) a.n_frames[b] = nfr;
  float v = 0.f, e2 = 0.f;
  if (m < a.n_mel && f < nfr) {
    const float* row = a.mag + ((long)b * a.T + f) * a.nb_pad;
    const float* w = a.fb + (long)m * a.nbins;
    float s = 0.f;
    for (int k = a.lo[m]; k < a.hi[m]; ++k) s = fmaf(w[k], row[k], s);
    const float c = fmaxf(s, a.min_clip);
    v = logf(c);
    e2 = c * c;
  }
  if (m < a.n_mel) a.mel[((long)b * a.n_mel + m) * a.T + f] = v;
  e2 = dx_wave_sum(e2);
  if ((m & 63) == 0) red[m >> 6] = e2;
  __syncthreads();
  if (m == 0) a.energy[(long)b * a.T + f] = f < nfr ? sqrtf(red[0] + red[1]) : 0.f;
}

}  // namespace

extern "C" int dx_mel_tables(float* basis, float* window, int n_fft, void* stream) {
  DX_REQUIRE(basis && window && n_fft >= 32 && n_fft % 32 == 0, DX_ERR_ARG, "dx_mel_tables: bad arguments");
  const int ncols_pad = dx_cdiv(2 * (n_fft / 2 + 1), FE_BN) * FE_BN;
  hipLaunchKernelGGL(fe_tables_kernel, dim3(1024), dim3(256), 0, (hipStream_t)stream, basis, window, n_fft, ncols_pad);
  DX_LAUNCH_CHECK();
  return DX_OK;
}

extern "C" long dx_mel_basis_floats(int n_fft) { return (long)dx_cdiv(2 * (n_fft / 2 + 1), FE_BN) * FE_BN * n_fft; }

extern "C" int dx_mel_spectrogram(const float* wav, long ldw, const int64_t* n_samples, const float* basis, const float* window,
                                  const float* fb, const int* fb_lo, const int* fb_hi, float* mag_ws, float* mel, float* energy,
                                  int64_t* n_frames, int B, int T, int n_fft, int hop, int n_mel, int centered, float min_clip,
                                  void* stream) {
  DX_REQUIRE(wav && n_samples && basis && window && fb && fb_lo && fb_hi && mag_ws && mel && energy && n_frames, DX_ERR_ARG,
             "dx_mel_spectrogram: null pointer");
  DX_REQUIRE(B > 0 && T > 0 && n_fft >= 32 && n_fft % 32 == 0 && hop > 0 && n_mel > 0 && n_mel <= 128, DX_ERR_SHAPE,
             "dx_mel_spectrogram: bad shape B=%d T=%d n_fft=%d hop=%d n_mel=%d", B, T, n_fft, hop, n_mel);
  const int nbins = n_fft / 2 + 1, ncols = 2 * nbins, nb_pad = dx_cdiv(nbins, 4) * 4;
  hipStream_t s = (hipStream_t)stream;
  StftArgs sa{wav, ldw, n_samples, basis, window, mag_ws, T, nb_pad, n_fft, hop, centered, dx_cdiv(ncols, FE_BN) * FE_BN};
  hipLaunchKernelGGL(fe_stft_kernel, dim3(dx_cdiv(T, FE_BM), dx_cdiv(ncols, FE_BN), B), dim3(FE_THREADS), 0, s, sa);
  MelArgs ma{mag_ws, fb, fb_lo, fb_hi, n_samples, mel, energy, n_frames, T, nb_pad, nbins, n_mel, n_fft, hop, centered, min_clip};
  hipLaunchKernelGGL(fe_mel_kernel, dim3(T, B), dim3(128), 0, s, ma);
  DX_LAUNCH_CHECK();
  return DX_OK;
}
