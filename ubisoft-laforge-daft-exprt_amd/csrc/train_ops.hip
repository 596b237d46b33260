// K13 losses (+ their gradients), K15 Adam / gradient norm, K16 float -> integer durations.
#include "dx_common.h"

namespace {

__device__ __forceinline__ float block_sum_256(float v, float* red) {
  v = dx_wave_sum(v);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  const float t = red[0] + red[1] + red[2] + red[3];
  __syncthreads();
  return t;
}

// ------------------------------------------------------------------ losses (loss.py:54-99)
// terms[0..6] = speaker, post_mult, duration, energy, pitch, mel_l1, mel_l2 (already weighted); terms[7] = total
struct SeqLossArgs {
  const float* pred[3]; const float* tgt[3]; float* dpred[3]; float w[3];
  const int64_t* lengths; float* terms; int B, L; float gscale;
  float* part;     // NULL: atomics on terms[2 + f]; else part[f * B + b] = this block's term (summed in a fixed order by the last launch)
};
// grid (B, 3)
__global__ __launch_bounds__(256) void seq_loss_kernel(SeqLossArgs a) {
  __shared__ float red[4];
  const int b = blockIdx.x, f = blockIdx.y;
  const float inv = 1.f / ((float)a.lengths[b] * (float)a.B);
  const float* p = a.pred[f] + (long)b * a.L; const float* t = a.tgt[f] + (long)b * a.L;
  float* dp = a.dpred[f] ? a.dpred[f] + (long)b * a.L : nullptr;
  float acc = 0.f;
  for (int l = threadIdx.x; l < a.L; l += 256) {
    const float d = p[l] - t[l];
    acc += d * d;
    if (dp) dp[l] = a.gscale * a.w[f] * 2.f * d * inv;
  }
  acc = block_sum_256(acc, red);
  if (threadIdx.x == 0) {
    if (a.part) a.part[f * a.B + b] = a.w[f] * acc * inv;
    else atomicAdd(a.terms + 2 + f, a.w[f] * acc * inv);
  }
}

// mel: pred/target (B, C, T); grid (chunks, B)
__global__ __launch_bounds__(256) void mel_loss_kernel(const float* __restrict__ pred, const float* __restrict__ tgt,
                                                       float* __restrict__ dpred, const int64_t* __restrict__ out_len,
                                                       float* terms, int B, int C, int T, float w, float gscale, int dtrans) {
  __shared__ float red[4];
  const int b = blockIdx.y;
  const float inv = 1.f / ((float)C * (float)out_len[b] * (float)B);
  const long n = (long)C * T, base = (long)b * n;
  float a1 = 0.f, a2 = 0.f;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += gridDim.x * 256L) {
    const float d = pred[base + i] - tgt[base + i];
    a1 += fabsf(d);
    a2 += d * d;
    if (dpred) dpred[dtrans ? base + (i % T) * C + i / T : base + i] = gscale * w * ((d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) + 2.f * d) * inv;
  }
  a1 = block_sum_256(a1, red);
  a2 = block_sum_256(a2, red);
  if (threadIdx.x == 0) { atomicAdd(terms + 5, w * a1 * inv); atomicAdd(terms + 6, w * a2 * inv); }
}

// the same terms with the gradient written TRANSPOSED, (B, T, C) for the channel-last decoder backward: a workgroup
// takes 64 frames x all channels, reads pred / tgt along t (coalesced), and writes the gradient tile along c through LDS
// (the direct transposed store above writes 4 bytes per 320-byte row: 76 us for 48 x 80 x 1000)
constexpr int MT_T = 64, MT_CMAX = 128, MT_TPW = 4;
__global__ __launch_bounds__(256) void mel_loss_t_kernel(const float* __restrict__ pred, const float* __restrict__ tgt,
                                                         float* __restrict__ dpred, const int64_t* __restrict__ out_len,
                                                         float* terms, int B, int C, int T, float w, float gscale, float* part, int tpw) {
  // part == NULL: `tpw` tiles per workgroup, two atomics on terms[5], terms[6] at the end (same-address atomics serialise at ~50 ns:
  // few, fat workgroups); part != NULL: one tile per workgroup (4x the workgroups for the 46 MB it streams), its two sums go to
  // part[2 * (b * gridDim.x + blockIdx.x)] and the last launch of dx_loss_fwd_bwd adds them in a fixed order
  __shared__ float tile[MT_CMAX][MT_T + 1];
  __shared__ float red[4];
  const int b = blockIdx.y;
  const float inv = 1.f / ((float)C * (float)out_len[b] * (float)B);
  const long base = (long)b * C * T;
  float a1 = 0.f, a2 = 0.f;
  constexpr int U = 10;                                 // loads in flight per thread and operand (80 bins: two rounds)
  // MT_TPW tiles of 64 frames per workgroup: every workgroup ends with two atomics on the SAME two addresses, and same-address atomics
  // serialise at ~50 ns each -- 768 one-tile workgroups spent more time queueing there (~30 us) than reading their 46 MB
  for (int tile_i = 0; tile_i < tpw; ++tile_i) {
    const int t0 = (blockIdx.x * tpw + tile_i) * MT_T;
    if (t0 >= T) break;
    for (int i0 = threadIdx.x; i0 < C * MT_T; i0 += 256 * U) {
      float pv[U], tv[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int i = i0 + u * 256, c = i / MT_T, t = t0 + i % MT_T;
        const bool ok = i < C * MT_T && t < T;
        pv[u] = ok ? pred[base + (long)c * T + t] : 0.f;
        tv[u] = ok ? tgt[base + (long)c * T + t] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int i = i0 + u * 256;
        if (i >= C * MT_T) break;
        const float d = pv[u] - tv[u];                    // 0 past T: contributes nothing
        a1 += fabsf(d);
        a2 += d * d;
        tile[i / MT_T][i % MT_T] = gscale * w * ((d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) + 2.f * d) * inv;
      }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < C * MT_T; i += 256) {
      const int tt = i / C, c = i % C, t = t0 + tt;
      if (t < T) dpred[base + (long)t * C + c] = tile[c][tt];
    }
    __syncthreads();
  }
  a1 = block_sum_256(a1, red);
  a2 = block_sum_256(a2, red);
  if (threadIdx.x == 0) {
    if (part) { part[2 * (b * gridDim.x + blockIdx.x)] = w * a1 * inv; part[2 * (b * gridDim.x + blockIdx.x) + 1] = w * a2 * inv; }
    else { atomicAdd(terms + 5, w * a1 * inv); atomicAdd(terms + 6, w * a2 * inv); }
  }
}

// speaker cross-entropy (mean over the batch) and post-multiplier L2 norm; one block
__global__ __launch_bounds__(256) void head_loss_kernel(const float* __restrict__ logits, const int64_t* __restrict__ ids,
                                                        float* __restrict__ dlogits, int B, int S, float w_spk,
                                                        const float* __restrict__ post, float* __restrict__ dpost, int npost,
                                                        float w_post, float* terms, float gscale, const DxStepScalars* step,
                                                        const float* seq_part, const float* mel_part, int n_mel_part) {
  // seq_part == NULL: FIRST launch of dx_loss_fwd_bwd: writes terms[0..1] and zeroes the accumulators terms[2..7] of the kernels behind
  // it (this replaces a hipMemsetAsync = one more dispatch per step).  seq_part != NULL: LAST launch: the other kernels left their
  // per-workgroup terms in seq_part [3][B] / mel_part [n_mel_part][2]; they are added here in a fixed order (run-to-run identical loss
  // terms, no same-address atomics) and the total is written: one launch less than the atomics form
  __shared__ float red[4];
  if (step) w_spk = step->w_speaker;   // captured steps: the adversarial weight of this iteration lives in device memory
  float ce = 0.f;
  for (int b = threadIdx.x; b < B; b += 256) {
    const float* z = logits + (long)b * S;
    float mx = -INFINITY;
    for (int s = 0; s < S; ++s) mx = fmaxf(mx, z[s]);
    float se = 0.f;
    for (int s = 0; s < S; ++s) se += expf(z[s] - mx);
    const int tgt = (int)ids[b];
    ce += (mx + logf(se)) - z[tgt];
    if (dlogits)
      for (int s = 0; s < S; ++s)
        dlogits[(long)b * S + s] = gscale * w_spk * (expf(z[s] - mx) / se - (s == tgt ? 1.f : 0.f)) / (float)B;
  }
  ce = block_sum_256(ce, red);
  float sq = 0.f;
  if (post) for (int i = threadIdx.x; i < npost; i += 256) sq += post[i] * post[i];
  sq = block_sum_256(sq, red);
  const float nrm = sqrtf(sq);
  if (post && dpost) for (int i = threadIdx.x; i < npost; i += 256) dpost[i] += nrm > 0.f ? gscale * w_post * post[i] / nrm : 0.f;
  float t5[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
  if (seq_part) {
    for (int f = 0; f < 3; ++f) {
      float v = 0.f;
      for (int b = threadIdx.x; b < B; b += 256) v += seq_part[f * B + b];
      t5[f] = block_sum_256(v, red);
    }
    float v1 = 0.f, v2 = 0.f;
    for (int i = threadIdx.x; i < n_mel_part; i += 256) { v1 += mel_part[2 * i]; v2 += mel_part[2 * i + 1]; }
    t5[3] = block_sum_256(v1, red);
    t5[4] = block_sum_256(v2, red);
  }
  if (threadIdx.x == 0) {
    terms[0] = w_spk * ce / (float)B;
    terms[1] = post ? w_post * nrm : 0.f;
    if (seq_part) {
      float tot = terms[0] + terms[1];
#pragma unroll
      for (int i = 0; i < 5; ++i) { terms[2 + i] = t5[i]; tot += t5[i]; }
      terms[7] = tot;
    } else {
#pragma unroll
      for (int i = 2; i < 8; ++i) terms[i] = 0.f;
    }
  }
}
__global__ void loss_total_kernel(float* terms) {
  float t = 0.f;
  for (int i = 0; i < 7; ++i) t += terms[i];
  terms[7] = t;
}

// ------------------------------------------------------------------ Adam (train.py:299-301) + gradient norm (train.py:399)
// 16-byte loads, four of them in flight per thread (the one-float-per-thread version ran at 1.7 TB/s); the head up to the
// first 16-byte boundary and the tail go through workgroup 0
__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ x, long n, float* out) {
  __shared__ float red[4];
  long head = (long)((16 - ((uintptr_t)x & 15)) & 15) / 4;
  if (head > n) head = n;
  const f32x4* xq = reinterpret_cast<const f32x4*>(x + head);
  const long nq = (n - head) >> 2, stride = gridDim.x * 256L;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  long i = blockIdx.x * 256L + threadIdx.x;
  for (; i + 3 * stride < nq; i += 4 * stride) {
    const f32x4 v0 = xq[i], v1 = xq[i + stride], v2 = xq[i + 2 * stride], v3 = xq[i + 3 * stride];
    a0 += v0[0] * v0[0] + v0[1] * v0[1] + v0[2] * v0[2] + v0[3] * v0[3];
    a1 += v1[0] * v1[0] + v1[1] * v1[1] + v1[2] * v1[2] + v1[3] * v1[3];
    a2 += v2[0] * v2[0] + v2[1] * v2[1] + v2[2] * v2[2] + v2[3] * v2[3];
    a3 += v3[0] * v3[0] + v3[1] * v3[1] + v3[2] * v3[2] + v3[3] * v3[3];
  }
  for (; i < nq; i += stride) { const f32x4 v = xq[i]; a0 += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3]; }
  if (blockIdx.x == 0) {
    for (long j = threadIdx.x; j < head; j += 256) a1 += x[j] * x[j];
    for (long j = head + nq * 4 + threadIdx.x; j < n; j += 256) a2 += x[j] * x[j];
  }
  float acc = block_sum_256((a0 + a1) + (a2 + a3), red);
  if (threadIdx.x == 0) atomicAdd(out, acc);
}
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, long n, float lr, float b1, float b2, float eps,
                                                   float wd, float bc1, float bc2_sqrt, const float* gnorm_sq, float clip,
                                                   float* gnorm_accum, const DxStepScalars* sc) {
  __shared__ float red[4];
  if (sc) { lr = sc->lr; bc1 = sc->bc1; bc2_sqrt = sc->bc2_sqrt; }   // captured steps: this iteration's scalars live in device memory
  float coef = 1.f;
  if (gnorm_sq && clip < INFINITY) coef = fminf(1.f, clip / (sqrtf(*gnorm_sq) + 1e-6f));  // clip_grad_norm_
  const float step = lr / bc1;
  float gsq = 0.f;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += gridDim.x * 256L) {
    const float pi = p[i], graw = g[i];
    gsq += graw * graw;
    const float gi = graw * coef + wd * pi;
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi; v[i] = vi;
    p[i] = pi - step * mi / (sqrtf(vi) / bc2_sqrt + eps);
  }
  if (gnorm_accum) {   // the gradient norm the trainer logs (clip_grad_norm_(inf), train.py:399), summed on the way: no separate pass over g
    gsq = block_sum_256(gsq, red);
    if (threadIdx.x == 0) atomicAdd(gnorm_accum, gsq);
  }
}
__global__ __launch_bounds__(256) void scale_kernel(float* __restrict__ x, long n, float s) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += gridDim.x * 256L) x[i] *= s;
}

// ------------------------------------------------------------------ integer durations (model.py:789-812 + extract_features.py:69-111)
__device__ __forceinline__ long floordiv(long a, long b) { long q = a / b; return (a % b != 0 && ((a < 0) != (b < 0))) ? q - 1 : q; }

// fp64 + integer arithmetic in the reference's order of operations: the walk over the symbols is SERIAL by definition (a running fp64
// sum whose rounding the integer frame counts depend on), so one lane per utterance does it -- but on a copy of the row in LDS: the
// first version walked global memory, one dependent ~1 us round trip per symbol and loop (207 us for B = 256, L = 160: 2 % of a
// synthesis call).  One wave per utterance: coalesced load (+ duration factors + threshold), lane 0 walks LDS, coalesced write-back.
constexpr int ID_LMAX = 2048;     // symbols per utterance held in LDS (24 KB); longer rows walk global memory as before
template <typename DP, typename OP>
__device__ __forceinline__ void int_durations_walk(DP d, OP o, int L, double sr, int fl, int hop, int centered, int64_t* total_out, int* status_out) {
  double end_prev = 0.0, total = 0.0;
  int nspans = 0;
  for (int l = 0; l < L; ++l) {
    o[l] = 0;
    if (d[l] != 0.f) {
      const double e = end_prev + (double)d[l];
      total = total + (e - end_prev);
      end_prev += (double)d[l];
      ++nspans;
    }
  }
  const long nb_samples = (long)(total * sr);
  const long nb_frames = 1 + (long)((double)(nb_samples - fl) / (double)hop);
  const long half = fl / 2;
  long assigned = 0;
  int consumed = 0, l = 0, first = -1, last = -1, st = 0;
  end_prev = 0.0;
  while (assigned + 1 <= nb_frames) {
    while (l < L && d[l] == 0.f) ++l;
    if (l >= L) { st = 1; break; }                       // IndexError: pop from empty list
    const double bg = end_prev, e = end_prev + (double)d[l];
    end_prev += (double)d[l];
    const long sb = (long)(bg * sr), se = (long)(e * sr);
    long lo = floordiv(sb + 1 - half + hop - 1, hop); if (lo < 0) lo = 0;   // ceil((sb + 1 - half) / hop)
    long hi = floordiv(se - half, hop); if (hi > nb_frames - 1) hi = nb_frames - 1;
    const long n = hi - lo + 1 > 0 ? hi - lo + 1 : 0;
    o[l] = n;
    assigned += n;
    if (first < 0) first = l;
    last = l;
    ++consumed; ++l;
  }
  if (!st && centered) {
    const long edge = (long)((double)fl / 2.0 / (double)hop);
    if (first < 0) st = 1;                               // IndexError: int_durations[0] on an empty list
    else {
      o[first] += edge;
      if (consumed < nspans) {
        while (l < L && d[l] == 0.f) ++l;
        o[l] = edge; ++consumed;
      } else o[last] += edge;
    }
  }
  if (!st && consumed != nspans) st = 2;                 // shape mismatch in the reference's index_put
  int64_t tot = 0;
  for (int i = 0; i < L; ++i) tot += o[i];
  *total_out = tot;
  *status_out = st;
}

__global__ __launch_bounds__(64) void int_durations_kernel(float* __restrict__ dur, const float* __restrict__ dur_factors, int64_t* __restrict__ out,
                                                           int64_t* __restrict__ totals, int* __restrict__ status, int B, int L, double sr, int fl,
                                                           int hop, int centered) {
  __shared__ float ds[ID_LMAX];
  __shared__ int64_t os[ID_LMAX];
  const int b = blockIdx.x, lane = threadIdx.x;
  float* d = dur + (long)b * L;
  int64_t* o = out + (long)b * L;
  const float dmin = (float)((double)fl / sr / 2.0);   // compared in fp32, like `tensor < python_float`
  const bool in_lds = L <= ID_LMAX;
  for (int l = lane; l < L; l += 64) {                   // model.py:891 (duration factors) and the threshold of extract_features.py:69-111
    float v = d[l];
    if (dur_factors) v *= dur_factors[(long)b * L + l];
    if (v < dmin) v = 0.f;
    d[l] = v;
    if (in_lds) ds[l] = v;
  }
  __syncthreads();
  if (lane == 0) {
    if (in_lds) int_durations_walk(ds, os, L, sr, fl, hop, centered, totals + b, status + b);
    else int_durations_walk(d, o, L, sr, fl, hop, centered, totals + b, status + b);
  }
  __syncthreads();
  if (in_lds)
    for (int l = lane; l < L; l += 64) o[l] = os[l];
}


// ------------------------------------------------------------------ inference-time prosody control (model.py:895-905, 814-864)
// one wave per utterance.  mode 0 = 'add' (pitch_shift), 1 = 'multiply' (pitch_multiply)
__global__ __launch_bounds__(64) void prosody_control_kernel(float* __restrict__ energy, float* __restrict__ pitch,
                                                             const float* __restrict__ energy_factors,
                                                             const float* __restrict__ pitch_factors,
                                                             const int64_t* __restrict__ dint, const int64_t* __restrict__ spk,
                                                             const float* __restrict__ spk_mean, const float* __restrict__ spk_std,
                                                             int mode, int L) {
  const int b = blockIdx.x, lane = threadIdx.x;
  float* e = energy + (long)b * L; float* p = pitch + (long)b * L;
  const float* ef = energy_factors + (long)b * L; const float* pf = pitch_factors + (long)b * L;
  const int64_t* di = dint + (long)b * L;
  float sum = 0.f, cnt = 0.f;
  for (int l = lane; l < L; l += 64) {
    const bool z = di[l] == 0;
    e[l] = z ? 0.f : e[l] * ef[l];
    const float pv = z ? 0.f : p[l];
    p[l] = pv;
    if (pv != 0.f) { sum += pv; cnt += 1.f; }
  }
  if (mode == 0) {
    const float mean = spk_mean[spk[b]], sd = spk_std[spk[b]];
    for (int l = lane; l < L; l += 64) {
      const float pv = p[l];
      if (pv != 0.f) p[l] = (logf(expf(sd * pv + mean) + pf[l]) - mean) / sd;
    }
  } else {
    sum = dx_wave_sum(sum); cnt = dx_wave_sum(cnt);
    const float mean = sum / cnt;
    for (int l = lane; l < L; l += 64) {
      const float pv = p[l];
      if (pv != 0.f) { const float dev = (pv - mean) * pf[l]; p[l] = pv + dev; }
    }
  }
}

inline int grid_for(long total, int cap = 2048) {
  long g = (total + 255) / 256;
  return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

}  // namespace

extern "C" long dx_loss_ws_floats(int B, int T) { return (B <= 0 || T <= 0) ? 0 : 3L * B + 2L * B * dx_cdiv(T, MT_T); }

extern "C" int dx_loss_fwd_bwd(const float* dur, const float* energy, const float* pitch, const float* dur_t,
                               const float* energy_t, const float* pitch_t, const int64_t* in_lengths, const float* mel,
                               const float* mel_t, const int64_t* out_lengths, const float* spk_logits,
                               const int64_t* spk_ids, const float* post_mult, float* d_dur, float* d_energy,
                               float* d_pitch, float* d_mel, float* d_spk_logits, float* d_post_mult, float* terms, float* ws,
                               int B, int L, int T, int n_mel, int n_spk_classes, int n_post, float w_spk, float w_post,
                               float w_dur, float w_energy, float w_pitch, float w_mel, float grad_scale,
                               int d_mel_transposed, const DxStepScalars* step, void* stream) {
  DX_REQUIRE(dur && energy && pitch && dur_t && energy_t && pitch_t && in_lengths && mel && mel_t && out_lengths && spk_logits &&
             spk_ids && terms, DX_ERR_ARG, "dx_loss_fwd_bwd: null pointer");
  DX_REQUIRE(B > 0 && L > 0 && T > 0, DX_ERR_SHAPE, "dx_loss_fwd_bwd: empty shape");
  hipStream_t s = (hipStream_t)stream;
  // ws (dx_loss_ws_floats(B, T) floats, no initialisation): per-workgroup terms added in a fixed order by the last launch (3 launches, no
  // atomics, 4x the workgroups on the mel term); only with the transposed mel gradient (the training step's form).  NULL: atomics, 4 launches.
  const bool part = ws && d_mel && d_mel_transposed && n_mel <= MT_CMAX;
  float* seq_part = part ? ws : nullptr;
  float* mel_part = part ? ws + 3L * B : nullptr;
  if (!part)
    hipLaunchKernelGGL(head_loss_kernel, dim3(1), dim3(256), 0, s, spk_logits, spk_ids, d_spk_logits, B, n_spk_classes, w_spk,
                       post_mult, d_post_mult, n_post, w_post, terms, grad_scale, step, nullptr, nullptr, 0);
  SeqLossArgs a{{dur, energy, pitch}, {dur_t, energy_t, pitch_t}, {d_dur, d_energy, d_pitch}, {w_dur, w_energy, w_pitch},
                in_lengths, terms, B, L, grad_scale, seq_part};
  hipLaunchKernelGGL(seq_loss_kernel, dim3(B, 3), dim3(256), 0, s, a);
  int chunks = dx_cdiv(n_mel * T, 256 * 8);
  if (chunks > 64) chunks = 64;
  if (d_mel && d_mel_transposed && n_mel <= MT_CMAX) {
    const int tpw = part ? 1 : MT_TPW;
    hipLaunchKernelGGL(mel_loss_t_kernel, dim3(dx_cdiv(T, MT_T * tpw), B), dim3(256), 0, s, mel, mel_t, d_mel, out_lengths, terms, B, n_mel, T, w_mel,
                       grad_scale, mel_part, tpw);
  } else {
    hipLaunchKernelGGL(mel_loss_kernel, dim3(chunks, B), dim3(256), 0, s, mel, mel_t, d_mel, out_lengths, terms, B, n_mel, T, w_mel, grad_scale, d_mel_transposed);
  }
  if (part)
    hipLaunchKernelGGL(head_loss_kernel, dim3(1), dim3(256), 0, s, spk_logits, spk_ids, d_spk_logits, B, n_spk_classes, w_spk,
                       post_mult, d_post_mult, n_post, w_post, terms, grad_scale, step, seq_part, mel_part, B * dx_cdiv(T, MT_T));
  else
    hipLaunchKernelGGL(loss_total_kernel, dim3(1), dim3(1), 0, s, terms);
  DX_LAUNCH_CHECK();
  return DX_OK;
}

extern "C" int dx_sumsq(const float* x, long n, float* out, void* stream) {
  DX_REQUIRE(x && out && n >= 0, DX_ERR_ARG, "dx_sumsq: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  if (int rc = dx_fill_zero(out, sizeof(float), s)) return rc;
  if (n) hipLaunchKernelGGL(sumsq_kernel, dim3(grid_for(n / 8 + 1, 1024)), dim3(256), 0, s, x, n, out);
  DX_LAUNCH_CHECK();
  return DX_OK;
}

extern "C" int dx_adam_step(float* p, const float* g, float* m, float* v, long n, float lr, float beta1, float beta2,
                            float eps, float weight_decay, int step, const float* grad_norm_sq, float clip_thresh,
                            float* grad_norm_sq_accum, const DxStepScalars* scalars, void* stream) {
  DX_REQUIRE(p && g && m && v && n > 0 && (step >= 1 || scalars), DX_ERR_ARG, "dx_adam_step: bad arguments");
  if (step < 1) step = 1;   // (unused when `scalars` carries the bias corrections)
  const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
  hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, lr, beta1, beta2, eps,
                     weight_decay, (float)bc1, (float)sqrt(bc2), grad_norm_sq, clip_thresh, grad_norm_sq_accum, scalars);
  DX_LAUNCH_CHECK();
  return DX_OK;
}

// ---- the device-side step block (DxStepScalars): one single-thread launch per optimizer step, in front of the (captured) step
__global__ void step_scalars_kernel(DxStepScalars* out, uint64_t salt, float lr, float bc1, float bc2_sqrt, float w_speaker, int step) {
  out->seed_salt = salt; out->lr = lr; out->bc1 = bc1; out->bc2_sqrt = bc2_sqrt; out->w_speaker = w_speaker; out->step = step;
}

extern "C" int dx_step_scalars_set(DxStepScalars* dev, uint64_t seed_salt, float lr, float beta1, float beta2, int step, float w_speaker,
                                   void* stream) {
  DX_REQUIRE(dev && step >= 1, DX_ERR_ARG, "dx_step_scalars_set: bad arguments");
  const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
  hipLaunchKernelGGL(step_scalars_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, dev, seed_salt, lr, (float)bc1, (float)sqrt(bc2), w_speaker, step);
  DX_LAUNCH_CHECK();
  return DX_OK;
}

extern "C" int dx_scale(float* x, long n, float s, void* stream) {
  DX_REQUIRE(x && n >= 0, DX_ERR_ARG, "dx_scale: bad arguments");
  if (n) hipLaunchKernelGGL(scale_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, x, n, s);
  DX_LAUNCH_CHECK();
  return DX_OK;
}

extern "C" int dx_int_durations(float* duration_preds, const float* dur_factors, int64_t* durations_int, int64_t* totals, int* status, int B, int L,
                                double sampling_rate, int filter_length, int hop_length, int centered, void* stream) {
  DX_REQUIRE(duration_preds && durations_int && totals && status, DX_ERR_ARG, "dx_int_durations: null pointer");
  DX_REQUIRE(B > 0 && L > 0 && hop_length > 0 && filter_length > 0, DX_ERR_SHAPE, "dx_int_durations: bad shape");
  hipLaunchKernelGGL(int_durations_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, duration_preds, dur_factors,
                     durations_int, totals, status, B, L, sampling_rate, filter_length, hop_length, centered);
  DX_LAUNCH_CHECK();
  return DX_OK;
}

extern "C" int dx_prosody_control(float* energy, float* pitch, const float* energy_factors, const float* pitch_factors,
                                  const int64_t* durations_int, const int64_t* speaker_ids, const float* spk_pitch_mean,
                                  const float* spk_pitch_std, int mode, int B, int L, void* stream) {
  DX_REQUIRE(energy && pitch && energy_factors && pitch_factors && durations_int, DX_ERR_ARG, "dx_prosody_control: null pointer");
  DX_REQUIRE(mode == 1 || (speaker_ids && spk_pitch_mean && spk_pitch_std), DX_ERR_ARG, "dx_prosody_control: 'add' needs speaker stats");
  DX_REQUIRE(mode == 0 || mode == 1, DX_ERR_UNSUPPORTED, "dx_prosody_control: pitch transform %d not implemented", mode);
  hipLaunchKernelGGL(prosody_control_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, energy, pitch, energy_factors, pitch_factors,
                     durations_int, speaker_ids, spk_pitch_mean, spk_pitch_std, mode, L);
  DX_LAUNCH_CHECK();
  return DX_OK;
}
