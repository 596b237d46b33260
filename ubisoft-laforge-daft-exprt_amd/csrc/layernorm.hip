// K5 -- LayerNorm over the channel axis of channel-last rows, fused with its neighbours:
//
//   s = dropout_pre(x) + residual           (attention / FF output path, model.py:189-191, 226-228)
//   y = ((s - mean) * rstd) * gamma + beta  (eps 1e-5, biased variance)
//   y = dropout_post(y)                     (prenet / predictor: conv -> ReLU -> LN -> Dropout, model.py:341-363, 528-543)
//   y = film_gamma[b] * y + film_beta[b]    (FiLM, model.py:230-235, 558-564)
//   y = 0 where n >= lengths[b]             (masked_fill, model.py:259, 262, 414, 566)
//
// One wave per row; a lane owns C/64 channels in 16-byte groups (coalesced float4 / bf16x4 accesses);
// statistics are fp32 wave reductions.  HBM-bound: reads x (+residual) once, writes y (+s, mean, rstd for
// the backward pass) once.  The backward kernel walks a contiguous slab of rows of ONE utterance per
// workgroup so that the per-channel reductions (dgamma, dbeta, dFiLM) stay in registers and cost one
// atomic per channel per workgroup.
#include <stdlib.h>

#include "dx_common.h"

namespace {

struct LNArgs {
  const void* x; const float* res; const float* gamma; const float* beta;
  const float* film; long ldf;        // film row b: [gamma(C) | beta(C)], rows ldf apart
  const int64_t* lengths; const int64_t* skip;
  void* y; void* y_lp; float* s_out; float* mean; float* rstd;   // y_lp: optional bf16 copy of y (MFMA operand of the next GEMM)
  int N; long rows;
  float p_pre, p_post; uint64_t seed_pre, seed_post;
  const DxStepScalars* step;   // NULL, or the device-side step block whose salt is added to both seeds
};

template <int C>
struct Lay {
  static constexpr int EPL = C / 64;                 // elements per lane
  static constexpr int V = EPL >= 4 ? 4 : EPL;       // vector width
  static constexpr int NV = EPL / V;                 // vectors per lane
  __device__ static __forceinline__ int col(int lane, int v) { return v * 64 * V + lane * V; }
};

template <typename T, int V>
__device__ __forceinline__ void load_vec(const T* p, float* out) {
#pragma unroll
  for (int i = 0; i < V; ++i) out[i] = (float)p[i];
}
template <>
__device__ __forceinline__ void load_vec<float, 4>(const float* p, float* out) {
  f32x4 v = *reinterpret_cast<const f32x4*>(p);
  out[0] = v[0]; out[1] = v[1]; out[2] = v[2]; out[3] = v[3];
}
template <>
__device__ __forceinline__ void load_vec<bf16_t, 4>(const bf16_t* p, float* out) {
  bf16x4 v = *reinterpret_cast<const bf16x4*>(p);
  out[0] = (float)v[0]; out[1] = (float)v[1]; out[2] = (float)v[2]; out[3] = (float)v[3];
}
template <typename T, int V>
__device__ __forceinline__ void store_vec(T* p, const float* in) {
#pragma unroll
  for (int i = 0; i < V; ++i) p[i] = (T)in[i];
}
template <>
__device__ __forceinline__ void store_vec<float, 4>(float* p, const float* in) {
  f32x4 v = {in[0], in[1], in[2], in[3]};
  *reinterpret_cast<f32x4*>(p) = v;
}
template <>
__device__ __forceinline__ void store_vec<bf16_t, 4>(bf16_t* p, const float* in) {
  bf16x4 v = {(bf16_t)in[0], (bf16_t)in[1], (bf16_t)in[2], (bf16_t)in[3]};
  *reinterpret_cast<bf16x4*>(p) = v;
}

// Dropout in front of / behind a LayerNorm: dx_keep_elem (dx_common.h) -- one hash per 4 consecutive channels.  One murmur hash per
// element (3 quarter-rate 32-bit multiplies) was 40 % of ln_fwd<1024> and a fifth of its backward.

template <typename TI, typename TO, int C>
__global__ __launch_bounds__(256) void ln_fwd_kernel(LNArgs a) {
  typedef Lay<C> L;
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= a.rows) return;
  const int b = (int)(row / a.N), n = (int)(row - (long)b * a.N);
  const TI* x = reinterpret_cast<const TI*>(a.x) + row * C;
  const uint32_t key_pre = dx_key32(dx_seed_eff(a.seed_pre, a.step), 0), key_post = dx_key32(dx_seed_eff(a.seed_post, a.step), 1);
  if (a.skip && n >= (int)a.skip[b] + 2) {   // rows past length + conv halo never reach a valid output: zeros, no reads
    if (n >= dx_fill_end((int)a.skip[b], a.N)) return;   // ... and past the fill end nobody reads them either (dx_common.h): unwritten
    TO* y0 = reinterpret_cast<TO*>(a.y) + row * C;
    float z[L::V];
#pragma unroll
    for (int i = 0; i < L::V; ++i) z[i] = 0.f;
#pragma unroll
    for (int k = 0; k < L::NV; ++k) {
      store_vec<TO, L::V>(y0 + L::col(lane, k), z);
      if (a.y_lp) store_vec<bf16_t, L::V>(reinterpret_cast<bf16_t*>(a.y_lp) + row * C + L::col(lane, k), z);
      if (a.s_out) store_vec<float, L::V>(a.s_out + row * C + L::col(lane, k), z);
    }
    if (a.mean && lane == 0) { a.mean[row] = 0.f; a.rstd[row] = 0.f; }
    return;
  }
  float v[L::EPL];
  const uint32_t th_pre = dx_drop_th8(a.p_pre);
  const float sc_pre = th_pre ? dx_drop_inv_keep8(th_pre) : 1.f;
#pragma unroll
  for (int k = 0; k < L::NV; ++k) {
    const int c0 = L::col(lane, k);
    load_vec<TI, L::V>(x + c0, v + k * L::V);
    if (th_pre) {
#pragma unroll
      for (int i = 0; i < L::V; ++i)
        v[k * L::V + i] = dx_keep_elem(key_pre, (uint32_t)row * C + c0 + i, th_pre) ? v[k * L::V + i] * sc_pre : 0.f;
    }
    if (a.res) {
      float r[L::V];
      load_vec<float, L::V>(a.res + row * C + c0, r);
#pragma unroll
      for (int i = 0; i < L::V; ++i) v[k * L::V + i] += r[i];
    }
    if (a.s_out) store_vec<float, L::V>(a.s_out + row * C + c0, v + k * L::V);
  }
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < L::EPL; ++i) sum += v[i];
  const float mean = dx_wave_sum(sum) * (1.f / C);
  float sq = 0.f;
#pragma unroll
  for (int i = 0; i < L::EPL; ++i) { const float d = v[i] - mean; sq += d * d; }
  const float rstd = rsqrtf(dx_wave_sum(sq) * (1.f / C) + 1e-5f);
  if (a.mean && lane == 0) { a.mean[row] = mean; a.rstd[row] = rstd; }
  const bool pad = a.lengths && n >= (int)a.lengths[b];
  const uint32_t th_post = dx_drop_th8(a.p_post);
  const float sc_post = th_post ? dx_drop_inv_keep8(th_post) : 1.f;
  TO* y = reinterpret_cast<TO*>(a.y) + row * C;
#pragma unroll
  for (int k = 0; k < L::NV; ++k) {
    const int c0 = L::col(lane, k);
    float g[L::V], bt[L::V], o[L::V];
    load_vec<float, L::V>(a.gamma + c0, g);
    load_vec<float, L::V>(a.beta + c0, bt);
#pragma unroll
    for (int i = 0; i < L::V; ++i) {
      float t = (v[k * L::V + i] - mean) * rstd * g[i] + bt[i];
      if (th_post) t = dx_keep_elem(key_post, (uint32_t)row * C + c0 + i, th_post) ? t * sc_post : 0.f;
      o[i] = t;
    }
    if (a.film) {
      float fg[L::V], fb[L::V];
      load_vec<float, L::V>(a.film + (long)b * a.ldf + c0, fg);
      load_vec<float, L::V>(a.film + (long)b * a.ldf + C + c0, fb);
#pragma unroll
      for (int i = 0; i < L::V; ++i) o[i] = fg[i] * o[i] + fb[i];
    }
    if (pad) {
#pragma unroll
      for (int i = 0; i < L::V; ++i) o[i] = 0.f;
    }
    store_vec<TO, L::V>(y + c0, o);
    if (a.y_lp) store_vec<bf16_t, L::V>(reinterpret_cast<bf16_t*>(a.y_lp) + row * C + c0, o);
  }
}

struct LNBwdArgs {
  const void* dy;            // grad wrt the kernel's output y, dtype TG, (rows, C)
  const void* s;             // the normalised tensor's input: s_out of the forward (fp32) or the raw x (TI)
  const float* mean; const float* rstd; const float* gamma; const float* beta;
  const float* film; long ldf; const int64_t* lengths; const int64_t* skip;
  void* ds;                  // out: grad wrt s (dtype TD) -- also the residual gradient
  void* dx_pre;              // out (nullable): grad wrt x before dropout_pre (dtype TD); only when p_pre > 0
  void* dx_pre_lp;           // out (nullable): the same gradient as bf16 (MFMA operand of the following dgrad / wgrad)
  float* dgamma; float* dbeta;          // (C) accumulated with atomics
  float* dfilm; long lddf;              // (B, 2C) accumulated with atomics (nullable)
  int N; int B; int rows_per_block;
  float p_pre, p_post; uint64_t seed_pre, seed_post;
  int relu_input;            // s = relu(conv): the returned ds is additionally gated by (s > 0)
  float* ws;                 // optional (B * chunks, 4, C) partial sums -> finished by ln_bwd_finish_kernel (no atomics)
  const DxStepScalars* step; // NULL, or the device-side step block whose salt is added to both seeds
};

// raw (unconverted) 4-element row segment: the prefetched next row stays in its storage type
template <typename T, int V> struct RawVec { typedef T type __attribute__((ext_vector_type(V))); };

// grid = (ceil(N / rows_per_block), B); 4 waves per block, wave w handles rows w, w+4, ... of the slab.
// FILM: the forward applied y = film_gamma * LN + film_beta (per-utterance gradients dfilm).  Register budget: the
// next row of dy / s is prefetched in its STORAGE type (half the registers of fp32 for the bf16 tensors), and for
// C = 1024 gamma / beta are re-read per row (L1-resident, 8 KB) instead of living in 32 registers -- the first version
// held everything in fp32 registers: 292 VGPRs, one wave per SIMD, 1.5 TB/s.
template <typename TI, typename TG, typename TD, int C, bool FILM>
// (fp32 rows at C = 1024 -- the exact-parity mode -- prefetch twice the registers: 2 waves per SIMD instead of 3, no scratch)
__global__ __launch_bounds__(256, C >= 1024 ? (FILM ? 1 : ((sizeof(TI) == 4 || sizeof(TG) == 4) ? 2 : 3)) : 4) void ln_bwd_kernel(LNBwdArgs a) {
  typedef Lay<C> L;
  constexpr bool REG_PARAMS = C <= 256;
  constexpr int NRED = FILM ? 4 : 2;
  __shared__ float red[4][NRED][C];  // [wave][dgamma, dbeta (, dfilm_g, dfilm_b)][C]
  // C = 1024: gamma / beta live in LDS, not in 32 registers and not re-read from global per row: vector memory returns in order, so a
  // global re-read issued behind fetch_row(n + 4) or behind the row's stores waits for them; LDS reads count separately (lgkmcnt).
  // (Measured: 55-57 vs 56-60 us alone, ~-0.01 ms per step: the compiler had already moved most of those re-reads up.)
  __shared__ float par_s[REG_PARAMS ? 1 : 2 * C];
  if (!REG_PARAMS) {
    for (int i = threadIdx.x; i < 2 * C; i += 256) par_s[i] = i < C ? a.gamma[i] : a.beta[i - C];
    __syncthreads();
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.y;
  const int n_begin = blockIdx.x * a.rows_per_block;
  const int n_end = min(a.N, n_begin + a.rows_per_block);
  const int len = a.lengths ? (int)a.lengths[b] : a.N;
  const uint32_t th_pre = dx_drop_th8(a.p_pre), th_post = dx_drop_th8(a.p_post);
  const uint32_t key_pre = dx_key32(dx_seed_eff(a.seed_pre, a.step), 0), key_post = dx_key32(dx_seed_eff(a.seed_post, a.step), 1);
  const float sc_pre = th_pre ? dx_drop_inv_keep8(th_pre) : 1.f;
  const float sc_post = th_post ? dx_drop_inv_keep8(th_post) : 1.f;      // (quantised with the threshold: see dx_keep_elem)
  const int nskip = a.skip ? (int)a.skip[b] + 2 : a.N;
  const int nfill = a.skip ? dx_fill_end((int)a.skip[b], a.N) : a.N;

  constexpr int NP = REG_PARAMS ? L::EPL : 1, NF = FILM ? L::EPL : 1;
  float gam[NP], bet[NP], fg[NF];
  float acc_g[L::EPL], acc_b[L::EPL], acc_fg[NF], acc_fb[NF];
  if (REG_PARAMS) {
#pragma unroll
    for (int k = 0; k < L::NV; ++k) {
      load_vec<float, L::V>(a.gamma + L::col(lane, k), gam + (REG_PARAMS ? k * L::V : 0));
      load_vec<float, L::V>(a.beta + L::col(lane, k), bet + (REG_PARAMS ? k * L::V : 0));
    }
  }
  if (FILM) {
#pragma unroll
    for (int k = 0; k < L::NV; ++k) load_vec<float, L::V>(a.film + (long)b * a.ldf + L::col(lane, k), fg + (FILM ? k * L::V : 0));
  }
#pragma unroll
  for (int i = 0; i < L::EPL; ++i) { acc_g[i] = 0.f; acc_b[i] = 0.f; }
#pragma unroll
  for (int i = 0; i < NF; ++i) { acc_fg[i] = 0.f; acc_fb[i] = 0.f; }

  typename RawVec<TG, L::V>::type g_nx[L::NV];
  typename RawVec<TI, L::V>::type x_nx[L::NV];
  float mean_nx = 0.f, rstd_nx = 0.f;
  auto fetch_row = [&](int n) {
    const long row = (long)b * a.N + n;
    const TG* dy = reinterpret_cast<const TG*>(a.dy) + row * C;
    const TI* sp = reinterpret_cast<const TI*>(a.s) + row * C;
#pragma unroll
    for (int k = 0; k < L::NV; ++k) {
      g_nx[k] = *reinterpret_cast<const typename RawVec<TG, L::V>::type*>(dy + L::col(lane, k));
      x_nx[k] = *reinterpret_cast<const typename RawVec<TI, L::V>::type*>(sp + L::col(lane, k));
    }
    mean_nx = a.mean[row];
    rstd_nx = a.rstd[row];
  };
  if (n_begin + wave < n_end && n_begin + wave < nskip) fetch_row(n_begin + wave);

  for (int n = n_begin + wave; n < n_end; n += 4) {
    const long row = (long)b * a.N + n;
    if (n >= nskip) {   // gradient rows past length + halo are exactly zero
      if (n >= nfill) break;   // (past the fill end nobody reads them, dx_common.h: unwritten; rows only grow from here)
      TD* ds0 = reinterpret_cast<TD*>(a.ds) + row * C;
      TD* dxp0 = a.dx_pre ? reinterpret_cast<TD*>(a.dx_pre) + row * C : nullptr;
      float z[L::V];
#pragma unroll
      for (int i = 0; i < L::V; ++i) z[i] = 0.f;
#pragma unroll
      for (int k = 0; k < L::NV; ++k) {
        store_vec<TD, L::V>(ds0 + L::col(lane, k), z);
        if (dxp0) store_vec<TD, L::V>(dxp0 + L::col(lane, k), z);
        if (a.dx_pre_lp) store_vec<bf16_t, L::V>(reinterpret_cast<bf16_t*>(a.dx_pre_lp) + row * C + L::col(lane, k), z);
      }
      continue;
    }
    // this row's operands were fetched one iteration ago; issue the loads of the wave's next row before computing
    float g[L::EPL], xh[L::EPL];
#pragma unroll
    for (int k = 0; k < L::NV; ++k)
#pragma unroll
      for (int i = 0; i < L::V; ++i) { g[k * L::V + i] = (float)g_nx[k][i]; xh[k * L::V + i] = (float)x_nx[k][i]; }
    const float mean = mean_nx, rstd = rstd_nx;
    if (n + 4 < n_end && n + 4 < nskip) fetch_row(n + 4);
    uint32_t pos = 0;   // bit j: s[j] > 0 (ReLU gate), EPL <= 16
    const bool pad = n >= len;
    float m1 = 0.f, m2 = 0.f;
#pragma unroll
    for (int k = 0; k < L::NV; ++k) {
      float gk[L::V], bk[L::V];
      if (!REG_PARAMS) {
#pragma unroll
        for (int i = 0; i < L::V; ++i) { gk[i] = par_s[L::col(lane, k) + i]; bk[i] = par_s[(REG_PARAMS ? 0 : C) + L::col(lane, k) + i]; }
      }
#pragma unroll
      for (int i = 0; i < L::V; ++i) {
        const int j = k * L::V + i;
        const int c = L::col(lane, k) + i;
        const float gmj = REG_PARAMS ? gam[REG_PARAMS ? j : 0] : gk[i], btj = REG_PARAMS ? bet[REG_PARAMS ? j : 0] : bk[i];
        float gy = pad ? 0.f : g[j];
        pos |= (xh[j] > 0.f ? 1u : 0u) << j;
        xh[j] = (xh[j] - mean) * rstd;
        float keep_post = 1.f;
        if (th_post) keep_post = dx_keep_elem(key_post, (uint32_t)row * C + c, th_post) ? sc_post : 0.f;
        if (FILM) {                                         // y = fg * (ln * keep) + fb
          const float ln = xh[j] * gmj + btj;               // LayerNorm output before dropout_post / FiLM
          acc_fg[FILM ? j : 0] += gy * ln * keep_post;
          acc_fb[FILM ? j : 0] += gy;
          gy *= fg[FILM ? j : 0];
        }
        gy *= keep_post;                                    // grad wrt ln
        acc_g[j] += gy * xh[j];
        acc_b[j] += gy;
        const float gx = gy * gmj;
        g[j] = gx;
        m1 += gx;
        m2 += gx * xh[j];
      }
    }
    m1 = dx_wave_sum(m1) * (1.f / C);
    m2 = dx_wave_sum(m2) * (1.f / C);
    TD* ds = reinterpret_cast<TD*>(a.ds) + row * C;
    TD* dxp = a.dx_pre ? reinterpret_cast<TD*>(a.dx_pre) + row * C : nullptr;
#pragma unroll
    for (int k = 0; k < L::NV; ++k) {
      const int c0 = L::col(lane, k);
      float o[L::V], o2[L::V];
#pragma unroll
      for (int i = 0; i < L::V; ++i) {
        const int j = k * L::V + i;
        o[i] = rstd * (g[j] - m1 - xh[j] * m2);
        if (a.relu_input && !((pos >> j) & 1u)) o[i] = 0.f;
        o2[i] = o[i];
        if (th_pre) o2[i] = dx_keep_elem(key_pre, (uint32_t)row * C + c0 + i, th_pre) ? o[i] * sc_pre : 0.f;
      }
      store_vec<TD, L::V>(ds + c0, o);
      if (dxp) store_vec<TD, L::V>(dxp + c0, o2);
      if (a.dx_pre_lp) store_vec<bf16_t, L::V>(reinterpret_cast<bf16_t*>(a.dx_pre_lp) + row * C + c0, o2);
    }
  }
  // ---- reduce the per-channel partials over the 4 waves, one atomic per channel per block
#pragma unroll
  for (int k = 0; k < L::NV; ++k) {
#pragma unroll
    for (int i = 0; i < L::V; ++i) {
      const int j = k * L::V + i, c = L::col(lane, k) + i;
      red[wave][0][c] = acc_g[j]; red[wave][1][c] = acc_b[j];
      if (FILM) { red[wave][FILM ? 2 : 0][c] = acc_fg[FILM ? j : 0]; red[wave][FILM ? 3 : 0][c] = acc_fb[FILM ? j : 0]; }
    }
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < NRED * C; idx += 256) {
    const int which = idx / C, c = idx - which * C;
    const float t = red[0][which][c] + red[1][which][c] + red[2][which][c] + red[3][which][c];
    if (a.ws) { a.ws[((long)(b * gridDim.x + blockIdx.x) * 4 + which) * C + c] = t; continue; }
    if (which == 0) atomicAdd(a.dgamma + c, t);
    else if (which == 1) atomicAdd(a.dbeta + c, t);
    else atomicAdd(a.dfilm + (long)b * a.lddf + (which == 3 ? C : 0) + c, t);
  }
  if (a.ws && !FILM) {   // the finish kernel reads all four slots
    for (int idx = threadIdx.x; idx < 2 * C; idx += 256) a.ws[((long)(b * gridDim.x + blockIdx.x) * 4 + 2) * C + idx] = 0.f;
  }
}
// Finishes the per-channel reductions of ln_bwd_kernel from its per-workgroup partial sums (deterministic, no atomics).
// blocks [0, nA): dgamma / dbeta over all (utterance, chunk) partials; blocks [nA, ...): dfilm[b] over the chunks of b.
__global__ __launch_bounds__(256) void ln_bwd_finish_kernel(const float* __restrict__ ws, float* dgamma, float* dbeta, float* dfilm,
                                                            long lddf, int B, int chunks, int C, int nA) {
  __shared__ float red[4][64];
  if ((int)blockIdx.x < nA) {
    const int col = blockIdx.x * 64 + (threadIdx.x & 63), grp = threadIdx.x >> 6;   // col in [0, 2C): gamma | beta
    float acc = 0.f;
    if (col < 2 * C) {
      const int which = col / C, c = col - which * C;
      const int total = B * chunks;
      for (int p = grp; p < total; p += 4) acc += ws[((long)p * 4 + which) * C + c];
    }
    red[grp][threadIdx.x & 63] = acc;
    __syncthreads();
    if (grp == 0 && col < 2 * C) {
      const float t = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
      if (col < C) dgamma[col] += t; else dbeta[col - C] += t;
    }
  } else if (dfilm) {
    const int per_b = (2 * C + 255) / 256;
    const int r = blockIdx.x - nA, b = r / per_b, col = (r % per_b) * 256 + threadIdx.x;
    if (col < 2 * C) {
      const int which = 2 + col / C, c = col % C;
      float acc = 0.f;
      for (int k = 0; k < chunks; ++k) acc += ws[((long)(b * chunks + k) * 4 + which) * C + c];
      dfilm[(long)b * lddf + col] += acc;
    }
  }
}

template <typename TI, typename TO>
int launch_fwd(const LNArgs& a, int C, hipStream_t s) {
  dim3 grid((unsigned)((a.rows + 3) / 4)), block(256);
  switch (C) {
    case 128: hipLaunchKernelGGL((ln_fwd_kernel<TI, TO, 128>), grid, block, 0, s, a); break;
    case 256: hipLaunchKernelGGL((ln_fwd_kernel<TI, TO, 256>), grid, block, 0, s, a); break;
    case 512: hipLaunchKernelGGL((ln_fwd_kernel<TI, TO, 512>), grid, block, 0, s, a); break;     // (non-default conv widths: same template)
    case 1024: hipLaunchKernelGGL((ln_fwd_kernel<TI, TO, 1024>), grid, block, 0, s, a); break;
    default: dx_set_error("dx_layernorm_fwd: C=%d unsupported (128, 256, 512, 1024)", C); return DX_ERR_UNSUPPORTED;
  }
  DX_LAUNCH_CHECK();
  return DX_OK;
}
template <typename TI, typename TG, typename TD>
int launch_bwd(const LNBwdArgs& a, int C, hipStream_t s) {
  dim3 grid(dx_cdiv(a.N, a.rows_per_block), a.B), block(256);
  const bool film = a.film != nullptr;
  switch (C) {
    case 128:
      if (film) hipLaunchKernelGGL((ln_bwd_kernel<TI, TG, TD, 128, true>), grid, block, 0, s, a);
      else hipLaunchKernelGGL((ln_bwd_kernel<TI, TG, TD, 128, false>), grid, block, 0, s, a);
      break;
    case 256:
      if (film) hipLaunchKernelGGL((ln_bwd_kernel<TI, TG, TD, 256, true>), grid, block, 0, s, a);
      else hipLaunchKernelGGL((ln_bwd_kernel<TI, TG, TD, 256, false>), grid, block, 0, s, a);
      break;
    case 512:
      if (film) hipLaunchKernelGGL((ln_bwd_kernel<TI, TG, TD, 512, true>), grid, block, 0, s, a);
      else hipLaunchKernelGGL((ln_bwd_kernel<TI, TG, TD, 512, false>), grid, block, 0, s, a);
      break;
    case 1024:
      if (film) hipLaunchKernelGGL((ln_bwd_kernel<TI, TG, TD, 1024, true>), grid, block, 0, s, a);   // not on the model's path
      else hipLaunchKernelGGL((ln_bwd_kernel<TI, TG, TD, 1024, false>), grid, block, 0, s, a);
      break;
    default: dx_set_error("dx_layernorm_bwd: C=%d unsupported (128, 256, 512, 1024)", C); return DX_ERR_UNSUPPORTED;
  }
  if (a.ws) {
    const int chunks = dx_cdiv(a.N, a.rows_per_block), nA = dx_cdiv(2 * C, 64);
    const int nB = a.dfilm ? a.B * dx_cdiv(2 * C, 256) : 0;
    hipLaunchKernelGGL(ln_bwd_finish_kernel, dim3(nA + nB), dim3(256), 0, s, a.ws, a.dgamma, a.dbeta, a.dfilm, a.lddf, a.B, chunks, C, nA);
  }
  DX_LAUNCH_CHECK();
  return DX_OK;
}

}  // namespace

extern "C" int dx_layernorm_fwd(const void* x, int x_dtype, const float* residual, const float* gamma, const float* beta,
                                const float* film, long ldf, const int64_t* lengths, const int64_t* skip_lengths, void* y,
                                int y_dtype, void* y_lp, float* s_out, float* mean, float* rstd, int B, int N, int C, float p_pre, uint64_t seed_pre,
                                float p_post, uint64_t seed_post, const DxStepScalars* step, void* stream) {
  DX_REQUIRE(x && gamma && beta && y, DX_ERR_ARG, "dx_layernorm_fwd: null pointer");
  DX_REQUIRE(B > 0 && N > 0, DX_ERR_SHAPE, "dx_layernorm_fwd: empty shape");
  DX_REQUIRE((mean == nullptr) == (rstd == nullptr), DX_ERR_ARG, "dx_layernorm_fwd: mean and rstd come together");
  DX_REQUIRE(p_pre >= 0.f && p_pre < 1.f && p_post >= 0.f && p_post < 1.f, DX_ERR_ARG, "dx_layernorm_fwd: dropout p out of [0,1)");
  LNArgs a{x, residual, gamma, beta, film, ldf, lengths, skip_lengths, y, y_lp, s_out, mean, rstd, N, (long)B * N, p_pre, p_post, seed_pre, seed_post, step};
  hipStream_t s = (hipStream_t)stream;
  if (x_dtype == DX_F32 && y_dtype == DX_F32) return launch_fwd<float, float>(a, C, s);
  if (x_dtype == DX_BF16 && y_dtype == DX_BF16) return launch_fwd<bf16_t, bf16_t>(a, C, s);
  if (x_dtype == DX_BF16 && y_dtype == DX_F32) return launch_fwd<bf16_t, float>(a, C, s);
  if (x_dtype == DX_F32 && y_dtype == DX_BF16) return launch_fwd<float, bf16_t>(a, C, s);
  dx_set_error("dx_layernorm_fwd: unsupported dtypes x=%d y=%d", x_dtype, y_dtype);
  return DX_ERR_DTYPE;
}

extern "C" int dx_layernorm_bwd(const void* dy, int dy_dtype, const void* s_in, int s_dtype, const float* mean,
                                const float* rstd, const float* gamma, const float* beta, const float* film, long ldf,
                                const int64_t* lengths, const int64_t* skip_lengths, void* ds, void* dx_pre, void* dx_pre_lp,
                                int d_dtype, float* dgamma, float* dbeta, float* dfilm, long lddf, int B, int N, int C, float p_pre, uint64_t seed_pre,
                                float p_post, uint64_t seed_post, int relu_input, float* ws, const DxStepScalars* step, void* stream) {
  DX_REQUIRE(dy && s_in && mean && rstd && gamma && beta && ds && dgamma && dbeta, DX_ERR_ARG, "dx_layernorm_bwd: null pointer");
  DX_REQUIRE(B > 0 && N > 0, DX_ERR_SHAPE, "dx_layernorm_bwd: empty shape");
  DX_REQUIRE((film == nullptr) == (dfilm == nullptr), DX_ERR_ARG, "dx_layernorm_bwd: film and dfilm come together");
  // enough workgroups to fill 256 CUs, few enough that the per-channel atomics stay cheap
  const int maxblk = 768;   // (dx_layernorm_bwd_ws_floats sizes the two-stage workspace for 768; 384 / 256 / 192 measured +0.01 / +0.15 / +0.18 ms per step; round 5: 1536 / 3072 for C <= 256 only: +0.06 / +0.06 ms)
  int rpb = 32;
  while (rpb < 1024 && (long)dx_cdiv(N, rpb) * B > maxblk) rpb *= 2;
  LNBwdArgs a{dy, s_in, mean, rstd, gamma, beta, film, ldf, lengths, skip_lengths, ds, dx_pre, dx_pre_lp, dgamma, dbeta, dfilm, lddf, N, B, rpb,
              p_pre, p_post, seed_pre, seed_post, relu_input, ws, step};
  hipStream_t s = (hipStream_t)stream;
  if (s_dtype == DX_F32 && dy_dtype == DX_F32 && d_dtype == DX_F32) return launch_bwd<float, float, float>(a, C, s);
  if (s_dtype == DX_BF16 && dy_dtype == DX_BF16 && d_dtype == DX_BF16) return launch_bwd<bf16_t, bf16_t, bf16_t>(a, C, s);
  if (s_dtype == DX_BF16 && dy_dtype == DX_F32 && d_dtype == DX_BF16) return launch_bwd<bf16_t, float, bf16_t>(a, C, s);
  if (s_dtype == DX_F32 && dy_dtype == DX_BF16 && d_dtype == DX_F32) return launch_bwd<float, bf16_t, float>(a, C, s);
  if (s_dtype == DX_F32 && dy_dtype == DX_F32 && d_dtype == DX_BF16) return launch_bwd<float, float, bf16_t>(a, C, s);
  dx_set_error("dx_layernorm_bwd: unsupported dtypes s=%d dy=%d d=%d", s_dtype, dy_dtype, d_dtype);
  return DX_ERR_DTYPE;
}

extern "C" long dx_layernorm_bwd_ws_floats(int B, int N, int C) {
  int rpb = 32;
  while (rpb < 1024 && (long)dx_cdiv(N, rpb) * B > 768) rpb *= 2;
  return (long)B * dx_cdiv(N, rpb) * 4 * C;
}
