// K4 -- multi-head self-attention with key-padding mask, flash-style (never materialises (B,H,N,N)).
//
// Reference math (nn.MultiheadAttention as called at model.py:182-186):
//   S = (q / sqrt(d_h)) k^T, S[:, pad keys] = -inf, P = softmax(S), P = dropout(P), o = P v
// Layout: qkv (B, N, 3E) straight out of the in-projection GEMM (q | k | v, heads contiguous d_h wide),
// o (B, N, E).  d_h in {16, 64}.
//
// MFMA formulation (32x32 tiles, wave64).  A workgroup = 4 waves; each wave owns 32 query rows (forward,
// dQ kernel) or 32 key rows (dK/dV kernel) and walks the other axis in LDS-staged tiles of 64 rows.
//   forward :  S^T = K Q^T  (lane = one query, 16 keys in registers -> softmax statistics are per-lane
//              scalars + one lane^32 exchange);  O^T += V^T P^T  (P^T feeds the B operand straight from
//              registers; the running-max rescale is a per-lane scalar multiply, no shuffles).
//   dQ      :  S^T, dP^T = V dO^T the same way;  dQ^T += K^T dS^T.
//   dK/dV   :  S = Q K^T, dP = dO V^T (lane = one key);  dV^T += dO^T P_drop,  dK^T += Q^T dS.
// Operands that must be read "k-major" from a row-major LDS tile (V^T, K^T, dO^T, Q^T) use the gfx950
// LDS transpose read ds_read_b64_tr_b16 (bf16) -- two reads give a lane its 8 k-values -- or eight
// ds_read_b32 in the exact-fp32 mode.
// Dropout on P: counter-based hash of (seed, (b, h, query, key)), regenerated in the backward kernels.
#include "dx_common.h"

namespace {


struct AttnArgs {
  const void* qkv;         // (B, N, 3E)
  void* o;                 // (B, N, E)           fwd out / bwd in
  float* lse;              // (B, H, N)           fwd out / bwd in
  const void* d_o;         // (B, N, E)           bwd in
  const float* delta;      // (B, H, N)           bwd in
  void* dqkv;              // (B, N, 3E)          bwd out
  const int64_t* lengths;
  int N, H, E;
  float scale, p_drop;
  uint64_t seed;
};

template <typename TC> struct APad;
template <> struct APad<bf16_t> { static constexpr int value = 8; };
template <> struct APad<float> { static constexpr int value = 4; };

// 2^x: the softmax scale and log2(e) are folded into one FMA in front of it (p = 2^(s*c - m), c = scale*log2 e), so a
// score costs FMA + v_exp_f32 instead of mul, sub, mul, v_exp.  The exact-fp32 mode uses the accurate exp2f.
template <typename TC> __device__ __forceinline__ float fast_exp2(float x);
template <> __device__ __forceinline__ float fast_exp2<bf16_t>(float x) { return __builtin_amdgcn_exp2f(x); }
template <> __device__ __forceinline__ float fast_exp2<float>(float x) { return exp2f(x); }
constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;

// stage `rows` rows x DH columns of a (.., ld_g)-strided global matrix into an LDS tile (zero beyond n_lim)
template <typename TC, int DH, int LD>
__device__ __forceinline__ void stage_tile(TC* dst, const TC* src, long ld_g, int row0, int rows, int n_lim, int tid) {
  constexpr int CPR = DH / 8;  // 8-element chunks per row
  for (int c = tid; c < rows * CPR; c += 256) {
    const int r = c / CPR, kc = (c - r * CPR) * 8;
    typename Vec8<TC>::type v = zero8<TC>();
    if (row0 + r < n_lim) v = *reinterpret_cast<const typename Vec8<TC>::type*>(src + (long)(row0 + r) * ld_g + kc);
    *reinterpret_cast<typename Vec8<TC>::type*>(dst + r * LD + kc) = v;
  }
}

// the same tile in two halves: global -> registers (issued one stage ahead, so the load latency hides under the MFMA /
// softmax work of the current stage) and registers -> LDS
template <typename TC, int DH, int ROWS> struct TileRegs {
  static constexpr int CPR = DH / 8, PT = (ROWS * CPR + 255) / 256;
  typename Vec8<TC>::type v[PT];
  __device__ __forceinline__ void fetch(const TC* src, long ld_g, int row0, int n_lim, int tid) {
#pragma unroll
    for (int t = 0; t < PT; ++t) {
      const int c = tid + t * 256, r = c / CPR, kc = (c - r * CPR) * 8;
      v[t] = zero8<TC>();
      if (c < ROWS * CPR && row0 + r < n_lim) v[t] = *reinterpret_cast<const typename Vec8<TC>::type*>(src + (long)(row0 + r) * ld_g + kc);
    }
  }
  template <int LD> __device__ __forceinline__ void commit(TC* dst, int tid) const {
#pragma unroll
    for (int t = 0; t < PT; ++t) {
      const int c = tid + t * 256, r = c / CPR, kc = (c - r * CPR) * 8;
      if (c < ROWS * CPR) *reinterpret_cast<typename Vec8<TC>::type*>(dst + r * LD + kc) = v[t];
    }
  }
};

__device__ __forceinline__ uint32_t athresh(float p) { return p <= 0.f ? 0u : (uint32_t)(p * 4294967296.0); }

// rows of the streamed axis per LDS stage: every stage exposes one global-load latency, so small heads take big stages
template <int DH> struct Stage { static constexpr int KT = DH <= 16 ? 256 : 128; };

// =============================================================================== forward
template <typename TC, int DH>
__global__ __launch_bounds__(256, 2) void attn_fwd_kernel(AttnArgs a) {
  constexpr int KT = Stage<DH>::KT;
  constexpr int LD = DH + APad<TC>::value, KS = DH / 16, MT = (DH + 31) / 32, WRAP = DH >= 32 ? 32 : 16;
  typedef typename Vec8<TC>::type frag_t;
  __shared__ __attribute__((aligned(16))) TC Ks[KT * LD];
  __shared__ __attribute__((aligned(16))) TC Vs[KT * LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, g = lane >> 5;
  const int b = blockIdx.z, h = blockIdx.y, N = a.N, E = a.E;
  const int len = (int)a.lengths[b];
  const int q = blockIdx.x * 128 + wave * 32 + l31;
  const long ld_g = 3L * E;
  const TC* base = reinterpret_cast<const TC*>(a.qkv) + (long)b * N * ld_g + h * DH;
  TC* O = reinterpret_cast<TC*>(a.o) + (long)b * N * E + h * DH;
  float* lse = a.lse ? a.lse + ((long)b * a.H + h) * N : nullptr;

  if (blockIdx.x * 128 >= len) {  // whole tile of pad queries: their rows are zeroed after the LayerNorm anyway
    if (q < N) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int d = dx_acc_row(r, g);
        if (d < DH) for (int mt = 0; mt < MT; ++mt) O[(long)q * E + mt * 32 + d] = (TC)0.f;
      }
      if (lse && g == 0) lse[q] = 0.f;
    }
    return;
  }

  frag_t qf[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks)
    qf[ks] = q < N ? *reinterpret_cast<const frag_t*>(base + (long)q * ld_g + ks * 16 + g * 8) : zero8<TC>();

  f32x16 oT[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) oT[mt][r] = 0.f;
  float m = -INFINITY, l = 0.f;
  const float c2 = a.scale * LOG2E;
  const uint32_t th = athresh(a.p_drop);
  const float inv_keep = a.p_drop > 0.f ? 1.f / (1.f - a.p_drop) : 1.f;   // applied once, to the output row
  // dropout counter of (q, key) = ctr_lane + key-dependent part that is a scalar + compile-time constant (dx_common.h)
  const uint32_t ctr_lane = ((uint32_t)q * (uint32_t)N + 4u * g) * DX_CTR_MUL + dx_key32(a.seed, (uint32_t)(b * a.H + h));

  constexpr bool AHEAD = sizeof(TC) == 2;   // exact-fp32 mode: twice the registers per tile, loads stay in place
  TileRegs<TC, DH, KT> kreg, vreg;
  if (AHEAD) {
    kreg.fetch(base + E, ld_g, 0, N, tid);
    vreg.fetch(base + 2 * E, ld_g, 0, N, tid);
  }
  for (int kt0 = 0; kt0 < len; kt0 += KT) {
    if (!AHEAD) {
      kreg.fetch(base + E, ld_g, kt0, N, tid);
      vreg.fetch(base + 2 * E, ld_g, kt0, N, tid);
    }
    kreg.template commit<LD>(Ks, tid);
    vreg.template commit<LD>(Vs, tid);
    __syncthreads();
    if (AHEAD && kt0 + KT < len) {
      kreg.fetch(base + E, ld_g, kt0 + KT, N, tid);
      vreg.fetch(base + 2 * E, ld_g, kt0 + KT, N, tid);
    }
#pragma unroll
    for (int sub = 0; sub < KT / 32; ++sub) {
      const int k0 = kt0 + sub * 32;
      if (k0 < len) {
        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          frag_t kf = *reinterpret_cast<const frag_t*>(&Ks[(sub * 32 + l31) * LD + ks * 16 + g * 8]);
          dx_mma(s, kf, qf[ks]);
        }
        float p[16], mx = -INFINITY;
        if (k0 + 32 > len) {   // boundary tile only: pad keys -> -inf
#pragma unroll
          for (int r = 0; r < 16; ++r) s[r] = (k0 + dx_acc_row(r, g) < len) ? s[r] : -INFINITY;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m, mx * c2);           // running max in the scaled log2 domain
        float alpha = 1.f;
        const bool moved = !__all(m_new == m);           // wave-uniform: most tiles after the first few skip the rescale
        if (moved) { alpha = fast_exp2<TC>(m - m_new); m = m_new; }
        float rs = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { p[r] = fast_exp2<TC>(fmaf(s[r], c2, -m)); rs += p[r]; }
        rs += __shfl_xor(rs, 32, 64);
        l = l * alpha + rs;
        if (th) {   // registers r, r + 1 hold the key pair (even, odd): one counter prefix, two fields
          const uint32_t ctr_tile = (uint32_t)k0 * DX_CTR_MUL;
#pragma unroll
          for (int r = 0; r < 16; r += 2) {
            const uint32_t pre = dx_drop_prefix(ctr_lane + (ctr_tile + (uint32_t)((r & 3) + 8 * (r >> 2)) * DX_CTR_MUL));
            p[r] = dx_drop_field(pre, DX_M24_EVEN) >= th ? p[r] : 0.f;
            p[r + 1] = dx_drop_field(pre, DX_M24_ODD) >= th ? p[r + 1] : 0.f;
          }
        }
        frag_t pf[2] = {pack8<TC>(p), pack8<TC>(p + 8)};
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          if (moved) {
#pragma unroll
            for (int r = 0; r < 16; ++r) oT[mt][r] *= alpha;
          }
#pragma unroll
          for (int kstep = 0; kstep < 2; ++kstep) {
            frag_t vf = gather8<TC, WRAP>(Vs + sub * 32 * LD, LD, kstep * 16 + 4 * g, kstep * 16 + 4 * g + 8, mt * 32, lane);
            dx_mma(oT[mt], vf, pf[kstep]);
          }
        }
      }
    }
    __syncthreads();
  }
  if (q < N) {
    const float inv_l = inv_keep / l;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        const int d0 = 8 * r4 + 4 * g;  // rows d0..d0+3 = registers 4*r4 .. 4*r4+3
        if (mt * 32 + d0 < DH) {
#pragma unroll
          for (int t = 0; t < 4; ++t) O[(long)q * E + mt * 32 + d0 + t] = (TC)(oT[mt][r4 * 4 + t] * inv_l);
        }
      }
    }
    if (lse && g == 0) lse[q] = m * LN2 + logf(l);
  }
}

// =============================================================================== delta = rowsum(dO * O)
template <typename TC>
__global__ void attn_delta_kernel(const TC* __restrict__ o, const TC* __restrict__ d_o, float* __restrict__ delta,
                                  int B, int N, int H, int DH) {
  const long total = (long)B * N * H;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int h = i % H; const long bn = i / H; const int n = bn % N; const int b = bn / N;
    const TC* po = o + bn * (long)(H * DH) + h * DH;
    const TC* pd = d_o + bn * (long)(H * DH) + h * DH;
    float acc = 0.f;
    for (int d = 0; d < DH; ++d) acc += (float)po[d] * (float)pd[d];
    delta[((long)b * H + h) * N + n] = acc;
  }
}

// =============================================================================== backward: dQ
template <typename TC, int DH>
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_kernel(AttnArgs a) {
  constexpr int KT = Stage<DH>::KT;
  constexpr int LD = DH + APad<TC>::value, KS = DH / 16, MT = (DH + 31) / 32, WRAP = DH >= 32 ? 32 : 16;
  typedef typename Vec8<TC>::type frag_t;
  __shared__ __attribute__((aligned(16))) TC Ks[KT * LD];
  __shared__ __attribute__((aligned(16))) TC Vs[KT * LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, g = lane >> 5;
  const int b = blockIdx.z, h = blockIdx.y, N = a.N, E = a.E;
  const int len = (int)a.lengths[b];
  const int q = blockIdx.x * 128 + wave * 32 + l31;
  const long ld_g = 3L * E;
  const TC* base = reinterpret_cast<const TC*>(a.qkv) + (long)b * N * ld_g + h * DH;
  const TC* dO = reinterpret_cast<const TC*>(a.d_o) + (long)b * N * E + h * DH;
  TC* dQ = reinterpret_cast<TC*>(a.dqkv) + (long)b * N * ld_g + h * DH;
  const bool q_valid = q < len;

  f32x16 dqT[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) dqT[mt][r] = 0.f;

  if (blockIdx.x * 128 >= len && q < N && g == 0)   // pad-query tiles: delta is never used, keep it finite
    const_cast<float*>(a.delta)[((long)b * a.H + h) * N + q] = 0.f;
  if (blockIdx.x * 128 < len) {
    frag_t qf[KS], dof[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      qf[ks] = q < N ? *reinterpret_cast<const frag_t*>(base + (long)q * ld_g + ks * 16 + g * 8) : zero8<TC>();
      dof[ks] = q < N ? *reinterpret_cast<const frag_t*>(dO + (long)q * E + ks * 16 + g * 8) : zero8<TC>();
    }
    const long stat = ((long)b * a.H + h) * N + q;
    const float lse_q = q < N ? a.lse[stat] : 0.f;
    // delta_q = sum_d dO[q][d] * O[q][d]: the lane pair (g = 0, 1) holds the whole row between them; published for
    // the dK/dV kernel (launched after this one) so that no separate pass over O / dO is needed
    float delta_q = 0.f;
    if (q < N) {
      const TC* Orow = reinterpret_cast<const TC*>(a.o) + ((long)b * N + q) * E + h * DH;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const frag_t of = *reinterpret_cast<const frag_t*>(Orow + ks * 16 + g * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) delta_q += (float)of[e] * (float)dof[ks][e];
      }
    }
    delta_q += __shfl_xor(delta_q, 32, 64);
    if (q < N && g == 0) const_cast<float*>(a.delta)[stat] = delta_q;
    const float c2 = a.scale * LOG2E, lse2 = lse_q * LOG2E;
    const bool tile_q_valid = blockIdx.x * 128 + wave * 32 + 32 <= len;
    const uint32_t th = athresh(a.p_drop);
    const float inv_keep = a.p_drop > 0.f ? 1.f / (1.f - a.p_drop) : 1.f;
    const uint32_t ctr_lane = ((uint32_t)q * (uint32_t)N + 4u * g) * DX_CTR_MUL + dx_key32(a.seed, (uint32_t)(b * a.H + h));

    constexpr bool AHEAD = sizeof(TC) == 2;
    TileRegs<TC, DH, KT> kreg, vreg;
    if (AHEAD) {
      kreg.fetch(base + E, ld_g, 0, N, tid);
      vreg.fetch(base + 2 * E, ld_g, 0, N, tid);
    }
    for (int kt0 = 0; kt0 < len; kt0 += KT) {
      if (!AHEAD) {
        kreg.fetch(base + E, ld_g, kt0, N, tid);
        vreg.fetch(base + 2 * E, ld_g, kt0, N, tid);
      }
      kreg.template commit<LD>(Ks, tid);
      vreg.template commit<LD>(Vs, tid);
      __syncthreads();
      if (AHEAD && kt0 + KT < len) {
        kreg.fetch(base + E, ld_g, kt0 + KT, N, tid);
        vreg.fetch(base + 2 * E, ld_g, kt0 + KT, N, tid);
      }
#pragma unroll
      for (int sub = 0; sub < KT / 32; ++sub) {
        const int k0 = kt0 + sub * 32;
        if (k0 < len) {
          f32x16 s, dp;
#pragma unroll
          for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) {
            frag_t kf = *reinterpret_cast<const frag_t*>(&Ks[(sub * 32 + l31) * LD + ks * 16 + g * 8]);
            frag_t vf = *reinterpret_cast<const frag_t*>(&Vs[(sub * 32 + l31) * LD + ks * 16 + g * 8]);
            dx_mma(s, kf, qf[ks]);
            dx_mma(dp, vf, dof[ks]);
          }
          float ds[16];
          const bool interior = tile_q_valid && k0 + 32 <= len;   // wave-uniform: no masking needed
          const uint32_t ctr_tile = (uint32_t)k0 * DX_CTR_MUL;
#pragma unroll
          for (int r = 0; r < 16; r += 2) {
            const int key = k0 + dx_acc_row(r, g);   // registers r, r+1 hold keys key, key+1 (one counter prefix, see forward)
            // th == 0 keeps everything; no branch around the hash (backward = training)
            const uint32_t pre = dx_drop_prefix(ctr_lane + (ctr_tile + (uint32_t)((r & 3) + 8 * (r >> 2)) * DX_CTR_MUL));
            const float x0 = dx_drop_field(pre, DX_M24_EVEN) >= th ? dp[r] : 0.f;
            const float x1 = dx_drop_field(pre, DX_M24_ODD) >= th ? dp[r + 1] : 0.f;
            float p0 = fast_exp2<TC>(fmaf(s[r], c2, -lse2)), p1 = fast_exp2<TC>(fmaf(s[r + 1], c2, -lse2));
            if (!interior) { p0 = (q_valid && key < len) ? p0 : 0.f; p1 = (q_valid && key + 1 < len) ? p1 : 0.f; }
            ds[r] = p0 * fmaf(x0, inv_keep, -delta_q);
            ds[r + 1] = p1 * fmaf(x1, inv_keep, -delta_q);
          }
          frag_t dsf[2] = {pack8<TC>(ds), pack8<TC>(ds + 8)};
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int kstep = 0; kstep < 2; ++kstep) {
              frag_t kT = gather8<TC, WRAP>(Ks + sub * 32 * LD, LD, kstep * 16 + 4 * g, kstep * 16 + 4 * g + 8, mt * 32, lane);
              dx_mma(dqT[mt], kT, dsf[kstep]);
            }
        }
      }
      __syncthreads();
    }
  }
  if (q < N) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int d = mt * 32 + dx_acc_row(r, g);
        if (d < DH) dQ[(long)q * ld_g + d] = (TC)(dqT[mt][r] * a.scale);
      }
  }
}

// =============================================================================== backward: dK, dV
template <typename TC, int DH>
__global__ __launch_bounds__(256, DH <= 16 ? 3 : 2) void attn_bwd_dkv_kernel(AttnArgs a) {
  constexpr int KT = Stage<DH>::KT;
  constexpr int LD = DH + APad<TC>::value, KS = DH / 16, MT = (DH + 31) / 32, WRAP = DH >= 32 ? 32 : 16;
  typedef typename Vec8<TC>::type frag_t;
  __shared__ __attribute__((aligned(16))) TC Qs[KT * LD];
  __shared__ __attribute__((aligned(16))) TC dOs[KT * LD];
  __shared__ float lse_s[KT], delta_s[KT];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, g = lane >> 5;
  const int b = blockIdx.z, h = blockIdx.y, N = a.N, E = a.E;
  const int len = (int)a.lengths[b];
  const int key = blockIdx.x * 128 + wave * 32 + l31;
  const long ld_g = 3L * E;
  const TC* base = reinterpret_cast<const TC*>(a.qkv) + (long)b * N * ld_g + h * DH;
  const TC* dO = reinterpret_cast<const TC*>(a.d_o) + (long)b * N * E + h * DH;
  TC* dK = reinterpret_cast<TC*>(a.dqkv) + (long)b * N * ld_g + E + h * DH;
  TC* dV = dK + E;
  const float* lse = a.lse + ((long)b * a.H + h) * N;
  const float* delta = a.delta + ((long)b * a.H + h) * N;
  const bool key_valid = key < len;
  const float inv_keep = a.p_drop > 0.f ? 1.f / (1.f - a.p_drop) : 1.f;   // dV: applied once at the end

  f32x16 dvT[MT], dkT[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dvT[mt][r] = 0.f; dkT[mt][r] = 0.f; }

  if (blockIdx.x * 128 < len) {
    frag_t kf[KS], vf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      kf[ks] = key < N ? *reinterpret_cast<const frag_t*>(base + E + (long)key * ld_g + ks * 16 + g * 8) : zero8<TC>();
      vf[ks] = key < N ? *reinterpret_cast<const frag_t*>(base + 2 * E + (long)key * ld_g + ks * 16 + g * 8) : zero8<TC>();
    }
    const uint32_t th = athresh(a.p_drop);
    // counter(q, key) = ctr_lane + q-dependent scalar; the field multiplier is a lane constant (key parity)
    const uint32_t ctr_q = (uint32_t)N * DX_CTR_MUL;
    const uint32_t ctr_lane = (uint32_t)(key & ~1) * DX_CTR_MUL + dx_key32(a.seed, (uint32_t)(b * a.H + h)) + 4u * g * ctr_q;
    const uint32_t mult_lane = (key & 1) ? DX_M24_ODD : DX_M24_EVEN;
    uint32_t ctr_row[16];   // wave-uniform (SGPR) per-register query offsets of the counter
#pragma unroll
    for (int r = 0; r < 16; ++r) ctr_row[r] = __builtin_amdgcn_readfirstlane((uint32_t)((r & 3) + 8 * (r >> 2)) * ctr_q);
    const float c2 = a.scale * LOG2E;
    const bool tile_k_valid = blockIdx.x * 128 + wave * 32 + 32 <= len;

    TileRegs<TC, DH, KT> qreg, doreg;
    float lse_r = 0.f, delta_r = 0.f;
    auto fetch_stats = [&](int qt0) {
      const int qq = qt0 + tid;
      lse_r = (tid < KT && qq < N) ? lse[qq] * LOG2E : 0.f;   // log2 domain, see fast_exp2
      delta_r = (tid < KT && qq < N) ? delta[qq] : 0.f;
    };
    constexpr bool AHEAD = sizeof(TC) == 2;
    if (AHEAD) {
      qreg.fetch(base, ld_g, 0, N, tid);
      doreg.fetch(dO, E, 0, N, tid);
      fetch_stats(0);
    }
    for (int qt0 = 0; qt0 < len; qt0 += KT) {
      if (!AHEAD) {
        qreg.fetch(base, ld_g, qt0, N, tid);
        doreg.fetch(dO, E, qt0, N, tid);
        fetch_stats(qt0);
      }
      qreg.template commit<LD>(Qs, tid);
      doreg.template commit<LD>(dOs, tid);
      if (tid < KT) { lse_s[tid] = lse_r; delta_s[tid] = delta_r; }
      __syncthreads();
      if (AHEAD && qt0 + KT < len) {
        qreg.fetch(base, ld_g, qt0 + KT, N, tid);
        doreg.fetch(dO, E, qt0 + KT, N, tid);
        fetch_stats(qt0 + KT);
      }
#pragma unroll (DH <= 16 ? 1 : 2)
      for (int sub = 0; sub < KT / 32; ++sub) {
        const int qb = qt0 + sub * 32;
        if (qb < len) {
          f32x16 s, dp;
#pragma unroll
          for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) {
            frag_t qf = *reinterpret_cast<const frag_t*>(&Qs[(sub * 32 + l31) * LD + ks * 16 + g * 8]);
            frag_t dof = *reinterpret_cast<const frag_t*>(&dOs[(sub * 32 + l31) * LD + ks * 16 + g * 8]);
            dx_mma(s, qf, kf[ks]);    // S[q][key]: lane = key column, registers = queries
            dx_mma(dp, dof, vf[ks]);  // dP[q][key]
          }
          float pd[16], ds[16];
          const bool interior = tile_k_valid && qb + 32 <= len;   // wave-uniform
          const uint32_t ctr_sub = ctr_lane + __builtin_amdgcn_readfirstlane((uint32_t)qb * ctr_q);
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = sub * 32 + dx_acc_row(r, g);
            const int qq = qt0 + row;
            float p = fast_exp2<TC>(fmaf(s[r], c2, -lse_s[row]));
            if (!interior) p = (key_valid && qq < len) ? p : 0.f;
            // same decision as the forward (th == 0 keeps everything; no branch around the hash: backward = training)
            const bool keep = dx_drop_field(dx_drop_prefix(ctr_sub + ctr_row[r]), mult_lane) >= th;
            pd[r] = keep ? p : 0.f;
            ds[r] = p * fmaf(keep ? dp[r] : 0.f, inv_keep, -delta_s[row]);
          }
          frag_t pf[2] = {pack8<TC>(pd), pack8<TC>(pd + 8)};
          frag_t dsf[2] = {pack8<TC>(ds), pack8<TC>(ds + 8)};
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int kstep = 0; kstep < 2; ++kstep) {
              const int kA = kstep * 16 + 4 * g;
              frag_t doT = gather8<TC, WRAP>(dOs + sub * 32 * LD, LD, kA, kA + 8, mt * 32, lane);
              dx_mma(dvT[mt], doT, pf[kstep]);
              frag_t qT = gather8<TC, WRAP>(Qs + sub * 32 * LD, LD, kA, kA + 8, mt * 32, lane);
              dx_mma(dkT[mt], qT, dsf[kstep]);
            }
        }
      }
      __syncthreads();
    }
  }
  if (key < N) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int d = mt * 32 + dx_acc_row(r, g);
        if (d < DH) {
          dV[(long)key * ld_g + d] = (TC)(dvT[mt][r] * inv_keep);
          dK[(long)key * ld_g + d] = (TC)(dkT[mt][r] * a.scale);
        }
      }
  }
}

template <typename TC>
int launch_fwd(const AttnArgs& a, int B, int dh, hipStream_t s) {
  dim3 grid(dx_cdiv(a.N, 128), a.H, B), block(256);
  if (dh == 16) hipLaunchKernelGGL((attn_fwd_kernel<TC, 16>), grid, block, 0, s, a);
  else if (dh == 64) hipLaunchKernelGGL((attn_fwd_kernel<TC, 64>), grid, block, 0, s, a);
  else { dx_set_error("attention: head dim %d unsupported (16, 64)", dh); return DX_ERR_UNSUPPORTED; }
  DX_LAUNCH_CHECK();
  return DX_OK;
}
template <typename TC>
int launch_bwd(const AttnArgs& a, int B, int dh, float* delta, hipStream_t s) {
  dim3 grid(dx_cdiv(a.N, 128), a.H, B), block(256);
  if (dh == 16) {
    hipLaunchKernelGGL((attn_bwd_dq_kernel<TC, 16>), grid, block, 0, s, a);
    hipLaunchKernelGGL((attn_bwd_dkv_kernel<TC, 16>), grid, block, 0, s, a);
  } else if (dh == 64) {
    hipLaunchKernelGGL((attn_bwd_dq_kernel<TC, 64>), grid, block, 0, s, a);
    hipLaunchKernelGGL((attn_bwd_dkv_kernel<TC, 64>), grid, block, 0, s, a);
  } else { dx_set_error("attention: head dim %d unsupported (16, 64)", dh); return DX_ERR_UNSUPPORTED; }
  DX_LAUNCH_CHECK();
  return DX_OK;
}

}  // namespace

extern "C" int dx_attention_fwd(const void* qkv, int dtype, const int64_t* lengths, void* o, float* lse, int B, int N,
                                int H, int E, float p_drop, uint64_t seed, void* stream) {
  DX_REQUIRE(qkv && lengths && o, DX_ERR_ARG, "dx_attention_fwd: null pointer");
  DX_REQUIRE(B > 0 && N > 0 && H > 0 && E % H == 0, DX_ERR_SHAPE, "dx_attention_fwd: bad shape B=%d N=%d H=%d E=%d", B, N, H, E);
  DX_REQUIRE(p_drop >= 0.f && p_drop < 1.f, DX_ERR_ARG, "dx_attention_fwd: dropout p out of [0,1)");
  const int dh = E / H;
  AttnArgs a{qkv, o, lse, nullptr, nullptr, nullptr, lengths, N, H, E, 1.f / sqrtf((float)dh), p_drop, seed};
  if (dtype == DX_BF16) return launch_fwd<bf16_t>(a, B, dh, (hipStream_t)stream);
  if (dtype == DX_F32) return launch_fwd<float>(a, B, dh, (hipStream_t)stream);
  dx_set_error("dx_attention_fwd: bad dtype %d", dtype);
  return DX_ERR_DTYPE;
}

extern "C" int dx_attention_bwd(const void* qkv, const void* o, const void* d_o, int dtype, const float* lse,
                                const int64_t* lengths, void* dqkv, float* delta_ws, int B, int N, int H, int E,
                                float p_drop, uint64_t seed, void* stream) {
  DX_REQUIRE(qkv && o && d_o && lse && lengths && dqkv && delta_ws, DX_ERR_ARG, "dx_attention_bwd: null pointer");
  DX_REQUIRE(B > 0 && N > 0 && H > 0 && E % H == 0, DX_ERR_SHAPE, "dx_attention_bwd: bad shape");
  const int dh = E / H;
  AttnArgs a{qkv, const_cast<void*>(o), const_cast<float*>(lse), d_o, delta_ws, dqkv, lengths, N, H, E,
             1.f / sqrtf((float)dh), p_drop, seed};
  if (dtype == DX_BF16) return launch_bwd<bf16_t>(a, B, dh, delta_ws, (hipStream_t)stream);
  if (dtype == DX_F32) return launch_bwd<float>(a, B, dh, delta_ws, (hipStream_t)stream);
  dx_set_error("dx_attention_bwd: bad dtype %d", dtype);
  return DX_ERR_DTYPE;
}
