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
#include <stdlib.h>

#include <type_traits>

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
  const int* order;        // launch rank -> utterance, longest first (dx_length_order); NULL = identity
  int B, gx;               // set by the launchers: utterances, tiles per (utterance, head) along the owned axis
  const DxStepScalars* step;   // NULL, or the device-side step block whose salt is added to seed (captured steps)
  int* counters;           // fused backward: B * H arrival counters (zero between launches)
};

// XCD-aware launch order.  Workgroups go to the 8 XCDs round-robin in flat index order, and every XCD has its own L2.  With the
// natural (tile, head, utterance) order the heads of one utterance land on eight different XCDs, and since a 128-byte line of
// the (B, N, 3E) / (B, N, E) rows holds the 16-channel slices of FOUR heads, every XCD fetched every line: FETCH_SIZE was 8x
// the tensors (290 MB per fused-backward launch for 37 MB of live rows).  Here all workgroups of one utterance -- every head,
// every tile -- decode to the same XCD (the ranks of the longest-first order dealt to the XCDs in a snake), so a line is fetched into one L2
// once and the other heads hit it.  `per` = workgroups per utterance; the grid is rounded up to 8 utterances (extras exit).
constexpr int DX_XCDS = 8;
__device__ __forceinline__ bool attn_decode(int flat, int per, int B, int& rank, int& inner) {
  const int xcd = flat % DX_XCDS, j = flat / DX_XCDS, grp = j / per;
  rank = grp * DX_XCDS + ((grp & 1) ? DX_XCDS - 1 - xcd : xcd);   // snake over the length-sorted ranks: the XCDs get equal shares of the long utterances
  inner = j - grp * per;
  return rank < B;
}
static inline unsigned attn_grid(int B, int per) { return (unsigned)(((B + DX_XCDS - 1) / DX_XCDS) * DX_XCDS * per); }

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
  // BRANCH-FREE on purpose: rows past the end of the tensor re-read its last row (their scores are masked / their
  // probabilities are exactly 0, and the row is finite, so nothing leaks) instead of being predicated to zero.  With the
  // predicated form every load sat in its own exec-masked block and the compiler closed the merge point with
  // s_waitcnt vmcnt(0) in front of the first MFMA of the stage -- the "prefetch" was waited for right after it was issued.
  __device__ __forceinline__ void fetch(const TC* src, long ld_g, int row0, int n_lim, int tid) {
    static_assert((ROWS * CPR) % 256 == 0, "tile must split evenly over the workgroup");
#pragma unroll
    for (int t = 0; t < PT; ++t) {
      const int c = tid + t * 256, r = c / CPR, kc = (c - r * CPR) * 8;
      const int row = min(row0 + r, n_lim - 1);
      v[t] = *reinterpret_cast<const typename Vec8<TC>::type*>(src + (long)row * ld_g + kc);
    }
  }
  template <int LD> __device__ __forceinline__ void commit(TC* dst, int tid) const {
#pragma unroll
    for (int t = 0; t < PT; ++t) {
      const int c = tid + t * 256, r = c / CPR, kc = (c - r * CPR) * 8;
      *reinterpret_cast<typename Vec8<TC>::type*>(dst + r * LD + kc) = v[t];
    }
  }
};

typedef float f32x2 __attribute__((ext_vector_type(2)));
// combine a value with the one held by the other half-wave's lane (lane ^ 32).  (A v_permlane32_swap form was tried: 1 % faster
// on the bf16 kernels, wrong results in the exact-fp32 instantiations -- not worth chasing.)
__device__ __forceinline__ float xhalf_max(float v) { return fmaxf(v, __shfl_xor(v, 32, 64)); }
__device__ __forceinline__ float xhalf_sum(float v) { return v + __shfl_xor(v, 32, 64); }
#ifndef DX_ATTN_KT16
#define DX_ATTN_KT16 128
#endif
#ifndef DX_ATTN_AHEAD
#define DX_ATTN_AHEAD 1
#endif
#ifndef DX_ATTN_OCC64
#define DX_ATTN_OCC64 2   // minimum waves per SIMD of the d_head = 64 forward / dQ kernels
#endif
#ifndef DX_ATTN_SPLIT64
#define DX_ATTN_SPLIT64 2 // d_head = 64: wave groups per workgroup that share the streamed axis (1 = every wave walks all of it)
#endif
#ifndef DX_ATTN_OCC16
#define DX_ATTN_OCC16 5   // workgroups (4 waves) per CU for the d_head = 16 forward / dQ kernels = waves per SIMD
#endif
// two scores per VALU instruction where the ISA has a packed fp32 form (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32)
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }

// rows of the streamed axis per LDS stage: every stage exposes one global-load latency, so small heads take big stages
template <int DH> struct Stage { static constexpr int KT = DH <= 16 ? DX_ATTN_KT16 : 128; };
// d_head = 64 has 2 heads: 128-row workgroups give a ragged batch of 48 utterances ~420 live workgroups = 1.6 waves per SIMD,
// and the kernels are latency-bound there (17-22 % VALU busy).  SP = 2 halves the rows a workgroup owns (64) and lets its two
// wave pairs take alternate 32-row sub-tiles of every LDS stage; the partial results meet in LDS once, after the last stage
// (forward: the usual two-way log-sum-exp merge; backward: plain sums).  Twice the waves for the same work per wave.
template <int DH> struct Split { static constexpr int value = DH >= 64 ? DX_ATTN_SPLIT64 : 1; };

// =============================================================================== forward
template <typename TC, int DH>
__global__ __launch_bounds__(256, sizeof(TC) == 2 ? (DH <= 16 ? DX_ATTN_OCC16 : DX_ATTN_OCC64) : 2) void attn_fwd_kernel(AttnArgs a) {
  constexpr int KT = Stage<DH>::KT;
  constexpr int LD = DH + APad<TC>::value, KS = DH / 16, MT = (DH + 31) / 32, WRAP = DH >= 32 ? 32 : 16;
  typedef typename Vec8<TC>::type frag_t;
  __shared__ __attribute__((aligned(16))) TC Ks[KT * LD];
  __shared__ __attribute__((aligned(16))) TC Vs[KT * LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, g = lane >> 5;
  constexpr int SP = Split<DH>::value, QB = 128 / SP, WQ = 4 / SP;
  // the wave index as a scalar (the compiler cannot see that tid >> 6 is wave-uniform) -- not for d_head = 64 here: the scalar
  // form of this kernel measured 45 vs 41 us
  const int wv = DH >= 64 ? wave : __builtin_amdgcn_readfirstlane(wave);
  const int wq = wv % WQ, kh = wv / WQ;
  int rank_, inner_;
  if (!attn_decode((int)blockIdx.x, a.gx * a.H, a.B, rank_, inner_)) return;      // (grid rounded up to 8 utterances)
  const int bx = inner_ % a.gx;                                                    // tile along the owned axis
  const int b = a.order ? a.order[rank_] : rank_, h = inner_ / a.gx, N = a.N, E = a.E;
  const int len = (int)a.lengths[b];
  const int q = bx * QB + wq * 32 + l31;
  const long ld_g = 3L * E;
  const TC* base = reinterpret_cast<const TC*>(a.qkv) + (long)b * N * ld_g + h * DH;
  TC* O = reinterpret_cast<TC*>(a.o) + (long)b * N * E + h * DH;
  float* lse = a.lse ? a.lse + ((long)b * a.H + h) * N : nullptr;

  const int fend = dx_fill_end(len, N);   // dead rows are written (as zeros) only below it: nobody reads past it (dx_common.h)
  if (bx * QB >= len) {  // whole tile of pad queries: their rows are zeroed after the LayerNorm anyway
    if (q < fend && kh == 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int d = dx_acc_row(r, g);
        if (d < DH) for (int mt = 0; mt < MT; ++mt) O[(long)q * E + mt * 32 + d] = (TC)0.f;
      }
      if (lse && g == 0) lse[q] = 0.f;
    }
    return;
  }

  // a wave whose 32 queries are all padding (the ragged end of an utterance: half the waves of its last tile on average)
  // still stages tiles and meets the barriers, but issues no MFMA / softmax work; its rows leave as zeros
  const bool wave_live = bx * QB + wq * 32 < len;
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
  const uint32_t th8 = dx_drop_th8(a.p_drop);
  const float inv_keep = dx_drop_inv_keep8(th8);   // applied once, to the output row
  // dropout block counter of (q, 4 keys) = lane part + a scalar that follows the key tile (dx_common.h)
  const uint32_t NB = (uint32_t)(N + 3) >> 2;
  const uint32_t ctr_lane = dx_opaque(((uint32_t)(q >> 2) * NB + g) * DX_CTR_MUL + dx_key32(dx_seed_eff(a.seed, a.step), (uint32_t)(b * a.H + h)));
  const uint32_t rot_lane = 8u * (q & 3), mult_lane = dx_blk_mult(q & 3);

  constexpr bool AHEAD = sizeof(TC) == 2 && DX_ATTN_AHEAD;   // exact-fp32 mode: twice the registers per tile, loads stay in place
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
    for (int s2 = 0; s2 < KT / 32 / SP; ++s2) {
      const int sub = s2 * SP + kh;   // SP = 2: the wave groups take alternate sub-tiles
      const int k0 = kt0 + sub * 32;
      if (k0 < len && wave_live) {
        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          frag_t kf = *reinterpret_cast<const frag_t*>(&Ks[(sub * 32 + l31) * LD + ks * 16 + g * 8]);
          dx_mma(s, kf, qf[ks]);
        }
        float p[16];
        if (k0 + 32 > len) {   // boundary tile only: pad keys -> -inf
#pragma unroll
          for (int r = 0; r < 16; ++r) s[r] = (k0 + dx_acc_row(r, g) < len) ? s[r] : -INFINITY;
        }
        float mx = fmaxf(fmaxf(s[0], s[1]), s[2]);     // v_max3_f32: two scores per instruction
#pragma unroll
        for (int r = 3; r < 15; r += 2) mx = fmaxf(fmaxf(mx, s[r]), s[r + 1]);
        mx = fmaxf(mx, s[15]);
        mx = xhalf_max(mx);
        const float m_new = fmaxf(m, mx * c2);           // running max in the scaled log2 domain
        float alpha = 1.f;
        const bool moved = !__all(m_new == m);           // wave-uniform: most tiles after the first few skip the rescale
        if (moved) { alpha = fast_exp2<TC>(m - m_new); m = m_new; }
        const f32x2 c22 = {c2, c2}, nm2 = {-m, -m};
        f32x2 rs2 = {0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          const f32x2 t = pk_fma(f32x2{s[r], s[r + 1]}, c22, nm2);
          const f32x2 e = {fast_exp2<TC>(t[0]), fast_exp2<TC>(t[1])};
          rs2 += e;
          p[r] = e[0]; p[r + 1] = e[1];
        }
        l = l * alpha + (rs2[0] + rs2[1]);             // this half-wave's 16 keys; the halves meet after the last stage
        if (th8) {   // registers 4j .. 4j + 3 hold 4 consecutive keys: one block hash, one row word, four byte fields
          const uint32_t ctr_tile = (uint32_t)(k0 >> 2) * DX_CTR_MUL;
          uint32_t w[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) w[j] = ctr_lane + (ctr_tile + (uint32_t)(2 * j) * DX_CTR_MUL);
          dx_drop_prefix4(w);
#pragma unroll
          for (int j = 0; j < 4; ++j) w[j] = __builtin_amdgcn_alignbit(w[j], w[j], rot_lane);
#pragma unroll
          for (int j = 0; j < 4; ++j) w[j] = __umul24(w[j], mult_lane);
#pragma unroll
          for (int j = 0; j < 4; ++j) dx_drop4(p[4 * j], p[4 * j + 1], p[4 * j + 2], p[4 * j + 3], w[j], th8);
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
  l = xhalf_sum(l);
  if (SP > 1) {   // the loop ended with a barrier: the K tile is dead, its LDS carries the second wave group's partial result
    constexpr int NR = MT * 16 + 2;
    static_assert(SP == 1 || WQ * NR * 64 * sizeof(float) <= KT * LD * sizeof(TC), "merge buffer must fit in the K tile");
    float* red = reinterpret_cast<float*>(Ks) + (long)wq * NR * 64 + lane;
#pragma unroll
    for (int k = 1; k < SP; ++k) {   // wave group k hands its partial to group 0, one group per round (fixed merge order)
      if (kh == k) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int r = 0; r < 16; ++r) red[(mt * 16 + r) * 64] = oT[mt][r];
        red[(MT * 16) * 64] = m;
        red[(MT * 16 + 1) * 64] = l;
      }
      __syncthreads();
      if (kh == 0) {
        const float m1 = red[(MT * 16) * 64], l1 = red[(MT * 16 + 1) * 64];   // m1 = -inf, l1 = 0 if that group saw no key
        const float mm = fmaxf(m, m1);
        const float a0 = fast_exp2<TC>(m - mm), a1 = fast_exp2<TC>(m1 - mm);
        l = l * a0 + l1 * a1;
        m = mm;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int r = 0; r < 16; ++r) oT[mt][r] = oT[mt][r] * a0 + red[(mt * 16 + r) * 64] * a1;
      }
      if (k + 1 < SP) __syncthreads();
    }
    if (kh) return;
  }
  if (q < fend) {
    const float inv_l = wave_live ? inv_keep / l : 0.f;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        const int d0 = 8 * r4 + 4 * g;  // rows d0..d0+3 = registers 4*r4 .. 4*r4+3
        if (mt * 32 + d0 < DH) {
#pragma unroll
          for (int t = 0; t < 4; ++t) O[(long)q * E + mt * 32 + d0 + t] = (TC)(wave_live ? oT[mt][r4 * 4 + t] * inv_l : 0.f);
        }
      }
    }
    if (lse && g == 0) lse[q] = wave_live ? m * LN2 + logf(l) : 0.f;
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
__global__ __launch_bounds__(256, sizeof(TC) == 2 ? (DH <= 16 ? 4 : DX_ATTN_OCC64) : 2) void attn_bwd_dq_kernel(AttnArgs a) {   // (d_head 16: 4 waves per SIMD -- at 5 the kernel kept 20 bytes of scratch; it is the N > 1024 fallback of the fused backward)
  constexpr int KT = Stage<DH>::KT;
  constexpr int LD = DH + APad<TC>::value, KS = DH / 16, MT = (DH + 31) / 32, WRAP = DH >= 32 ? 32 : 16;
  typedef typename Vec8<TC>::type frag_t;
  __shared__ __attribute__((aligned(16))) TC Ks[KT * LD];
  __shared__ __attribute__((aligned(16))) TC Vs[KT * LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, g = lane >> 5;
  constexpr int SP = Split<DH>::value, QB = 128 / SP, WQ = 4 / SP;
  const int wv = __builtin_amdgcn_readfirstlane(wave);   // scalar: the row-group tests below become s_cbranch (5 % on the backward kernels)
  const int wq = wv % WQ, kh = wv / WQ;
  int rank_, inner_;
  if (!attn_decode((int)blockIdx.x, a.gx * a.H, a.B, rank_, inner_)) return;      // (grid rounded up to 8 utterances)
  const int bx = inner_ % a.gx;                                                    // tile along the owned axis
  const int b = a.order ? a.order[rank_] : rank_, h = inner_ / a.gx, N = a.N, E = a.E;
  const int len = (int)a.lengths[b];
  const int q = bx * QB + wq * 32 + l31;
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

  if (bx * QB >= len && q < N && g == 0)   // pad-query tiles: delta is never used, keep it finite
    const_cast<float*>(a.delta)[((long)b * a.H + h) * N + q] = 0.f;
  if (bx * QB < len) {
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
    delta_q = xhalf_sum(delta_q);
    if (q < N && g == 0 && kh == 0) const_cast<float*>(a.delta)[stat] = delta_q;
    const float c2 = a.scale * LOG2E, lse2 = lse_q * LOG2E;
    const bool tile_q_valid = bx * QB + wq * 32 + 32 <= len;
    const bool wave_live = bx * QB + wq * 32 < len;
    const uint32_t th8 = dx_drop_th8(a.p_drop);
    const float inv_keep = dx_drop_inv_keep8(th8);
    const uint32_t NB = (uint32_t)(N + 3) >> 2;
    const uint32_t ctr_lane = dx_opaque(((uint32_t)(q >> 2) * NB + g) * DX_CTR_MUL + dx_key32(dx_seed_eff(a.seed, a.step), (uint32_t)(b * a.H + h)));
    const uint32_t rot_lane = 8u * (q & 3), mult_lane = dx_blk_mult(q & 3);

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
      for (int s2 = 0; s2 < KT / 32 / SP; ++s2) {
        const int sub = s2 * SP + kh;   // SP = 2: the wave groups take alternate sub-tiles
        const int k0 = kt0 + sub * 32;
        if (k0 < len && wave_live) {   // all-padding waves: dQ stays zero (see the forward)
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
          const uint32_t ctr_tile = (uint32_t)(k0 >> 2) * DX_CTR_MUL;
          const f32x2 c22 = {c2, c2}, nl2 = {-lse2, -lse2}, ik2 = {inv_keep, inv_keep}, nd2 = {-delta_q, -delta_q};
          // th8 == 0 keeps everything; no branch around the hashes (backward = training)
          uint32_t w[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) w[j] = ctr_lane + (ctr_tile + (uint32_t)(2 * j) * DX_CTR_MUL);
          dx_drop_prefix4(w);
#pragma unroll
          for (int j = 0; j < 4; ++j) w[j] = __builtin_amdgcn_alignbit(w[j], w[j], rot_lane);
#pragma unroll
          for (int j = 0; j < 4; ++j) w[j] = __umul24(w[j], mult_lane);
#pragma unroll
          for (int j = 0; j < 4; ++j) {   // registers 4j .. 4j + 3 = keys k0 + 8j + 4g + {0..3}: one row word (see forward)
            float x[4] = {dp[4 * j], dp[4 * j + 1], dp[4 * j + 2], dp[4 * j + 3]};
            dx_drop4(x[0], x[1], x[2], x[3], w[j], th8);
#pragma unroll
            for (int i = 0; i < 4; i += 2) {
              const int r = 4 * j + i;
              const int key = k0 + dx_acc_row(r, g);
              const f32x2 t = pk_fma(f32x2{s[r], s[r + 1]}, c22, nl2);
              f32x2 pr = {fast_exp2<TC>(t[0]), fast_exp2<TC>(t[1])};
              if (!interior) { pr[0] = (q_valid && key < len) ? pr[0] : 0.f; pr[1] = (q_valid && key + 1 < len) ? pr[1] : 0.f; }
              const f32x2 d2 = pr * pk_fma(f32x2{x[i], x[i + 1]}, ik2, nd2);
              ds[r] = d2[0]; ds[r + 1] = d2[1];
            }
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
    if (SP > 1) {   // sum the two wave groups' partial dQ^T through the (dead) K tile
      static_assert(SP == 1 || WQ * MT * 16 * 64 * sizeof(float) <= KT * LD * sizeof(TC), "merge buffer must fit in the K tile");
      float* red = reinterpret_cast<float*>(Ks) + (long)wq * (MT * 16) * 64 + lane;
#pragma unroll
      for (int k = 1; k < SP; ++k) {   // one wave group per round, added in group order
        if (kh == k) {
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) red[(mt * 16 + r) * 64] = dqT[mt][r];
        }
        __syncthreads();
        if (!kh) {
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) dqT[mt][r] += red[(mt * 16 + r) * 64];
        }
        if (k + 1 < SP) __syncthreads();
      }
    }
  }
  if (q < dx_fill_end(len, N) && kh == 0) {   // (dead rows past the fill end stay unwritten, dx_common.h)
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
__global__ __launch_bounds__(256, DH <= 16 ? 3 : (sizeof(TC) == 4 ? 1 : 2)) void attn_bwd_dkv_kernel(AttnArgs a) {   // (exact-fp32 parity path, d_head 64: twice the operand registers -- one wave per SIMD instead of 148 bytes of scratch)
  constexpr int KT = Stage<DH>::KT;
  constexpr int LD = DH + APad<TC>::value, KS = DH / 16, MT = (DH + 31) / 32, WRAP = DH >= 32 ? 32 : 16;
  typedef typename Vec8<TC>::type frag_t;
  __shared__ __attribute__((aligned(16))) TC Qs[KT * LD];
  __shared__ __attribute__((aligned(16))) TC dOs[KT * LD];
  __shared__ float lse_s[KT], delta_s[KT];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, g = lane >> 5;
  constexpr int SP = Split<DH>::value, QB = 128 / SP, WQ = 4 / SP;
  const int wv = __builtin_amdgcn_readfirstlane(wave);   // scalar: the row-group tests below become s_cbranch (5 % on the backward kernels)
  const int wq = wv % WQ, kh = wv / WQ;
  int rank_, inner_;
  if (!attn_decode((int)blockIdx.x, a.gx * a.H, a.B, rank_, inner_)) return;      // (grid rounded up to 8 utterances)
  const int bx = inner_ % a.gx;                                                    // tile along the owned axis
  const int b = a.order ? a.order[rank_] : rank_, h = inner_ / a.gx, N = a.N, E = a.E;
  const int len = (int)a.lengths[b];
  const int key = bx * QB + wq * 32 + l31;
  const long ld_g = 3L * E;
  const TC* base = reinterpret_cast<const TC*>(a.qkv) + (long)b * N * ld_g + h * DH;
  const TC* dO = reinterpret_cast<const TC*>(a.d_o) + (long)b * N * E + h * DH;
  TC* dK = reinterpret_cast<TC*>(a.dqkv) + (long)b * N * ld_g + E + h * DH;
  TC* dV = dK + E;
  const float* lse = a.lse + ((long)b * a.H + h) * N;
  const float* delta = a.delta + ((long)b * a.H + h) * N;
  const bool key_valid = key < len;
  const float inv_keep = dx_drop_inv_keep8(dx_drop_th8(a.p_drop));   // dV: applied once at the end (the forward's quantised keep rate)

  f32x16 dvT[MT], dkT[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dvT[mt][r] = 0.f; dkT[mt][r] = 0.f; }

  if (bx * QB < len) {
    frag_t kf[KS], vf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      kf[ks] = key < N ? *reinterpret_cast<const frag_t*>(base + E + (long)key * ld_g + ks * 16 + g * 8) : zero8<TC>();
      vf[ks] = key < N ? *reinterpret_cast<const frag_t*>(base + 2 * E + (long)key * ld_g + ks * 16 + g * 8) : zero8<TC>();
    }
    const uint32_t th8 = dx_drop_th8(a.p_drop), th_top = th8 << 24;
    // block counter ((q >> 2) * NB + (key >> 2)) * MUL + stream = lane part (key block, lane group) + a wave-uniform part that
    // follows the query rows; the lane reads byte (key & 3) of each row word (dx_common.h)
    const uint32_t NB = (uint32_t)(N + 3) >> 2;
    const uint32_t ctr_q = NB * DX_CTR_MUL;   // one query block further
    const uint32_t ctr_lane = dx_opaque(((uint32_t)(key >> 2) + g * NB) * DX_CTR_MUL + dx_key32(dx_seed_eff(a.seed, a.step), (uint32_t)(b * a.H + h)));
    const uint32_t shl_lane = 24u - 8u * (key & 3);
    const float c2 = a.scale * LOG2E;
    const bool tile_k_valid = bx * QB + wq * 32 + 32 <= len;
    const bool wave_live = bx * QB + wq * 32 < len;

    TileRegs<TC, DH, KT> qreg, doreg;
    float lse_r = 0.f, delta_r = 0.f;
    auto fetch_stats = [&](int qt0) {
      const int qq = qt0 + tid;
      const int qc = min(qq, N - 1);                           // branch-free, like TileRegs::fetch
      lse_r = lse[qc] * LOG2E;                                 // log2 domain, see fast_exp2
      delta_r = delta[qc];
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
      for (int s2 = 0; s2 < KT / 32 / SP; ++s2) {
        const int sub = s2 * SP + kh;   // SP = 2: the wave groups take alternate sub-tiles
        const int qb = qt0 + sub * 32;
        if (qb < len && wave_live) {   // all-padding key waves: dK / dV stay zero
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
          const uint32_t ctr_sub = ctr_lane + __builtin_amdgcn_readfirstlane((uint32_t)(qb >> 2) * ctr_q);
          const f32x2 c22 = {c2, c2}, ik2 = {inv_keep, inv_keep};
          uint32_t bases[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) bases[j] = ctr_sub + (uint32_t)(2 * j) * ctr_q;
          dx_drop_prefix4(bases);
#pragma unroll
          for (int j = 0; j < 4; ++j) {   // registers 4j .. 4j + 3 = queries qb + 8j + 4g + {0..3}: one block hash, four row words
            const uint32_t base = bases[j];
            f32x2 pr[2], nd2[2];
#pragma unroll
            for (int i = 0; i < 4; i += 2) {
              const int r = 4 * j + i;
              const int row = sub * 32 + dx_acc_row(r, g);
              const int qq = qt0 + row;
              const f32x2 nl2 = {-lse_s[row], -lse_s[row + 1]};
              nd2[i >> 1] = f32x2{-delta_s[row], -delta_s[row + 1]};
              const f32x2 t = pk_fma(f32x2{s[r], s[r + 1]}, c22, nl2);
              pr[i >> 1] = f32x2{fast_exp2<TC>(t[0]), fast_exp2<TC>(t[1])};
              if (!interior) {
                pr[i >> 1][0] = (key_valid && qq < len) ? pr[i >> 1][0] : 0.f;
                pr[i >> 1][1] = (key_valid && qq + 1 < len) ? pr[i >> 1][1] : 0.f;
              }
            }
            // same decisions as the forward (th8 == 0 keeps everything; no branch around the hash: backward = training)
            float pk[4] = {pr[0][0], pr[0][1], pr[1][0], pr[1][1]};
            float x[4] = {dp[4 * j], dp[4 * j + 1], dp[4 * j + 2], dp[4 * j + 3]};
            dx_drop4x2_var(pk, x, dx_drop_row(base, 0u, DX_BLK_M0), dx_drop_row(base, 8u, DX_BLK_M1), dx_drop_row(base, 16u, DX_BLK_M2),
                           dx_drop_row(base, 24u, DX_BLK_M3), shl_lane, th_top);
#pragma unroll
            for (int i = 0; i < 4; i += 2) {
              const int r = 4 * j + i;
              pd[r] = pk[i]; pd[r + 1] = pk[i + 1];
              const f32x2 d2 = pr[i >> 1] * pk_fma(f32x2{x[i], x[i + 1]}, ik2, nd2[i >> 1]);
              ds[r] = d2[0]; ds[r + 1] = d2[1];
            }
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
    if (SP > 1) {   // sum the two wave groups' partial dV^T / dK^T through the (dead) Q and dO tiles
      static_assert(SP == 1 || WQ * MT * 16 * 64 * sizeof(float) <= KT * LD * sizeof(TC), "merge buffer must fit in a tile");
      float* red_v = reinterpret_cast<float*>(Qs) + (long)wq * (MT * 16) * 64 + lane;
      float* red_k = reinterpret_cast<float*>(dOs) + (long)wq * (MT * 16) * 64 + lane;
#pragma unroll
      for (int k = 1; k < SP; ++k) {   // one wave group per round, added in group order
        if (kh == k) {
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) { red_v[(mt * 16 + r) * 64] = dvT[mt][r]; red_k[(mt * 16 + r) * 64] = dkT[mt][r]; }
        }
        __syncthreads();
        if (!kh) {
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) { dvT[mt][r] += red_v[(mt * 16 + r) * 64]; dkT[mt][r] += red_k[(mt * 16 + r) * 64]; }
        }
        if (k + 1 < SP) __syncthreads();
      }
    }
  }
  if (key < dx_fill_end(len, N) && kh == 0) {   // (dead rows past the fill end stay unwritten, dx_common.h)
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


// =============================================================================== backward, fused (bf16, d_head = 16)
// ONE recomputation of S and dP for dQ, dK and dV (the two kernels above recompute them once each, and the d_head = 16 kernels
// are bound by that VALU work, not by their MFMAs).  A workgroup (4 waves, two resident per CU) owns the KEYS of one (utterance,
// head) -- all of them up to FB_KEYS, an even share of a longer utterance -- and walks ALL of its queries:
//   * wave w keeps the K / V fragments of its <= 4 key blocks (32 keys each) in registers for the whole kernel together with
//     their dK^T / dV^T accumulators (16x16x32 MFMAs: 16 d x 16 keys per accumulator, no padding);
//   * the queries stream through LDS in stages of 128 rows (Q and dO, next stage prefetched in registers; delta = rowsum(dO * O)
//     and the log-sum-exp are staged with them); per 32-query block every wave computes S^T = K Q^T and dP^T = V dO^T for its key
//     blocks (lane = query, keys in registers: log-sum-exp and delta are per-lane scalars, the cheap form of the dropout hash
//     applies), P, the dropout mask and dS once;
//   * dQ^T += K^T dS^T takes dS^T straight from the registers (B operand) and K^T from the workgroup's K rows in LDS;
//   * dV^T += dO^T P_drop and dK^T += Q^T dS contract over the QUERIES, which sit in the lanes: P_drop and dS pass through a
//     wave-private 32 x 32 bf16 LDS tile and come back key-major with ds_read_b64_tr_b16 (8 b64 writes + 8 transposed reads
//     per tile instead of a second softmax recomputation);
//   * the 4 partial dQ^T of a query block (one per wave) meet in LDS and are summed in wave order: deterministic, no atomics;
//   * an utterance of more than FB_KEYS keys has up to FB_PARTS workgroups (even shares of its key blocks): each leaves its fp32 dQ
//     partial in the workspace, and the one that arrives LAST (an arrival counter per (utterance, head), release / acquire fences at
//     agent scope) adds them in key order and writes dQ -- a fixed order, so the sum does not depend on who arrives when.
//     FB_KB = 4 key blocks per wave = 512 keys per workgroup.  Measured with 2 (256 keys, up to 4 workgroups per (utterance, head):
//     the longest workgroup half as long): 170 vs 120 us on the bench batch -- nearly every utterance then leaves partials
//     (98 MB written + read per launch instead of ~30) and the per-query-block work that does not depend on the key count
//     (fragments, transposes, the dQ hand-over) doubles.
// N <= 1024; longer batches take the two-pass kernels.
#ifndef DX_FB_KB
#define DX_FB_KB 4
#endif
constexpr int FB_W = 4, FB_T = FB_W * 64, FB_KB = DX_FB_KB, FB_QT = 128, FB_KEYS = FB_W * FB_KB * 32, FB_MAXN = 1024, FB_PARTS = FB_MAXN / FB_KEYS;
#ifndef DX_FB_LDT
#define DX_FB_LDT 36
#endif
// FB_LDT: row stride of the wave-private [query][key] tiles.  36 elements = 18 banks: the 32 rows of a 64-bit tile store start on 32
// distinct even banks and the 4 rows x 32 bytes of one 16-lane transposed read fall on 32 distinct banks; 40 (20 banks) made rows r
// and r + 16 collide on the stores and rows 0 and 3 of every transposed read overlap
constexpr int FB_LDK = 24, FB_LDT = DX_FB_LDT;

__device__ __forceinline__ f32x4 dx_mma16(f32x4 acc, const bf16x8& a, const bf16x8& b) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
}
// operand of a 16x16x32 MFMA read k-major from a row-major bf16 tile: lane (i = lane & 15, G = lane >> 4) gets column col0 + i,
// rows row0 + 8 G .. + 7 (two 4 x 16 transposed reads per 16-lane group)
__device__ __forceinline__ bf16x8 tr8(const bf16_t* tile, int ld, int row0, int col0, int lane) {
  const int i = lane & 15, G = lane >> 4;
  const bf16_t* p = tile + (row0 + 8 * G + (i >> 2)) * ld + col0 + 4 * (i & 3);
  s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p));
  s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p + 4 * ld));
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  s16x8 r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, r);
}
// pm[i] = byte i of w >= th8 ? pr[i] : 0 and x[i] likewise in place (one compare per decision, two selects; see dx_drop4)
__device__ __forceinline__ void dx_drop4x2(float* pm, const float* pr, float* x, uint32_t w, uint32_t th8) {
  uint64_t m0, m1, m2, m3;
  asm("v_cmp_ge_u32_sdwa %8, %16, %17 src0_sel:BYTE_0 src1_sel:DWORD\n\t"
      "v_cmp_ge_u32_sdwa %9, %16, %17 src0_sel:BYTE_1 src1_sel:DWORD\n\t"
      "v_cmp_ge_u32_sdwa %10, %16, %17 src0_sel:BYTE_2 src1_sel:DWORD\n\t"
      "v_cmp_ge_u32_sdwa %11, %16, %17 src0_sel:BYTE_3 src1_sel:DWORD\n\t"
      "v_cndmask_b32_e64 %0, 0, %12, %8\n\t"
      "v_cndmask_b32_e64 %1, 0, %13, %9\n\t"
      "v_cndmask_b32_e64 %2, 0, %14, %10\n\t"
      "v_cndmask_b32_e64 %3, 0, %15, %11\n\t"
      "v_cndmask_b32_e64 %4, 0, %4, %8\n\t"
      "v_cndmask_b32_e64 %5, 0, %5, %9\n\t"
      "v_cndmask_b32_e64 %6, 0, %6, %10\n\t"
      "v_cndmask_b32_e64 %7, 0, %7, %11"
      : "=&v"(pm[0]), "=&v"(pm[1]), "=&v"(pm[2]), "=&v"(pm[3]), "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]),
        "=&s"(m0), "=&s"(m1), "=&s"(m2), "=&s"(m3)
      : "v"(pr[0]), "v"(pr[1]), "v"(pr[2]), "v"(pr[3]), "v"(w), "v"(th8));
}

// workspace of the fused kernel behind the (B, H, N) floats of delta: B * H * FB_PARTS fp32 dQ partials of (N_pad x 16); nothing in it has to
// survive a launch.  The B * H arrival counters live in a buffer of their own (`counters`): zero before the first launch, and the last
// arrival of a group puts its counter back to zero, so the same small buffer serves every shape and every later launch on the stream.
__host__ __device__ static inline long fb_ws_floats(int B, int N, int H) {
  const long npad = (N + 31) & ~31;
  return (long)B * H * FB_PARTS * npad * 16;
}

__global__ __launch_bounds__(FB_T, 2) void attn_bwd_fused16_kernel(AttnArgs a, float* ws) {   // grid: attn_grid(B, FB_PARTS * H)
  constexpr int DH = 16;
  typedef bf16_t TC;
  typedef bf16x8 frag_t;
  __shared__ __attribute__((aligned(16))) TC Kall[FB_KEYS * FB_LDK];     // this workgroup's K rows (one head): K^T operand of dQ^T
  __shared__ __attribute__((aligned(16))) TC Qs[FB_QT * FB_LDK];
  __shared__ __attribute__((aligned(16))) TC dOs[FB_QT * FB_LDK];
  __shared__ float lse_s[FB_QT], delta_s[FB_QT];
  __shared__ __attribute__((aligned(16))) TC Tp[FB_W][32 * FB_LDT];      // wave-private transposition tiles [query][key]
  __shared__ __attribute__((aligned(16))) TC Td[FB_W][32 * FB_LDT];
  __shared__ float red[2][FB_W][8][64];                                   // partial dQ^T of a query block, one slot per wave
  __shared__ int arrived;
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, g = lane >> 5, i16 = lane & 15, G = lane >> 4;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int H = a.H, N = a.N, E = a.E;
  int bi, inner_;
  if (!attn_decode((int)blockIdx.x, FB_PARTS * H, a.B, bi, inner_)) return;
  const int part = inner_ % FB_PARTS, h = inner_ / FB_PARTS;
  const int b = a.order ? a.order[bi] : bi;
  int len = (int)a.lengths[b];
  len = len < 0 ? 0 : (len > N ? N : len);
  const int nkb_all = (len + 31) >> 5, rows_live = nkb_all * 32;
  const int nparts = nkb_all > 0 ? (nkb_all + FB_KEYS / 32 - 1) / (FB_KEYS / 32) : 1;   // workgroups that share this (utterance, head)
  const bool split = nparts > 1;
  if (part >= nparts) return;
  const long ld_g = 3L * E;
  const TC* base = reinterpret_cast<const TC*>(a.qkv) + (long)b * N * ld_g + h * DH;
  const TC* dO = reinterpret_cast<const TC*>(a.d_o) + (long)b * N * E + h * DH;
  const TC* Og = reinterpret_cast<const TC*>(a.o) + (long)b * N * E + h * DH;
  TC* dQ = reinterpret_cast<TC*>(a.dqkv) + (long)b * N * ld_g + h * DH;
  const float* lse = a.lse + ((long)b * H + h) * N;
  const long npad = (N + 31) & ~31;
  float* pbuf = ws + (((long)b * H + h) * FB_PARTS + part) * npad * 16;   // split utterances: this workgroup's dQ partial [query][d]
  int* counter = a.counters + (b * H + h);
  if (part == 0) {   // rows past the last live block: dQ | dK | dV are zero (this head's 16 columns of each)
    const frag_t z = zero8<TC>();
    const int nz = (max(dx_fill_end(len, N), rows_live) - rows_live) * 6;   // (only below the fill end: nobody reads past it, dx_common.h)
    for (int c = tid; c < nz; c += FB_T) {
      const int row = rows_live + c / 6, pt = c % 6;
      *reinterpret_cast<frag_t*>(dQ + (long)row * ld_g + (pt >> 1) * E + (pt & 1) * 8) = z;
    }
  }
  if (len == 0) return;
  // key blocks of this workgroup [kbA, kbB) and of this wave [kb0, kb0 + cnt): spread evenly over the waves that get any
  const int kbA = (part * nkb_all) / nparts, kbB = ((part + 1) * nkb_all) / nparts;
  const int nkb = kbB - kbA;
  for (int c = tid; c < nkb * 64; c += FB_T) {
    const int r = c >> 1, hf = (c & 1) * 8, key = kbA * 32 + r;
    frag_t v = zero8<TC>();
    if (key < len) v = *reinterpret_cast<const frag_t*>(base + E + (long)key * ld_g + hf);
    *reinterpret_cast<frag_t*>(&Kall[r * FB_LDK + hf]) = v;
  }
  const int nwa = nkb < FB_W ? nkb : FB_W;
  int kb0 = kbA, cnt = 0;
  if (w < nwa) { kb0 = kbA + (w * nkb) / nwa; cnt = kbA + ((w + 1) * nkb) / nwa - kb0; }
  frag_t kf[FB_KB], vf[FB_KB];
  f32x4 dk[FB_KB][2], dv[FB_KB][2];
#pragma unroll
  for (int j = 0; j < FB_KB; ++j) {
    const int key = (kb0 + j) * 32 + l31;
    const bool ok = j < cnt && key < len;
    kf[j] = ok ? *reinterpret_cast<const frag_t*>(base + E + (long)key * ld_g + g * 8) : zero8<TC>();
    vf[j] = ok ? *reinterpret_cast<const frag_t*>(base + 2 * E + (long)key * ld_g + g * 8) : zero8<TC>();
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) { dk[j][t][r] = 0.f; dv[j][t][r] = 0.f; }
  }
  const float c2 = a.scale * LOG2E;
  const uint32_t th8 = dx_drop_th8(a.p_drop);
  const float inv_keep = dx_drop_inv_keep8(th8);
  const uint32_t NB = (uint32_t)(N + 3) >> 2;
  const uint32_t skey = dx_key32(dx_seed_eff(a.seed, a.step), (uint32_t)(b * H + h));

  const int srow = tid >> 1, shf = (tid & 1) * 8;   // this thread's 16-byte piece of a 128-row stage
  frag_t qreg, doreg;
  float dreg, lreg;
  auto fetch = [&](int qs0) {                         // branch-free: rows past the tensor re-read its last row (masked later)
    const int r = min(qs0 + srow, N - 1);
    qreg = *reinterpret_cast<const frag_t*>(base + (long)r * ld_g + shf);
    doreg = *reinterpret_cast<const frag_t*>(dO + (long)r * E + shf);
    const frag_t og = *reinterpret_cast<const frag_t*>(Og + (long)r * E + shf);
    float d = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) d += (float)doreg[e] * (float)og[e];
    dreg = d;                                         // half of delta_q = sum_d dO[q][d] O[q][d]; the lane pair adds up at commit
    lreg = lse[r] * LOG2E;                            // log2 domain, see fast_exp2
  };
  fetch(0);
  int buf = 0;
  for (int qs0 = 0; qs0 < len; qs0 += FB_QT) {
    *reinterpret_cast<frag_t*>(&Qs[srow * FB_LDK + shf]) = qreg;
    *reinterpret_cast<frag_t*>(&dOs[srow * FB_LDK + shf]) = doreg;
    {
      const float d = dreg + __shfl_xor(dreg, 1, 64);
      if (!(tid & 1)) { const bool ok = qs0 + srow < len; delta_s[srow] = ok ? d : 0.f; lse_s[srow] = ok ? lreg : 0.f; }
    }
    __syncthreads();
    if (qs0 + FB_QT < len) fetch(qs0 + FB_QT);
    const int left = (len - qs0 + 31) >> 5, nsub = left < FB_QT / 32 ? left : FB_QT / 32;
    for (int sub = 0; sub < nsub; ++sub) {
      const int qb = qs0 + sub * 32, q = qb + l31;
      f32x16 dqT;
#pragma unroll
      for (int r = 0; r < 16; ++r) dqT[r] = 0.f;
      if (cnt > 0) {
        const bool q_valid = q < len;
        const frag_t qf = *reinterpret_cast<const frag_t*>(&Qs[(sub * 32 + l31) * FB_LDK + g * 8]);
        const frag_t dof = *reinterpret_cast<const frag_t*>(&dOs[(sub * 32 + l31) * FB_LDK + g * 8]);
        const float lse2 = lse_s[sub * 32 + l31], delta_q = delta_s[sub * 32 + l31];
        const frag_t doA = tr8(dOs, FB_LDK, sub * 32, 0, lane);   // dO^T / Q^T: 16 d x 32 queries of this block
        const frag_t qA = tr8(Qs, FB_LDK, sub * 32, 0, lane);
        const uint32_t ctr_lane = dx_opaque(((uint32_t)(q >> 2) * NB + g) * DX_CTR_MUL + skey);
        const uint32_t rot_lane = 8u * (q & 3), mult_lane = dx_blk_mult(q & 3);
        const f32x2 c22 = {c2, c2}, nl2 = {-lse2, -lse2}, ik2 = {inv_keep, inv_keep}, nd2 = {-delta_q, -delta_q};
#pragma unroll
        for (int j = 0; j < FB_KB; ++j) {
          if (j < cnt) {
            const int k0 = (kb0 + j) * 32;
            f32x16 s, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
            dx_mma(s, kf[j], qf);      // S^T[key][q]: lane = query, registers = keys
            dx_mma(dp, vf[j], dof);    // dP^T[key][q]
            if (!(qb + 32 <= len && k0 + 32 <= len)) {   // wave-uniform, boundary tiles only: pad keys / pad queries -> P = 2^-inf = 0
              asm volatile("; boundary tile" ::: "memory");   // keeps this a branch (if-converted, the 32 compare / select pairs ran on every tile)
#pragma unroll
              for (int r = 0; r < 16; ++r) s[r] = (q_valid && k0 + dx_acc_row(r, g) < len) ? s[r] : -INFINITY;
            }
            const uint32_t ctr_tile = (uint32_t)(k0 >> 2) * DX_CTR_MUL;
            uint32_t hw[4];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) hw[jj] = ctr_lane + (ctr_tile + (uint32_t)(2 * jj) * DX_CTR_MUL);
            dx_drop_prefix4(hw);
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) hw[jj] = __builtin_amdgcn_alignbit(hw[jj], hw[jj], rot_lane);
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) hw[jj] = __umul24(hw[jj], mult_lane);
            float pd[16], ds[16];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {   // registers 4jj .. 4jj + 3 = keys k0 + 8jj + 4g + {0..3}: one row word
              float x[4] = {dp[4 * jj], dp[4 * jj + 1], dp[4 * jj + 2], dp[4 * jj + 3]};
              float pr[4];
#pragma unroll
              for (int i = 0; i < 4; i += 2) {
                const int r = 4 * jj + i;
                const f32x2 t = pk_fma(f32x2{s[r], s[r + 1]}, c22, nl2);
                pr[i] = fast_exp2<TC>(t[0]);
                pr[i + 1] = fast_exp2<TC>(t[1]);
              }
              dx_drop4x2(pd + 4 * jj, pr, x, hw[jj], th8);
#pragma unroll
              for (int i = 0; i < 4; i += 2) {
                const int r = 4 * jj + i;
                const f32x2 d2 = f32x2{pr[i], pr[i + 1]} * pk_fma(f32x2{x[i], x[i + 1]}, ik2, nd2);
                ds[r] = d2[0]; ds[r + 1] = d2[1];
              }
            }
            const frag_t pdf[2] = {pack8<TC>(pd), pack8<TC>(pd + 8)};
            const frag_t dsf[2] = {pack8<TC>(ds), pack8<TC>(ds + 8)};
            // [query][key] tiles for the contractions over the queries: fragment element e of k-step ks = key 16 ks + 8 (e >> 2) + 4 g + (e & 3)
            TC* tp = &Tp[w][l31 * FB_LDT + 4 * g];
            TC* td = &Td[w][l31 * FB_LDT + 4 * g];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
              *reinterpret_cast<bf16x4*>(tp + 16 * ks) = __builtin_shufflevector(pdf[ks], pdf[ks], 0, 1, 2, 3);
              *reinterpret_cast<bf16x4*>(tp + 16 * ks + 8) = __builtin_shufflevector(pdf[ks], pdf[ks], 4, 5, 6, 7);
              *reinterpret_cast<bf16x4*>(td + 16 * ks) = __builtin_shufflevector(dsf[ks], dsf[ks], 0, 1, 2, 3);
              *reinterpret_cast<bf16x4*>(td + 16 * ks + 8) = __builtin_shufflevector(dsf[ks], dsf[ks], 4, 5, 6, 7);
            }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {   // dQ^T += K^T dS^T
              const frag_t kT = gather8<TC, 16>(Kall + (k0 - kbA * 32) * FB_LDK, FB_LDK, ks * 16 + 4 * g, ks * 16 + 4 * g + 8, 0, lane);
              dx_mma(dqT, kT, dsf[ks]);
            }
            asm volatile("" ::: "memory");      // the transposed reads below follow this wave's own tile writes (LDS is in order per wave)
#pragma unroll
            for (int t = 0; t < 2; ++t) {      // key halves of the block: dV^T += dO^T P_drop, dK^T += Q^T dS
              const frag_t pB = tr8(Tp[w], FB_LDT, 0, 16 * t, lane);
              const frag_t dB = tr8(Td[w], FB_LDT, 0, 16 * t, lane);
              dv[j][t] = dx_mma16(dv[j][t], doA, pB);
              dk[j][t] = dx_mma16(dk[j][t], qA, dB);
            }
            asm volatile("" ::: "memory");
          }
        }
      }
      if (w < nwa) {
#pragma unroll
        for (int r = 0; r < 8; ++r) red[buf][w][r][lane] = dqT[r];   // rows d < 16 of the 32 x 32 accumulator
      }
      __syncthreads();
      if (tid < 128) {   // thread = (query, 4 consecutive d): sums the waves' partials in wave order, one store
        const int qq = tid & 31, dg = tid >> 5, rhi = dg >> 1, gg = dg & 1;
        float acc4[4] = {0.f, 0.f, 0.f, 0.f};
        for (int ww = 0; ww < nwa; ++ww)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc4[e] += red[buf][ww][rhi * 4 + e][gg * 32 + qq];
        if (split) {
          *reinterpret_cast<f32x4*>(pbuf + (long)(qb + qq) * 16 + 4 * dg) = f32x4{acc4[0], acc4[1], acc4[2], acc4[3]};
        } else if (qb + qq < N) {
          bf16x4 o4;
#pragma unroll
          for (int e = 0; e < 4; ++e) o4[e] = (TC)(acc4[e] * a.scale);
          *reinterpret_cast<bf16x4*>(dQ + (long)(qb + qq) * ld_g + 4 * dg) = o4;
        }
      }
      buf ^= 1;
    }
  }
#pragma unroll
  for (int j = 0; j < FB_KB; ++j) {
    if (j < cnt) {
#pragma unroll
      for (int t = 0; t < 2; ++t) {   // accumulator (16 d x 16 keys): lane = key, registers = d 4 G .. 4 G + 3
        const int key = (kb0 + j) * 32 + 16 * t + i16;
        if (key < N) {
          bf16x4 k4, v4;
#pragma unroll
          for (int r = 0; r < 4; ++r) { k4[r] = (TC)(dk[j][t][r] * a.scale); v4[r] = (TC)(dv[j][t][r] * inv_keep); }
          *reinterpret_cast<bf16x4*>(dQ + E + (long)key * ld_g + 4 * G) = k4;
          *reinterpret_cast<bf16x4*>(dQ + 2 * E + (long)key * ld_g + 4 * G) = v4;
        }
      }
    }
  }
  if (!split) return;
  // the last arrival of the group adds the partials in key order and writes dQ
  // (hand-off recipe of the CDNA guide, section 6 G16: drain every wave's stores, ONE lane releases at agent scope, then the
  // ticket; the second arrival acquires once -- a __threadfence() per thread made this kernel 1.8x slower)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const int last = __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nparts - 1;
    if (last) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      __hip_atomic_store(counter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // all arrivals are in: zero again for the next launch
    }
    arrived = last;
  }
  __syncthreads();
  if (!arrived) return;
  const float* p0 = ws + (((long)b * H + h) * FB_PARTS) * npad * 16;
  for (int c = tid; c < rows_live * 4; c += FB_T) {
    const int qq = c >> 2, dg = c & 3;
    if (qq < N) {
      f32x4 x = *reinterpret_cast<const f32x4*>(p0 + (long)qq * 16 + 4 * dg);
      for (int pp = 1; pp < nparts; ++pp) x += *reinterpret_cast<const f32x4*>(p0 + (long)pp * npad * 16 + (long)qq * 16 + 4 * dg);
      bf16x4 o4;
#pragma unroll
      for (int e = 0; e < 4; ++e) o4[e] = (TC)(x[e] * a.scale);
      *reinterpret_cast<bf16x4*>(dQ + (long)qq * ld_g + 4 * dg) = o4;
    }
  }
}

template <typename TC>
int launch_fwd(const AttnArgs& a0, int B, int dh, hipStream_t s) {
  AttnArgs a = a0;
  a.B = B;
  a.gx = dx_cdiv(a.N, dh >= 64 ? 128 / Split<64>::value : 128);
  dim3 grid(attn_grid(B, a.gx * a.H)), block(256);
  if (dh == 16) hipLaunchKernelGGL((attn_fwd_kernel<TC, 16>), grid, block, 0, s, a);
  else if (dh == 64) hipLaunchKernelGGL((attn_fwd_kernel<TC, 64>), grid, block, 0, s, a);
  else if (dh == 32) hipLaunchKernelGGL((attn_fwd_kernel<TC, 32>), grid, block, 0, s, a);      // 4 heads of a 128-wide model: the generic
  else if (dh == 128) hipLaunchKernelGGL((attn_fwd_kernel<TC, 128>), grid, block, 0, s, a);   // templates, not tuned (1 head)
  else { dx_set_error("attention: head dim %d unsupported (16, 32, 64, 128)", dh); return DX_ERR_UNSUPPORTED; }
  DX_LAUNCH_CHECK();
  return DX_OK;
}
template <typename TC>
int launch_bwd(const AttnArgs& a0, int B, int dh, float* delta, int algo, hipStream_t s) {
  AttnArgs a = a0;
  a.B = B;
  a.gx = dx_cdiv(a.N, dh >= 64 ? 128 / Split<64>::value : 128);
  const bool can_fuse = std::is_same<TC, bf16_t>::value && dh == 16 && a.N <= FB_MAXN;
  if (algo == DX_ATTN_FUSED && !can_fuse) {
    dx_set_error("dx_attention_bwd: the fused kernel needs bf16, d_head 16, N <= %d (got d_head %d, N %d)", FB_MAXN, dh, a.N);
    return DX_ERR_UNSUPPORTED;
  }
  if (can_fuse && (algo == DX_ATTN_FUSED || algo == DX_ATTN_AUTO)) {
    if (!a.counters) { dx_set_error("dx_attention_bwd: the fused kernel needs the arrival counters (dx_attention_bwd_counters(B, H) ints, zeroed once)"); return DX_ERR_ARG; }
    hipLaunchKernelGGL(attn_bwd_fused16_kernel, dim3(attn_grid(B, FB_PARTS * a.H)), dim3(FB_T), 0, s, a, delta + (long)B * a.H * a.N);
    DX_LAUNCH_CHECK();
    return DX_OK;
  }
  dim3 grid(attn_grid(B, a.gx * a.H)), block(256);
  if (dh == 16) {
    hipLaunchKernelGGL((attn_bwd_dq_kernel<TC, 16>), grid, block, 0, s, a);
    hipLaunchKernelGGL((attn_bwd_dkv_kernel<TC, 16>), grid, block, 0, s, a);
  } else if (dh == 64) {
    hipLaunchKernelGGL((attn_bwd_dq_kernel<TC, 64>), grid, block, 0, s, a);
    hipLaunchKernelGGL((attn_bwd_dkv_kernel<TC, 64>), grid, block, 0, s, a);
  } else if (dh == 32) {
    hipLaunchKernelGGL((attn_bwd_dq_kernel<TC, 32>), grid, block, 0, s, a);
    hipLaunchKernelGGL((attn_bwd_dkv_kernel<TC, 32>), grid, block, 0, s, a);
  } else if (dh == 128) {
    hipLaunchKernelGGL((attn_bwd_dq_kernel<TC, 128>), grid, block, 0, s, a);
    hipLaunchKernelGGL((attn_bwd_dkv_kernel<TC, 128>), grid, block, 0, s, a);
  } else { dx_set_error("attention: head dim %d unsupported (16, 32, 64, 128)", dh); return DX_ERR_UNSUPPORTED; }
  DX_LAUNCH_CHECK();
  return DX_OK;
}

// ---- longest-first launch order.  A workgroup's work is proportional to its utterance's length, all workgroups of a
// ragged batch are resident at once, and the hardware hands them to the CUs in blockIdx order: with the utterances in
// collate order the CUs that drew the long ones finish last while the rest idle (measured on a 1..1000-frame batch of 48:
// forward 84 -> 79 us for d_head = 16, 47 -> 40 us for d_head = 64; backward 236 -> 217 us).  One tiny launch per batch.
__global__ void length_order_kernel(const int64_t* __restrict__ lens, int B, int* __restrict__ order) {
  for (int i = threadIdx.x; i < B; i += blockDim.x) {
    const int64_t li = lens[i];
    int rank = 0;
    for (int j = 0; j < B; ++j) { const int64_t lj = lens[j]; rank += (lj > li) || (lj == li && j < i); }
    order[rank] = i;
  }
}
}  // namespace

extern "C" int dx_length_order(const int64_t* lengths, int B, int* order, void* stream) {
  DX_REQUIRE(lengths && order, DX_ERR_ARG, "dx_length_order: null pointer");
  DX_REQUIRE(B > 0 && B <= 65536, DX_ERR_SHAPE, "dx_length_order: B=%d (1..65536)", B);
  hipLaunchKernelGGL(length_order_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, lengths, B, order);
  DX_LAUNCH_CHECK();
  return DX_OK;
}

extern "C" int dx_attention_fwd(const void* qkv, int dtype, const int64_t* lengths, const int* order, void* o, float* lse, int B, int N,
                                int H, int E, float p_drop, uint64_t seed, const DxStepScalars* step, void* stream) {
  DX_REQUIRE(qkv && lengths && o, DX_ERR_ARG, "dx_attention_fwd: null pointer");
  DX_REQUIRE(B > 0 && N > 0 && H > 0 && E % H == 0, DX_ERR_SHAPE, "dx_attention_fwd: bad shape B=%d N=%d H=%d E=%d", B, N, H, E);
  DX_REQUIRE(p_drop >= 0.f && p_drop < 1.f, DX_ERR_ARG, "dx_attention_fwd: dropout p out of [0,1)");
  const int dh = E / H;
  AttnArgs a{qkv, o, lse, nullptr, nullptr, nullptr, lengths, N, H, E, 1.f / sqrtf((float)dh), p_drop, seed, order};
  a.step = step; a.counters = nullptr;
  if (dtype == DX_BF16) return launch_fwd<bf16_t>(a, B, dh, (hipStream_t)stream);
  if (dtype == DX_F32) return launch_fwd<float>(a, B, dh, (hipStream_t)stream);
  dx_set_error("dx_attention_fwd: bad dtype %d", dtype);
  return DX_ERR_DTYPE;
}

extern "C" long dx_attention_bwd_ws_floats(int B, int N, int H) {
  if (B <= 0 || N <= 0 || H <= 0) return 0;
  return (long)B * H * N + fb_ws_floats(B, N, H);
}

extern "C" long dx_attention_bwd_counters(int B, int H) { return (B <= 0 || H <= 0) ? 0 : (long)B * H; }

extern "C" int dx_attention_bwd(const void* qkv, const void* o, const void* d_o, int dtype, const float* lse,
                                const int64_t* lengths, const int* order, void* dqkv, float* delta_ws, int* counters, int B, int N, int H, int E,
                                float p_drop, uint64_t seed, const DxStepScalars* step, int algo, void* stream) {
  DX_REQUIRE(qkv && o && d_o && lse && lengths && dqkv && delta_ws, DX_ERR_ARG, "dx_attention_bwd: null pointer");
  DX_REQUIRE(B > 0 && N > 0 && H > 0 && E % H == 0, DX_ERR_SHAPE, "dx_attention_bwd: bad shape");
  DX_REQUIRE(algo >= DX_ATTN_AUTO && algo <= DX_ATTN_FUSED, DX_ERR_ARG, "dx_attention_bwd: algo %d", algo);
  const int dh = E / H;
  AttnArgs a{qkv, const_cast<void*>(o), const_cast<float*>(lse), d_o, delta_ws, dqkv, lengths, N, H, E,
             1.f / sqrtf((float)dh), p_drop, seed, order};
  a.step = step; a.counters = counters;
  if (dtype == DX_BF16) return launch_bwd<bf16_t>(a, B, dh, delta_ws, algo, (hipStream_t)stream);
  if (dtype == DX_F32) return launch_bwd<float>(a, B, dh, delta_ws, algo, (hipStream_t)stream);
  dx_set_error("dx_attention_bwd: bad dtype %d", dtype);
  return DX_ERR_DTYPE;
}
