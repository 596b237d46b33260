// K2 -- the position-wise feed-forward half of an FFT block in ONE launch (bf16 operands):
//
//   h = ReLU(conv_k3(a; W1, b1))            128 -> C hidden channels          (reference model.py:226-229)
//   z = conv_k3(h; W2, b2)                  C -> 128
//   u = [FiLM] LayerNorm(dropout(z) + a) [masked]                             (model.py:230-235, 262)
//
// The two-kernel path writes h (B, N, C) to HBM and reads it straight back (61 MB each way per frame-level block at
// B = 48); here a workgroup owns a tile of <= 126 positions and walks the hidden axis in chunks of 64 channels: the
// chunk of h it needs (tile + 1 halo row on either side, recomputed from the activation tile that sits in LDS for the
// workgroup's lifetime) goes from the conv-1 accumulators into LDS in the A-operand layout of conv 2 and is consumed
// from there.  h is still written ONCE (the backward pass needs it: ReLU gate, weight gradient of conv 2) but never
// read back in the forward pass, and conv 1 / conv 2 / the LayerNorm epilogue share one launch, one prologue and one
// pass over the tile plan.
//
// Structure (512 threads): waves 0-3 issue MFMAs only, waves 4-7 stream the weights.  Both weight matrices pass through
// a 3-stage LDS ring of 24 KB stages, filled by global_load_lds_dwordx4 (1 KiB pieces of 16 rows x 64 B, XOR swizzle on
// the source side, same image as conv_gemm.hip).  Per chunk c of 128 hidden channels there are 8 stages:
//   j = 0..3   W1[tap][128 c .. 128 c + 127][32 j .. 32 j + 31]       conv 1 partial over input channels 32 j ..
//   j = 4..7   W2[tap][0 .. 127][128 c + 32 (j - 4) ..]               conv 2 partial over hidden channels 128 c + 32 (j - 4) ..
// one workgroup barrier per stage (loaders arrive when their pieces of stage k have landed, MFMA waves when they are done
// with stage k - 1).  Conv 1 runs with swapped operands (D[channel][position]) so that a lane ends up with 4 consecutive
// hidden channels of one position: bias + ReLU + bf16 pack in registers, four ds_write_b64 per 32 x 32 tile into the
// swizzled A image of conv 2.  MFMA wave (wp, wc) owns a 2 x 2 block of 32 x 32 tiles in both GEMMs (positions 64 wp ..,
// channels 64 wc ..): four fragment reads feed four MFMAs, fragments of k-step s + 1 are requested before the MFMAs of s.
// Per tile a workgroup streams both weight matrices once (1.57 MB from L2) for 2 x 96 MFMAs per wave and chunk: 32 B/clk,
// right at what a CU fetches from L2 (~30 B/clk, tools/probes/lds_dma_rate_probe.hip), and reads 1 KB of fragments from LDS
// per MFMA (4 waves: 128 B/clk, the LDS peak) -- a tile of <= 126 rows does not allow a bigger register block.
//
// Tiles come from dx_ff_plan: every utterance cut into equal pieces of <= 126 rows, the batch padded to a multiple of 256
// workgroups; the padding rows [length, N) of the 128-wide outputs are zero-filled by the loader waves, an equal share per
// workgroup, and the last tile of an utterance zeroes the rows of h just past the utterance that the backward kernels'
// tiles may touch (they only ever multiply them by exact zeros, but the buffer comes uninitialised).
#include <type_traits>

#include "dx_common.h"

namespace {

constexpr int FF_THREADS = 512, FF_MAXH = 126, FF_HC = 128, FF_RING = 3, FF_D = 128;
constexpr int A_ROWS = 144, H_ROWS = 144, STAGE_ROWS = 384, STAGE_EL = STAGE_ROWS * 32;
constexpr int A_EL = 4 * A_ROWS * 32, H_EL = 4 * H_ROWS * 32, STG_LD = 132;
constexpr int RING_BYTES = FF_RING * STAGE_EL * 2, STG_BYTES = 128 * STG_LD * 4;
#ifndef FF_ABL
#define FF_ABL 0   // compile-time ablation (development): 1 no fragment reads / MFMAs, 2 no weight DMA, 4 no h store, 8 no finalize
#endif
constexpr int FF_MAXC = 2048;               // hidden channels (bias staged in LDS)
constexpr int FF_TAIL_ROWS = 130;            // rows of h past an utterance's end that a backward tile may read
constexpr int FF_ZERO_EL = 4096;
__device__ __attribute__((aligned(16))) unsigned short ff_zero_page[FF_ZERO_EL + 32];

struct FFArgs {
  const bf16_t* x;          // (B, N, 128) bf16: the attention sub-layer's output (GEMM operand copy)
  const bf16_t* w1;         // [3][C][128]   forward packing of convs.0
  const float* b1;          // (C)
  const bf16_t* w2;         // [3][128][C]   forward packing of convs.2
  const float* b2;          // (128)
  bf16_t* h;                // (B, N, C) out (saved for the backward pass)
  const int64_t* lengths;
  const float* residual;    // (B, N, 128) fp32
  const float* gamma; const float* beta; const float* film; long ldf;
  float* y; bf16_t* y_lp; float* s_out; float* mean; float* rstd;
  float p_pre; uint64_t seed_pre;
  const int4* plan; int plan_tiles;
  int B, N, C;
};

__device__ __forceinline__ int lds_at(int row, int chunk) { return row * 32 + ((chunk ^ ((row >> 2) & 3)) << 3); }

__device__ __forceinline__ void store8f(float* p, const float* v) {
  *reinterpret_cast<f32x4*>(p) = f32x4{v[0], v[1], v[2], v[3]};
  *reinterpret_cast<f32x4*>(p + 4) = f32x4{v[4], v[5], v[6], v[7]};
}
__device__ __forceinline__ void store8b(bf16_t* p, const float* v) {
  bf16x8 r;
#pragma unroll
  for (int e = 0; e < 8; ++e) r[e] = (bf16_t)v[e];
  *reinterpret_cast<bf16x8*>(p) = r;
}
__device__ __forceinline__ f32x8 load8f(const float* p) {
  const f32x4 lo = *reinterpret_cast<const f32x4*>(p), hi = *reinterpret_cast<const f32x4*>(p + 4);
  return f32x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

__device__ __forceinline__ void ff_wait_vmcnt(int n) {
  switch (n) {
#define FF_VMW(n) case n: asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory"); break;
    FF_VMW(1) FF_VMW(2) FF_VMW(3) FF_VMW(4) FF_VMW(5) FF_VMW(6) FF_VMW(7) FF_VMW(8) FF_VMW(9) FF_VMW(10) FF_VMW(11) FF_VMW(12)
#undef FF_VMW
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
}

__global__ __launch_bounds__(FF_THREADS, 1) void ff_fused_fwd_kernel(FFArgs p) {
  typedef bf16x8 frag_t;
  __shared__ __attribute__((aligned(16))) char smem[(A_EL + H_EL) * 2 + (RING_BYTES > STG_BYTES ? RING_BYTES : STG_BYTES)];
  __shared__ __attribute__((aligned(16))) float b1s[FF_MAXC];      // conv-1 bias: read per chunk in the accumulator layout
  bf16_t* As = reinterpret_cast<bf16_t*>(smem);
  bf16_t* Hs = As + A_EL;
  bf16_t* ring = Hs + H_EL;
  float* stage = reinterpret_cast<float*>(ring);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, g = lane >> 5;
  const int4 e = p.plan[blockIdx.x];
  const int b = e.x, n0 = e.y, h = e.z, fill_per = e.w;
  const int N = p.N, C = p.C;
  const int nchunks = C / FF_HC, S = nchunks * 8;
  if (h <= 0 && wave < 4) return;                         // an empty tile: only its loader waves work (padding fill)
  if (wave < 4)                                           // (visible to every MFMA wave after the first stage barrier)
    for (int i = tid; i < C; i += 256) b1s[i] = p.b1[i];
  const int len = (int)p.lengths[b] < 0 ? 0 : ((int)p.lengths[b] > N ? N : (int)p.lengths[b]);
  const bool last_tile = h > 0 && n0 + h >= len;          // owns the first padding row of h (model.py: pads are not masked between the convs)

  if (wave >= 4) {
    // ------------------------------------------------------------------ loader waves
    const int lw = __builtin_amdgcn_readfirstlane(wave) - 4;
    const int ltid = lw * 64 + lane;
    // activation tile: rows n0 - 2 .. n0 + h + 1 (two halo rows per side: one for conv 2's halo of h, one for conv 1), 4 K chunks
    if (h > 0) {
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int q = lw + 4 * t;                          // 36 pieces: K chunk q / 9, 16-row piece q % 9
        const int kc = q / 9, rp = q - kc * 9;
        const int r = rp * 16 + (lane >> 2);
        const int cc = (lane & 3) ^ ((r >> 2) & 3);
        const int n = n0 - 2 + r;
        const bf16_t* sp = reinterpret_cast<const bf16_t*>(ff_zero_page) + cc * 8;
        if (r < h + 4 && n >= 0 && n < N) sp = p.x + ((size_t)b * N + n) * FF_D + kc * 32 + cc * 8;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)sp,
                                         (__attribute__((address_space(3))) void*)(As + kc * A_ROWS * 32 + rp * 512), 16, 0, 0);
      }
    }
    // weight pieces of a stage: rows 16 (lw + 4 t) + lane / 4 of the 384-row image (tap = row / 128), t = 0..5
    const bf16_t* s1[6];
    const bf16_t* s2[6];
    int dst[6];
#pragma unroll
    for (int t = 0; t < 6; ++t) {
      const int r = (lw + 4 * t) * 16 + (lane >> 2);
      const int cc = (lane & 3) ^ ((r >> 2) & 3);
      const int tap = r >> 7, ch = r & 127;
      s1[t] = p.w1 + ((size_t)tap * C + ch) * FF_D + cc * 8;
      s2[t] = p.w2 + ((size_t)tap * FF_D + ch) * C + cc * 8;
      dst[t] = __builtin_amdgcn_readfirstlane((lw + 4 * t) * 512);
    }
    auto issue_stage = [&](int s, int buf) {
      if (FF_ABL & 2) return;
      const int c = s >> 3, j = s & 7;
      if (j < 4) {
        const size_t off = (size_t)c * FF_HC * FF_D + j * 32;
#pragma unroll
        for (int t = 0; t < 6; ++t)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(s1[t] + off),
                                           (__attribute__((address_space(3))) void*)(ring + buf * STAGE_EL + dst[t]), 16, 0, 0);
      } else {
        const size_t off = (size_t)c * FF_HC + (j - 4) * 32;
#pragma unroll
        for (int t = 0; t < 6; ++t)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(s2[t] + off),
                                           (__attribute__((address_space(3))) void*)(ring + buf * STAGE_EL + dst[t]), 16, 0, 0);
      }
    };
    if (h > 0) {
#pragma unroll
      for (int st = 0; st < FF_RING - 1; ++st)
        if (st < S) issue_stage(st, st);
    }
    // ---- padding fill: the batch's padding rows [length, N), flattened utterance by utterance, are split evenly over the
    // workgroups; this one owns [lo, hi).  Each loader wave finds the utterances its range touches with a wave scan.
    {
      const long lo = (long)blockIdx.x * fill_per, hi = lo + fill_per;
      long carry = 0;
      for (int base = 0; base < p.B && carry < hi; base += 64) {
        const int ub = base + lane;
        const int ulen = ub < p.B ? (int)p.lengths[ub] : N;
        const int dead = ub < p.B ? N - (ulen < 0 ? 0 : (ulen > N ? N : ulen)) : 0;
        int incl = dead;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
        const long ustart = carry + incl - dead, uend = carry + incl;
        const long fs = ustart > lo ? ustart : lo, fe = uend < hi ? uend : hi;
        unsigned long long todo = __ballot(fs < fe);
        while (todo) {
          const int src_lane = __ffsll((long long)todo) - 1;
          todo &= todo - 1;
          const int fb = base + src_lane;
          const int first = __shfl(N - dead + (int)(fs - ustart), src_lane, 64), cnt = __shfl((int)(fe - fs), src_lane, 64);
          const float z[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
          for (int c = ltid; c < cnt * (FF_D / 8); c += 256) {
            const int n = first + (c >> 4), cl = (c & 15) * 8;
            const size_t off = ((size_t)fb * N + n) * FF_D + cl;
            store8f(p.y + off, z);
            if (p.y_lp) store8b(p.y_lp + off, z);
            if (p.s_out) store8f(p.s_out + off, z);
            if (p.mean && cl == 0) { p.mean[(size_t)fb * N + n] = 0.f; p.rstd[(size_t)fb * N + n] = 0.f; }
          }
        }
        carry += __shfl(incl, 63, 64);
      }
    }
    if (h <= 0) return;
    if (last_tile) {     // rows of h past the utterance (the first one, row `len`, is computed by the MFMA waves)
      const float z[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      const int first = len + 1, cnt = min(N - first, FF_TAIL_ROWS);
      const int segs = C / 8;
      for (long c = ltid; c < (long)cnt * segs; c += 256) {
        const int n = first + (int)(c / segs), cl = (int)(c % segs) * 8;
        store8b(p.h + ((size_t)b * N + n) * C + cl, z);
      }
    }
    int nbuf = FF_RING - 1, k = 0;
    for (; k + FF_RING - 1 < S; ++k) {
      // (the fill's stores share the counter and may retire out of order with the loads: drain everything once)
      if (k == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else ff_wait_vmcnt(6 * (FF_RING - 2));
      __builtin_amdgcn_s_barrier();
      issue_stage(k + FF_RING - 1, nbuf);
      nbuf = nbuf + 1 == FF_RING ? 0 : nbuf + 1;
    }
    for (; k < S; ++k) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
  } else {
    // ------------------------------------------------------------------ MFMA waves: wave (wp, wc) owns a 2 x 2 block of tiles
    const int wp = wave >> 1, wc = wave & 1;
    // 32-row blocks of this wave that hold rows of the tile: hidden rows 0 .. h + 1, output rows 0 .. h - 1 (wave-uniform)
    const int nb1 = __builtin_amdgcn_readfirstlane(min(2, max(0, (h + 2 + 31) / 32 - 2 * wp)));
    const int nb2 = __builtin_amdgcn_readfirstlane(min(2, max(0, (h + 31) / 32 - 2 * wp)));
    f32x16 acc2[2][2], acc1[2][2];                            // [position block][channel tile]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc2[i][t][r] = 0.f; acc1[i][t][r] = 0.f; }
    // LDS offsets (elements) of this lane's fragments, loop-invariant.  Activation / hidden rows: one per (tap, k-step) -- the
    // swizzle term ((row >> 2) & 3) moves with the tap -- and the second position block is 32 rows = 1024 elements further
    // (a multiple of 4 rows: same swizzle, an immediate).  Weight rows: tap * 128 and tile * 32 are multiples of 4 rows, so
    // one offset per k-step and immediates for (tap, tile).
    int offA[6], offW[2];
#pragma unroll
    for (int st = 0; st < 6; ++st) offA[st] = lds_at(2 * wp * 32 + l31 + (st >> 1), (st & 1) * 2 + g);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) offW[ks] = lds_at(2 * wc * 32 + l31, ks * 2 + g);
    int buf = 0;
    for (int k = 0; k < S; ++k) {
      asm volatile("" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      const int c = k >> 3, j = k & 7;
      const bf16_t* Wr = ring + buf * STAGE_EL;
      if (j < 4) {
        // ---- conv 1, input channels 32 j ..: D[hidden channel][position] += W1 * a^T
        const bf16_t* Ak = As + j * A_ROWS * 32;
        auto run1 = [&](auto nb_tag) {
          constexpr int NB = decltype(nb_tag)::value;
          if constexpr (NB > 0 && !(FF_ABL & 1)) {
            frag_t af[2][NB], wf[2][2];
            auto load = [&](int st, frag_t* a_, frag_t* w_) {
#pragma unroll
              for (int i = 0; i < NB; ++i) a_[i] = *reinterpret_cast<const frag_t*>(&Ak[offA[st] + i * 1024]);
#pragma unroll
              for (int t = 0; t < 2; ++t) w_[t] = *reinterpret_cast<const frag_t*>(&Wr[offW[st & 1] + ((st >> 1) * 128 + t * 32) * 32]);
            };
            load(0, af[0], wf[0]);
#pragma unroll
            for (int st = 0; st < 6; ++st) {
              if (st + 1 < 6) load(st + 1, af[(st + 1) & 1], wf[(st + 1) & 1]);
              __builtin_amdgcn_sched_barrier(0);
#pragma unroll
              for (int i = 0; i < NB; ++i)
#pragma unroll
                for (int t = 0; t < 2; ++t) dx_mma(acc1[i][t], wf[st & 1][t], af[st & 1][i]);
            }
            if (j == 3 && !(FF_ABL & 8)) {   // the chunk of h is complete: bias + ReLU; rows outside the utterance are conv 2's zero padding
#pragma unroll
              for (int i = 0; i < NB; ++i) {
                const int hr = (2 * wp + i) * 32 + l31, npos = n0 - 1 + hr;
                const bool pos_ok = npos >= 0 && npos < N;
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                  const int kc = 2 * wc + t;                    // K chunk of conv 2 = 32-channel tile of the hidden chunk
#pragma unroll
                  for (int q = 0; q < 4; ++q) {
                    const f32x4 bv = *reinterpret_cast<const f32x4*>(&b1s[c * FF_HC + kc * 32 + 8 * q + 4 * g]);
                    bf16x4 o;
#pragma unroll
                    for (int e2 = 0; e2 < 4; ++e2) {
                      const float v = fmaxf(acc1[i][t][4 * q + e2] + bv[e2], 0.f);
                      o[e2] = (bf16_t)(pos_ok ? v : 0.f);
                      acc1[i][t][4 * q + e2] = 0.f;
                    }
                    *reinterpret_cast<bf16x4*>(&Hs[kc * H_ROWS * 32 + lds_at(hr, q) + 4 * g]) = o;
                  }
                }
              }
            }
          }
        };
        if (nb1 >= 2) run1(std::integral_constant<int, 2>{});
        else if (nb1 == 1) run1(std::integral_constant<int, 1>{});
      } else {
        if (!(FF_ABL & 4)) {   // h chunk c is in LDS (the barrier of stage j = 4): save it for the backward pass, 16 bytes per
          // lane, a quarter of the rows in each of the four conv-2 stages so that the stores trickle out beside the MFMAs
#pragma unroll
          for (int it = 0; it < 2; ++it) {
            const int row = ((j - 4) * 2 + it) * 16 + (tid >> 4), seg = tid & 15;
            const int n = n0 - 1 + row;
            if (row >= 1 && row <= h + (last_tile ? 1 : 0) && n < N) {
              const bf16x8 v = *reinterpret_cast<const bf16x8*>(&Hs[(seg >> 2) * H_ROWS * 32 + lds_at(row, seg & 3)]);
              *reinterpret_cast<bf16x8*>(p.h + ((size_t)b * N + n) * C + c * FF_HC + seg * 8) = v;
            }
          }
        }
        // ---- conv 2, hidden channels 128 c + 32 (j - 4) ..: D[position][out channel] += h * W2^T
        const bf16_t* Hk = Hs + (j - 4) * H_ROWS * 32;
        auto run2 = [&](auto nb_tag) {
          constexpr int NB = decltype(nb_tag)::value;
          if constexpr (NB > 0 && !(FF_ABL & 1)) {
            frag_t hf[2][NB], wf[2][2];
            auto load = [&](int st, frag_t* a_, frag_t* w_) {
#pragma unroll
              for (int i = 0; i < NB; ++i) a_[i] = *reinterpret_cast<const frag_t*>(&Hk[offA[st] + i * 1024]);
#pragma unroll
              for (int t = 0; t < 2; ++t) w_[t] = *reinterpret_cast<const frag_t*>(&Wr[offW[st & 1] + ((st >> 1) * 128 + t * 32) * 32]);
            };
            load(0, hf[0], wf[0]);
#pragma unroll
            for (int st = 0; st < 6; ++st) {
              if (st + 1 < 6) load(st + 1, hf[(st + 1) & 1], wf[(st + 1) & 1]);
              __builtin_amdgcn_sched_barrier(0);
#pragma unroll
              for (int i = 0; i < NB; ++i)
#pragma unroll
                for (int t = 0; t < 2; ++t) dx_mma(acc2[i][t], hf[st & 1][i], wf[st & 1][t]);
            }
          }
        };
        if (nb2 >= 2) run2(std::integral_constant<int, 2>{});
        else if (nb2 == 1) run2(std::integral_constant<int, 1>{});
      }
      buf = buf + 1 == FF_RING ? 0 : buf + 1;
    }
    __syncthreads();                                           // every wave is done with the ring: the epilogue stages through it
    // bias in the MFMA layout, accumulators -> LDS stage (128 rows x 128 channels, fp32)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if (i < nb2) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const int cl = (2 * wc + t) * 32 + l31;
          const float bv = p.b2[cl];
#pragma unroll
          for (int r = 0; r < 16; ++r) stage[((2 * wp + i) * 32 + dx_acc_row(r, g)) * STG_LD + cl] = acc2[i][t][r] + bv;
        }
      }
    }
  }
  if (wave >= 4) __syncthreads();                              // (matches the MFMA waves' barrier after the main loop)
  __syncthreads();
  // ---- LayerNorm epilogue, all 512 threads: 16 lanes hold one complete 128-channel row (same arithmetic, same dropout
  // counters as the LayerNorm epilogue of conv_gemm.hip)
#pragma unroll
  for (int pass = 0; pass < 4; ++pass) {
    const int sr = (tid >> 4) + pass * 32, cl = (tid & 15) * 8;
    const int n = n0 + sr;
    if (sr >= h || n >= N) continue;
    float v[8];
    {
      const f32x4 lo = *reinterpret_cast<const f32x4*>(&stage[sr * STG_LD + cl]);
      const f32x4 hi = *reinterpret_cast<const f32x4*>(&stage[sr * STG_LD + cl + 4]);
      v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3]; v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
    }
    const size_t rowg = (size_t)b * N + n, offl = rowg * FF_D + cl;
    if (p.p_pre > 0.f) {
      const uint32_t th = (uint32_t)(p.p_pre * 4294967296.0), key = dx_key32(p.seed_pre, 0);
      const float sc = 1.f / (1.f - p.p_pre);
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] = dx_keep(key, (uint32_t)rowg * FF_D + cl + q, th) ? v[q] * sc : 0.f;
    }
    {
      const f32x8 r = load8f(p.residual + offl);
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] += r[q];
    }
    if (p.s_out) store8f(p.s_out + offl, v);
    float sum = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) sum += v[q];
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) sum += __shfl_xor(sum, o, 64);
    const float mean = sum * (1.f / FF_D);
    float sq = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) { const float d = v[q] - mean; sq += d * d; }
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) sq += __shfl_xor(sq, o, 64);
    const float rstd = rsqrtf(sq * (1.f / FF_D) + 1e-5f);
    if (p.mean && cl == 0) { p.mean[rowg] = mean; p.rstd[rowg] = rstd; }
    const f32x8 gm = load8f(p.gamma + cl), bt = load8f(p.beta + cl);
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = (v[q] - mean) * rstd * gm[q] + bt[q];
    if (p.film) {
      const f32x8 fg = load8f(p.film + (size_t)b * p.ldf + cl), fb = load8f(p.film + (size_t)b * p.ldf + FF_D + cl);
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] = fg[q] * v[q] + fb[q];
    }
    if (n >= len) {
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] = 0.f;
    }
    store8f(p.y + offl, v);
    if (p.y_lp) store8b(p.y_lp + offl, v);
  }
}

// ---- tile plan: every utterance cut into equal pieces of <= 126 rows; the number of non-empty tiles is rounded up to a
// multiple of the 256 CUs and the tallest piece made as short as that allows; entries past it are empty (padding fill only)
constexpr int FF_NUM_CU = 256;
__global__ __launch_bounds__(64) void ff_plan_kernel(const int64_t* __restrict__ lens, int B, int N, int T, int4* __restrict__ table) {
  __shared__ int first[4096 + 1];
  const int lane = threadIdx.x;
  auto len_of = [&](int b) { const int l = (int)lens[b]; return l < 0 ? 0 : (l > N ? N : l); };
  auto tiles_at = [&](int H) {
    int c = 0;
    for (int b = lane; b < B; b += 64) c += (len_of(b) + H - 1) / H;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
    return c;
  };
  const int need = tiles_at(FF_MAXH);
  int Teff = (need + FF_NUM_CU - 1) / FF_NUM_CU * FF_NUM_CU;
  if (Teff > T) Teff = T;
  if (Teff < 1) Teff = 1;
  int lo = 1, hi = FF_MAXH;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (tiles_at(mid) <= Teff) hi = mid; else lo = mid + 1;
  }
  const int H = lo;
  if (lane == 0) {
    int acc = 0;
    for (int b = 0; b < B; ++b) { first[b] = acc; acc += (len_of(b) + H - 1) / H; }
    first[B] = acc;
  }
  __syncthreads();
  long dead = 0;
  for (int b = lane; b < B; b += 64) dead += N - len_of(b);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) dead += __shfl_xor(dead, o, 64);
  const int per = (int)((dead + T - 1) / T);
  for (int b = lane; b < B; b += 64) {
    const int l = len_of(b), t = first[b + 1] - first[b];
    if (t == 0) continue;
    const int hb = (l + t - 1) / t;
    for (int j = 0; j < t; ++j) {
      const int n0 = j * hb, rows = l - n0 < hb ? l - n0 : hb;
      if (first[b] + j < T) table[first[b] + j] = make_int4(b, n0, rows > 0 ? rows : 0, per);
    }
  }
  for (int i = first[B] + lane; i < T; i += 64) table[i] = make_int4(0, 0, 0, per);
}

}  // namespace

extern "C" int dx_ff_plan_size(int B, int N) {
  if (B <= 0 || N <= 0) return 0;
  const long worst = (long)B * dx_cdiv(N, FF_MAXH);
  return (int)((worst + FF_NUM_CU - 1) / FF_NUM_CU * FF_NUM_CU);
}

extern "C" int dx_ff_plan(const int64_t* lengths, int B, int N, int n_tiles, int* table, void* stream) {
  DX_REQUIRE(lengths && table, DX_ERR_ARG, "dx_ff_plan: null pointer");
  DX_REQUIRE(B > 0 && B <= 4096 && N > 0, DX_ERR_SHAPE, "dx_ff_plan: B=%d (1..4096), N=%d", B, N);
  DX_REQUIRE(n_tiles >= dx_ff_plan_size(B, N), DX_ERR_ARG, "dx_ff_plan: n_tiles=%d < dx_ff_plan_size=%d", n_tiles, dx_ff_plan_size(B, N));
  hipLaunchKernelGGL(ff_plan_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, lengths, B, N, n_tiles, reinterpret_cast<int4*>(table));
  DX_LAUNCH_CHECK();
  return DX_OK;
}

extern "C" int dx_ff_fused_fwd(const void* x_lp, const void* w1_packed, const float* b1, const void* w2_packed, const float* b2,
                               const float* residual, const float* gamma, const float* beta, const float* film, long ldf,
                               const int64_t* lengths, void* h_out, float* y, void* y_lp, float* s_out, float* mean, float* rstd,
                               int B, int N, int C, float p_pre, uint64_t seed_pre, const int* plan, int plan_tiles, void* stream) {
  DX_REQUIRE(x_lp && w1_packed && b1 && w2_packed && b2 && residual && gamma && beta && lengths && h_out && y && plan, DX_ERR_ARG,
             "dx_ff_fused_fwd: null pointer");
  DX_REQUIRE(B > 0 && N > 0 && C >= FF_HC && C % FF_HC == 0 && C <= FF_MAXC, DX_ERR_SHAPE, "dx_ff_fused_fwd: B=%d N=%d C=%d (C a multiple of 128, <= 2048)", B, N, C);
  DX_REQUIRE((mean == nullptr) == (rstd == nullptr), DX_ERR_ARG, "dx_ff_fused_fwd: mean and rstd come together");
  DX_REQUIRE(p_pre >= 0.f && p_pre < 1.f, DX_ERR_ARG, "dx_ff_fused_fwd: dropout p out of [0,1)");
  DX_REQUIRE(plan_tiles >= dx_ff_plan_size(B, N), DX_ERR_ARG, "dx_ff_fused_fwd: plan_tiles=%d < dx_ff_plan_size(B, N)=%d", plan_tiles,
             dx_ff_plan_size(B, N));
  FFArgs a{reinterpret_cast<const bf16_t*>(x_lp), reinterpret_cast<const bf16_t*>(w1_packed), b1,
           reinterpret_cast<const bf16_t*>(w2_packed), b2, reinterpret_cast<bf16_t*>(h_out), lengths, residual, gamma, beta, film, ldf,
           y, reinterpret_cast<bf16_t*>(y_lp), s_out, mean, rstd, p_pre, seed_pre, reinterpret_cast<const int4*>(plan), plan_tiles, B, N, C};
  hipLaunchKernelGGL(ff_fused_fwd_kernel, dim3((unsigned)plan_tiles), dim3(FF_THREADS), 0, (hipStream_t)stream, a);
  DX_LAUNCH_CHECK();
  return DX_OK;
}
