// K1 / K3 / K12 -- k-tap conv on channel-last activations as an implicit GEMM on MFMA (gfx950).
//
//   Y[b, n, co] = bias[co] + sum_{tap, ci} X[b, n + tap - taps/2, ci] * W[tap][co][ci]
//
// Tiling: one workgroup (4 MFMA waves) computes a 64 / 128 / 256-position x 128-channel tile of ONE utterance, so
// the conv halo is simply rows n0-1 .. n0+rows of that utterance (rows outside [0, N) are zero padding).
// The K loop walks Cin in chunks of 32; per chunk the haloed activation tile and the weight tile (taps x 128 x 32)
// are staged in LDS (bf16: unpadded XOR-swizzled rows, fp32: rows padded by 16 B -> conflict-free ds_read_b128
// fragment reads), and every tap reuses the same activation tile at a row offset -- no im2col is materialised.
// Each wave owns a (32 MI) x 64 sub-tile = MI x 2 MFMA 32x32 accumulators.
// Operand type TC: bf16 (v_mfma_f32_32x32x16_bf16) or fp32 (v_mfma_f32_32x32x2_f32, exact fp32 mode).
// Variants in this file: conv_gemm_kernel (register-staged pipeline; RING: LDS-DMA ring with loader waves, and on top
// of it the balanced variable-height tiles of dx_conv_tile_plan), conv_wreg_kernel (weights in registers, K <= 384),
// conv_sk_kernel (split-K inside the workgroup + LayerNorm epilogues), conv_wide_kernel (256 x 256 tiles).
// The weight gradients live in conv_wgrad.hip, the weight packers in conv_pack.hip, shared helpers in conv_common.h.
#include <stdlib.h>

#include <type_traits>

#include "dx_common.h"

namespace {

constexpr int BN = 128, NTHREADS = 256;

#include "conv_common.h"

// LayerNorm epilogue (Cout == 128: a tile holds complete rows): s = dropout(conv) + residual; y = LN(s) [* FiLM] [masked]
// (LN template parameter of conv_gemm_kernel: 0 none, 1 forward LayerNorm, 2 backward LayerNorm)
// Backward (dx_conv1d_lnbwd, the data-gradient GEMM that completes dL/dy of a LayerNorm carries that LayerNorm's
// backward): y = residual gradient in / ds out (in place), y_lp = bf16 dx_pre out, s_out = the saved LayerNorm
// input (read), mean / rstd read, dgamma / dbeta / dfilm accumulated with one atomic per channel per workgroup.
struct LNEpi {
  const float* gamma; const float* beta; const float* residual; const float* film; long ldf;
  float* y; void* y_lp; float* s_out; float* mean; float* rstd;
  float p_pre; uint64_t seed_pre;
  int enabled;
  float* dgamma; float* dbeta; float* dfilm; long lddf;
  const void* w2; void* y2;   // split-K kernel: y2 = y_lp . w2^T (+ b2), a 128 -> n2 k = 1 GEMM on the rows the epilogue has just produced
  const float* b2; int n2;    // (n2 = 128: output-projection data gradient behind the LayerNorm backward; 384: the next block's QKV projection)
  const DxStepScalars* step;  // NULL, or the device-side step block whose salt is added to seed_pre (captured steps)
  // "virtual" residual (split-K forward kernel, dx_conv1d_ln_vres): `residual` is the saved INPUT s of the LayerNorm that produced the
  // residual stream, and the epilogue re-applies that LayerNorm (+ mask) -- its fp32 output then never has to be stored (y = NULL there)
  const float* res_mean; const float* res_rstd; const float* res_gamma; const float* res_beta;
};

struct ConvArgs {
  const void* x; long ldx;
  const void* w; const float* bias;
  void* y; long ldy;
  const void* gate;
  const int64_t* mask_len;
  const int64_t* skip_len;
  int N, Cin, Cout, flags, B;
  LNEpi ln;
  const int* plan; int plan_tiles;     // balanced position tiles {b, n0, rows, 0} (dx_conv_tile_plan), ring kernels only
  const void* w_frag;                  // the same weights in MFMA-fragment order (dx_pack_frag_major): split-K kernel, or NULL
  uint32_t* relu_bits;                 // conv_wreg_kernel<BITS>: sign bits of the ReLU output, (B, Cout / 32, N) words -- written (RELU) ...
  const uint32_t* gate_bits;           // ... or read as the gate of the data gradient (GATE) instead of the activation itself
};

// Pipeline: the global loads of K-chunk k+1 are issued into registers (raw element type, converted only when they are
// written to LDS) BEFORE the MFMAs of chunk k, so HBM/L2 latency hides under the matrix work; one LDS buffer, two
// barriers per chunk.  Epilogue: accumulators are staged through LDS (reusing the operand buffers) 64 rows at a time
// and leave as whole 16-byte row segments (16 lanes cover a 128-channel row) -- the MFMA C layout would otherwise
// emit 64 two-byte stores per lane.
#ifndef DX_CONV_WPS
#define DX_CONV_WPS 2
#endif
// MI = 32-row MFMA tiles per wave along the position axis: 2 -> 128-row workgroup tile; 1 -> 64-row tile, used when
// Cout <= 128 (one channel tile): twice the workgroups for the GEMMs whose grid would otherwise under-fill 256 CUs.
// BK = channels per K chunk: 32, or 64 for the narrow-output kernels whose long serial K loop is latency-bound.
#ifndef DX_CONV_WPS_NARROW
#define DX_CONV_WPS_NARROW 4
#endif
// LNM: 0 none, 1 forward LayerNorm, 2 backward LayerNorm with FiLM gradients, 3 backward LayerNorm without FiLM
// LDS-DMA ring pipeline (conv_gemm_kernel<..., RING>): bf16 operands, long contractions.  OPT-IN (DX_CONV_RING=1):
// measured on MI355X it ties with the register-staged pipeline (1024 -> 1024 k3: 814 vs 818 TFLOP/s; 1024 -> 128 k3
// 55 vs 61 us plain, 78 vs 72 us with the LayerNorm epilogue; training step 9.35 vs 9.36 ms) because neither is bound by
// its pipeline: a CU fetches at most ~30 B/clk from L2 (tools/probes/lds_dma_rate_probe.hip: 16-17 TB/s chip-wide for
// global_load_lds_dwordx4, 14 TB/s for loads to registers, independent of row width and of the number of pieces in
// flight), a 128 x 128 x (3 x 32) chunk needs 36 KB for 768 MFMA cycles = 47 B/clk, and the ablations of this kernel
// give 306 us with the MFMA waves idle, 237 us with the loaders idle, 367 us together.  Past ~800 TFLOP/s the lever
// is bytes per FLOP per CU (taller position tiles when the batch has enough of them), not the pipeline.
#ifndef DX_PLAN_RING
#define DX_PLAN_RING 3   // stages of the balanced-tile (plan) kernels: 3 x 41 KB (2: main loop 39.5 vs 36.1 us)
#endif
// RING: 0 = register-staged single-buffer pipeline; S >= 2 = S-stage LDS ring filled by four loader waves (512 threads, bf16)
#ifndef CG_K1_BK
#define CG_K1_BK 32   // K chunk of the LayerNorm-fused k = 1 GEMMs (64: A/B build; the operand image stays below the epilogue stage)
#endif
#ifndef CG_K1_PF
#define CG_K1_PF 1   // chunks in flight of the register-staged k = 1 GEMMs (2: measured +-0, 29.8 vs 30.5 us / 20.2 vs 19.6 us: the chunk period is its barrier / LDS chain, not the global round trip)
#endif
template <typename TA, typename TC, typename TO, typename TG, int TAPS, int MI, int BK, int LNM = 0, int RING = 0>
// (fp32 activations feeding bf16 MFMAs at k = 3 -- instantiations off the bf16 step path, the LayerNorm kernels hand the GEMMs bf16 copies --
// prefetch their K chunk as raw fp32: one wave per SIMD less than the bf16-input form instead of 28-52 bytes of scratch)
__global__ __launch_bounds__(RING ? 2 * NTHREADS : NTHREADS, RING ? 2 : ((sizeof(TA) == 4 && sizeof(TC) == 2 && TAPS == 3) ? (MI == 1 ? DX_CONV_WPS_NARROW - 1 : 1) : (MI == 1 ? DX_CONV_WPS_NARROW : DX_CONV_WPS))) void conv_gemm_kernel(ConvArgs p) {
  constexpr int LN = LNM == 3 ? 2 : LNM;
  constexpr bool LNFILM = LNM == 2;
  constexpr int BM = 64 * MI, KC = BK / 8;   // KC = 8-element chunks per row of a K chunk
  constexpr int HALO = TAPS / 2;
  constexpr int AROWS = BM + TAPS - 1;
  // LDS image of the operand tiles.  bf16 (BK = 32: four 16-byte chunks per row): NO padding, chunk c of row r sits at
  // chunk position c ^ ((r >> 2) & 3) -- the 16 rows of a ds_read_b128 lane group then cover all 64 banks, and the
  // 8 lanes of a ds_write_b128 group (2 rows x 4 chunks) cover 32 distinct banks.  (The padded 80-byte rows read
  // conflict-free but staged with 2-way write conflicts: SQ_LDS_BANK_CONFLICT was 30 % of the LDS cycles of a kernel
  // whose LDS pipe -- ds_write_b128 of the weight tile above all -- is busier than its matrix pipe.)  fp32: padded rows.
  constexpr bool SWZ = sizeof(TC) == 2 && BK == 32;
  constexpr int LDS_K = SWZ ? BK : BK + Pad<TC>::value;
  auto lds_at = [](int row, int chunk) { return SWZ ? row * LDS_K + ((chunk ^ ((row >> 2) & 3)) << 3) : row * LDS_K + (chunk << 3); };
  constexpr int A_CH = AROWS * (BK / 8), A_PT = (A_CH + NTHREADS - 1) / NTHREADS;
  constexpr int W_PT = TAPS * BN * (BK / 8) / NTHREADS;
  constexpr int STG_LD = BN + 4;
  // ring image of one K chunk: activation rows rounded up to whole 16-row DMA pieces, then the taps x 128 weight rows
  constexpr int AR16 = (AROWS + 15) & ~15, STAGE_EL = (AR16 + TAPS * BN) * 32;
  constexpr int OPER_BYTES = RING ? RING * STAGE_EL * 2 : (AROWS + TAPS * BN) * LDS_K * (int)sizeof(TC), STG_BYTES = 64 * STG_LD * 4 * (RING > 0 && MI == 4 ? 2 : 1);
  typedef typename Vec8<TC>::type frag_t;
  typedef typename VecN<TA, 8>::type raw_t;
  __shared__ __attribute__((aligned(16))) char smem[OPER_BYTES > STG_BYTES ? OPER_BYTES : STG_BYTES];
  TC* As = reinterpret_cast<TC*>(smem);
  TC* Ws = As + AROWS * LDS_K;
  float* stage = reinterpret_cast<float*>(smem);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, g = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  // XCD-aware, weight-stationary order.  Workgroup L runs on XCD L % 8 (observed dispatch order).  Every workgroup
  // streams the whole weight slice of its channel tile (taps x 128 x Cin, up to 786 KB) through LDS, so the slice must
  // stay in that XCD's 4 MB L2 across workgroups: each XCD walks ALL of its position tiles for channel tile 0, then
  // for channel tile 1, ...  (Measured: no difference vs the channel-tile-fastest order on MI355X -- the kernel is bound
  // by its LDS->MFMA issue pattern at ~800 TFLOP/s, the known ceiling of a 128x128-tile two-barrier structure -- but this
  // order keeps the weight working set of an XCD at one slice, which matters once the inner loop gets faster.)
  // PLAN (ring kernels with 256-row tiles, one channel tile): the position tiles come from a table that cuts every
  // utterance into equal pieces of <= 256 rows such that the whole batch is a multiple of 256 workgroups of (nearly) the
  // same height -- a workgroup costs one pass over the weights whatever its height.  The padding rows [length, N) of the
  // batch are zero-filled by the loader waves, an equal share per workgroup, while the first chunks are in flight.
  constexpr bool PLAN = RING > 0 && MI == 4 && LNM != 0;
  int n0, b, co0, h = BM;              // h = rows of this tile
  int fill_per = 0;                    // PLAN: padding rows (flattened over the batch) this workgroup zero-fills
  if constexpr (PLAN) {
    co0 = 0;
    const int4 e = reinterpret_cast<const int4*>(p.plan)[blockIdx.x];
    b = e.x; n0 = e.y; h = e.z; fill_per = e.w;
    if (h <= 0 && threadIdx.x < NTHREADS) return;       // an empty tile: only its loader waves work (padding fill)
  } else {
    const int ztiles = dx_cdiv(p.Cout, BN), ptiles = dx_cdiv(p.N, BM);
    const int Lid = blockIdx.x, jj = Lid >> 3;
    const int per_xcd = (ptiles * p.B + 7) >> 3;         // position tiles owned by one XCD
    const int pt = (Lid & 7) + 8 * (jj % per_xcd);
    if (pt >= ptiles * p.B) return;
    n0 = (pt % ptiles) * BM; b = pt / ptiles; co0 = (jj / per_xcd) * BN;
    (void)ztiles;
  }
  const int N = p.N, Cin = p.Cin, Cout = p.Cout;
  const TA* X = reinterpret_cast<const TA*>(p.x) + (size_t)b * N * p.ldx;
  const TC* W = reinterpret_cast<const TC*>(p.w);
  const bool relu = p.flags & DX_CONV_RELU, trans = p.flags & DX_CONV_TRANSPOSED_OUT, accum = p.flags & DX_CONV_ACCUMULATE;
  const int len = p.mask_len ? (int)p.mask_len[b] : N;
  TO* Y = reinterpret_cast<TO*>(p.y);
  const TG* G = reinterpret_cast<const TG*>(p.gate);
  const bool vec_out = !trans && (Cout % 8 == 0) && (p.ldy % 8 == 0);

  // padding early-out: a tile that starts past length + conv halo cannot reach a valid output -> zeros, no MFMA
  if (!PLAN && p.skip_len && n0 >= (int)p.skip_len[b] + 2) {
    if (RING && tid >= NTHREADS) return;             // loader waves
    if (!trans && n0 >= dx_fill_end((int)p.skip_len[b], N)) return;   // past the fill end: nobody reads these rows (dx_common.h); the
                                                                        // transposed (B, C, N) form is the user-visible mel: fully padded
    if (LN == 2) {   // incoming residual gradient rows are zero there and stay; the bf16 dx_pre rows must exist as zeros
      float z[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      for (int c = tid; c < BM * (BN / 8); c += NTHREADS) {
        const int n = n0 + (c >> 4), cl = (c & 15) * 8;
        if (n >= N) continue;
        const size_t off = ((size_t)b * N + n) * BN + cl;
        store8<float>(p.ln.y + off, z);
        store8<bf16_t>(reinterpret_cast<bf16_t*>(p.ln.y_lp) + off, z);
      }
      return;
    }
    if (accum) return;
    if (LN == 1) {
      float z[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      for (int c = tid; c < BM * (BN / 8); c += NTHREADS) {
        const int n = n0 + (c >> 4), cl = (c & 15) * 8;
        if (n >= N) continue;
        const size_t off = ((size_t)b * N + n) * BN + cl;
        if (p.ln.y) store8<float>(p.ln.y + off, z);
        if (p.ln.y_lp) store8<bf16_t>(reinterpret_cast<bf16_t*>(p.ln.y_lp) + off, z);
        if (p.ln.s_out) store8<float>(p.ln.s_out + off, z);
        if (p.ln.mean && cl == 0) { p.ln.mean[(size_t)b * N + n] = 0.f; p.ln.rstd[(size_t)b * N + n] = 0.f; }
      }
      return;
    }
    if (vec_out) {
      float z[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      for (int c = tid; c < BM * (BN / 8); c += NTHREADS) {
        const int n = n0 + (c >> 4), co = co0 + (c & 15) * 8;
        if (n < N && co < Cout) store8<TO>(Y + ((size_t)b * N + n) * p.ldy + co, z);
      }
    } else {
      for (int c = tid; c < BM * BN; c += NTHREADS) {
        const int n = n0 + (trans ? c % BM : c / BN), co = co0 + (trans ? c / BM : c % BN);
        if (n < N && co < Cout) Y[trans ? ((size_t)b * Cout + co) * p.ldy + n : ((size_t)b * N + n) * p.ldy + co] = (TO)0.f;
      }
    }
    return;
  }

  f32x16 acc[MI][2];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  if constexpr (RING > 0) {
    // ---- LDS-DMA ring with dedicated loader waves (bf16 operands, Cin % 32 == 0; 512-thread workgroup).
    // Waves 0-3 run the MFMAs exactly as in the register-staged pipeline; waves 4-7 (one per SIMD, next to an MFMA
    // wave) only move data: measured with the MFMA waves issuing their own loads, a 1-wave-per-SIMD workgroup spends
    // more time ISSUING global -> LDS pieces (~100 cycles each, in order with its MFMAs) than the MFMAs take.
    // A K chunk's image is (AR16 + TAPS * 128) rows of 64 bytes in the swizzled layout of lds_at, written by
    // global_load_lds_dwordx4 pieces of 16 rows (1 KiB per wave instruction; LDS destination = piece base + 16 * lane,
    // so the swizzle is applied to each lane's SOURCE chunk).  Piece q belongs to loader q % 4; rows outside the
    // utterance / beyond Cout read a zero page.
    // Per chunk k, ONE workgroup barrier: a loader arrives after its pieces of chunk k have landed (counted vmcnt: the
    // RING - 2 younger chunks stay in flight), an MFMA wave after it has finished reading chunk k - 1.  Past the
    // barrier the MFMA waves read chunk k and the loaders refill the buffer chunk k - 1 just left with chunk k + RING - 1.
    static_assert(sizeof(TA) == 2 && sizeof(TC) == 2 && BK == 32, "ring pipeline: bf16 operands, 32-channel chunks");
    // pieces of a chunk: the first nA cover the h + TAPS - 1 activation rows of this tile, then TAPS * 8 weight pieces
    constexpr int W_INS = TAPS * BN / 16, MAXP = (AR16 / 16 + W_INS + 3) / 4, NSTEP = TAPS * 2;
    const int nA = (h + TAPS - 1 + 15) >> 4, nP = nA + W_INS;
    TC* ring = reinterpret_cast<TC*>(smem);
    const int nk = Cin >> 5;
    if (wave >= 4) {
      const int lw = __builtin_amdgcn_readfirstlane(wave) - 4;
      const int mine = (nP - lw + 3) >> 2;                       // pieces lw, lw + 4, ... of every chunk are this loader's
      const TC* src[MAXP];
      int dst[MAXP];
#pragma unroll
      for (int t = 0; t < MAXP; ++t) {
        const int q = lw + 4 * t;
        const bool isw = q >= nA;
        const int r = (isw ? q - nA : q) * 16 + (lane >> 2);     // row of the activation / weight image this lane fills
        const int c = (lane & 3) ^ ((r >> 2) & 3);                // source chunk that belongs at position lane & 3
        const TC* sp = reinterpret_cast<const TC*>(dx_zero_page) + c * 8;
        const int n = n0 + r - HALO, co = co0 + (r & (BN - 1));
        const TC* xa = reinterpret_cast<const TC*>(X) + (long)n * p.ldx + c * 8;
        const TC* wa = W + ((size_t)(r / BN) * Cout + co) * Cin + c * 8;
        sp = (!isw && r < h + TAPS - 1 && n >= 0 && n < N) ? xa : sp;
        sp = (isw && q < nP && co < Cout) ? wa : sp;
        src[t] = sp;
        dst[t] = __builtin_amdgcn_readfirstlane((isw ? AR16 / 16 + q - nA : q) * 512);
      }
      auto issue_chunk = [&](int kc, int buf) {
#pragma unroll
        for (int t = 0; t < MAXP; ++t)
          if (t < mine)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[t] + kc * 32),
                                             (__attribute__((address_space(3))) void*)(ring + buf * STAGE_EL + dst[t]), 16, 0, 0);
      };
      auto wait_landed = [&](int keep) {                         // s_waitcnt vmcnt(keep), keep wave-uniform
        switch (keep) {
#define DX_VMW(n) case n: asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory"); break;
          DX_VMW(1) DX_VMW(2) DX_VMW(3) DX_VMW(4) DX_VMW(5) DX_VMW(6) DX_VMW(7) DX_VMW(8) DX_VMW(9) DX_VMW(10) DX_VMW(11) DX_VMW(12)
          DX_VMW(13) DX_VMW(14) DX_VMW(15) DX_VMW(16) DX_VMW(17) DX_VMW(18) DX_VMW(19) DX_VMW(20) DX_VMW(21) DX_VMW(22) DX_VMW(23) DX_VMW(24)
          DX_VMW(25) DX_VMW(26) DX_VMW(27) DX_VMW(28) DX_VMW(29) DX_VMW(30) DX_VMW(31) DX_VMW(32) DX_VMW(33) DX_VMW(34) DX_VMW(35) DX_VMW(36)
#undef DX_VMW
          default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
      };
      static_assert(MAXP * (RING - 2) <= 36, "vmcnt switch too short");
#pragma unroll
      for (int st = 0; st < RING - 1; ++st)
        if (st < nk && h > 0) issue_chunk(st, st);
      bool stores_in_flight = false;
      if constexpr (PLAN) {
        // padding fill: the batch's padding rows, flattened utterance by utterance, are split evenly over the workgroups;
        // this one owns [lo, hi).  Each loader wave finds the utterances its range touches with a wave scan over the
        // lengths, and the 256 loader threads share the 16-byte segments of those rows.
        const long lo = (long)blockIdx.x * fill_per, hi = lo + fill_per;
        const int ltid = lw * 64 + lane;
        long carry = 0;
        for (int base = 0; base < p.B && carry < hi; base += 64) {
          const int ub = base + lane;
          const int ulen = ub < p.B ? (int)p.skip_len[ub] : N;
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
            const int first = __shfl(N - dead + (int)(fs - ustart), src_lane, 64);
            int cnt = __shfl((int)(fe - fs), src_lane, 64);
            cnt = min(cnt, dx_fill_end((int)p.skip_len[fb], N) - first);   // dead rows past the fill end stay unwritten (dx_common.h)
            float z[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            for (int c = ltid; c < cnt * (BN / 8); c += NTHREADS) {
              const int n = first + (c >> 4), cl = (c & 15) * 8;
              const size_t off = ((size_t)fb * N + n) * BN + cl;
              if (LN == 2 || p.ln.y) store8<float>(p.ln.y + off, z);
              if (LN == 2 || p.ln.y_lp) store8<bf16_t>(reinterpret_cast<bf16_t*>(p.ln.y_lp) + off, z);
              if (LN == 1) {
                if (p.ln.s_out) store8<float>(p.ln.s_out + off, z);
                if (p.ln.mean && cl == 0) { p.ln.mean[(size_t)fb * N + n] = 0.f; p.ln.rstd[(size_t)fb * N + n] = 0.f; }
              }
            }
            stores_in_flight = true;
          }
          carry += __shfl(incl, 63, 64);
        }
        if (h <= 0) return;
      }
      int nbuf = RING - 1, k = 0;                                // buffer that chunk k + RING - 1 goes to
      for (; k + RING - 1 < nk; ++k) {
        // (the fill's stores share the counter and may retire out of order with the loads: drain everything once)
        if (PLAN && k == 0 && stores_in_flight) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else wait_landed(mine * (RING - 2));
        __builtin_amdgcn_s_barrier();
        issue_chunk(k + RING - 1, nbuf);
        nbuf = nbuf + 1 == RING ? 0 : nbuf + 1;
      }
      for (; k < nk; ++k) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
      }
      if (!PLAN) return;                                          // fixed tiles: the epilogue belongs to the MFMA waves
    }
    // MFMA waves.  PLAN: wave (wm, wn) owns the 32-row blocks wm, wm + 2, ... (interleaved, so a short tile still
    // spreads over both wave rows) and skips the blocks past the tile's height; the loader waves come back for the
    // epilogue as a second 256-thread team (one 64-row slab each per round).
    if (wave < 4) {
    auto row_of = [&](int i) { return PLAN ? (2 * i + wm) * 32 : wm * 32 * MI + i * 32; };
    const int nact = PLAN ? __builtin_amdgcn_readfirstlane((((h + 31) >> 5) - wm + 1) >> 1) : MI;
    auto mainloop = [&](auto na_tag) {
      constexpr int NA = decltype(na_tag)::value;
      int buf = 0;
      for (int k = 0; k < nk; ++k) {
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if constexpr (NA > 0) {
          const TC* Ar = ring + buf * STAGE_EL;
          const TC* Wr = Ar + AR16 * 32;
          frag_t a[2][NA], bf[2][2];
          auto load_frags = [&](int step, frag_t* af, frag_t* bfr) {
            const int tap = step >> 1, ks = step & 1;
#pragma unroll
            for (int i = 0; i < NA; ++i) af[i] = *reinterpret_cast<const frag_t*>(&Ar[lds_at(row_of(i) + l31 + tap, ks * 2 + g)]);
#pragma unroll
            for (int j = 0; j < 2; ++j) bfr[j] = *reinterpret_cast<const frag_t*>(&Wr[lds_at(tap * BN + wn * 64 + j * 32 + l31, ks * 2 + g)]);
          };
          load_frags(0, a[0], bf[0]);
#pragma unroll
          for (int step = 0; step < NSTEP; ++step) {   // fragments of k-step s + 1 are read before the MFMAs of k-step s
            if (step + 1 < NSTEP) load_frags(step + 1, a[(step + 1) & 1], bf[(step + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < NA; ++i)
#pragma unroll
              for (int j = 0; j < 2; ++j) dx_mma(acc[i][j], a[step & 1][i], bf[step & 1][j]);
          }
        }
        buf = buf + 1 == RING ? 0 : buf + 1;
      }
    };
    if (nact >= MI) mainloop(std::integral_constant<int, MI>{});
    else if (MI > 3 && nact == 3) mainloop(std::integral_constant<int, (MI > 3 ? 3 : MI)>{});
    else if (MI > 2 && nact == 2) mainloop(std::integral_constant<int, (MI > 2 ? 2 : MI)>{});
    else if (MI > 1 && nact == 1) mainloop(std::integral_constant<int, 1>{});
    else mainloop(std::integral_constant<int, 0>{});
    }
    __syncthreads();                                 // every MFMA wave is done with the ring: the epilogue stages through it
  } else {
  // register-staged pipeline.  CG_K1_PF = 2 keeps TWO chunks of the k = 1 GEMMs in flight in two static register sets: no gain (see the
    // macro) -- their chunk loop (0.74 us per chunk, tools/cg_timing_k1.py: 8.9 us of an 18 us QKV data gradient, 3.6 us of a 14 us
    // out-projection + LayerNorm whose epilogue runs at 5 TB/s) is bound by its two barriers and the LDS round trip per chunk
    constexpr int PF = TAPS == 1 ? CG_K1_PF : 1;
    raw_t ra[PF][A_PT];
    frag_t rw[PF][W_PT];
    auto fetch = [&](auto slot, int k0) {
      constexpr int S = decltype(slot)::value;
#pragma unroll
      for (int t = 0; t < A_PT; ++t) {
        const int c = tid + t * NTHREADS;
        const int r = c / KC, kc = (c % KC) * 8;
        const int n = n0 + r - HALO, ci = k0 + kc;
#pragma unroll
        for (int e = 0; e < 8; ++e) ra[S][t][e] = (TA)0.f;
        if (c < A_CH && n >= 0 && n < N && ci < Cin) ra[S][t] = raw_load8<TA>(X + (size_t)n * p.ldx + ci);
      }
#pragma unroll
      for (int t = 0; t < W_PT; ++t) {
        const int c = tid + t * NTHREADS;
        const int tap = c / (BN * KC), rem = c - tap * (BN * KC);
        const int row = rem / KC, kc = (rem % KC) * 8;
        const int co = co0 + row, ci = k0 + kc;
        rw[S][t] = zero8<TC>();
        if (co < Cout && ci < Cin) rw[S][t] = *reinterpret_cast<const frag_t*>(W + ((size_t)tap * Cout + co) * Cin + ci);
      }
    };
    auto commit = [&](auto slot) {
      constexpr int S = decltype(slot)::value;
#pragma unroll
      for (int t = 0; t < A_PT; ++t) {
        const int c = tid + t * NTHREADS;
        if (c < A_CH) *reinterpret_cast<frag_t*>(&As[lds_at(c / KC, c % KC)]) = cvt8<TA, TC>(ra[S][t]);
      }
#pragma unroll
      for (int t = 0; t < W_PT; ++t) {
        const int c = tid + t * NTHREADS;
        const int tap = c / (BN * KC), rem = c - tap * (BN * KC);
        *reinterpret_cast<frag_t*>(&Ws[lds_at(tap * BN + rem / KC, rem % KC)]) = rw[S][t];
      }
    };
    auto compute = [&]() {
#pragma unroll
      for (int tap = 0; tap < TAPS; ++tap) {
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
          frag_t a[MI], bf[2];
#pragma unroll
          for (int i = 0; i < MI; ++i)
            a[i] = *reinterpret_cast<const frag_t*>(&As[lds_at(wm * 32 * MI + i * 32 + l31 + tap, ks * 2 + g)]);
#pragma unroll
          for (int j = 0; j < 2; ++j)
            bf[j] = *reinterpret_cast<const frag_t*>(&Ws[lds_at(tap * BN + wn * 64 + j * 32 + l31, ks * 2 + g)]);
#pragma unroll
          for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) dx_mma(acc[i][j], a[i], bf[j]);
        }
      }
    };
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, PF - 1>;

    fetch(S0{}, 0);
    if constexpr (PF == 2) {
      if (BK < Cin) fetch(S1{}, BK);
      commit(S0{});
      __syncthreads();
      // chunk k0 is in LDS, chunk k0 + BK in the registers of the other set, chunk k0 + 2 BK is requested into the set just committed
      auto step = [&](auto cur, auto nxt, int k0) {
        if (k0 + 2 * BK < Cin) fetch(cur, k0 + 2 * BK);
        compute();
        __syncthreads();
        if (k0 + BK < Cin) {
          commit(nxt);
          __syncthreads();
        }
      };
      for (int k0 = 0; k0 < Cin; k0 += 2 * BK) {
        step(S0{}, S1{}, k0);
        if (k0 + BK < Cin) step(S1{}, S0{}, k0 + BK);
      }
    } else {
      commit(S0{});
      __syncthreads();
      for (int k0 = 0; k0 < Cin; k0 += BK) {
        const bool more = k0 + BK < Cin;
        if (more) fetch(S0{}, k0 + BK);
        compute();
        __syncthreads();
        if (more) {
          commit(S0{});
          __syncthreads();
        }
      }
    }
  }

  // ---- epilogue
  constexpr int NCS = LNM == 2 ? 4 : (LNM == 3 ? 2 : 1);
  float csum[NCS][8];   // LN backward: this thread's column sums (dgamma, dbeta [, dfilm_g, dfilm_b]) over its rows
#pragma unroll
  for (int q = 0; q < NCS; ++q)
#pragma unroll
    for (int e = 0; e < 8; ++e) csum[q][e] = 0.f;
  // PLAN: two teams of 256 threads (MFMA waves / loader waves) take one 64-row slab each per round
  constexpr int ETEAMS = PLAN ? 2 : 1;
  const int team = PLAN ? tid >> 8 : 0, etid = PLAN ? tid & 255 : tid;
  float* const mystage = stage + team * (64 * STG_LD);
  if (vec_out) {
#pragma unroll
    for (int ip = 0; ip < MI / ETEAMS; ++ip) {
      if (PLAN && ip * 64 * ETEAMS >= h) break;                   // workgroup-uniform: the barriers below stay matched
      const int i = ip * ETEAMS + team;                           // this team's slab
      // phase 1: bias + ReLU in the MFMA layout, accumulators -> LDS stage (64 rows x 128 channels, fp32, one per team)
      if (!PLAN || tid < NTHREADS) {
#pragma unroll
        for (int sl = 0; sl < ETEAMS; ++sl) {
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const int cl = wn * 64 + j * 32 + l31, co = co0 + cl;
            const float bv = (p.bias && co < Cout) ? p.bias[co] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              float v = acc[ip * ETEAMS + sl][j][r] + bv;
              if (relu) v = fmaxf(v, 0.f);
              stage[sl * (64 * STG_LD) + (wm * 32 + dx_acc_row(r, g)) * STG_LD + cl] = v;
            }
          }
        }
      }
      // LayerNorm epilogues: the global inputs of all four passes are requested BEFORE the barrier (one round trip per
      // 64-row slab; issued pass by pass they are four dependent trips, because the in-place stores of a pass may alias
      // the loads of the next one as far as the compiler knows -- they never do: every row belongs to one thread group)
      constexpr bool PFB = !(LNFILM && MI != 2);     // FiLM-gradient variants at the 128 / 256 register caps: loads stay in their pass
      f32x8 pf_a[4], pf_b[4];
      float pf_m[4], pf_r[4];
      // the per-channel operands of the row passes (gamma, beta, FiLM row of this utterance) depend on the thread's channel segment
      // only: requested once per slab in front of the barrier.  Inside the passes every one of them sat behind the stores of the
      // pass before -- y / s_out may alias them as far as the compiler knows -- one exposed L2 round trip per pass (conv_sk_kernel:
      // 8.9 -> 7.3 us of epilogue).  Not for the variants at their register caps (PFB).
      const int cl_h = (etid & 15) * 8;
      f32x8 gm_h, bt_h, fg_h, fb_h;
      if (LN != 0 && PFB) {
        gm_h = raw_load8<float>(p.ln.gamma + cl_h);
        if (LN == 1 || LNFILM) bt_h = raw_load8<float>(p.ln.beta + cl_h);
        if (p.ln.film && (LN == 1 || LNFILM)) fg_h = raw_load8<float>(p.ln.film + (size_t)b * p.ln.ldf + cl_h);
        if (p.ln.film && LN == 1) fb_h = raw_load8<float>(p.ln.film + (size_t)b * p.ln.ldf + BN + cl_h);
      }
      if (LN != 0 && PFB) {
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
          const int sr = (etid >> 4) + pass * 16;
          const int trow = PLAN ? i * 64 + sr : (sr >> 5) * 32 * MI + i * 32 + (sr & 31);
          const int n = n0 + trow, cl = (etid & 15) * 8;
          if (n < N && trow < h) {
            const size_t rowg = (size_t)b * N + n, offl = rowg * BN + cl;
            if (LN == 2) {
              pf_a[pass] = raw_load8<float>(p.ln.y + offl);
              pf_b[pass] = raw_load8<float>(p.ln.s_out + offl);
              pf_m[pass] = p.ln.mean[rowg];
              pf_r[pass] = p.ln.rstd[rowg];
            } else {
              pf_a[pass] = raw_load8<float>(p.ln.residual + offl);
            }
          }
        }
      }
      __syncthreads();
      // phase 2: whole 16-byte row segments: gate, mask, accumulate, store
#pragma unroll
      for (int pass = 0; pass < 4; ++pass) {
        const int sr = (etid >> 4) + pass * 16;                    // stage row 0..63
        const int trow = PLAN ? i * 64 + sr : (sr >> 5) * 32 * MI + i * 32 + (sr & 31);   // row inside the tile
        const int n = n0 + trow, cl = (etid & 15) * 8, co = co0 + cl;
        if (n < N && co < Cout && trow < h) {
          float v[8];
          const f32x4 lo = *reinterpret_cast<const f32x4*>(&mystage[sr * STG_LD + cl]);
          const f32x4 hi = *reinterpret_cast<const f32x4*>(&mystage[sr * STG_LD + cl + 4]);
          v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3]; v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
          if (LN == 2) {        // fused LayerNorm BACKWARD: v + residual gradient = dL/d(LN output) of this row
            const size_t rowg = (size_t)b * N + n, offl = rowg * BN + cl;
            {
              const f32x8 r = PFB ? pf_a[pass] : raw_load8<float>(p.ln.y + offl);
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] = n < len ? v[e] + r[e] : 0.f;     // masked_fill rows carry no gradient
            }
            const f32x8 sv = PFB ? pf_b[pass] : raw_load8<float>(p.ln.s_out + offl);
            const float mean = PFB ? pf_m[pass] : p.ln.mean[rowg], rstd = PFB ? pf_r[pass] : p.ln.rstd[rowg];
            const f32x8 gm = PFB ? gm_h : raw_load8<float>(p.ln.gamma + cl);
            float xh[8], s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) xh[e] = (sv[e] - mean) * rstd;
            if (LNFILM) {                                         // y = fg * LN + fb
              const f32x8 fg = PFB ? fg_h : raw_load8<float>(p.ln.film + (size_t)b * p.ln.ldf + cl), bt = PFB ? bt_h : raw_load8<float>(p.ln.beta + cl);
#pragma unroll
              for (int e = 0; e < 8; ++e) {
                csum[LNFILM ? 2 : 0][e] += v[e] * (xh[e] * gm[e] + bt[e]);
                csum[LNFILM ? 3 : 0][e] += v[e];
                v[e] *= fg[e];
              }
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              csum[0][e] += v[e] * xh[e];
              csum[1][e] += v[e];
              v[e] *= gm[e];
              s1 += v[e];
              s2 += v[e] * xh[e];
            }
            s1 = dx_row16_sum(s1); s2 = dx_row16_sum(s2);
            s1 *= 1.f / BN; s2 *= 1.f / BN;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = rstd * (v[e] - s1 - xh[e] * s2);
            store8<float>(p.ln.y + offl, v);                      // ds, in place of the residual gradient
            if (p.ln.p_pre > 0.f) {
              const uint32_t th = dx_drop_th8(p.ln.p_pre), key = dx_key32(dx_seed_eff(p.ln.seed_pre, p.ln.step), 0);
              const float sc = dx_drop_inv_keep8(th);
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] = dx_keep_elem(key, (uint32_t)rowg * BN + cl + e, th) ? v[e] * sc : 0.f;
            }
            store8<bf16_t>(reinterpret_cast<bf16_t*>(p.ln.y_lp) + offl, v);
            continue;
          }
          if (LN == 1) {        // fused LayerNorm: 16 lanes hold one complete 128-channel row
            const size_t rowg = (size_t)b * N + n, offl = rowg * BN + cl;
            if (p.ln.p_pre > 0.f) {
              const uint32_t th = dx_drop_th8(p.ln.p_pre), key = dx_key32(dx_seed_eff(p.ln.seed_pre, p.ln.step), 0);
              const float sc = dx_drop_inv_keep8(th);
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] = dx_keep_elem(key, (uint32_t)rowg * BN + cl + e, th) ? v[e] * sc : 0.f;
            }
            {
              const f32x8 r = pf_a[pass];
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] += r[e];
            }
            if (p.ln.s_out) store8<float>(p.ln.s_out + offl, v);
            float sum = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) sum += v[e];
            sum = dx_row16_sum(sum);
            const float mean = sum * (1.f / BN);
            float sq = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float d = v[e] - mean; sq += d * d; }
            sq = dx_row16_sum(sq);
            const float rstd = rsqrtf(sq * (1.f / BN) + 1e-5f);
            if (p.ln.mean && cl == 0) { p.ln.mean[rowg] = mean; p.ln.rstd[rowg] = rstd; }
            const f32x8 gm = gm_h, bt = bt_h;                      // (LN == 1: PFB is always true)
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (v[e] - mean) * rstd * gm[e] + bt[e];
            if (p.ln.film) {
              const f32x8 fg = fg_h, fb = fb_h;
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] = fg[e] * v[e] + fb[e];
            }
            if (n >= len) {
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] = 0.f;
            }
            if (p.ln.y) store8<float>(p.ln.y + offl, v);       // (NULL: the consumer of the fp32 stream re-derives it, LNEpi::res_mean)
            if (p.ln.y_lp) store8<bf16_t>(reinterpret_cast<bf16_t*>(p.ln.y_lp) + offl, v);
            continue;
          }
          const size_t off = ((size_t)b * N + n) * p.ldy + co;
          if (G) {
            const typename VecN<TG, 8>::type gv = raw_load8<TG>(G + off);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = ((float)gv[e] > 0.f) ? v[e] : 0.f;
          }
          if (n >= len) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = 0.f;
          }
          if (accum) {
            const typename VecN<TO, 8>::type old = raw_load8<TO>(Y + off);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += (float)old[e];
          }
          store8<TO>(Y + off, v);
        }
      }
      __syncthreads();
    }
    if (LN == 2) {   // column sums: 16 row-threads per channel segment -> LDS -> one atomic per channel per workgroup
      constexpr int nq = NCS;
      constexpr int RG = 16 * ETEAMS;                               // row groups (16 threads each) that hold partial sums
      for (int q = 0; q < nq; ++q)
#pragma unroll
        for (int e = 0; e < 8; ++e) stage[(q * RG + (tid >> 4)) * BN + (tid & 15) * 8 + e] = csum[q][e];
      __syncthreads();
      for (int idx = tid; idx < nq * BN; idx += NTHREADS * ETEAMS) {
        const int q = idx / BN, c = idx - q * BN;
        float t = 0.f;
#pragma unroll
        for (int r = 0; r < RG; ++r) t += stage[(q * RG + r) * BN + c];
        if (q == 0) atomicAdd(p.ln.dgamma + c, t);
        else if (q == 1) atomicAdd(p.ln.dbeta + c, t);
        else atomicAdd(p.ln.dfilm + (size_t)b * p.ln.lddf + (q == 3 ? BN : 0) + c, t);
      }
    }
    return;
  }
  // scalar path: transposed output (mel projection) or channel counts that are not multiples of 8
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int co = co0 + wn * 64 + j * 32 + l31;
    if (co >= Cout) continue;
    const float bv = p.bias ? p.bias[co] : 0.f;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = n0 + wm * 32 * MI + i * 32 + dx_acc_row(r, g);
        if (n >= N) continue;
        float v = acc[i][j][r] + bv;
        if (relu) v = fmaxf(v, 0.f);
        const size_t off = trans ? ((size_t)b * Cout + co) * p.ldy + n : ((size_t)b * N + n) * p.ldy + co;
        if (G) v = ((float)G[off] > 0.f) ? v : 0.f;
        if (n >= len) v = 0.f;
        if (accum) v += (float)Y[off];
        Y[off] = (TO)v;
      }
    }
  }
}

// ---- weight-stationary variant for short contractions (Cin = 128, Cout a multiple of 256: the FF block's 128 -> 1024
// conv and the data gradient of its 1024 -> 128 partner).  With K = taps * Cin <= 384 the tiled kernel above spends a
// workgroup's life waiting: 4 K-chunks of 0.3 us of MFMA work, each behind a ~1.5 us global -> LDS round trip, plus a
// prologue and an epilogue (measured 26 % MFMA utilisation at 3 workgroups / CU).  Here a 512-thread workgroup owns 256
// output channels for its lifetime: wave w keeps the weights of channels [32w, 32w + 32) for the WHOLE contraction in
// registers as ready-made MFMA B fragments (taps * 8 k-steps * 4 VGPRs = 96), so the weights are read once per
// workgroup instead of once per position tile and never touch LDS.  128-position tiles of the input stream past:
// the A tile (130 x 128) is double-buffered in LDS, fetched into registers one tile ahead, and shared by the 8 waves;
// per k-step a wave reads 4 A fragments for 4 MFMAs (128 rows x 32 channels).  One barrier per tile.  The epilogue is
// wave-private and register-only: the MFMA operands are swapped (D[co][pos]) so that a lane ends up with 8 consecutive
// channels of one position after four v_permlane32_swap, bias / ReLU / gate are applied in that layout and the block
// leaves through 16-byte buffer stores (out-of-range rows dropped by the descriptor), issued in slices between the
// MFMAs of the next tile, so no wave waits for another between tiles.  The live position tiles of the batch
// (skip_lengths) are split evenly over the workgroups of a channel block; dead tiles are zero-filled in a second pass.
constexpr int WR_THREADS = 512, WR_BN = 256, WR_BM = 128;
// BITS (bf16 output only): the ReLU of the FF block's first conv also leaves ONE BIT per output element -- a 32-bit word per
// (position, 32-channel block of a wave), bit layout = the wave's own post-swap register order -- and the data gradient of the second
// conv gates with that word instead of re-reading the 2 KB activation row: 128 B instead of 2 KB per row of gate traffic.
template <typename TO, typename TG, int TAPS, bool RELU, bool GATE, bool BITS = false>
__global__ __launch_bounds__(WR_THREADS, 2) void conv_wreg_kernel(ConvArgs p, int ngrp) {
  typedef bf16_t TC;
  constexpr int BM = WR_BM, HALO = TAPS / 2, AROWS = BM + TAPS - 1, CIN = 128, LDK = CIN + Pad<TC>::value, KCH = CIN / 8;
  constexpr int KSTEPS = CIN / 16;
  constexpr int A_CH = AROWS * KCH, A_PT = (A_CH + WR_THREADS - 1) / WR_THREADS;
  constexpr int A_BYTES = AROWS * LDK * (int)sizeof(TC);
  typedef typename Vec8<TC>::type frag_t;
  __shared__ __attribute__((aligned(16))) char smem[2 * A_BYTES];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, g = lane >> 5;
  const int ztiles = p.Cout / WR_BN, ptiles = dx_cdiv(p.N, BM);
  // workgroup -> (position group, channel slice), slice-major: the ztiles workgroups of one position group sit 64 indices apart = on
  // the SAME XCD (round-robin dispatch, ngrp % 8 == 0) and read their common activation tiles through one L2
  const bool zmajor = ngrp % 8 == 0;
  const int grp = zmajor ? (int)blockIdx.x % ngrp : (int)blockIdx.x / ztiles, zt = zmajor ? (int)blockIdx.x / ngrp : (int)blockIdx.x % ztiles;
  const int co0 = zt * WR_BN + wave * 32;
  const int N = p.N, Cout = p.Cout;
  const TC* W = reinterpret_cast<const TC*>(p.w);
  TO* Y = reinterpret_cast<TO*>(p.y);
  const TG* G = reinterpret_cast<const TG*>(p.gate);

  // ---- this wave's weights, once: B fragment of k-step (tap, ks) = W[tap][co0 + l31][16 ks + 8 g .. + 8]
  // Read straight from global memory a fragment load touches 32 rows x 2 x 16 bytes -- 64 sectors for 1 KB -- and the prologue
  // took 9.3 us of a 43 us launch (s_memrealtime stamps per workgroup, tools/wreg_timing.py): five dependent global round trips
  // (three taps of weights, bias, the first A tile) behind the kernel-argument load.  Now the workgroup's slice of each tap,
  // W[tap][256 channels][128] = 64 KB CONTIGUOUS, is requested with whole-row 16-byte loads at the very top, the bias, the tile
  // bookkeeping and the first A tile are requested behind it, and only then do the slices pass through the (still unused) A
  // buffers, one tap at a time, for the waves to pick up their fragments.
  frag_t wreg[TAPS][KSTEPS];
  static_assert(WR_BN * LDK * (int)sizeof(TC) <= 2 * A_BYTES, "a tap's weight slice must fit in the A buffers");
  static_assert((WR_BN * KCH) % WR_THREADS == 0, "weight slice must split evenly over the workgroup");
  constexpr int W_PT = WR_BN * KCH / WR_THREADS;
  bf16x8 wtmp[TAPS][W_PT];
  // With a fragment-order copy of the weights (dx_pack_frag_major: a fragment is one contiguous KiB, and it is exactly
  // wreg[tap][ks] of the wave that owns channel block co0 / 32) the wave loads its 8 x TAPS fragments straight into their
  // registers: one round trip, no pass through LDS, none of the 2 TAPS barriers below.
  const bool wfrag = TAPS == 3 && p.w_frag != nullptr;
  if (wfrag) {
    const TC* wf = reinterpret_cast<const TC*>(p.w_frag) + (size_t)(co0 >> 5) * 512 + lane * 8;
    const size_t fstride = (size_t)(Cout >> 5) * 512;            // fragments of one (chunk, tap, half): all channel blocks
#pragma unroll
    for (int tap = 0; tap < TAPS; ++tap)
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ++ks)
        wreg[tap][ks] = *reinterpret_cast<const frag_t*>(wf + (size_t)((((ks >> 1) * TAPS + tap) << 1) + (ks & 1)) * fstride);
  } else {
    const int cblk = zt * WR_BN;
#pragma unroll
    for (int tap = 0; tap < TAPS; ++tap) {
      const TC* src = W + ((size_t)tap * Cout + cblk) * CIN;
#pragma unroll
      for (int t = 0; t < W_PT; ++t) wtmp[tap][t] = *reinterpret_cast<const bf16x8*>(src + (size_t)(tid + t * WR_THREADS) * 8);
    }
  }
  // The MFMAs run with the operands swapped (weights as A, activations as B), so the accumulator tile is D[co][position]:
  // a lane holds ONE position (l31) and, per group of 4 registers, 4 CONSECUTIVE output channels (rows (r & 3) + 8 (r >> 2)
  // + 4 g) -- row-major output leaves the registers without an LDS transpose.
  // (the data-gradient instantiation has no bias: its 16 registers hold the prefetched gate values instead, see gpre)
  float bvr[GATE ? 1 : 16];
  if constexpr (!GATE) {
#pragma unroll
    for (int r = 0; r < 16; ++r) bvr[r] = p.bias ? p.bias[co0 + dx_acc_row(r, g)] : 0.f;
  }

  // ---- this workgroup's share of the live position tiles (flat list over the batch)
  // (cooperative count + prefix sums in LDS, dx_block_count_scan: the serial walks over the lengths -- count, locate the first live
  //  tile, locate the first dead tile -- were most of this kernel's 8 us prologue)
  __shared__ int s_live[DX_SCAN_MAXB + 1], s_cum[DX_SCAN_MAXB + 1], s_part[WR_THREADS / 64];
  const bool scan = p.B <= DX_SCAN_MAXB;
  auto live_g = [&](int b) { return p.skip_len ? min(ptiles, dx_cdiv(min(N, (int)p.skip_len[b] + 2), BM)) : ptiles; };
  auto live_of = [&](int b) { return scan ? s_live[b] : live_g(b); };
  int total = 0, b = 0, pt = 0, nlive = 0, i0, i1;
  if (scan) {
    dx_block_count_scan<WR_THREADS>(p.B, live_g, [](int v) { return v; }, s_live, s_cum, s_part);
    total = s_cum[p.B];
    i0 = (int)((long)total * grp / ngrp); i1 = (int)((long)total * (grp + 1) / ngrp);
    b = dx_locate_item(s_cum, p.B, i0);
    nlive = s_live[b];
    pt = i0 - s_cum[b];
  } else {
    for (int bb = 0; bb < p.B; ++bb) total += live_g(bb);
    i0 = (int)((long)total * grp / ngrp); i1 = (int)((long)total * (grp + 1) / ngrp);
    for (int cum = 0; b < p.B; ++b) {
      nlive = live_g(b);
      if (i0 < cum + nlive) { pt = i0 - cum; break; }
      cum += nlive;
    }
  }
  int left = i1 - i0;

  // All global accesses of the tile loop are BUFFER loads / stores on a per-utterance resource: rows outside [0, N) are
  // dropped / read as zero by the hardware bounds check, so the loop body has no divergent branches and hipcc can count
  // the outstanding memory operations exactly (with `if (n < N)` around the stores it fell back to `s_waitcnt vmcnt(0)`
  // in front of every epilogue block, which also drained the A-tile prefetch issued at the top of the tile).
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  const uint32_t xbytes = (uint32_t)((size_t)N * p.ldx * sizeof(TC)), ybytes = (uint32_t)((size_t)N * p.ldy * sizeof(TO));
  bf16x8 ra[A_PT];
  auto fetch = [&](int fb, int fpt) {
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<TC*>(reinterpret_cast<const TC*>(p.x)) + (size_t)fb * N * p.ldx, 0, xbytes, 0x00020000);
#pragma unroll
    for (int t = 0; t < A_PT; ++t) {
      const int c = tid + t * WR_THREADS;
      const int n = fpt * BM + (c >> 4) - HALO;                       // -1 (halo of the first tile) wraps to out-of-range
      const uint32_t voff = c < A_CH ? (uint32_t)(n * (int)p.ldx + (c & 15) * 8) * (uint32_t)sizeof(TC) : 0xffffff00u;
      ra[t] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rx, (int)voff, 0, 0));
    }
  };
  auto commit = [&](int buf) {
    TC* As = reinterpret_cast<TC*>(smem + buf * A_BYTES);
#pragma unroll
    for (int t = 0; t < A_PT; ++t) {
      const int c = tid + t * WR_THREADS;
      if (c < A_CH) *reinterpret_cast<bf16x8*>(&As[(c >> 4) * LDK + (c & 15) * 8]) = ra[t];
    }
  };

  // Software pipeline inside a wave: the matrix pipe runs asynchronously, so the epilogue of one 64-row half (VALU +
  // LDS + stores) is issued in slices BETWEEN the MFMAs of the other half:
  //   phase A(t): MFMAs of rows 0..63 of tile t    ||  epilogue of rows 64..127 of tile t-1
  //   phase B(t): MFMAs of rows 64..127 of tile t  ||  epilogue of rows 0..63 of tile t
  // (measured before: MFMA loop 22 us + epilogue 15 us back to back; the two waves of a SIMD ran them in lockstep)
  struct Epi { int cb, n0, len; };   // utterance, first row of the tile, mask length
  // Epilogue of one 32-position accumulator tile, straight from registers.  bf16 output: two v_permlane32_swap per
  // 8-channel group gather a lane's 8 consecutive channels (16-byte stores; lanes g = 0 / 1 of a position write
  // channels [0, 8) / [8, 16) and [16, 24) / [24, 32) of the wave's 32); fp32 output: one 16-byte store per register group.
  // gate words of ONE 32-position tile (two 16-byte loads per lane), requested by gate_fetch one or more k-steps before the
  // epilogue slice that consumes them: issued inside epi_tile they were consumed by the very next instruction, a full
  // memory round trip with the wave unable to issue MFMAs, four times per position tile (the GATE variant ran 62 us where
  // the same GEMM without a gate runs 42).
  typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
  u32x4_t gpre[2][2];
  auto gate_fetch = [&](u32x4_t* dst, const Epi& e, int row0) {
    if constexpr (GATE && BITS) {
      const int n = e.n0 + row0 + l31;
      const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<uint32_t*>(p.gate_bits) + ((size_t)e.cb * (Cout >> 5) + (co0 >> 5)) * N, 0, (uint32_t)((size_t)N * 4), 0x00020000);
      dst[0][0] = __builtin_amdgcn_raw_buffer_load_b32(rb, n * 4, 0, 0);      // (rows outside [0, N): zero = gate closed; their stores are dropped)
    } else if constexpr (GATE && sizeof(TO) == 2) {
      const int n = e.n0 + row0 + l31;
      const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<TG*>(G) + (size_t)e.cb * N * p.ldy, 0, (uint32_t)((size_t)N * p.ldy * sizeof(TG)), 0x00020000);
      const uint32_t eoff = (uint32_t)n * (uint32_t)p.ldy + (uint32_t)co0;
#pragma unroll
      for (int h2 = 0; h2 < 2; ++h2) dst[h2] = __builtin_amdgcn_raw_buffer_load_b128(rg, (int)((eoff + 16 * h2 + 8 * g) * 2u), 0, 0);
    }
  };
  auto epi_tile = [&](const f32x16& ac, const Epi& e, int row0, const u32x4_t* gw2) {
    const int n = e.n0 + row0 + l31;
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(Y + (size_t)e.cb * N * p.ldy, 0, ybytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<TG*>(GATE ? G : reinterpret_cast<const TG*>(Y)) + (size_t)e.cb * N * p.ldy, 0, (uint32_t)((size_t)N * p.ldy * sizeof(TG)), 0x00020000);
    const uint32_t eoff = (uint32_t)n * (uint32_t)p.ldy + (uint32_t)co0;     // element offset inside the utterance
    const bool zero_row = n >= e.len;              // mask_lengths
    float v[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      v[r] = GATE ? ac[r] : ac[r] + bvr[GATE ? 0 : r];
      if (GATE && p.bias) v[r] += p.bias[co0 + dx_acc_row(r, g)];      // (no caller on the step path gates AND biases: loaded in place)
      if (RELU) v[r] = fmaxf(v[r], 0.f);
    }
    if constexpr (sizeof(TO) == 4) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const uint32_t o = (eoff + 8 * q + 4 * g) * 4u;
        f32x4 w = {v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
        if (GATE) {
          const f32x4 gv = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rg, (int)o, 0, 0));
#pragma unroll
          for (int j = 0; j < 4; ++j) w[j] = gv[j] > 0.f ? w[j] : 0.f;
        }
        if (zero_row) w = f32x4{0.f, 0.f, 0.f, 0.f};
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, w), ry, (int)o, 0, 0);
      }
    } else {
      typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
      typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
      uint32_t P[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const bf16x2 pr = {(bf16_t)v[2 * k], (bf16_t)v[2 * k + 1]};
        P[k] = __builtin_bit_cast(uint32_t, pr);
      }
      // (P0,P1 | P2,P3) and (P4,P5 | P6,P7): hi half of the first pair <-> lo half of the second
#pragma unroll
      for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const u32x2 sw = __builtin_amdgcn_permlane32_swap(P[4 * h2 + k], P[4 * h2 + 2 + k], false, false);
          P[4 * h2 + k] = sw[0];
          P[4 * h2 + 2 + k] = sw[1];
        }
      // now (P0, P1, P2, P3) = channel pairs (0,1)(2,3)(4,5)(6,7) + 8 g and (P4 .. P7) the same + 16
      if constexpr (RELU && BITS) {   // bit k / 16 + k of a lane's word: low / high half of P[k] is non-zero (values are >= 0: + 0x7fff carries into bit 15)
        uint32_t m = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) m |= (((P[k] + 0x7fff7fffu) >> (15 - k)) & (0x00010001u << k));
        const uint32_t mp = (uint32_t)__shfl_xor((int)m, 32, 64);
        uint32_t word = g ? (mp | (m << 8)) : (m | (mp << 8));      // lane group 0 in bits 0-7 / 16-23, group 1 in 8-15 / 24-31
        if (zero_row) word = 0u;
        const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(
            p.relu_bits + ((size_t)e.cb * (Cout >> 5) + (co0 >> 5)) * N, 0, (uint32_t)((size_t)N * 4), 0x00020000);
        __builtin_amdgcn_raw_buffer_store_b32(word, rb, g ? (int)0xffffff00u : n * 4, 0, 0);   // one lane of the pair stores (the other one out of range)
      }
      uint32_t own = 0;
      if constexpr (GATE && BITS) own = g ? (gw2[0][0] >> 8) : gw2[0][0];
#pragma unroll
      for (int h2 = 0; h2 < 2; ++h2) {
        const uint32_t o = (eoff + 16 * h2 + 8 * g) * 2u;
        u32x4 w = {P[4 * h2], P[4 * h2 + 1], P[4 * h2 + 2], P[4 * h2 + 3]};
        if constexpr (GATE && BITS) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int k = 4 * h2 + j;
            const uint32_t lo = (uint32_t)__builtin_amdgcn_sbfe((int)own, k, 1), hi = (uint32_t)__builtin_amdgcn_sbfe((int)own, 16 + k, 1);
            w[j] &= (lo & 0x0000ffffu) | (hi & 0xffff0000u);
          }
        } else if (GATE) {   // gate > 0 on the packed bf16 bits: sign clear and magnitude non-zero  <=>  bits - 1 < 0x7fff (unsigned)
          const u32x4 gw = gw2[h2];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const uint32_t m = (((gw[j] & 0xffffu) - 1u) < 0x7fffu ? 0x0000ffffu : 0u) | (((gw[j] >> 16) - 1u) < 0x7fffu ? 0xffff0000u : 0u);
            w[j] &= m;
          }
        }
        if (zero_row) w = u32x4{0u, 0u, 0u, 0u};
        __builtin_amdgcn_raw_buffer_store_b128(w, ry, (int)o, 0, 0);
      }
    }
  };
  // slice kk (0 .. TAPS*KSTEPS-1) of the epilogue of accumulator pair ac[0], ac[1] (rows [64 h, 64 h + 64) of tile e)
  // `nx` / `nh`: the rows whose epilogue runs in the NEXT phase (the accumulators being filled now): their gate words are
  // requested right after this phase's second drain has freed gpre -- two thirds of a phase plus the head of the next one ahead
  auto epi_slice = [&](int kk, const f32x16* ac, const Epi& e, int h, const Epi& nx, int nh) {
    constexpr int NS = TAPS * KSTEPS;
    // early in the phase: the end-of-tile wait for the prefetched A tile (vmcnt) also covers these stores
    if (kk == 1) epi_tile(ac[0], e, h * 64, gpre[0]);
    else if (kk == NS / 3) epi_tile(ac[1], e, h * 64 + 32, gpre[1]);
    else if (kk == NS / 3 + 1) { gate_fetch(gpre[0], nx, nh * 64); gate_fetch(gpre[1], nx, nh * 64 + 32); }
  };

  int buf = 0;
  if (left > 0) fetch(b, pt);
  if (!wfrag) {   // weights: registers (whole rows) -> LDS -> registers (MFMA fragments), see the top of the kernel
    TC* Ws = reinterpret_cast<TC*>(smem);
#pragma unroll
    for (int tap = 0; tap < TAPS; ++tap) {
      if (tap) __syncthreads();   // the previous tap's fragments have been read
#pragma unroll
      for (int t = 0; t < W_PT; ++t) {
        const int c = tid + t * WR_THREADS;
        *reinterpret_cast<bf16x8*>(&Ws[(c >> 4) * LDK + (c & 15) * 8]) = wtmp[tap][t];
      }
      __syncthreads();
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ++ks)
        wreg[tap][ks] = *reinterpret_cast<const frag_t*>(&Ws[(wave * 32 + l31) * LDK + ks * 16 + g * 8]);
    }
    __syncthreads();   // the A tile of the first position tile goes into the same memory
  }
  if (left > 0) commit(0);
  __syncthreads();
  f32x16 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  Epi prev{0, N, 0};   // n0 = N: every row out of range, nothing is stored before the first tile
  while (left > 0) {
    const Epi cur{b, pt * BM, p.mask_len ? (int)p.mask_len[b] : N};
    const TC* As = reinterpret_cast<const TC*>(smem + buf * A_BYTES);
    --left;
    if (left > 0) {
      if (++pt >= nlive) { ++b; pt = 0; nlive = live_of(b); }
      fetch(b, pt);
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      // epilogue partner: phase A drains acc[2..3] of the previous tile, phase B drains acc[0..1] of this tile
      const Epi& ep = h == 0 ? prev : cur;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[h * 2 + i][r] = 0.f;   // drained one phase ago
#pragma unroll
      for (int tap = 0; tap < TAPS; ++tap) {
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
          frag_t a[2];
#pragma unroll
          for (int i = 0; i < 2; ++i) a[i] = *reinterpret_cast<const frag_t*>(&As[(h * 64 + i * 32 + l31 + tap) * LDK + ks * 16 + g * 8]);
#pragma unroll
          for (int i = 0; i < 2; ++i) dx_mma(acc[h * 2 + i], wreg[tap][ks], a[i]);
          epi_slice(tap * KSTEPS + ks, &acc[h == 0 ? 2 : 0], ep, h == 0 ? 1 : 0, cur, h);
        }
      }
    }
    prev = cur;
    if (left > 0) commit(buf ^ 1);
    buf ^= 1;
    __syncthreads();
  }
  {   // drain: rows 64..127 of the last tile
#pragma unroll
    for (int kk = 0; kk < TAPS * KSTEPS; ++kk) epi_slice(kk, &acc[2], prev, 1, Epi{0, N, 0}, 0);
  }

  // ---- dead tiles (start past length + conv halo): zeros, no reads; split evenly like the live ones
  if (p.skip_len) {
    const int cblk = zt * WR_BN;
    const int dead = ptiles * p.B - total;
    const int j0 = (int)((long)dead * grp / ngrp), j1 = (int)((long)dead * (grp + 1) / ngrp);
    int db = 0, dpt = 0, cum = 0;
    if (scan && j0 < j1) {   // dead tiles before utterance u: u * ptiles - s_cum[u] (monotone): the largest u with that <= j0
      int lo = 0, hi = p.B - 1;
      while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (mid * ptiles - s_cum[mid] <= j0) lo = mid; else hi = mid - 1; }
      db = lo;
      dpt = s_live[db] + (j0 - (db * ptiles - s_cum[db]));
    } else if (!scan) {
      for (; db < p.B; ++db) {
        const int nd = ptiles - live_of(db);
        if (j0 < cum + nd) { dpt = live_of(db) + (j0 - cum); break; }
        cum += nd;
      }
    }
    const float z[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int j = j0; j < j1; ++j) {
      const int fend = db < p.B ? dx_fill_end((int)p.skip_len[db], N) : 0;   // dead tiles past the fill end stay unwritten (dx_common.h)
      for (int c = tid; c < BM * (WR_BN / 8) && dpt * BM < fend; c += WR_THREADS) {
        const int n = dpt * BM + (c >> 5), co = cblk + (c & 31) * 8;
        if (n < fend) store8<TO>(Y + ((size_t)db * N + n) * p.ldy + co, z);
      }
      if (++dpt >= ptiles) { ++db; while (db < p.B && live_of(db) >= ptiles) ++db; dpt = db < p.B ? live_of(db) : 0; }
    }
  }
}


// ---- narrow-output k = 3 GEMM with the LayerNorm epilogues, split-K INSIDE the workgroup (conv_sk_kernel) ------------------------
// The balanced-tile ring kernel above gives every CU one pass over the 786 KB weight slice of a 1024 -> 128 k = 3 GEMM, but its
// main loop runs at a third of the matrix rate: weights (24 KB per K chunk) and activations (8 KB) share one LDS ring filled by
// LDS-DMA, whose issue -> landed latency is ~1 us under load, so the bytes a CU can have in flight (two ring stages) cap the
// stream at ~30 GB/s per CU; 4 x 2 register blocking needs 0.75 KB of LDS fragments per MFMA on top.  Here
//   * the workgroup is 4 waves, ONE per SIMD, with the full 512-register file each (accumulators in AGPRs), and the contraction
//     is split between the waves (the comment inside the kernel has the details): the register file is the weight ring -- the
//     weights are stored in fragment order (dx_pack_frag_major: a fragment is one contiguous KiB), come straight from L2 into
//     registers, every fragment read by exactly ONE wave of the workgroup: 786 KB per CU per launch -- and LDS holds activations only;
//   * the haloed activation slabs go through per-wave LDS-DMA rings issued by the waves themselves (inline asm: hipcc would drain
//     every counted load before the first LDS read that follows a DMA it knows of);
//   * after the last step the partial tiles of the K slices meet through LDS, which leaves wave w with the complete rows of channel
//     block w for the LayerNorm epilogues (forward LayerNorm: dx_conv1d_ln; backward: dx_conv1d_lnbwd), the row-wise code of
//     conv_gemm_kernel run by one 256-thread team.
// The padding rows of the batch (an equal share per workgroup, as in the ring kernel) are zero-filled after the epilogue.
// (Rounds 3-5 split the contraction two ways x two channel halves over ONE shared activation ring with a workgroup barrier per
//  32-channel chunk: 36 % matrix-pipe issue inside its compute phase, 620 cycles of hand-over per chunk; same-box A/B against the
//  loop below: 6.39 -> 6.27 ms per training step, DESIGN 5.  That loop, the main-loop ablation switches and an LDS-staged store of
//  the second GEMM's rows (53.9 vs 51.1 us) were deleted after their measurements.)
constexpr int SK_THREADS = 256;
constexpr int SK4_MAXNA = 6, SK4_S = 3, SK4_NPMAX = (SK4_MAXNA * 32 + 2 + 15) / 16, SK4_WAVE_EL = SK4_S * SK4_NPMAX * 512;
template <int N>
__device__ __forceinline__ void sk_wait_vmcnt_c() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

__device__ __forceinline__ void sk_dma16(const void* gsrc, unsigned lds_dst) {   // one 1-KiB LDS-DMA piece (16 B per lane)
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
template <int LNM>
__global__ __launch_bounds__(SK_THREADS, 1) void conv_sk_kernel(ConvArgs p) {
  typedef bf16_t TC;
  typedef bf16x8 frag_t;
  constexpr int LN = LNM == 3 ? 2 : LNM;
  constexpr bool LNFILM = LNM == 2;
  constexpr int TAPS = 3, HALO = 1, STG_LD = BN + 4, STG_BYTES = 64 * STG_LD * 4;
  constexpr int MAXBLK = SK4_MAXNA;
  constexpr int RING_BYTES = 4 * SK4_WAVE_EL * 2, XCH_BYTES = 24 * 4096;
  constexpr int SMEM_BYTES = RING_BYTES > XCH_BYTES ? (RING_BYTES > STG_BYTES ? RING_BYTES : STG_BYTES) : (XCH_BYTES > STG_BYTES ? XCH_BYTES : STG_BYTES);
  __shared__ __attribute__((aligned(16))) char smem[SMEM_BYTES];
  TC* ring = reinterpret_cast<TC*>(smem);
  float* stage = reinterpret_cast<float*>(smem);
  float* xch = reinterpret_cast<float*>(smem);
  auto lds_at = [](int row, int chunk) { return row * 32 + ((chunk ^ ((row >> 2) & 3)) << 3); };
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, g = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), wk = wave & 1, wc = wave >> 1;
  const int4 e = reinterpret_cast<const int4*>(p.plan)[blockIdx.x];
  const int b = e.x, n0_tile = e.y, h_tile = e.z, fill_per = e.w;
  const int N = p.N, Cin = p.Cin;
  const int len = p.mask_len ? (int)p.mask_len[b] : N;
  // a tile taller than the accumulators of the main loop hold (6 row blocks) is two workgroups' work: blockIdx.y = 1 takes the
  // rows from 128 on (and exits at once for every other tile); the grid is (tiles, 2) there
  const bool tall = h_tile > 32 * SK4_MAXNA;
  if (blockIdx.y && !tall) return;
  const int n0 = n0_tile + (int)blockIdx.y * 128, h = tall ? (blockIdx.y ? h_tile - 128 : 128) : h_tile;
  if (h > 0) {
    const TC* X = reinterpret_cast<const TC*>(p.x) + (size_t)b * N * p.ldx;
    const int nk = Cin >> 5;
    // ---- main loop (round 6).  The contraction (Cin x 3 taps) is split FOUR ways inside the workgroup: wave w takes the 32-channel
    // chunks 4 s + w (s = "step") with all three taps, for ALL rows of the tile (<= 5 blocks of 32) and ALL 128 output channels:
    // <= 5 x 4 MFMA tiles = 320 accumulator registers.  Nothing is shared between the waves until the end:
    //   * the activation slab of a wave's chunk (<= 162 rows x 32 channels, <= 11 KiB) goes through the wave's OWN 3-stage LDS-DMA
    //     ring -- there is no workgroup barrier in the loop, only the wave's own vmcnt waits (hand-counted below);
    //   * the weight fragments (fragment order, dx_pack_frag_major: the four channel blocks of one (chunk, tap, k half) are 4 KiB
    //     contiguous) come from L2 straight into registers, every fragment read by exactly ONE wave: 786 KB per workgroup as before;
    //   * per (tap, k half) "sub-step" a wave reads NA activation fragments from LDS for 4 NA MFMAs (0.25 KB of LDS per MFMA; the
    //     rounds 3-5 loop: 0.5), the fragments of the next sub-step are requested before the MFMAs of this one;
    //   * after the last step the four partial tiles of every (row block, channel block) meet through LDS, two row blocks per pass;
    //     local channel block j of wave w is block j ^ w, so that local 0 is the one the wave keeps (static register indices) and
    //     the sum runs in the fixed order own + (w ^ 1) + (w ^ 2) + (w ^ 3): results stay run-to-run reproducible.
    // Why: one wave per SIMD in lock step with three others (the rounds 3-5 loop: a workgroup barrier per chunk) exposes every latency.
    // Tiles of 129..160 rows (the balanced plan of a B = 48 batch: H = 124..135) also ran that loop's 8-block code path: 48 MFMAs per chunk for 30.
    f32x16 fin[SK4_MAXNA];
    {
      const int nblk = __builtin_amdgcn_readfirstlane((h + 31) >> 5);     // live 32-row blocks, 1 .. 6
      const unsigned ring_base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)smem) +
                                 (unsigned)(wave * SK4_WAVE_EL * 2);
      const TC* ringw = ring + wave * SK4_WAVE_EL;
      // LDS-DMA pieces: 16 rows x 64 B; lane -> row lane >> 2, slot lane & 3, which holds source chunk slot ^ ((row >> 2) & 3) (lds_at;
      // (row >> 2) & 3 == (lane >> 4) & 3 for every piece).  Rows outside the utterance or past the halo read the zero page.
      const int lr = lane >> 2, csrc = (lane & 3) ^ ((lane >> 4) & 3);
      const TC* zp = reinterpret_cast<const TC*>(dx_zero_page);
      asm volatile("" : "+s"(zp));                                        // (an SGPR pair, not a GOT load per piece)
      const int rowoff0 = (n0 - HALO + lr) * (int)p.ldx + csrc * 8, ld16 = 16 * (int)p.ldx;   // elements from X; < 2^31 (plan_check: B * N * ldx)
      const int rlo = n0 == 0 ? 1 : 0, rhi = min(h + TAPS - 1, N - n0 + HALO);
      const TC* wl = reinterpret_cast<const TC*>(p.w_frag) + lane * 8;
      auto mainloop = [&](auto na_tag, auto ks_tag) {
        // KS K slices x (4 / KS) channel groups: wave = wk + KS wc takes the chunks KS s + wk and the channel blocks NCB wc + (j ^ wk),
        // j = 0 .. NCB - 1 (NCB = KS): local block 0 is global block `wave`, the one the wave keeps after the exchange
        constexpr int NA = decltype(na_tag)::value, KS = decltype(ks_tag)::value, NCB = KS;
        constexpr int NP = (NA * 32 + TAPS - 1 + 15) / 16, STAGE_EL = NP * 512;
        constexpr int RQ = NA * NCB >= 10 ? 3 : 6;                         // weight-fragment ring, in sub-steps (6 per step)
        auto dist = [](int q) constexpr { return (NP + 5 - q) / 6; };      // pieces issued in sub-step q
        auto first = [](int q) constexpr { int f = 0; for (int i = 0; i < q; ++i) f += (NP + 5 - i) / 6; return f; };
        constexpr int QL = NP >= 6 ? 5 : NP - 1;                           // last sub-step that issues a piece
        constexpr int P5 = NP - NP / 6;                                    // pieces issued before sub-step 5
        static_assert(3 * STAGE_EL <= SK4_WAVE_EL, "ring stage");
        const int skw = wave % KS, scw = wave / KS;
        const int ns = nk / KS;                                             // steps (launcher: Cin % 128 == 0, Cin >= 256)
        // every workgroup walks the steps in its own rotation: workgroup L runs on XCD L % 8, and the 32 workgroups of an XCD start within a
        // microsecond of each other -- in the same order they would all ask the XCD's L2 for the same weight lines at the same time (one channel
        // serves them one after the other: measured 30 GB/s per CU of L2 hits, a quarter of what the L2 delivers to CUs that read different
        // lines).  The fp32 sums of different tiles then run in different step orders (each still fixed, so results stay reproducible)
        const int soff = (int)((blockIdx.x >> 3) % (unsigned)ns);
        auto kc_of = [&](int s) { int t = s + soff; if (t >= ns) t -= ns; return KS * t + skw; };
        const int jx = skw * 512, jb = scw * NCB * 512;
        f32x16 acc[NA][NCB];
#pragma unroll
        for (int i = 0; i < NA; ++i)
#pragma unroll
          for (int j = 0; j < NCB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        frag_t bq[RQ][NCB], a[2][NA];
        auto piece = [&](int t, int kc, int stg) {
          // (laundered: as loop invariants hipcc keeps 2 NP hoisted pointers in registers -- and spills them; re-deriving one costs ~8 VALU)
          int lrv = lr, ro = rowoff0;
          asm volatile("" : "+v"(lrv), "+v"(ro));
          const int r = t * 16 + lrv;
          const bool ok = (unsigned)(r - rlo) < (unsigned)(rhi - rlo);
          const TC* sp = (ok ? X : zp) + ((ok ? ro + t * ld16 : csrc * 8) + kc * 32);
          sk_dma16(sp, __builtin_amdgcn_readfirstlane(ring_base + (unsigned)((stg * STAGE_EL + t * 512) * 2)));
        };
        auto load_b = [&](int kc, int q, frag_t* d) {
          const TC* wq = wl + (size_t)kc * 12288 + q * 2048 + jb;
#pragma unroll
          for (int j = 0; j < NCB; ++j) d[j] = *reinterpret_cast<const frag_t*>(wq + ((j * 512) ^ jx));
        };
        auto read_a = [&](const TC* Ar, int q, frag_t* d) {
          const int tap = q >> 1, half = q & 1;
#pragma unroll
          for (int i = 0; i < NA; ++i) d[i] = *reinterpret_cast<const frag_t*>(&Ar[lds_at(i * 32 + l31 + tap, half * 2 + g)]);
        };
        // prologue: the slabs of steps 0 and 1, the fragments of the first RQ sub-steps, then the first activation fragments
#pragma unroll
        for (int t = 0; t < NP; ++t) piece(t, kc_of(0), 0);
#pragma unroll
        for (int t = 0; t < NP; ++t) piece(t, kc_of(1), 1);
#pragma unroll
        for (int q = 0; q < RQ; ++q) load_b(kc_of(0), q, bq[q]);
        sk_wait_vmcnt_c<NP + NCB * RQ>();                                  // behind the slab of step 0: the slab of step 1, NCB RQ fragments
        read_a(ringw, 0, a[0]);
        int stg = 0;                                                        // ring stage of step s
        for (int s = 0; s < ns; ++s) {
          const int stg1 = stg + 1 == SK4_S ? 0 : stg + 1, stg2 = stg1 + 1 == SK4_S ? 0 : stg1 + 1;
          const bool dma = s + 2 < ns, more = s + 1 < ns;
          const int kc0 = kc_of(s), kc1 = kc_of(more ? s + 1 : s), kc2 = dma ? kc_of(s + 2) : 0;
          const TC* Ar = ringw + stg * STAGE_EL;
          const TC* An = ringw + stg1 * STAGE_EL;
          auto sub = [&](auto q_tag) {
            constexpr int Q = decltype(q_tag)::value, SL = Q % RQ;
            if (Q < 5) read_a(Ar, Q + 1, a[(Q + 1) & 1]);
            else if (more) {
              // the slab of step s + 1 must have landed.  Behind its last piece in this wave's queue: the fragment loads of the rest
              // of that step (s >= 1: NCB (6 - QL); s == 0: the prologue's NCB RQ), and of this step's sub-steps 0..4 (5 NCB) with
              // the pieces issued beside them (P5, when step s + 2 exists).  hipcc does not see the pieces: its own waits are early.
              if (s == 0) { if (dma) sk_wait_vmcnt_c<NCB * RQ + 5 * NCB + P5>(); else sk_wait_vmcnt_c<NCB * RQ + 5 * NCB>(); }
              else { if (dma) sk_wait_vmcnt_c<NCB * (6 - QL) + 5 * NCB + P5>(); else sk_wait_vmcnt_c<NCB * (6 - QL) + 5 * NCB>(); }
              read_a(An, 0, a[0]);
            }
            __builtin_amdgcn_sched_barrier(0);
            constexpr int HALF = NA / 2;
#pragma unroll
            for (int i = 0; i < HALF; ++i)
#pragma unroll
              for (int j = 0; j < NCB; ++j) dx_mma(acc[i][j], a[Q & 1][i], bq[SL][j]);
            __builtin_amdgcn_sched_barrier(0);
            if (dma) {
#pragma unroll
              for (int t = first(Q); t < first(Q) + dist(Q); ++t) piece(t, kc2, stg2);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = HALF; i < NA; ++i)
#pragma unroll
              for (int j = 0; j < NCB; ++j) dx_mma(acc[i][j], a[Q & 1][i], bq[SL][j]);
            __builtin_amdgcn_sched_barrier(0);
            // refill the slots just read with sub-step Q + RQ (UNCONDITIONAL: past the end it re-reads -- a load that may not execute
            // makes hipcc assume the worst at every use: with a condition around them it drained the whole queue, vmcnt(0), in front of the
            // first MFMA that follows)
            load_b(Q + RQ < 6 ? kc0 : kc1, (Q + RQ) % 6, bq[SL]);
          };
          sub(std::integral_constant<int, 0>{});
          sub(std::integral_constant<int, 1>{});
          sub(std::integral_constant<int, 2>{});
          sub(std::integral_constant<int, 3>{});
          sub(std::integral_constant<int, 4>{});
          sub(std::integral_constant<int, 5>{});
          stg = stg1;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                 // every wave is done with its ring: the exchange buffer reuses it
        // ---- the K slices meet, two row blocks per pass: slot ((wave (KS - 1) + j - 1) * 2 + rbl) of 4 KiB = [register quad][lane][4];
        // fixed order own + (wk ^ 1) [+ (wk ^ 2) + (wk ^ 3)]: results stay run-to-run reproducible
#pragma unroll
        for (int pp = 0; pp < (NA + 1) / 2; ++pp) {
#pragma unroll
          for (int rbl = 0; rbl < 2; ++rbl)
            if (2 * pp + rbl < NA) {
#pragma unroll
              for (int j = 1; j < KS; ++j)
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                  const f32x16& t = acc[2 * pp + rbl][j];
                  *reinterpret_cast<f32x4*>(xch + ((((wave * (KS - 1) + j - 1) * 2 + rbl) * 4 + r4) * 64 + lane) * 4) =
                      f32x4{t[4 * r4], t[4 * r4 + 1], t[4 * r4 + 2], t[4 * r4 + 3]};
                }
            }
          __syncthreads();
#pragma unroll
          for (int rbl = 0; rbl < 2; ++rbl)
            if (2 * pp + rbl < NA) {
              f32x16 t = acc[2 * pp + rbl][0];
#pragma unroll
              for (int d = 1; d < KS; ++d)
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                  const f32x4 v = *reinterpret_cast<const f32x4*>(xch + (((((wave ^ d) * (KS - 1) + d - 1) * 2 + rbl) * 4 + r4) * 64 + lane) * 4);
#pragma unroll
                  for (int e2 = 0; e2 < 4; ++e2) t[4 * r4 + e2] += v[e2];
                }
              fin[2 * pp + rbl] = t;
            }
          __syncthreads();
        }
#pragma unroll
        for (int i = NA; i < SK4_MAXNA; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) fin[i][r] = 0.f;
      };
      constexpr std::integral_constant<int, 4> K4{};
      constexpr std::integral_constant<int, 2> K2{};
      if (nblk > 5) mainloop(std::integral_constant<int, 6>{}, K2);
      else if (nblk == 5) mainloop(std::integral_constant<int, 5>{}, K2);
      else if (nblk == 4) mainloop(std::integral_constant<int, 4>{}, K4);
      else if (nblk == 3) mainloop(std::integral_constant<int, 3>{}, K4);
      else if (nblk == 2) mainloop(std::integral_constant<int, 2>{}, K4);
      else mainloop(std::integral_constant<int, 1>{}, K4);
    }
#define DX_SK_FIN(i) fin[i]
    // ---- LayerNorm epilogue (the PLAN epilogue of conv_gemm_kernel with one 256-thread team)
    constexpr int NCS = LNM == 2 ? 4 : (LNM == 3 ? 2 : 1);
    float csum[NCS][8];
#pragma unroll
    for (int q = 0; q < NCS; ++q)
#pragma unroll
      for (int e2 = 0; e2 < 8; ++e2) csum[q][e2] = 0.f;
    const int cb = 2 * wc + wk;                        // == wave: the channel block whose complete rows this wave holds
    const float bv = p.bias ? p.bias[cb * 32 + l31] : 0.f;
    // second GEMM of the backward variant (LayerNorm backward -> output-projection data gradient, model.py:182-186): the rows this
    // epilogue writes as y_lp are the operand of a 128 -> 128 k = 1 GEMM that used to be the next launch (18 us for 3 us of work).
    // Weights as the A operand (wave = 32 output channels, its 8 fragments in registers), the 64 freshly written rows as B from an
    // LDS image beside the staging buffer: D[channel][row], a lane owns one row and 4 x 4 consecutive channels (8-byte stores).
    // The forward variant does the same with the NEXT block's QKV projection (128 -> 384, model.py:165-171): three channel blocks per
    // wave, bias added in the store.
    constexpr int A2_LD = BN + 8, A2_OFF = 64 * 1024, NC2 = LN == 2 ? 1 : 3;
    static_assert(A2_OFF >= STG_BYTES && A2_OFF + 64 * A2_LD * 2 <= SMEM_BYTES, "second-GEMM operand tile must fit beside the staging buffer");
    const bool gemm2 = p.ln.y2 != nullptr;
    const int n2 = p.ln.n2, ncb2 = __builtin_amdgcn_readfirstlane(n2 >> 7);      // channel blocks per wave (1 or 3)
    TC* a2 = reinterpret_cast<TC*>(smem + A2_OFF);
    frag_t w2f[NC2][8];
    if (gemm2) {
#pragma unroll
      for (int c = 0; c < NC2; ++c)
        if (c < ncb2) {
          const TC* w2 = reinterpret_cast<const TC*>(p.ln.w2) + (size_t)((c * 4 + wave) * 32 + l31) * BN + g * 8;
#pragma unroll
          for (int ks = 0; ks < 8; ++ks) w2f[c][ks] = *reinterpret_cast<const frag_t*>(w2 + ks * 16);
        }
    }
    // the per-channel operands of the row passes depend on (b, channel segment) only: loaded ONCE here.  Inside the passes they sat
    // behind the stores of the pass before (the compiler cannot prove that y / s_out do not alias gamma / beta / film), one exposed
    // L2 round trip per pass and slab
    const int cl_h = (tid & 15) * 8;
    const f32x8 gm_h = raw_load8<float>(p.ln.gamma + cl_h);
    f32x8 bt_h = gm_h, fg_h = gm_h, fb_h = gm_h;
    if (LN == 1 || LNFILM) bt_h = raw_load8<float>(p.ln.beta + cl_h);
    if (p.ln.film && (LN == 1 || LNFILM)) fg_h = raw_load8<float>(p.ln.film + (size_t)b * p.ln.ldf + cl_h);
    if (p.ln.film && LN == 1) fb_h = raw_load8<float>(p.ln.film + (size_t)b * p.ln.ldf + BN + cl_h);
    const bool vres = LN == 1 && p.ln.res_mean != nullptr;
    f32x8 rg_h = gm_h, rb_h = gm_h;
    if (vres) { rg_h = raw_load8<float>(p.ln.res_gamma + cl_h); rb_h = raw_load8<float>(p.ln.res_beta + cl_h); }
#pragma unroll
    for (int i = 0; i < (MAXBLK + 1) / 2; ++i) {
      if (i * 64 >= h) break;                          // workgroup-uniform: the barriers below stay matched
#pragma unroll
      for (int rb = 0; rb < 2; ++rb)
        if (2 * i + rb < MAXBLK) {
#pragma unroll
          for (int r = 0; r < 16; ++r) stage[(rb * 32 + dx_acc_row(r, g)) * STG_LD + cb * 32 + l31] = DX_SK_FIN(2 * i + rb)[r] + bv;
        }
      f32x8 pf_a[4], pf_b[4];
      float pf_m[4], pf_r[4];
#pragma unroll
      for (int pass = 0; pass < 4; ++pass) {           // the global inputs of all four passes are requested before the barrier
        const int sr = (tid >> 4) + pass * 16, trow = i * 64 + sr;
        const int n = n0 + trow, cl = (tid & 15) * 8;
        if (n < N && trow < h) {
          const size_t rowg = (size_t)b * N + n, offl = rowg * BN + cl;
          if (LN == 2) {
            pf_a[pass] = raw_load8<float>(p.ln.y + offl);
            pf_b[pass] = raw_load8<float>(p.ln.s_out + offl);
            pf_m[pass] = p.ln.mean[rowg];
            pf_r[pass] = p.ln.rstd[rowg];
          } else {
            pf_a[pass] = raw_load8<float>(p.ln.residual + offl);
            if (vres) { pf_m[pass] = p.ln.res_mean[rowg]; pf_r[pass] = p.ln.res_rstd[rowg]; }
          }
        }
      }
      __syncthreads();
#pragma unroll
      for (int pass = 0; pass < 4; ++pass) {
        const int sr = (tid >> 4) + pass * 16, trow = i * 64 + sr;
        const int n = n0 + trow, cl = (tid & 15) * 8;
        if (n < N && trow < h) {
          float v[8];
          const f32x4 lo = *reinterpret_cast<const f32x4*>(&stage[sr * STG_LD + cl]);
          const f32x4 hi = *reinterpret_cast<const f32x4*>(&stage[sr * STG_LD + cl + 4]);
          v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3]; v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
          const size_t rowg = (size_t)b * N + n, offl = rowg * BN + cl;
          if (LN == 2) {        // fused LayerNorm BACKWARD: v + residual gradient = dL/d(LN output) of this row
            {
              const f32x8 r = pf_a[pass];
#pragma unroll
              for (int e2 = 0; e2 < 8; ++e2) v[e2] = n < len ? v[e2] + r[e2] : 0.f;     // masked_fill rows carry no gradient
            }
            const f32x8 sv = pf_b[pass];
            const float mean = pf_m[pass], rstd = pf_r[pass];
            const f32x8 gm = gm_h;
            float xh[8], s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int e2 = 0; e2 < 8; ++e2) xh[e2] = (sv[e2] - mean) * rstd;
            if (LNFILM) {                                         // y = fg * LN + fb
              const f32x8 fg = fg_h, bt = bt_h;
#pragma unroll
              for (int e2 = 0; e2 < 8; ++e2) {
                csum[LNFILM ? 2 : 0][e2] += v[e2] * (xh[e2] * gm[e2] + bt[e2]);
                csum[LNFILM ? 3 : 0][e2] += v[e2];
                v[e2] *= fg[e2];
              }
            }
#pragma unroll
            for (int e2 = 0; e2 < 8; ++e2) {
              csum[0][e2] += v[e2] * xh[e2];
              csum[NCS > 1 ? 1 : 0][e2] += v[e2];
              v[e2] *= gm[e2];
              s1 += v[e2];
              s2 += v[e2] * xh[e2];
            }
            s1 = dx_row16_sum(s1); s2 = dx_row16_sum(s2);
            s1 *= 1.f / BN; s2 *= 1.f / BN;
#pragma unroll
            for (int e2 = 0; e2 < 8; ++e2) v[e2] = rstd * (v[e2] - s1 - xh[e2] * s2);
            store8<float>(p.ln.y + offl, v);                      // ds, in place of the residual gradient
            if (p.ln.p_pre > 0.f) {
              const uint32_t th = dx_drop_th8(p.ln.p_pre), key = dx_key32(dx_seed_eff(p.ln.seed_pre, p.ln.step), 0);
              const float sc = dx_drop_inv_keep8(th);
#pragma unroll
              for (int e2 = 0; e2 < 8; ++e2) v[e2] = dx_keep_elem(key, (uint32_t)rowg * BN + cl + e2, th) ? v[e2] * sc : 0.f;
            }
            store8<bf16_t>(reinterpret_cast<bf16_t*>(p.ln.y_lp) + offl, v);
            if (gemm2) store8<bf16_t>(a2 + sr * A2_LD + cl, v);
          } else {              // fused LayerNorm: 16 lanes hold one complete 128-channel row
            if (p.ln.p_pre > 0.f) {
              const uint32_t th = dx_drop_th8(p.ln.p_pre), key = dx_key32(dx_seed_eff(p.ln.seed_pre, p.ln.step), 0);
              const float sc = dx_drop_inv_keep8(th);
#pragma unroll
              for (int e2 = 0; e2 < 8; ++e2) v[e2] = dx_keep_elem(key, (uint32_t)rowg * BN + cl + e2, th) ? v[e2] * sc : 0.f;
            }
            {
              f32x8 r = pf_a[pass];
              if (vres) {   // the residual stream = mask(LayerNorm(s)) of the launch that produced s: same expression as its epilogue
                const float rm = pf_m[pass], rr = pf_r[pass];
#pragma unroll
                for (int e2 = 0; e2 < 8; ++e2) r[e2] = n < len ? (r[e2] - rm) * rr * rg_h[e2] + rb_h[e2] : 0.f;
              }
#pragma unroll
              for (int e2 = 0; e2 < 8; ++e2) v[e2] += r[e2];
            }
            if (p.ln.s_out) store8<float>(p.ln.s_out + offl, v);
            float sum = 0.f;
#pragma unroll
            for (int e2 = 0; e2 < 8; ++e2) sum += v[e2];
            sum = dx_row16_sum(sum);
            const float mean = sum * (1.f / BN);
            float sq = 0.f;
#pragma unroll
            for (int e2 = 0; e2 < 8; ++e2) { const float d = v[e2] - mean; sq += d * d; }
            sq = dx_row16_sum(sq);
            const float rstd = rsqrtf(sq * (1.f / BN) + 1e-5f);
            if (p.ln.mean && cl == 0) { p.ln.mean[rowg] = mean; p.ln.rstd[rowg] = rstd; }
            const f32x8 gm = gm_h, bt = bt_h;
#pragma unroll
            for (int e2 = 0; e2 < 8; ++e2) v[e2] = (v[e2] - mean) * rstd * gm[e2] + bt[e2];
            if (p.ln.film) {
              const f32x8 fg = fg_h, fb = fb_h;
#pragma unroll
              for (int e2 = 0; e2 < 8; ++e2) v[e2] = fg[e2] * v[e2] + fb[e2];
            }
            if (n >= len) {
#pragma unroll
              for (int e2 = 0; e2 < 8; ++e2) v[e2] = 0.f;
            }
            if (p.ln.y) store8<float>(p.ln.y + offl, v);
            if (p.ln.y_lp) store8<bf16_t>(reinterpret_cast<bf16_t*>(p.ln.y_lp) + offl, v);
            if (gemm2) store8<bf16_t>(a2 + sr * A2_LD + cl, v);
          }
        } else if (gemm2) {                            // rows outside the tile / the tensor: zeros in the operand image
          const float z8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
          store8<bf16_t>(a2 + sr * A2_LD + cl, z8);
        }
      }
      __syncthreads();
      if (gemm2) {   // (the next iteration writes the image only behind its own barrier, i.e. after every wave has read it)
        // all (channel block, row block) products of the slab at once: 2 NC2 independent accumulators, every operand fragment of the
        // slab read from LDS ONCE (round 5 ran them one after the other: 8 dependent MFMAs per tile, the fragments re-read per channel block)
        typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
        f32x16 d2[NC2][2];
#pragma unroll
        for (int c = 0; c < NC2; ++c)
#pragma unroll
          for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int r = 0; r < 16; ++r) d2[c][rb][r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
          const frag_t x0 = *reinterpret_cast<const frag_t*>(a2 + l31 * A2_LD + ks * 16 + g * 8);
          const frag_t x1 = *reinterpret_cast<const frag_t*>(a2 + (32 + l31) * A2_LD + ks * 16 + g * 8);
#pragma unroll
          for (int c = 0; c < NC2; ++c)
            if (c < ncb2) {
              dx_mma(d2[c][0], w2f[c][ks], x0);
              dx_mma(d2[c][1], w2f[c][ks], x1);
            }
        }
#pragma unroll
        for (int c = 0; c < NC2; ++c) {
          if (c >= ncb2) break;
          const int cw = (c * 4 + wave) * 32;                     // this wave's 32 output channels of channel group c
          f32x4 bj4[4];                                           // bias of the lane's channels cw + 4 g + 8 j + 0..3 (MFMA layout)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            bj4[j] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (p.ln.b2) bj4[j] = *reinterpret_cast<const f32x4*>(p.ln.b2 + cw + 4 * g + 8 * j);
          }
#pragma unroll
          for (int rb = 0; rb < 2; ++rb) {
            // bf16 pairs, then two v_permlane32_swap per 16 channels (conv_wreg_kernel's epilogue): the lane ends up with channels
            // cw + 8 g + 0..7 and cw + 16 + 8 g + 0..7 of its row -- two 16-byte stores instead of four 8-byte ones
            typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
            typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
            uint32_t P[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
              const bf16x2 pr = {(bf16_t)(d2[c][rb][2 * k] + bj4[k >> 1][(2 * k) & 3]), (bf16_t)(d2[c][rb][2 * k + 1] + bj4[k >> 1][(2 * k + 1) & 3])};
              P[k] = __builtin_bit_cast(uint32_t, pr);
            }
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
              for (int k = 0; k < 2; ++k) {
                const u32x2 sw = __builtin_amdgcn_permlane32_swap(P[4 * h2 + k], P[4 * h2 + 2 + k], false, false);
                P[4 * h2 + k] = sw[0];
                P[4 * h2 + 2 + k] = sw[1];
              }
            // (through an LDS image of the slab, whole rows per store instruction: measured 53.9 vs 51.1 us -- the 8 us the QKV rows cost are
            //  their 23 MB in a write-bound epilogue, not the shape of the store instructions)
            const int trow = i * 64 + rb * 32 + l31, n = n0 + trow;
            if (trow < h && n < N) {
              TC* yo = reinterpret_cast<TC*>(p.ln.y2) + ((size_t)b * N + n) * n2 + cw + 8 * g;
              *reinterpret_cast<u32x4*>(yo) = u32x4{P[0], P[1], P[2], P[3]};
              *reinterpret_cast<u32x4*>(yo + 16) = u32x4{P[4], P[5], P[6], P[7]};
            }
          }
        }
      }
    }
    if (LN == 2) {   // column sums: 16 row-threads per channel segment -> LDS -> one atomic per channel per workgroup
      for (int q = 0; q < NCS; ++q)
#pragma unroll
        for (int e2 = 0; e2 < 8; ++e2) stage[(q * 16 + (tid >> 4)) * BN + (tid & 15) * 8 + e2] = csum[q][e2];
      __syncthreads();
      for (int idx = tid; idx < NCS * BN; idx += SK_THREADS) {
        const int q = idx / BN, c = idx - q * BN;
        float t = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) t += stage[(q * 16 + r) * BN + c];
        if (q == 0) atomicAdd(p.ln.dgamma + c, t);
        else if (q == 1) atomicAdd(p.ln.dbeta + c, t);
        else atomicAdd(p.ln.dfilm + (size_t)b * p.ln.lddf + (q == 3 ? BN : 0) + c, t);
      }
    }
  }
#undef DX_SK_FIN
  // ---- padding fill: the batch's padding rows, flattened utterance by utterance, are split evenly over the workgroups; this one
  // owns [lo, hi).  Each wave finds the utterances its range touches with a wave scan over the lengths, and the 256 threads share
  // the 16-byte segments of those rows.
  if (blockIdx.y == 0) {
    const long lo = (long)blockIdx.x * fill_per, hi = lo + fill_per;
    long carry = 0;
    for (int base = 0; base < p.B && carry < hi; base += 64) {
      const int ub = base + lane;
      const int ulen = ub < p.B ? (int)p.skip_len[ub] : N;
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
        const int first = __shfl(N - dead + (int)(fs - ustart), src_lane, 64);
        int cntr = __shfl((int)(fe - fs), src_lane, 64);
        cntr = min(cntr, dx_fill_end((int)p.skip_len[fb], N) - first);   // dead rows past the fill end stay unwritten (dx_common.h)
        float z[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (p.ln.y2) {                                   // rows of the second GEMM's output (n2 channels, bf16)
          const int segs = p.ln.n2 >> 3;
          for (int c = tid; c < cntr * segs; c += SK_THREADS) {
            const int n = first + c / segs, cl = (c % segs) * 8;
            store8<bf16_t>(reinterpret_cast<bf16_t*>(p.ln.y2) + ((size_t)fb * N + n) * p.ln.n2 + cl, z);
          }
        }
        for (int c = tid; c < cntr * (BN / 8); c += SK_THREADS) {
          const int n = first + (c >> 4), cl = (c & 15) * 8;
          const size_t off = ((size_t)fb * N + n) * BN + cl;
          if (LN == 2 || p.ln.y) store8<float>(p.ln.y + off, z);
          if (LN == 2 || p.ln.y_lp) store8<bf16_t>(reinterpret_cast<bf16_t*>(p.ln.y_lp) + off, z);
          if (LN == 1) {
            if (p.ln.s_out) store8<float>(p.ln.s_out + off, z);
            if (p.ln.mean && cl == 0) { p.ln.mean[(size_t)fb * N + n] = 0.f; p.ln.rstd[(size_t)fb * N + n] = 0.f; }
          }
        }
      }
      carry += __shfl(incl, 63, 64);
    }
  }
}


// ---- wide k = 3 GEMM (Cout a multiple of 256, long contraction: the pre-net's 1024 -> 1024 conv and its data gradient) on the
// full register file: 256 rows x 256 channels per 4-wave workgroup, ONE wave per SIMD, wave (wr, wc) = 128 rows x 128 channels =
// 4 x 4 MFMA tiles = 256 accumulator registers.  Same ingredients as conv_sk_kernel: the haloed activation tile (258 rows x 32
// channels per chunk) through a 3-stage LDS-DMA ring issued by the waves themselves, the weights in fragment order
// (dx_pack_frag_major) from L2 straight into registers -- a ring of 6 k-steps = one chunk (4 fragments each, 96 registers): the
// slot a k-step has just read is refilled with the same k-step of the next chunk -- and the per-workgroup rotation of the chunk order.  Per k-step a wave
// reads 4 activation fragments from LDS for 16 MFMAs (0.25 KB of LDS per MFMA; the 128-channel tiles of conv_gemm_kernel need 0.75).
// L2 -> CU traffic per launch = 2 bytes x M N K x (1 / 256 + 1 / 256): half of what 256 x 128 tiles fetch.
// Epilogue: bias, ReLU, rows past length + 2 zeroed; a wave stages one 32-row x 128-channel slab at a time through its own LDS
// region and stores whole 256-byte row segments in bf16.
constexpr int WD_THREADS = 256, WD_S = 3, WD_RING = 6, WD_MAXP = 5;
__device__ __forceinline__ void wd_wait_vmcnt(int n) {
  switch (n) {
#define DX_VMW(n) case n: asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory"); break;
    DX_VMW(48) DX_VMW(49) DX_VMW(50) DX_VMW(51) DX_VMW(52) DX_VMW(53)
#undef DX_VMW
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
}

__global__ __launch_bounds__(WD_THREADS, 1) void conv_wide_kernel(ConvArgs p) {
  typedef bf16_t TC;
  typedef bf16x8 frag_t;
  constexpr int TAPS = 3, HALO = 1, BMW = 256, AROWS = BMW + TAPS - 1, AR16 = (AROWS + 15) & ~15, STAGE_EL = AR16 * 32;
  constexpr int SLAB_LD = 128 + 4;
  constexpr int RING_BYTES = WD_S * STAGE_EL * 2, SLAB_BYTES = 4 * 32 * SLAB_LD * 4;
  __shared__ __attribute__((aligned(16))) char smem[RING_BYTES > SLAB_BYTES ? RING_BYTES : SLAB_BYTES];
  TC* ring = reinterpret_cast<TC*>(smem);
  auto lds_at = [](int row, int chunk) { return row * 32 + ((chunk ^ ((row >> 2) & 3)) << 3); };
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, g = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), wr = wave >> 1, wc = wave & 1;
  const int N = p.N, Cin = p.Cin, Cout = p.Cout;
  // position tiles of the batch from dx_conv_tile_plan (rows < length + halo of every utterance, cut into equal pieces of <= 256 rows such
  // that the tile count is a multiple of 64 = 256 CUs / 4 channel tiles); channel tile slowest: consecutive workgroups (one per XCD in
  // turn) share a channel tile, so an XCD's L2 holds one 1.5 MB weight slice at a time
  const int pt = blockIdx.x % p.plan_tiles, ct = blockIdx.x / p.plan_tiles;
  const int4 e = reinterpret_cast<const int4*>(p.plan)[pt];
  const int b = e.x, n0 = e.y, h = e.z, fill_per = e.w;
  const int co_w = ct * 256 + wc * 128;                              // first channel of this wave
  TC* Y = reinterpret_cast<TC*>(p.y);
  if (h > 0) {
  f32x16 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][c][r] = 0.f;
  const TC* X = reinterpret_cast<const TC*>(p.x) + (size_t)b * N * p.ldx;
  const int nk = Cin >> 5;
  const int nA = (h + TAPS - 1 + 15) >> 4;
  const int mine = __builtin_amdgcn_readfirstlane(nA > wave ? (nA - wave + 3) >> 2 : 0);
  const TC* src[WD_MAXP];
  unsigned dst[WD_MAXP];
#pragma unroll
  for (int t = 0; t < WD_MAXP; ++t) {
    const int q = wave + 4 * t;
    const int r = q * 16 + (lane >> 2);
    const int c = (lane & 3) ^ ((r >> 2) & 3);
    const int n = n0 + r - HALO;
    const TC* sp = reinterpret_cast<const TC*>(dx_zero_page) + c * 8;
    if (q < nA && r < h + TAPS - 1 && n >= 0 && n < N) sp = X + (long)n * p.ldx + c * 8;
    src[t] = sp;
    dst[t] = (unsigned)(q * 512 * 2);
  }
  const unsigned ring_base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)smem);
  auto issue_dma = [&](int kc, int buf) {
#pragma unroll
    for (int t = 0; t < WD_MAXP; ++t)
      if (t < mine) sk_dma16(src[t] + kc * 32, __builtin_amdgcn_readfirstlane(ring_base + (unsigned)(buf * STAGE_EL * 2) + dst[t]));
  };
  // rotation of the chunk order (see conv_sk_kernel) by POSITION tile only: the channel tiles of one position tile run on the same
  // XCD (plan_tiles % 8 == 0) at the same time and read the same activation chunks -- in the same chunk order the first one
  // pulls a chunk into the XCD's L2 and the others hit it (with the rotation keyed on blockIdx they walked the chunks 8 apart and
  // each fetched the activation tile for itself: FETCH_SIZE 188 MB per launch for 61 MB of activations)
  const int koff = (int)((pt >> 3) % (unsigned)nk);
  auto kc_of = [&](int it) { const int k = it + koff; return k >= nk ? k - nk : k; };
  // fragment (k-step q = chunk * 6 + tap * 2 + half, channel block c) at q * (Cout / 32) * 512 + c * 512 elements
  const size_t qstride = (size_t)(Cout >> 5) * 512;
  const TC* wp = reinterpret_cast<const TC*>(p.w_frag) + (size_t)(co_w >> 5) * 512 + lane * 8;
  frag_t bq[WD_RING][4];
  auto load_b = [&](int it, int ks6, frag_t* d) {   // k-step ks6 of chunk `it` (in this workgroup's rotation; past the end: the last chunk again)
    const TC* base = wp + (size_t)(kc_of(it < nk ? it : nk - 1) * 6 + ks6) * qstride;
#pragma unroll
    for (int c = 0; c < 4; ++c) d[c] = *reinterpret_cast<const frag_t*>(base + c * 512);
  };
  // 32-row blocks interleaved over the two wave rows (block 2 i + wr is wave row wr's i-th), so that a tile of any height splits evenly
  const int nblk = (h + 31) >> 5;
  const int nact = __builtin_amdgcn_readfirstlane((nblk - wr + 1) >> 1);
  const bool counted = nk >= 8;
#pragma unroll
  for (int st = 0; st < WD_S - 1; ++st)
    if (st < nk) issue_dma(kc_of(st), st);
#pragma unroll
  for (int s0 = 0; s0 < WD_RING; ++s0) load_b(0, s0, bq[s0]);
  // per chunk a wave issues [DMA pieces of chunk it + S - 1] [24 fragment loads]: when the pieces of chunk `it` must have landed,
  // 24 (S - 1) fragment loads + the pieces of the iteration in between may be in flight: 48 + mine * min(1, nk - 1 - it) (S = 3)
  auto chunk = [&](int it, auto na_tag) {
    constexpr int NA = decltype(na_tag)::value;
    const int behind = nk - 1 - it;
    if (counted) wd_wait_vmcnt(48 + mine * (behind > 1 ? 1 : behind));
    else wd_wait_vmcnt(0);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (it + WD_S - 1 < nk) issue_dma(kc_of(it + WD_S - 1), (it + WD_S - 1) % WD_S);
    const TC* Ar = ring + (it % WD_S) * STAGE_EL;
    frag_t a[2][NA > 0 ? NA : 1];
    if constexpr (NA > 0) {
#pragma unroll
      for (int i = 0; i < NA; ++i) a[0][i] = *reinterpret_cast<const frag_t*>(&Ar[lds_at((2 * i + wr) * 32 + l31, g)]);
    }
#pragma unroll
    for (int ks6 = 0; ks6 < 6; ++ks6) {
      if constexpr (NA > 0) {
        if (ks6 + 1 < 6) {                           // the next k-step's activation fragments are requested before this one's MFMAs
          const int tn = (ks6 + 1) >> 1, kn = (ks6 + 1) & 1;
#pragma unroll
          for (int i = 0; i < NA; ++i) a[(ks6 + 1) & 1][i] = *reinterpret_cast<const frag_t*>(&Ar[lds_at((2 * i + wr) * 32 + l31 + tn, kn * 2 + g)]);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < NA; ++i)
#pragma unroll
          for (int c = 0; c < 4; ++c) dx_mma(acc[i][c], a[ks6 & 1][i], bq[ks6][c]);
        __builtin_amdgcn_sched_barrier(0);
      }
      load_b(it + 1, ks6, bq[ks6]);                  // the same k-step of the next chunk
    }
  };
  auto mainloop = [&](auto na_tag) {
    for (int it = 0; it < nk; ++it) chunk(it, na_tag);
  };
  if (nact >= 4) mainloop(std::integral_constant<int, 4>{});
  else if (nact == 3) mainloop(std::integral_constant<int, 3>{});
  else if (nact == 2) mainloop(std::integral_constant<int, 2>{});
  else if (nact == 1) mainloop(std::integral_constant<int, 1>{});
  else mainloop(std::integral_constant<int, 0>{});
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();                                   // the ring is dead: every wave stages its slabs through its own region

  const bool relu = p.flags & DX_CONV_RELU;
  float* slab = reinterpret_cast<float*>(smem) + (size_t)wave * (32 * SLAB_LD);
  float bv[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) bv[c] = p.bias ? p.bias[co_w + c * 32 + l31] : 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r0 = (2 * i + wr) * 32;                // first tile row of this 32-row block
    if (r0 >= h) break;                              // wave-uniform; the staging region is wave-private: no workgroup barrier
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float v = acc[i][c][r] + bv[c];
        if (relu) v = fmaxf(v, 0.f);
        slab[dx_acc_row(r, g) * SLAB_LD + c * 32 + l31] = v;
      }
    asm volatile("" ::: "memory");
#pragma unroll
    for (int pass = 0; pass < 8; ++pass) {           // 16 lanes x 16 bytes = one 256-byte row segment per store instruction
      const int row = pass * 4 + (lane >> 4), cl = (lane & 15) * 8;
      const f32x4 lo = *reinterpret_cast<const f32x4*>(&slab[row * SLAB_LD + cl]);
      const f32x4 hi = *reinterpret_cast<const f32x4*>(&slab[row * SLAB_LD + cl + 4]);
      if (r0 + row < h) {
        const float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        store8<bf16_t>(Y + ((size_t)b * N + n0 + r0 + row) * p.ldy + co_w + cl, v);
      }
    }
    asm volatile("" ::: "memory");
  }
  }
  // ---- padding fill (rows past length + halo of every utterance: zeros), an equal share of the flattened padding rows per position tile,
  // this channel tile's 256 columns of them (see conv_sk_kernel)
  {
    const long lo = (long)pt * fill_per, hi = lo + fill_per;
    const int halo = p.flags >> 8;
    long carry = 0;
    for (int base = 0; base < p.B && carry < hi; base += 64) {
      const int ub = base + lane;
      int ulen = ub < p.B ? (int)p.skip_len[ub] : N;
      ulen = (ulen < 0 ? 0 : ulen) + halo;
      const int dead = ub < p.B ? N - (ulen > N ? N : ulen) : 0;
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
        const int first = __shfl(N - dead + (int)(fs - ustart), src_lane, 64);
        int cntr = __shfl((int)(fe - fs), src_lane, 64);
        cntr = min(cntr, dx_fill_end((int)p.skip_len[fb], N) - first);   // dead rows past the fill end stay unwritten (dx_common.h)
        const float z[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int c = tid; c < cntr * 32; c += WD_THREADS)
          store8<bf16_t>(Y + ((size_t)fb * N + first + (c >> 5)) * p.ldy + ct * 256 + (c & 31) * 8, z);
      }
      carry += __shfl(incl, 63, 64);
    }
  }
}

template <typename TA, typename TC, typename TO, typename TG>
bool try_weight_stationary(const ConvArgs& a, int B, int taps, hipStream_t s) {
  if constexpr (sizeof(TC) != 2 || sizeof(TA) != 2) return false;
  else {
    if (a.ln.enabled || (a.flags & (DX_CONV_TRANSPOSED_OUT | DX_CONV_ACCUMULATE))) return false;
    if (a.Cin != 128 || a.Cout % WR_BN || a.ldy % 8 || a.ldx % 8) return false;
    const int ztiles = a.Cout / WR_BN;
    // about one workgroup per CU; more position groups than tiles only adds weight loads
    const long tiles = (long)dx_cdiv(a.N, WR_BM) * B;
    int ngrp = 256 / ztiles;
    if (ngrp > tiles) ngrp = (int)tiles;
    if (ngrp < 1) ngrp = 1;
    dim3 grid(ngrp * ztiles), block(WR_THREADS);
    const bool relu = a.flags & DX_CONV_RELU, gate = a.gate != nullptr;
    if ((size_t)a.N * a.ldy * 4 >= (1ull << 32) || (size_t)a.N * a.ldx * 2 >= (1ull << 32)) return false;   // 32-bit buffer offsets
    if (a.relu_bits || a.gate_bits) {   // dx_conv1d_relu_bits: one bit per element written by the ReLU / read as the gate
      if constexpr (sizeof(TO) == 2) {
        if (taps != 3 || gate || (a.relu_bits != nullptr) == (a.gate_bits != nullptr) || (a.relu_bits && !relu) || (a.gate_bits && relu)) return false;
        if (a.relu_bits) hipLaunchKernelGGL((conv_wreg_kernel<TO, TG, 3, true, false, true>), grid, block, 0, s, a, ngrp);
        else hipLaunchKernelGGL((conv_wreg_kernel<TO, TG, 3, false, true, true>), grid, block, 0, s, a, ngrp);
        return true;
      } else {
        return false;
      }
    }
#define DX_WREG_LAUNCH(T, R, GT) hipLaunchKernelGGL((conv_wreg_kernel<TO, TG, T, R, GT>), grid, block, 0, s, a, ngrp)
    if (taps == 3) {
      if (relu && gate) DX_WREG_LAUNCH(3, true, true); else if (relu) DX_WREG_LAUNCH(3, true, false);
      else if (gate) DX_WREG_LAUNCH(3, false, true); else DX_WREG_LAUNCH(3, false, false);
    } else {
      if (relu && gate) DX_WREG_LAUNCH(1, true, true); else if (relu) DX_WREG_LAUNCH(1, true, false);
      else if (gate) DX_WREG_LAUNCH(1, false, true); else DX_WREG_LAUNCH(1, false, false);
    }
#undef DX_WREG_LAUNCH
    return true;
  }
}

template <typename TA, typename TC, typename TO, typename TG, int LN = 0>
int launch_taps(const ConvArgs& a, int B, int taps, hipStream_t s) {
  const int ztiles = dx_cdiv(a.Cout, BN);
  // Narrow-output GEMMs (Cout <= 128, k = 3): 128-row tiles stage the weight chunk once per 128 rows (the LDS write of the
  // weight tile is the busiest part of the kernel: 818 vs 609 TFLOP/s on a dense B = 256 problem) but need enough tiles to
  // fill the chip; 64-row tiles otherwise.  Measured in the training step: B = 48 equal, B = 128 +2 % for 128 rows.
  const int narrow_mi = (long)B * a.N > 64000 ? 2 : 1;
  if constexpr (LN != 0) {   // LayerNorm epilogues: one channel tile (Cout = 128)
    constexpr int LNB = LN == 2 ? 3 : LN;             // backward without FiLM gradients: fewer registers
    const bool film = LN == 2 && a.ln.film != nullptr;
    if constexpr (sizeof(TA) == 2 && sizeof(TC) == 2) {
      // split-K workgroups on the same balanced tiles, weights in fragment order -- while the batch is ONE round of tiles (B * N <= 256 CUs x
      // 256 rows): measured 0.35 % of the B = 48 step faster than the ring kernel (frame level 46 vs 50 us, phoneme level 27 vs 33 us),
      // 2.5 % of the B = 256 step slower (several rounds of 256-row tiles: the ring kernel's two epilogue teams win there)
      if (a.plan && taps == 3 && a.w_frag && a.Cin >= 256 && a.Cin % 128 == 0 && (long)B * a.N <= 256L * 256) {
        // tiles of more than 6 row blocks (possible when N > 192) are split between blockIdx.y = 0 and 1
        dim3 gridp((unsigned)a.plan_tiles, a.N > 32 * SK4_MAXNA ? 2u : 1u);
        if (film) hipLaunchKernelGGL((conv_sk_kernel<LN>), gridp, dim3(SK_THREADS), 0, s, a);
        else hipLaunchKernelGGL((conv_sk_kernel<LNB>), gridp, dim3(SK_THREADS), 0, s, a);
        DX_LAUNCH_CHECK();
        return DX_OK;
      }
      if (a.plan && taps == 3) {   // balanced 256-row tiles + padding-fill workgroups (dx_conv_tile_plan)
        dim3 gridp((unsigned)a.plan_tiles);
        if (film) hipLaunchKernelGGL((conv_gemm_kernel<TA, TC, TO, TG, 3, 4, 32, LN, DX_PLAN_RING>), gridp, dim3(2 * NTHREADS), 0, s, a);
        else hipLaunchKernelGGL((conv_gemm_kernel<TA, TC, TO, TG, 3, 4, 32, LNB, DX_PLAN_RING>), gridp, dim3(2 * NTHREADS), 0, s, a);
        DX_LAUNCH_CHECK();
        return DX_OK;
      }
      if constexpr (LN == 2) {
        if (a.plan && taps == 1) {   // k = 1 data gradient + LayerNorm backward (QKV projection, K = 384) on the same tiles
          dim3 gridp((unsigned)a.plan_tiles);
          if (film) hipLaunchKernelGGL((conv_gemm_kernel<TA, TC, TO, TG, 1, 4, 32, LN, 3>), gridp, dim3(2 * NTHREADS), 0, s, a);
          else hipLaunchKernelGGL((conv_gemm_kernel<TA, TC, TO, TG, 1, 4, 32, LNB, 3>), gridp, dim3(2 * NTHREADS), 0, s, a);
          DX_LAUNCH_CHECK();
          return DX_OK;
        }
      }
    }
    if (narrow_mi == 2 && taps == 3) {
      const long pt2 = (long)dx_cdiv(a.N, 128) * B;
      dim3 grid2((unsigned)(((pt2 + 7) / 8) * 8));
      if (film) hipLaunchKernelGGL((conv_gemm_kernel<TA, TC, TO, TG, 3, 2, 32, LN>), grid2, dim3(NTHREADS), 0, s, a);
      else hipLaunchKernelGGL((conv_gemm_kernel<TA, TC, TO, TG, 3, 2, 32, LNB>), grid2, dim3(NTHREADS), 0, s, a);
      DX_LAUNCH_CHECK();
      return DX_OK;
    }
    const long ptiles = (long)dx_cdiv(a.N, 64) * B;
    dim3 grid((unsigned)(((ptiles + 7) / 8) * 8)), block(NTHREADS);
    if (taps == 1) {
      if (film) hipLaunchKernelGGL((conv_gemm_kernel<TA, TC, TO, TG, 1, 1, CG_K1_BK, LN>), grid, block, 0, s, a);
      else hipLaunchKernelGGL((conv_gemm_kernel<TA, TC, TO, TG, 1, 1, CG_K1_BK, LNB>), grid, block, 0, s, a);
    } else {
      if (film) hipLaunchKernelGGL((conv_gemm_kernel<TA, TC, TO, TG, 3, 1, 32, LN>), grid, block, 0, s, a);
      else hipLaunchKernelGGL((conv_gemm_kernel<TA, TC, TO, TG, 3, 1, 32, LNB>), grid, block, 0, s, a);
    }
    DX_LAUNCH_CHECK();
    return DX_OK;
  } else {
    if (try_weight_stationary<TA, TC, TO, TG>(a, B, taps, s)) { DX_LAUNCH_CHECK(); return DX_OK; }
    // Wide k = 3 GEMMs with a long contraction (prenet 1024 -> 1024): 256-row tiles (MI = 4, a wave owns 128 x 64) when
    // that still leaves >= 4 workgroups per CU.  The kernel is bound by what a CU can fetch from L2 (~30 B/clk),
    // and a taller tile re-uses the taps x 128-channel weight chunk for twice the positions: 930 vs 810 TFLOP/s.
    static int forced_wide = getenv("DX_CONV_WIDE_MI") ? atoi(getenv("DX_CONV_WIDE_MI")) : 0;
    const int wide_mi = forced_wide ? forced_wide : ((long)dx_cdiv(a.N, 256) * B * ztiles >= 1024 ? 4 : 2);
    const int mi = ztiles == 1 ? (taps == 3 ? narrow_mi : 1) : ((taps == 3 && a.Cin >= 512 && sizeof(TC) == 2) ? wide_mi : 2);
    const long ptiles = (long)dx_cdiv(a.N, 64 * mi) * B;
    dim3 grid((unsigned)(((ptiles + 7) / 8) * 8 * ztiles)), block(NTHREADS);
    if constexpr (sizeof(TC) == 2) {
      if (mi == 4) {
        // (the loader-wave ring at this tile shape measured 918 vs 942 TFLOP/s: the activation stream comes from HBM / Infinity
        // Cache at ~10 B/clk/CU, the pipeline is not the limit)
        hipLaunchKernelGGL((conv_gemm_kernel<TA, TC, TO, TG, 3, 4, 32>), grid, block, 0, s, a);
        DX_LAUNCH_CHECK();
        return DX_OK;
      }
    }
    if (taps == 1 && mi == 1) hipLaunchKernelGGL((conv_gemm_kernel<TA, TC, TO, TG, 1, 1, 32>), grid, block, 0, s, a);
    else if (taps == 1) hipLaunchKernelGGL((conv_gemm_kernel<TA, TC, TO, TG, 1, 2, 32>), grid, block, 0, s, a);
    else if (mi == 1) hipLaunchKernelGGL((conv_gemm_kernel<TA, TC, TO, TG, 3, 1, 32>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((conv_gemm_kernel<TA, TC, TO, TG, 3, 2, 32>), grid, block, 0, s, a);
    DX_LAUNCH_CHECK();
    return DX_OK;
  }
}


}  // namespace

static int conv1d_impl(const void* x, int x_dtype, long ldx, const void* w_packed, int w_dtype, const float* bias,
                       void* y, int y_dtype, long ldy, const void* relu_gate, int gate_dtype,
                       const int64_t* mask_lengths, const int64_t* skip_lengths, int B, int N, int Cin, int Cout,
                       int taps, int flags, const void* w_frag, void* stream);

extern "C" int dx_conv1d(const void* x, int x_dtype, long ldx, const void* w_packed, int w_dtype, const float* bias,
                         void* y, int y_dtype, long ldy, const void* relu_gate, int gate_dtype,
                         const int64_t* mask_lengths, const int64_t* skip_lengths, int B, int N, int Cin, int Cout,
                         int taps, int flags, void* stream) {
  return conv1d_impl(x, x_dtype, ldx, w_packed, w_dtype, bias, y, y_dtype, ldy, relu_gate, gate_dtype, mask_lengths, skip_lengths, B, N,
                     Cin, Cout, taps, flags, nullptr, stream);
}

extern "C" int dx_conv1d_wfrag(const void* x, int x_dtype, long ldx, const void* w_packed, int w_dtype, const void* w_frag,
                               const float* bias, void* y, int y_dtype, long ldy, const void* relu_gate, int gate_dtype,
                               const int64_t* mask_lengths, const int64_t* skip_lengths, int B, int N, int Cin, int Cout,
                               int taps, int flags, void* stream) {
  DX_REQUIRE(!w_frag || (taps == 3 && w_dtype == DX_BF16 && Cin % 32 == 0 && Cout % 32 == 0), DX_ERR_ARG,
             "dx_conv1d_wfrag: a fragment-order copy goes with bf16 weights, taps = 3, Cin %% 32 == 0 and Cout %% 32 == 0");
  return conv1d_impl(x, x_dtype, ldx, w_packed, w_dtype, bias, y, y_dtype, ldy, relu_gate, gate_dtype, mask_lengths, skip_lengths, B, N,
                     Cin, Cout, taps, flags, w_frag, stream);
}

static int conv1d_impl(const void* x, int x_dtype, long ldx, const void* w_packed, int w_dtype, const float* bias,
                       void* y, int y_dtype, long ldy, const void* relu_gate, int gate_dtype,
                       const int64_t* mask_lengths, const int64_t* skip_lengths, int B, int N, int Cin, int Cout,
                       int taps, int flags, const void* w_frag, void* stream) {
  DX_REQUIRE(x && w_packed && y, DX_ERR_ARG, "dx_conv1d: null pointer");
  DX_REQUIRE(B > 0 && N > 0 && Cin > 0 && Cout > 0, DX_ERR_SHAPE, "dx_conv1d: empty shape B=%d N=%d Cin=%d Cout=%d", B, N, Cin, Cout);
  DX_REQUIRE(Cin % 8 == 0 && ldx % 8 == 0, DX_ERR_SHAPE, "dx_conv1d: Cin (%d) and ldx (%ld) must be multiples of 8", Cin, ldx);
  DX_REQUIRE(taps == 1 || taps == 3, DX_ERR_UNSUPPORTED, "dx_conv1d: taps=%d (only 1 and 3)", taps);
  ConvArgs a{x, ldx, w_packed, bias, y, ldy, relu_gate, mask_lengths, skip_lengths, N, Cin, Cout, flags, B, LNEpi{}};
  a.w_frag = w_frag;
  hipStream_t s = (hipStream_t)stream;
  const int gd = relu_gate ? gate_dtype : y_dtype;
  if (w_dtype == DX_BF16) {
    if (x_dtype == DX_F32 && y_dtype == DX_F32 && gd == DX_F32) return launch_taps<float, bf16_t, float, float>(a, B, taps, s);
    if (x_dtype == DX_F32 && y_dtype == DX_F32 && gd == DX_BF16) return launch_taps<float, bf16_t, float, bf16_t>(a, B, taps, s);
    if (x_dtype == DX_F32 && y_dtype == DX_BF16 && gd == DX_BF16) return launch_taps<float, bf16_t, bf16_t, bf16_t>(a, B, taps, s);
    if (x_dtype == DX_BF16 && y_dtype == DX_F32 && gd == DX_F32) return launch_taps<bf16_t, bf16_t, float, float>(a, B, taps, s);
    if (x_dtype == DX_BF16 && y_dtype == DX_BF16 && gd == DX_BF16) return launch_taps<bf16_t, bf16_t, bf16_t, bf16_t>(a, B, taps, s);
  } else if (w_dtype == DX_F32) {
    if (x_dtype == DX_F32 && y_dtype == DX_F32 && gd == DX_F32) return launch_taps<float, float, float, float>(a, B, taps, s);
  }
  dx_set_error("dx_conv1d: unsupported dtype combination x=%d w=%d y=%d gate=%d", x_dtype, w_dtype, y_dtype, gd);
  return DX_ERR_DTYPE;
}

extern "C" int dx_conv1d_relu_bits(const void* x, long ldx, const void* w_packed, const void* w_frag, const float* bias, void* y, long ldy,
                                   uint32_t* bits_out, const uint32_t* bits_in, const int64_t* mask_lengths, const int64_t* skip_lengths,
                                   int B, int N, int Cout, void* stream) {
  DX_REQUIRE(x && w_packed && y, DX_ERR_ARG, "dx_conv1d_relu_bits: null pointer");
  DX_REQUIRE((bits_out != nullptr) != (bits_in != nullptr), DX_ERR_ARG, "dx_conv1d_relu_bits: exactly one of bits_out (ReLU forward) / bits_in (gated data gradient)");
  DX_REQUIRE(B > 0 && N > 0 && Cout > 0 && Cout % WR_BN == 0 && ldx % 8 == 0 && ldy % 8 == 0, DX_ERR_SHAPE,
             "dx_conv1d_relu_bits: Cout %% 256 == 0 and row strides multiples of 8 (got B=%d N=%d Cout=%d ldx=%ld ldy=%ld)", B, N, Cout, ldx, ldy);
  ConvArgs a{x, ldx, w_packed, bits_out ? bias : nullptr, y, ldy, nullptr, mask_lengths, skip_lengths, N, 128, Cout, bits_out ? DX_CONV_RELU : 0, B, LNEpi{}};
  a.w_frag = w_frag;
  a.relu_bits = bits_out;
  a.gate_bits = bits_in;
  if (!try_weight_stationary<bf16_t, bf16_t, bf16_t, bf16_t>(a, B, 3, (hipStream_t)stream)) {
    dx_set_error("dx_conv1d_relu_bits: shape not taken by the register-weights kernel (N * ld beyond 32-bit buffer offsets?)");
    return DX_ERR_UNSUPPORTED;
  }
  DX_LAUNCH_CHECK();
  return DX_OK;
}

// ---- balanced position tiles (dx_conv_tile_plan) ----------------------------------------------------------------
// The k = 3, 1024 -> 128 GEMMs are bound by what a CU can fetch from L2, and a workgroup fetches the whole 786 KB weight
// slice whatever the height of its tile: the cost of a launch is (weight passes per CU) x 12 us.  Fixed 128-row tiles
// give a ragged batch a few tiles more than 256 (a second pass on a handful of CUs doubles the kernel); the plan cuts
// every utterance into equal pieces of at most DX_PLAN_ROWS rows such that the batch is exactly n_tiles (a multiple of
// the 256 CUs) pieces and the tallest piece is as short as possible.
constexpr int DX_PLAN_ROWS = 256, DX_NUM_CU = 256;
__device__ __forceinline__ void conv_plan_body(const int64_t* __restrict__ lens, int B, int N, int T, int4* __restrict__ table, int halo) {
  __shared__ int first[4096 + 1];
  const int lane = threadIdx.x;
  auto len_of = [&](int b) { const int l0 = (int)lens[b], l = (l0 < 0 ? 0 : l0) + halo; return l > N ? N : l; };   // rows that carry work
  auto tiles_at = [&](int H) {
    int c = 0;
    for (int b = lane; b < B; b += 64) c += (len_of(b) + H - 1) / H;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
    return c;
  };
  int lo = 1, hi = DX_PLAN_ROWS;                       // smallest height whose tile count fits (T >= B * ceil(N / 256) by contract)
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (tiles_at(mid) <= T) hi = mid; else lo = mid + 1;
  }
  const int H = lo;
  if (lane == 0) {
    int acc = 0;
    for (int b = 0; b < B; ++b) { first[b] = acc; acc += (len_of(b) + H - 1) / H; }
    first[B] = acc;
  }
  __syncthreads();
  // padding rows of the batch, split evenly over the T workgroups (entry.w)
  long dead = 0;
  for (int b = lane; b < B; b += 64) dead += N - len_of(b);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) dead += __shfl_xor(dead, o, 64);
  const int per = (int)((dead + T - 1) / T);
  for (int b = lane; b < B; b += 64) {
    const int l = len_of(b), t = first[b + 1] - first[b];
    if (t == 0) continue;
    const int hb = (l + t - 1) / t;                    // equal pieces inside the utterance
    for (int j = 0; j < t; ++j) {
      const int n0 = j * hb, rows = l - n0 < hb ? l - n0 : hb;
      if (first[b] + j < T) table[first[b] + j] = make_int4(b, n0, rows > 0 ? rows : 0, per);
    }
  }
  for (int i = first[B] + lane; i < T; i += 64) table[i] = make_int4(0, 0, 0, per);
}

__global__ __launch_bounds__(64) void conv_plan_kernel(const int64_t* __restrict__ lens, int B, int N, int T, int4* __restrict__ table, int halo) {
  conv_plan_body(lens, B, N, T, table, halo);
}
// Everything the step derives from one lengths tensor, in one launch of three single-wave workgroups: the halo-0 plan of the
// LayerNorm-fused GEMMs, the halo-2 plan of the wide GEMMs, and the longest-first launch order of the attention kernels
// (dx_length_order) -- three 5-9 us launches per lengths tensor otherwise, each a dispatch boundary on the launch stream.
__global__ __launch_bounds__(64) void batch_prep_kernel(const int64_t* __restrict__ lens, int B, int N, int T0, int4* __restrict__ table0,
                                                        int T2, int4* __restrict__ table2, int* __restrict__ order) {
  if (blockIdx.x == 0) { if (table0) conv_plan_body(lens, B, N, T0, table0, 0); }
  else if (blockIdx.x == 1) { if (table2) conv_plan_body(lens, B, N, T2, table2, 2); }
  else if (order) {
    for (int i = threadIdx.x; i < B; i += blockDim.x) {    // rank by (length descending, index ascending), as dx_length_order
      const int64_t li = lens[i];
      int rank = 0;
      for (int j = 0; j < B; ++j) { const int64_t lj = lens[j]; rank += (lj > li) || (lj == li && j < i); }
      order[rank] = i;
    }
  }
}

extern "C" int dx_conv_tile_plan_size(int B, int N) {
  if (B <= 0 || N <= 0) return 0;
  const long worst = (long)B * dx_cdiv(N, DX_PLAN_ROWS);
  return (int)((worst + DX_NUM_CU - 1) / DX_NUM_CU * DX_NUM_CU);
}

extern "C" int dx_conv_tile_plan(const int64_t* lengths, int B, int N, int n_tiles, int* table, int halo, void* stream) {
  DX_REQUIRE(lengths && table, DX_ERR_ARG, "dx_conv_tile_plan: null pointer");
  DX_REQUIRE(B > 0 && B <= 4096 && N > 0 && halo >= 0 && halo <= 8, DX_ERR_SHAPE, "dx_conv_tile_plan: B=%d (1..4096), N=%d, halo=%d (0..8)", B, N, halo);
  DX_REQUIRE(n_tiles >= B * dx_cdiv(N, DX_PLAN_ROWS), DX_ERR_ARG, "dx_conv_tile_plan: n_tiles=%d < B * ceil(N / 256) = %d", n_tiles,
             B * dx_cdiv(N, DX_PLAN_ROWS));
  hipLaunchKernelGGL(conv_plan_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, lengths, B, N, n_tiles, reinterpret_cast<int4*>(table), halo);
  DX_LAUNCH_CHECK();
  return DX_OK;
}

extern "C" int dx_batch_prep(const int64_t* lengths, int B, int N, int n_tiles0, int* table0, int n_tiles2, int* table2, int* order,
                             void* stream) {
  DX_REQUIRE(lengths, DX_ERR_ARG, "dx_batch_prep: null pointer");
  DX_REQUIRE(B > 0 && B <= 4096 && N > 0, DX_ERR_SHAPE, "dx_batch_prep: B=%d (1..4096), N=%d", B, N);
  const int need = B * dx_cdiv(N, DX_PLAN_ROWS);
  DX_REQUIRE((!table0 || n_tiles0 >= need) && (!table2 || n_tiles2 >= need), DX_ERR_ARG, "dx_batch_prep: a tile count below B * ceil(N / 256) = %d", need);
  hipLaunchKernelGGL(batch_prep_kernel, dim3(3), dim3(64), 0, (hipStream_t)stream, lengths, B, N, n_tiles0, reinterpret_cast<int4*>(table0),
                     n_tiles2, reinterpret_cast<int4*>(table2), order);
  DX_LAUNCH_CHECK();
  return DX_OK;
}

// plan validity for the LayerNorm-fused GEMMs: bf16 operands, k = 3, whole 32-channel chunks
static int plan_check(const char* who, const int* plan, int plan_tiles, const int64_t* lengths, int x_dtype, int w_dtype, long ldx, int Cin,
                      int taps, int B, int N, bool lnbwd = false) {
  if (!plan) return DX_OK;
  DX_REQUIRE(lengths, DX_ERR_ARG, "%s: a tile plan needs lengths", who);
  DX_REQUIRE(x_dtype == DX_BF16 && w_dtype == DX_BF16 && (taps == 3 || (taps == 1 && lnbwd)) && Cin % 32 == 0 && Cin <= DX_ZERO_PAGE_EL && ldx % 8 == 0, DX_ERR_UNSUPPORTED,
             "%s: tile plans are for bf16 operands, taps = 3 (or 1 for the backward variant), Cin %% 32 == 0 (got x=%d w=%d taps=%d Cin=%d)", who, x_dtype, w_dtype, taps, Cin);
  // any tile count the plan kernel accepts (callers may trade tile height against tile count for small batches)
  DX_REQUIRE(plan_tiles >= B * dx_cdiv(N, DX_PLAN_ROWS), DX_ERR_ARG, "%s: plan_tiles=%d < B * ceil(N / 256) = %d", who, plan_tiles,
             B * dx_cdiv(N, DX_PLAN_ROWS));
  return DX_OK;
}

extern "C" int dx_conv1d_ln(const void* x, int x_dtype, long ldx, const void* w_packed, int w_dtype, const float* bias,
                            const float* residual, const float* gamma, const float* beta, const float* film, long ldf,
                            const int64_t* lengths, float* y, void* y_lp, float* s_out, float* mean, float* rstd, int B, int N,
                            int Cin, int taps, float p_pre, uint64_t seed_pre, const int* plan, int plan_tiles, const void* w_frag,
                            const void* w2_packed, const float* b2, void* y2, int n2, const DxStepScalars* step, void* stream) {
  return dx_conv1d_ln_vres(x, x_dtype, ldx, w_packed, w_dtype, bias, residual, nullptr, nullptr, nullptr, nullptr, gamma, beta, film, ldf, lengths,
                           y, y_lp, s_out, mean, rstd, B, N, Cin, taps, p_pre, seed_pre, plan, plan_tiles, w_frag, w2_packed, b2, y2, n2, step, stream);
}

extern "C" int dx_conv1d_ln_vres(const void* x, int x_dtype, long ldx, const void* w_packed, int w_dtype, const float* bias,
                                 const float* residual, const float* res_mean, const float* res_rstd, const float* res_gamma, const float* res_beta,
                                 const float* gamma, const float* beta, const float* film, long ldf,
                                 const int64_t* lengths, float* y, void* y_lp, float* s_out, float* mean, float* rstd, int B, int N,
                                 int Cin, int taps, float p_pre, uint64_t seed_pre, const int* plan, int plan_tiles, const void* w_frag,
                                 const void* w2_packed, const float* b2, void* y2, int n2, const DxStepScalars* step, void* stream) {
  DX_REQUIRE(x && w_packed && residual && gamma && beta && (y || y_lp), DX_ERR_ARG, "dx_conv1d_ln: null pointer");
  DX_REQUIRE(y || (x_dtype == DX_BF16 && w_dtype == DX_BF16), DX_ERR_ARG, "dx_conv1d_ln: y = NULL (bf16 copy only) goes with bf16 operands");
  if (res_mean) {
    DX_REQUIRE(res_rstd && res_gamma && res_beta && lengths, DX_ERR_ARG, "dx_conv1d_ln_vres: res_mean / res_rstd / res_gamma / res_beta / lengths come together");
    DX_REQUIRE(plan && w_frag && taps == 3 && Cin >= 256 && Cin % 128 == 0 && (long)B * N <= 256L * 256 && w_dtype == DX_BF16 && x_dtype == DX_BF16,
               DX_ERR_UNSUPPORTED, "dx_conv1d_ln_vres: the re-derived residual exists on the split-K path only (bf16, taps = 3, plan + fragment-order "
               "weights, Cin %% 128 == 0, B * N <= 65536)");
  }
  DX_REQUIRE(!w_frag || plan, DX_ERR_ARG, "dx_conv1d_ln: fragment-order weights go with a tile plan");
  if (int rc = plan_check("dx_conv1d_ln", plan, plan_tiles, lengths, x_dtype, w_dtype, ldx, Cin, taps, B, N)) return rc;
  DX_REQUIRE(B > 0 && N > 0 && Cin > 0, DX_ERR_SHAPE, "dx_conv1d_ln: empty shape");
  DX_REQUIRE(Cin % 8 == 0 && ldx % 8 == 0, DX_ERR_SHAPE, "dx_conv1d_ln: Cin (%d) and ldx (%ld) must be multiples of 8", Cin, ldx);
  DX_REQUIRE(taps == 1 || taps == 3, DX_ERR_UNSUPPORTED, "dx_conv1d_ln: taps=%d (only 1 and 3)", taps);
  DX_REQUIRE((mean == nullptr) == (rstd == nullptr), DX_ERR_ARG, "dx_conv1d_ln: mean and rstd come together");
  DX_REQUIRE(p_pre >= 0.f && p_pre < 1.f, DX_ERR_ARG, "dx_conv1d_ln: dropout p out of [0,1)");
  ConvArgs a{x, ldx, w_packed, bias, nullptr, BN, nullptr, lengths, lengths, N, Cin, BN, 0, B,
             LNEpi{gamma, beta, residual, film, ldf, y, y_lp, s_out, mean, rstd, p_pre, seed_pre, 1}};
  a.plan = plan; a.plan_tiles = plan_tiles; a.w_frag = w_frag; a.ln.step = step;
  a.ln.res_mean = res_mean; a.ln.res_rstd = res_rstd; a.ln.res_gamma = res_gamma; a.ln.res_beta = res_beta;
  if (y2) {   // second GEMM in the epilogue (the next block's QKV projection): only the split-K workgroups carry it (gate of launch_taps)
    DX_REQUIRE(w2_packed && (n2 == 128 || n2 == 384) && plan && w_frag && taps == 3 && Cin >= 256 && Cin % 128 == 0 && (long)B * N <= 256L * 256 &&
               w_dtype == DX_BF16 && x_dtype == DX_BF16, DX_ERR_UNSUPPORTED, "dx_conv1d_ln: y2 needs n2 in {128, 384} and the split-K path (bf16, "
               "taps = 3, plan + fragment-order weights, Cin %% 128 == 0, B * N <= 65536)");
    a.ln.w2 = w2_packed; a.ln.y2 = y2; a.ln.b2 = b2; a.ln.n2 = n2;
  }
  hipStream_t s = (hipStream_t)stream;
  if (w_dtype == DX_BF16 && x_dtype == DX_BF16) return launch_taps<bf16_t, bf16_t, float, float, 1>(a, B, taps, s);
  if (w_dtype == DX_BF16 && x_dtype == DX_F32) return launch_taps<float, bf16_t, float, float, 1>(a, B, taps, s);
  if (w_dtype == DX_F32 && x_dtype == DX_F32) return launch_taps<float, float, float, float, 1>(a, B, taps, s);
  dx_set_error("dx_conv1d_ln: unsupported dtype combination x=%d w=%d", x_dtype, w_dtype);
  return DX_ERR_DTYPE;
}

extern "C" int dx_conv1d_lnbwd(const void* x, int x_dtype, long ldx, const void* w_packed, int w_dtype, float* y_inout,
                               const float* s_in, const float* mean, const float* rstd, const float* gamma,
                               const float* beta, const float* film, long ldf, const int64_t* lengths, void* dx_pre_lp,
                               float* dgamma, float* dbeta, float* dfilm, long lddf, int B, int N, int Cin, int taps,
                               float p_pre, uint64_t seed_pre, const int* plan, int plan_tiles, const void* w_frag, const void* w2_packed,
                               void* y2, const DxStepScalars* step, void* stream) {
  if (int rc = plan_check("dx_conv1d_lnbwd", plan, plan_tiles, lengths, x_dtype, w_dtype, ldx, Cin, taps, B, N, true)) return rc;
  DX_REQUIRE(!w_frag || plan, DX_ERR_ARG, "dx_conv1d_lnbwd: fragment-order weights go with a tile plan");
  DX_REQUIRE(x && w_packed && y_inout && s_in && mean && rstd && gamma && beta && lengths && dx_pre_lp && dgamma && dbeta,
             DX_ERR_ARG, "dx_conv1d_lnbwd: null pointer");
  DX_REQUIRE((film == nullptr) == (dfilm == nullptr), DX_ERR_ARG, "dx_conv1d_lnbwd: film and dfilm come together");
  DX_REQUIRE(B > 0 && N > 0 && Cin > 0, DX_ERR_SHAPE, "dx_conv1d_lnbwd: empty shape");
  DX_REQUIRE(Cin % 8 == 0 && ldx % 8 == 0, DX_ERR_SHAPE, "dx_conv1d_lnbwd: Cin (%d) and ldx (%ld) must be multiples of 8", Cin, ldx);
  DX_REQUIRE(taps == 1 || taps == 3, DX_ERR_UNSUPPORTED, "dx_conv1d_lnbwd: taps=%d (only 1 and 3)", taps);
  DX_REQUIRE(p_pre >= 0.f && p_pre < 1.f, DX_ERR_ARG, "dx_conv1d_lnbwd: dropout p out of [0,1)");
  ConvArgs a{x, ldx, w_packed, nullptr, y_inout, BN, nullptr, lengths, lengths, N, Cin, BN, 0, B,
             LNEpi{gamma, beta, nullptr, film, ldf, y_inout, dx_pre_lp, const_cast<float*>(s_in), const_cast<float*>(mean),
                   const_cast<float*>(rstd), p_pre, seed_pre, 2, dgamma, dbeta, dfilm, lddf}};
  a.plan = plan; a.plan_tiles = plan_tiles; a.w_frag = w_frag; a.ln.step = step;
  if (y2) {   // second GEMM in the epilogue: only the split-K workgroups carry it (same gate as launch_taps)
    DX_REQUIRE(w2_packed && plan && w_frag && taps == 3 && Cin >= 256 && Cin % 128 == 0 && (long)B * N <= 256L * 256 && w_dtype == DX_BF16 &&
               x_dtype == DX_BF16, DX_ERR_UNSUPPORTED, "dx_conv1d_lnbwd: y2 needs the split-K path (bf16, taps = 3, plan + fragment-order weights, "
               "Cin %% 128 == 0, B * N <= 65536)");
    a.ln.w2 = w2_packed; a.ln.y2 = y2; a.ln.b2 = nullptr; a.ln.n2 = BN;
  }
  hipStream_t s = (hipStream_t)stream;
  if (w_dtype == DX_BF16 && x_dtype == DX_BF16) return launch_taps<bf16_t, bf16_t, float, float, 2>(a, B, taps, s);
  if (w_dtype == DX_BF16 && x_dtype == DX_F32) return launch_taps<float, bf16_t, float, float, 2>(a, B, taps, s);
  if (w_dtype == DX_F32 && x_dtype == DX_F32) return launch_taps<float, float, float, float, 2>(a, B, taps, s);
  dx_set_error("dx_conv1d_lnbwd: unsupported dtype combination x=%d w=%d", x_dtype, w_dtype);
  return DX_ERR_DTYPE;
}

extern "C" int dx_conv1d_wide(const void* x, long ldx, const void* w_frag, const float* bias, void* y, long ldy, const int64_t* lengths,
                              const int* plan, int plan_tiles, int halo, int B, int N, int Cin, int Cout, int flags, void* stream) {
  DX_REQUIRE(x && w_frag && y && lengths && plan, DX_ERR_ARG, "dx_conv1d_wide: null pointer");
  DX_REQUIRE(B > 0 && N > 0, DX_ERR_SHAPE, "dx_conv1d_wide: empty shape");
  DX_REQUIRE(Cin % 128 == 0 && Cin >= 256 && Cin <= DX_ZERO_PAGE_EL && Cout % 256 == 0 && ldx % 8 == 0 && ldy % 8 == 0, DX_ERR_UNSUPPORTED,
             "dx_conv1d_wide: Cin %% 128 == 0 (256..4096), Cout %% 256 == 0, row strides multiples of 8 (got Cin=%d Cout=%d)", Cin, Cout);
  DX_REQUIRE((flags & ~DX_CONV_RELU) == 0 && halo >= 0 && halo <= 8, DX_ERR_UNSUPPORTED, "dx_conv1d_wide: only DX_CONV_RELU is supported (flags=%d), halo 0..8", flags);
  DX_REQUIRE(plan_tiles >= B * dx_cdiv(N, DX_PLAN_ROWS), DX_ERR_ARG, "dx_conv1d_wide: plan_tiles=%d < B * ceil(N / 256)", plan_tiles);
  ConvArgs a{x, ldx, nullptr, bias, y, ldy, nullptr, nullptr, lengths, N, Cin, Cout, flags | (halo << 8), B, LNEpi{}};
  a.w_frag = w_frag; a.plan = plan; a.plan_tiles = plan_tiles;
  dim3 grid((unsigned)(plan_tiles * (Cout / 256)));
  hipLaunchKernelGGL(conv_wide_kernel, grid, dim3(WD_THREADS), 0, (hipStream_t)stream, a);
  DX_LAUNCH_CHECK();
  return DX_OK;
}
