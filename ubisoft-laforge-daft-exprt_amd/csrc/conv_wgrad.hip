// Weight / bias gradients of the conv GEMMs (conv_gemm.hip): conv_wgrad_kernel (register-staged, any operand type),
// conv_wgrad_ring_kernel / conv_wgrad_ring_pair_kernel (bf16 operands, LDS-DMA ring with loader waves), the partial-tile reductions
// and the dx_conv1d_wgrad* entry points.
#include <stdlib.h>

#include <type_traits>

#include "dx_common.h"

namespace {

#include "conv_common.h"

#ifndef DX_WGRAD_WPS
#define DX_WGRAD_WPS 2
#endif

// ---- weight gradient ------------------------------------------------------------------------
//   dW[co][ci][tap] += sum_{b, n} dY[b, n, co] * X[b, n + tap - taps/2, ci]        (fp32, PyTorch layout)
//   db[co]          += sum_{b, n} dY[b, n, co]
// GEMM with the positions as the contraction axis.  These kernels are bound by streaming their operands: every output
// tile re-reads the rows of dY and X it contracts, so the tile is as large as the accumulators allow -- a 512-thread
// workgroup owns 128 (co) x 128 (ci) x taps outputs (wave (wm, wn) of 2 x 4 holds 64 co x 32 ci x taps = 2*taps MFMA
// 32x32 tiles), which for the 128 <-> 1024 convolutions reads the wide operand ONCE and the narrow one 8 times (the
// first version's 128 x 64 tiles: 2 and 16 times).  The flat list of VALID 64-row items of the batch is split evenly
// over the workgroups of a tile (split-K balanced over valid rows, not utterances: lengths differ by 10x inside a batch).
// Per item the dY tile [64][128] and the haloed X tile [66][128] are staged in LDS in their natural row-major layout
// (fetched into registers one item ahead); both MFMA operands need "8 consecutive positions for one channel", which the
// LDS transpose read (ds_read_b64_tr_b16, gather8) delivers without a software transpose; all taps reuse the same X
// tile at a row offset.  The partial tile goes to a workspace in register order (coalesced) and wgrad_reduce_kernel
// sums the partials of all splits into dW -- a fixed summation order, no fp32 atomics on dW (measured: the atomics
// were 1/3 of the kernel).  Without a workspace the partial tile is added with atomics.
constexpr int WG_CO = 128, WG_CI = 128, WG_P = 64, WG_THREADS = 512;

struct WgradArgs {
  const void* dy; long lddy; const void* x; long ldx;
  float* dw; float* db; const int64_t* lengths; float* ws;
  int B, N, Cin, Cout, nsplit, tiles_ci;
};

template <typename TC, int TAPS> struct WgradSmem {
  static constexpr int XROWS = WG_P + TAPS - 1;
  // row strides = 16 banks (mod 64) apart: the 4 rows x 2 halves x 4 chunks touched by one 32-lane group of a
  // transpose read (ds_read_b64_tr_b16) then fall on 64 distinct banks
  static constexpr int LDA = WG_CO + 4 * Pad<TC>::value, LDB = WG_CI + 4 * Pad<TC>::value;
  static constexpr int A_ELEMS = WG_P * LDA, B_ELEMS = XROWS * LDB;
  static constexpr int TILE_FLOATS = TAPS * 2 * 16 * WG_THREADS;   // one partial tile in register order
};

template <typename TA, typename TB, typename TC, int TAPS>
__global__ __launch_bounds__(WG_THREADS, DX_WGRAD_WPS) void conv_wgrad_kernel(WgradArgs p) {
  typedef WgradSmem<TC, TAPS> SM;
  constexpr int HALO = TAPS / 2, XROWS = SM::XROWS, LDA = SM::LDA, LDB = SM::LDB;
  constexpr int A_PT = WG_P * (WG_CO / 8) / WG_THREADS;                                 // 2
  constexpr int B_CH = XROWS * (WG_CI / 8), B_PT = (B_CH + WG_THREADS - 1) / WG_THREADS;
  typedef typename Vec8<TC>::type frag_t;
  typedef typename VecN<TA, 8>::type rawa_t;
  typedef typename VecN<TB, 8>::type rawb_t;
  __shared__ __attribute__((aligned(16))) TC dYs[SM::A_ELEMS];
  __shared__ __attribute__((aligned(16))) TC Xs[SM::B_ELEMS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, g = lane >> 5;
  const int wm = wave >> 2, wn = wave & 3;
  const int ntiles = dx_cdiv(p.Cout, WG_CO) * p.tiles_ci;
  const bool by_split = ntiles > 1 && (p.nsplit & 7) == 0;      // the tiles of a split on one XCD (see wgrad_ring_body)
  const int split = by_split ? (int)blockIdx.x % p.nsplit : (int)blockIdx.x / ntiles, tile = by_split ? (int)blockIdx.x / p.nsplit : (int)blockIdx.x % ntiles;
  const int co0 = (tile / p.tiles_ci) * WG_CO, ci0 = (tile % p.tiles_ci) * WG_CI;
  const int N = p.N, Cin = p.Cin, Cout = p.Cout;

  // rows beyond len + halo carry exactly-zero gradients (masked upstream): not part of the item list
  __shared__ int s_nl[DX_SCAN_MAXB + 1], s_cum[DX_SCAN_MAXB + 1], s_part[WG_THREADS / 64];
  const bool scan = p.B <= DX_SCAN_MAXB;               // (else: the serial walk over the lengths)
  auto nlim_g = [&](int b) { return p.lengths ? min(N, (int)p.lengths[b] + 2) : N; };
  auto nlim_of = [&](int b) { return scan ? s_nl[b] : nlim_g(b); };
  int total = 0, b = 0, n0 = 0, nlim = 0, i0, i1;
  if (scan) {
    dx_block_count_scan<WG_THREADS>(p.B, nlim_g, [](int nl) { return dx_cdiv(nl, WG_P); }, s_nl, s_cum, s_part);
    total = s_cum[p.B];
    i0 = (int)((long)total * split / p.nsplit); i1 = (int)((long)total * (split + 1) / p.nsplit);
    b = dx_locate_item(s_cum, p.B, i0);
    nlim = s_nl[b];
    n0 = (i0 - s_cum[b]) * WG_P;
  } else {
    for (int bb = 0; bb < p.B; ++bb) total += dx_cdiv(nlim_g(bb), WG_P);
    i0 = (int)((long)total * split / p.nsplit); i1 = (int)((long)total * (split + 1) / p.nsplit);
    for (int cum = 0; b < p.B; ++b) {                 // locate item i0
      nlim = nlim_g(b);
      const int c = dx_cdiv(nlim, WG_P);
      if (i0 < cum + c) { n0 = (i0 - cum) * WG_P; break; }
      cum += c;
    }
  }
  int left = i1 - i0;

  f32x16 acc[TAPS][2];
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][i][r] = 0.f;
  // bias gradient = dY^T . 1: one extra MFMA per k-step against an all-ones B fragment (published only by the
  // ci0 == 0 tiles' wn == 0 waves) instead of a serial LDS column-sum loop
  const bool do_bias = p.db && ci0 == 0 && wn == 0;
  f32x16 bacc[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) bacc[i][r] = 0.f;
  frag_t ones;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones[e] = (TC)1.f;

  rawa_t ra[A_PT];
  rawb_t rb[B_PT];
  auto fetch = [&](int fb, int fn0, int flim) {
    const TA* dY = reinterpret_cast<const TA*>(p.dy) + (size_t)fb * N * p.lddy;
    const TB* X = reinterpret_cast<const TB*>(p.x) + (size_t)fb * N * p.ldx;
#pragma unroll
    for (int t = 0; t < A_PT; ++t) {
      const int c = tid + t * WG_THREADS;
      const int n = fn0 + (c >> 4), co = co0 + (c & 15) * 8;
#pragma unroll
      for (int e = 0; e < 8; ++e) ra[t][e] = (TA)0.f;
      if (n < flim && co < Cout) ra[t] = raw_load8<TA>(dY + (size_t)n * p.lddy + co);   // rows >= flim are zero
    }
#pragma unroll
    for (int t = 0; t < B_PT; ++t) {
      const int c = tid + t * WG_THREADS;
      const int n = fn0 + (c >> 4) - HALO, ci = ci0 + (c & 15) * 8;
#pragma unroll
      for (int e = 0; e < 8; ++e) rb[t][e] = (TB)0.f;
      // x rows > flim only ever meet zero dy rows
      if (c < B_CH && n >= 0 && n < N && n <= flim && ci < Cin) rb[t] = raw_load8<TB>(X + (size_t)n * p.ldx + ci);
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int t = 0; t < A_PT; ++t) {
      const int c = tid + t * WG_THREADS;
      *reinterpret_cast<frag_t*>(&dYs[(c >> 4) * LDA + (c & 15) * 8]) = cvt8<TA, TC>(ra[t]);
    }
#pragma unroll
    for (int t = 0; t < B_PT; ++t) {
      const int c = tid + t * WG_THREADS;
      if (c < B_CH) *reinterpret_cast<frag_t*>(&Xs[(c >> 4) * LDB + (c & 15) * 8]) = cvt8<TB, TC>(rb[t]);
    }
  };

  if (left > 0) {
    fetch(b, n0, nlim);
    commit();
    __syncthreads();
  }
  // two copies of the item loop, with and without the bias MFMAs (a quarter of the matrix work, needed by 1 wave in 4
  // of the ci0 == 0 tiles only); the choice is wave-uniform and made once, outside the loop
  auto items = [&](auto bias_tag) {
    constexpr bool BIAS = decltype(bias_tag)::value;
    while (left > 0) {
      --left;
      if (left > 0) {                                 // next item
        n0 += WG_P;
        if (n0 >= nlim) { ++b; n0 = 0; nlim = nlim_of(b); }
        fetch(b, n0, nlim);
      }
#pragma unroll
      for (int ks = 0; ks < WG_P / 16; ++ks) {
        const int kA = ks * 16 + 8 * g, kB = kA + 4;
        frag_t a[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) a[i] = gather8<TC, 32>(dYs, LDA, kA, kB, wm * 64 + i * 32, lane);
        if (BIAS) {
#pragma unroll
          for (int i = 0; i < 2; ++i) dx_mma(bacc[i], a[i], ones);
        }
#pragma unroll
        for (int t = 0; t < TAPS; ++t) {
          frag_t bx = gather8<TC, 32>(Xs, LDB, kA + t, kB + t, wn * 32, lane);
#pragma unroll
          for (int i = 0; i < 2; ++i) dx_mma(acc[t][i], a[i], bx);
        }
      }
      __syncthreads();
      if (left > 0) {
        commit();
        __syncthreads();
      }
    }
  };
  if (__builtin_amdgcn_readfirstlane((int)do_bias)) items(std::true_type{});
  else items(std::false_type{});
  if (p.ws) {
    float* out = p.ws + ((size_t)split * ntiles + tile) * SM::TILE_FLOATS;
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) out[((t * 2 + i) * 16 + r) * WG_THREADS + tid] = acc[t][i][r];
  } else {
    const int ci = ci0 + wn * 32 + l31;
    if (ci < Cin) {
#pragma unroll
      for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int co = co0 + wm * 64 + i * 32 + dx_acc_row(r, g);
            if (co < Cout) atomicAdd(p.dw + ((size_t)co * Cin + ci) * TAPS + t, acc[t][i][r]);
          }
    }
  }
  if (do_bias && l31 == 0) {   // every column of bacc holds the same row sums; column 0 publishes them
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co0 + wm * 64 + i * 32 + dx_acc_row(r, g);
        if (co < Cout) atomicAdd(p.db + co, bacc[i][r]);
      }
  }
}

// s_waitcnt vmcnt(n) for a wave-uniform run-time n (LDS-DMA rings: "all but my pieces of the younger stages have landed")
__device__ __forceinline__ void dx_wait_vmcnt(int n) {
  switch (n) {
#define DX_VMW(n) case n: asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory"); break;
    DX_VMW(1) DX_VMW(2) DX_VMW(3) DX_VMW(4) DX_VMW(5) DX_VMW(6) DX_VMW(7) DX_VMW(8) DX_VMW(9) DX_VMW(10) DX_VMW(11) DX_VMW(12)
    DX_VMW(13) DX_VMW(14) DX_VMW(15) DX_VMW(16) DX_VMW(17) DX_VMW(18) DX_VMW(19) DX_VMW(20) DX_VMW(21) DX_VMW(22) DX_VMW(23) DX_VMW(24)
    DX_VMW(25) DX_VMW(26) DX_VMW(27) DX_VMW(28) DX_VMW(29) DX_VMW(30) DX_VMW(31) DX_VMW(32) DX_VMW(33) DX_VMW(34) DX_VMW(35) DX_VMW(36)
#undef DX_VMW
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
}

// transpose read from an UNPADDED row-major [rows][128] bf16 tile whose 64-byte segments are XOR-swizzled by row & 3
// (segment s of row r sits at s ^ (r & 3)): the 4 rows x 64 bytes one 32-lane group of ds_read_b64_tr_b16 touches then
// cover all 64 banks, and a 16-row x 64-byte LDS-DMA piece lands as 4 whole rows with the swizzle on the source side.
__device__ __forceinline__ bf16x8 gather8_swz(const bf16_t* tile, int kA, int kB, int col0, int lane) {
  const int i = lane & 15, half = (lane >> 4) & 1, j = i >> 2, q = i & 3;
  const int ra = kA + j, rb = kB + j, seg = col0 >> 5, within = (col0 & 31) + 16 * half + 4 * q;
  const bf16_t* pa = tile + ra * 128 + ((seg ^ (ra & 3)) << 5) + within;
  const bf16_t* pb = tile + rb * 128 + ((seg ^ (rb & 3)) << 5) + within;
  s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(pa));
  s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(pb));
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  s16x8 r = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  return __builtin_bit_cast(bf16x8, r);
}

// ---- weight gradient, bf16 operands: LDS-DMA ring with loader waves (same tiles, item list, partial-tile layout and
// reduce kernel as conv_wgrad_kernel).  768 threads: waves 0-7 own the 128 x 128 x taps accumulators (wave (wm, wn) of
// 2 x 4: 64 co x 32 ci) and only read LDS and issue MFMAs; waves 8-11 stream the items -- dY [64][128] and the haloed X
// [64 + taps - 1][128], unpadded rows with the gather8_swz swizzle -- into a 4-deep ring with global_load_lds_dwordx4
// (1 KiB = 4 rows per piece; rows outside the item's utterance read a zero page), one workgroup barrier per item.  The
// register-staged kernel spent 4800 cycles per item on 2048 cycles of MFMA work (two barriers, ds_write staging, and
// the bias MFMAs in every wave); here the bias gradient is a column sum the loader waves take from the LDS tile.
constexpr int WGR_THREADS = 768, WGR_RING = 4;
template <int TAPS>
__device__ __forceinline__ void wgrad_ring_body(const WgradArgs& p, const int bid) {
  constexpr int HALO = TAPS / 2, XROWS = WG_P + TAPS - 1;
  constexpr int A_PIECES = WG_P / 4, X_PIECES = (XROWS + 3) / 4, NP = A_PIECES + X_PIECES, ITEM_EL = NP * 512, MAXP = (NP + 3) / 4;
  static_assert(MAXP * (WGR_RING - 2) <= 36, "vmcnt switch too short");
  __shared__ __attribute__((aligned(16))) bf16_t ring[WGR_RING * ITEM_EL];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, g = lane >> 5;
  const int ntiles = dx_cdiv(p.Cout, WG_CO) * p.tiles_ci;
  // Workgroup L runs on XCD L % 8, and every XCD has its own L2.  The 8 tiles of one split of a 128 <-> 1024 gradient walk the SAME rows of
  // the narrow operand (dz / a: 128 channels): in (split, tile) index order they sat on 8 different XCDs and the narrow operand was fetched
  // 8 times (~54 MB per frame-level launch, 0.9 GB per step of FETCH_SIZE).  With the split index fastest (nsplit % 8 == 0, see
  // wgrad_nsplit) all tiles of a split share an XCD -- also the 3 tiles of a QKV gradient's split (and, in the pair launch, whatever the
  // first problem's workgroup count is: the offset moves all tiles of a split alike).  Only the assignment of (split, tile) to workgroups
  // changes: same partial tiles, same sums.
  const bool by_split = ntiles > 1 && ntiles != 64 && (p.nsplit & 7) == 0;
  const int split = by_split ? bid % p.nsplit : bid / ntiles;
  int tile = by_split ? bid / p.nsplit : bid % ntiles;
  // XCD-aware tile order for the 8 x 8 tile grid of a 1024 x 1024 weight (workgroup L runs on XCD L % 8, ntiles % 8 == 0): in index
  // order an XCD owns one ci column of tiles and reads ALL of dY (8 x 61 MB per launch over the chip); dealt as 4 (co) x 2 (ci)
  // blocks it reads half of dY and a quarter of X.  Only the assignment of tile ids to workgroups changes.
  if (ntiles == 64 && p.tiles_ci == 8) {
    const int x = tile & 7, k8 = tile >> 3;
    tile = (4 * (x >> 2) + (k8 & 3)) * 8 + 2 * (x & 3) + (k8 >> 2);
  }
  const int co0 = (tile / p.tiles_ci) * WG_CO, ci0 = (tile % p.tiles_ci) * WG_CI;
  const int N = p.N, Cin = p.Cin, Cout = p.Cout;
  __shared__ int s_nl[DX_SCAN_MAXB + 1], s_cum[DX_SCAN_MAXB + 1], s_part[WGR_THREADS / 64];
  const bool scan = p.B <= DX_SCAN_MAXB;               // (else: the serial walk over the lengths)
  auto nlim_g = [&](int b) { return p.lengths ? min(N, (int)p.lengths[b] + 2) : N; };
  auto nlim_of = [&](int b) { return scan ? s_nl[b] : nlim_g(b); };
  int total = 0;
  if (scan) {
    dx_block_count_scan<WGR_THREADS>(p.B, nlim_g, [](int nl) { return dx_cdiv(nl, WG_P); }, s_nl, s_cum, s_part);
    total = s_cum[p.B];
  } else {
    for (int b = 0; b < p.B; ++b) total += dx_cdiv(nlim_g(b), WG_P);
  }
  const int i0 = (int)((long)total * split / p.nsplit), i1 = (int)((long)total * (split + 1) / p.nsplit);
  const int count = i1 - i0;

  if (wave >= 8) {
    // ---- loader waves
    const int lw = __builtin_amdgcn_readfirstlane(wave) - 8;
    const int mine = (NP - lw + 3) >> 2;
    int ib = 0, in0 = 0, ilim = 0;
    if (scan) {
      ib = dx_locate_item(s_cum, p.B, i0);
      ilim = s_nl[ib];
      in0 = (i0 - s_cum[ib]) * WG_P;
    } else {
      for (int cum = 0; ib < p.B; ++ib) {             // locate item i0
        ilim = nlim_g(ib);
        const int c = dx_cdiv(ilim, WG_P);
        if (i0 < cum + c) { in0 = (i0 - cum) * WG_P; break; }
        cum += c;
      }
    }
    // per-piece constants (A_PIECES is a multiple of 4: slot t < A_PIECES / 4 is a dY piece for every loader): the row
    // inside the item and this lane's column; per item only the row number changes -- the loaders are instruction-bound
    // (the first version recomputed everything per piece and needed 4400 cycles per item for 1500 cycles of MFMA work)
    static_assert(A_PIECES % 4 == 0, "dY pieces per loader must not depend on the loader");
    constexpr int TA_SLOTS = A_PIECES / 4;
    const bf16_t* zp = reinterpret_cast<const bf16_t*>(dx_zero_page);
    int roff[MAXP], coff[MAXP];
    bool cok[MAXP];
#pragma unroll
    for (int t = 0; t < MAXP; ++t) {
      const bool isx = t >= TA_SLOTS;
      const int q = lw + 4 * t, r = (isx ? q - A_PIECES : q) * 4 + (lane >> 4), c16 = (lane & 15) ^ ((r & 3) << 2);
      roff[t] = isx ? r - HALO : r;
      coff[t] = (isx ? ci0 : co0) + c16 * 8;
      cok[t] = isx ? (r < XROWS && coff[t] < Cin) : (coff[t] < Cout);
    }
    const uint32_t lddy = (uint32_t)p.lddy, ldx = (uint32_t)p.ldx;
    auto issue_item = [&](int buf) {
      const bf16_t* dY = reinterpret_cast<const bf16_t*>(p.dy) + (size_t)ib * N * p.lddy;
      const bf16_t* X = reinterpret_cast<const bf16_t*>(p.x) + (size_t)ib * N * p.ldx;
      const int xlim = ilim < N - 1 ? ilim : N - 1;          // X rows are valid for 0 <= n <= min(ilim, N - 1)
#pragma unroll
      for (int t = 0; t < MAXP; ++t) {
        if (lw + 4 * t < NP) {
          const bool isx = t >= TA_SLOTS;
          const int n = in0 + roff[t];
          const bool ok = cok[t] && (isx ? (unsigned)n <= (unsigned)xlim : n < ilim);
          const uint32_t off = __umul24((uint32_t)n, isx ? ldx : lddy) + (uint32_t)coff[t];
          const bf16_t* sp = ok ? (isx ? X : dY) + off : zp + (coff[t] & 127);
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)sp,
                                           (__attribute__((address_space(3))) void*)(ring + buf * ITEM_EL + (lw + 4 * t) * 512), 16, 0, 0);
        }
      }
      in0 += WG_P;                                     // advance to the next item of the list
      if (in0 >= ilim) { ++ib; in0 = 0; ilim = ib < p.B ? nlim_of(ib) : 0; }
    };
#pragma unroll
    for (int st = 0; st < WGR_RING - 1; ++st)
      if (st < count) issue_item(st);
    // Barrier j (j = 0 .. count) promises the MFMA waves that items <= j + 1 have landed and that item j - 1's slot is free: one item
    // of slack, so that they can read the NEXT item's first fragments during the last k-step of the current one (no LDS round trip
    // and no barrier wait in the open at every item boundary: that was ~40 % of their loop).  Two items in flight instead of three.
    // (The bias gradient -- column sums of the dY tile -- is taken by the MFMA waves from the fragments they hold.)
    if (count > 2) dx_wait_vmcnt(mine); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    int nbuf = WGR_RING - 1;
    for (int k = 0; k < count; ++k) {
      const bool more = k + WGR_RING - 1 < count;
      if (more) issue_item(nbuf);
      if (more) dx_wait_vmcnt(mine); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      nbuf = nbuf + 1 == WGR_RING ? 0 : nbuf + 1;
    }
    return;
  }

  // ---- MFMA waves
  const int wm = wave >> 2, wn = wave & 3;
  f32x16 acc[TAPS][2];
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][i][r] = 0.f;
  // Per-lane LDS offsets of the transpose reads, computed once: the rows a lane touches are (multiple of 4) + tap + j, so
  // the swizzle term (row & 3) = (tap + j) & 3 is a lane constant per tap and the k-step only adds a compile-time
  // immediate -- 5 address registers instead of one per (k-step, tap, half), which leaves room to read the fragments of
  // k-step s + 1 before the MFMAs of k-step s (two MFMA waves per SIMD do not cover a ~190-cycle LDS round trip).
  typedef short s16x8v __attribute__((ext_vector_type(8)));
  const int li = lane & 15, lhalf = (lane >> 4) & 1, lj = li >> 2, lq = li & 3, lwithin = 16 * lhalf + 4 * lq;
  int offA[2], offX[TAPS];
#pragma unroll
  for (int i = 0; i < 2; ++i) offA[i] = (8 * g + lj) * 128 + (((wm * 2 + i) ^ lj) << 5) + lwithin;
#pragma unroll
  for (int t = 0; t < TAPS; ++t) offX[t] = (8 * g + t + lj) * 128 + ((wn ^ ((t + lj) & 3)) << 5) + lwithin;
  auto tr8 = [&](const bf16_t* tile, int off, int ks) {     // rows ks * 16 + {0..3} and + {4..7} (+ the lane part in off)
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(tile + off + ks * 16 * 128));
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(tile + off + ks * 16 * 128 + 4 * 128));
    s16x8v r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, r);
  };
  constexpr int NKS = WG_P / 16;
  __builtin_amdgcn_s_setprio(2);   // the loader wave of this SIMD takes the issue slots the MFMA waves leave
  int buf = 0;
  {
    // software pipeline ACROSS items (see the loader loop: barrier j guarantees item j + 1): the fragments of (item k + 1, k-step 0)
    // are requested in front of the MFMAs of (item k, last k-step); NKS is even, so the fragment double-buffer keeps its parity
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    typedef uint32_t u32x2v __attribute__((ext_vector_type(2)));
    static_assert(NKS % 2 == 0, "fragment parity across items");
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    bf16x8 a[2][2];
    u32x4 f[2];
    // bias gradient = column sums of the dY tile.  The loader waves used to take them from the LDS tile between their barriers (16 LDS
    // reads + 32 adds per item ON THE LOADERS' critical path: 5 us of a 41 us launch, 10 us of the MFMA-waves-only ablation).  The four
    // waves of a channel-row group hold the same dY fragments: wave wn sums the fragments of k-step ks == wn (8 positions of one channel
    // per lane and fragment: 16 VALU ops per fragment beside the MFMAs), one register per channel block.
    const bool bias_here = p.db && ci0 == 0;
    float bsum[2] = {0.f, 0.f};
    if (count > 0) {
#pragma unroll
      for (int i = 0; i < 2; ++i) a[0][i] = tr8(ring, offA[i], 0);
      f[0] = __builtin_bit_cast(u32x4, tr8(ring + A_PIECES * 512, offX[0], 0));
    }
    for (int k = 0; k < count; ++k) {
      const bf16_t* A = ring + buf * ITEM_EL;
      const bf16_t* Xs = A + A_PIECES * 512;
      const int nb = buf + 1 == WGR_RING ? 0 : buf + 1;
      const bf16_t* An = ring + nb * ITEM_EL;
      const bf16_t* Xn = An + A_PIECES * 512;
      const bool more = k + 1 < count;
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
        uint32_t tail0 = 0;
        if (ks + 1 < NKS) {
#pragma unroll
          for (int i = 0; i < 2; ++i) a[(ks + 1) & 1][i] = tr8(A, offA[i], ks + 1);
          f[(ks + 1) & 1] = __builtin_bit_cast(u32x4, tr8(Xs, offX[0], ks + 1));
        } else {
          if constexpr (TAPS == 3) {   // rows 64 .. 67 of the haloed tile: the dword behind the last 8-position block (lanes 0 - 31 are consumed)
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(Xs + offX[0] + NKS * 16 * 128 - g * 8 * 128));
            tail0 = __builtin_bit_cast(u32x2v, lo)[0];
          }
          if (more) {
#pragma unroll
            for (int i = 0; i < 2; ++i) a[(ks + 1) & 1][i] = tr8(An, offA[i], 0);
            f[(ks + 1) & 1] = __builtin_bit_cast(u32x4, tr8(Xn, offX[0], 0));
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (bias_here && ks == wn) {
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            const u32x4 av = __builtin_bit_cast(u32x4, a[ks & 1][i]);
#pragma unroll
            for (int d = 0; d < 4; ++d) bsum[i] += __uint_as_float(av[d] << 16) + __uint_as_float(av[d] & 0xffff0000u);
          }
        }
        const u32x4 c = f[ks & 1];
        const bf16x8 bx0 = __builtin_bit_cast(bf16x8, c);
#pragma unroll
        for (int i = 0; i < 2; ++i) dx_mma(acc[0][i], a[ks & 1][i], bx0);
        if constexpr (TAPS == 3) {
          const uint32_t nxt = ks + 1 < NKS ? f[(ks + 1) & 1][0] : tail0;
          const u32x2v sw = __builtin_amdgcn_permlane32_swap(c[0], nxt, false, false);
          const uint32_t n0 = g ? sw[0] : sw[1];        // first two positions of the next 8-position block of this lane's channel
          const u32x4 t1 = {__builtin_amdgcn_alignbit(c[1], c[0], 16), __builtin_amdgcn_alignbit(c[2], c[1], 16),
                            __builtin_amdgcn_alignbit(c[3], c[2], 16), __builtin_amdgcn_alignbit(n0, c[3], 16)};
          const u32x4 t2 = {c[1], c[2], c[3], n0};
          const bf16x8 bx1 = __builtin_bit_cast(bf16x8, t1), bx2 = __builtin_bit_cast(bf16x8, t2);
#pragma unroll
          for (int i = 0; i < 2; ++i) dx_mma(acc[TAPS > 1 ? 1 : 0][i], a[ks & 1][i], bx1);
#pragma unroll
          for (int i = 0; i < 2; ++i) dx_mma(acc[TAPS > 2 ? 2 : 0][i], a[ks & 1][i], bx2);
        }
      }
      asm volatile("" ::: "memory");
      __builtin_amdgcn_s_barrier();      // item k consumed (its slot may be refilled); item k + 2 has landed
      asm volatile("" ::: "memory");
      buf = nb;
    }
    if (bias_here) {   // the two half-waves hold the two position halves of every k-step
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const float t = bsum[i] + __shfl_xor(bsum[i], 32, 64);
        const int co = co0 + wm * 64 + i * 32 + l31;
        if (g == 0 && co < Cout) atomicAdd(p.db + co, t);
      }
    }
  }
  __builtin_amdgcn_s_setprio(0);
  if (p.ws) {
    float* out = p.ws + ((size_t)split * ntiles + tile) * (TAPS * 2 * 16 * WG_THREADS);
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) out[((t * 2 + i) * 16 + r) * WG_THREADS + tid] = acc[t][i][r];
  } else {
    const int ci = ci0 + wn * 32 + l31;
    if (ci < Cin) {
#pragma unroll
      for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int co = co0 + wm * 64 + i * 32 + dx_acc_row(r, g);
            if (co < Cout) atomicAdd(p.dw + ((size_t)co * Cin + ci) * TAPS + t, acc[t][i][r]);
          }
    }
  }
}

template <int TAPS>
__global__ __launch_bounds__(WGR_THREADS, 3) void conv_wgrad_ring_kernel(WgradArgs p) { wgrad_ring_body<TAPS>(p, (int)blockIdx.x); }
// TWO weight gradients over the same rows in one launch: workgroups [0, na) run problem a, the rest problem b (the k = 1 pair of an FFT
// block -- output projection, 48 workgroups, and QKV projection, 63 -- used to run back to back on the side queue, each on a fifth of
// the CUs; side by side they take the time of the longer one).  One call site of the body: one LDS ring.
struct WgradPair { WgradArgs a, b; int na; };
template <int TAPS>
__global__ __launch_bounds__(WGR_THREADS, 3) void conv_wgrad_ring_pair_kernel(WgradPair q) {
  const bool second = (int)blockIdx.x >= q.na;
  wgrad_ring_body<TAPS>(second ? q.b : q.a, second ? (int)blockIdx.x - q.na : (int)blockIdx.x);
}

// dW += sum over the splits of the partial tiles (register order, see conv_wgrad_kernel).  One thread per 4 consecutive
// tile elements (= 4 consecutive lanes of one accumulator register: same co, 4 consecutive ci), 16-byte loads, four
// splits in flight per thread (the first version read one float per thread per split: 1.3 TB/s).
// Second launch of a weight gradient: dW += sum over the splits of the partial tiles (fixed order: no fp32 atomics on dW).
// A block owns 64 output quads; its 256 threads are 4 groups that each walk a quarter of the splits (8 loads in flight) and
// meet in LDS -- the first version gave every quad ONE thread that walked all 24-64 splits: a chain of 3-8 dependent HBM
// round trips, 24-28 us per launch for 13-38 MB (0.5-1.5 TB/s), 1.4 ms of side-stream time per training step.
template <int TAPS>
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw, int nsplit,
                                                           int ntiles, int tiles_ci, int Cout, int Cin) {
  constexpr int TILE_FLOATS = TAPS * 2 * 16 * WG_THREADS, QUADS = TILE_FLOATS / 4;
  __shared__ f32x4 part[4][64];
  const int grp = threadIdx.x >> 6, ql = threadIdx.x & 63;
  const long q = blockIdx.x * 64L + ql;
  const bool live = q < (long)ntiles * QUADS;
  const int tile = live ? (int)(q / QUADS) : 0, e0 = live ? (int)(q - (long)tile * QUADS) * 4 : 0;      // element offset inside the tile
  const float* src = ws + (size_t)tile * TILE_FLOATS + e0;
  const size_t stride = (size_t)ntiles * TILE_FLOATS;
  const int per = (nsplit + 3) >> 2, k_lo = grp * per, k_hi = min(nsplit, k_lo + per);
  f32x4 acc8[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) acc8[u] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (live) {
    int k = k_lo;
    for (; k + 7 < k_hi; k += 8) {
      f32x4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const f32x4*>(src + (size_t)(k + u) * stride);
#pragma unroll
      for (int u = 0; u < 8; ++u) acc8[u] += v[u];
    }
    f32x4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = (k + u < k_hi) ? *reinterpret_cast<const f32x4*>(src + (size_t)(k + u) * stride) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < 8; ++u) acc8[u] += v[u];
  }
  part[grp][ql] = ((acc8[0] + acc8[1]) + (acc8[2] + acc8[3])) + ((acc8[4] + acc8[5]) + (acc8[6] + acc8[7]));
  __syncthreads();
  if (grp != 0 || !live) return;
  const f32x4 sum = (part[0][ql] + part[1][ql]) + (part[2][ql] + part[3][ql]);
  const int slot = e0 / WG_THREADS, tid = e0 % WG_THREADS;
  const int t = slot / 32, i = (slot >> 4) & 1, r = slot & 15;
  const int lane = tid & 63, wave = tid >> 6, wm = wave >> 2, wn = wave & 3;
  const int co = (tile / tiles_ci) * WG_CO + wm * 64 + i * 32 + dx_acc_row(r, lane >> 5);
  const int ci = (tile % tiles_ci) * WG_CI + wn * 32 + (lane & 31);               // .. ci + 3 (lane % 4 == 0)
  if (co < Cout) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (ci + j < Cin) dw[((size_t)co * Cin + ci + j) * TAPS + t] += sum[j];
  }
}

// several weight gradients' partial tiles added to their dW by ONE launch (dx_conv1d_wgrad_multi): the entries ride in the kernel
// arguments; a block finds its entry by its first-block table and then runs the body of wgrad_reduce_kernel with run-time taps
constexpr int WG_MULTI_MAX = 8;
struct MultiReduceArgs {
  const float* ws[WG_MULTI_MAX]; float* dw[WG_MULTI_MAX];
  int nsplit[WG_MULTI_MAX], ntiles[WG_MULTI_MAX], tiles_ci[WG_MULTI_MAX], Cout[WG_MULTI_MAX], Cin[WG_MULTI_MAX], taps[WG_MULTI_MAX];
  int begin[WG_MULTI_MAX + 1];
  int n;
};
__global__ __launch_bounds__(256) void wgrad_reduce_multi_kernel(MultiReduceArgs m) {
  __shared__ f32x4 part[4][64];
  int e = 0;
#pragma unroll
  for (int i = 1; i < WG_MULTI_MAX; ++i) if (i < m.n && (int)blockIdx.x >= m.begin[i]) e = i;
  const int TAPS = m.taps[e], nsplit = m.nsplit[e], ntiles = m.ntiles[e], tiles_ci = m.tiles_ci[e], Cout = m.Cout[e], Cin = m.Cin[e];
  const float* __restrict__ ws = m.ws[e];
  float* __restrict__ dw = m.dw[e];
  const int TILE_FLOATS = TAPS * 2 * 16 * WG_THREADS, QUADS = TILE_FLOATS / 4;
  const int grp = threadIdx.x >> 6, ql = threadIdx.x & 63;
  const long q = ((long)blockIdx.x - m.begin[e]) * 64L + ql;
  const bool live = q < (long)ntiles * QUADS;
  const int tile = live ? (int)(q / QUADS) : 0, e0 = live ? (int)(q - (long)tile * QUADS) * 4 : 0;
  const float* src = ws + (size_t)tile * TILE_FLOATS + e0;
  const size_t stride = (size_t)ntiles * TILE_FLOATS;
  const int per = (nsplit + 3) >> 2, k_lo = grp * per, k_hi = min(nsplit, k_lo + per);
  f32x4 acc8[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) acc8[u] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (live) {
    int k = k_lo;
    for (; k + 7 < k_hi; k += 8) {
      f32x4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const f32x4*>(src + (size_t)(k + u) * stride);
#pragma unroll
      for (int u = 0; u < 8; ++u) acc8[u] += v[u];
    }
    f32x4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = (k + u < k_hi) ? *reinterpret_cast<const f32x4*>(src + (size_t)(k + u) * stride) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < 8; ++u) acc8[u] += v[u];
  }
  part[grp][ql] = ((acc8[0] + acc8[1]) + (acc8[2] + acc8[3])) + ((acc8[4] + acc8[5]) + (acc8[6] + acc8[7]));
  __syncthreads();
  if (grp != 0 || !live) return;
  const f32x4 sum = (part[0][ql] + part[1][ql]) + (part[2][ql] + part[3][ql]);
  const int slot = e0 / WG_THREADS, tid = e0 % WG_THREADS;
  const int t = slot / 32, i = (slot >> 4) & 1, r = slot & 15;
  const int lane = tid & 63, wave = tid >> 6, wm = wave >> 2, wn = wave & 3;
  const int co = (tile / tiles_ci) * WG_CO + wm * 64 + i * 32 + dx_acc_row(r, lane >> 5);
  const int ci = (tile % tiles_ci) * WG_CI + wn * 32 + (lane & 31);
  if (co < Cout) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (ci + j < Cin) dw[((size_t)co * Cin + ci + j) * TAPS + t] += sum[j];
  }
}

template <typename TA, typename TB, typename TC>
int launch_wgrad(const WgradArgs& a, int taps, hipStream_t s, bool reduce = true) {
  const int ntiles = dx_cdiv(a.Cout, WG_CO) * a.tiles_ci;
  dim3 grid(ntiles * a.nsplit), block(WG_THREADS);
  const bool ring = sizeof(TA) == 2 && sizeof(TB) == 2 && sizeof(TC) == 2 && a.lddy % 8 == 0 && a.ldx % 8 == 0 &&
                    a.Cout % 8 == 0 && a.Cin % 8 == 0;
  // k = 1 (QKV / output projections): round 3 kept the register-staged kernel at 192 workgroups (the ring kernel at 192 was 0.15 % slower).
  // With FEWER, longer-lived workgroups the ring kernel wins: a 64-split launch has 7 items per workgroup and is all prologue + partial
  // tile; 64 workgroups (21 splits of the 3 QKV tiles) on the 4-deep ring: 7.61 vs 7.64 ms per step, a third of the partial-tile traffic
  if (taps == 1) {
    if (ring) hipLaunchKernelGGL((conv_wgrad_ring_kernel<1>), grid, dim3(WGR_THREADS), 0, s, a);
    else hipLaunchKernelGGL((conv_wgrad_kernel<TA, TB, TC, 1>), grid, block, 0, s, a);
    if (reduce && a.ws)
      hipLaunchKernelGGL((wgrad_reduce_kernel<1>), dim3(ntiles * (1 * 2 * 16 * WG_THREADS / 4 / 64)), dim3(256), 0, s, a.ws, a.dw, a.nsplit, ntiles, a.tiles_ci, a.Cout, a.Cin);
  } else {
    if (ring) hipLaunchKernelGGL((conv_wgrad_ring_kernel<3>), grid, dim3(WGR_THREADS), 0, s, a);
    else hipLaunchKernelGGL((conv_wgrad_kernel<TA, TB, TC, 3>), grid, block, 0, s, a);
    if (reduce && a.ws)
      hipLaunchKernelGGL((wgrad_reduce_kernel<3>), dim3(ntiles * (3 * 2 * 16 * WG_THREADS / 4 / 64)), dim3(256), 0, s, a.ws, a.dw, a.nsplit, ntiles, a.tiles_ci, a.Cout, a.Cin);
  }
  DX_LAUNCH_CHECK();
  return DX_OK;
}

}  // namespace

// number of workgroup splits of the position axis for a wgrad problem (shared by the launcher and the workspace query)
static int wgrad_nsplit(int B, int N, int Cin, int Cout, int taps) {
  // Workgroup target: the weight gradients run on a side stream UNDER the data-gradient chain, so the question is not
  // how fast they finish alone but how little they slow the main stream down.  Measured per training step with the
  // 128 x 128 tiles (B = 48, T <= 1000): 64 -> 10.73 ms, 128 -> 10.46, 160 -> 10.45, 192 -> 10.30, 224 -> 10.35,
  // 256 -> 10.42, 320 -> 10.69; B = 128: 192 -> 22.1, 256 -> 22.5, 384 -> 23.2.  3/4 of the CUs, 8 waves each.
  // Round 5 (split counts rounded to multiples of 8, 6.70 ms step): 128 -> +0.02 ms (16 instead of 24 splits of the FF gradients: 0.6 GB fewer
  // partial tiles, less parallelism), 256 -> +0.13 ms (32 splits).
#ifndef DX_WG_TARGET
#define DX_WG_TARGET 192
#endif
  const int target = taps == 1 ? 64 : DX_WG_TARGET;   // k = 1: see launch_wgrad (24: +0.2 ms per step, 96 / 128: +0.02, 192: +0.05)
  const int tiles = dx_cdiv(Cout, WG_CO) * dx_cdiv(Cin, WG_CI);
  // every split costs one more partial tile to write and re-read: keep >= ~8 items (64 positions each) per workgroup,
  // 16 for the linear layers (a third of the MFMA work per item)
  int ns = target / tiles;
  const long by_work = (long)B * dx_cdiv(N, WG_P) / 16;      // (B * N rows is an upper bound of the valid rows; round 5: 16 for k = 3 too -- the phoneme-level
  // gradients had 18 splits of ~4 items each, i.e. 28 MB of partial tiles for 11 MB of operands: 9 splits, time-neutral, -0.2 GB per step)
  if (ns > by_work) ns = (int)by_work;
  if (ns >= 8) ns = (ns + 3) / 8 * 8;   // a multiple of 8: the tiles of a split then share an XCD (see wgrad_ring_body)
  return ns < 1 ? 1 : ns;
}

extern "C" long dx_conv1d_wgrad_ws_floats(int B, int N, int Cin, int Cout, int taps) {
  if (B <= 0 || N <= 0 || Cin <= 0 || Cout <= 0 || (taps != 1 && taps != 3)) return 0;
  const long tiles = (long)dx_cdiv(Cout, WG_CO) * dx_cdiv(Cin, WG_CI);
  return (long)wgrad_nsplit(B, N, Cin, Cout, taps) * tiles * taps * 2 * 16 * WG_THREADS;
}

static int wgrad_one(const void* dy, int dy_dtype, long lddy, const void* x, int x_dtype, long ldx, int compute_dtype, float* dw, float* db,
                     const int64_t* lengths, float* ws, int B, int N, int Cin, int Cout, int taps, hipStream_t s, bool reduce, WgradArgs* out,
                     bool launch = true) {
  DX_REQUIRE(dy && x && dw, DX_ERR_ARG, "dx_conv1d_wgrad: null pointer");
  DX_REQUIRE(B > 0 && N > 0 && Cin > 0 && Cout > 0, DX_ERR_SHAPE, "dx_conv1d_wgrad: empty shape");
  DX_REQUIRE(Cin % 8 == 0 && Cout % 8 == 0 && ldx % 8 == 0 && lddy % 8 == 0, DX_ERR_SHAPE,
             "dx_conv1d_wgrad: Cin, Cout and the row strides must be multiples of 8");
  DX_REQUIRE(taps == 1 || taps == 3, DX_ERR_UNSUPPORTED, "dx_conv1d_wgrad: taps=%d (only 1 and 3)", taps);
  // k = 1 weight gradients (QKV / output projections, 16 k - 49 k elements): partial tiles + reduce launch, or fp32 atomics on dW
  WgradArgs a{dy, lddy, x, ldx, dw, db, lengths, ws, B, N, Cin, Cout, wgrad_nsplit(B, N, Cin, Cout, taps), dx_cdiv(Cin, WG_CI)};
  if (out) *out = a;
  if (!launch) {   // (the caller launches: only the dtype combination is checked here)
    const bool ok = compute_dtype == DX_BF16 ? ((dy_dtype == DX_F32 || dy_dtype == DX_BF16) && (x_dtype == DX_F32 || x_dtype == DX_BF16))
                                             : (compute_dtype == DX_F32 && dy_dtype == DX_F32 && x_dtype == DX_F32);
    DX_REQUIRE(ok, DX_ERR_DTYPE, "dx_conv1d_wgrad: unsupported dtype combination dy=%d x=%d compute=%d", dy_dtype, x_dtype, compute_dtype);
    return DX_OK;
  }
  if (compute_dtype == DX_BF16) {
    if (dy_dtype == DX_F32 && x_dtype == DX_F32) return launch_wgrad<float, float, bf16_t>(a, taps, s, reduce);
    if (dy_dtype == DX_F32 && x_dtype == DX_BF16) return launch_wgrad<float, bf16_t, bf16_t>(a, taps, s, reduce);
    if (dy_dtype == DX_BF16 && x_dtype == DX_F32) return launch_wgrad<bf16_t, float, bf16_t>(a, taps, s, reduce);
    if (dy_dtype == DX_BF16 && x_dtype == DX_BF16) return launch_wgrad<bf16_t, bf16_t, bf16_t>(a, taps, s, reduce);
  } else if (compute_dtype == DX_F32) {
    if (dy_dtype == DX_F32 && x_dtype == DX_F32) return launch_wgrad<float, float, float>(a, taps, s, reduce);
  }
  dx_set_error("dx_conv1d_wgrad: unsupported dtype combination dy=%d x=%d compute=%d", dy_dtype, x_dtype, compute_dtype);
  return DX_ERR_DTYPE;
}

extern "C" int dx_conv1d_wgrad(const void* dy, int dy_dtype, long lddy, const void* x, int x_dtype, long ldx,
                               int compute_dtype, float* dw, float* db, const int64_t* lengths, float* ws, int B, int N,
                               int Cin, int Cout, int taps, void* stream) {
  return wgrad_one(dy, dy_dtype, lddy, x, x_dtype, ldx, compute_dtype, dw, db, lengths, ws, B, N, Cin, Cout, taps, (hipStream_t)stream, true, nullptr);
}

extern "C" int dx_wgrad_desc_size(void) { return (int)sizeof(DxWgradDesc); }

extern "C" long dx_conv1d_wgrad_multi_ws_floats(const DxWgradDesc* d, int n, int B, int N) {
  long total = 0;
  for (int i = 0; i < n; ++i) total += dx_conv1d_wgrad_ws_floats(B, N, d[i].Cin, d[i].Cout, d[i].taps);
  return total;
}

extern "C" int dx_conv1d_wgrad_multi(const DxWgradDesc* d, int n, int compute_dtype, const int64_t* lengths, float* ws, int B, int N,
                                     void* stream) {
  DX_REQUIRE(d && n > 0 && n <= WG_MULTI_MAX && ws, DX_ERR_ARG, "dx_conv1d_wgrad_multi: 1..%d descriptors and a workspace", WG_MULTI_MAX);
  hipStream_t s = (hipStream_t)stream;
  MultiReduceArgs m{};
  int blocks = 0;
  float* wsp = ws;
  auto ring1 = [&](int i) {   // a k = 1 weight gradient the ring kernel takes (launch_wgrad's condition)
    return i < n && d[i].taps == 1 && compute_dtype == DX_BF16 && d[i].dy_dtype == DX_BF16 && d[i].x_dtype == DX_BF16 && d[i].lddy % 8 == 0 &&
           d[i].ldx % 8 == 0 && d[i].Cout % 8 == 0 && d[i].Cin % 8 == 0;
  };
  bool paired = false;   // descriptor i was launched together with i - 1
  for (int i = 0; i < n; ++i) {
    WgradArgs a;
    const bool pair = !paired && ring1(i) && ring1(i + 1);
    if (int rc = wgrad_one(d[i].dy, d[i].dy_dtype, d[i].lddy, d[i].x, d[i].x_dtype, d[i].ldx, compute_dtype, d[i].dw, d[i].db, lengths, wsp, B, N,
                           d[i].Cin, d[i].Cout, d[i].taps, s, false, &a, !(pair || paired))) return rc;
    if (pair) {
      WgradArgs b2;
      float* wsb = wsp + dx_conv1d_wgrad_ws_floats(B, N, d[i].Cin, d[i].Cout, d[i].taps);
      if (int rc = wgrad_one(d[i + 1].dy, d[i + 1].dy_dtype, d[i + 1].lddy, d[i + 1].x, d[i + 1].x_dtype, d[i + 1].ldx, compute_dtype, d[i + 1].dw,
                             d[i + 1].db, lengths, wsb, B, N, d[i + 1].Cin, d[i + 1].Cout, d[i + 1].taps, s, false, &b2, false)) return rc;
      WgradPair q{a, b2, dx_cdiv(a.Cout, WG_CO) * a.tiles_ci * a.nsplit};
      const int nb = dx_cdiv(b2.Cout, WG_CO) * b2.tiles_ci * b2.nsplit;
      hipLaunchKernelGGL((conv_wgrad_ring_pair_kernel<1>), dim3(q.na + nb), dim3(WGR_THREADS), 0, s, q);
    }
    paired = pair;
    const int ntiles = dx_cdiv(a.Cout, WG_CO) * a.tiles_ci;
    if (a.ws) {
      const int k = m.n++;
      m.ws[k] = a.ws; m.dw[k] = a.dw; m.nsplit[k] = a.nsplit; m.ntiles[k] = ntiles; m.tiles_ci[k] = a.tiles_ci; m.Cout[k] = a.Cout; m.Cin[k] = a.Cin;
      m.taps[k] = d[i].taps; m.begin[k] = blocks;
      blocks += ntiles * (d[i].taps * 2 * 16 * WG_THREADS / 4 / 64);
    }
    wsp += dx_conv1d_wgrad_ws_floats(B, N, d[i].Cin, d[i].Cout, d[i].taps);
  }
  m.begin[m.n] = blocks;
  if (m.n) hipLaunchKernelGGL(wgrad_reduce_multi_kernel, dim3(blocks), dim3(256), 0, s, m);
  DX_LAUNCH_CHECK();
  return DX_OK;
}
