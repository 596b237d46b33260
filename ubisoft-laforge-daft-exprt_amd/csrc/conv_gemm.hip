// K1 / K3 / K12 -- k-tap conv on channel-last activations as an implicit GEMM on MFMA (gfx950).
//
//   Y[b, n, co] = bias[co] + sum_{tap, ci} X[b, n + tap - taps/2, ci] * W[tap][co][ci]
//
// Tiling: one workgroup (4 waves) computes a 128-position x 128-channel tile of ONE utterance, so the
// conv halo is simply rows n0-1 .. n0+128 of that utterance (rows outside [0, N) are zero padding).
// The K loop walks Cin in chunks of 32; per chunk the haloed activation tile (130 x 32) and the weight
// tile (taps x 128 x 32) are staged in LDS (rows padded by 16 B -> conflict-free ds_read_b128 fragment
// reads), and every tap reuses the same activation tile at a row offset -- no im2col is materialised.
// Each wave owns a 64 x 64 sub-tile = 2 x 2 MFMA 32x32 accumulators (64 VGPRs).
// Operand type TC: bf16 (v_mfma_f32_32x32x16_bf16) or fp32 (v_mfma_f32_32x32x2_f32, exact fp32 mode).
#include "dx_common.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 32, NTHREADS = 256;

template <typename TC> struct Pad;
template <> struct Pad<bf16_t> { static constexpr int value = 8; };
template <> struct Pad<float> { static constexpr int value = 4; };

struct ConvArgs {
  const void* x; long ldx;
  const void* w; const float* bias;
  void* y; long ldy;
  const void* gate;
  const int64_t* mask_len;
  int N, Cin, Cout, flags;
};

template <typename TA, typename TC, typename TO, typename TG, int TAPS>
__global__ __launch_bounds__(NTHREADS) void conv_gemm_kernel(ConvArgs p) {
  constexpr int HALO = TAPS / 2;
  constexpr int AROWS = BM + TAPS - 1;
  constexpr int LDS_K = BK + Pad<TC>::value;
  typedef typename Vec8<TC>::type frag_t;
  __shared__ __attribute__((aligned(16))) TC As[AROWS * LDS_K];
  __shared__ __attribute__((aligned(16))) TC Ws[TAPS * BN * LDS_K];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, g = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  const int n0 = blockIdx.x * BM, b = blockIdx.y, co0 = blockIdx.z * BN;
  const int N = p.N, Cin = p.Cin, Cout = p.Cout;
  const TA* X = reinterpret_cast<const TA*>(p.x) + (size_t)b * N * p.ldx;
  const TC* W = reinterpret_cast<const TC*>(p.w);

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  for (int k0 = 0; k0 < Cin; k0 += BK) {
    // ---- stage the haloed activation tile
    for (int c = tid; c < AROWS * (BK / 8); c += NTHREADS) {
      const int r = c >> 2, kc = (c & 3) * 8;
      const int n = n0 + r - HALO, ci = k0 + kc;
      frag_t v;
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = (TC)0.f;
      if (n >= 0 && n < N && ci < Cin) v = dx_load8<TA, TC>(X + (size_t)n * p.ldx + ci);
      *reinterpret_cast<frag_t*>(&As[r * LDS_K + kc]) = v;
    }
    // ---- stage the weight tile
    for (int c = tid; c < TAPS * BN * (BK / 8); c += NTHREADS) {
      const int tap = c / (BN * 4), rem = c - tap * (BN * 4);
      const int row = rem >> 2, kc = (rem & 3) * 8;
      const int co = co0 + row, ci = k0 + kc;
      frag_t v;
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = (TC)0.f;
      if (co < Cout && ci < Cin) v = *reinterpret_cast<const frag_t*>(W + ((size_t)tap * Cout + co) * Cin + ci);
      *reinterpret_cast<frag_t*>(&Ws[(tap * BN + row) * LDS_K + kc]) = v;
    }
    __syncthreads();
#pragma unroll
    for (int tap = 0; tap < TAPS; ++tap) {
#pragma unroll
      for (int ks = 0; ks < BK / 16; ++ks) {
        frag_t a[2], bf[2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
          a[i] = *reinterpret_cast<const frag_t*>(&As[(wm * 64 + i * 32 + l31 + tap) * LDS_K + ks * 16 + g * 8]);
#pragma unroll
        for (int j = 0; j < 2; ++j)
          bf[j] = *reinterpret_cast<const frag_t*>(&Ws[(tap * BN + wn * 64 + j * 32 + l31) * LDS_K + ks * 16 + g * 8]);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) dx_mma(acc[i][j], a[i], bf[j]);
      }
    }
    __syncthreads();
  }

  // ---- epilogue: bias, ReLU, ReLU-derivative gate, length mask, store
  const bool relu = p.flags & DX_CONV_RELU, trans = p.flags & DX_CONV_TRANSPOSED_OUT, accum = p.flags & DX_CONV_ACCUMULATE;
  const int len = p.mask_len ? (int)p.mask_len[b] : N;
  TO* Y = reinterpret_cast<TO*>(p.y);
  const TG* G = reinterpret_cast<const TG*>(p.gate);
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int co = co0 + wn * 64 + j * 32 + l31;
    if (co >= Cout) continue;
    const float bv = p.bias ? p.bias[co] : 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = n0 + wm * 64 + i * 32 + dx_acc_row(r, g);
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

template <typename TA, typename TC, typename TO, typename TG>
int launch_taps(const ConvArgs& a, int B, int taps, hipStream_t s) {
  dim3 grid(dx_cdiv(a.N, BM), B, dx_cdiv(a.Cout, BN)), block(NTHREADS);
  if (taps == 1)
    hipLaunchKernelGGL((conv_gemm_kernel<TA, TC, TO, TG, 1>), grid, block, 0, s, a);
  else
    hipLaunchKernelGGL((conv_gemm_kernel<TA, TC, TO, TG, 3>), grid, block, 0, s, a);
  DX_LAUNCH_CHECK();
  return DX_OK;
}


// ---- weight gradient ------------------------------------------------------------------------
//   dW[co][ci][tap] += sum_{b, n} dY[b, n, co] * X[b, n + tap - taps/2, ci]        (fp32, PyTorch layout)
//   db[co]          += sum_{b, n} dY[b, n, co]
// GEMM with the positions as the contraction axis.  A workgroup owns a 128 (co) x 64 (ci) x taps output tile and a
// slice of the utterances (split-K over the batch; partial tiles are combined with fp32 atomics).  Per step of 32
// positions the dY tile [32][128] and the haloed X tile [34][64] are staged in LDS in their natural row-major
// layout; both MFMA operands need "8 consecutive positions for one channel", which the LDS transpose read
// (ds_read_b64_tr_b16, gather8) delivers without a software transpose; all taps reuse the same X tile at a row
// offset.  Wave (wm, wn) accumulates 64 co x 32 ci x taps = 2*taps MFMA 32x32 tiles.
constexpr int WG_CO = 128, WG_CI = 64, WG_P = 32;

struct WgradArgs {
  const void* dy; long lddy; const void* x; long ldx;
  float* dw; float* db; const int64_t* lengths;
  int B, N, Cin, Cout, nsplit, tiles_ci;
};

template <typename TA, typename TB, typename TC, int TAPS>
__global__ __launch_bounds__(NTHREADS) void conv_wgrad_kernel(WgradArgs p) {
  constexpr int HALO = TAPS / 2, XROWS = WG_P + TAPS - 1;
  constexpr int LDA = WG_CO + Pad<TC>::value, LDB = WG_CI + Pad<TC>::value;
  typedef typename Vec8<TC>::type frag_t;
  __shared__ __attribute__((aligned(16))) TC dYs[WG_P * LDA];
  __shared__ __attribute__((aligned(16))) TC Xs[XROWS * LDB];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, g = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  const int tile = blockIdx.x, co0 = (tile / p.tiles_ci) * WG_CO, ci0 = (tile % p.tiles_ci) * WG_CI;
  const int N = p.N, Cin = p.Cin, Cout = p.Cout;
  const int b_begin = (int)((long)p.B * blockIdx.y / p.nsplit), b_end = (int)((long)p.B * (blockIdx.y + 1) / p.nsplit);

  f32x16 acc[TAPS][2];
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][i][r] = 0.f;
  float bsum = 0.f;
  const bool do_bias = p.db && ci0 == 0 && tid < WG_CO;

  for (int b = b_begin; b < b_end; ++b) {
    const TA* dY = reinterpret_cast<const TA*>(p.dy) + (size_t)b * N * p.lddy;
    const TB* X = reinterpret_cast<const TB*>(p.x) + (size_t)b * N * p.ldx;
    // rows beyond len + halo carry exactly-zero gradients (masked upstream): skip them
    const int nlim = p.lengths ? min(N, (int)p.lengths[b] + 2) : N;
    for (int n0 = 0; n0 < nlim; n0 += WG_P) {
      for (int c = tid; c < WG_P * (WG_CO / 8); c += NTHREADS) {
        const int r = c >> 4, kc = (c & 15) * 8;
        const int n = n0 + r, co = co0 + kc;
        frag_t v = zero8<TC>();
        if (n < N && co < Cout) v = dx_load8<TA, TC>(dY + (size_t)n * p.lddy + co);
        *reinterpret_cast<frag_t*>(&dYs[r * LDA + kc]) = v;
      }
      for (int c = tid; c < XROWS * (WG_CI / 8); c += NTHREADS) {
        const int r = c >> 3, kc = (c & 7) * 8;
        const int n = n0 + r - HALO, ci = ci0 + kc;
        frag_t v = zero8<TC>();
        if (n >= 0 && n < N && ci < Cin) v = dx_load8<TB, TC>(X + (size_t)n * p.ldx + ci);
        *reinterpret_cast<frag_t*>(&Xs[r * LDB + kc]) = v;
      }
      __syncthreads();
      if (do_bias) {
#pragma unroll 8
        for (int r = 0; r < WG_P; ++r) bsum += (float)dYs[r * LDA + tid];
      }
#pragma unroll
      for (int ks = 0; ks < WG_P / 16; ++ks) {
        const int kA = ks * 16 + 8 * g, kB = kA + 4;
        frag_t a[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) a[i] = gather8<TC, 32>(dYs, LDA, kA, kB, wm * 64 + i * 32, lane);
#pragma unroll
        for (int t = 0; t < TAPS; ++t) {
          frag_t bx = gather8<TC, 32>(Xs, LDB, kA + t, kB + t, wn * 32, lane);
#pragma unroll
          for (int i = 0; i < 2; ++i) dx_mma(acc[t][i], a[i], bx);
        }
      }
      __syncthreads();
    }
  }
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
  if (do_bias && co0 + tid < Cout) atomicAdd(p.db + co0 + tid, bsum);
}

template <typename TA, typename TB, typename TC>
int launch_wgrad(const WgradArgs& a, int taps, hipStream_t s) {
  dim3 grid(dx_cdiv(a.Cout, WG_CO) * a.tiles_ci, a.nsplit), block(NTHREADS);
  if (taps == 1)
    hipLaunchKernelGGL((conv_wgrad_kernel<TA, TB, TC, 1>), grid, block, 0, s, a);
  else
    hipLaunchKernelGGL((conv_wgrad_kernel<TA, TB, TC, 3>), grid, block, 0, s, a);
  DX_LAUNCH_CHECK();
  return DX_OK;
}

// ---- weight packing -------------------------------------------------------------------------
template <typename TC>
__global__ void pack_weight_kernel(const float* __restrict__ w, TC* __restrict__ out, int Cout, int Cin, int taps, int tf) {
  const size_t total = (size_t)Cout * Cin * taps;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    // i enumerates the OUTPUT linearly
    if (!tf) {
      const int ci = i % Cin; const size_t t = i / Cin; const int co = t % Cout; const int tap = t / Cout;
      out[i] = (TC)w[((size_t)co * Cin + ci) * taps + tap];
    } else {
      const int co = i % Cout; const size_t t = i / Cout; const int ci = t % Cin; const int tap = t / Cin;
      out[i] = (TC)w[((size_t)co * Cin + ci) * taps + (taps - 1 - tap)];
    }
  }
}

}  // namespace

extern "C" int dx_conv1d(const void* x, int x_dtype, long ldx, const void* w_packed, int w_dtype, const float* bias,
                         void* y, int y_dtype, long ldy, const void* relu_gate, int gate_dtype,
                         const int64_t* mask_lengths, int B, int N, int Cin, int Cout, int taps, int flags, void* stream) {
  DX_REQUIRE(x && w_packed && y, DX_ERR_ARG, "dx_conv1d: null pointer");
  DX_REQUIRE(B > 0 && N > 0 && Cin > 0 && Cout > 0, DX_ERR_SHAPE, "dx_conv1d: empty shape B=%d N=%d Cin=%d Cout=%d", B, N, Cin, Cout);
  DX_REQUIRE(Cin % 8 == 0 && ldx % 8 == 0, DX_ERR_SHAPE, "dx_conv1d: Cin (%d) and ldx (%ld) must be multiples of 8", Cin, ldx);
  DX_REQUIRE(taps == 1 || taps == 3, DX_ERR_UNSUPPORTED, "dx_conv1d: taps=%d (only 1 and 3)", taps);
  DX_REQUIRE(B <= 65535 && dx_cdiv(Cout, BN) <= 65535, DX_ERR_SHAPE, "dx_conv1d: grid too large");
  ConvArgs a{x, ldx, w_packed, bias, y, ldy, relu_gate, mask_lengths, N, Cin, Cout, flags};
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

extern "C" int dx_pack_conv_weight(const float* w, void* out, int out_dtype, int Cout, int Cin, int taps,
                                   int transpose_flip, void* stream) {
  DX_REQUIRE(w && out, DX_ERR_ARG, "dx_pack_conv_weight: null pointer");
  DX_REQUIRE(Cout > 0 && Cin > 0 && taps > 0, DX_ERR_SHAPE, "dx_pack_conv_weight: empty shape");
  const size_t total = (size_t)Cout * Cin * taps;
  const int grid = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
  hipStream_t s = (hipStream_t)stream;
  if (out_dtype == DX_BF16)
    hipLaunchKernelGGL(pack_weight_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, w, (bf16_t*)out, Cout, Cin, taps, transpose_flip);
  else if (out_dtype == DX_F32)
    hipLaunchKernelGGL(pack_weight_kernel<float>, dim3(grid), dim3(256), 0, s, w, (float*)out, Cout, Cin, taps, transpose_flip);
  else {
    dx_set_error("dx_pack_conv_weight: bad out_dtype %d", out_dtype);
    return DX_ERR_DTYPE;
  }
  DX_LAUNCH_CHECK();
  return DX_OK;
}

extern "C" int dx_conv1d_wgrad(const void* dy, int dy_dtype, long lddy, const void* x, int x_dtype, long ldx,
                               int compute_dtype, float* dw, float* db, const int64_t* lengths, int B, int N, int Cin,
                               int Cout, int taps, void* stream) {
  DX_REQUIRE(dy && x && dw, DX_ERR_ARG, "dx_conv1d_wgrad: null pointer");
  DX_REQUIRE(B > 0 && N > 0 && Cin > 0 && Cout > 0, DX_ERR_SHAPE, "dx_conv1d_wgrad: empty shape");
  DX_REQUIRE(Cin % 8 == 0 && Cout % 8 == 0 && ldx % 8 == 0 && lddy % 8 == 0, DX_ERR_SHAPE,
             "dx_conv1d_wgrad: Cin, Cout and the row strides must be multiples of 8");
  DX_REQUIRE(taps == 1 || taps == 3, DX_ERR_UNSUPPORTED, "dx_conv1d_wgrad: taps=%d (only 1 and 3)", taps);
  WgradArgs a{dy, lddy, x, ldx, dw, db, lengths, B, N, Cin, Cout, 1, dx_cdiv(Cin, WG_CI)};
  const int tiles = dx_cdiv(Cout, WG_CO) * a.tiles_ci;
  int ns = 1024 / tiles;                 // enough workgroups for 256 CUs, as few atomic passes as possible
  a.nsplit = ns < 1 ? 1 : (ns > B ? B : ns);
  hipStream_t s = (hipStream_t)stream;
  if (compute_dtype == DX_BF16) {
    if (dy_dtype == DX_F32 && x_dtype == DX_F32) return launch_wgrad<float, float, bf16_t>(a, taps, s);
    if (dy_dtype == DX_F32 && x_dtype == DX_BF16) return launch_wgrad<float, bf16_t, bf16_t>(a, taps, s);
    if (dy_dtype == DX_BF16 && x_dtype == DX_F32) return launch_wgrad<bf16_t, float, bf16_t>(a, taps, s);
    if (dy_dtype == DX_BF16 && x_dtype == DX_BF16) return launch_wgrad<bf16_t, bf16_t, bf16_t>(a, taps, s);
  } else if (compute_dtype == DX_F32) {
    if (dy_dtype == DX_F32 && x_dtype == DX_F32) return launch_wgrad<float, float, float>(a, taps, s);
  }
  dx_set_error("dx_conv1d_wgrad: unsupported dtype combination dy=%d x=%d compute=%d", dy_dtype, x_dtype, compute_dtype);
  return DX_ERR_DTYPE;
}
