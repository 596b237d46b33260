// Weight packing: fp32 PyTorch-layout conv / linear weights -> the operand copies the GEMMs read ([tap][co][ci] and its transposed /
// flipped form for the data gradients, MFMA-fragment order for the register- and split-K kernels), single and batched.
#include <stdlib.h>

#include "dx_common.h"

namespace {

#include "conv_common.h"

// fragment-order copy of a packed [3][Cout][Cin] bf16 weight (Cout a multiple of 32): out[chunk][tap][half][block][lane][8] =
// w[tap][32 block + (lane & 31)][32 chunk + 16 half + 8 (lane >> 5) + 0..7], block < Cout / 32
__device__ __forceinline__ void frag_major_copy(const bf16_t* __restrict__ w, bf16_t* __restrict__ out, int Cin, int Cout, long i) {
  const int nblk = Cout >> 5;
  const int lane = i & 63;
  const long r1 = i >> 6;
  const int c = (int)(r1 % nblk);
  const long r2 = r1 / nblk;
  const int half = (int)(r2 & 1);
  const long rest = r2 >> 1;
  const int tap = (int)(rest % 3), kc = (int)(rest / 3);
  const bf16_t* src = w + ((size_t)tap * Cout + c * 32 + (lane & 31)) * Cin + kc * 32 + half * 16 + (lane >> 5) * 8;
  *reinterpret_cast<bf16x8*>(out + i * 8) = *reinterpret_cast<const bf16x8*>(src);
}
__global__ void pack_frag_major_kernel(const bf16_t* __restrict__ w, bf16_t* __restrict__ out, int Cin, int Cout) {
  const long total = (long)(Cin >> 5) * 3 * 2 * (Cout >> 5) * 64;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) frag_major_copy(w, out, Cin, Cout, i);
}
// weight-stationary dispatch: bf16 operands, plain row-major vectorised output, no fused LayerNorm / accumulate
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


// all GEMM weights of the model in ONE launch: descriptor table on the device, flat element index -> (descriptor, element)
// begin = running count of 32 (co) x 32 (ci) bricks over the table.  A workgroup moves ONE brick (all taps) through LDS:
// the fp32 source rows are read as contiguous 32 * taps floats, both packed layouts leave as 64-byte row segments
// (the first version computed one output element per thread with a per-element table search and, for the
// transposed layout, reads 1.5 KB apart: 85 + 107 us per step for 88 MB each).
struct PackDesc { const float* w; void* out; int Cout, Cin, taps, tf; long begin; };
template <typename TC>
__global__ __launch_bounds__(256) void pack_batched_kernel(const PackDesc* __restrict__ descs, int n, long total) {
  __shared__ float tile[32][3 * 32 + 1];
  const long u = blockIdx.x;
  int lo = 0, hi = n - 1;
  while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (descs[mid].begin <= u) lo = mid; else hi = mid - 1; }
  const PackDesc d = descs[lo];
  const int taps = d.taps > 3 ? 3 : d.taps;
  const int cib = dx_cdiv(d.Cin, 32), brick = (int)(u - d.begin);
  const int co0 = (brick / cib) * 32, ci0 = (brick % cib) * 32;
  const int rowlen = 32 * taps;                       // floats of one source row segment: [ci0 .. ci0+32) x taps
  for (int i = threadIdx.x; i < 32 * rowlen; i += 256) {
    const int r = i / rowlen, k = i - r * rowlen;     // k = (ci - ci0) * taps + tap
    const int co = co0 + r, ci = ci0 + k / taps;
    tile[r][k] = (co < d.Cout && ci < d.Cin) ? d.w[((long)co * d.Cin + ci0) * taps + k] : 0.f;
  }
  __syncthreads();
  TC* out = reinterpret_cast<TC*>(d.out);
  for (int i = threadIdx.x; i < taps * 1024; i += 256) {
    const int tap = i >> 10, r = (i >> 5) & 31, c = i & 31;
    if (!d.tf) {            // out[tap][co][ci]: r = co row, c = ci
      if (co0 + r < d.Cout && ci0 + c < d.Cin) out[((long)tap * d.Cout + co0 + r) * d.Cin + ci0 + c] = (TC)tile[r][c * taps + tap];
    } else {                // out[taps-1-tap][ci][co]: r = ci row, c = co
      if (ci0 + r < d.Cin && co0 + c < d.Cout) out[((long)(taps - 1 - tap) * d.Cin + ci0 + r) * d.Cout + co0 + c] = (TC)tile[c][r * taps + tap];
    }
  }
}

}  // namespace

extern "C" int dx_pack_desc_size(void) { return (int)sizeof(PackDesc); }

extern "C" int dx_pack_conv_weights_batched(const void* descs_dev, int n, long total_bricks, int out_dtype, void* stream) {
  DX_REQUIRE(descs_dev && n > 0 && total_bricks > 0 && total_bricks < (1L << 31), DX_ERR_ARG, "dx_pack_conv_weights_batched: bad arguments");
  const unsigned grid = (unsigned)total_bricks;
  if (out_dtype == DX_BF16)
    hipLaunchKernelGGL(pack_batched_kernel<bf16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const PackDesc*)descs_dev, n, total_bricks);
  else if (out_dtype == DX_F32)
    hipLaunchKernelGGL(pack_batched_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const PackDesc*)descs_dev, n, total_bricks);
  else { dx_set_error("dx_pack_conv_weights_batched: bad out_dtype %d", out_dtype); return DX_ERR_DTYPE; }
  DX_LAUNCH_CHECK();
  return DX_OK;
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

struct FragDesc { const void* src; void* dst; int Cin; int Cout; };
namespace {
__global__ void pack_frag_major_batched_kernel(const FragDesc* __restrict__ descs) {
  const FragDesc d = descs[blockIdx.y];
  const long total = (long)(d.Cin >> 5) * 3 * 2 * (d.Cout >> 5) * 64;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x)
    frag_major_copy(reinterpret_cast<const bf16_t*>(d.src), reinterpret_cast<bf16_t*>(d.dst), d.Cin, d.Cout, i);
}
}  // namespace

extern "C" int dx_frag_desc_size(void) { return (int)sizeof(FragDesc); }

extern "C" int dx_pack_frag_major_batched(const void* descs_dev, int n, long max_elems, void* stream) {
  DX_REQUIRE(descs_dev && n > 0, DX_ERR_ARG, "dx_pack_frag_major_batched: empty table");
  DX_REQUIRE(max_elems > 0 && max_elems % 8 == 0, DX_ERR_SHAPE, "dx_pack_frag_major_batched: max_elems=%ld", max_elems);
  long blocks = (max_elems / 8 + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(pack_frag_major_batched_kernel, dim3((unsigned)blocks, (unsigned)n), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const FragDesc*>(descs_dev));
  DX_LAUNCH_CHECK();
  return DX_OK;
}

extern "C" int dx_pack_frag_major(const void* w_packed, void* out, int Cin, int Cout, void* stream) {
  DX_REQUIRE(w_packed && out, DX_ERR_ARG, "dx_pack_frag_major: null pointer");
  DX_REQUIRE(Cin > 0 && Cin % 32 == 0 && Cout > 0 && Cout % 32 == 0, DX_ERR_SHAPE, "dx_pack_frag_major: Cin=%d, Cout=%d must be multiples of 32", Cin, Cout);
  const long total = (long)(Cin >> 5) * 3 * 2 * (Cout >> 5) * 64;
  long blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(pack_frag_major_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const bf16_t*>(w_packed), reinterpret_cast<bf16_t*>(out), Cin, Cout);
  DX_LAUNCH_CHECK();
  return DX_OK;
}
