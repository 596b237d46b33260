// Adam over the flat parameter buffer FUSED with the refresh of the MFMA operand copies of the GEMM weights.
//
// The reference's step ends with torch.optim.Adam (train.py:299-301, 399-401) and its next forward pass reads the fp32 weights
// directly.  Here every GEMM reads a bf16 (or fp32) copy of its weight in an MFMA-friendly layout -- forward packing
// [tap][Cout][Cin], data-gradient packing [tap'][Cin][Cout] with flipped taps, and for some layers a fragment-order copy of either
// (dx_pack_frag_major) -- and those copies have to follow every optimizer step.  As separate launches that is one pass over the
// parameters for Adam (28 B / parameter) plus two more reads of all GEMM weights by the pack kernels (2 x 59 MB) and two more
// launches for the fragment-order copies.  Fused, a workgroup owns one 32 (co) x 32 (ci) x taps brick of one weight: it applies Adam
// to the brick (same arithmetic, element for element, as adam_kernel), keeps the updated values in LDS and writes every copy of the
// brick from there -- the weights are read once, by Adam.  Everything that is not a GEMM weight (biases, LayerNorm parameters,
// embeddings, the small heads: 3 % of the parameters) is updated by the same launch through a table of flat ranges.
#include "dx_common.h"

namespace {

struct AdamPackDesc {
  long off;                 // offset of the weight (Cout, Cin, taps) fp32 in the flat parameter / gradient / moment buffers
  void* fwd;                // [taps][Cout][Cin]                      (NULL = not kept)
  void* tr;                 // [taps][Cin][Cout], taps flipped        (NULL = not kept)
  void* frag_fwd;           // dx_pack_frag_major of fwd              (NULL = not kept; bf16, taps = 3, Cout % 32 == 0, Cin % 32 == 0)
  void* frag_tr;            // dx_pack_frag_major of tr
  int Cout, Cin, taps, pad;
  long begin;               // running count of 32 x 32 bricks over the table
};
struct AdamFlatDesc { long off, n, begin; };   // a range of other parameters; begin = running count of FLAT_BLOCK-element blocks
constexpr int FLAT_BLOCK = 4096;

struct AdamHyper { float lr, b1, b2, eps, wd, bc1, bc2_sqrt; };

__device__ __forceinline__ float block_sum_256(float v, float* red) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

// one parameter: the arithmetic of adam_kernel (train_ops.hip) with coef = 1 (infinite clipping threshold)
__device__ __forceinline__ float adam_one(float pi, float graw, float& mi, float& vi, const AdamHyper& h, float step) {
  const float gi = graw + h.wd * pi;
  mi = h.b1 * mi + (1.f - h.b1) * gi;
  vi = h.b2 * vi + (1.f - h.b2) * gi * gi;
  return pi - step * mi / (sqrtf(vi) / h.bc2_sqrt + h.eps);
}

template <typename TC>
__global__ __launch_bounds__(256) void adam_pack_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                        float* __restrict__ v, const AdamPackDesc* __restrict__ bricks, int nb,
                                                        long total_bricks, const AdamFlatDesc* __restrict__ flats, int nf, AdamHyper h,
                                                        float* gnorm_accum, const DxStepScalars* sc) {
  __shared__ float tile[32][3 * 32 + 1];
  __shared__ float red[4];
  if (sc) { h.lr = sc->lr; h.bc1 = sc->bc1; h.bc2_sqrt = sc->bc2_sqrt; }
  const float step = h.lr / h.bc1;
  const long u = blockIdx.x;
  float gsq = 0.f;
  if (u >= total_bricks) {
    // ---- a block of a flat range
    const long fb = u - total_bricks;
    int lo = 0, hi = nf - 1;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (flats[mid].begin <= fb) lo = mid; else hi = mid - 1; }
    const AdamFlatDesc d = flats[lo];
    const long e0 = (fb - d.begin) * FLAT_BLOCK, e1 = e0 + FLAT_BLOCK < d.n ? e0 + FLAT_BLOCK : d.n;
    for (long e = e0 + threadIdx.x; e < e1; e += 256) {
      const long i = d.off + e;
      const float graw = g[i];
      float mi = m[i], vi = v[i];
      gsq += graw * graw;
      p[i] = adam_one(p[i], graw, mi, vi, h, step);
      m[i] = mi; v[i] = vi;
    }
  } else {
    // ---- a 32 x 32 x taps brick of a GEMM weight
    int lo = 0, hi = nb - 1;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (bricks[mid].begin <= u) lo = mid; else hi = mid - 1; }
    const AdamPackDesc d = bricks[lo];
    const int taps = d.taps > 3 ? 3 : d.taps;
    const int cib = dx_cdiv(d.Cin, 32), brick = (int)(u - d.begin);
    const int cb = brick / cib, kb = brick % cib, co0 = cb * 32, ci0 = kb * 32;
    const int rowlen = 32 * taps;                     // floats of one source row segment: [ci0 .. ci0 + 32) x taps
    for (int i = threadIdx.x; i < 32 * rowlen; i += 256) {
      const int r = i / rowlen, k = i - r * rowlen;   // k = (ci - ci0) * taps + tap
      const int co = co0 + r, ci = ci0 + k / taps;
      float out = 0.f;
      if (co < d.Cout && ci < d.Cin) {
        const long idx = d.off + ((long)co * d.Cin + ci0) * taps + k;
        const float graw = g[idx];
        float mi = m[idx], vi = v[idx];
        gsq += graw * graw;
        out = adam_one(p[idx], graw, mi, vi, h, step);
        p[idx] = out; m[idx] = mi; v[idx] = vi;
      }
      tile[r][k] = out;
    }
    __syncthreads();
    TC* fwd = reinterpret_cast<TC*>(d.fwd);
    TC* tr = reinterpret_cast<TC*>(d.tr);
    for (int i = threadIdx.x; i < taps * 1024; i += 256) {
      const int tap = i >> 10, r = (i >> 5) & 31, c = i & 31;
      if (fwd && co0 + r < d.Cout && ci0 + c < d.Cin) fwd[((long)tap * d.Cout + co0 + r) * d.Cin + ci0 + c] = (TC)tile[r][c * taps + tap];
      if (tr && ci0 + r < d.Cin && co0 + c < d.Cout) tr[((long)(taps - 1 - tap) * d.Cin + ci0 + r) * d.Cout + co0 + c] = (TC)tile[c][r * taps + tap];
    }
    if constexpr (sizeof(TC) == 2) {
      // fragment-order copies (dx_pack_frag_major): out[chunk][tap][half][block][lane][8], one 16-byte vector per (tap, half, lane)
      if (d.frag_fwd || d.frag_tr) {
        typedef TC v8 __attribute__((ext_vector_type(8)));
        for (int i = threadIdx.x; i < 3 * 2 * 64; i += 256) {
          const int lane = i & 63, half = (i >> 6) & 1, tap = i >> 7;
          const int a = lane & 31, b0 = 16 * half + 8 * (lane >> 5);
          if (d.frag_fwd) {   // = fwd[tap][32 cb + a][32 kb + b0 + e]: chunk = kb, block = cb of Cout / 32
            v8 x;
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = (TC)tile[a][(b0 + e) * 3 + tap];
            *reinterpret_cast<v8*>(reinterpret_cast<TC*>(d.frag_fwd) + (((((long)kb * 3 + tap) * 2 + half) * (d.Cout >> 5) + cb) * 64 + lane) * 8) = x;
          }
          if (d.frag_tr) {    // = tr[tap][32 kb + a][32 cb + b0 + e] = W[32 cb + b0 + e][32 kb + a][2 - tap]: chunk = cb, block = kb of Cin / 32
            v8 x;
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = (TC)tile[b0 + e][a * 3 + (2 - tap)];
            *reinterpret_cast<v8*>(reinterpret_cast<TC*>(d.frag_tr) + (((((long)cb * 3 + tap) * 2 + half) * (d.Cin >> 5) + kb) * 64 + lane) * 8) = x;
          }
        }
      }
    }
  }
  if (gnorm_accum) {   // the gradient norm the trainer logs (clip_grad_norm_(inf), train.py:399), summed on the way
    gsq = block_sum_256(gsq, red);
    if (threadIdx.x == 0) atomicAdd(gnorm_accum, gsq);
  }
}

}  // namespace

extern "C" int dx_adam_pack_desc_size(void) { return (int)sizeof(AdamPackDesc); }
extern "C" int dx_adam_flat_desc_size(void) { return (int)sizeof(AdamFlatDesc); }
extern "C" int dx_adam_flat_block(void) { return FLAT_BLOCK; }

extern "C" int dx_adam_pack_step(float* p, const float* g, float* m, float* v, const void* bricks_dev, int n_weights, long total_bricks,
                                 const void* flats_dev, int n_flats, long total_flat_blocks, int out_dtype, float lr, float beta1,
                                 float beta2, float eps, float weight_decay, int step, float* grad_norm_sq_accum,
                                 const DxStepScalars* scalars, void* stream) {
  DX_REQUIRE(p && g && m && v && bricks_dev && n_weights > 0 && total_bricks > 0, DX_ERR_ARG, "dx_adam_pack_step: bad arguments");
  DX_REQUIRE((n_flats == 0) == (total_flat_blocks == 0) && (n_flats == 0 || flats_dev), DX_ERR_ARG, "dx_adam_pack_step: flat table");
  DX_REQUIRE(step >= 1 || scalars, DX_ERR_ARG, "dx_adam_pack_step: step count");
  DX_REQUIRE(total_bricks + total_flat_blocks < (1L << 31), DX_ERR_SHAPE, "dx_adam_pack_step: too many blocks");
  if (step < 1) step = 1;
  const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
  const AdamHyper h{lr, beta1, beta2, eps, weight_decay, (float)bc1, (float)sqrt(bc2)};
  const dim3 grid((unsigned)(total_bricks + total_flat_blocks));
  const AdamPackDesc* b = reinterpret_cast<const AdamPackDesc*>(bricks_dev);
  const AdamFlatDesc* f = reinterpret_cast<const AdamFlatDesc*>(flats_dev);
  if (out_dtype == DX_BF16)
    hipLaunchKernelGGL(adam_pack_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, p, g, m, v, b, n_weights, total_bricks, f, n_flats, h,
                       grad_norm_sq_accum, scalars);
  else if (out_dtype == DX_F32)
    hipLaunchKernelGGL(adam_pack_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, p, g, m, v, b, n_weights, total_bricks, f, n_flats, h,
                       grad_norm_sq_accum, scalars);
  else { dx_set_error("dx_adam_pack_step: bad out_dtype %d", out_dtype); return DX_ERR_DTYPE; }
  DX_LAUNCH_CHECK();
  return DX_OK;
}
