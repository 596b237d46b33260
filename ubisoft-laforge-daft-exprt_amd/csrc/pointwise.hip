// K6 / K7 / K8 / K9 / K10 / K14 -- the HBM-bound pointwise and tiny-GEMM pieces around the FFT blocks.
//   dx_scalar_embed_*  : sum of 1->C k=3 convs of scalar features (+ base tensor, + positional table, + mask)
//                        (energy/pitch embeddings model.py:402-414; duration/energy/pitch projections 618-628)
//   dx_embed_pos_*     : symbol embedding gather + positional table + mask (model.py:497-504)
//   dx_masked_mean_*   : sum over time / length (model.py:419)
//   dx_film_assemble_* : gamma = post_g * g_raw + 1, beta = post_b * b_raw, split per module/block (model.py:430-462)
//   dx_linear_small_*  : exact-fp32 nn.Linear for the small heads (FiLM projections 427-428, classifier 276-283,
//                        predictor projection 568, range projection 634) -- VALU, no operand rounding, so that the
//                        duration head keeps the reference's fp32 values for the integer-duration conversion.
// All kernels are fp32, coalesced along the channel axis (C = 128 -> one float2 per lane of a wave64 row, or one
// float per thread of a 128-thread row group).
#include "dx_common.h"

namespace {

constexpr int C128 = 128;

// ------------------------------------------------------------------ scalar-feature conv embedding
struct ScalarEmbedArgs {
  const float* base;          // (B, N, C) or null
  const float* feat[3];       // (B, N) each
  const float* w[3];          // (C, 1, taps) each
  const float* bias[3];       // (C) each
  int nfeat;
  const float* pos;           // (max_len, C) or null
  const int64_t* lengths;     // mask (rows n >= len -> 0) or null
  float* out;                 // (B, N, C)
  int N;
  long rows;
  int taps;                   // 3 ("same" padding, one halo row each side) or 1
};

__global__ __launch_bounds__(256) void scalar_embed_fwd_kernel(ScalarEmbedArgs a) {
  // 256 threads = 2 rows x 128 channels
  const int c = threadIdx.x & 127;
  const long row = (long)blockIdx.x * 2 + (threadIdx.x >> 7);
  if (row >= a.rows) return;
  const int b = (int)(row / a.N), n = (int)(row - (long)b * a.N);
  float v = 0.f;
  if (!a.lengths || n < (int)a.lengths[b]) {
    if (a.base) v = a.base[row * C128 + c];
    for (int f = 0; f < a.nfeat; ++f) {
      const float* ft = a.feat[f] + (long)b * a.N;
      if (a.taps == 3) {
        const float xm = n > 0 ? ft[n - 1] : 0.f, x0 = ft[n], xp = n + 1 < a.N ? ft[n + 1] : 0.f;
        const float* w = a.w[f] + c * 3;
        v += w[0] * xm + w[1] * x0 + w[2] * xp + a.bias[f][c];
      } else {
        v += a.w[f][c] * ft[n] + a.bias[f][c];
      }
    }
    if (a.pos) v += a.pos[(long)n * C128 + c];
  }
  a.out[row * C128 + c] = v;
}

struct ScalarEmbedBwdArgs {
  const float* dout;          // (B, N, C)
  const float* feat[3];
  int nfeat;
  const int64_t* lengths;     // the forward's mask or null
  float* dbase;               // (B, N, C) = dout * mask, or null
  float* dw[3];               // (C, 1, taps) accumulated
  float* dbias[3];            // (C) accumulated
  int N; int rows_per_block;
  int taps;
};

constexpr int SE_MAX_ROWS = 1024;
// grid (ceil(N / rows_per_block), B); 1024 threads = 8 row lanes x 128 channels: row lane q walks rows q, q + 8, ... of the
// slab, the eight partial sums meet in LDS, and the workgroup ends with ONE atomic per (feature, tap, channel) -- all
// workgroups hit the same 1024 addresses, so few fat workgroups (the 2048-workgroup launch spent ~100 us queueing atomics)
__global__ __launch_bounds__(1024) void scalar_embed_bwd_kernel(ScalarEmbedBwdArgs a) {
  __shared__ float red[8][3][4][C128];
  __shared__ float fs[3][SE_MAX_ROWS + 2];      // the slab's feature values with a one-row halo: one LDS read instead of three global loads per tap
  const int c = threadIdx.x & 127, q = threadIdx.x >> 7, b = blockIdx.y;
  const int n0 = blockIdx.x * a.rows_per_block;
  const int len = a.lengths ? (int)a.lengths[b] : a.N;
  const int n1 = min(min(a.N, n0 + a.rows_per_block), a.dbase ? a.N : len);
  for (int i = threadIdx.x; i < a.nfeat * (a.rows_per_block + 2); i += 1024) {
    const int f = i / (a.rows_per_block + 2), r = i - f * (a.rows_per_block + 2), n = n0 + r - 1;
    fs[f][r] = (n >= 0 && n < a.N) ? a.feat[f][(long)b * a.N + n] : 0.f;
  }
  __syncthreads();
  float aw[3][3], ab[3];
#pragma unroll
  for (int f = 0; f < 3; ++f) { ab[f] = 0.f; aw[f][0] = aw[f][1] = aw[f][2] = 0.f; }
#pragma unroll 4
  for (int n = n0 + q; n < n1; n += 8) {
    const long row = (long)b * a.N + n;
    const float g = n < len ? a.dout[row * C128 + c] : 0.f;
    if (a.dbase) a.dbase[row * C128 + c] = g;
    const int r = n - n0;
#pragma unroll
    for (int f = 0; f < 3; ++f) {
      if (f < a.nfeat) {
        const float xm = fs[f][r], x0 = fs[f][r + 1], xp = fs[f][r + 2];
        aw[f][0] += g * xm; aw[f][1] += g * x0; aw[f][2] += g * xp; ab[f] += g;
      }
    }
  }
#pragma unroll
  for (int f = 0; f < 3; ++f) { red[q][f][0][c] = aw[f][0]; red[q][f][1][c] = aw[f][1]; red[q][f][2][c] = aw[f][2]; red[q][f][3][c] = ab[f]; }
  __syncthreads();
  for (int idx = threadIdx.x; idx < a.nfeat * 4 * C128; idx += 1024) {
    const int f = idx / (4 * C128), k = (idx / C128) & 3, ch = idx & (C128 - 1);
    const float t = ((red[0][f][k][ch] + red[1][f][k][ch]) + (red[2][f][k][ch] + red[3][f][k][ch])) +
                    ((red[4][f][k][ch] + red[5][f][k][ch]) + (red[6][f][k][ch] + red[7][f][k][ch]));
    if (k == 3) atomicAdd(a.dbias[f] + ch, t);
    else if (a.taps == 3) atomicAdd(a.dw[f] + ch * 3 + k, t);
    else if (k == 1) atomicAdd(a.dw[f] + ch, t);                 // one tap: the centre sum
  }
}

// ------------------------------------------------------------------ embedding + positional table
__global__ __launch_bounds__(256) void embed_pos_fwd_kernel(const int64_t* __restrict__ ids, const float* __restrict__ table,
                                                            const float* __restrict__ pos, const int64_t* __restrict__ lengths,
                                                            float* __restrict__ out, int N, long rows) {
  const int c = threadIdx.x & 127;
  const long row = (long)blockIdx.x * 2 + (threadIdx.x >> 7);
  if (row >= rows) return;
  const int b = (int)(row / N), n = (int)(row - (long)b * N);
  float v = 0.f;
  if (n < (int)lengths[b]) v = table[ids[row] * C128 + c] + pos[(long)n * C128 + c];
  out[row * C128 + c] = v;
}
__global__ __launch_bounds__(256) void embed_pos_bwd_kernel(const int64_t* __restrict__ ids, const float* __restrict__ dout,
                                                            const int64_t* __restrict__ lengths, float* __restrict__ dtable,
                                                            int N, long rows) {
  const int c = threadIdx.x & 127;
  const long row = (long)blockIdx.x * 2 + (threadIdx.x >> 7);
  if (row >= rows) return;
  const int b = (int)(row / N), n = (int)(row - (long)b * N);
  if (n < (int)lengths[b]) atomicAdd(dtable + ids[row] * C128 + c, dout[row * C128 + c]);
}

// ------------------------------------------------------------------ out[b] = a[b] + table[ids[b]]  (speaker embedding add, model.py:423-424)
__global__ void gather_add_fwd_kernel(const float* __restrict__ a, const float* __restrict__ table, const int64_t* __restrict__ ids,
                                      float* __restrict__ out, int B, int C) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * C) return;
  const int b = i / C, c = i - b * C;
  out[i] = a[i] + table[ids[b] * C + c];
}
__global__ void gather_add_bwd_kernel(const float* __restrict__ dz, const int64_t* __restrict__ ids, float* __restrict__ dtable, int B, int C) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * C) return;
  const int b = i / C, c = i - b * C;
  atomicAdd(dtable + ids[b] * C + c, dz[i]);
}

// ------------------------------------------------------------------ masked mean over time
// grid (B); 1024 threads = 8 row lanes x 128 channels; fixed summation order (lane-strided rows, then the 8 lanes in LDS):
// the utterance embedding feeds every FiLM parameter, and with bf16 GEMM operands downstream a 1e-7 run-to-run wobble
// here (fp32 atomics in the first version) flips operand roundings and moves individual mel values by 0.1+
__global__ __launch_bounds__(1024) void masked_mean_fwd_kernel(const float* __restrict__ x, const int64_t* __restrict__ lengths,
                                                               float* __restrict__ out, int N) {
  __shared__ float red[8][C128];
  const int c = threadIdx.x & 127, q = threadIdx.x >> 7, b = blockIdx.x;
  const int len = (int)lengths[b], n1 = min(N, len);             // pads are zeros (masked upstream)
  const float* p = x + (long)b * N * C128 + c;
  float acc = 0.f;
#pragma unroll 4
  for (int n = q; n < n1; n += 8) acc += p[(long)n * C128];
  red[q][c] = acc;
  __syncthreads();
  if (q == 0) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) t += red[k][c];
    out[b * C128 + c] = len > 0 ? t / (float)len : 0.f;
  }
}
__global__ __launch_bounds__(256) void masked_mean_bwd_kernel(const float* __restrict__ dy, const int64_t* __restrict__ lengths,
                                                              float* __restrict__ dx, int N, long rows) {
  const int c = threadIdx.x & 127;
  const long row = (long)blockIdx.x * 2 + (threadIdx.x >> 7);
  if (row >= rows) return;
  const int b = (int)(row / N);
  dx[row * C128 + c] = dy[b * C128 + c] / (float)lengths[b];
}

// ------------------------------------------------------------------ FiLM assembly
// raw gammas / betas (B, W) with W = sum(nb_blocks_m * ch_m); film_m (B, nb_m, 2*ch_m); post (2, nblk_total) or null
struct FilmArgs {
  const float* g_raw; const float* b_raw; const float* post;
  float* film[3];
  int nb[3], ch[3];
  int B, W, nblk;
};
__global__ void film_assemble_fwd_kernel(FilmArgs a) {
  const long total = (long)a.B * a.W;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int b = (int)(i / a.W);
    int col = (int)(i - (long)b * a.W), m = 0, blk0 = 0;
    while (col >= a.nb[m] * a.ch[m]) { col -= a.nb[m] * a.ch[m]; blk0 += a.nb[m]; ++m; }
    const int blk = col / a.ch[m], c = col - blk * a.ch[m];
    const float pg = a.post ? a.post[blk0 + blk] : 1.f, pb = a.post ? a.post[a.nblk + blk0 + blk] : 1.f;
    float* dst = a.film[m] + ((long)b * a.nb[m] + blk) * 2 * a.ch[m];
    dst[c] = pg * a.g_raw[i] + 1.f;
    dst[a.ch[m] + c] = pb * a.b_raw[i];
  }
}
struct FilmBwdArgs {
  const float* g_raw; const float* b_raw; const float* post;
  const float* dfilm[3];
  float* dg_raw; float* db_raw; float* dpost;   // dpost accumulated (atomics)
  int nb[3], ch[3];
  int B, W, nblk;
};
__global__ void film_assemble_bwd_kernel(FilmBwdArgs a) {
  // (+ the post-multiplier gradient: dpost[0][blk] += sum_{b,c} dfilm_gamma * g_raw, dpost[1][blk] += sum_{b,c} dfilm_beta * b_raw.  A wave's 64
  //  consecutive columns lie in ONE FiLM block whenever the block widths are multiples of 64, as in every published config: one wave
  //  reduction + two atomics per wave; the separate one-workgroup-per-block reduction launch took 34 us for 120 k products)
  const long total = (long)a.B * a.W;
  for (long i0 = blockIdx.x * (long)blockDim.x; i0 < total; i0 += (long)gridDim.x * blockDim.x) {
    const long i = i0 + threadIdx.x;
    const bool live = i < total;
    float tg = 0.f, tb = 0.f;
    int key = -1;
    if (live) {
      const int b = (int)(i / a.W);
      int col = (int)(i - (long)b * a.W), m = 0, blk0 = 0;
      while (col >= a.nb[m] * a.ch[m]) { col -= a.nb[m] * a.ch[m]; blk0 += a.nb[m]; ++m; }
      const int blk = col / a.ch[m], c = col - blk * a.ch[m];
      const float* src = a.dfilm[m] + ((long)b * a.nb[m] + blk) * 2 * a.ch[m];
      const float dg = src[c], db = src[a.ch[m] + c];
      const float pg = a.post ? a.post[blk0 + blk] : 1.f, pb = a.post ? a.post[a.nblk + blk0 + blk] : 1.f;
      a.dg_raw[i] = dg * pg;
      a.db_raw[i] = db * pb;
      if (a.dpost) { tg = dg * a.g_raw[i]; tb = db * a.b_raw[i]; key = blk0 + blk; }
    }
    if (a.dpost) {
      const int key0 = __builtin_amdgcn_readfirstlane(key);
      if (__all(key == key0 || !live)) {
        const float sg = dx_wave_sum(tg), sb = dx_wave_sum(tb);
        if ((threadIdx.x & 63) == 0 && key0 >= 0) { atomicAdd(a.dpost + key0, sg); atomicAdd(a.dpost + a.nblk + key0, sb); }
      } else if (live) {
        atomicAdd(a.dpost + key, tg);
        atomicAdd(a.dpost + a.nblk + key, tb);
      }
    }
  }
}
// ------------------------------------------------------------------ exact-fp32 small linear
// y[m][o] = act(sum_k x[m][k] W[o][k] + bias[o]); rows m >= mask_len[b] (b = m / N) are written as zeros
__global__ __launch_bounds__(256) void linear_small_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                               const float* __restrict__ bias, float* __restrict__ y,
                                                               const int64_t* __restrict__ mask_len, int N, long M, int K,
                                                               int O, int relu) {
  const long total = M * O;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long m = i / O; const int o = (int)(i - m * O);
    float acc = 0.f;
    if (!mask_len || (int)(m % N) < (int)mask_len[m / N]) {
      const f32x4* xr = reinterpret_cast<const f32x4*>(x + m * K);
      const f32x4* wr = reinterpret_cast<const f32x4*>(w + (long)o * K);
      for (int k = 0; k < K / 4; ++k) {
        const f32x4 a = xr[k], bq = wr[k];
        acc = fmaf(a[0], bq[0], acc); acc = fmaf(a[1], bq[1], acc); acc = fmaf(a[2], bq[2], acc); acc = fmaf(a[3], bq[3], acc);
      }
      acc += bias ? bias[o] : 0.f;
      if (relu) acc = fmaxf(acc, 0.f);
    }
    y[i] = acc;
  }
}
// dx[m][k] = sum_o dyg[m][o] W[o][k],  dyg = dy * (y > 0 if relu) * (row valid)
__global__ __launch_bounds__(256) void linear_small_bwd_dx_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                                  const float* __restrict__ w, float* __restrict__ dx,
                                                                  const int64_t* __restrict__ mask_len, int N, long M, int K,
                                                                  int O, int relu, float scale, int o_chunk) {
  // gridDim.y splits the output-feature loop (FiLM projections: O = 1280 for only M*K = 6144 threads); partial
  // sums are combined with atomics into a zeroed dx when there is more than one chunk
  const long total = M * K;
  const int o0 = blockIdx.y * o_chunk, o1 = min(O, o0 + o_chunk);
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long m = i / K; const int k = (int)(i - m * K);
    float acc = 0.f;
    if (!mask_len || (int)(m % N) < (int)mask_len[m / N]) {
#pragma unroll 8
      for (int o = o0; o < o1; ++o) {
        float g = dy[m * O + o];
        if (relu && !(y[m * O + o] > 0.f)) g = 0.f;
        acc = fmaf(g, w[(long)o * K + k], acc);
      }
    }
    if (gridDim.y == 1) dx[i] = acc * scale; else atomicAdd(dx + i, acc * scale);
  }
}
// dW[o][k] += sum_m dyg[m][o] x[m][k]; db[o] += sum_m dyg[m][o].  grid (ceil(O*K/256), mchunks)
__global__ __launch_bounds__(256) void linear_small_bwd_dw_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                                  const float* __restrict__ x, float* __restrict__ dw,
                                                                  float* __restrict__ db, const int64_t* __restrict__ mask_len,
                                                                  int N, long M, int K, int O, int relu, int rows_per_chunk) {
  const long idx = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (idx >= (long)O * K) return;
  const int o = (int)(idx / K), k = (int)(idx - (long)o * K);
  const long m0 = (long)blockIdx.y * rows_per_chunk, m1 = min(M, m0 + rows_per_chunk);
  // the row -> (utterance, position) split is carried along instead of a 64-bit division per row, and four independent partial sums
  // keep four rows of loads in flight (the first version's chain of one division + three dependent loads per row: 38 us for 8 MB)
  int b = mask_len ? (int)(m0 / N) : 0, n = mask_len ? (int)(m0 - (long)b * N) : 0;
  int len = mask_len ? (int)mask_len[b] : 0;
  float a4[4] = {0.f, 0.f, 0.f, 0.f}, b4[4] = {0.f, 0.f, 0.f, 0.f};
  long m = m0;
  for (; m + 3 < m1; m += 4) {
    float g[4], xv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      bool ok = true;
      if (mask_len) {
        ok = n < len;
        if (++n == N) { n = 0; ++b; len = (m + u + 1 < M) ? (int)mask_len[b] : 0; }
      }
      g[u] = ok ? dy[(m + u) * O + o] : 0.f;
      if (relu && !(y[(m + u) * O + o] > 0.f)) g[u] = 0.f;
      xv[u] = x[(m + u) * K + k];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) { a4[u] = fmaf(g[u], xv[u], a4[u]); b4[u] += g[u]; }
  }
  for (; m < m1; ++m) {
    bool ok = true;
    if (mask_len) {
      ok = n < len;
      if (++n == N) { n = 0; ++b; len = (m + 1 < M) ? (int)mask_len[b] : 0; }
    }
    float g = ok ? dy[m * O + o] : 0.f;
    if (relu && !(y[m * O + o] > 0.f)) g = 0.f;
    a4[0] = fmaf(g, x[m * K + k], a4[0]);
    b4[0] += g;
  }
  const float acc = (a4[0] + a4[1]) + (a4[2] + a4[3]), accb = (b4[0] + b4[1]) + (b4[2] + b4[3]);
  atomicAdd(dw + idx, acc);
  if (db && k == 0) atomicAdd(db + o, accb);
}

// ------------------------------------------------------------------ misc
__global__ void add_inplace_kernel(float* __restrict__ dst, const float* __restrict__ src, long n) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) dst[i] += src[i];
}
// column sums of a (rows, C) matrix of type T accumulated into out (C) -- conv / linear bias gradients
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ x, float* __restrict__ out, long rows, int C, int rows_per_block) {
  const long r0 = (long)blockIdx.y * rows_per_block, r1 = min(rows, r0 + rows_per_block);
  for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < C; c += gridDim.x * blockDim.x) {
    float acc = 0.f;
    for (long r = r0; r < r1; ++r) acc += (float)x[r * C + c];
    atomicAdd(out + c, acc);
  }
}

inline int grid_for(long total, int block = 256, int cap = 4096) {
  long g = (total + block - 1) / block;
  return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}


// ---- the small heads with FEW ROWS (M = batch size: classifier, FiLM projections): LDS-tiled fp32 weight gradient.  The generic
// kernels above give every output element one thread that walks the rows serially -- 48 dependent global round trips for a
// 48-utterance batch, 20-28 us per launch for a few kFLOP (profiles/r02_step_timeline.txt); here a 16 x 16 thread block
// stages 32-row tiles of both operands in LDS and every thread owns a 4 x 4 block of outputs.
constexpr int LT = 64, LR = 32;
// dW[o][k] += sum_m G[m][o] x[m][k], db[o] += sum_m G[m][o], G = dy * relu'(y)
__global__ __launch_bounds__(256) void linear_rows_bwd_dw_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                                 const float* __restrict__ x, float* __restrict__ dw,
                                                                 float* __restrict__ db, long M, int K, int O, int relu) {
  __shared__ float Gs[LR][LT + 4], Xs[LR][LT + 4];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int k0 = blockIdx.x * LT, o0 = blockIdx.y * LT;
  float acc[4][4] = {}, accb[4] = {};
  for (long m0 = 0; m0 < M; m0 += LR) {
    for (int i = threadIdx.x; i < LR * LT; i += 256) {
      const int r = i / LT, c = i - r * LT;
      const long m = m0 + r;
      float g = 0.f, xv = 0.f;
      if (m < M) {
        if (o0 + c < O) { g = dy[m * O + o0 + c]; if (relu && !(y[m * O + o0 + c] > 0.f)) g = 0.f; }
        if (k0 + c < K) xv = x[m * K + k0 + c];
      }
      Gs[r][c] = g; Xs[r][c] = xv;
    }
    __syncthreads();
#pragma unroll 8
    for (int r = 0; r < LR; ++r) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(&Gs[r][ty * 4]), b = *reinterpret_cast<const f32x4*>(&Xs[r][tx * 4]);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        accb[i] += a[i];
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int o = o0 + ty * 4 + i;
    if (o >= O) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (k0 + tx * 4 + j < K) atomicAdd(dw + (long)o * K + k0 + tx * 4 + j, acc[i][j]);
    if (db && blockIdx.x == 0 && tx == 0) atomicAdd(db + o, accb[i]);
  }
}
// ---- layout helpers that used to be ATen copies on the step path ---------------------------------------------------------
// (B, R, C) -> (B, C, R) through a 32 x 33 LDS tile: coalesced on both sides.  The reference hands the mel batch over as
// (B, n_mel, T) (model.py:744) while every kernel here wants channel-last rows; the loss gradient of the autograd bridge
// comes back the same way.
template <typename TY>
__global__ __launch_bounds__(256) void transpose_last2_kernel(const float* __restrict__ x, TY* __restrict__ y, int R, int C) {
  __shared__ float tile[32][33];
  const int b = blockIdx.z, r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const float* xb = x + (size_t)b * R * C;
  TY* yb = y + (size_t)b * R * C;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = r0 + ty + 8 * i, c = c0 + tx;
    if (r < R && c < C) tile[ty + 8 * i][tx] = xb[(size_t)r * C + c];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = c0 + ty + 8 * i, r = r0 + tx;
    if (r < R && c < C) yb[(size_t)c * R + r] = (TY)tile[tx][ty + 8 * i];
  }
}
// K interleaved planes <-> K separate planes: y (M, K) <-> a_k (M)   (K <= 4: the three prosody heads share one projection)
struct PlaneArgs { float* planes[4]; };
__global__ __launch_bounds__(256) void unstack_kernel(const float* __restrict__ y, PlaneArgs a, long M, int K) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < M * K; i += (long)gridDim.x * 256) a.planes[i % K][i / K] = y[i];
}
__global__ __launch_bounds__(256) void stack_kernel(float* __restrict__ y, PlaneArgs a, long M, int K) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < M * K; i += (long)gridDim.x * 256) y[i] = a.planes[i % K][i / K];
}
}  // namespace

extern "C" int dx_scalar_embed_fwd(const float* base, const float* const* feats, const float* const* ws,
                                   const float* const* biases, int nfeat, const float* pos_table,
                                   const int64_t* lengths, float* out, int B, int N, int C, int taps, void* stream) {
  DX_REQUIRE(out && nfeat >= 0 && nfeat <= 3, DX_ERR_ARG, "dx_scalar_embed_fwd: bad arguments");
  DX_REQUIRE(taps == 1 || taps == 3, DX_ERR_UNSUPPORTED, "dx_scalar_embed_fwd: taps=%d (only 1 and 3)", taps);
  DX_REQUIRE(C == C128, DX_ERR_UNSUPPORTED, "dx_scalar_embed_fwd: C=%d (only 128)", C);
  DX_REQUIRE(B > 0 && N > 0, DX_ERR_SHAPE, "dx_scalar_embed_fwd: empty shape");
  ScalarEmbedArgs a{};
  a.base = base; a.nfeat = nfeat; a.pos = pos_table; a.lengths = lengths; a.out = out; a.N = N; a.rows = (long)B * N; a.taps = taps;
  for (int f = 0; f < nfeat; ++f) { a.feat[f] = feats[f]; a.w[f] = ws[f]; a.bias[f] = biases[f]; }
  hipLaunchKernelGGL(scalar_embed_fwd_kernel, dim3((unsigned)((a.rows + 1) / 2)), dim3(256), 0, (hipStream_t)stream, a);
  DX_LAUNCH_CHECK();
  return DX_OK;
}

extern "C" int dx_scalar_embed_bwd(const float* dout, const float* const* feats, int nfeat, const int64_t* lengths,
                                   float* dbase, float* const* dws, float* const* dbiases, int B, int N, int C, int taps,
                                   void* stream) {
  DX_REQUIRE(dout && nfeat >= 0 && nfeat <= 3, DX_ERR_ARG, "dx_scalar_embed_bwd: bad arguments");
  DX_REQUIRE(taps == 1 || taps == 3, DX_ERR_UNSUPPORTED, "dx_scalar_embed_bwd: taps=%d (only 1 and 3)", taps);
  DX_REQUIRE(C == C128, DX_ERR_UNSUPPORTED, "dx_scalar_embed_bwd: C=%d (only 128)", C);
  ScalarEmbedBwdArgs a{};
  a.dout = dout; a.nfeat = nfeat; a.lengths = lengths; a.dbase = dbase; a.N = N; a.taps = taps;
  // every workgroup ends with 512-1024 atomics on the same few addresses: few, fat workgroups
  // (same-address atomics serialise at ~50 ns each: ~256 workgroups keep that tail under the time the rows take to stream)
  int rpb = 64;
  while (rpb < 1024 && (long)dx_cdiv(N, rpb) * B > 256) rpb *= 2;
  a.rows_per_block = rpb;
  for (int f = 0; f < nfeat; ++f) { a.feat[f] = feats[f]; a.dw[f] = dws[f]; a.dbias[f] = dbiases[f]; }
  hipLaunchKernelGGL(scalar_embed_bwd_kernel, dim3(dx_cdiv(N, rpb), B), dim3(1024), 0, (hipStream_t)stream, a);
  DX_LAUNCH_CHECK();
  return DX_OK;
}

extern "C" int dx_embed_pos_fwd(const int64_t* ids, const float* table, const float* pos_table, const int64_t* lengths,
                                float* out, int B, int N, int C, void* stream) {
  DX_REQUIRE(ids && table && pos_table && lengths && out, DX_ERR_ARG, "dx_embed_pos_fwd: null pointer");
  DX_REQUIRE(C == C128, DX_ERR_UNSUPPORTED, "dx_embed_pos_fwd: C=%d (only 128)", C);
  const long rows = (long)B * N;
  hipLaunchKernelGGL(embed_pos_fwd_kernel, dim3((unsigned)((rows + 1) / 2)), dim3(256), 0, (hipStream_t)stream, ids, table,
                     pos_table, lengths, out, N, rows);
  DX_LAUNCH_CHECK();
  return DX_OK;
}
extern "C" int dx_embed_pos_bwd(const int64_t* ids, const float* dout, const int64_t* lengths, float* dtable, int B,
                                int N, int C, void* stream) {
  DX_REQUIRE(ids && dout && lengths && dtable, DX_ERR_ARG, "dx_embed_pos_bwd: null pointer");
  DX_REQUIRE(C == C128, DX_ERR_UNSUPPORTED, "dx_embed_pos_bwd: C=%d (only 128)", C);
  const long rows = (long)B * N;
  hipLaunchKernelGGL(embed_pos_bwd_kernel, dim3((unsigned)((rows + 1) / 2)), dim3(256), 0, (hipStream_t)stream, ids, dout,
                     lengths, dtable, N, rows);
  DX_LAUNCH_CHECK();
  return DX_OK;
}

extern "C" int dx_masked_mean_fwd(const float* x, const int64_t* lengths, float* out, int B, int N, int C, void* stream) {
  DX_REQUIRE(x && lengths && out, DX_ERR_ARG, "dx_masked_mean_fwd: null pointer");
  DX_REQUIRE(C == C128, DX_ERR_UNSUPPORTED, "dx_masked_mean_fwd: C=%d (only 128)", C);
  hipLaunchKernelGGL(masked_mean_fwd_kernel, dim3(B), dim3(1024), 0, (hipStream_t)stream, x, lengths, out, N);
  DX_LAUNCH_CHECK();
  return DX_OK;
}
extern "C" int dx_masked_mean_bwd(const float* dy, const int64_t* lengths, float* dx, int B, int N, int C, void* stream) {
  DX_REQUIRE(dy && lengths && dx, DX_ERR_ARG, "dx_masked_mean_bwd: null pointer");
  DX_REQUIRE(C == C128, DX_ERR_UNSUPPORTED, "dx_masked_mean_bwd: C=%d (only 128)", C);
  const long rows = (long)B * N;
  hipLaunchKernelGGL(masked_mean_bwd_kernel, dim3((unsigned)((rows + 1) / 2)), dim3(256), 0, (hipStream_t)stream, dy, lengths,
                     dx, N, rows);
  DX_LAUNCH_CHECK();
  return DX_OK;
}

extern "C" int dx_film_assemble_fwd(const float* g_raw, const float* b_raw, const float* post, float* film_enc,
                                    float* film_pp, float* film_dec, const int* nb, const int* ch, int B, void* stream) {
  DX_REQUIRE(g_raw && b_raw && film_enc && film_pp && film_dec && nb && ch, DX_ERR_ARG, "dx_film_assemble_fwd: null pointer");
  FilmArgs a{};
  a.g_raw = g_raw; a.b_raw = b_raw; a.post = post; a.film[0] = film_enc; a.film[1] = film_pp; a.film[2] = film_dec; a.B = B;
  for (int m = 0; m < 3; ++m) { a.nb[m] = nb[m]; a.ch[m] = ch[m]; a.W += nb[m] * ch[m]; a.nblk += nb[m]; }
  hipLaunchKernelGGL(film_assemble_fwd_kernel, dim3(grid_for((long)B * a.W)), dim3(256), 0, (hipStream_t)stream, a);
  DX_LAUNCH_CHECK();
  return DX_OK;
}
extern "C" int dx_film_assemble_bwd(const float* g_raw, const float* b_raw, const float* post, const float* dfilm_enc,
                                    const float* dfilm_pp, const float* dfilm_dec, float* dg_raw, float* db_raw,
                                    float* dpost, const int* nb, const int* ch, int B, void* stream) {
  DX_REQUIRE(g_raw && b_raw && dfilm_enc && dfilm_pp && dfilm_dec && dg_raw && db_raw, DX_ERR_ARG, "dx_film_assemble_bwd: null pointer");
  FilmBwdArgs a{};
  a.g_raw = g_raw; a.b_raw = b_raw; a.post = post; a.dfilm[0] = dfilm_enc; a.dfilm[1] = dfilm_pp; a.dfilm[2] = dfilm_dec;
  a.dg_raw = dg_raw; a.db_raw = db_raw; a.dpost = post ? dpost : nullptr; a.B = B;
  for (int m = 0; m < 3; ++m) { a.nb[m] = nb[m]; a.ch[m] = ch[m]; a.W += nb[m] * ch[m]; a.nblk += nb[m]; }
  hipLaunchKernelGGL(film_assemble_bwd_kernel, dim3(grid_for((long)B * a.W)), dim3(256), 0, (hipStream_t)stream, a);
  DX_LAUNCH_CHECK();
  return DX_OK;
}

extern "C" int dx_linear_small_fwd(const float* x, const float* w, const float* bias, float* y,
                                   const int64_t* mask_lengths, int N, long M, int K, int O, int relu, void* stream) {
  DX_REQUIRE(x && w && y, DX_ERR_ARG, "dx_linear_small_fwd: null pointer");
  DX_REQUIRE(M > 0 && K > 0 && O > 0 && K % 4 == 0, DX_ERR_SHAPE, "dx_linear_small_fwd: bad shape M=%ld K=%d O=%d", M, K, O);
  hipLaunchKernelGGL(linear_small_fwd_kernel, dim3(grid_for(M * O)), dim3(256), 0, (hipStream_t)stream, x, w, bias, y,
                     mask_lengths, N, M, K, O, relu);
  DX_LAUNCH_CHECK();
  return DX_OK;
}
extern "C" int dx_linear_small_bwd(const float* dy, const float* y, const float* x, const float* w, float* dx, float dx_scale,
                                   float* dw, float* db, const int64_t* mask_lengths, int N, long M, int K, int O,
                                   int relu, void* stream) {
  DX_REQUIRE(dy && x && w && dw, DX_ERR_ARG, "dx_linear_small_bwd: null pointer");
  DX_REQUIRE(!relu || y, DX_ERR_ARG, "dx_linear_small_bwd: relu needs the forward output y");
  hipStream_t s = (hipStream_t)stream;
  const bool few_rows = !mask_lengths && M <= 512;   // M = batch size: LDS-tiled weight-gradient kernel
  if (dx) {
    // few outputs (M * K small): short chains of 16 output features per thread, partial sums meet in atomics (a thread that walks
    // 64+ features is a chain of dependent global round trips: 25 us for a 48 x 128 x 128 layer)
    const int chunks = (O > 16 && M * K < (1L << 18)) ? dx_cdiv(O, 16) : 1;
    if (chunks > 1) { if (int rc = dx_fill_zero(dx, (size_t)M * K * sizeof(float), s)) return rc; }
    hipLaunchKernelGGL(linear_small_bwd_dx_kernel, dim3(grid_for(M * K), chunks), dim3(256), 0, s, dy, y, w, dx, mask_lengths, N, M, K, O,
                       relu, dx_scale, dx_cdiv(O, chunks));
  }
  if (few_rows) {
    hipLaunchKernelGGL(linear_rows_bwd_dw_kernel, dim3(dx_cdiv(K, LT), dx_cdiv(O, LT)), dim3(256), 0, s, dy, y, x, dw, db, M, K, O, relu);
    DX_LAUNCH_CHECK();
    return DX_OK;
  }
  int rpc = 64;
  while (rpc < 4096 && (M + rpc - 1) / rpc > 256) rpc *= 2;
  dim3 grid(dx_cdiv(O * K, 256), (unsigned)((M + rpc - 1) / rpc));
  hipLaunchKernelGGL(linear_small_bwd_dw_kernel, grid, dim3(256), 0, s, dy, y, x, dw, db, mask_lengths, N, M, K, O, relu, rpc);
  DX_LAUNCH_CHECK();
  return DX_OK;
}

extern "C" int dx_add_inplace(float* dst, const float* src, long n, void* stream) {
  DX_REQUIRE(dst && src && n >= 0, DX_ERR_ARG, "dx_add_inplace: bad arguments");
  if (n == 0) return DX_OK;
  hipLaunchKernelGGL(add_inplace_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, dst, src, n);
  DX_LAUNCH_CHECK();
  return DX_OK;
}

extern "C" int dx_colsum(const void* x, int dtype, float* out, long rows, int C, void* stream) {
  DX_REQUIRE(x && out && rows > 0 && C > 0, DX_ERR_ARG, "dx_colsum: bad arguments");
  int rpb = 64;
  while (rpb < 8192 && (rows + rpb - 1) / rpb > 512) rpb *= 2;
  dim3 grid(dx_cdiv(C, 256), (unsigned)((rows + rpb - 1) / rpb));
  if (dtype == DX_F32) hipLaunchKernelGGL(colsum_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)x, out, rows, C, rpb);
  else if (dtype == DX_BF16) hipLaunchKernelGGL(colsum_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, out, rows, C, rpb);
  else { dx_set_error("dx_colsum: bad dtype %d", dtype); return DX_ERR_DTYPE; }
  DX_LAUNCH_CHECK();
  return DX_OK;
}

extern "C" int dx_gather_add_fwd(const float* a, const float* table, const int64_t* ids, float* out, int B, int C, void* stream) {
  DX_REQUIRE(a && table && ids && out && B > 0 && C > 0, DX_ERR_ARG, "dx_gather_add_fwd: bad arguments");
  hipLaunchKernelGGL(gather_add_fwd_kernel, dim3(dx_cdiv(B * C, 256)), dim3(256), 0, (hipStream_t)stream, a, table, ids, out, B, C);
  DX_LAUNCH_CHECK();
  return DX_OK;
}
extern "C" int dx_gather_add_bwd(const float* dz, const int64_t* ids, float* dtable, int B, int C, void* stream) {
  DX_REQUIRE(dz && ids && dtable && B > 0 && C > 0, DX_ERR_ARG, "dx_gather_add_bwd: bad arguments");
  hipLaunchKernelGGL(gather_add_bwd_kernel, dim3(dx_cdiv(B * C, 256)), dim3(256), 0, (hipStream_t)stream, dz, ids, dtable, B, C);
  DX_LAUNCH_CHECK();
  return DX_OK;
}

extern "C" int dx_transpose_last2(const float* x, void* y, int y_dtype, int B, int R, int C, void* stream) {
  DX_REQUIRE(x && y && B > 0 && R > 0 && C > 0, DX_ERR_ARG, "dx_transpose_last2: bad arguments");
  DX_REQUIRE(y_dtype == DX_F32 || y_dtype == DX_BF16, DX_ERR_DTYPE, "dx_transpose_last2: y dtype %d", y_dtype);
  const dim3 grid(dx_cdiv(C, 32), dx_cdiv(R, 32), B);
  if (y_dtype == DX_F32) hipLaunchKernelGGL(transpose_last2_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, x, static_cast<float*>(y), R, C);
  else hipLaunchKernelGGL(transpose_last2_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, x, static_cast<bf16_t*>(y), R, C);
  DX_LAUNCH_CHECK();
  return DX_OK;
}

extern "C" int dx_unstack(const float* y, float* const* planes, long M, int K, void* stream) {
  DX_REQUIRE(y && planes && M > 0 && K > 0 && K <= 4, DX_ERR_ARG, "dx_unstack: bad arguments (K <= 4)");
  PlaneArgs a{};
  for (int k = 0; k < K; ++k) { DX_REQUIRE(planes[k], DX_ERR_ARG, "dx_unstack: null plane"); a.planes[k] = planes[k]; }
  hipLaunchKernelGGL(unstack_kernel, dim3(grid_for(M * K)), dim3(256), 0, (hipStream_t)stream, y, a, M, K);
  DX_LAUNCH_CHECK();
  return DX_OK;
}

extern "C" int dx_stack(float* y, const float* const* planes, long M, int K, void* stream) {
  DX_REQUIRE(y && planes && M > 0 && K > 0 && K <= 4, DX_ERR_ARG, "dx_stack: bad arguments (K <= 4)");
  PlaneArgs a{};
  for (int k = 0; k < K; ++k) { DX_REQUIRE(planes[k], DX_ERR_ARG, "dx_stack: null plane"); a.planes[k] = const_cast<float*>(planes[k]); }
  hipLaunchKernelGGL(stack_kernel, dim3(grid_for(M * K)), dim3(256), 0, (hipStream_t)stream, y, a, M, K);
  DX_LAUNCH_CHECK();
  return DX_OK;
}

// zero fill with 16-byte stores (the head up to the first 16-byte boundary and the tail byte by byte from workgroup 0); a kernel of
// this library rather than hipMemsetAsync so that every dispatch of a step is one of ours (and a graph capture records a kernel node)
__global__ __launch_bounds__(256) void fill_zero_kernel(unsigned char* __restrict__ p, size_t bytes) {
  size_t head = (16 - ((uintptr_t)p & 15)) & 15;
  if (head > bytes) head = bytes;
  const size_t nvec = (bytes - head) / 16;
  uint4* v = reinterpret_cast<uint4*>(p + head);
  const uint4 z = make_uint4(0u, 0u, 0u, 0u);
  for (size_t i = blockIdx.x * 256UL + threadIdx.x; i < nvec; i += gridDim.x * 256UL) v[i] = z;
  if (blockIdx.x == 0) {
    for (size_t i = threadIdx.x; i < head; i += 256) p[i] = 0;
    for (size_t i = head + nvec * 16 + threadIdx.x; i < bytes; i += 256) p[i] = 0;
  }
}

__global__ void anchor_kernel() {}
extern "C" int dx_anchor(void* stream) {
  hipLaunchKernelGGL(anchor_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream);
  DX_LAUNCH_CHECK();
  return DX_OK;
}

// one wave that keeps its hardware queue occupied for `microseconds` (100 MHz constant clock): the probe behind
// daft_exprt.streams.runs_beside -- "do these two HIP streams sit on different hardware queues?"
__global__ void spin_kernel(long ticks) {
  const long t0 = wall_clock64();
  while ((long)wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}
extern "C" int dx_spin(int microseconds, void* stream) {
  DX_REQUIRE(microseconds >= 0 && microseconds <= 100000, DX_ERR_ARG, "dx_spin: 0 .. 100000 us");
  hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (long)microseconds * 100);
  DX_LAUNCH_CHECK();
  return DX_OK;
}

extern "C" int dx_fill_zero(void* p, size_t bytes, void* stream) {
  DX_REQUIRE(p || bytes == 0, DX_ERR_ARG, "dx_fill_zero: null pointer");
  if (bytes == 0) return DX_OK;
  size_t blocks = (bytes / 16 + 255) / 256;
  blocks = blocks < 1 ? 1 : (blocks > 2048 ? 2048 : blocks);
  hipLaunchKernelGGL(fill_zero_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, static_cast<unsigned char*>(p), bytes);
  DX_LAUNCH_CHECK();
  return DX_OK;
}
