// The two per-utterance heads that hang off the prosody embedding (B rows of 128 floats), each as ONE forward and TWO backward launches.
//   FiLM head (model.py:419-462): z = emb + spk_embedding[id];  g_raw = z Wg^T + bg,  b_raw = z Wb^T + bb  (W = 1280 columns each);
//                                 gamma = post_g[blk] * g_raw + 1,  beta = post_b[blk] * b_raw, split per (module, block).
//   speaker classifier (model.py:27-54, 276-292): gradient reversal -> Linear 128 -> 128, ReLU, Linear 128 -> 128, ReLU,
//                                 Linear 128 -> n_speakers - 1.
// As separate launches (gather_add, 2 x linear_small, film_assemble; 3 x linear_small; and their backward counterparts with their
// fills and adds) these were ~10 dispatches forward and ~22 backward of 5-15 us each -- latency chains of a few dependent global round
// trips with the chip idle: 75 + 155 us per training step.  Exact fp32 arithmetic; the forward keeps the summation order of
// linear_small_fwd_kernel (k ascending, bias last), so its outputs are bit-identical to the separate launches.
#include "dx_common.h"

namespace {

constexpr int HD = 128;          // hidden_embed_dim of the prosody / phoneme encoders (model.DaftExprt rejects anything else)

struct FilmLayout { int nb[3], ch[3]; int W, nblk; };

__device__ __forceinline__ void film_col(const FilmLayout& L, int col, int& m, int& blk0, int& blk, int& c) {
  m = 0; blk0 = 0;
  while (col >= L.nb[m] * L.ch[m]) { col -= L.nb[m] * L.ch[m]; blk0 += L.nb[m]; ++m; }
  blk = col / L.ch[m];
  c = col - blk * L.ch[m];
}

__device__ __forceinline__ float dot128(const float* __restrict__ z_lds, const float* __restrict__ wrow, float bias) {
  float acc = 0.f;
  const f32x4* wr = reinterpret_cast<const f32x4*>(wrow);
#pragma unroll 8
  for (int k = 0; k < HD / 4; ++k) {
    const f32x4 w = wr[k];
    acc = fmaf(z_lds[4 * k], w[0], acc); acc = fmaf(z_lds[4 * k + 1], w[1], acc);
    acc = fmaf(z_lds[4 * k + 2], w[2], acc); acc = fmaf(z_lds[4 * k + 3], w[3], acc);
  }
  return acc + bias;
}

// ---- FiLM head forward: grid (ceil(W / 256), B)
struct FilmFwdArgs {
  const float* emb; const float* spk; const int64_t* ids; const float* wg; const float* bg; const float* wb; const float* bb; const float* post;
  float* z; float* g_raw; float* b_raw; float* film[3];
  FilmLayout L;
};
__global__ __launch_bounds__(256) void film_head_fwd_kernel(FilmFwdArgs a) {
  __shared__ float zs[HD];
  const int b = blockIdx.y, col = blockIdx.x * 256 + threadIdx.x;
  if (threadIdx.x < HD) {
    const float v = a.emb[(long)b * HD + threadIdx.x] + a.spk[a.ids[b] * HD + threadIdx.x];
    zs[threadIdx.x] = v;
    if (blockIdx.x == 0) a.z[(long)b * HD + threadIdx.x] = v;
  }
  __syncthreads();
  if (col >= a.L.W) return;
  const float g = dot128(zs, a.wg + (long)col * HD, a.bg[col]), bt = dot128(zs, a.wb + (long)col * HD, a.bb[col]);
  a.g_raw[(long)b * a.L.W + col] = g;
  a.b_raw[(long)b * a.L.W + col] = bt;
  int m, blk0, blk, c;
  film_col(a.L, col, m, blk0, blk, c);
  const float pg = a.post ? a.post[blk0 + blk] : 1.f, pb = a.post ? a.post[a.L.nblk + blk0 + blk] : 1.f;
  float* dst = a.film[m] + ((long)b * a.L.nb[m] + blk) * 2 * a.L.ch[m];
  dst[c] = pg * g + 1.f;
  dst[a.L.ch[m] + c] = pb * bt;
}

// ---- FiLM head backward, data side: grid (ceil(W / 128), B), 128 threads.  A block owns 128 columns of one utterance: thread t first
// turns dfilm into (dg_raw, db_raw) of column c0 + t (stored for the weight-gradient launch; the post-multiplier gradient leaves as one
// wave reduction + two atomics per wave), then thread k adds the block's share of dz[k] = sum_col dg_raw Wg[col][k] + db_raw Wb[col][k]
// to d_emb[b][k] and to the speaker-embedding gradient row (10 blocks per utterance meet in fp32 atomics).
struct FilmBwdDxArgs {
  const float* g_raw; const float* b_raw; const float* post; const int64_t* ids; const float* wg; const float* wb; const float* dfilm[3];
  float* dg_raw; float* db_raw; float* d_emb; float* d_spk; float* dpost;
  FilmLayout L;
};
__global__ __launch_bounds__(128) void film_head_bwd_dx_kernel(FilmBwdDxArgs a) {
  __shared__ float gs[128], bs[128];
  const int b = blockIdx.y, c0 = blockIdx.x * 128, col = c0 + threadIdx.x;
  float dgr = 0.f, dbr = 0.f, tg = 0.f, tb = 0.f;
  int key = -1;
  if (col < a.L.W) {
    int m, blk0, blk, c;
    film_col(a.L, col, m, blk0, blk, c);
    const float* src = a.dfilm[m] + ((long)b * a.L.nb[m] + blk) * 2 * a.L.ch[m];
    const float dg = src[c], db = src[a.L.ch[m] + c];
    const float pg = a.post ? a.post[blk0 + blk] : 1.f, pb = a.post ? a.post[a.L.nblk + blk0 + blk] : 1.f;
    dgr = dg * pg; dbr = db * pb;
    a.dg_raw[(long)b * a.L.W + col] = dgr;
    a.db_raw[(long)b * a.L.W + col] = dbr;
    if (a.dpost) { tg = dg * a.g_raw[(long)b * a.L.W + col]; tb = db * a.b_raw[(long)b * a.L.W + col]; key = blk0 + blk; }
  }
  gs[threadIdx.x] = dgr; bs[threadIdx.x] = dbr;
  if (a.dpost) {   // a wave's 64 columns lie in one FiLM block whenever the block widths are multiples of 64 (every published config)
    const int key0 = __builtin_amdgcn_readfirstlane(key);
    if (__all(key == key0)) {
      const float sg = dx_wave_sum(tg), sb = dx_wave_sum(tb);
      if ((threadIdx.x & 63) == 0 && key0 >= 0) { atomicAdd(a.dpost + key0, sg); atomicAdd(a.dpost + a.L.nblk + key0, sb); }
    } else if (key >= 0) {
      atomicAdd(a.dpost + key, tg);
      atomicAdd(a.dpost + a.L.nblk + key, tb);
    }
  }
  __syncthreads();
  const int k = threadIdx.x, ncol = min(128, a.L.W - c0);
  float acc = 0.f;
#pragma unroll 4
  for (int j = 0; j < ncol; ++j) {
    acc = fmaf(gs[j], a.wg[(long)(c0 + j) * HD + k], acc);
    acc = fmaf(bs[j], a.wb[(long)(c0 + j) * HD + k], acc);
  }
  atomicAdd(a.d_emb + (long)b * HD + k, acc);
  atomicAdd(a.d_spk + a.ids[b] * HD + k, acc);
}

// ---- weight gradients of a set of "few rows" linear layers in ONE launch: dW[o][k] += sum_b G[b][o] X[b][k], db[o] += sum_b G[b][o].
// Every (o, k) belongs to one thread, which walks the B rows: deterministic, no atomics.  Layer table in the kernel arguments.
constexpr int DWL_MAX = 4;
struct DwLayers {
  const float* g[DWL_MAX]; const float* x[DWL_MAX]; float* dw[DWL_MAX]; float* db[DWL_MAX];
  int O[DWL_MAX], ldg[DWL_MAX]; int begin[DWL_MAX + 1]; int n, B;
};
__global__ __launch_bounds__(256) void heads_dw_kernel(DwLayers a) {
  int e = 0;
#pragma unroll
  for (int i = 1; i < DWL_MAX; ++i) if (i < a.n && (int)blockIdx.x >= a.begin[i]) e = i;
  const int idx = ((int)blockIdx.x - a.begin[e]) * 256 + threadIdx.x, o = idx / HD, k = idx - o * HD;
  if (o >= a.O[e]) return;
  const float* __restrict__ g = a.g[e] + o;
  const float* __restrict__ x = a.x[e] + k;
  const int ldg = a.ldg[e];
  float s0 = 0.f, s1 = 0.f, t0 = 0.f, t1 = 0.f;
  int b = 0;
  for (; b + 1 < a.B; b += 2) {
    const float g0 = g[(long)b * ldg], g1 = g[(long)(b + 1) * ldg];
    s0 = fmaf(g0, x[(long)b * HD], s0); s1 = fmaf(g1, x[(long)(b + 1) * HD], s1);
    t0 += g0; t1 += g1;
  }
  if (b < a.B) { const float g0 = g[(long)b * ldg]; s0 = fmaf(g0, x[(long)b * HD], s0); t0 += g0; }
  a.dw[e][(long)o * HD + k] += s0 + s1;
  if (k == 0 && a.db[e]) a.db[e][o] += t0 + t1;
}

// ---- classifier forward: grid (B), 128 threads
__global__ __launch_bounds__(128) void classifier_fwd_kernel(const float* __restrict__ emb, const float* __restrict__ w1, const float* __restrict__ b1,
                                                            const float* __restrict__ w2, const float* __restrict__ b2, const float* __restrict__ w3,
                                                            const float* __restrict__ b3, float* __restrict__ h1, float* __restrict__ h2,
                                                            float* __restrict__ logits, int S) {
  __shared__ float xs[HD], ys[HD];
  const int b = blockIdx.x, t = threadIdx.x;
  xs[t] = emb[(long)b * HD + t];                     // (the gradient reversal in front of the first layer is an identity here)
  __syncthreads();
  const float v1 = fmaxf(dot128(xs, w1 + (long)t * HD, b1[t]), 0.f);
  h1[(long)b * HD + t] = v1;
  ys[t] = v1;
  __syncthreads();
  const float v2 = fmaxf(dot128(ys, w2 + (long)t * HD, b2[t]), 0.f);
  h2[(long)b * HD + t] = v2;
  xs[t] = v2;
  __syncthreads();
  if (t < S) logits[(long)b * S + t] = dot128(xs, w3 + (long)t * HD, b3[t]);
}

// ---- classifier backward, data side: grid (B), 128 threads.  g3 = d_logits, g2 = (g3 W3) * (h2 > 0), g1 = (g2 W2) * (h1 > 0),
// d_emb = -lambda * (g1 W1) (the gradient reversal, model.py:34-38).  g2 / g1 are stored for the weight-gradient launch.
__global__ __launch_bounds__(128) void classifier_bwd_dx_kernel(const float* __restrict__ dlogits, const float* __restrict__ h1, const float* __restrict__ h2,
                                                               const float* __restrict__ w1, const float* __restrict__ w2, const float* __restrict__ w3,
                                                               float* __restrict__ g1, float* __restrict__ g2, float* __restrict__ d_emb, float lambda,
                                                               int S) {
  __shared__ float gs[HD];
  const int b = blockIdx.x, k = threadIdx.x;
  if (k < S) gs[k] = dlogits[(long)b * S + k];
  __syncthreads();
  float acc = 0.f;
  for (int o = 0; o < S; ++o) acc = fmaf(gs[o], w3[(long)o * HD + k], acc);
  const float v2 = h2[(long)b * HD + k] > 0.f ? acc : 0.f;
  g2[(long)b * HD + k] = v2;
  __syncthreads();
  gs[k] = v2;
  __syncthreads();
  acc = 0.f;
#pragma unroll 8
  for (int o = 0; o < HD; ++o) acc = fmaf(gs[o], w2[(long)o * HD + k], acc);
  const float v1 = h1[(long)b * HD + k] > 0.f ? acc : 0.f;
  g1[(long)b * HD + k] = v1;
  __syncthreads();
  gs[k] = v1;
  __syncthreads();
  acc = 0.f;
#pragma unroll 8
  for (int o = 0; o < HD; ++o) acc = fmaf(gs[o], w1[(long)o * HD + k], acc);
  d_emb[(long)b * HD + k] = -lambda * acc;
}

int fill_layout(FilmLayout& L, const int* nb, const int* ch) {
  L.W = 0; L.nblk = 0;
  for (int m = 0; m < 3; ++m) { L.nb[m] = nb[m]; L.ch[m] = ch[m]; L.W += nb[m] * ch[m]; L.nblk += nb[m]; }
  return L.W;
}

}  // namespace

extern "C" int dx_film_head_fwd(const float* emb, const float* spk_table, const int64_t* spk_ids, const float* wg, const float* bg,
                                const float* wb, const float* bb, const float* post, float* z, float* g_raw, float* b_raw,
                                float* film_enc, float* film_pp, float* film_dec, const int* nb, const int* ch, int B, int C, void* stream) {
  DX_REQUIRE(emb && spk_table && spk_ids && wg && bg && wb && bb && z && g_raw && b_raw && film_enc && film_pp && film_dec && nb && ch,
             DX_ERR_ARG, "dx_film_head_fwd: null pointer");
  DX_REQUIRE(C == HD && B > 0, DX_ERR_UNSUPPORTED, "dx_film_head_fwd: C=%d (only 128)", C);
  FilmFwdArgs a{emb, spk_table, spk_ids, wg, bg, wb, bb, post, z, g_raw, b_raw, {film_enc, film_pp, film_dec}, {}};
  fill_layout(a.L, nb, ch);
  hipLaunchKernelGGL(film_head_fwd_kernel, dim3(dx_cdiv(a.L.W, 256), B), dim3(256), 0, (hipStream_t)stream, a);
  DX_LAUNCH_CHECK();
  return DX_OK;
}

extern "C" int dx_film_head_bwd(const float* g_raw, const float* b_raw, const float* post, const float* z, const int64_t* spk_ids,
                                const float* wg, const float* wb, const float* dfilm_enc, const float* dfilm_pp, const float* dfilm_dec,
                                float* dg_raw_ws, float* db_raw_ws, float* d_emb, float* d_spk_table, float* dpost, float* dwg, float* dbg,
                                float* dwb, float* dbb, const int* nb, const int* ch, int B, int C, void* stream) {
  DX_REQUIRE(g_raw && b_raw && z && spk_ids && wg && wb && dfilm_enc && dfilm_pp && dfilm_dec && dg_raw_ws && db_raw_ws && d_emb && d_spk_table &&
             dwg && dbg && dwb && dbb && nb && ch, DX_ERR_ARG, "dx_film_head_bwd: null pointer");
  DX_REQUIRE(C == HD && B > 0, DX_ERR_UNSUPPORTED, "dx_film_head_bwd: C=%d (only 128)", C);
  hipStream_t s = (hipStream_t)stream;
  FilmBwdDxArgs a{g_raw, b_raw, post, spk_ids, wg, wb, {dfilm_enc, dfilm_pp, dfilm_dec}, dg_raw_ws, db_raw_ws, d_emb, d_spk_table, post ? dpost : nullptr, {}};
  const int W = fill_layout(a.L, nb, ch);
  hipLaunchKernelGGL(film_head_bwd_dx_kernel, dim3(dx_cdiv(W, 128), B), dim3(128), 0, s, a);
  DwLayers d{};
  d.n = 2; d.B = B;
  d.g[0] = dg_raw_ws; d.x[0] = z; d.dw[0] = dwg; d.db[0] = dbg; d.O[0] = W; d.ldg[0] = W; d.begin[0] = 0;
  d.g[1] = db_raw_ws; d.x[1] = z; d.dw[1] = dwb; d.db[1] = dbb; d.O[1] = W; d.ldg[1] = W; d.begin[1] = dx_cdiv(W * HD, 256);
  d.begin[2] = 2 * dx_cdiv(W * HD, 256);
  hipLaunchKernelGGL(heads_dw_kernel, dim3(d.begin[2]), dim3(256), 0, s, d);
  DX_LAUNCH_CHECK();
  return DX_OK;
}

extern "C" int dx_classifier_fwd(const float* emb, const float* w1, const float* b1, const float* w2, const float* b2, const float* w3,
                                 const float* b3, float* h1, float* h2, float* logits, int B, int C, int S, void* stream) {
  DX_REQUIRE(emb && w1 && b1 && w2 && b2 && w3 && b3 && h1 && h2 && logits, DX_ERR_ARG, "dx_classifier_fwd: null pointer");
  DX_REQUIRE(C == HD && B > 0 && S > 0 && S <= HD, DX_ERR_UNSUPPORTED, "dx_classifier_fwd: C=%d (only 128), S=%d (1..128)", C, S);
  hipLaunchKernelGGL(classifier_fwd_kernel, dim3(B), dim3(128), 0, (hipStream_t)stream, emb, w1, b1, w2, b2, w3, b3, h1, h2, logits, S);
  DX_LAUNCH_CHECK();
  return DX_OK;
}

extern "C" int dx_classifier_bwd(const float* d_logits, const float* emb, const float* h1, const float* h2, const float* w1, const float* w2,
                                 const float* w3, float* g1_ws, float* g2_ws, float* d_emb, float lambda, float* dw1, float* db1, float* dw2,
                                 float* db2, float* dw3, float* db3, int B, int C, int S, void* stream) {
  DX_REQUIRE(d_logits && emb && h1 && h2 && w1 && w2 && w3 && g1_ws && g2_ws && d_emb && dw1 && db1 && dw2 && db2 && dw3 && db3, DX_ERR_ARG,
             "dx_classifier_bwd: null pointer");
  DX_REQUIRE(C == HD && B > 0 && S > 0 && S <= HD, DX_ERR_UNSUPPORTED, "dx_classifier_bwd: C=%d (only 128), S=%d (1..128)", C, S);
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(classifier_bwd_dx_kernel, dim3(B), dim3(128), 0, s, d_logits, h1, h2, w1, w2, w3, g1_ws, g2_ws, d_emb, lambda, S);
  DwLayers d{};
  d.n = 3; d.B = B;
  const int blk = dx_cdiv(HD * HD, 256);
  d.g[0] = g1_ws; d.x[0] = emb; d.dw[0] = dw1; d.db[0] = db1; d.O[0] = HD; d.ldg[0] = HD; d.begin[0] = 0;
  d.g[1] = g2_ws; d.x[1] = h1; d.dw[1] = dw2; d.db[1] = db2; d.O[1] = HD; d.ldg[1] = HD; d.begin[1] = blk;
  d.g[2] = d_logits; d.x[2] = h2; d.dw[2] = dw3; d.db[2] = db3; d.O[2] = S; d.ldg[2] = S; d.begin[2] = 2 * blk;
  d.begin[3] = 2 * blk + dx_cdiv(S * HD, 256);
  hipLaunchKernelGGL(heads_dw_kernel, dim3(d.begin[3]), dim3(256), 0, s, d);
  DX_LAUNCH_CHECK();
  return DX_OK;
}
