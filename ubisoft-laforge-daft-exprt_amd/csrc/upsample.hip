// K11 -- Gaussian upsampling (GaussianUpsamplingModule.forward, model.py:608-662), fp32 throughout
// (the alignment weights lose ~10 % in bf16), integer part exact:
//
//   x'      = enc + conv1->128(energy) + conv1->128(pitch)                          (model.py:618-628)
//   range_l = softplus(w_r . (x'_l + conv1->128(dur_float)_l) + b_r), 1 at pad l    (model.py:633-637)
//   mu_l    = float(d_l)/2 + float(sum_{j<l} d_j)      d = int64 durations          (model.py:641-643)
//   p[l,t]  = exp(-(t+0.5-mu_l)^2 / (2 range_l^2) - log(range_l) - log(sqrt(2 pi))), 0 at pad l
//   w[l,t]  = p[l,t] / (sum_l p[l,t] + 1e-20)                                        (model.py:653-657)
//   x_up[t] = sum_l w[l,t] x'_l                                                      (model.py:659)
//
// The reference materialises a (B, L, 128, T) product (3.4 GB at B=48); here a workgroup owns 32 frames of one
// utterance, keeps the (L-chunk x 32) weight tile in LDS and contracts it against x' from L2 -- only w (B,L,T)
// (it is a model output, "alignments") and x_up ever reach HBM.  The kernel also applies the decoder's
// positional add + mask (model.py:696-701) so the decoder input is produced in the same pass.
#include "dx_common.h"

namespace {

constexpr int D = 128;
constexpr float LOG_SQRT_2PI = 0.91893853320467274178f;

// ---------------------------------------------------------------- prepare: x', ranges (one wave per (b,l) row)
struct PrepArgs {
  const float* enc; const float* dur_f; const float* energy; const float* pitch; const int64_t* lengths;
  const float* w_dur; const float* b_dur; const float* w_en; const float* b_en; const float* w_pi; const float* b_pi;
  const float* w_r; const float* b_r;
  float* xp; float* ranges; float* r_pre; float* rin;
  int L; long rows;
};
__device__ __forceinline__ float conv3(const float* ft, int l, int L, const float* w) {
  const float xm = l > 0 ? ft[l - 1] : 0.f, x0 = ft[l], xq = l + 1 < L ? ft[l + 1] : 0.f;
  return w[0] * xm + w[1] * x0 + w[2] * xq;
}
__global__ __launch_bounds__(256) void gu_prepare_kernel(PrepArgs a) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= a.rows) return;
  const int b = (int)(row / a.L), l = (int)(row - (long)b * a.L);
  float acc = 0.f;
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int c = lane * 2 + k;
    const float xp = a.enc[row * D + c] + conv3(a.energy + (long)b * a.L, l, a.L, a.w_en + c * 3) + a.b_en[c] +
                     conv3(a.pitch + (long)b * a.L, l, a.L, a.w_pi + c * 3) + a.b_pi[c];
    a.xp[row * D + c] = xp;
    const float rin = xp + conv3(a.dur_f + (long)b * a.L, l, a.L, a.w_dur + c * 3) + a.b_dur[c];
    if (a.rin) a.rin[row * D + c] = rin;
    acc += rin * a.w_r[c];
  }
  acc = dx_wave_sum(acc) + a.b_r[0];
  if (lane == 0) {
    const bool pad = l >= (int)a.lengths[b];
    const float sp = acc > 20.f ? acc : log1pf(expf(acc));  // torch Softplus(beta=1, threshold=20)
    a.ranges[row] = pad ? 1.f : sp;
    if (a.r_pre) a.r_pre[row] = acc;
  }
}

// ---------------------------------------------------------------- means: exact int64 prefix sums (one wave per utterance)
// mu[l] = d[l] / 2 + sum_{k<l} d[k]   (`model.py:632-640`: cumsum of the integer durations, centre of each segment)
__global__ __launch_bounds__(64) void gu_means_kernel(const int64_t* __restrict__ dur_int, float* __restrict__ means,
                                                       int64_t* __restrict__ totals, int B, int L) {
  const int b = blockIdx.x, lane = threadIdx.x;
  long long carry = 0;
  for (int l0 = 0; l0 < L; l0 += 64) {
    const int l = l0 + lane;
    const long long d = l < L ? (long long)dur_int[(long)b * L + l] : 0;
    long long incl = d;                       // inclusive wave scan, exact in int64
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const long long up = __shfl_up(incl, o, 64);
      if (lane >= o) incl += up;
    }
    const long long excl = carry + incl - d;
    if (l < L) {
      float mu = (float)d / 2.f;
      if (l > 0) mu += (float)excl;
      means[(long)b * L + l] = mu;
    }
    carry += __shfl(incl, 63, 64);
  }
  if (lane == 0) totals[b] = carry;
}

// ---------------------------------------------------------------- upsample forward
struct UpArgs {
  const float* xp; const float* ranges; const float* means; const int64_t* in_len; const int64_t* out_len;
  const float* pos;   // positional table or null
  float* weights;     // (B, L, T)
  float* out;         // (B, T, D): (x_up + pos) masked by out_len when pos != null, else raw x_up
  int L, T;
};
constexpr int TT = 32;   // frames per workgroup
constexpr int LC = 64;   // phoneme rows per LDS chunk

__global__ __launch_bounds__(256) void gu_upsample_fwd_kernel(UpArgs a) {
  __shared__ float Wt[LC][TT + 1];
  __shared__ float part[8][TT];
  __shared__ float denom[TT];
  __shared__ int lrange[2];   // [first, last] phoneme whose Gaussian is not exactly zero on this 32-frame tile
  const int tid = threadIdx.x, b = blockIdx.y, t0 = blockIdx.x * TT;
  if (tid == 0) { lrange[0] = 1 << 30; lrange[1] = -1; }
  __syncthreads();
  const int tt = tid & 31, lg = tid >> 5;          // phase 1/2: (frame, phoneme-group)
  const int L = a.L, T = a.T;
  const int len = (int)a.in_len[b];
  const float tc = (float)(t0 + tt) + 0.5f;
  const float* mu = a.means + (long)b * L;
  const float* sg = a.ranges + (long)b * L;
  // phase 1: denominators
  float s = 0.f;
  int lo = 1 << 30, hi = -1;
  for (int l = lg; l < len; l += 8) {
    const float sd = sg[l], dlt = tc - mu[l];
    const float pv = expf(-(dlt * dlt) / (2.f * sd * sd) - logf(sd) - LOG_SQRT_2PI);
    s += pv;
    if (pv != 0.f) { lo = min(lo, l); hi = max(hi, l); }   // far-away Gaussians underflow to exactly 0 in fp32
  }
  part[lg][tt] = s;
  if (hi >= 0) { atomicMin(&lrange[0], lo); atomicMax(&lrange[1], hi); }
  __syncthreads();
  if (tid < TT) {
    float d = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) d += part[k][tid];
    denom[tid] = d + 1e-20f;
  }
  __syncthreads();
  // phase 2: weights out + contraction against x'
  const int ft = tid >> 3, cg = tid & 7;            // (frame, 16-channel group)
  float acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  const float den = denom[tt];
  for (int l0 = 0; l0 < L; l0 += LC) {
    for (int l = l0 + lg; l < min(L, l0 + LC); l += 8) {
      float w = 0.f;
      if (l < len) {
        const float sd = sg[l], dlt = tc - mu[l];
        w = expf(-(dlt * dlt) / (2.f * sd * sd) - logf(sd) - LOG_SQRT_2PI) / den;
      }
      Wt[l - l0][tt] = w;
      if (t0 + tt < T) a.weights[((long)b * L + l) * T + t0 + tt] = w;
    }
    __syncthreads();
    // weights outside [lrange] are exactly zero on this tile: the contraction skips them (exact, ~10x less work)
    const int lmax = min(min(min(L, l0 + LC), len), lrange[1] + 1);
    for (int l = max(l0, lrange[0]); l < lmax; ++l) {
      const float w = Wt[l - l0][ft];
      const f32x4* xr = reinterpret_cast<const f32x4*>(a.xp + ((long)b * L + l) * D + cg * 16);
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const f32x4 x = xr[v];
        acc[v * 4 + 0] = fmaf(w, x[0], acc[v * 4 + 0]); acc[v * 4 + 1] = fmaf(w, x[1], acc[v * 4 + 1]);
        acc[v * 4 + 2] = fmaf(w, x[2], acc[v * 4 + 2]); acc[v * 4 + 3] = fmaf(w, x[3], acc[v * 4 + 3]);
      }
    }
    __syncthreads();
  }
  const int t = t0 + ft;
  if (t < T) {
    float* o = a.out + ((long)b * T + t) * D + cg * 16;
    if (a.pos) {
      const bool valid = t < (int)a.out_len[b];
      const float* pr = a.pos + (long)t * D + cg * 16;
#pragma unroll
      for (int i = 0; i < 16; ++i) o[i] = valid ? acc[i] + pr[i] : 0.f;
    } else {
#pragma unroll
      for (int i = 0; i < 16; ++i) o[i] = acc[i];
    }
  }
}

// ---------------------------------------------------------------- backward, phase 1: dw[l,t] = g[t,:] . x'[l,:], Dsum[t] = sum_l w dw
struct UpBwd1Args {
  const float* g;        // (B, T, D) grad wrt the kernel output (already zero where masked)
  const float* xp; const float* weights; const int64_t* in_len; const int64_t* out_len;
  float* dw;             // (B, L, T)
  float* dsum;           // (B, T)
  int L, T;
};
__global__ __launch_bounds__(256) void gu_upsample_bwd1_kernel(UpBwd1Args a) {
  __shared__ float G[TT][D + 1];
  __shared__ float part[8][TT];
  __shared__ int lrange[2];
  const int tid = threadIdx.x, b = blockIdx.y, t0 = blockIdx.x * TT;
  const int L = a.L, T = a.T, len = (int)a.in_len[b];
  const int tvalid = a.out_len ? (int)a.out_len[b] : T;
  if (tid == 0) { lrange[0] = 1 << 30; lrange[1] = -1; }
  __syncthreads();
  {   // phonemes with a non-zero weight on this tile (dw is only ever used multiplied by w, see bwd2)
    const int tt = tid & 31, lg = tid >> 5;
    int lo = 1 << 30, hi = -1;
    if (t0 + tt < T)
      for (int l = lg; l < len; l += 8)
        if (a.weights[((long)b * L + l) * T + t0 + tt] != 0.f) { lo = min(lo, l); hi = max(hi, l); }
    if (hi >= 0) { atomicMin(&lrange[0], lo); atomicMax(&lrange[1], hi); }
  }
  for (int i = tid; i < TT * D; i += 256) {
    const int r = i / D, c = i - r * D, t = t0 + r;
    G[r][c] = (t < T && t < tvalid) ? a.g[((long)b * T + t) * D + c] : 0.f;
  }
  __syncthreads();
  const int tt = tid & 31, lg = tid >> 5;
  float dacc = 0.f;
  const int lfirst = lrange[0], llast = min(lrange[1], len - 1);
  for (int l = lfirst + lg; l <= llast; l += 8) {
    float dot = 0.f;
    {
      const float* xr = a.xp + ((long)b * L + l) * D;
      for (int c = 0; c < D; ++c) dot = fmaf(G[tt][c], xr[c], dot);
    }
    if (t0 + tt < T) {
      const long idx = ((long)b * L + l) * T + t0 + tt;
      a.dw[idx] = dot;
      dacc += a.weights[idx] * dot;
    }
  }
  part[lg][tt] = dacc;
  __syncthreads();
  if (tid < TT && t0 + tid < T) {
    float d = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) d += part[k][tid];
    a.dsum[(long)b * T + t0 + tid] = d;
  }
}

// ---------------------------------------------------------------- backward, phase 2 (one workgroup per (b,l)):
//   dxp[l,:] = sum_t w[l,t] g[t,:];   dsigma[l] = sum_t w[l,t] (dw[l,t] - Dsum[t]) ((t+.5-mu)^2/s^3 - 1/s)
//   dr = dsigma * sigmoid(r_pre);  dxp += dr * w_r;  d(range weight) and d(duration projection) handled by the caller
struct UpBwd2Args {
  const float* g; const float* weights; const float* dw; const float* dsum; const float* means; const float* ranges;
  const float* r_pre; const float* w_r; const int64_t* in_len; const int64_t* out_len;
  float* dxp;        // (B, L, D) grad wrt x' (upsample path + range path)
  float* drin;       // (B, L, D) grad wrt the range-projection input (= dr * w_r), feeds the duration projection
  float* dr;         // (B, L)
  int L, T;
};
__global__ __launch_bounds__(128) void gu_upsample_bwd2_kernel(UpBwd2Args a) {
  __shared__ float red[2];
  const int c = threadIdx.x, l = blockIdx.x, b = blockIdx.y;
  const int L = a.L, T = a.T, len = (int)a.in_len[b];
  const long row = (long)b * L + l;
  float* dxp = a.dxp + row * D;
  float* drin = a.drin + row * D;
  if (l >= len) {
    dxp[c] = 0.f; drin[c] = 0.f;
    if (c == 0) a.dr[row] = 0.f;
    return;
  }
  const int tvalid = min(T, a.out_len ? (int)a.out_len[b] : T);
  const float* w = a.weights + row * T;
  const float* dwr = a.dw + row * T;
  const float* ds = a.dsum + (long)b * T;
  const float mu = a.means[row], sd = a.ranges[row];
  // frames on which this phoneme's weight is not exactly zero (a band around its Gaussian mean)
  __shared__ int trange[2];
  if (c == 0) { trange[0] = 1 << 30; trange[1] = -1; }
  __syncthreads();
  {
    int lo = 1 << 30, hi = -1;
    for (int t = c; t < tvalid; t += 128) if (w[t] != 0.f) { lo = min(lo, t); hi = max(hi, t); }
    if (hi >= 0) { atomicMin(&trange[0], lo); atomicMax(&trange[1], hi); }
  }
  __syncthreads();
  const int tfirst = trange[0], tlast = trange[1];
  float acc = 0.f, dsig = 0.f;
  for (int t = tfirst; t <= tlast; ++t) acc = fmaf(w[t], a.g[((long)b * T + t) * D + c], acc);
  for (int t = tfirst + c; t <= tlast; t += 128) {
    const float wt = w[t];
    if (wt != 0.f) {
      const float dlt = (float)t + 0.5f - mu;
      dsig += wt * (dwr[t] - ds[t]) * (dlt * dlt / (sd * sd * sd) - 1.f / sd);
    }
  }
  dsig = dx_wave_sum(dsig);
  if ((c & 63) == 0) red[c >> 6] = dsig;
  __syncthreads();
  const float rp = a.r_pre[row];
  const float dr = (red[0] + red[1]) * (rp > 20.f ? 1.f : 1.f / (1.f + expf(-rp)));
  const float dri = dr * a.w_r[c];
  dxp[c] = acc + dri;
  drin[c] = dri;
  if (c == 0) a.dr[row] = dr;
}

}  // namespace

extern "C" int dx_gu_prepare(const float* enc, const float* dur_float, const float* energy, const float* pitch,
                             const int64_t* in_lengths, const float* w_dur, const float* b_dur, const float* w_en,
                             const float* b_en, const float* w_pi, const float* b_pi, const float* w_range,
                             const float* b_range, float* xp, float* ranges, float* r_pre, float* rin, int B, int L,
                             int C, void* stream) {
  DX_REQUIRE(enc && dur_float && energy && pitch && in_lengths && xp && ranges, DX_ERR_ARG, "dx_gu_prepare: null pointer");
  DX_REQUIRE(C == D, DX_ERR_UNSUPPORTED, "dx_gu_prepare: C=%d (only 128)", C);
  DX_REQUIRE(B > 0 && L > 0, DX_ERR_SHAPE, "dx_gu_prepare: empty shape");
  PrepArgs a{enc, dur_float, energy, pitch, in_lengths, w_dur, b_dur, w_en, b_en, w_pi, b_pi, w_range, b_range, xp, ranges, r_pre, rin, L, (long)B * L};
  hipLaunchKernelGGL(gu_prepare_kernel, dim3((unsigned)((a.rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, a);
  DX_LAUNCH_CHECK();
  return DX_OK;
}

extern "C" int dx_gu_means(const int64_t* durations_int, float* means, int64_t* totals, int B, int L, void* stream) {
  DX_REQUIRE(durations_int && means && totals, DX_ERR_ARG, "dx_gu_means: null pointer");
  hipLaunchKernelGGL(gu_means_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, durations_int, means, totals, B, L);
  DX_LAUNCH_CHECK();
  return DX_OK;
}

extern "C" int dx_gu_upsample_fwd(const float* xp, const float* ranges, const float* means, const int64_t* in_lengths,
                                  const int64_t* out_lengths, const float* pos_table, float* weights, float* out, int B,
                                  int L, int T, int C, void* stream) {
  DX_REQUIRE(xp && ranges && means && in_lengths && weights && out, DX_ERR_ARG, "dx_gu_upsample_fwd: null pointer");
  DX_REQUIRE(!pos_table || out_lengths, DX_ERR_ARG, "dx_gu_upsample_fwd: pos_table needs out_lengths");
  DX_REQUIRE(C == D, DX_ERR_UNSUPPORTED, "dx_gu_upsample_fwd: C=%d (only 128)", C);
  DX_REQUIRE(B > 0 && L > 0 && T > 0, DX_ERR_SHAPE, "dx_gu_upsample_fwd: empty shape");
  UpArgs a{xp, ranges, means, in_lengths, out_lengths, pos_table, weights, out, L, T};
  hipLaunchKernelGGL(gu_upsample_fwd_kernel, dim3(dx_cdiv(T, TT), B), dim3(256), 0, (hipStream_t)stream, a);
  DX_LAUNCH_CHECK();
  return DX_OK;
}

extern "C" int dx_gu_upsample_bwd(const float* g, const float* xp, const float* weights, const float* means,
                                  const float* ranges, const float* r_pre, const float* w_range,
                                  const int64_t* in_lengths, const int64_t* out_lengths, float* dw_ws, float* dsum_ws,
                                  float* dxp, float* drin, float* dr, int B, int L, int T, int C, void* stream) {
  DX_REQUIRE(g && xp && weights && means && ranges && r_pre && w_range && in_lengths && dw_ws && dsum_ws && dxp && drin && dr,
             DX_ERR_ARG, "dx_gu_upsample_bwd: null pointer");
  DX_REQUIRE(C == D, DX_ERR_UNSUPPORTED, "dx_gu_upsample_bwd: C=%d (only 128)", C);
  hipStream_t s = (hipStream_t)stream;
  UpBwd1Args a1{g, xp, weights, in_lengths, out_lengths, dw_ws, dsum_ws, L, T};
  hipLaunchKernelGGL(gu_upsample_bwd1_kernel, dim3(dx_cdiv(T, TT), B), dim3(256), 0, s, a1);
  UpBwd2Args a2{g, weights, dw_ws, dsum_ws, means, ranges, r_pre, w_range, in_lengths, out_lengths, dxp, drin, dr, L, T};
  hipLaunchKernelGGL(gu_upsample_bwd2_kernel, dim3(L, B), dim3(128), 0, s, a2);
  DX_LAUNCH_CHECK();
  return DX_OK;
}
