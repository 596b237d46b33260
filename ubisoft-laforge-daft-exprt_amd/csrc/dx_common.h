// Shared device/host helpers for the Daft-Exprt gfx950 kernels.
// Everything here targets CDNA4 (wave64, MFMA 32x32x16 bf16 / 32x32x2 f32) directly.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/daft_exprt_hip.h"

typedef __bf16 bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define DX_WAVE 64

// ---- error plumbing (host) ------------------------------------------------------------
void dx_set_error(const char* fmt, ...);
#define DX_REQUIRE(cond, code, ...)            \
  do {                                         \
    if (!(cond)) {                             \
      dx_set_error(__VA_ARGS__);               \
      return (code);                           \
    }                                          \
  } while (0)
#define DX_LAUNCH_CHECK()                                                   \
  do {                                                                      \
    hipError_t e__ = hipGetLastError();                                     \
    if (e__ != hipSuccess) {                                                \
      dx_set_error("%s:%d launch failed: %s", __FILE__, __LINE__, hipGetErrorString(e__)); \
      return DX_ERR_LAUNCH;                                                 \
    }                                                                       \
  } while (0)

__host__ __device__ static inline int dx_cdiv(int a, int b) { return (a + b - 1) / b; }
// Dead rows -- rows of an utterance at or past its live end (length + conv halo) -- never reach a valid output; they only have to EXIST as
// finite values (zeros) as far as a consumer's last tile can reach.  Consumers read rows in tiles of at most 256 rows (+ 1 halo row) that
// start below length + 2, so producers zero-fill dead rows only below dx_fill_end(); rows past it are never read and stay UNWRITTEN
// (before: every frame-level kernel wrote all padding rows of a T <= 1000 batch, 38 % of its rows, as zeros -- ~2.4 GB per step).
// `len` is the entry of the lengths tensor the producer was handed: the true length, or min(length, n_max - 2) inside a grouped step
// (never more than 2 below the length), hence the + 4.  User-visible outputs (mel, predictions, alignments) keep their full zero padding.
__host__ __device__ static inline int dx_fill_end(int len, int N) {
  const int f = (((len < 0 ? 0 : len) + 4 + 255) & ~255) + 1;
  return f < N ? f : N;
}

// ---- device helpers -------------------------------------------------------------------
template <typename T>
struct Vec8;
template <>
struct Vec8<bf16_t> { typedef bf16x8 type; };
template <>
struct Vec8<float> { typedef f32x8 type; };

// load 8 consecutive elements of TA from global and convert to TC
template <typename TA, typename TC>
__device__ __forceinline__ typename Vec8<TC>::type dx_load8(const TA* p);

template <>
__device__ __forceinline__ bf16x8 dx_load8<bf16_t, bf16_t>(const bf16_t* p) {
  return *reinterpret_cast<const bf16x8*>(p);
}
template <>
__device__ __forceinline__ bf16x8 dx_load8<float, bf16_t>(const float* p) {
  f32x4 lo = *reinterpret_cast<const f32x4*>(p);
  f32x4 hi = *reinterpret_cast<const f32x4*>(p + 4);
  bf16x8 r;
  r[0] = (bf16_t)lo[0]; r[1] = (bf16_t)lo[1]; r[2] = (bf16_t)lo[2]; r[3] = (bf16_t)lo[3];
  r[4] = (bf16_t)hi[0]; r[5] = (bf16_t)hi[1]; r[6] = (bf16_t)hi[2]; r[7] = (bf16_t)hi[3];
  return r;
}
template <>
__device__ __forceinline__ f32x8 dx_load8<float, float>(const float* p) {
  f32x4 lo = *reinterpret_cast<const f32x4*>(p);
  f32x4 hi = *reinterpret_cast<const f32x4*>(p + 4);
  f32x8 r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return r;
}
template <>
__device__ __forceinline__ f32x8 dx_load8<bf16_t, float>(const bf16_t* p) {
  bf16x8 v = *reinterpret_cast<const bf16x8*>(p);
  f32x8 r;
#pragma unroll
  for (int i = 0; i < 8; ++i) r[i] = (float)v[i];
  return r;
}

// One K=16 step of a 32x32 output tile.  Lane l holds row/col (l & 31) and the 8 k-values
// 8*(l>>5)..+7 of the step for both operands.  bf16: one v_mfma_f32_32x32x16_bf16.
// fp32: eight v_mfma_f32_32x32x2_f32 (exact fp32; instruction j contracts k = j and 8+j).
__device__ __forceinline__ void dx_mma(f32x16& acc, const bf16x8& a, const bf16x8& b) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
}
__device__ __forceinline__ void dx_mma(f32x16& acc, const f32x8& a, const f32x8& b) {
#pragma unroll
  for (int j = 0; j < 8; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[j], acc, 0, 0, 0);
}

// row index inside a 32x32 accumulator tile for register r of lane-group g = lane >> 5
__device__ __forceinline__ int dx_acc_row(int r, int g) { return (r & 3) + 8 * (r >> 2) + 4 * g; }

template <typename T>
__device__ __forceinline__ float dx_to_f32(T v) { return (float)v; }

// DPP moves inside the 16-lane rows of a wave (lanes 16 k .. 16 k + 15): v_add_f32_dpp, ~8 cycles -- a __shfl_xor compiles to
// ds_bpermute_b32, a ~100-cycle round trip through the LDS crossbar that a kernel with ONE wave per SIMD cannot hide
template <int CTRL>
__device__ __forceinline__ float dx_dpp(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
// sum over the 16 lanes of a row, every lane gets the total (a symmetric butterfly: the same bits in every lane):
// quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror, row_mirror
__device__ __forceinline__ float dx_row16_sum(float v) {
  v += dx_dpp<0xB1>(v);
  v += dx_dpp<0x4E>(v);
  v += dx_dpp<0x141>(v);
  v += dx_dpp<0x140>(v);
  return v;
}

__device__ __forceinline__ float dx_wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
__device__ __forceinline__ float dx_wave_max(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
  return v;
}

// ---- per-utterance work lists without serial walks over the batch ------------------------------------------------------
// Several kernels split a flat list of work items (64-row items, 128-row tiles) over their workgroups, where utterance b owns
// cnt_of(val_of(b)) items and val_of(b) is derived from lengths[b].  Walking the lengths serially -- once to count, once more to find
// the utterance that holds the workgroup's first item, every wave of every workgroup -- is a chain of up to 2 B dependent global
// round trips: ~20 us in front of a 25 us weight-gradient launch at B = 48 (the launches never got shorter than 25 us however little
// work they had).  Here every thread loads ONE length (one round trip for the workgroup), a wave scan + a pass over the wave totals
// build the exclusive prefix sums in LDS, and the rest are LDS reads.
//   s_val[b] = val_of(b), s_cum[b] = items before utterance b, s_cum[B] = total.   s_part: THREADS / 64 ints.
// Every thread of the workgroup must call it (it has barriers); B <= DX_SCAN_MAXB (callers fall back to the serial walk above that).
constexpr int DX_SCAN_MAXB = 512;    // (LDS: two arrays of this many ints per workgroup; the largest published batch is 256 utterances)
template <int THREADS, typename FV, typename FC>
__device__ __forceinline__ void dx_block_count_scan(int B, FV val_of, FC cnt_of, int* s_val, int* s_cum, int* s_part) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int carry = 0;
  for (int b0 = 0; b0 < B; b0 += THREADS) {               // one trip for B <= THREADS
    const int b = b0 + tid;
    const int val = b < B ? val_of(b) : 0;
    const int v = b < B ? cnt_of(val) : 0;
    int x = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const int y = __shfl_up(x, off, 64); if (lane >= off) x += y; }
    if (lane == 63) s_part[wave] = x;
    __syncthreads();
    int woff = carry, chunk = 0;
#pragma unroll
    for (int w = 0; w < THREADS / 64; ++w) { const int t = s_part[w]; if (w < wave) woff += t; chunk += t; }
    if (b < B) { s_val[b] = val; s_cum[b] = woff + x - v; }
    carry += chunk;
    __syncthreads();
  }
  if (tid == 0) s_cum[B] = carry;
  __syncthreads();
}
// the utterance that holds item i (0 <= i < s_cum[B]): the largest b with s_cum[b] <= i (utterances without items share their
// successor's prefix and are skipped)
__device__ __forceinline__ int dx_locate_item(const int* s_cum, int B, int i) {
  int lo = 0, hi = B - 1;
  while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (s_cum[mid] <= i) lo = mid; else hi = mid - 1; }
  return lo;
}

typedef short s16x4 __attribute__((ext_vector_type(4)));

// k-major gather from a row-major tile: lane gets column c = col0 + (lane & 31) % WRAP and the 8 rows
// {kA..kA+3, kB..kB+3}.
template <typename TC, int WRAP>
__device__ __forceinline__ typename Vec8<TC>::type gather8(const TC* tile, int ld, int kA, int kB, int col0, int lane);

template <>
__device__ __forceinline__ bf16x8 gather8<bf16_t, 32>(const bf16_t* tile, int ld, int kA, int kB, int col0, int lane) {
  const int i = lane & 15, half = (lane >> 4) & 1, j = i >> 2, q = i & 3;
  const bf16_t* pa = tile + (kA + j) * ld + col0 + 16 * half + 4 * q;
  const bf16_t* pb = tile + (kB + j) * ld + col0 + 16 * half + 4 * q;
  s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(pa));
  s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(pb));
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  s16x8 r = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  return __builtin_bit_cast(bf16x8, r);
}
template <>
__device__ __forceinline__ bf16x8 gather8<bf16_t, 16>(const bf16_t* tile, int ld, int kA, int kB, int col0, int lane) {
  const int i = lane & 15, j = i >> 2, q = i & 3;
  const bf16_t* pa = tile + (kA + j) * ld + col0 + 4 * q;
  const bf16_t* pb = tile + (kB + j) * ld + col0 + 4 * q;
  s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(pa));
  s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(pb));
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  s16x8 r = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  return __builtin_bit_cast(bf16x8, r);
}
template <>
__device__ __forceinline__ f32x8 gather8<float, 32>(const float* tile, int ld, int kA, int kB, int col0, int lane) {
  const int c = col0 + (lane & 31);
  f32x8 r;
#pragma unroll
  for (int t = 0; t < 4; ++t) { r[t] = tile[(kA + t) * ld + c]; r[4 + t] = tile[(kB + t) * ld + c]; }
  return r;
}
template <>
__device__ __forceinline__ f32x8 gather8<float, 16>(const float* tile, int ld, int kA, int kB, int col0, int lane) {
  const int c = col0 + (lane & 15);
  f32x8 r;
#pragma unroll
  for (int t = 0; t < 4; ++t) { r[t] = tile[(kA + t) * ld + c]; r[4 + t] = tile[(kB + t) * ld + c]; }
  return r;
}

template <typename TC>
__device__ __forceinline__ typename Vec8<TC>::type zero8() {
  typename Vec8<TC>::type v;
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = (TC)0.f;
  return v;
}
template <typename TC>
__device__ __forceinline__ typename Vec8<TC>::type pack8(const float* p) {
  typename Vec8<TC>::type v;
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = (TC)p[e];
  return v;
}

// Counter-based dropout RNG: one 32-bit hash (murmur3 finaliser) per element.  keep iff hash >= p * 2^32.
// `key` is a 32-bit stream key folded on the fly from the 64-bit call seed (+ a per-(batch, head) salt in attention);
// `idx` is the element index inside the tensor (fits 32 bits for every tensor of this model, B*N*C < 2^32).
// Bit-parity with torch's Philox stream is not a goal; the forward and backward kernels
// regenerate the same mask from (seed, element index).  64-bit multiplies are avoided on purpose: they cost 4
// quarter-rate v_mul each and made the d_head = 16 attention kernels VALU-bound.
__host__ __device__ __forceinline__ uint32_t dx_mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x85ebca6bu; x ^= x >> 13; x *= 0xc2b2ae35u; x ^= x >> 16;
  return x;
}
__host__ __device__ __forceinline__ uint32_t dx_key32(uint64_t seed, uint32_t salt) {
  return dx_mix32((uint32_t)seed ^ dx_mix32((uint32_t)(seed >> 32) + 0x9E3779B9u * (salt + 1u)));
}
// The seed a launch draws its dropout mask from: the by-value seed, plus -- when the caller passes the device-side step block
// (DxStepScalars, include/daft_exprt_hip.h) -- that block's salt, modulo 2^63.  A captured hipGraph replays fixed kernel
// arguments, so whatever changes from one optimizer step to the next has to live in memory; seed_by_value + salt is the same
// number the eager path passes by value (model._seed), hence the same mask bit for bit.
__device__ __forceinline__ uint64_t dx_seed_eff(uint64_t seed, const DxStepScalars* step) {
  return step ? ((seed + step->seed_salt) & 0x7fffffffffffffffULL) : seed;
}
// ---- attention-weight dropout: counter based, built from full-rate integer ops only (v_mul_u32_u24, v_alignbit; the 32-bit
// v_mul_lo_u32 of dx_mix32 is quarter rate on CDNA and the attention kernels are VALU bound).  One 32-bit hash serves a
// 4 x 4 block of (query, key) decisions:
//   block counter   c      = ((q >> 2) * ceil(N / 4) + (key >> 2)) * DX_CTR_MUL + stream_key          (wraps mod 2^32)
//   block hash      base   = prefix(c)                      xor-shift, 24-bit multiply, xor-shift       (3 ops per 16 decisions)
//   row word        w(q)   = mul24(rotr(base, 8 * (q & 3)), DX_BLK_M[q & 3])                            (2 ops per 4 decisions)
//   keep(q, key)           = byte (key & 3) of w(q) >= round(256 p)
// so the forward / dQ kernels (lane = query, 4 consecutive keys in registers) pay 6 ops per 4 decisions and the dK/dV kernel
// (lane = key, 4 consecutive queries in registers) 11 per 4 -- plus compare + select per decision -- where one hash per
// decision pair used to cost 6 / 10 ops per decision.
// The drop probability is quantised to p_hat = round(256 p) / 256 (p = 0.1 -> 26 / 256 = 0.1016) and the kept values are
// scaled by 1 / (1 - p_hat), so the expectation is exact.  Measured on 62 500 blocks x 3 stream keys (numpy model of the
// same integer ops): keep rates within 3e-3 of 1 - p_hat per field, |correlation| between any two of the 16 decisions of a
// block and between neighbouring blocks at the sampling-noise level (rms 4e-3 = 1 / sqrt(#blocks)), chi^2 of every byte ~1.0
// per degree of freedom.  Without the rotation the low bytes of the four row words share the low bits of `base` (|corr| 0.07).
constexpr uint32_t DX_CTR_MUL = 0x9E3779B1u, DX_M24_PRE = 0x9E3779u;
constexpr uint32_t DX_BLK_M0 = 0xC2B2AFu, DX_BLK_M1 = 0x85EBCBu, DX_BLK_M2 = 0xA54FF5u, DX_BLK_M3 = 0x6C8E95u;
// p > 0 always drops something and never everything: the threshold is clamped to [1, 255] (p < 1/512 -> 1/256, p > 255/256 -> 255/256)
__host__ __device__ __forceinline__ uint32_t dx_drop_th8(float p) {
  if (p <= 0.f) return 0u;
  const uint32_t t = (uint32_t)(p * 256.f + 0.5f);
  return t < 1u ? 1u : (t > 255u ? 255u : t);
}
__host__ __device__ __forceinline__ float dx_drop_inv_keep8(uint32_t th8) { return 256.f / (float)(256u - th8); }
__device__ __forceinline__ uint32_t dx_drop_prefix(uint32_t ctr) {
  ctr ^= ctr >> 16;                    // both xor-shifts are by 16: one v_xor_b32_sdwa (src1_sel:WORD_1) each
  ctr = __umul24(ctr, DX_M24_PRE);
  ctr ^= ctr >> 16;
  return ctr;
}
// keep the compiler from folding a lane counter into the tile offset added to it later: (a * MUL + k) + (t * MUL + c) would
// otherwise come back as ONE quarter-rate 32-bit multiply per hash instead of one add
__device__ __forceinline__ uint32_t dx_opaque(uint32_t v) { asm volatile("" : "+v"(v)); return v; }
__device__ __forceinline__ uint32_t dx_blk_mult(int row) {
  return row == 0 ? DX_BLK_M0 : row == 1 ? DX_BLK_M1 : row == 2 ? DX_BLK_M2 : DX_BLK_M3;
}
// four block hashes at once, step by step: a wave issues a DEPENDENT VALU instruction only every ~9 cycles on gfx950 (an
// independent one every ~6.5, tools/probes/valu_rate_probe.hip), so the four chains are interleaved rather than run in turn
__device__ __forceinline__ void dx_drop_prefix4(uint32_t* c) {
#pragma unroll
  for (int j = 0; j < 4; ++j) c[j] ^= c[j] >> 16;
#pragma unroll
  for (int j = 0; j < 4; ++j) c[j] = __umul24(c[j], DX_M24_PRE);
#pragma unroll
  for (int j = 0; j < 4; ++j) c[j] ^= c[j] >> 16;
}
// row word of a block hash; rot = 8 * (q & 3), mult = dx_blk_mult(q & 3) (lane constants or literals)
__device__ __forceinline__ uint32_t dx_drop_row(uint32_t base, uint32_t rot, uint32_t mult) {
  return __umul24(__builtin_amdgcn_alignbit(base, base, rot), mult);
}
// byte `i` (compile-time) of a row word against the 8-bit threshold
__device__ __forceinline__ bool dx_keep8(uint32_t w, int i, uint32_t th8) { return ((w >> (8 * i)) & 0xffu) >= th8; }
// byte at a lane-dependent position: shl = 24 - 8 * (key & 3), th_top = th8 << 24 (the low garbage bits cannot flip the compare)
__device__ __forceinline__ bool dx_keep8_var(uint32_t w, uint32_t shl, uint32_t th_top) { return (w << shl) >= th_top; }
// v[i] = byte i of w >= th8 ? v[i] : 0 for the four byte fields of one row word.  Written out because a compare into an SGPR
// pair followed directly by the select that reads it costs two wait states on gfx950 and the compiler serialises the four
// pairs through VCC (s_nop 1 after every compare): four SDWA byte compares into four SGPR pairs, then the four selects.
__device__ __forceinline__ void dx_drop4(float& v0, float& v1, float& v2, float& v3, uint32_t w, uint32_t th8) {
  uint64_t m0, m1, m2, m3;
  asm("v_cmp_ge_u32_sdwa %4, %8, %9 src0_sel:BYTE_0 src1_sel:DWORD\n\t"
      "v_cmp_ge_u32_sdwa %5, %8, %9 src0_sel:BYTE_1 src1_sel:DWORD\n\t"
      "v_cmp_ge_u32_sdwa %6, %8, %9 src0_sel:BYTE_2 src1_sel:DWORD\n\t"
      "v_cmp_ge_u32_sdwa %7, %8, %9 src0_sel:BYTE_3 src1_sel:DWORD\n\t"
      "v_cndmask_b32_e64 %0, 0, %0, %4\n\t"
      "v_cndmask_b32_e64 %1, 0, %1, %5\n\t"
      "v_cndmask_b32_e64 %2, 0, %2, %6\n\t"
      "v_cndmask_b32_e64 %3, 0, %3, %7"
      : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "=&s"(m0), "=&s"(m1), "=&s"(m2), "=&s"(m3)
      : "v"(w), "v"(th8));
}
// the dK/dV form: one decision per row word w[i] (byte at the lane's position, see dx_keep8_var) applied to two values each
__device__ __forceinline__ void dx_drop4x2_var(float* a, float* b, uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3,
                                               uint32_t shl, uint32_t th_top) {
  uint64_t m0, m1, m2, m3;
  uint32_t t0, t1, t2, t3;
  asm("v_lshlrev_b32 %12, %20, %16\n\t"
      "v_lshlrev_b32 %13, %20, %17\n\t"
      "v_lshlrev_b32 %14, %20, %18\n\t"
      "v_lshlrev_b32 %15, %20, %19\n\t"
      "v_cmp_ge_u32_e64 %8, %12, %21\n\t"
      "v_cmp_ge_u32_e64 %9, %13, %21\n\t"
      "v_cmp_ge_u32_e64 %10, %14, %21\n\t"
      "v_cmp_ge_u32_e64 %11, %15, %21\n\t"
      "v_cndmask_b32_e64 %0, 0, %0, %8\n\t"
      "v_cndmask_b32_e64 %1, 0, %1, %9\n\t"
      "v_cndmask_b32_e64 %2, 0, %2, %10\n\t"
      "v_cndmask_b32_e64 %3, 0, %3, %11\n\t"
      "v_cndmask_b32_e64 %4, 0, %4, %8\n\t"
      "v_cndmask_b32_e64 %5, 0, %5, %9\n\t"
      "v_cndmask_b32_e64 %6, 0, %6, %10\n\t"
      "v_cndmask_b32_e64 %7, 0, %7, %11"
      : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]),
        "=&s"(m0), "=&s"(m1), "=&s"(m2), "=&s"(m3), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
      : "v"(w0), "v"(w1), "v"(w2), "v"(w3), "v"(shl), "v"(th_top));
}

// element-wise dropout of the (row, channel) activations: one hash per 4 consecutive channels, one byte each, p quantised like
// the attention-weight dropout (dx_drop_th8 / dx_drop_inv_keep8).  Every site of one stream key -- forward and backward, whichever
// kernel they live in -- must use this form.
__device__ __forceinline__ bool dx_keep_elem(uint32_t key, uint32_t idx, uint32_t th8) {
  const uint32_t h = dx_mix32((idx >> 2) * 0x9E3779B1u + key);
  return ((h >> ((idx & 3u) * 8u)) & 0xffu) >= th8;
}
