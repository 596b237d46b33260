/* daft_exprt_hip.h -- C ABI of libdaftexprt_hip.so, the MI355X (gfx950) kernel library behind the
 * Daft-Exprt acoustic-model hot path.
 *
 * The reference (ubisoft/ubisoft-laforge-daft-exprt) has no native boundary: every device op is an
 * ATen call made from src/daft_exprt/model.py / loss.py / train.py.  Each entry point below replaces
 * the ATen op sequence of the reference lines cited next to it.  The Python binding a maintainer
 * would add on the reference side is shown in INTEGRATION.md (ctypes; no torch types cross this ABI).
 *
 * Conventions
 *   - plain pointers + sizes; all pointers are DEVICE pointers unless stated otherwise;
 *   - the caller owns every buffer (inputs, outputs, workspaces); nothing is allocated or freed here;
 *   - all work is enqueued on `stream` (a hipStream_t passed as void*); no hidden synchronisation;
 *   - activations are channel-last (B, N, C); "lengths" are int64 (B,) device arrays;
 *   - return 0 on success, a negative DX_ERR_* code otherwise; dx_last_error() gives the message
 *     (thread-local).  Argument violations mirror the reference's asserts as errors, never UB.
 *   - dtype codes: DX_F32 / DX_BF16 / DX_I64.
 *   - DEAD ROWS (ABI v11).  Rows of an utterance at or past length + conv halo (`skip_lengths[b] + 2`, or the length of a
 *     masked output) never reach a valid output.  Producers write them as zeros only below
 *         dx_fill_end(len, N) = min(N, roundup(len + 4, 256) + 1)
 *     -- as far as a consumer's last tile (<= 256 rows + 1 halo row, starting below length + 2) can reach; rows at or past
 *     that index are never read by any entry point and are left UNWRITTEN (up to v10 they were written as zeros).  Callers
 *     that inspect whole (B, N, C) intermediate tensors must not look past it.  User-visible outputs -- the transposed
 *     (B, C, N) form of dx_conv1d (the mel), dx_linear_small_*, dx_gu_upsample_fwd -- keep their full zero padding.
 *     A hard sequence end n_max[b] < N (an utterance that belongs to a shorter-padded micro-batch of a grouped step) is
 *     expressed with the same arguments: skip_lengths[b] = min(length, n_max - 2), mask_lengths[b] = n_max.
 */
#ifndef DAFT_EXPRT_HIP_H
#define DAFT_EXPRT_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DX_ABI_VERSION 13

enum { DX_F32 = 0, DX_BF16 = 1, DX_I64 = 2 };
enum { DX_OK = 0, DX_ERR_ARG = -1, DX_ERR_SHAPE = -2, DX_ERR_DTYPE = -3, DX_ERR_LAUNCH = -4, DX_ERR_UNSUPPORTED = -5 };

/* flags of dx_conv1d */
enum { DX_CONV_RELU = 1, DX_CONV_TRANSPOSED_OUT = 2, DX_CONV_ACCUMULATE = 4 /* y += result */ };

int dx_abi_version(void);
const char* dx_last_error(void);

/* ---- the step block: what changes from one optimizer step to the next, in DEVICE memory.
 * The reference's trainer recomputes these on the host every iteration and they reach its kernels as scalar arguments: the
 * learning rate (train.py:139-151, 491-494), Adam's bias corrections (torch.optim.Adam, train.py:299-301), the adversarial
 * loss weight (loss.py:22-28), and -- through torch's Philox offset -- the dropout streams.  A captured step (hipGraph) replays
 * FIXED kernel arguments, so every entry point that consumes one of them takes an optional `const DxStepScalars* step`
 * (NULL = use the by-value arguments, the eager path):
 *   dropout:  effective seed = (seed argument + step->seed_salt) mod 2^63     (dx_conv1d_ln, dx_conv1d_lnbwd, dx_layernorm_fwd / _bwd,
 *             dx_attention_fwd / _bwd) -- the same number the eager path passes by value, hence the same masks bit for bit;
 *   dx_adam_step: lr, bc1 = 1 - beta1^t, bc2_sqrt = sqrt(1 - beta2^t) replace the lr / step arguments;
 *   dx_loss_fwd_bwd: w_speaker replaces w_spk.
 * dx_step_scalars_set fills the block with one single-thread launch (stream-ordered in front of the step that reads it). */
typedef struct DxStepScalars {
  uint64_t seed_salt;
  float lr, bc1, bc2_sqrt, w_speaker;
  int step;
  int reserved[3];
} DxStepScalars;
int dx_step_scalars_set(DxStepScalars* dev, uint64_t seed_salt, float lr, float beta1, float beta2, int step, float w_speaker,
                        void* stream);

/* ---- K1/K3/K12: k-tap (1 or 3) stride-1 "same" conv on channel-last activations as an implicit GEMM
 * on MFMA; also nn.Linear (taps = 1).  Replaces ConvNorm1D.forward (model.py:86-94: transpose, nn.Conv1d,
 * transpose), LinearNorm.forward (model.py:66-72) and the in/out projections of nn.MultiheadAttention
 * (model.py:182-186).
 *   x        (B, N, Cin) rows `ldx` elements apart, dtype x_dtype
 *   w_packed [taps][Cout][Cin], dtype w_dtype = the MFMA operand type (DX_BF16 -> v_mfma_f32_32x32x16_bf16,
 *            DX_F32 -> v_mfma_f32_32x32x2_f32); produced by dx_pack_conv_weight
 *   bias     (Cout) fp32 or NULL
 *   y        (B, N, Cout) rows `ldy` apart -- or (B, Cout, N) rows `ldy` apart with DX_CONV_TRANSPOSED_OUT
 *   relu_gate NULL, or a tensor laid out exactly like y (dtype gate_dtype): y is multiplied by (gate > 0)
 *            (the ReLU derivative when this call is the data-gradient of a conv that fed a ReLU)
 *   mask_lengths NULL, or int64 (B): rows n >= mask_lengths[b] are written as zeros
 *            (masked_fill of model.py:259,262,569,707)
 *   skip_lengths NULL, or int64 (B): padding early-out -- 128-row tiles that start at n0 >= skip_lengths[b] + 2 are
 *            written as zeros without touching the MFMA pipe (rows past length + conv halo never reach a valid
 *            output) -- and not written at all from dx_fill_end on (DEAD ROWS above)
 * Positions outside [0, N) are zero padding (N = the batch's max length).  Cin % 8 == 0.
 */
int dx_conv1d(const void* x, int x_dtype, long ldx, const void* w_packed, int w_dtype, const float* bias,
              void* y, int y_dtype, long ldy, const void* relu_gate, int gate_dtype, const int64_t* mask_lengths,
              const int64_t* skip_lengths, int B, int N, int Cin, int Cout, int taps, int flags, void* stream);

/* dx_conv1d with an optional fragment-order copy of the weights (dx_pack_frag_major(_batched) of w_packed; NULL = dx_conv1d):
 * the register-weights kernel of the Cin = 128, Cout % 256 == 0 GEMMs (FF conv 128 -> 1024 and the data gradient of
 * 1024 -> 128) then loads every MFMA fragment of its weight slice straight into its registers -- one round trip instead of
 * a pass through LDS in front of the first position tile.  Same results bit for bit. */
int dx_conv1d_wfrag(const void* x, int x_dtype, long ldx, const void* w_packed, int w_dtype, const void* w_frag, const float* bias,
                    void* y, int y_dtype, long ldy, const void* relu_gate, int gate_dtype, const int64_t* mask_lengths,
                    const int64_t* skip_lengths, int B, int N, int Cin, int Cout, int taps, int flags, void* stream);

/* The ReLU gate of the FF block as one BIT per element (ABI v12).  The data gradient of PositionWiseConvFF's second conv
 * (model.py:220-237: conv k3 -> ReLU -> conv k3) is gated by `h > 0`, h = the ReLU output of the first conv: a (B, N, Cout) tensor
 * that the gate would re-read in full.  Instead the forward conv leaves `bits` next to h and the data gradient reads those:
 *   bits_out != NULL:  y = relu(conv_k3(x, w) + bias) as dx_conv1d_wfrag with DX_CONV_RELU, and bits_out = (y > 0);
 *   bits_in  != NULL:  y = conv_k3(x, w) where bits_in, 0 elsewhere (the data gradient: w = the transposed / flipped copy, no bias).
 * bf16 x / w / y, Cin = 128, taps = 3, Cout % 256 == 0 (the register-weights kernel); w_frag optional as in dx_conv1d_wfrag.
 * bits: uint32 (B, Cout / 32, N); the bit order inside a word is private to the kernel pair (its own register order) -- only
 * these two calls read or write it.  mask_lengths / skip_lengths as in dx_conv1d; rows at or past mask_lengths have no bit set. */
int dx_conv1d_relu_bits(const void* x, long ldx, const void* w_packed, const void* w_frag, const float* bias, void* y, long ldy,
                        uint32_t* bits_out, const uint32_t* bits_in, const int64_t* mask_lengths, const int64_t* skip_lengths,
                        int B, int N, int Cout, void* stream);

/* dx_conv1d with Cout = 128 and the following LayerNorm fused into its epilogue (a 128-row tile holds complete rows):
 *   s = dropout_pre(conv(x) + bias) + residual;  y = LN(s) * gamma + beta;  y = film[b,:128] * y + film[b,128:];
 *   y = 0 where n >= lengths[b]
 * i.e. the attention out-projection + Dropout + residual + LayerNorm + mask (model.py:186-191, 259) and the second FF conv
 * + Dropout + residual + LayerNorm + FiLM + mask (model.py:226-235, 262) in one launch.  Outputs: y fp32, y_lp optional
 * bf16 copy, s_out / mean / rstd for dx_layernorm_bwd (NULL at inference).  lengths also drives the padding early-out.  * y2 (NULL = off; bf16 (B, N, n2), n2 = 128 or 384) with w2_packed (bf16 [1][n2][128]) and b2 (fp32 (n2) or NULL):
 * y2 = y_lp . w2^T + b2, the k = 1 projection that reads this LayerNorm's output next (the QKV projection of the following FFT
 * block, model.py:165-171), computed by the epilogue from the rows it has just normalised; split-K path only (see dx_conv1d_lnbwd). */
int dx_conv1d_ln(const void* x, int x_dtype, long ldx, const void* w_packed, int w_dtype, const float* bias,
                 const float* residual, const float* gamma, const float* beta, const float* film, long ldf,
                 const int64_t* lengths, float* y, void* y_lp, float* s_out, float* mean, float* rstd, int B, int N,
                 int Cin, int taps, float p_pre, uint64_t seed_pre, const int* plan, int plan_tiles, const void* w_frag,
                 const void* w2_packed, const float* b2, void* y2, int n2, const DxStepScalars* step, void* stream);

/* dx_conv1d_ln whose residual stream is re-derived instead of read (ABI v12).  In an FFT block the residual of the second LayerNorm
 * is the OUTPUT of the first one, a = mask(LN1(s1)) (model.py:186-191 -> 226-235): with res_mean != NULL `residual` is s1 (the saved
 * LayerNorm input of the launch that produced the stream), res_mean / res_rstd its row statistics, res_gamma / res_beta that
 * LayerNorm's parameters, and the epilogue computes  a = (s1 - mean) * rstd * gamma + beta  (0 where n >= lengths[b]) itself -- so
 * the producing launch may pass y = NULL and store only its bf16 copy and s1: 512 bytes per row less to write.  Split-K path only
 * (bf16, taps = 3, plan + fragment-order weights, B * N <= 65536); y = NULL is accepted by every path of both entry points as long
 * as y_lp is given.  res_mean = NULL: identical to dx_conv1d_ln. */
int dx_conv1d_ln_vres(const void* x, int x_dtype, long ldx, const void* w_packed, int w_dtype, const float* bias,
                      const float* residual, const float* res_mean, const float* res_rstd, const float* res_gamma, const float* res_beta,
                      const float* gamma, const float* beta, const float* film, long ldf,
                      const int64_t* lengths, float* y, void* y_lp, float* s_out, float* mean, float* rstd, int B, int N,
                      int Cin, int taps, float p_pre, uint64_t seed_pre, const int* plan, int plan_tiles, const void* w_frag,
                      const void* w2_packed, const float* b2, void* y2, int n2, const DxStepScalars* step, void* stream);

/* Data gradient of a conv / linear INTO a 128-channel residual stream, fused with the BACKWARD of the LayerNorm that
 * consumed that stream in the forward pass (autograd of model.py:189-191 resp. 226-235 + 259/262, i.e. what
 * dx_conv1d(..., ACCUMULATE) followed by dx_layernorm_bwd compute in two launches and two extra passes over the tensor):
 *   g  = y_inout + conv(x, w_packed)            (x = gradient of the conv's output, w_packed = transpose_flip packing)
 *   g  = 0 where n >= lengths[b];  FiLM: dfilm[b] += (sum_n g * LN, sum_n g), g *= film_gamma[b]
 *   dgamma += sum g * xhat, dbeta += sum g;  ds = rstd * (g*gamma - mean_c(g*gamma) - xhat * mean_c(g*gamma*xhat))
 *   y_inout <- ds (fp32, the residual gradient that flows on);  dx_pre_lp <- dropout_pre(ds) as bf16 (MFMA operand of the
 *   previous layer's data / weight gradient).   s_in / mean / rstd: saved by dx_conv1d_ln / dx_layernorm_fwd.
 * Cout is 128 by construction.  lengths also drives the padding early-out (rows >= lengths[b] + 2 stay zero).
 * y2 (NULL = off; bf16 (B, N, 128)) with w2_packed (bf16 [1][128][128], forward packing of the map to apply): y2 = dx_pre_lp . w2^T,
 * the 128 -> 128 linear layer whose data gradient consumes dx_pre_lp next (the attention output projection, model.py:182-186),
 * computed by the epilogue from the rows it has just produced instead of by a launch of its own; needs the split-K path (bf16,
 * taps = 3, plan + w_frag, Cin % 128 == 0, B * N <= 65536: DX_ERR_UNSUPPORTED otherwise).  Same values as dx_conv1d on dx_pre_lp. */
int dx_conv1d_lnbwd(const void* x, int x_dtype, long ldx, const void* w_packed, int w_dtype, float* y_inout,
                    const float* s_in, const float* mean, const float* rstd, const float* gamma, const float* beta,
                    const float* film, long ldf, const int64_t* lengths, void* dx_pre_lp, float* dgamma, float* dbeta,
                    float* dfilm, long lddf, int B, int N, int Cin, int taps, float p_pre, uint64_t seed_pre,
                    const int* plan, int plan_tiles, const void* w_frag, const void* w2_packed, void* y2, const DxStepScalars* step,
                    void* stream);

/* The weights of a k = 3 conv in MFMA-fragment order: out[chunk][tap][half][block][lane][8] = w_packed[tap][32 block + (lane & 31)]
 * [32 chunk + 16 half + 8 (lane >> 5) + 0..7], block < Cout / 32; w_packed = the [3][Cout][Cin] bf16 packing of dx_pack_conv_weight
 * (either orientation), Cin and Cout multiples of 32.  A fragment (the B operand of one v_mfma_f32_32x32x16_bf16) is one contiguous
 * KiB, so a wave reads it from L2 straight into registers with one fully coalesced load and the weights never pass through LDS.
 * Users: the optional `w_frag` of dx_conv1d_ln / dx_conv1d_lnbwd (Cout = 128, with a tile plan, Cin >= 256, Cin % 128 == 0: split-K
 * workgroups of 4 waves with 512 registers each -- all rows x 64 channels x half of the contraction per wave, accumulators in AGPRs,
 * four K chunks of fragments in flight per wave, LDS carries the activation tile only, the two K halves added through LDS in a
 * fixed order; results differ from the w_frag = NULL path by fp32 summation order only) and dx_conv1d_wide. */
int dx_pack_frag_major(const void* w_packed, void* out, int Cin, int Cout, void* stream);
/* The same for n weights in one launch.  descs_dev: DEVICE array of n records {const void* src; void* dst; int Cin; int Cout;}
 * (dx_frag_desc_size() bytes each); max_elems = the largest Cin * Cout * 3 in the table. */
int dx_frag_desc_size(void);
int dx_pack_frag_major_batched(const void* descs_dev, int n, long max_elems, void* stream);

/* dx_conv1d for the WIDE k = 3 GEMMs (ConvNorm1D with 1024 -> 1024 channels in the prosody encoder's pre-net, model.py:341-363, and its
 * data gradient): bf16 x (B, N, Cin) and y (B, N, Cout), weights in fragment order, Cin % 128 == 0, Cout % 256 == 0, flags = 0 or
 * DX_CONV_RELU, bias fp32 or NULL.  256 rows x 256 channels per 4-wave workgroup (one wave per SIMD, 128 x 128 per wave = 256
 * accumulator registers), activation tile through an LDS-DMA ring, weight fragments from L2 into registers one chunk ahead:
 * 0.25 KB of LDS traffic per MFMA and half the L2 -> CU bytes of 256 x 128 tiles.  Position tiles: `plan` = dx_conv_tile_plan(lengths,
 * ..., n_tiles, table, halo) with ANY n_tiles >= B * ceil(N / 256) -- a multiple of 64 (256 CUs / 4 channel tiles of a 1024-channel output)
 * fills the chip in whole rounds; rows n >= lengths[b] + halo never reach a valid output and are written as zeros. */
int dx_conv1d_wide(const void* x, long ldx, const void* w_frag, const float* bias, void* y, long ldy, const int64_t* lengths,
                   const int* plan, int plan_tiles, int halo, int B, int N, int Cin, int Cout, int flags, void* stream);

/* Balanced position tiles for dx_conv1d_ln / dx_conv1d_lnbwd (optional `plan`; bf16 operands, Cin % 32 == 0, taps = 3 --
 * dx_conv1d_lnbwd also taps = 1).
 * The k = 3 GEMMs into the 128-channel stream are bound by what a compute unit fetches from L2, and every workgroup
 * fetches the whole weight slice whatever the height of its tile; a ragged batch (model.py:21-23 masks, lengths differ
 * 10x inside a batch) cut into fixed 128-row tiles ends a few tiles above a multiple of the 256 CUs.  The plan cuts
 * each utterance into equal pieces of <= 256 rows so that the batch is at most n_tiles pieces with the smallest possible
 * maximum height.  n_tiles: any count >= B * ceil(N / 256); dx_conv_tile_plan_size(B, N) is that bound rounded up to a
 * multiple of the 256 CUs (the default of the Python host side).  table: n_tiles x 4 int32 {b, n0, rows, f}
 * (f = padding rows of the batch each workgroup zero-fills beside its tile),
 * device memory, valid for every GEMM over the same lengths (one plan per batch).  Results are identical to the
 * unplanned call: same rows, same summation order per output; rows >= lengths[b] are written as zeros. */
int dx_conv_tile_plan_size(int B, int N);
int dx_conv_tile_plan(const int64_t* lengths, int B, int N, int n_tiles, int* table, int halo /* rows past the length that still
                      carry work: 0 for the LayerNorm-fused GEMMs (masked), 2 for the pre-net convs of dx_conv1d_wide */, void* stream);
/* One launch for everything a step derives from one lengths tensor: dx_conv_tile_plan(halo 0) into table0 (n_tiles0 entries),
 * dx_conv_tile_plan(halo 2) into table2 (n_tiles2 entries) and dx_length_order into order (B ints); any of the three
 * outputs may be NULL.  Same results as the three separate calls. */
int dx_batch_prep(const int64_t* lengths, int B, int N, int n_tiles0, int* table0, int n_tiles2, int* table2, int* order,
                  void* stream);

/* Pack an fp32 (Cout, Cin, taps) PyTorch conv / (Cout, Cin) linear weight for dx_conv1d.
 *   transpose_flip = 0: out[tap][co][ci] = w[co][ci][tap]                (forward operand)
 *   transpose_flip = 1: out[tap][ci][co] = w[co][ci][taps-1-tap]         (data-gradient operand) */
int dx_pack_conv_weight(const float* w, void* out, int out_dtype, int Cout, int Cin, int taps, int transpose_flip,
                        void* stream);
/* Every GEMM weight of the model in one launch.  descs_dev: DEVICE array of n records
 * {const float* w; void* out; int Cout, Cin, taps, transpose_flip; long begin;} (dx_pack_desc_size() bytes each); a
 * workgroup packs one 32 (Cout) x 32 (Cin) brick of one weight (all taps): begin = running sum of
 * ceil(Cout / 32) * ceil(Cin / 32) over the records before this one, total_bricks = that sum over all records. */
int dx_pack_desc_size(void);
int dx_pack_conv_weights_batched(const void* descs_dev, int n, long total_bricks, int out_dtype, void* stream);

/* Weight / bias gradient of dx_conv1d (and of nn.Linear with taps = 1), accumulated into dw (Cout, Cin, taps)
 * [PyTorch layout, fp32] and db (Cout) [NULL to skip]:
 *   dw[co][ci][tap] += sum_{b,n} dy[b, n, co] * x[b, n + tap - taps/2, ci];   db[co] += sum_{b,n} dy[b, n, co]
 * dy (B, N, Cout) rows lddy apart, x (B, N, Cin) rows ldx apart; compute_dtype = MFMA operand type.
 * lengths (NULL = none): rows n >= lengths[b] + 2 of dy are known to be zero and are skipped (the position axis is
 * split over workgroups by valid rows).  ws: scratch of dx_conv1d_wgrad_ws_floats(...) floats (contents irrelevant, must
 * stay untouched until the call has completed on `stream`): the per-workgroup partial sums go there and a second
 * launch adds them to dw in a fixed order.  ws = NULL: partial sums are added to dw with fp32 atomics instead
 * (slower, summation order varies run to run).  db always uses atomics (Cout values per workgroup). */
long dx_conv1d_wgrad_ws_floats(int B, int N, int Cin, int Cout, int taps);
int dx_conv1d_wgrad(const void* dy, int dy_dtype, long lddy, const void* x, int x_dtype, long ldx,
                    int compute_dtype, float* dw, float* db, const int64_t* lengths, float* ws, int B, int N, int Cin,
                    int Cout, int taps, void* stream);

/* n (<= 8) weight gradients over the SAME batch rows (B, N, lengths) in one call -- e.g. the four of an FFT block's backward pass
 * (model.py:182-186, 226-235: QKV and output projections, the two feed-forward convolutions): the n GEMM launches of dx_conv1d_wgrad,
 * then ONE launch that adds every partial tile to its dw (instead of one reduce launch behind each GEMM; same summation order per
 * element, so the results are those of n dx_conv1d_wgrad calls bit for bit).  descs: HOST array of n records; ws: scratch of
 * dx_conv1d_wgrad_multi_ws_floats(descs, n, B, N) floats (required). */
typedef struct DxWgradDesc {
  const void* dy; const void* x; float* dw; float* db;
  long lddy, ldx;
  int dy_dtype, x_dtype, Cin, Cout, taps, pad;
} DxWgradDesc;
int dx_wgrad_desc_size(void);
long dx_conv1d_wgrad_multi_ws_floats(const DxWgradDesc* descs, int n, int B, int N);
int dx_conv1d_wgrad_multi(const DxWgradDesc* descs, int n, int compute_dtype, const int64_t* lengths, float* ws, int B, int N,
                          void* stream);

/* ---- K5: LayerNorm(C) over channel-last rows fused with its neighbours (C in {128, 256, 1024}):
 *   s = dropout_pre(x) + residual;  y = LN(s) * gamma + beta;  y = dropout_post(y);
 *   y = film[b, :C] * y + film[b, C:];  y = 0 where n >= lengths[b]
 * Replaces nn.LayerNorm + nn.Dropout + residual add + FiLM + masked_fill at model.py:189-191 (attention),
 * 226-235 + 262 (FF block), 346-348/353-355/360-362 (prenet), 533-535/540-542 + 558-566 (predictor).
 * residual / film / lengths / s_out / mean+rstd may be NULL (feature off).  s_out, mean, rstd (fp32) are what
 * dx_layernorm_bwd needs.  Dropout masks are a counter-based hash of (seed, element index), regenerated
 * identically by the backward kernel; p = 0 disables.  skip_lengths (NULL = off): rows n >= skip_lengths[b] + 2 are
 * written as zeros without being read (padding early-out, same rule as dx_conv1d: unwritten from dx_fill_end on).  C in {128, 256, 512, 1024}. */
int dx_layernorm_fwd(const void* x, int x_dtype, const float* residual, const float* gamma, const float* beta,
                     const float* film, long ldf, const int64_t* lengths, const int64_t* skip_lengths, void* y,
                     int y_dtype, void* y_lp /* optional bf16 copy of y */, float* s_out, float* mean, float* rstd, int B, int N, int C, float p_pre,
                     uint64_t seed_pre, float p_post, uint64_t seed_post, const DxStepScalars* step, void* stream);

/* Backward of dx_layernorm_fwd.  dy: grad wrt y.  s_in: s_out of the forward (or x itself when there was no
 * residual / pre-dropout).  Outputs: ds = grad wrt s (equals the residual-branch gradient); dx_pre = grad wrt x
 * through dropout_pre (NULL when p_pre == 0: then it equals ds); dgamma, dbeta (C) and dfilm (B, 2C) are
 * ACCUMULATED with fp32 atomics (zero them first).  relu_input != 0: the normalised tensor was relu(conv) (prenet,
 * predictor); ds is then gated by (s_in > 0) so that it is the gradient of the conv's pre-activation. */
int dx_layernorm_bwd(const void* dy, int dy_dtype, const void* s_in, int s_dtype, const float* mean,
                     const float* rstd, const float* gamma, const float* beta, const float* film, long ldf,
                     const int64_t* lengths, const int64_t* skip_lengths, void* ds, void* dx_pre,
                     void* dx_pre_lp /* optional bf16 copy of dx_pre (of ds when p_pre == 0) */, int d_dtype,
                     float* dgamma, float* dbeta, float* dfilm, long lddf, int B, int N, int C, float p_pre, uint64_t seed_pre,
                     float p_post, uint64_t seed_post, int relu_input,
                     float* ws /* NULL: fp32 atomics; else dx_layernorm_bwd_ws_floats(B,N,C) floats: deterministic two-stage reduction */,
                     const DxStepScalars* step, void* stream);
long dx_layernorm_bwd_ws_floats(int B, int N, int C);

/* ---- K4: multi-head self-attention with key-padding mask, flash-style on MFMA (d_head 16 and 64: the tuned kernels of the
 * published 8- / 2-head configurations; 32 and 128 -- 4 heads / 1 head of a 128-wide model -- run the same templates untuned).
 * Replaces nn.MultiheadAttention's core (model.py:182-186): S = (q/sqrt(d)) k^T, pad keys -> -inf, softmax,
 * dropout on the probabilities, P v -- without materialising (B, H, N, N).
 *   qkv (B, N, 3E) = in-projection output [q | k | v], dtype = MFMA operand type (DX_BF16 or DX_F32)
 *   o   (B, N, E) same dtype;  lse (B, H, N) fp32 log-sum-exp per query (NULL at inference)
 * Queries n >= lengths[b] are not attended (their rows are zeroed by the following masked LayerNorm).
 * order: NULL, or the (B) int32 table of dx_length_order -- workgroups are then launched longest utterance first
 * (same results; a ragged batch finishes sooner). */
int dx_attention_fwd(const void* qkv, int dtype, const int64_t* lengths, const int* order, void* o, float* lse, int B, int N,
                     int H, int E, float p_drop, uint64_t seed, const DxStepScalars* step, void* stream);

/* Backward of dx_attention_fwd (autograd of model.py:182-186): dqkv (B, N, 3E) <- d_o (B, N, E).  delta_ws: (B, H, N) fp32 workspace.
 * algo: DX_ATTN_AUTO picks the fused kernel where it exists (bf16, d_head 16, N <= 1024: one workgroup per (utterance, head)
 * recomputes S / dP once for dQ, dK and dV) and the two-pass pair (dQ kernel, dK/dV kernel) elsewhere; DX_ATTN_TWO_PASS forces the
 * pair; DX_ATTN_FUSED returns DX_ERR_UNSUPPORTED where the fused kernel does not apply.  Both draw the forward's dropout mask.
 * delta_ws: dx_attention_bwd_ws_floats(B, N, H) floats of scratch (delta + the fused kernel's dQ partials); nothing in it has to be
 * initialised or to survive the call, so one grow-only buffer per stream serves every batch shape.
 * counters: dx_attention_bwd_counters(B, H) ints, the arrival counters of the fused kernel (an utterance of more than 512 keys is
 * shared by two workgroups): zero before the FIRST launch, and every launch puts the counters it used back to zero (the second
 * arrival of a pair resets its counter), so the caller zeroes a buffer of max(B * H) ints once per stream and keeps it.  May be
 * NULL when the fused kernel cannot run (fp32 operands, d_head 64, N > 1024, or DX_ATTN_TWO_PASS).  Calls that may run
 * CONCURRENTLY (different streams) must not share either buffer. */
long dx_attention_bwd_ws_floats(int B, int N, int H);
long dx_attention_bwd_counters(int B, int H);
enum { DX_ATTN_AUTO = 0, DX_ATTN_TWO_PASS = 1, DX_ATTN_FUSED = 2 };
int dx_attention_bwd(const void* qkv, const void* o, const void* d_o, int dtype, const float* lse,
                     const int64_t* lengths, const int* order, void* dqkv, float* delta_ws, int* counters, int B, int N, int H, int E,
                     float p_drop, uint64_t seed, const DxStepScalars* step, int algo, void* stream);

/* order[r] = index of the utterance with the r-th largest length (ties: lower index first), B <= 65536.  Host-side
 * analogue: the reference's collate sorts a batch by decreasing phoneme count (data_loader.py:246-250); the attention
 * kernels only use it as a launch order, so any batch order stays valid. */
int dx_length_order(const int64_t* lengths, int B, int* order, void* stream);

/* ---- K6: out = base + sum_f conv1d(1 -> 128, k = taps)(feat_f) + pos_table[n], zero where n >= lengths[b].
 * Energy / pitch embeddings + positional add + mask of the prosody encoder (model.py:400-414); the duration /
 * energy / pitch projections of the upsampler (model.py:618-628).  feats / ws / biases are HOST arrays of
 * nfeat (<= 3) device pointers: feat (B, N), w (128, 1, taps), bias (128).  base, pos_table, lengths may be NULL.
 * taps (ABI v13): 3 (the published conv_kernel) or 1. */
int dx_scalar_embed_fwd(const float* base, const float* const* feats, const float* const* ws,
                        const float* const* biases, int nfeat, const float* pos_table, const int64_t* lengths,
                        float* out, int B, int N, int C, int taps, void* stream);
/* dbase = dout * mask (may be NULL); dws / dbiases are accumulated with atomics. */
int dx_scalar_embed_bwd(const float* dout, const float* const* feats, int nfeat, const int64_t* lengths,
                        float* dbase, float* const* dws, float* const* dbiases, int B, int N, int C, int taps, void* stream);

/* ---- K7/K8: out[b, n] = table[ids[b, n]] + pos_table[n] for n < lengths[b], else 0 (model.py:497-504; the
 * positional gather replaces the host loops of PositionalEncoding.forward, model.py:132-150). */
int dx_embed_pos_fwd(const int64_t* ids, const float* table, const float* pos_table, const int64_t* lengths,
                     float* out, int B, int N, int C, void* stream);
int dx_embed_pos_bwd(const int64_t* ids, const float* dout, const int64_t* lengths, float* dtable, int B, int N, int C,
                     void* stream);

/* ---- K9: out[b] = sum_n x[b, n] / lengths[b] (model.py:419) and its backward. */
int dx_masked_mean_fwd(const float* x, const int64_t* lengths, float* out, int B, int N, int C, void* stream);
int dx_masked_mean_bwd(const float* dy, const int64_t* lengths, float* dx, int B, int N, int C, void* stream);

/* ---- K9: FiLM assembly (model.py:430-462).  g_raw / b_raw (B, W) are the gammas / betas projections, W = sum_m
 * nb[m] * ch[m] over the 3 FiLM-ed modules (encoder, predictor, decoder; nb / ch are HOST int[3]); post (2, sum nb)
 * or NULL.  film_m (B, nb[m], 2 ch[m]) = [post_g * g + 1 | post_b * b]. */
int dx_film_assemble_fwd(const float* g_raw, const float* b_raw, const float* post, float* film_enc, float* film_pp,
                         float* film_dec, const int* nb, const int* ch, int B, void* stream);
int dx_film_assemble_bwd(const float* g_raw, const float* b_raw, const float* post, const float* dfilm_enc,
                         const float* dfilm_pp, const float* dfilm_dec, float* dg_raw, float* db_raw, float* dpost,
                         const int* nb, const int* ch, int B, void* stream);

/* ---- K10/K14: exact-fp32 nn.Linear for the small heads (VALU, no operand rounding): FiLM projections
 * (model.py:427-428), speaker classifier (276-283), prosody projection 256 -> 3 (568-569), range projection
 * (634).  x (M, K), w (O, K), y (M, O); rows with (m % N) >= mask_lengths[m / N] are zero when mask_lengths != NULL.
 * Backward: dx = (dy * relu'(y)) W * dx_scale (NULL to skip; dx_scale = -lambda implements the gradient
 * reversal of model.py:34-38); dw / db accumulated with atomics. */
int dx_linear_small_fwd(const float* x, const float* w, const float* bias, float* y, const int64_t* mask_lengths,
                        int N, long M, int K, int O, int relu, void* stream);
int dx_linear_small_bwd(const float* dy, const float* y, const float* x, const float* w, float* dx, float dx_scale,
                        float* dw, float* db, const int64_t* mask_lengths, int N, long M, int K, int O, int relu,
                        void* stream);

/* out[b] = a[b] + table[ids[b]] (speaker-embedding add, model.py:423-424); backward scatter-adds dz into dtable. */
int dx_gather_add_fwd(const float* a, const float* table, const int64_t* ids, float* out, int B, int C, void* stream);
int dx_gather_add_bwd(const float* dz, const int64_t* ids, float* dtable, int B, int C, void* stream);

/* The two per-utterance heads behind the prosody embedding as single launches (C = 128).
 * dx_film_head_fwd = dx_gather_add_fwd + 2 x dx_linear_small_fwd + dx_film_assemble_fwd (model.py:419-462): z = emb + spk_table[ids]
 * (B, C), g_raw / b_raw = z W^T + bias (B, W), the three FiLM tensors assembled; bit-identical to the separate launches.
 * dx_film_head_bwd: from dfilm_* -- (dg_raw, db_raw) into the two (B, W) scratch buffers, dpost +=, d_emb += dz and d_spk_table[ids] += dz
 * (fp32 atomics), then the weight / bias gradients of both projections (+=, one thread per element: deterministic).
 * dx_classifier_fwd / _bwd (model.py:27-54, 276-292): three linear layers with ReLU; the backward WRITES d_emb = -lambda * (...) (the
 * gradient reversal), stores the gated gradients of layers 1 / 2 in two (B, C) scratch buffers and accumulates the six parameter
 * gradients.  S = number of logits (<= 128). */
int dx_film_head_fwd(const float* emb, const float* spk_table, const int64_t* spk_ids, const float* wg, const float* bg, const float* wb,
                     const float* bb, const float* post, float* z, float* g_raw, float* b_raw, float* film_enc, float* film_pp,
                     float* film_dec, const int* nb, const int* ch, int B, int C, void* stream);
int dx_film_head_bwd(const float* g_raw, const float* b_raw, const float* post, const float* z, const int64_t* spk_ids, const float* wg,
                     const float* wb, const float* dfilm_enc, const float* dfilm_pp, const float* dfilm_dec, float* dg_raw_ws,
                     float* db_raw_ws, float* d_emb, float* d_spk_table, float* dpost, float* dwg, float* dbg, float* dwb, float* dbb,
                     const int* nb, const int* ch, int B, int C, void* stream);
int dx_classifier_fwd(const float* emb, const float* w1, const float* b1, const float* w2, const float* b2, const float* w3,
                      const float* b3, float* h1, float* h2, float* logits, int B, int C, int S, void* stream);
int dx_classifier_bwd(const float* d_logits, const float* emb, const float* h1, const float* h2, const float* w1, const float* w2,
                      const float* w3, float* g1_ws, float* g2_ws, float* d_emb, float lambda, float* dw1, float* db1, float* dw2,
                      float* db2, float* dw3, float* db3, int B, int C, int S, void* stream);
int dx_add_inplace(float* dst, const float* src, long n, void* stream);           /* dst += src */
int dx_colsum(const void* x, int dtype, float* out, long rows, int C, void* stream); /* out[c] += sum_r x[r][c] (bias grads) */
int dx_scale(float* x, long n, float s, void* stream);
/* Layout helpers (the reference gets these from ATen views / copies around its modules):
 *   dx_transpose_last2: fp32 (B, R, C) -> (B, C, R) in y_dtype (DX_F32, or DX_BF16 = the rounding the first pre-net GEMM applies
 *   at operand load) -- the mel batch arrives as (B, n_mel, T) (model.py:744), kernels want rows;
 *   dx_unstack / dx_stack: y (M, K) interleaved <-> K planes of M floats, K <= 4 -- the duration / energy / pitch heads share one
 *   projection (model.py:567-575);  planes = host array of K device pointers;
 *   dx_fill_zero: stream-ordered memset of a device buffer. */
int dx_transpose_last2(const float* x, void* y, int y_dtype, int B, int R, int C, void* stream);
int dx_unstack(const float* y, float* const* planes, long M, int K, void* stream);
int dx_stack(float* y, const float* const* planes, long M, int K, void* stream);
int dx_fill_zero(void* p, size_t bytes, void* stream);
/* an empty single-thread launch: inside a stream capture, the first node recorded behind a fork decides which branch the graph
 * executor keeps on the forking node's queue -- the capturing trainer records one on the launch stream right behind every fork to
 * the weight-gradient stream (train.CapturedStep) */
int dx_anchor(void* stream);
/* one wave that occupies its hardware queue for `microseconds` (<= 100 000): HIP maps streams onto a few hardware queues
 * (GPU_MAX_HW_QUEUES, 4 by default) and two streams on the SAME queue run in submission order -- the weight-gradient stream, the
 * optimizer stream and RCCL's stream only overlap with the launch stream when they sit on other queues.  daft_exprt.streams probes
 * that with this kernel and picks streams that do (no reference counterpart: torch / DDP leave it to chance). */
int dx_spin(int microseconds, void* stream);

/* ---- K11: Gaussian upsampling (GaussianUpsamplingModule.forward, model.py:608-662), fp32, integer prefix sums exact.
 * dx_gu_prepare : xp = enc + conv(energy) + conv(pitch); rin = xp + conv(dur_float); r_pre = w_range . rin + b_range;
 *                 ranges = softplus(r_pre), 1 at l >= in_lengths[b].  r_pre / rin may be NULL (inference).
 * dx_gu_means   : means[l] = float(d[l]) / 2 + float(sum_{j<l} d[j]) from int64 durations; totals[b] = sum_l d.
 * dx_gu_upsample_fwd : weights (B, L, T) and out (B, T, 128).  With pos_table != NULL the decoder's positional add +
 *                 mask (model.py:696-701) is applied: out = (x_up + pos) where t < out_lengths[b], else 0. */
int dx_gu_prepare(const float* enc, const float* dur_float, const float* energy, const float* pitch,
                  const int64_t* in_lengths, const float* w_dur, const float* b_dur, const float* w_en,
                  const float* b_en, const float* w_pi, const float* b_pi, const float* w_range,
                  const float* b_range, float* xp, float* ranges, float* r_pre, float* rin, int B, int L,
                  int C, void* stream);
int dx_gu_means(const int64_t* durations_int, float* means, int64_t* totals, int B, int L, void* stream);
int dx_gu_upsample_fwd(const float* xp, const float* ranges, const float* means, const int64_t* in_lengths,
                       const int64_t* out_lengths, const float* pos_table, float* weights, float* out, int B,
                       int L, int T, int C, void* stream);
/* g = grad wrt `out`.  Outputs: dxp (grad wrt xp incl. the range path), drin (grad wrt rin), dr (B, L) grad wrt r_pre.
 * dw_ws (B, L, T) and dsum_ws (B, T) are fp32 workspaces. */
int dx_gu_upsample_bwd(const float* g, const float* xp, const float* weights, const float* means,
                       const float* ranges, const float* r_pre, const float* w_range,
                       const int64_t* in_lengths, const int64_t* out_lengths, float* dw_ws, float* dsum_ws,
                       float* dxp, float* drin, float* dr, int B, int L, int T, int C, void* stream);

/* ---- K13: the 7-term training loss and its gradients in one pass (DaftExprtLoss.forward, loss.py:54-99).
 * terms (8 floats, device): speaker, post_mult, duration, energy, pitch, mel_l1, mel_l2 (weighted), total.
 * mel / mel_t are (B, n_mel, T).  Gradient outputs (NULL to skip) are d(total * grad_scale)/d(pred);
 * d_post_mult is ACCUMULATED.  w_spk is the ramped adversarial weight (loss.py:22-28).
 * ws: NULL, or dx_loss_ws_floats(B, T) floats of scratch (no initialisation): with it (and the transposed mel gradient) the
 * per-workgroup terms are added in a fixed order by the last of THREE launches -- run-to-run identical loss terms, no same-address
 * atomics; without it four launches and fp32 atomics on terms[2..6]. */
long dx_loss_ws_floats(int B, int T);
int dx_loss_fwd_bwd(const float* dur, const float* energy, const float* pitch, const float* dur_t,
                    const float* energy_t, const float* pitch_t, const int64_t* in_lengths, const float* mel,
                    const float* mel_t, const int64_t* out_lengths, const float* spk_logits,
                    const int64_t* spk_ids, const float* post_mult, float* d_dur, float* d_energy,
                    float* d_pitch, float* d_mel, float* d_spk_logits, float* d_post_mult, float* terms, float* ws,
                    int B, int L, int T, int n_mel, int n_spk_classes, int n_post, float w_spk, float w_post,
                    float w_dur, float w_energy, float w_pitch, float w_mel, float grad_scale,
                    int d_mel_transposed /* write d_mel as (B, T, n_mel) */,
                    const DxStepScalars* step /* NULL, or the step block: its w_speaker replaces w_spk */, void* stream);

/* ---- K15: torch.optim.Adam as configured at train.py:299-301 (coupled L2, bias correction, amsgrad off) over a
 * flat fp32 buffer; clip_grad_norm_ (train.py:399) folded in: grad_norm_sq (device scalar from dx_sumsq) and
 * clip_thresh (INFINITY = log only). */
int dx_sumsq(const float* x, long n, float* out, void* stream);
int dx_adam_step(float* p, const float* g, float* m, float* v, long n, float lr, float beta1, float beta2,
                 float eps, float weight_decay, int step, const float* grad_norm_sq, float clip_thresh,
                 float* grad_norm_sq_accum /* NULL, or a device scalar that receives += sum(g^2) of this slice: the norm the
                 trainer logs (train.py:399 with an infinite threshold) without a pass of its own; per-section calls add up */,
                 const DxStepScalars* scalars /* NULL, or the step block: its lr / bias corrections replace `lr` and `step` */,
                 void* stream);

/* dx_adam_step over the whole flat buffer FUSED with the refresh of the MFMA operand copies of the GEMM weights (what
 * dx_pack_conv_weights_batched + dx_pack_frag_major_batched do after every optimizer step): one launch, the weights are read once.
 * bricks_dev: DEVICE array of n_weights records {long off; void* fwd; void* tr; void* frag_fwd; void* frag_tr; int Cout, Cin, taps,
 * pad; long begin;} (dx_adam_pack_desc_size() bytes): off = offset of the fp32 weight (Cout, Cin, taps) in p / g / m / v; fwd =
 * [taps][Cout][Cin], tr = [taps][Cin][Cout] with flipped taps (dx_pack_conv_weight layouts), frag_* = dx_pack_frag_major of those
 * (bf16, taps = 3, Cout and Cin multiples of 32), any of them NULL; begin = running count of 32 x 32 bricks, total_bricks their sum.
 * flats_dev: DEVICE array of n_flats records {long off, n, begin;} (dx_adam_flat_desc_size() bytes) covering every parameter that is
 * not a GEMM weight; begin = running count of dx_adam_flat_block()-element blocks, total_flat_blocks their sum.  out_dtype = dtype of
 * the copies (DX_BF16 / DX_F32).  Same arithmetic per element as dx_adam_step with an infinite clipping threshold; grad_norm_sq_accum
 * and scalars as there. */
int dx_adam_pack_desc_size(void);
int dx_adam_flat_desc_size(void);
int dx_adam_flat_block(void);
int dx_adam_pack_step(float* p, const float* g, float* m, float* v, const void* bricks_dev, int n_weights, long total_bricks,
                      const void* flats_dev, int n_flats, long total_flat_blocks, int out_dtype, float lr, float beta1, float beta2,
                      float eps, float weight_decay, int step, float* grad_norm_sq_accum, const DxStepScalars* scalars, void* stream);

/* ---- K16: float -> integer frame durations on the device (DaftExprt.get_int_durations model.py:789-812 +
 * duration_to_integer extract_features.py:69-111): optional dur_factors multiply (model.py:891), in-place
 * thresholding at filter_length / sampling_rate / 2, fp64 cumulative times, sample/frame counting in integers.
 * status[b]: 0 ok, 1 = the reference would raise IndexError (utterance shorter than one analysis window),
 * 2 = the reference's scatter would fail (symbol / duration count mismatch). */
int dx_int_durations(float* duration_preds, const float* dur_factors, int64_t* durations_int, int64_t* totals, int* status, int B, int L,
                     double sampling_rate, int filter_length, int hop_length, int centered, void* stream);

/* Inference-time prosody control (model.py:895-905, pitch_shift 814-834, pitch_multiply 836-864), in place.
 * mode 0 = 'add', 1 = 'multiply'. */
int dx_prosody_control(float* energy, float* pitch, const float* energy_factors, const float* pitch_factors,
                       const int64_t* durations_int, const int64_t* speaker_ids, const float* spk_pitch_mean,
                       const float* spk_pitch_std, int mode, int B, int L, void* stream);

/* ---- K17: mel / energy front-end of the synthesis path (extract_features.py:330-359 `mel_spectrogram_HiFi`,
 * 299-304 `extract_energy` as applied at generate.py:457 and extract_features.py:465-466).
 *   mel[b, m, f]   = log(max(sum_k fb[m, k] * sqrt(re^2 + im^2 + 1e-9), min_clip)),  X = STFT(wav[b], n_fft, hop, Hann,
 *                    center / reflect padding as torch.stft)            frames past the utterance: zeros
 *   energy[b, f]   = || exp(mel[b, :, f]) ||_2;      n_frames[b] = centered ? 1 + n / hop : 1 + (n - n_fft) / hop
 * wav (B, ldw) fp32, n_samples (B) int64.  twiddle (2 * n_fft floats) / window (n_fft floats): device tables filled
 * once by dx_mel_tables.  fb (n_mel, n_fft/2 + 1) dense filterbank with the non-zero bin range [fb_lo[m], fb_hi[m]) of
 * every filter.  mel (B, n_mel, T), energy (B, T).  T >= max n_frames.  n_fft in {256, 1024, 4096} (radix-4 FFT in LDS,
 * one workgroup per frame). */
int dx_mel_tables(float* twiddle, float* window, int n_fft, void* stream);
int dx_mel_spectrogram(const float* wav, long ldw, const int64_t* n_samples, const float* twiddle, const float* window,
                       const float* fb, const int* fb_lo, const int* fb_hi, float* mel, float* energy,
                       int64_t* n_frames, int B, int T, int n_fft, int hop, int n_mel, int centered, float min_clip,
                       void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DAFT_EXPRT_HIP_H */
