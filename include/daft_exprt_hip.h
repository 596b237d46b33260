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
 */
#ifndef DAFT_EXPRT_HIP_H
#define DAFT_EXPRT_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DX_ABI_VERSION 1

enum { DX_F32 = 0, DX_BF16 = 1, DX_I64 = 2 };
enum { DX_OK = 0, DX_ERR_ARG = -1, DX_ERR_SHAPE = -2, DX_ERR_DTYPE = -3, DX_ERR_LAUNCH = -4, DX_ERR_UNSUPPORTED = -5 };

/* flags of dx_conv1d */
enum { DX_CONV_RELU = 1, DX_CONV_TRANSPOSED_OUT = 2 };

int dx_abi_version(void);
const char* dx_last_error(void);

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
 * Positions outside [0, N) are zero padding (N = the batch's max length).  Cin % 8 == 0.
 */
int dx_conv1d(const void* x, int x_dtype, long ldx, const void* w_packed, int w_dtype, const float* bias,
              void* y, int y_dtype, long ldy, const void* relu_gate, int gate_dtype, const int64_t* mask_lengths,
              int B, int N, int Cin, int Cout, int taps, int flags, void* stream);

/* Pack an fp32 (Cout, Cin, taps) PyTorch conv / (Cout, Cin) linear weight for dx_conv1d.
 *   transpose_flip = 0: out[tap][co][ci] = w[co][ci][tap]                (forward operand)
 *   transpose_flip = 1: out[tap][ci][co] = w[co][ci][taps-1-tap]         (data-gradient operand) */
int dx_pack_conv_weight(const float* w, void* out, int out_dtype, int Cout, int Cin, int taps, int transpose_flip,
                        void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DAFT_EXPRT_HIP_H */
