"""Thin Python wrappers over the C ABI: allocate outputs with torch (device memory plumbing),
pass raw pointers + the current stream.  One function per kernel family."""
import torch

from daft_exprt import _hip as H


def pack_conv_weight(w, dtype, transpose_flip=False):
    ''' fp32 (Cout, Cin, taps) or (Cout, Cin) -> packed MFMA operand, see dx_pack_conv_weight '''
    H.require_gpu(w)
    w = w.contiguous()
    cout, cin = w.shape[0], w.shape[1]
    taps = w.shape[2] if w.dim() == 3 else 1
    shape = (taps, cin, cout) if transpose_flip else (taps, cout, cin)
    out = torch.empty(shape, dtype=dtype, device=w.device)
    H.check(H.lib().dx_pack_conv_weight(H.ptr(w), H.ptr(out), H.dt(out), cout, cin, taps, int(transpose_flip), H.stream()))
    return out


def conv1d(x, w_packed, bias=None, out_dtype=None, relu=False, relu_gate=None, mask_lengths=None,
           transposed_out=False, out=None):
    ''' x (B, N, Cin) [last dim contiguous]; w_packed (taps, Cout, Cin) -> (B, N, Cout) or (B, Cout, N) '''
    H.require_gpu(x, w_packed)
    B, N, Cin = x.shape
    taps, Cout, Cin_w = w_packed.shape
    assert Cin_w == Cin, (Cin_w, Cin)
    assert x.stride(2) == 1 and x.stride(0) == N * x.stride(1)
    out_dtype = out_dtype or x.dtype
    if out is None:
        out = torch.empty((B, Cout, N) if transposed_out else (B, N, Cout), dtype=out_dtype, device=x.device)
    flags = (H.CONV_RELU if relu else 0) | (H.CONV_TRANSPOSED_OUT if transposed_out else 0)
    ldy = out.stride(1)
    H.check(H.lib().dx_conv1d(H.ptr(x), H.dt(x), x.stride(1), H.ptr(w_packed), H.dt(w_packed), H.ptr(bias),
                              H.ptr(out), H.dt(out), ldy, H.ptr(relu_gate), H.dt(relu_gate) if relu_gate is not None else 0,
                              H.ptr(mask_lengths), B, N, Cin, Cout, taps, flags, H.stream()))
    return out
