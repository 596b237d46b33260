"""Thin Python wrappers over the C ABI (include/daft_exprt_hip.h): allocate outputs with torch
(device-memory plumbing only), pass raw pointers + the current HIP stream.  One function per entry point."""
import ctypes

import torch

from daft_exprt import _hip as H
from daft_exprt import config

_INF = float('inf')
DETERMINISTIC_LN = False
WGRAD_WORKSPACE = True   # False: fp32 atomics on dW instead of partial tiles + fixed-order reduce (tests compare the two)

# Device pointer of the step block (DxStepScalars, include/daft_exprt_hip.h) while a step is being CAPTURED into a hipGraph
# (`train.CapturedStep`): the dropout kernels add its salt to their by-value seeds, Adam and the loss read this iteration's learning
# rate / bias corrections / adversarial weight from it.  None = eager launches, every scalar by value.
STEP_PTR = None

# Optional per-kernel timing probe used by bench.py: {family: [(start_event, end_event, padded_flops, N), ...]}.
# Events are recorded on torch's current stream, which is the stream every kernel is launched on.
PROBE = None


class _Probe(object):
    def __init__(self, family, flops, n_axis):
        self.family, self.flops, self.n_axis = family, flops, n_axis

    def __enter__(self):
        if PROBE is not None:
            self.start = torch.cuda.Event(enable_timing=True)
            self.start.record()

    def __exit__(self, *exc):
        if PROBE is not None:
            end = torch.cuda.Event(enable_timing=True)
            end.record()
            PROBE.setdefault(self.family, []).append((self.start, end, self.flops, self.n_axis))


class _NoProbe(object):
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NO_PROBE = _NoProbe()


def _probe(family, flops_fn, n_axis):
    ''' timing probe of bench.py when armed, a shared no-op otherwise (the step makes ~150 of these calls and its
        phoneme-level stretches are host-bound) '''
    return _NO_PROBE if PROBE is None else _Probe(family, flops_fn(), n_axis)


def _empty(*shape, **kw):
    ''' torch.empty -- or, with DX_POISON=1, a buffer pre-filled with NaN (floating types) / a large negative number (integers): the
        parity tests then prove that no kernel reads a row its producer did not write (dead rows past dx_fill_end stay unwritten) '''
    t = torch.empty(*shape, **kw)
    if config.POISON and t.is_cuda:
        if t.is_floating_point():
            t.fill_(float('nan'))
        elif t.dtype in (torch.int32, torch.int64):
            t.fill_(-(1 << 30))
    return t


def _empty_like(x, **kw):
    return _empty(x.shape, dtype=kw.get('dtype', x.dtype), device=x.device)


def _ptr_array(tensors):
    arr = (ctypes.c_void_p * max(1, len(tensors)))()
    for i, t in enumerate(tensors):
        arr[i] = t.data_ptr()
    return arr


def _int_array(vals):
    return (ctypes.c_int * len(vals))(*vals)


# ----------------------------------------------------------------------------- conv / linear on MFMA
def pack_conv_weight(w, dtype, transpose_flip=False, out=None):
    ''' fp32 (Cout, Cin, taps) or (Cout, Cin) -> packed MFMA operand, see dx_pack_conv_weight '''
    H.require_gpu(w)
    cout, cin = w.shape[0], w.shape[1]
    taps = w.shape[2] if w.dim() == 3 else 1
    shape = (taps, cin, cout) if transpose_flip else (taps, cout, cin)
    if out is None:
        out = _empty(shape, dtype=dtype, device=w.device)
    assert w.is_contiguous()
    H.check(H.lib().dx_pack_conv_weight(H.ptr(w), H.ptr(out), H.dt(out), cout, cin, taps, int(transpose_flip), H.stream()))
    return out




def relu_bits_ok(x, w_packed, out_dtype=None):
    ''' the shapes `conv1d(..., relu_bits=True)` / a bit-mask `relu_gate` take: the register-weights kernel (dx_conv1d_relu_bits) '''
    taps, Cout, Cin = w_packed.shape
    return (x.is_cuda and taps == 3 and Cin == 128 and Cout % 256 == 0 and x.dtype == torch.bfloat16 and w_packed.dtype == torch.bfloat16
            and (out_dtype or x.dtype) == torch.bfloat16 and x.stride(1) % 8 == 0)


def conv1d(x, w_packed, bias=None, out_dtype=None, relu=False, relu_gate=None, mask_lengths=None,
           transposed_out=False, out=None, accumulate=False, skip_lengths=None, w_frag=None, wide_plan=None, relu_bits=False):
    ''' x (B, N, Cin) [last dim contiguous]; w_packed (taps, Cout, Cin) -> (B, N, Cout) or (B, Cout, N).
        relu_bits (with relu, shapes of `relu_bits_ok`): returns (y, bits) -- bits = int32 (B, Cout / 32, N), one bit per element of
        y > 0; a later call passes it as `relu_gate` (an int32 tensor instead of the activation) and reads 1 / 16 of the gate bytes.
        w_frag + wide_plan: the same weights in fragment order (pack_frag_major) and the balanced tiles of the batch
        (conv_tile_plan(skip_lengths, N, halo=2, round_to=64)): the wide k = 3 GEMMs (bf16 in / out, Cin % 128 == 0,
        Cout % 256 == 0, nothing but bias / ReLU in the epilogue) then run on dx_conv1d_wide; w_frag alone with Cin = 128, k = 3:
        the register-weights kernel loads its fragments from it (dx_conv1d_wfrag) '''
    H.require_gpu(x, w_packed)
    B, N, Cin = x.shape
    taps, Cout, Cin_w = w_packed.shape
    assert Cin_w == Cin, (Cin_w, Cin)
    assert x.stride(2) == 1 and (B == 1 or x.stride(0) == N * x.stride(1))
    out_dtype = out_dtype or x.dtype
    if (w_frag is not None and wide_plan is not None and taps == 3 and x.dtype == torch.bfloat16 and out_dtype == torch.bfloat16
            and out is None and relu_gate is None and mask_lengths is None and skip_lengths is not None and not transposed_out
            and Cin % 128 == 0 and Cin >= 256 and Cout % 256 == 0):
        table, pb, pn = wide_plan
        assert (pb, pn) == (B, N), 'tile plan built for another batch geometry'
        y = _empty((B, N, Cout), dtype=torch.bfloat16, device=x.device)
        with _probe('conv_gemm', lambda: 2. * B * N * Cin * Cout * taps, N):
            H.check(H.lib().dx_conv1d_wide(H.ptr(x), x.stride(1), H.ptr(w_frag), H.ptr(bias), H.ptr(y), y.stride(1), H.ptr(skip_lengths),
                                           H.ptr(table), table.shape[0], 2, B, N, Cin, Cout, H.CONV_RELU if relu else 0, H.stream()))
        return y
    gate_bits = relu_gate is not None and relu_gate.dtype == torch.int32
    if relu_bits or gate_bits:
        assert relu_bits_ok(x, w_packed, out_dtype) and out is None and not transposed_out and not accumulate and relu == bool(relu_bits) and \
            (relu_gate is None or gate_bits), 'relu_bits / bit-mask gate: bf16 128 -> 256 k channels, k = 3, plain output'
        y = _empty((B, N, Cout), dtype=torch.bfloat16, device=x.device)
        bits = relu_gate if gate_bits else _empty((B, Cout // 32, N), dtype=torch.int32, device=x.device)
        assert tuple(bits.shape) == (B, Cout // 32, N) and bits.is_contiguous()
        frag = w_frag if (w_frag is not None and w_frag.dtype == torch.bfloat16 and w_frag.numel() == w_packed.numel()) else None
        with _probe('conv_gemm', lambda: 2. * B * N * Cin * Cout * taps, N):
            H.check(H.lib().dx_conv1d_relu_bits(H.ptr(x), x.stride(1), H.ptr(w_packed), H.ptr(frag), H.ptr(bias), H.ptr(y), y.stride(1),
                                                None if gate_bits else H.ptr(bits), H.ptr(bits) if gate_bits else None,
                                                H.ptr(mask_lengths), H.ptr(skip_lengths), B, N, Cout, H.stream()))
        return (y, bits) if relu_bits else y
    if out is None:
        assert not accumulate
        out = _empty((B, Cout, N) if transposed_out else (B, N, Cout), dtype=out_dtype, device=x.device)
    flags = (H.CONV_RELU if relu else 0) | (H.CONV_TRANSPOSED_OUT if transposed_out else 0) | (4 if accumulate else 0)
    frag = None
    if w_frag is not None and taps == 3 and Cin == 128 and w_packed.dtype == torch.bfloat16:
        assert w_frag.dtype == torch.bfloat16 and w_frag.numel() == w_packed.numel()
        frag = w_frag                                 # register-weights kernel: fragments straight into registers (dx_conv1d_wfrag)
    with _probe('conv_gemm', lambda: 2. * B * N * Cin * Cout * taps, N):
      H.check(H.lib().dx_conv1d_wfrag(H.ptr(x), H.dt(x), x.stride(1), H.ptr(w_packed), H.dt(w_packed), H.ptr(frag), H.ptr(bias),
                                    H.ptr(out), H.dt(out), out.stride(1), H.ptr(relu_gate),
                                    H.dt(relu_gate) if relu_gate is not None else 0,
                                    H.ptr(mask_lengths), H.ptr(skip_lengths), B, N, Cin, Cout, taps, flags, H.stream()))
    return out


def splitk_ln_ok(B, N, x_dtype, w_packed, plan, w_frag):
    ''' does conv1d_ln on a (B, N, Cin) input of x_dtype with these weights run on the split-K workgroups (the path that can take
        `residual_ln`)?  Mirrors the gate of `launch_taps` in csrc/conv_gemm.hip '''
    Cin = w_packed.shape[2]
    return (USE_SPLITK and plan is not None and w_frag is not None and x_dtype == torch.bfloat16 and w_packed.dtype == torch.bfloat16
            and w_packed.shape[0] == 3 and Cin >= 256 and Cin % 128 == 0 and B * N <= 65536 and tuple(plan[1:]) == (B, N))


def conv1d_ln(x, w_packed, bias, residual, gamma, beta, lengths, film=None, save=False, p_pre=0., seed_pre=0, lp_copy=False, plan=None,
              w_frag=None, w2_packed=None, b2=None, store_y=True, residual_ln=None):
    ''' conv / linear to 128 channels with the following LayerNorm (+dropout, residual, FiLM, mask) fused into the epilogue.
        plan: conv_tile_plan(lengths, N) of this batch (bf16 k = 3 GEMMs only); w_frag: the weights in fragment order
        (pack_frag_major) -> split-K workgroups on the plan's tiles.  Returns (y, y_lp, s_out, mean, rstd) -- and a sixth element with
        w2_packed (bf16 (1, n2, 128), n2 = 128 or 384; b2 fp32 (n2) or None): y_lp . w2^T + b2 (bf16), the k = 1 projection that reads
        this output next, from the same launch when the split-K path takes it, else None (the caller launches it).
        store_y = False (with lp_copy): the fp32 output is not stored (y = None) -- for a stream whose only fp32 reader re-derives it:
        residual_ln = (mean, rstd, gamma, beta) of the LayerNorm that produced the residual stream, `residual` then being that
        LayerNorm's saved INPUT (`splitk_ln_ok` shapes only, see dx_conv1d_ln_vres) '''
    B, N, Cin = x.shape
    taps, Cout, _ = w_packed.shape
    assert Cout == 128 and x.stride(2) == 1 and residual.is_contiguous()
    assert store_y or lp_copy
    dev = x.device
    y = _empty((B, N, 128), dtype=torch.float32, device=dev) if store_y else None
    y_lp = _empty((B, N, 128), dtype=torch.bfloat16, device=dev) if lp_copy else None
    s_out = _empty((B, N, 128), dtype=torch.float32, device=dev) if save else None
    mean = _empty(B * N, dtype=torch.float32, device=dev) if save else None
    rstd = _empty(B * N, dtype=torch.float32, device=dev) if save else None
    pargs = _plan_args(plan, x, w_packed, B, N, w_frag=w_frag)
    r_mean = r_rstd = r_gamma = r_beta = None
    if residual_ln is not None:
        assert splitk_ln_ok(B, N, x.dtype, w_packed, plan, w_frag), 'residual_ln: split-K path only'
        r_mean, r_rstd, r_gamma, r_beta = residual_ln
    y2, n2 = None, 0
    if (w2_packed is not None and lp_copy and pargs[2] is not None and taps == 3 and Cin % 128 == 0 and B * N <= 65536
            and w2_packed.dtype == torch.bfloat16 and w2_packed.shape[0] == 1 and w2_packed.shape[2] == 128 and w2_packed.shape[1] in (128, 384)):
        n2 = w2_packed.shape[1]
        y2 = _empty((B, N, n2), dtype=torch.bfloat16, device=dev)
    with _probe('conv_gemm', lambda: 2. * B * N * Cin * Cout * taps + 2. * B * N * 128 * n2, N):
        H.check(H.lib().dx_conv1d_ln_vres(H.ptr(x), H.dt(x), x.stride(1), H.ptr(w_packed), H.dt(w_packed), H.ptr(bias), H.ptr(residual),
                                          H.ptr(r_mean), H.ptr(r_rstd), H.ptr(r_gamma), H.ptr(r_beta),
                                          H.ptr(gamma), H.ptr(beta), H.ptr(film), film.stride(0) if film is not None else 0, H.ptr(lengths),
                                          H.ptr(y), H.ptr(y_lp), H.ptr(s_out), H.ptr(mean), H.ptr(rstd), B, N, Cin, taps, float(p_pre),
                                          int(seed_pre), *pargs, H.ptr(w2_packed if y2 is not None else None),
                                          H.ptr(b2 if y2 is not None else None), H.ptr(y2), n2, STEP_PTR, H.stream()))
    if w2_packed is not None:
        return y, y_lp, s_out, mean, rstd, y2
    return y, y_lp, s_out, mean, rstd


def conv1d_lnbwd(x, w_packed, y_inout, s_in, mean, rstd, gamma, beta, lengths, dgamma, dbeta, film=None, dfilm=None,
                 p_pre=0., seed_pre=0, plan=None, w_frag=None, w2_packed=None):
    ''' data gradient of a conv / linear into a 128-channel residual stream + the backward of the LayerNorm that consumed
        that stream, one launch (see dx_conv1d_lnbwd).  y_inout (B, N, 128) fp32: residual gradient in, ds out (in place).
        Returns the bf16 dropout_pre(ds) -- and, with w2_packed (bf16 (1, 128, 128): the packed weight of the 128 -> 128 linear map
        to apply to it), the pair (dropout_pre(ds), conv1d(dropout_pre(ds), w2_packed)): the second product comes out of the same
        launch when the split-K path takes it, else from a conv1d call.  dgamma / dbeta / dfilm accumulate. '''
    B, N, Cin = x.shape
    taps, Cout, _ = w_packed.shape
    assert Cout == 128 and x.stride(2) == 1 and y_inout.is_contiguous() and y_inout.dtype == torch.float32
    dx_lp = _empty((B, N, 128), dtype=torch.bfloat16, device=x.device)
    ldf = film.stride(0) if film is not None else 0
    lddf = dfilm.stride(0) if dfilm is not None else 0
    pargs = _plan_args(plan, x, w_packed, B, N, k1_ok=True, w_frag=w_frag)
    y2 = None
    if (w2_packed is not None and pargs[2] is not None and taps == 3 and Cin % 128 == 0 and B * N <= 65536
            and w2_packed.dtype == torch.bfloat16 and tuple(w2_packed.shape) == (1, 128, 128)):
        y2 = _empty((B, N, 128), dtype=torch.bfloat16, device=x.device)
    with _probe('conv_gemm', lambda: 2. * B * N * Cin * Cout * taps + (2. * B * N * 128 * 128 if y2 is not None else 0.), N):
        H.check(H.lib().dx_conv1d_lnbwd(H.ptr(x), H.dt(x), x.stride(1), H.ptr(w_packed), H.dt(w_packed), H.ptr(y_inout), H.ptr(s_in),
                                        H.ptr(mean), H.ptr(rstd), H.ptr(gamma), H.ptr(beta), H.ptr(film), ldf, H.ptr(lengths),
                                        H.ptr(dx_lp), H.ptr(dgamma), H.ptr(dbeta), H.ptr(dfilm), lddf, B, N, Cin, taps,
                                        float(p_pre), int(seed_pre), *pargs, H.ptr(w2_packed if y2 is not None else None), H.ptr(y2),
                                        STEP_PTR, H.stream()))
    if w2_packed is None:
        return dx_lp
    if y2 is None:
        y2 = conv1d(dx_lp, w2_packed, None, out_dtype=dx_lp.dtype, skip_lengths=lengths)
    return dx_lp, y2


def conv_tile_plan(lengths, N, halo=0, round_to=None, tiles=None):
    ''' balanced position tiles of one batch (dx_conv_tile_plan): build once per batch, pass as plan= to conv1d_ln / conv1d_lnbwd
        (halo 0: rows past the length are masked) or to the wide GEMMs of conv1d (halo 2: the pre-net convs compute two rows past
        the length; round_to 64: their 4 channel tiles x 64 position tiles fill the 256 CUs in whole rounds; tiles: an explicit
        tile count >= B * ceil(N / 256) -- fewer, taller tiles for a batch too small to give 256 workgroups a useful height).
        Returns (int32 table (n_tiles, 4) on the device, B, N) '''
    B = lengths.shape[0]
    if tiles is not None:
        n = max(int(tiles), B * ((N + 255) // 256))
    elif round_to is None:
        n = H.lib().dx_conv_tile_plan_size(B, N)
    else:
        worst = B * ((N + 255) // 256)
        n = (worst + round_to - 1) // round_to * round_to
    table = _empty((n, 4), dtype=torch.int32, device=lengths.device)
    H.check(H.lib().dx_conv_tile_plan(H.ptr(lengths), B, N, n, H.ptr(table), int(halo), H.stream()))
    return table, B, N


def batch_prep(lengths, N, plan=True, wide=True, order=True):
    ''' (conv_tile_plan(lengths, N), conv_tile_plan(lengths, N, halo=2, round_to=64), length_order(lengths)) from ONE launch
        (dx_batch_prep); an output that is switched off comes back as None '''
    B = lengths.shape[0]
    worst = B * ((N + 255) // 256)
    n0 = H.lib().dx_conv_tile_plan_size(B, N)
    n2 = (worst + 63) // 64 * 64
    dev = lengths.device
    t0 = _empty((n0, 4), dtype=torch.int32, device=dev) if plan else None
    t2 = _empty((n2, 4), dtype=torch.int32, device=dev) if wide else None
    od = _empty((B,), dtype=torch.int32, device=dev) if order else None
    H.check(H.lib().dx_batch_prep(H.ptr(lengths), B, N, n0, H.ptr(t0), n2, H.ptr(t2), H.ptr(od), H.stream()))
    return (t0, B, N) if plan else None, (t2, B, N) if wide else None, od


def _plan_args(plan, x, w_packed, B, N, k1_ok=False, w_frag=None):
    ''' (table pointer, n_tiles, fragment-order weights) when the plan applies to this GEMM (bf16 operands, k = 3, same batch
        geometry); the fragment-order copy only goes with a plan, k = 3 and Cin >= 256 '''
    taps = w_packed.shape[0]
    if plan is None or x.dtype != torch.bfloat16 or w_packed.dtype != torch.bfloat16 or x.shape[2] % 32 or \
            not (taps == 3 or (taps == 1 and k1_ok and x.shape[2] >= 256)):
        return None, 0, None
    table, pb, pn = plan
    assert (pb, pn) == (B, N), 'tile plan built for another batch geometry'
    frag = None
    if w_frag is not None and taps == 3 and x.shape[2] >= 256 and USE_SPLITK:
        assert w_frag.dtype == torch.bfloat16 and w_frag.numel() == w_packed.numel()
        frag = H.ptr(w_frag)
    return H.ptr(table), table.shape[0], frag


USE_SPLITK = True   # False: the LayerNorm-fused k = 3 GEMMs stay on the ring kernel (tests compare the two)


def pack_frag_major(w_packed, out=None):
    ''' fragment-order copy of a packed [3][Cout][Cin] bf16 weight (dx_pack_frag_major) '''
    taps, cout, cin = w_packed.shape
    assert taps == 3 and cout % 32 == 0 and cin % 32 == 0 and w_packed.dtype == torch.bfloat16 and w_packed.is_contiguous()
    out = _empty(w_packed.numel(), dtype=torch.bfloat16, device=w_packed.device) if out is None else out
    H.check(H.lib().dx_pack_frag_major(H.ptr(w_packed), H.ptr(out), cin, cout, H.stream()))
    return out


def frag_table(pairs, device):
    ''' device table for pack_frag_major_batched: pairs = [(w_packed [3][Cout][Cin], out)] '''
    import struct
    rec = H.lib().dx_frag_desc_size()
    buf = bytearray(rec * len(pairs))
    for i, (w, out) in enumerate(pairs):
        struct.pack_into('<QQii', buf, i * rec, w.data_ptr(), out.data_ptr(), w.shape[2], w.shape[1])
    table = torch.frombuffer(buf, dtype=torch.uint8).clone().to(device)
    return table, len(pairs), max(w.numel() for w, _ in pairs)


def pack_frag_major_batched(table, n, max_elems, stream=None):
    H.check(H.lib().dx_pack_frag_major_batched(H.ptr(table), n, max_elems, H.stream() if stream is None else stream))


def pack_table(entries, device):
    ''' entries: [(w fp32 tensor, out tensor, transpose_flip)] -> (device descriptor table, n, total_bricks) for
        pack_weights_batched (one launch for every GEMM weight of the model) '''
    import numpy as np
    dt = np.dtype([('w', '<u8'), ('out', '<u8'), ('Cout', '<i4'), ('Cin', '<i4'), ('taps', '<i4'), ('tf', '<i4'), ('begin', '<i8')])
    assert dt.itemsize == H.lib().dx_pack_desc_size()
    arr, begin = np.zeros(len(entries), dtype=dt), 0
    for i, (w, out, tf) in enumerate(entries):
        taps = w.shape[2] if w.dim() == 3 else 1
        arr[i] = (w.data_ptr(), out.data_ptr(), w.shape[0], w.shape[1], taps, int(tf), begin)
        begin += ((w.shape[0] + 31) // 32) * ((w.shape[1] + 31) // 32)   # 32 x 32 bricks, see dx_pack_conv_weights_batched
    table = torch.from_numpy(arr.view(np.uint8).copy()).to(device)
    return table, len(entries), begin


def pack_weights_batched(table, n, total, dtype, stream=None):
    ''' stream: raw hipStream_t to launch on (default: torch's current stream) '''
    H.check(H.lib().dx_pack_conv_weights_batched(H.ptr(table), n, total, H._DT[dtype], H.stream() if stream is None else stream))


def conv1d_wgrad(dy, x, dw, db, compute_dtype, lengths=None, stream=None, ws=None):
    ''' dw (Cout, Cin, taps) / (Cout, Cin) fp32 and db (Cout) are ACCUMULATED. dy (B,N,Cout), x (B,N,Cin).
        stream: raw hipStream_t to launch on (default: torch's current stream); ws: caller-owned scratch of at least
        `wgrad_ws_floats(...)` floats that stays valid until the launch has completed on that stream. '''
    B, N, Cout = dy.shape
    Cin = x.shape[2]
    taps = dw.shape[2] if dw.dim() == 3 else 1
    assert dw.shape[0] == Cout and dw.shape[1] == Cin and dy.stride(2) == 1 and x.stride(2) == 1
    # scratch for the per-workgroup partial tiles; allocated on the current stream (the caching allocator keeps it
    # stream-ordered), ~25 MB for the wide convolutions
    if not WGRAD_WORKSPACE:
        ws = None
    elif ws is None:
        ws = _empty(H.lib().dx_conv1d_wgrad_ws_floats(B, N, Cin, Cout, taps), dtype=torch.float32, device=dy.device)
    with _probe('conv_wgrad', lambda: 2. * B * N * Cin * Cout * taps, N):
      H.check(H.lib().dx_conv1d_wgrad(H.ptr(dy), H.dt(dy), dy.stride(1), H.ptr(x), H.dt(x), x.stride(1),
                                    H._DT[compute_dtype], H.ptr(dw), H.ptr(db), H.ptr(lengths), H.ptr(ws), B, N, Cin, Cout, taps,
                                    H.stream() if stream is None else stream))


class _WgradDesc(ctypes.Structure):
    _fields_ = [('dy', ctypes.c_void_p), ('x', ctypes.c_void_p), ('dw', ctypes.c_void_p), ('db', ctypes.c_void_p), ('lddy', ctypes.c_long),
                ('ldx', ctypes.c_long), ('dy_dtype', ctypes.c_int), ('x_dtype', ctypes.c_int), ('Cin', ctypes.c_int), ('Cout', ctypes.c_int),
                ('taps', ctypes.c_int), ('pad', ctypes.c_int)]


WGRAD_MULTI_MAX = 8


def conv1d_wgrad_multi(items, compute_dtype, lengths, stream=None, ws=None):
    ''' items: [(dy (B, N, Cout), x (B, N, Cin), dw, db)] over the same (B, N, lengths), at most WGRAD_MULTI_MAX: the GEMM launches of
        conv1d_wgrad for each, then ONE launch that adds all their partial tiles to the dw's (dx_conv1d_wgrad_multi).  ws: scratch of at
        least `wgrad_multi_ws_floats(items)` floats, valid until the call has completed on the stream. '''
    n = len(items)
    assert 0 < n <= WGRAD_MULTI_MAX and WGRAD_WORKSPACE
    B, N = items[0][0].shape[0], items[0][0].shape[1]
    arr = (_WgradDesc * n)()
    flops = 0.
    for i, (dy, x, dw, db) in enumerate(items):
        assert dy.shape[0] == B and dy.shape[1] == N and x.shape[0] == B and x.shape[1] == N and dy.stride(2) == 1 and x.stride(2) == 1
        taps = dw.shape[2] if dw.dim() == 3 else 1
        assert dw.shape[0] == dy.shape[2] and dw.shape[1] == x.shape[2]
        arr[i] = _WgradDesc(dy.data_ptr(), x.data_ptr(), dw.data_ptr(), db.data_ptr() if db is not None else None, dy.stride(1), x.stride(1),
                            H.dt(dy), H.dt(x), x.shape[2], dy.shape[2], taps, 0)
        flops += 2. * B * N * x.shape[2] * dy.shape[2] * taps
    if ws is None:
        ws = _empty(H.lib().dx_conv1d_wgrad_multi_ws_floats(arr, n, B, N), dtype=torch.float32, device=items[0][0].device)
    with _probe('conv_wgrad', lambda: flops, N):
        H.check(H.lib().dx_conv1d_wgrad_multi(arr, n, H._DT[compute_dtype], H.ptr(lengths), H.ptr(ws), B, N,
                                              H.stream() if stream is None else stream))


def wgrad_multi_ws_floats(shapes):
    ''' scratch floats of conv1d_wgrad_multi for [(B, N, Cin, Cout, taps)] '''
    return sum(H.lib().dx_conv1d_wgrad_ws_floats(*s) for s in shapes)


_WS_FLOATS = {}


def wgrad_ws_floats(B, N, Cin, Cout, taps):
    ''' (a pure function of the shape, asked four times per flush of an FFT block's weight gradients: remembered -- the phoneme-level
        stretches of the backward pass are host-bound) '''
    key = (B, N, Cin, Cout, taps)
    v = _WS_FLOATS.get(key)
    if v is None:
        v = _WS_FLOATS[key] = H.lib().dx_conv1d_wgrad_ws_floats(B, N, Cin, Cout, taps)
    return v



# ----------------------------------------------------------------------------- LayerNorm (+ residual, dropout, FiLM, mask)
def layernorm_fwd(x, gamma, beta, residual=None, film=None, lengths=None, out_dtype=torch.float32, save=False,
                  save_s=False, p_pre=0., seed_pre=0, p_post=0., seed_post=0, skip_lengths=None, lp_copy=False):
    ''' returns (y, s_out, mean, rstd) -- or (y, y_bf16, s_out, mean, rstd) with lp_copy '''
    B, N, C = x.shape
    assert x.is_contiguous()
    y = _empty((B, N, C), dtype=out_dtype, device=x.device)
    y_lp = _empty((B, N, C), dtype=torch.bfloat16, device=x.device) if lp_copy else None
    mean = rstd = s_out = None
    if save:
        mean = _empty(B * N, dtype=torch.float32, device=x.device)
        rstd = _empty_like(mean)
    if save_s:
        s_out = _empty((B, N, C), dtype=torch.float32, device=x.device)
    ldf = film.stride(0) if film is not None else 0
    H.check(H.lib().dx_layernorm_fwd(H.ptr(x), H.dt(x), H.ptr(residual), H.ptr(gamma), H.ptr(beta), H.ptr(film), ldf,
                                     H.ptr(lengths), H.ptr(skip_lengths), H.ptr(y), H.dt(y), H.ptr(y_lp), H.ptr(s_out), H.ptr(mean), H.ptr(rstd), B, N, C,
                                     float(p_pre), int(seed_pre), float(p_post), int(seed_post), STEP_PTR, H.stream()))
    if lp_copy:
        return y, y_lp, s_out, mean, rstd
    return y, s_out, mean, rstd


def layernorm_bwd(dy, s_in, mean, rstd, gamma, beta, dgamma, dbeta, film=None, dfilm=None, lengths=None,
                  d_dtype=torch.float32, p_pre=0., seed_pre=0, p_post=0., seed_post=0, relu_input=False, skip_lengths=None, lp_only=False, separate=False):
    ''' returns (ds, dx_pre); dx_pre is ds itself when there is no pre-dropout. dgamma/dbeta/dfilm accumulate.
        lp_only: dx_pre is returned as a bf16 tensor only (it feeds MFMA operands exclusively).
        separate: dx_pre gets its own buffer even without pre-dropout (the caller accumulates into ds in place while
        another stream still reads dx_pre). '''
    B, N, C = dy.shape
    ds = _empty((B, N, C), dtype=d_dtype, device=dy.device)
    dx_pre = _empty_like(ds) if ((p_pre > 0. or separate) and not lp_only) else None
    dx_lp = _empty((B, N, C), dtype=torch.bfloat16, device=dy.device) if lp_only else None
    ldf = film.stride(0) if film is not None else 0
    lddf = dfilm.stride(0) if dfilm is not None else 0
    # two-stage (atomic-free, run-to-run deterministic) reduction of dgamma/dbeta/dfilm; measured 7 % slower per step than
    # the fp32 atomics (one more dependent launch per LayerNorm), so it is opt-in
    ws = _empty(H.lib().dx_layernorm_bwd_ws_floats(B, N, C), dtype=torch.float32, device=dy.device) if DETERMINISTIC_LN else None
    H.check(H.lib().dx_layernorm_bwd(H.ptr(dy), H.dt(dy), H.ptr(s_in), H.dt(s_in), H.ptr(mean), H.ptr(rstd), H.ptr(gamma),
                                     H.ptr(beta), H.ptr(film), ldf, H.ptr(lengths), H.ptr(skip_lengths), H.ptr(ds), H.ptr(dx_pre), H.ptr(dx_lp), H.dt(ds),
                                     H.ptr(dgamma), H.ptr(dbeta), H.ptr(dfilm), lddf, B, N, C, float(p_pre), int(seed_pre),
                                     float(p_post), int(seed_post), int(relu_input), H.ptr(ws), STEP_PTR, H.stream()))
    if lp_only:
        return ds, dx_lp
    return ds, (dx_pre if dx_pre is not None else ds)


# ----------------------------------------------------------------------------- attention
def length_order(lengths):
    ''' int32 (B): utterance indices by decreasing length (dx_length_order) -- the launch order of the attention kernels '''
    order = _empty((lengths.shape[0],), dtype=torch.int32, device=lengths.device)
    H.check(H.lib().dx_length_order(H.ptr(lengths), lengths.shape[0], H.ptr(order), H.stream()))
    return order


def attention_fwd(qkv, lengths, nb_heads, p_drop=0., seed=0, need_lse=True, order=None):
    B, N, E3 = qkv.shape
    E = E3 // 3
    assert qkv.is_contiguous()
    o = _empty((B, N, E), dtype=qkv.dtype, device=qkv.device)
    lse = _empty((B, nb_heads, N), dtype=torch.float32, device=qkv.device) if need_lse else None
    H.check(H.lib().dx_attention_fwd(H.ptr(qkv), H.dt(qkv), H.ptr(lengths), H.ptr(order), H.ptr(o), H.ptr(lse), B, N, nb_heads, E,
                                     float(p_drop), int(seed), STEP_PTR, H.stream()))
    return o, lse


ATTN_AUTO, ATTN_TWO_PASS, ATTN_FUSED = 0, 1, 2
_ATTN_WS = {}        # (device, raw stream) -> [scratch floats (grow-only), arrival counters (int32, zeroed once)]
ATTN_WS_OWNER = None   # a graph capture in progress sets this to a dict of its own: its launches must not share (or outlive) the eager buffers


def _attn_workspace(B, N, nb_heads, device):
    ''' (scratch, counters) of dx_attention_bwd for the current stream.  The scratch (delta + the fused kernel's dQ partials, ~50 MB at
        B = 48 / T = 1000) needs no initialisation and nothing in it survives a call, so ONE grow-only buffer per stream serves every
        batch shape (real data has a new N every batch); the arrival counters sit in a small buffer of their own, zeroed once -- every
        launch leaves the counters it used at zero.  Calls on one stream run in order, so they can share both. '''
    table = _ATTN_WS if ATTN_WS_OWNER is None else ATTN_WS_OWNER
    key = (device, H.stream())
    need, need_c = H.lib().dx_attention_bwd_ws_floats(B, N, nb_heads), H.lib().dx_attention_bwd_counters(B, nb_heads)
    ent = table.get(key)
    if ent is None:
        ent = table[key] = [None, None]
    if ent[0] is None or ent[0].numel() < need:
        ent[0] = _empty((max(need, 1 << 20),), dtype=torch.float32, device=device)
    if ent[1] is None or ent[1].numel() < need_c:
        ent[1] = zeros((max(need_c, 4096),), device, dtype=torch.int32)
    return ent[0], ent[1]


def attention_bwd(qkv, o, d_o, lse, lengths, nb_heads, p_drop=0., seed=0, order=None, algo=ATTN_AUTO):
    B, N, E3 = qkv.shape
    E = E3 // 3
    assert d_o.is_contiguous() and d_o.dtype == qkv.dtype
    dqkv = _empty_like(qkv)
    ws, counters = _attn_workspace(B, N, nb_heads, qkv.device)
    H.check(H.lib().dx_attention_bwd(H.ptr(qkv), H.ptr(o), H.ptr(d_o), H.dt(qkv), H.ptr(lse), H.ptr(lengths), H.ptr(order), H.ptr(dqkv),
                                     H.ptr(ws), H.ptr(counters), B, N, nb_heads, E, float(p_drop), int(seed), STEP_PTR, int(algo), H.stream()))
    return dqkv


# ----------------------------------------------------------------------------- pointwise / small heads
def scalar_embed_fwd(feats, ws, biases, base=None, pos_table=None, lengths=None):
    B, N = feats[0].shape
    C, taps = ws[0].shape[0], ws[0].shape[2]
    assert all(tuple(w.shape) == (C, 1, taps) for w in ws)
    out = _empty((B, N, C), dtype=torch.float32, device=feats[0].device)
    H.check(H.lib().dx_scalar_embed_fwd(H.ptr(base), _ptr_array(feats), _ptr_array(ws), _ptr_array(biases), len(feats),
                                        H.ptr(pos_table), H.ptr(lengths), H.ptr(out), B, N, C, taps, H.stream()))
    return out


def scalar_embed_bwd(dout, feats, dws, dbiases, lengths=None, need_dbase=False):
    B, N, C = dout.shape
    dbase = _empty_like(dout) if need_dbase else None
    taps = dws[0].shape[2]
    assert all(tuple(w.shape) == (C, 1, taps) for w in dws)
    H.check(H.lib().dx_scalar_embed_bwd(H.ptr(dout), _ptr_array(feats), len(feats), H.ptr(lengths), H.ptr(dbase),
                                        _ptr_array(dws), _ptr_array(dbiases), B, N, C, taps, H.stream()))
    return dbase


def embed_pos_fwd(ids, table, pos_table, lengths):
    B, N = ids.shape
    C = table.shape[1]
    assert pos_table.shape[1] == C and table.is_contiguous() and pos_table.is_contiguous()
    out = _empty((B, N, C), dtype=torch.float32, device=ids.device)
    H.check(H.lib().dx_embed_pos_fwd(H.ptr(ids), H.ptr(table), H.ptr(pos_table), H.ptr(lengths), H.ptr(out), B, N, C, H.stream()))
    return out


def embed_pos_bwd(ids, dout, lengths, dtable):
    B, N = ids.shape
    C = dtable.shape[1]
    assert dout.shape[2] == C and dout.is_contiguous()
    H.check(H.lib().dx_embed_pos_bwd(H.ptr(ids), H.ptr(dout), H.ptr(lengths), H.ptr(dtable), B, N, C, H.stream()))


def masked_mean_fwd(x, lengths):
    B, N, C = x.shape
    out = _empty((B, C), dtype=torch.float32, device=x.device)
    H.check(H.lib().dx_masked_mean_fwd(H.ptr(x), H.ptr(lengths), H.ptr(out), B, N, C, H.stream()))
    return out


def masked_mean_bwd(dy, lengths, N):
    B, C = dy.shape
    dx = _empty((B, N, C), dtype=torch.float32, device=dy.device)
    H.check(H.lib().dx_masked_mean_bwd(H.ptr(dy), H.ptr(lengths), H.ptr(dx), B, N, C, H.stream()))
    return dx


def film_assemble_fwd(g_raw, b_raw, post, nb, ch):
    B = g_raw.shape[0]
    films = [_empty((B, nb[m], 2 * ch[m]), dtype=torch.float32, device=g_raw.device) for m in range(3)]
    H.check(H.lib().dx_film_assemble_fwd(H.ptr(g_raw), H.ptr(b_raw), H.ptr(post), H.ptr(films[0]), H.ptr(films[1]),
                                         H.ptr(films[2]), _int_array(nb), _int_array(ch), B, H.stream()))
    return films


def film_assemble_bwd(g_raw, b_raw, post, dfilms, dpost, nb, ch):
    B = g_raw.shape[0]
    dg, db = _empty_like(g_raw), _empty_like(b_raw)
    H.check(H.lib().dx_film_assemble_bwd(H.ptr(g_raw), H.ptr(b_raw), H.ptr(post), H.ptr(dfilms[0]), H.ptr(dfilms[1]),
                                         H.ptr(dfilms[2]), H.ptr(dg), H.ptr(db), H.ptr(dpost), _int_array(nb), _int_array(ch),
                                         B, H.stream()))
    return dg, db


USE_FUSED_HEADS = True   # False: the FiLM head / speaker classifier as their separate small launches (tests compare the two)


def film_head_fwd(emb, spk_table, spk_ids, wg, bg, wb, bb, post, nb, ch):
    ''' (z, g_raw, b_raw, [film_enc, film_pp, film_dec]) in one launch (dx_film_head_fwd) '''
    B, C = emb.shape
    W = sum(n * c for n, c in zip(nb, ch))
    dev = emb.device
    z = _empty((B, C), dtype=torch.float32, device=dev)
    g_raw, b_raw = _empty((B, W), dtype=torch.float32, device=dev), _empty((B, W), dtype=torch.float32, device=dev)
    films = [_empty((B, nb[m], 2 * ch[m]), dtype=torch.float32, device=dev) for m in range(3)]
    H.check(H.lib().dx_film_head_fwd(H.ptr(emb), H.ptr(spk_table), H.ptr(spk_ids), H.ptr(wg), H.ptr(bg), H.ptr(wb), H.ptr(bb), H.ptr(post),
                                     H.ptr(z), H.ptr(g_raw), H.ptr(b_raw), H.ptr(films[0]), H.ptr(films[1]), H.ptr(films[2]),
                                     _int_array(nb), _int_array(ch), B, C, H.stream()))
    return z, g_raw, b_raw, films


def film_head_bwd(g_raw, b_raw, post, z, spk_ids, wg, wb, dfilms, d_emb, d_spk_table, dpost, dwg, dbg, dwb, dbb, nb, ch):
    ''' everything behind the FiLM tensors' gradients in two launches (dx_film_head_bwd); d_emb, d_spk_table, dpost and the four
        parameter gradients are accumulated '''
    B, C = z.shape
    ws = _empty((2,) + tuple(g_raw.shape), dtype=torch.float32, device=z.device)
    H.check(H.lib().dx_film_head_bwd(H.ptr(g_raw), H.ptr(b_raw), H.ptr(post), H.ptr(z), H.ptr(spk_ids), H.ptr(wg), H.ptr(wb), H.ptr(dfilms[0]),
                                     H.ptr(dfilms[1]), H.ptr(dfilms[2]), H.ptr(ws[0]), H.ptr(ws[1]), H.ptr(d_emb), H.ptr(d_spk_table), H.ptr(dpost),
                                     H.ptr(dwg), H.ptr(dbg), H.ptr(dwb), H.ptr(dbb), _int_array(nb), _int_array(ch), B, C, H.stream()))


def classifier_fwd(emb, w1, b1, w2, b2, w3, b3):
    B, C = emb.shape
    S = w3.shape[0]
    h1, h2 = _empty_like(emb), _empty_like(emb)
    logits = _empty((B, S), dtype=torch.float32, device=emb.device)
    H.check(H.lib().dx_classifier_fwd(H.ptr(emb), H.ptr(w1), H.ptr(b1), H.ptr(w2), H.ptr(b2), H.ptr(w3), H.ptr(b3), H.ptr(h1), H.ptr(h2),
                                      H.ptr(logits), B, C, S, H.stream()))
    return logits, h1, h2


def classifier_bwd(d_logits, emb, h1, h2, w1, w2, w3, lambda_, dw1, db1, dw2, db2, dw3, db3):
    ''' returns d_emb = -lambda * dL/d(classifier input); parameter gradients accumulated (dx_classifier_bwd, two launches) '''
    B, C = emb.shape
    ws = _empty((2, B, C), dtype=torch.float32, device=emb.device)
    d_emb = _empty_like(emb)
    assert d_logits.is_contiguous()
    H.check(H.lib().dx_classifier_bwd(H.ptr(d_logits), H.ptr(emb), H.ptr(h1), H.ptr(h2), H.ptr(w1), H.ptr(w2), H.ptr(w3), H.ptr(ws[0]), H.ptr(ws[1]),
                                      H.ptr(d_emb), float(lambda_), H.ptr(dw1), H.ptr(db1), H.ptr(dw2), H.ptr(db2), H.ptr(dw3), H.ptr(db3), B, C,
                                      w3.shape[0], H.stream()))
    return d_emb


def linear_small_fwd(x, w, bias, relu=False, mask_lengths=None, N=1):
    K = x.shape[-1]
    M = x.numel() // K
    O = w.shape[0]
    assert x.is_contiguous() and w.is_contiguous()
    y = _empty(x.shape[:-1] + (O,), dtype=torch.float32, device=x.device)
    H.check(H.lib().dx_linear_small_fwd(H.ptr(x), H.ptr(w), H.ptr(bias), H.ptr(y), H.ptr(mask_lengths), N, M, K, O, int(relu), H.stream()))
    return y


def linear_small_bwd(dy, y, x, w, dw, db, relu=False, mask_lengths=None, N=1, need_dx=True, dx_scale=1.):
    K = x.shape[-1]
    M = x.numel() // K
    O = w.shape[0]
    assert dy.is_contiguous() and x.is_contiguous()
    dx = _empty_like(x) if need_dx else None
    H.check(H.lib().dx_linear_small_bwd(H.ptr(dy), H.ptr(y), H.ptr(x), H.ptr(w), H.ptr(dx), float(dx_scale), H.ptr(dw), H.ptr(db),
                                        H.ptr(mask_lengths), N, M, K, O, int(relu), H.stream()))
    return dx


def gather_add_fwd(a, table, ids):
    B, C = a.shape
    out = _empty_like(a)
    H.check(H.lib().dx_gather_add_fwd(H.ptr(a), H.ptr(table), H.ptr(ids), H.ptr(out), B, C, H.stream()))
    return out


def gather_add_bwd(dz, ids, dtable):
    B, C = dz.shape
    H.check(H.lib().dx_gather_add_bwd(H.ptr(dz), H.ptr(ids), H.ptr(dtable), B, C, H.stream()))


def add_(dst, src):
    assert dst.is_contiguous() and src.is_contiguous() and dst.numel() == src.numel()
    H.check(H.lib().dx_add_inplace(H.ptr(dst), H.ptr(src), dst.numel(), H.stream()))
    return dst


def transpose_last2(x, out_dtype=torch.float32):
    ''' (B, R, C) fp32 -> (B, C, R) in out_dtype (fp32 / bf16), contiguous '''
    B, R, C = x.shape
    assert x.is_contiguous() and x.dtype == torch.float32
    y = _empty((B, C, R), dtype=out_dtype, device=x.device)
    H.check(H.lib().dx_transpose_last2(H.ptr(x), H.ptr(y), H.dt(y), B, R, C, H.stream()))
    return y


def unstack(y, K):
    ''' (..., K) interleaved fp32 -> K contiguous tensors of shape (...) '''
    assert y.is_contiguous() and y.shape[-1] == K and y.dtype == torch.float32
    outs = [_empty(y.shape[:-1], dtype=torch.float32, device=y.device) for _ in range(K)]
    H.check(H.lib().dx_unstack(H.ptr(y), _ptr_array(outs), y.numel() // K, K, H.stream()))
    return outs


def stack(planes):
    ''' K contiguous fp32 tensors of one shape (...) -> (..., K) interleaved '''
    K = len(planes)
    assert all(p.is_contiguous() and p.dtype == torch.float32 and p.shape == planes[0].shape for p in planes)
    y = _empty(planes[0].shape + (K,), dtype=torch.float32, device=planes[0].device)
    H.check(H.lib().dx_stack(H.ptr(y), _ptr_array(planes), planes[0].numel(), K, H.stream()))
    return y


def zeros(shape, device, dtype=torch.float32):
    ''' device buffer cleared by a stream-ordered memset (dx_fill_zero) '''
    t = _empty(shape, dtype=dtype, device=device)
    H.check(H.lib().dx_fill_zero(H.ptr(t), t.numel() * t.element_size(), H.stream()))
    return t


def colsum_(x, out):
    C = x.shape[-1]
    H.check(H.lib().dx_colsum(H.ptr(x), H.dt(x), H.ptr(out), x.numel() // C, C, H.stream()))


# ----------------------------------------------------------------------------- Gaussian upsampling
def gu_prepare(enc, dur_float, energy, pitch, in_lengths, P, save=False):
    ''' P: dict with w_dur, b_dur, w_en, b_en, w_pi, b_pi, w_range, b_range '''
    B, L, C = enc.shape
    dev = enc.device
    xp = _empty((B, L, C), dtype=torch.float32, device=dev)
    ranges = _empty((B, L), dtype=torch.float32, device=dev)
    r_pre = _empty((B, L), dtype=torch.float32, device=dev) if save else None
    rin = _empty((B, L, C), dtype=torch.float32, device=dev) if save else None
    H.check(H.lib().dx_gu_prepare(H.ptr(enc), H.ptr(dur_float), H.ptr(energy), H.ptr(pitch), H.ptr(in_lengths),
                                  H.ptr(P['w_dur']), H.ptr(P['b_dur']), H.ptr(P['w_en']), H.ptr(P['b_en']), H.ptr(P['w_pi']),
                                  H.ptr(P['b_pi']), H.ptr(P['w_range']), H.ptr(P['b_range']), H.ptr(xp), H.ptr(ranges),
                                  H.ptr(r_pre), H.ptr(rin), B, L, C, H.stream()))
    return xp, ranges, r_pre, rin


def gu_means(durations_int):
    B, L = durations_int.shape
    means = _empty((B, L), dtype=torch.float32, device=durations_int.device)
    totals = _empty((B,), dtype=torch.int64, device=durations_int.device)
    H.check(H.lib().dx_gu_means(H.ptr(durations_int), H.ptr(means), H.ptr(totals), B, L, H.stream()))
    return means, totals


def gu_upsample_fwd(xp, ranges, means, in_lengths, T, out_lengths=None, pos_table=None):
    B, L, C = xp.shape
    weights = _empty((B, L, T), dtype=torch.float32, device=xp.device)
    out = _empty((B, T, C), dtype=torch.float32, device=xp.device)
    H.check(H.lib().dx_gu_upsample_fwd(H.ptr(xp), H.ptr(ranges), H.ptr(means), H.ptr(in_lengths), H.ptr(out_lengths),
                                       H.ptr(pos_table), H.ptr(weights), H.ptr(out), B, L, T, C, H.stream()))
    return out, weights


def gu_upsample_bwd(g, xp, weights, means, ranges, r_pre, w_range, in_lengths, out_lengths):
    B, L, C = xp.shape
    T = weights.shape[2]
    dev = xp.device
    dw_ws = _empty((B, L, T), dtype=torch.float32, device=dev)
    dsum_ws = _empty((B, T), dtype=torch.float32, device=dev)
    dxp, drin = _empty_like(xp), _empty_like(xp)
    dr = _empty((B, L), dtype=torch.float32, device=dev)
    H.check(H.lib().dx_gu_upsample_bwd(H.ptr(g), H.ptr(xp), H.ptr(weights), H.ptr(means), H.ptr(ranges), H.ptr(r_pre),
                                       H.ptr(w_range), H.ptr(in_lengths), H.ptr(out_lengths), H.ptr(dw_ws), H.ptr(dsum_ws),
                                       H.ptr(dxp), H.ptr(drin), H.ptr(dr), B, L, T, C, H.stream()))
    return dxp, drin, dr


# ----------------------------------------------------------------------------- loss / optimizer / durations
def loss_fwd_bwd(dur, energy, pitch, dur_t, energy_t, pitch_t, in_lengths, mel, mel_t, out_lengths, spk_logits, spk_ids,
                 post_mult, weights, grads=None, d_post_mult=None, grad_scale=1., d_mel_transposed=False):
    ''' weights = (w_spk, w_post, w_dur, w_energy, w_pitch, w_mel). grads: None or dict of output tensors
        (d_dur, d_energy, d_pitch, d_mel, d_spk).  Returns the (8,) device tensor of loss terms. '''
    B, L = dur.shape
    n_mel, T = mel.shape[1], mel.shape[2]
    terms = _empty(8, dtype=torch.float32, device=dur.device)
    g = grads or {}
    assert mel.is_contiguous() and mel_t.is_contiguous()
    # per-workgroup terms + a fixed-order sum in the last launch (3 launches, no atomics) when the mel gradient is written transposed
    ws = _empty(H.lib().dx_loss_ws_floats(B, T), dtype=torch.float32, device=dur.device) if (d_mel_transposed and g.get('d_mel') is not None) else None
    H.check(H.lib().dx_loss_fwd_bwd(H.ptr(dur), H.ptr(energy), H.ptr(pitch), H.ptr(dur_t), H.ptr(energy_t), H.ptr(pitch_t),
                                    H.ptr(in_lengths), H.ptr(mel), H.ptr(mel_t), H.ptr(out_lengths), H.ptr(spk_logits),
                                    H.ptr(spk_ids), H.ptr(post_mult), H.ptr(g.get('d_dur')), H.ptr(g.get('d_energy')),
                                    H.ptr(g.get('d_pitch')), H.ptr(g.get('d_mel')), H.ptr(g.get('d_spk')), H.ptr(d_post_mult),
                                    H.ptr(terms), H.ptr(ws), B, L, T, n_mel, spk_logits.shape[1],
                                    post_mult.numel() if post_mult is not None else 0, *[float(w) for w in weights],
                                    float(grad_scale), int(d_mel_transposed), STEP_PTR, H.stream()))
    return terms


def sumsq(x, out=None):
    out = out if out is not None else _empty(1, dtype=torch.float32, device=x.device)
    H.check(H.lib().dx_sumsq(H.ptr(x), x.numel(), H.ptr(out), H.stream()))
    return out


def adam_step(p, g, m, v, lr, betas, eps, weight_decay, step, grad_norm_sq=None, clip_thresh=_INF, norm_accum=None):
    ''' with ops.STEP_PTR set (captured step) lr and the bias corrections come from the device-side step block '''
    H.check(H.lib().dx_adam_step(H.ptr(p), H.ptr(g), H.ptr(m), H.ptr(v), p.numel(), float(lr), float(betas[0]), float(betas[1]),
                                 float(eps), float(weight_decay), int(step), H.ptr(grad_norm_sq), float(clip_thresh),
                                 H.ptr(norm_accum), STEP_PTR, H.stream()))


def adam_pack_table(weights, flats, device):
    ''' device tables of dx_adam_pack_step.  weights: [(offset, (Cout, Cin, taps), fwd, tr, frag_fwd, frag_tr)] (copies: tensors or None);
        flats: [(offset, numel)].  Returns (bricks table, n_weights, total_bricks, flats table, n_flats, total_flat_blocks). '''
    import numpy as np
    bd = np.dtype([('off', '<i8'), ('fwd', '<u8'), ('tr', '<u8'), ('ffwd', '<u8'), ('ftr', '<u8'), ('Cout', '<i4'), ('Cin', '<i4'),
                   ('taps', '<i4'), ('pad', '<i4'), ('begin', '<i8')])
    fd = np.dtype([('off', '<i8'), ('n', '<i8'), ('begin', '<i8')])
    assert bd.itemsize == H.lib().dx_adam_pack_desc_size() and fd.itemsize == H.lib().dx_adam_flat_desc_size()
    ptr = lambda t: 0 if t is None else t.data_ptr()
    arr, begin = np.zeros(len(weights), dtype=bd), 0
    for i, (off, (cout, cin, taps), fwd, tr, ffwd, ftr) in enumerate(weights):
        assert taps in (1, 3), f'adam_pack_table: a GEMM weight with {taps} taps (the brick kernel holds 1 or 3 taps per 32 x 32 brick)'
        if ffwd is not None or ftr is not None:
            assert taps == 3 and cout % 32 == 0 and cin % 32 == 0
        arr[i] = (off, ptr(fwd), ptr(tr), ptr(ffwd), ptr(ftr), cout, cin, taps, 0, begin)
        begin += ((cout + 31) // 32) * ((cin + 31) // 32)
    blk = H.lib().dx_adam_flat_block()
    farr, fbegin = np.zeros(max(1, len(flats)), dtype=fd), 0
    for i, (off, n) in enumerate(flats):
        farr[i] = (off, n, fbegin)
        fbegin += (n + blk - 1) // blk
    up = lambda a: torch.from_numpy(a.view(np.uint8).copy()).to(device)
    return up(arr), len(weights), begin, up(farr), len(flats), fbegin


def adam_pack_step(p, g, m, v, table, out_dtype, lr, betas, eps, weight_decay, step, norm_accum=None):
    ''' whole-buffer Adam + refresh of every MFMA operand copy in one launch (dx_adam_pack_step) '''
    bricks, nw, nbricks, flats, nf, nfb = table
    H.check(H.lib().dx_adam_pack_step(H.ptr(p), H.ptr(g), H.ptr(m), H.ptr(v), H.ptr(bricks), nw, nbricks, H.ptr(flats) if nf else None, nf, nfb,
                                      H._DT[out_dtype], float(lr), float(betas[0]), float(betas[1]), float(eps), float(weight_decay), int(step),
                                      H.ptr(norm_accum), STEP_PTR, H.stream()))


def step_scalars_set(block, seed_salt, lr, betas, step, w_speaker):
    ''' fill the device-side step block (a 48-byte uint8 tensor) for the step about to be replayed (dx_step_scalars_set) '''
    H.check(H.lib().dx_step_scalars_set(H.ptr(block), int(seed_salt), float(lr), float(betas[0]), float(betas[1]), int(step),
                                        float(w_speaker), H.stream()))


def scale_(x, s):
    H.check(H.lib().dx_scale(H.ptr(x), x.numel(), float(s), H.stream()))


def int_durations(duration_preds, hparams, dur_factors=None):
    ''' in-place thresholding of duration_preds; returns (durations_int, totals, status) device tensors '''
    B, L = duration_preds.shape
    dev = duration_preds.device
    assert duration_preds.is_contiguous()
    dint = _empty((B, L), dtype=torch.int64, device=dev)
    totals = _empty((B,), dtype=torch.int64, device=dev)
    status = _empty((B,), dtype=torch.int32, device=dev)
    H.check(H.lib().dx_int_durations(H.ptr(duration_preds), H.ptr(dur_factors), H.ptr(dint), H.ptr(totals), H.ptr(status), B, L,
                                     float(hparams.sampling_rate), int(hparams.filter_length), int(hparams.hop_length),
                                     int(bool(hparams.centered)), H.stream()))
    return dint, totals, status


def prosody_control(energy, pitch, energy_factors, pitch_factors, durations_int, mode, speaker_ids=None, spk_mean=None,
                    spk_std=None):
    B, L = energy.shape
    H.check(H.lib().dx_prosody_control(H.ptr(energy), H.ptr(pitch), H.ptr(energy_factors), H.ptr(pitch_factors),
                                       H.ptr(durations_int), H.ptr(speaker_ids), H.ptr(spk_mean), H.ptr(spk_std), int(mode),
                                       B, L, H.stream()))
