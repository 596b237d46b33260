"""ctypes binding of libdaftexprt_hip.so (C ABI: include/daft_exprt_hip.h).

No torch types cross the ABI: tensors are handed over as raw device pointers + sizes and
the current HIP stream.  There is NO fallback: if the library is missing this module raises.
"""
import ctypes
import os
import re

import torch

from daft_exprt import config

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = config.HIP_LIB or os.path.join(os.path.dirname(_HERE), 'csrc', 'libdaftexprt_hip.so')   # DX_HIP_LIB: A/B builds
HEADER_PATH = os.path.join(os.path.dirname(os.path.dirname(_HERE)), 'include', 'daft_exprt_hip.h')

F32, BF16, I64 = 0, 1, 2
_DT = {torch.float32: F32, torch.bfloat16: BF16, torch.int64: I64}
CONV_RELU, CONV_TRANSPOSED_OUT = 1, 2

_C = {'int': ctypes.c_int, 'long': ctypes.c_long, 'float': ctypes.c_float, 'double': ctypes.c_double,
      'size_t': ctypes.c_size_t, 'uint64_t': ctypes.c_uint64, 'int64_t': ctypes.c_int64, 'unsigned': ctypes.c_uint}
_lib = None


def header_prototypes():
    ''' [(name, restype, [argtype, ...])] parsed from the public header (single source of truth). '''
    text = open(HEADER_PATH).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    protos = []
    for m in re.finditer(r'\b(int|long|const char\*|size_t|void)\s+(dx_\w+)\s*\(([^;{]*?)\)\s*;', text, flags=re.S):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        argtypes = []
        if args and args != 'void':
            for a in args.split(','):
                a = a.strip()
                if '*' in a:
                    argtypes.append(ctypes.c_void_p)
                else:
                    base = a.replace('const ', '').split()[0]
                    argtypes.append(_C[base])
        restype = {'int': ctypes.c_int, 'long': ctypes.c_long, 'const char*': ctypes.c_char_p, 'size_t': ctypes.c_size_t, 'void': None}[ret]
        protos.append((name, restype, argtypes))
    return protos


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f'{LIB_PATH} is missing: build it with `python -c "import __graft_entry__ as g; g.build()"` '
                               f'(hipcc --offload-arch=gfx950). There is no CPU fallback.')
        L = ctypes.CDLL(LIB_PATH)
        for name, restype, argtypes in header_prototypes():
            fn = getattr(L, name)
            fn.restype = restype
            fn.argtypes = argtypes
        _lib = L
    return _lib


AFTER_LAUNCH = None     # one-shot callback run after the next entry point returns (model._flush_wgrads, capturing only)


def check(rc):
    global AFTER_LAUNCH
    if rc != 0:
        raise RuntimeError(f'libdaftexprt_hip error {rc}: {lib().dx_last_error().decode()}')
    if AFTER_LAUNCH is not None:
        cb, AFTER_LAUNCH = AFTER_LAUNCH, None
        cb()


def ptr(t):
    return None if t is None else t.data_ptr()


def dt(t):
    return _DT[t.dtype]


_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)


_current_device = torch.cuda.current_device


def stream():
    ''' raw hipStream_t of torch's current stream (the C hook is ~10x cheaper than building a torch.cuda.Stream object;
        ~340 calls per training step) '''
    if _raw_stream is not None:
        return _raw_stream(_current_device())
    return torch.cuda.current_stream().cuda_stream


def require_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError('daft_exprt HIP kernels need device tensors (no CPU fallback)')
