#!/usr/bin/env python
"""Host-side cost of one wrapped C-ABI call (development aid): tiny tensors, no synchronisation inside the loop."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'ubisoft-laforge-daft-exprt_amd'))
from daft_exprt import ops, _hip as H
dev = torch.device('cuda:0')
x = torch.randn(2, 64, 128, device=dev).to(torch.bfloat16)
w = ops.pack_conv_weight(torch.randn(128, 128, 1, device=dev), torch.bfloat16)
lens = torch.tensor([64, 50], device=dev)
g, b = torch.ones(128, device=dev), torch.zeros(128, device=dev)
xf = torch.randn(2, 64, 128, device=dev)
def t(fn, n=2000):
    for _ in range(50): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    dt = (time.perf_counter() - t0) / n
    torch.cuda.synchronize()
    return dt * 1e6
print('conv1d           %.1f us' % t(lambda: ops.conv1d(x, w, None, out_dtype=torch.bfloat16, skip_lengths=lens)))
print('layernorm_fwd    %.1f us' % t(lambda: ops.layernorm_fwd(xf, g, b, lengths=lens, save=True, save_s=True, skip_lengths=lens)))
print('torch.empty      %.1f us' % t(lambda: torch.empty((2, 64, 128), dtype=torch.float32, device=dev)))
print('H.stream()       %.1f us' % t(lambda: H.stream()))
print('x.data_ptr()     %.1f us' % t(lambda: x.data_ptr()))
lib = H.lib()
print('dx_abi_version   %.1f us' % t(lambda: lib.dx_abi_version()))
print('dx_scale(ctypes) %.1f us' % t(lambda: lib.dx_scale(xf.data_ptr(), 16, 1.0, H.stream())))
s2 = torch.cuda.Stream()
def sw():
    s2.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s2):
        pass
print('side-stream hop  %.1f us' % t(sw))
