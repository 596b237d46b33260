#!/usr/bin/env python
"""Development: per-wave phase time stamps of conv_wreg_kernel (build with DX_EXTRA_HIPCC_FLAGS="-DWR_TIMING=<workgroup + 1>")."""
import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'ubisoft-laforge-daft-exprt_amd'))
from daft_exprt import ops
from daft_exprt import _hip as H
dev = torch.device('cuda:0')
B, N, cin, cout = 48, 1000, 128, 1024
x = torch.randn(B, N, cin, device=dev).to(torch.bfloat16)
w = torch.randn(cout, cin, 3, device=dev) / (cin * 3) ** 0.5
wp = ops.pack_conv_weight(w, torch.bfloat16)
bias = torch.zeros(cout, device=dev)
out = torch.empty(B, N, cout, device=dev, dtype=torch.bfloat16)
for _ in range(3):
    ops.conv1d(x, wp, bias, out_dtype=torch.bfloat16, relu=True, out=out)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 512)()
lib = ctypes.CDLL(H.LIB_PATH)
rc = lib.dx_debug_wreg_timing(buf)
print('rc', rc)
for wv in range(8):
    ts = [buf[wv * 64 + i] for i in range(30)]
    ts = [t for t in ts if t]
    print('wave', wv, 'deltas', [ts[i + 1] - ts[i] for i in range(len(ts) - 1)])

wg = (ctypes.c_ulonglong * 4096)()
lib.dx_debug_wreg_wg(wg)
import numpy as np
a = np.array(list(wg), dtype=np.int64).reshape(1024, 4)[:256]
t0 = a[:, 0].min()
a = (a - t0) / 100.0   # us
print('workgroup start us: min %.2f max %.2f; after prologue: mean %.2f max %.2f; end of loop: mean %.2f max %.2f; end: mean %.2f max %.2f' % (
    a[:, 0].min(), a[:, 0].max(), a[:, 1].mean(), a[:, 1].max(), a[:, 2].mean(), a[:, 2].max(), a[:, 3].mean(), a[:, 3].max()))
print('per-workgroup loop duration us: min %.2f mean %.2f max %.2f' % ((a[:, 2] - a[:, 1]).min(), (a[:, 2] - a[:, 1]).mean(), (a[:, 2] - a[:, 1]).max()))
order = np.argsort(a[:, 3])
print('slowest workgroups (id, start, prologue end, loop end, end):', [(int(i), *[round(float(v), 1) for v in a[i]]) for i in order[-5:]])
print('fastest:', [(int(i), *[round(float(v), 1) for v in a[i]]) for i in order[:3]])
