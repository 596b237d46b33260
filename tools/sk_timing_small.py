#!/usr/bin/env python
"""Development: phases of the split-K conv + LayerNorm kernel at a PHONEME-LEVEL shape (B = 48, N = 160; build with
DX_EXTRA_HIPCC_FLAGS="-DSK_TIMING=1"), SK_PLAN = 256 (the default plan) | 64 (fewer, taller tiles)."""
import ctypes, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'ubisoft-laforge-daft-exprt_amd'))
from daft_exprt import ops, _hip as H
dev = torch.device('cuda:0')
B, N = 48, 160
torch.manual_seed(0)
lens = torch.randint(40, N + 1, (B,), device=dev); lens[0] = N
x = torch.randn(B, N, 1024, device=dev).to(torch.bfloat16)
wp = ops.pack_conv_weight(torch.randn(128, 1024, 3, device=dev) / 3072 ** 0.5, torch.bfloat16); wf = ops.pack_frag_major(wp)
bias, g, bt = torch.zeros(128, device=dev), torch.ones(128, device=dev), torch.zeros(128, device=dev)
res = torch.randn(B, N, 128, device=dev)
w2 = ops.pack_conv_weight(torch.randn(384, 128, device=dev) / 128 ** 0.5, torch.bfloat16); b2 = torch.zeros(384, device=dev)
which = os.environ.get('SK_PLAN', '256')
plan = {'256': lambda: ops.conv_tile_plan(lens, N), '64': lambda: ops.conv_tile_plan(lens, N, tiles=64)}[which]()
for _ in range(5):
    ops.conv1d_ln(x, wp, bias, res, g, bt, lens, save=True, p_pre=0.1, seed_pre=5, lp_copy=True, plan=plan, w_frag=wf, w2_packed=w2, b2=b2)
torch.cuda.synchronize()
lib = ctypes.CDLL(H.LIB_PATH)
ts = (ctypes.c_ulonglong * 8192)()
lib.dx_debug_sk_ts(ts)
a = np.array(list(ts), dtype=np.int64).reshape(1024, 8)[:, :6]
a = a[(a[:, 0] > 0) & (a[:, 5] > 0)]
a = a[a[:, 0] > a[:, 0].max() - 20000]
live = a[a[:, 1] > 0]
t0 = a[:, 0].min()
print('plan %s: %d workgroups (%d with a tile); start spread %.2f us; end: mean %.2f max %.2f us' % (which, len(a), len(live), (a[:, 0].max() - t0) / 100., (a[:, 5] - t0).mean() / 100., (a[:, 5] - t0).max() / 100.))
for i, n in enumerate(['prologue', 'main loop', 'K-half exchange', 'LayerNorm epilogue', 'padding fill']):
    d = (live[:, i + 1] - live[:, i]) / 100.
    print('  %-36s mean %6.2f  min %6.2f  max %6.2f us' % (n, d.mean(), d.min(), d.max()))
