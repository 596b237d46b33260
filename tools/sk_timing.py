#!/usr/bin/env python
"""Development: per-workgroup time stamps of the split-K LayerNorm-fused conv GEMM (build with DX_EXTRA_HIPCC_FLAGS="-DSK_TIMING=<LNM>",
1 = forward LayerNorm epilogue, 3 = backward): runs a few training steps, then prints the phases of the LAST such launch."""
import ctypes, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'ubisoft-laforge-daft-exprt_amd'))
sys.argv = ['bench.py', '--steps', '3', '--warmup', '3', '--no-cpu-baseline', '--no-probe']
import bench
bench.main()
torch.cuda.synchronize()
from daft_exprt import _hip as H
lib = ctypes.CDLL(H.LIB_PATH)
ts = (ctypes.c_ulonglong * 8192)()
print('rc', lib.dx_debug_sk_ts(ts))
a = np.array(list(ts), dtype=np.int64).reshape(1024, 8)[:, :6]
a = a[(a[:, 0] > 0) & (a[:, 5] > 0)]
a = a[a[:, 0] > a[:, 0].max() - 20000]      # the last launch only
live = a[a[:, 1] > 0]
t0 = a[:, 0].min()
print('%d workgroups (%d with a tile); start spread %.2f us; end: mean %.2f max %.2f us' % (len(a), len(live), (a[:, 0].max() - t0) / 100., (a[:, 5] - t0).mean() / 100., (a[:, 5] - t0).max() / 100.))
names = ['prologue', 'main loop', 'K-half exchange', 'LayerNorm epilogue', 'padding fill']
for i, n in enumerate(names):
    d = (live[:, i + 1] - live[:, i]) / 100.
    print('  %-20s mean %6.2f  min %6.2f  max %6.2f us' % (n, d.mean(), d.min(), d.max()))

ch = (ctypes.c_ulonglong * 2048)()
lib.dx_debug_sk_chunk(ch)
c = np.array(list(ch), dtype=np.int64).reshape(4, 64, 8)
print('workgroup 40, s_memtime ticks per chunk: DMA wait | barrier | DMA issue | fragments + MFMAs | (chunk period)')
for w in (0, 3):
    print(' wave', w)
    for k in list(range(0, 6)) + list(range(14, 18)) + list(range(28, 32)):
        x = c[w, k]
        print('  chunk %2d: %5d | %5d | %5d | %5d | %5d' % (k, x[1] - x[0], x[2] - x[1], x[3] - x[2], x[4] - x[3], c[w, k + 1, 0] - x[0] if k < 31 else 0))
