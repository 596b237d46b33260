#!/usr/bin/env python
"""Development: per-workgroup time stamps of the LayerNorm-fused conv GEMM (build with DX_EXTRA_HIPCC_FLAGS="-DCG_TIMING=<LNM>",
1 = forward LayerNorm epilogue, 3 = backward): runs a few training steps, then prints the timeline of the LAST such launch."""
import ctypes, os, subprocess, sys
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
wg = (ctypes.c_ulonglong * 4096)()
print('rc', lib.dx_debug_cg_wg(wg))
a = np.array(list(wg), dtype=np.int64).reshape(1024, 4)
a = a[(a[:, 0] > 0) & (a[:, 3] > 0)]
a = a[a[:, 0] > a[:, 0].max() - 20000]      # the last launch only (stamps within 200 us of the newest)
t0 = a[:, 0].min()
a = (a - t0) / 100.0
print('%d workgroups; start: max %.2f us | main loop start: mean %.2f max %.2f | main loop end: mean %.2f max %.2f | end: mean %.2f max %.2f' % (
    len(a), a[:, 0].max(), a[:, 1].mean(), a[:, 1].max(), a[:, 2].mean(), a[:, 2].max(), a[:, 3].mean(), a[:, 3].max()))
print('main loop: mean %.2f us (min %.2f max %.2f); epilogue: mean %.2f us (min %.2f max %.2f)' % (
    (a[:, 2] - a[:, 1]).mean(), (a[:, 2] - a[:, 1]).min(), (a[:, 2] - a[:, 1]).max(), (a[:, 3] - a[:, 2]).mean(), (a[:, 3] - a[:, 2]).min(), (a[:, 3] - a[:, 2]).max()))

ch = (ctypes.c_ulonglong * 512)()
lib.dx_debug_cg_chunk(ch)
c = np.array(list(ch), dtype=np.int64).reshape(2, 64, 4)
ld, mf = c[0, :30], c[1, :30]
print('loader wave 0 of workgroup 40, s_memtime ticks per chunk: wait for landing | barrier | issue | (chunk period)')
for k in range(2, 14):
    print('  chunk %2d: %5d | %5d | %5d | %5d' % (k, ld[k, 1] - ld[k, 0], ld[k, 2] - ld[k, 1], ld[k, 3] - ld[k, 2], ld[k + 1, 0] - ld[k, 0]))
print('MFMA wave 0: barrier wait | fragment reads + MFMAs | (chunk period)')
for k in range(2, 14):
    print('  chunk %2d: %5d | %5d | %5d' % (k, mf[k, 2] - mf[k, 1], mf[k, 3] - mf[k, 2], mf[k + 1, 1] - mf[k, 1]))
