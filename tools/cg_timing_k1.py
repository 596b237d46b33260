#!/usr/bin/env python
"""Development: per-workgroup time stamps of the register-staged k = 1 LayerNorm-fused GEMMs (build with
DX_EXTRA_HIPCC_FLAGS="-DCG_TIMING=<LNM> -DCG_TIMING_K1", LNM 1 = out-projection + LayerNorm forward, 3 = QKV data gradient + LayerNorm
backward): runs the micro-benchmark, prints the phases of the last stamped launch."""
import ctypes, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'ubisoft-laforge-daft-exprt_amd'))
sys.path.insert(0, os.path.join(ROOT, 'tools'))
sys.argv = ['bench_ops.py']
import bench_ops
(bench_ops.bench_lnbwd if os.environ.get('CG_K1', 'fwd') == 'bwd' else bench_ops.bench_convln)()
torch.cuda.synchronize()
from daft_exprt import _hip as H
lib = ctypes.CDLL(H.LIB_PATH)
wg = (ctypes.c_ulonglong * 4096)()
print('rc', lib.dx_debug_cg_wg(wg))
a = np.array(list(wg), dtype=np.int64).reshape(1024, 4)
a = a[(a[:, 0] > 0) & (a[:, 3] > 0)]
a = a[a[:, 0] > a[:, 0].max() - 20000]
t0 = a[:, 0].min()
a = (a - t0) / 100.0
live = a[:, 2] > a[:, 1]
print('%d stamped workgroups (%d with a main loop); start: max %.2f us; end: mean %.2f max %.2f' % (len(a), live.sum(), a[:, 0].max(), a[:, 3].mean(), a[:, 3].max()))
a = a[live]
print('prologue %.2f us | main loop: mean %.2f (min %.2f max %.2f) | epilogue: mean %.2f (min %.2f max %.2f)' % (
    (a[:, 1] - a[:, 0]).mean(), (a[:, 2] - a[:, 1]).mean(), (a[:, 2] - a[:, 1]).min(), (a[:, 2] - a[:, 1]).max(),
    (a[:, 3] - a[:, 2]).mean(), (a[:, 3] - a[:, 2]).min(), (a[:, 3] - a[:, 2]).max()))
order = np.argsort(a[:, 0])
q = a[order]
for lo, hi in ((0, 8), (len(q) // 2, len(q) // 2 + 4), (len(q) - 8, len(q))):
    for r in q[lo:hi]:
        print('   start %.2f  loop %.2f..%.2f  end %.2f' % tuple(r))
