#!/usr/bin/env python
"""Development: the split-K conv + LayerNorm launch (dx_conv1d_ln with a plan and fragment-order weights, y2 epilogue on) alone, at the
frame level (B = 48, N = 1000) and the phoneme level (B = 48, N = 160) of the B = 48 training step."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'ubisoft-laforge-daft-exprt_amd')); sys.path.insert(0, os.path.join(ROOT, 'tools'))
from daft_exprt import ops
from bench_ops import timeit
dev = torch.device('cuda:0')
torch.manual_seed(0)
for B, N, lo in ((48, 1000, 250), (48, 160, 40)):
    lens = torch.randint(lo, N + 1, (B,), device=dev); lens[0] = N
    x = torch.randn(B, N, 1024, device=dev).to(torch.bfloat16)
    wp = ops.pack_conv_weight(torch.randn(128, 1024, 3, device=dev) / 3072 ** 0.5, torch.bfloat16); wf = ops.pack_frag_major(wp)
    bias, g, bt = torch.zeros(128, device=dev), torch.ones(128, device=dev), torch.zeros(128, device=dev)
    res = torch.randn(B, N, 128, device=dev)
    w2 = ops.pack_conv_weight(torch.randn(384, 128, device=dev) / 128 ** 0.5, torch.bfloat16); b2 = torch.zeros(384, device=dev)
    plan = ops.conv_tile_plan(lens, N)
    t = timeit(lambda: ops.conv1d_ln(x, wp, bias, res, g, bt, lens, save=True, p_pre=0.1, seed_pre=5, lp_copy=True, plan=plan, w_frag=wf, w2_packed=w2, b2=b2))
    print(f'conv_sk forward B={B} N={N} rows={int(lens.sum())}: {t * 1e3:6.1f} us')
