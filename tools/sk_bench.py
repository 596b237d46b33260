#!/usr/bin/env python
"""Development: the split-K conv + LayerNorm launch (dx_conv1d_ln with a plan and fragment-order weights, y2 epilogue on) alone:
  full128  B = 64, N = 512, every utterance full: 256 tiles of exactly 128 rows (4 row blocks)
  c2       the four B = 48 bench batches' frame-level lengths (tiles of 124..135 rows)
  phoneme  B = 48, N = 160
Set DX_HIP_LIB to time another build of the library."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'ubisoft-laforge-daft-exprt_amd')); sys.path.insert(0, os.path.join(ROOT, 'tools'))
from daft_exprt import ops
from bench_ops import timeit
import bench
from daft_exprt.data_loader import synthetic_batch
dev = torch.device('cuda:0')
torch.manual_seed(0)
hp = bench.make_hparams(48, 'bf16')
cases = [('full128', torch.full((64,), 512), 512)]
for i in range(4):
    cb = synthetic_batch(hp, 48, seed=1234 + 1000 * i, t_min=1, t_max=1000, force_first_full=True)
    cases.append((f'c2[{i}]', cb[9].clone(), int(cb[9].max())))
    if i == 0:
        cases.append(('phoneme', cb[5].clone(), int(cb[5].max())))
out = []
for name, lens, N in cases:
    lens = lens.to(dev).long(); B = lens.numel()
    x = torch.randn(B, N, 1024, device=dev).to(torch.bfloat16)
    wp = ops.pack_conv_weight(torch.randn(128, 1024, 3, device=dev) / 3072 ** 0.5, torch.bfloat16); wf = ops.pack_frag_major(wp)
    bias, g, bt = torch.zeros(128, device=dev), torch.ones(128, device=dev), torch.zeros(128, device=dev)
    res = torch.randn(B, N, 128, device=dev)
    w2 = ops.pack_conv_weight(torch.randn(384, 128, device=dev) / 128 ** 0.5, torch.bfloat16); b2 = torch.zeros(384, device=dev)
    plan = ops.conv_tile_plan(lens, N)
    t = timeit(lambda: ops.conv1d_ln(x, wp, bias, res, g, bt, lens, save=True, p_pre=0.1, seed_pre=5, lp_copy=True, plan=plan, w_frag=wf, w2_packed=w2, b2=b2))
    rows = int(lens.sum())
    out.append(f'{name}: B={B} N={N} rows={rows}: {t * 1e3:6.1f} us  {2 * rows * 3072 * 128 / t / 1e9:6.0f} TFLOP/s')
print(os.environ.get('DX_HIP_LIB', 'default'), ' | '.join(out))
