#!/usr/bin/env python
"""Development: what each output of the split-K conv + LayerNorm launch's epilogue costs (the main loop is the same in every row)."""
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
cb = synthetic_batch(hp, 48, seed=1234 + 3000, t_min=1, t_max=1000, force_first_full=True)
for name, lens, N in (('full128', torch.full((64,), 512), 512), ('c2[3]', cb[9].clone(), int(cb[9].max()))):
    lens = lens.to(dev).long(); B = lens.numel()
    x = torch.randn(B, N, 1024, device=dev).to(torch.bfloat16)
    wp = ops.pack_conv_weight(torch.randn(128, 1024, 3, device=dev) / 3072 ** 0.5, torch.bfloat16); wf = ops.pack_frag_major(wp)
    bias, g, bt = torch.zeros(128, device=dev), torch.ones(128, device=dev), torch.zeros(128, device=dev)
    res = torch.randn(B, N, 128, device=dev)
    w2 = ops.pack_conv_weight(torch.randn(384, 128, device=dev) / 128 ** 0.5, torch.bfloat16); b2 = torch.zeros(384, device=dev)
    plan = ops.conv_tile_plan(lens, N)
    dy = torch.randn(B, N, 128, device=dev); s_in = torch.randn(B, N, 128, device=dev); mean = torch.zeros(B * N, device=dev); rstd = torch.ones(B * N, device=dev)
    dg, db = torch.zeros(128, device=dev), torch.zeros(128, device=dev)
    wo = ops.pack_conv_weight(torch.randn(128, 128, device=dev) / 128 ** 0.5, torch.bfloat16)
    rows = []
    for label, kw in (('all (y, s, lp, qkv)', dict(save=True, lp_copy=True, w2_packed=w2, b2=b2)),
                      ('no qkv', dict(save=True, lp_copy=True)),
                      ('no qkv, no s', dict(save=False, lp_copy=True)),
                      ('lp only', dict(save=False, lp_copy=True, store_y=False)),
                      ('lp + qkv', dict(save=False, lp_copy=True, store_y=False, w2_packed=w2, b2=b2)),
                      ('all, no dropout', dict(save=True, lp_copy=True, w2_packed=w2, b2=b2, p_pre=0.))):
        kw.setdefault('p_pre', 0.1)
        t = timeit(lambda: ops.conv1d_ln(x, wp, bias, res, g, bt, lens, seed_pre=5, plan=plan, w_frag=wf, **kw))
        rows.append(f'{label}: {t * 1e3:5.1f}')
    t = timeit(lambda: ops.conv1d_lnbwd(x, wp, dy, s_in, mean, rstd, g, bt, lens, dg, db, p_pre=0.1, seed_pre=5, plan=plan, w_frag=wf, w2_packed=wo))
    rows.append(f'lnbwd + d_o: {t * 1e3:5.1f}')
    t = timeit(lambda: ops.conv1d_lnbwd(x, wp, dy, s_in, mean, rstd, g, bt, lens, dg, db, p_pre=0.1, seed_pre=5, plan=plan, w_frag=wf))
    rows.append(f'lnbwd: {t * 1e3:5.1f}')
    print(os.environ.get('DX_HIP_LIB', 'default')[-20:], name, ' | '.join(rows), 'us')
