#!/usr/bin/env python
"""How long does the HOST need to enqueue one training step (development aid)?  Runs steps back-to-back without
synchronising and reports host time per step vs GPU time per step."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'ubisoft-laforge-daft-exprt_amd'))
import bench
from daft_exprt.data_loader import synthetic_batch
from daft_exprt.model import DaftExprt
from daft_exprt.train import Trainer
from daft_exprt import _hip as H

hp = bench.make_hparams(48, 'bf16')
dev = torch.device('cuda:0')
model = DaftExprt(hp).to(dev).train()
tr = Trainer(model, hp, 1)
cb = synthetic_batch(hp, 48, seed=1234, t_max=1000, force_first_full=True)
inputs, targets, _ = model.parse_batch(dev, cb)
for i in range(3):
    tr.step([(inputs, targets)], 20000 + i)
torch.cuda.synchronize()
# count launches through the C ABI
calls = {'n': 0}
lib = H.lib()
t0 = time.perf_counter()
for i in range(5):
    tr.step([(inputs, targets)], 20000 + i)
t_host = (time.perf_counter() - t0) / 5
torch.cuda.synchronize()
t_all = (time.perf_counter() - t0) / 5
print(f'host enqueue time per step: {t_host*1e3:.2f} ms; wall per step incl. GPU drain: {t_all*1e3:.2f} ms')
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
tr.step([(inputs, targets)], 20010)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('cumulative').print_stats(18)
