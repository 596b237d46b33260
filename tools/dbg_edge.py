import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/ubisoft-laforge-daft-exprt_amd')
import torch, numpy as np
from tests.test_gpu_edge_cases import _batch, CASES
from tests.util import make_hparams, no_dropout
from oracle import daft_exprt_cpu as O
from daft_exprt.loss import DaftExprtLoss
from daft_exprt.model import DaftExprt
DEV='cuda:0'
case = sys.argv[1] if len(sys.argv) > 1 else 'tile_boundaries'
Ls, Ts = CASES[case]
if len(sys.argv) > 2:
    Ls = [int(x) for x in sys.argv[2].split(',')]; Ts = [int(x) for x in sys.argv[3].split(',')]
hp = no_dropout(make_hparams(compute_dtype='fp32'))
torch.manual_seed(11)
model = DaftExprt(hp)
P = {k: v.detach().clone().requires_grad_(True) for k, v in model.state_dict().items()}
model = model.to(DEV).train()
batch = _batch(hp, Ls, Ts, seed=len(Ls) * 100 + Ts[0])
inputs, targets, _ = model.parse_batch(DEV, batch)
crit = DaftExprtLoss(0, hp)
model.zero_grad()
loss, _ = crit(model(inputs), targets, 20000)
loss.backward(); torch.cuda.synchronize()
cin = tuple(t.cpu() for t in inputs)
ref = O.forward(P, hp, cin, training=True)
ref_loss, _ = O.loss(hp, ref, (cin[1], cin[3], cin[4], cin[8], cin[10]), 20000)
grads = torch.autograd.grad(ref_loss, list(P.values()))
rows = []
for (name, p), g in zip(model.named_parameters(), grads):
    gn, rn = float(p.grad.norm()), float(g.norm())
    rows.append((abs(gn - rn) / (rn + 1e-12), name, gn, rn))
rows.sort(reverse=True)
for r in rows[:14]: print(f'{r[0]:.4f} {r[1]:70s} {r[2]:.5e} {r[3]:.5e}')
