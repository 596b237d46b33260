"""Development aid: per-tensor gradient errors of the HIP path vs the CPU oracle for one synthetic training batch.
usage: python tools/dbg_parity.py MODE BATCH T_MIN T_MAX SEED N_SPEAKERS [emul]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'ubisoft-laforge-daft-exprt_amd')]
from tests import test_gpu_parity_at_size as T   # noqa: E402
from tests.util import make_hparams, no_dropout   # noqa: E402

mode, B, tmin, tmax, seed, nspk = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
emul = len(sys.argv) > 7
from daft_exprt.data_loader import synthetic_batch
from daft_exprt.loss import DaftExprtLoss
from daft_exprt.model import DaftExprt
hp = no_dropout(make_hparams(speakers=[f's{i}' for i in range(nspk)], batch_size=B, accumulation_steps=1, compute_dtype=mode))
torch.manual_seed(hp.seed)
model = DaftExprt(hp)
state = {k: v.detach().clone() for k, v in model.state_dict().items()}
model = model.to('cuda:0').train()
cb = synthetic_batch(hp, B, seed=seed, t_min=tmin, t_max=tmax, force_first_full=True)
inputs, targets, _ = model.parse_batch('cuda:0', cb)
rows = T._keep_rows(inputs, min(B, 8))
w = DaftExprtLoss(0, hp).weights(20000)
hip = T._hip_full_batch(model, inputs, targets, w, rows)
ora = T._oracle_slice(hp, state, inputs, rows, B, 20000, torch.bfloat16 if emul else None)
print('preds', {k: f'{T._rel(hip[0][k], ora[0][k]):.2e}' for k in ora[0]})
gmax = max(float(g.abs().max()) for g in ora[2].values())
for name, ref in ora[2].items():
    got = hip[2][name]
    err = float((got - ref).abs().max())
    print(f'{err / (float(ref.abs().max()) + 1e-30):9.2e} {err:9.2e} {float(ref.abs().max()):9.2e}  {name}')
print('gmax', gmax, 'lengths T', inputs[9].tolist(), 'L', inputs[5].tolist())
