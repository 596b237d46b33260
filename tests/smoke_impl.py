"""smoke(): one tiny training step + one tiny synthesis call of the HIP hot path on the GPU, checked against the
CPU oracle (fp32 operand mode so that the comparison is tight)."""
import numpy as np
import torch

from oracle import daft_exprt_cpu as O
from oracle.fill import fill_params
from tests.util import make_hparams, no_dropout


def run_smoke(device):
    from daft_exprt.data_loader import synthetic_batch
    from daft_exprt.loss import DaftExprtLoss
    from daft_exprt.model import DaftExprt
    hp = no_dropout(make_hparams(compute_dtype='fp32'))
    model = DaftExprt(hp)
    P = fill_params(O.param_shapes(hp))
    model.load_state_dict(P)
    model = model.to(device).train()
    batch = synthetic_batch(hp, 3, seed=7, t_max=96, force_first_full=True, l_range=(8, 20))
    inputs, targets, _ = model.parse_batch(device, batch)
    crit = DaftExprtLoss(0, hp)
    model.zero_grad()
    loss, terms = crit(model(inputs), targets, 20000)
    loss.backward()
    torch.cuda.synchronize()
    # oracle on the same batch
    Pc = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    cin = tuple(t.cpu() for t in inputs)
    out = O.forward(Pc, hp, cin, training=True)
    ref, _ = O.loss(hp, out, (cin[1], cin[3], cin[4], cin[8], cin[10]), 20000)
    grads = torch.autograd.grad(ref, list(Pc.values()))
    assert abs(float(loss) - float(ref)) <= 2e-4 * abs(float(ref)), (float(loss), float(ref))
    for (name, p), g in zip(model.named_parameters(), grads):
        gn, rn = float(p.grad.norm()), float(g.norm())
        assert abs(gn - rn) <= 5e-3 * rn + 1e-7, (name, gn, rn)
    print(f'[smoke] HIP train step on {torch.cuda.get_device_name(0)}: loss {float(loss):.6f} == oracle {float(ref):.6f}; '
          f'193 gradient norms match the CPU oracle')
