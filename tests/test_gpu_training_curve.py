"""Training BEHAVIOUR of the bf16 operand mode (the bench's) against the exact-fp32 operand mode of the same kernels: 300 optimizer
steps of the full model on a small synthetic corpus (8 utterances per batch, a pool of 12 batches cycled: the model can fit it),
same initial weights, same batches, same dropout seeds (dropout ON, the reference's p = 0.1), the reference's schedule from
iteration 1 (`train.py:139-151`: 1e-4 warming up).  The reference trains 370 k iterations in fp32 (hparams.py:63,
train.py:368-401); what a user of the bf16 mode needs is that the loss CURVE is the fp32 one: the total loss and the mel terms,
averaged over windows of 25 steps, stay within 2 % of the fp32 run from step 100 on, both curves fall, and nothing goes
non-finite.  (Individual steps differ: bf16 rounding moves dropout-perturbed activations, and the two trajectories decorrelate
at the 1e-3 level within tens of steps -- the windowed mean is the quantity with a meaning.)"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _curve(mode, steps=300):
    from daft_exprt.data_loader import synthetic_batch
    from daft_exprt.model import DaftExprt
    from daft_exprt.train import Trainer
    from tests.util import make_hparams
    hp = make_hparams(compute_dtype=mode, batch_size=8, accumulation_steps=1)
    torch.manual_seed(hp.seed)
    model = DaftExprt(hp).to(DEV).train()
    trainer = Trainer(model, hp, 1)
    pool = []
    for i in range(12):
        cb = synthetic_batch(hp, 8, seed=4000 + i, t_max=240, force_first_full=False, l_range=(12, 40))
        inputs, targets, _ = model.parse_batch(DEV, cb)
        pool.append((inputs, targets))
    hist = []
    for it in range(1, steps + 1):
        terms, gn = trainer.step([pool[(it - 1) % len(pool)]], it)
        hist.append(torch.cat((terms, gn.sqrt())))
    out = torch.stack(hist).cpu().numpy()
    assert np.isfinite(out).all()
    return out


def test_bf16_loss_curve_tracks_the_fp32_curve_over_300_steps():
    a, b = _curve('fp32'), _curve('bf16')
    win = 25
    sm = lambda x: np.convolve(x, np.ones(win) / win, mode='valid')
    worst = {}
    for name, col in (('total', 7), ('mel_l1', 5), ('mel_l2', 6), ('duration', 2)):
        fa, fb = sm(a[:, col]), sm(b[:, col])
        rel = np.abs(fb - fa)[100 - win:] / np.abs(fa[100 - win:])
        worst[name] = float(rel.max())
        assert fa[-1] < 0.9 * fa[0] and fb[-1] < 0.9 * fb[0], (name, fa[0], fa[-1], fb[0], fb[-1])     # both runs learn
    print('windowed (25 steps) relative deviation of the bf16 curve from the fp32 curve, steps 100..300:', {k: f'{v:.3%}' for k, v in worst.items()},
          '| total loss fp32 %.4f -> %.4f, bf16 %.4f -> %.4f' % (a[0, 7], a[-1, 7], b[0, 7], b[-1, 7]))
    # (the duration term is ~1 % of the total and noisy at this corpus size: measured 14 % apart in windows where the total differs by 0.2 %)
    assert worst['total'] <= 0.02 and worst['mel_l1'] <= 0.02 and worst['mel_l2'] <= 0.02 and worst['duration'] <= 0.3, worst
