"""No kernel may read a row its producer did not write.  Dead rows (past length + conv halo) are zero-filled only below
`dx_fill_end` (csrc/dx_common.h: as far as a consumer's last tile can reach); everything past it stays UNWRITTEN.  With
`config.POISON` every buffer `ops` allocates is pre-filled with NaN, so a kernel that reads an unwritten row -- a tile that reaches
past the fill end, a contraction that multiplies a zero gradient row with a garbage activation row -- turns predictions or
gradients into NaN or moves them.  The poisoned step must reproduce the clean step: predictions bit for bit (fully padded,
user-visible tensors included), loss terms and gradients to atomics order."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _step(model, inputs, targets, weights, poison, bounds=None):
    from daft_exprt import config
    old = config.POISON
    config.POISON = poison
    try:
        model.zero_grad()
        model._step_id = 3
        terms = model.forward_backward(inputs, targets, weights, bounds=bounds)
        torch.cuda.synchronize()
    finally:
        config.POISON = old
    logits, films, (dur, energy, pitch), mel, w = model.last_outputs
    return terms.clone(), [t.clone() for t in (mel, dur, energy, pitch, logits, w)], model._gflat.clone()


@pytest.mark.parametrize('mode,dropout', [('bf16', True), ('fp32', False)])
def test_poisoned_buffers_do_not_reach_any_output(mode, dropout):
    from daft_exprt.data_loader import synthetic_batch
    from daft_exprt.loss import DaftExprtLoss
    from daft_exprt.model import DaftExprt
    from tests.util import make_hparams, no_dropout
    hp = make_hparams(compute_dtype=mode, batch_size=12)
    if not dropout:
        hp = no_dropout(hp)
    torch.manual_seed(3)
    model = DaftExprt(hp).to(DEV).train()
    # utterance 0 pins the padded length at 900 frames; the others are short: most of their padding lies past the fill end
    cb = synthetic_batch(hp, 12, seed=5, t_max=900, force_first_full=True, l_range=(10, 50))
    inputs, targets, _ = model.parse_batch(DEV, cb)
    assert int(inputs[9].max()) == 900 and int(inputs[9].min()) < 400
    weights = DaftExprtLoss(0, hp).weights(20000)
    t0, o0, g0 = _step(model, inputs, targets, weights, False)
    t1, o1, g1 = _step(model, inputs, targets, weights, True)
    assert bool(torch.isfinite(g1).all()) and bool(torch.isfinite(t1).all())
    for a, b in zip(o0, o1):
        assert bool(torch.isfinite(b).all())
        assert torch.equal(a, b), float((a - b).abs().max())
    assert torch.allclose(t0, t1, rtol=1e-5, atol=0.)
    assert float((g0 - g1).norm()) <= 1e-4 * float(g0.norm()), float((g0 - g1).norm()) / float(g0.norm())


@pytest.mark.parametrize('mode', ['fp32', 'bf16'])
def test_poisoned_grouped_step(mode):
    ''' (bf16: the kernels that take the bounds only in that mode -- bit-mask ReLU conv, split-K LayerNorm GEMMs, the wide plan: ADVICE r5) '''
    from daft_exprt.data_loader import group_micro_batches, synthetic_batch
    from daft_exprt.loss import DaftExprtLoss
    from daft_exprt.model import DaftExprt
    from tests.util import make_hparams, no_dropout
    hp = no_dropout(make_hparams(compute_dtype=mode, batch_size=4, accumulation_steps=3))
    torch.manual_seed(3)
    model = DaftExprt(hp).to(DEV).train()
    mbs = []
    for k, t_max in enumerate((700, 256, 130)):
        cb = synthetic_batch(hp, 4, seed=50 + k, t_max=t_max, force_first_full=True, l_range=(10 + 10 * k, 40 + 10 * k))
        inputs, targets, _ = model.parse_batch(DEV, cb)
        mbs.append((inputs, targets))
    g = group_micro_batches(mbs)
    weights = DaftExprtLoss(0, hp).weights(20000)
    t0, o0, g0 = _step(model, g.inputs, g.targets, weights, False, bounds=g.bounds)
    t1, o1, g1 = _step(model, g.inputs, g.targets, weights, True, bounds=g.bounds)
    assert bool(torch.isfinite(g1).all())
    for a, b in zip(o0, o1):
        assert torch.equal(a, b), float((a - b).abs().max())
    assert float((g0 - g1).norm()) <= 1e-4 * float(g0.norm())


def test_poisoned_inference():
    from daft_exprt import config
    from daft_exprt.data_loader import centre_duration_head, synthetic_inference_batch
    from daft_exprt.model import DaftExprt
    from tests.util import make_hparams
    hp = make_hparams(compute_dtype='bf16')
    torch.manual_seed(3)
    model = DaftExprt(hp).to(DEV).eval()
    centre_duration_head(model)
    hp.stats = {f'spk {i}': {'pitch': {'mean': 5.0, 'std': 0.3}} for i in range(hp.n_speakers)}
    inputs = tuple(t.to(DEV) for t in synthetic_inference_batch(hp, 10, seed=9, l_range=(8, 120), t_ref_range=(60, 900)))
    outs = []
    for poison in (False, True):
        config.POISON = poison
        try:
            enc, dec, w = model.inference(tuple(t.clone() for t in inputs), 'add', hp)
            torch.cuda.synchronize()
        finally:
            config.POISON = False
        outs.append([dec[0].clone(), dec[1].clone(), enc[0].clone(), enc[1].clone(), w.clone()])
    for a, b in zip(*outs):
        assert torch.equal(a, b)
