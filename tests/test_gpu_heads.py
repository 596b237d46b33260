"""The FiLM head and the speaker classifier as fused launches (`dx_film_head_fwd / _bwd`, `dx_classifier_fwd / _bwd`) against the chain of
small launches they replace (gather_add, linear_small, film_assemble and their backward counterparts, each pinned against the reference
in test_gpu_model / test_gpu_parity_at_size): same forward values bit for bit (same summation order), gradients equal up to the fp32
summation order of the backward reductions.  Reference: model.py:27-54, 276-292 (classifier + gradient reversal), 419-462 (FiLM)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _run(fused, speakers, mode='fp32', B=6):
    from daft_exprt import ops
    from daft_exprt.data_loader import synthetic_batch
    from daft_exprt.model import DaftExprt
    from tests.util import make_hparams, no_dropout
    hp = no_dropout(make_hparams(compute_dtype=mode, batch_size=B, speakers=speakers))
    torch.manual_seed(5)
    model = DaftExprt(hp).to(DEV).train()
    cb = synthetic_batch(hp, B, seed=77, t_max=180, force_first_full=True, l_range=(6, 24))
    inputs, targets, _ = model.parse_batch(DEV, cb)
    old = ops.USE_FUSED_HEADS
    ops.USE_FUSED_HEADS = fused
    try:
        model.zero_grad()
        terms = model.forward_backward(inputs, targets, (1e-2, 1e-3, 1., 1., 1., 1.))
        torch.cuda.synchronize()
    finally:
        ops.USE_FUSED_HEADS = old
    logits, films, _, mel, _ = model.last_outputs
    return terms.clone(), logits.clone(), [f.clone() for f in films], mel.clone(), {n: p.grad.detach().clone() for n, p in model.named_parameters()}


@pytest.mark.parametrize('speakers', [[f'spk{i:02d}' for i in range(11)], ['LJ']])
def test_fused_heads_match_the_separate_launches(speakers):
    t0, l0, f0, m0, g0 = _run(False, speakers)
    t1, l1, f1, m1, g1 = _run(True, speakers)
    assert torch.equal(l0, l1) and all(torch.equal(a, b) for a, b in zip(f0, f1)) and torch.equal(m0, m1)       # forward: bit-identical
    assert torch.allclose(t0, t1, rtol=1e-6, atol=1e-7)
    gmax = max(float(g.abs().max()) for g in g0.values())
    for n in g0:
        err = float((g0[n] - g1[n]).abs().max())
        assert err <= 2e-5 * float(g0[n].abs().max()) + 2e-7 * gmax, (n, err, float(g0[n].abs().max()))
    touched = [n for n in g0 if n.startswith(('speaker_classifier', 'prosody_encoder.gammas', 'prosody_encoder.betas', 'prosody_encoder.spk_embedding',
                                              'prosody_encoder.post_multipliers'))]
    assert len(touched) == 12 and all(float(g1[n].abs().max()) > 0. for n in touched if 'classifier' not in n or len(speakers) > 1)


def test_more_speakers_than_the_fused_classifier_holds():
    ''' n_speakers - 1 > 128 logits do not fit the one-launch classifier's 128-wide tile: the model then takes the
        `linear_small_*` launches instead of raising, and the logits are those of the plain three-layer MLP (model.py:276-292) '''
    speakers = [f'spk{i:03d}' for i in range(140)]
    t1, l1, f1, m1, g1 = _run(True, speakers)
    assert l1.shape == (6, 140) and bool(torch.isfinite(t1).all())
    from daft_exprt.model import DaftExprt
    from tests.util import make_hparams, no_dropout
    hp = no_dropout(make_hparams(compute_dtype='fp32', batch_size=6, speakers=speakers))
    torch.manual_seed(5)
    model = DaftExprt(hp).to(DEV).train()
    assert not model._fused_classifier()
    for n in ('1', '3', '5'):
        assert float(g1[f'speaker_classifier.classifier.{n}.linear_layer.weight'].abs().max()) > 0.
