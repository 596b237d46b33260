"""Architectures other than the published one (`hparams.py:90-128` builds any head count / FF width): the variants this build
accepts -- 4 heads and 1 head (head sizes 32 / 128 on the generic attention templates), other FF / pre-net / predictor widths
through the generic GEMM kernels -- against the CPU oracle (autograd of the restated reference), forward and every
parameter gradient, fp32 operand mode (exact-parity arithmetic) and bf16."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import daft_exprt_cpu as O
from oracle.fill import fill_params
from tests.util import gradient_report, make_hparams, no_dropout

DEV = 'cuda:0'


def _hparams(mode, heads, ff, pre, pred, kernels=(3, 3, 3, 3)):
    hp = no_dropout(make_hparams(compute_dtype=mode, batch_size=4))
    hp.prosody_encoder.update(attn_nb_heads=heads[0], conv_channels=pre, conv_kernel=kernels[0])
    hp.phoneme_encoder.update(attn_nb_heads=heads[1], conv_channels=ff, conv_kernel=kernels[1])
    hp.frame_decoder.update(attn_nb_heads=heads[2], conv_channels=ff, conv_kernel=kernels[2])
    hp.local_prosody_predictor.update(conv_channels=pred, conv_kernel=kernels[3] if len(kernels) > 3 else 3)
    return hp


def _run(mode, heads, ff, pre, pred, kernels=(3, 3, 3, 3)):
    from daft_exprt.data_loader import synthetic_batch
    from daft_exprt.loss import DaftExprtLoss
    from daft_exprt.model import DaftExprt
    hp = _hparams(mode, heads, ff, pre, pred, kernels)
    state = fill_params(O.param_shapes(hp))
    model = DaftExprt(hp)
    model.load_state_dict(state)
    model = model.to(DEV).train()
    cb = synthetic_batch(hp, 4, seed=31, t_max=200, force_first_full=True, l_range=(12, 40))
    inputs, targets, _ = model.parse_batch(DEV, cb)
    weights = DaftExprtLoss(0, hp).weights(20000)
    model.zero_grad()
    terms = model.forward_backward(inputs, targets, weights)
    torch.cuda.synchronize()
    logits, films, (dur, energy, pitch), mel, _ = model.last_outputs
    hip = ({'speaker': logits, 'duration': dur, 'energy': energy, 'pitch': pitch, 'mel': mel}, terms.cpu(),
           {n: p.grad.detach().cpu().clone() for n, p in model.named_parameters()})
    P = {k: v.detach().clone().float().requires_grad_(True) for k, v in state.items()}
    cin = tuple(t.cpu() for t in inputs)
    out = O.forward(P, hp, cin, training=True)
    total, oterms = O.loss(hp, out, (cin[1], cin[3], cin[4], cin[8], cin[10]), 20000)
    grads = torch.autograd.grad(total, list(P.values()), allow_unused=True)
    ora = ({'speaker': out[0], 'duration': out[2][0], 'energy': out[2][1], 'pitch': out[2][2], 'mel': out[3][0]}, float(total),
           {k: (g if g is not None else torch.zeros_like(P[k])) for k, g in zip(P, grads)})
    return hip, ora


@pytest.mark.parametrize('heads,ff,pre,pred', [((4, 1, 4), 1024, 1024, 256),      # 4 heads (d_h 32) and 1 head (d_h 128)
                                               ((8, 2, 2), 512, 512, 128),        # the published heads on other conv widths
                                               ((2, 4, 1), 256, 256, 256)])
def test_variant_architectures_match_the_oracle_fp32(heads, ff, pre, pred):
    (hp_p, hp_t, hp_g), (or_p, or_t, or_g) = _run('fp32', heads, ff, pre, pred)
    for k in or_p:
        a, b = hp_p[k].detach().float().cpu(), or_p[k].detach().float()
        assert a.shape == b.shape
        err = float((a - b).abs().max() / (b.abs().max() + 1e-12))
        assert err <= 2e-4, (k, err)
    assert abs(float(hp_t[7]) - or_t) <= 1e-4 * abs(or_t), (float(hp_t[7]), or_t)
    worst = gradient_report(hp_g, or_g, rel=2e-3, floor=2e-5)
    assert worst[0][0] <= 1., worst[:5]


@pytest.mark.parametrize('kernels', [(1, 1, 1, 1), (3, 1, 3, 3), (1, 3, 1, 3), (3, 3, 3, 1)])   # prosody encoder, phoneme encoder, decoder, predictor
def test_conv_kernel_1_matches_the_oracle_fp32(kernels):
    ''' `conv_kernel` 1 (hparams.py:90-128 takes any odd size; model.py:75-94): the FF block's convolutions as linear layers, the
        pre-net and the scalar embeddings of the prosody encoder with one tap -- on the k = 1 GEMM kernels '''
    (hp_p, hp_t, hp_g), (or_p, or_t, or_g) = _run('fp32', (8, 2, 2), 1024, 1024, 256, kernels)
    for k in or_p:
        a, b = hp_p[k].detach().float().cpu(), or_p[k].detach().float()
        assert a.shape == b.shape
        err = float((a - b).abs().max() / (b.abs().max() + 1e-12))
        assert err <= 2e-4, (k, err)
    assert abs(float(hp_t[7]) - or_t) <= 1e-4 * abs(or_t), (float(hp_t[7]), or_t)
    worst = gradient_report(hp_g, or_g, rel=2e-3, floor=2e-5)
    assert worst[0][0] <= 1., worst[:5]


def test_conv_kernel_1_runs_in_bf16():
    (hp_p, hp_t, hp_g), (or_p, or_t, or_g) = _run('bf16', (8, 2, 2), 1024, 1024, 256, (1, 1, 1))
    for k in or_p:
        a, b = hp_p[k].detach().float().cpu(), or_p[k].detach().float()
        err = float((a - b).abs().max() / (b.abs().max() + 1e-12))
        assert err <= 6e-2, (k, err)
    assert abs(float(hp_t[7]) - or_t) <= 2e-2 * abs(or_t)


def test_variant_architecture_runs_in_bf16():
    (hp_p, hp_t, hp_g), (or_p, or_t, or_g) = _run('bf16', (4, 1, 4), 512, 512, 128)
    for k in or_p:
        a, b = hp_p[k].detach().float().cpu(), or_p[k].detach().float()
        err = float((a - b).abs().max() / (b.abs().max() + 1e-12))
        assert err <= 6e-2, (k, err)
    assert abs(float(hp_t[7]) - or_t) <= 2e-2 * abs(or_t)


def test_unsupported_architectures_say_so():
    from daft_exprt.model import DaftExprt
    hp = make_hparams()
    hp.phoneme_encoder['attn_nb_heads'] = 3
    with pytest.raises(NotImplementedError):
        DaftExprt(hp)
    hp = make_hparams()
    hp.prosody_encoder['hidden_embed_dim'] = 256
    with pytest.raises(NotImplementedError):
        DaftExprt(hp)
    hp = make_hparams()
    hp.frame_decoder['conv_kernel'] = 5
    with pytest.raises(NotImplementedError):
        DaftExprt(hp)
    hp = make_hparams()
    hp.gaussian_upsampling_module['conv_kernel'] = 1
    with pytest.raises(NotImplementedError):
        DaftExprt(hp)
    hp = make_hparams()
    hp.local_prosody_predictor['conv_kernel'] = 5
    with pytest.raises(NotImplementedError):
        DaftExprt(hp)

