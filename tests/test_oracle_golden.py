"""Pin the CPU oracle against the golden fixtures captured from the imported reference
(tools/gen_goldens.py).  CPU only; these are the tests that make `oracle/` trustworthy.
Tolerances: fp32 forward 2e-5 abs / 1e-4 rel (summation order differs between the explicit
restatement and ATen's fused kernels); integer tensors exact."""
import os

import numpy as np
import pytest
import torch

from oracle import daft_exprt_cpu as O
from oracle.fill import fill_params
from tests.util import make_hparams, load_inputs, INPUT_NAMES, no_dropout

torch.set_num_threads(8)


def _close(a, b, rtol=1e-4, atol=2e-5, what=''):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    err = np.abs(a - b)
    tol = atol + rtol * np.abs(b)
    assert (err <= tol).all(), f'{what}: max abs err {err.max():.3e} (ref max {np.abs(b).max():.3e})'


@pytest.fixture(scope='module')
def fx_eval(golden_dir):
    return np.load(os.path.join(golden_dir, 'forward_eval.npz'))


def test_param_table_matches_reference_count():
    hp = make_hparams()
    shapes = O.param_shapes(hp)
    assert len(shapes) == 193
    assert sum(int(np.prod(s)) for s in shapes.values()) == 14727153


def test_forward_eval(fx_eval):
    hp = make_hparams()
    P = fill_params(O.param_shapes(hp))
    inputs = load_inputs(fx_eval)
    with torch.no_grad():
        spk, film, enc, dec, weights = O.forward(P, hp, inputs, training=False)
    _close(spk, fx_eval['out_speaker_preds'], what='speaker_preds')
    _close(film[1], fx_eval['out_encoder_film'], what='encoder_film')
    _close(film[2], fx_eval['out_prosody_pred_film'], what='pp_film')
    _close(film[3], fx_eval['out_decoder_film'], what='decoder_film')
    _close(enc[0], fx_eval['out_duration'], what='duration')
    _close(enc[1], fx_eval['out_energy'], what='energy')
    _close(enc[2], fx_eval['out_pitch'], what='pitch')
    _close(dec[0], fx_eval['out_mel'], rtol=2e-4, atol=5e-5, what='mel')
    _close(weights, fx_eval['out_weights'], what='weights')
    # intermediates
    emb, ef, pf, df = O.prosody_encoder(P, hp, inputs[6], inputs[7], inputs[8], inputs[10], inputs[9], False)
    _close(emb.detach(), fx_eval['mid_prosody_embeddings'], what='prosody_embeddings')
    enc_out = O.phoneme_encoder(P, hp, inputs[0], ef, inputs[5], False)
    _close(enc_out.detach(), fx_eval['mid_enc_outputs'], what='enc_outputs')


def test_loss_terms(fx_eval):
    hp = make_hparams()
    inputs = load_inputs(fx_eval)
    out = (torch.from_numpy(fx_eval['out_speaker_preds']),
           [torch.from_numpy(fx_eval['out_post_multipliers']), None, None, None],
           [torch.from_numpy(fx_eval[f'out_{k}']) for k in ('duration', 'energy', 'pitch')] + [inputs[5]],
           [torch.from_numpy(fx_eval['out_mel']), inputs[9]], None)
    targets = (inputs[1], inputs[3], inputs[4], inputs[8], inputs[10])
    keys = ('speaker_loss', 'post_mult_loss', 'duration_loss', 'energy_loss', 'pitch_loss', 'mel_spec_l1_loss', 'mel_spec_l2_loss')
    for it in (0, 1, 5000, 10000, 20000):
        total, terms = O.loss(hp, out, targets, it)
        _close(float(total), fx_eval[f'loss_total_it{it}'], rtol=1e-5, atol=1e-6, what=f'total@{it}')
        _close([float(terms[k]) for k in keys], fx_eval[f'loss_terms_it{it}'], rtol=1e-5, atol=1e-7, what=f'terms@{it}')


def test_collate_train_contract(fx_eval):
    items = []
    for i in range(4):
        g = lambda nm: torch.from_numpy(fx_eval[f'item{i}_{nm}'])
        items.append([g('symbols'), g('dur_float'), g('dur_int'), g('sym_energy'), g('sym_pitch'), g('frames_energy'),
                      g('frames_pitch'), g('mel'), int(fx_eval[f'item{i}_speaker']), f'dir{i}', f'file{i}'])
    batch = O.collate_train(items)
    for name, got in zip(INPUT_NAMES, batch[:11]):
        want = fx_eval[f'in_{name}']
        assert got.numpy().dtype == want.dtype, name
        assert np.array_equal(got.numpy(), want), name
    assert list(batch[11]) == list(fx_eval['collate_dirs'])
    assert list(batch[12]) == list(fx_eval['collate_files'])


def test_gradients_and_adam(golden_dir):
    fx = np.load(os.path.join(golden_dir, 'train_nodrop.npz'))
    hp = no_dropout(make_hparams())
    shapes = O.param_shapes(hp)
    P = {k: v.requires_grad_(True) for k, v in fill_params(shapes).items()}
    assert list(P.keys()) == list(fx['param_names'])
    inputs = load_inputs(fx)
    targets = (inputs[1], inputs[3], inputs[4], inputs[8], inputs[10])
    out = O.forward(P, hp, inputs, training=True)
    total, _ = O.loss(hp, out, targets, 20000)
    _close(float(total), fx['loss_total'], rtol=2e-5, what='train loss')
    grads = torch.autograd.grad(total, list(P.values()))
    norms = np.array([g.norm().item() for g in grads])
    _close(norms, fx['grad_norms'], rtol=2e-3, atol=1e-6, what='grad norms')
    heads = np.stack([np.pad(g.reshape(-1)[:32].numpy(), (0, max(0, 32 - g.numel()))) for g in grads])
    scale = np.abs(fx['grad_heads']).max(axis=1, keepdims=True) + 1e-12
    assert (np.abs(heads - fx['grad_heads']) / scale).max() < 5e-3
    tot = float(torch.sqrt(sum((g.double() ** 2).sum() for g in grads)))
    _close(tot, fx['grad_total_norm'], rtol=1e-3, what='total grad norm')
    for key in fx.files:
        if key.startswith('grad_full__'):
            g = grads[list(P.keys()).index(key[len('grad_full__'):])]
            ref = fx[key]
            assert np.abs(g.numpy() - ref).max() <= 5e-3 * np.abs(ref).max() + 1e-7, key
    # three Adam steps from iteration 1 (train.py:299-301, 391-401, 486-494)
    P = {k: v.detach().clone().requires_grad_(True) for k, v in fill_params(shapes).items()}
    before = {k: v.detach().clone() for k, v in P.items()}
    state = {'step': 0, 'm': {k: torch.zeros_like(v) for k, v in P.items()}, 'v': {k: torch.zeros_like(v) for k, v in P.items()}}
    losses, gnorms = [], []
    for step in range(3):
        out = O.forward(P, hp, inputs, training=True)
        total, _ = O.loss(hp, out, targets, 1 + step)
        grads = torch.autograd.grad(total, list(P.values()))
        gnorms.append(float(torch.sqrt(sum((g.double() ** 2).sum() for g in grads))))
        with torch.no_grad():
            O.adam_step({k: v for k, v in P.items()}, dict(zip(P.keys(), grads)), state, O.learning_rate(hp, 1 + step),
                        hp.betas, hp.epsilon, hp.weight_decay)
        losses.append(float(total))
    _close(losses, fx['adam_losses'], rtol=5e-3, what='adam losses')
    _close(gnorms, fx['adam_grad_norms'], rtol=2e-2, what='adam grad norms')
    deltas = np.array([(P[k].detach() - before[k]).norm().item() for k in P])
    _close(deltas, fx['adam_delta_norms'], rtol=2e-2, atol=1e-7, what='adam param deltas')


def test_inference(golden_dir):
    fx = np.load(os.path.join(golden_dir, 'inference.npz'))
    hp = make_hparams()
    hp.stats = {f'spk {i}': {'pitch': {'mean': float(fx['stats_pitch_mean'][i]), 'std': float(fx['stats_pitch_std'][i])}}
                for i in range(11)}
    P = fill_params(O.param_shapes(hp))
    names = ['symbols', 'dur_factors', 'energy_factors', 'pitch_factors', 'input_lengths', 'energy_refs', 'pitch_refs',
             'mel_spec_refs', 'ref_lengths', 'speaker_ids']
    for transform in ('add', 'multiply'):
        inputs = tuple(torch.from_numpy(fx[f'{transform}_in_{n}']) for n in names)
        enc, dec, weights = O.inference(P, hp, inputs, transform)
        assert np.array_equal(enc[1].numpy(), fx[f'{transform}_out_durations_int'])       # bit-exact integer path
        assert np.array_equal(dec[1].numpy(), fx[f'{transform}_out_output_lengths'])
        _close(enc[0], fx[f'{transform}_out_duration'], what='duration')
        _close(enc[2], fx[f'{transform}_out_energy'], what='energy')
        _close(enc[3], fx[f'{transform}_out_pitch'], rtol=2e-4, atol=5e-5, what='pitch')
        _close(dec[0], fx[f'{transform}_out_mel'], rtol=2e-4, atol=5e-5, what='mel')
        _close(weights, fx[f'{transform}_out_weights'], what='weights')


@pytest.mark.parametrize('transform', ['add', 'multiply'])
def test_collate_inference_replays_the_reference(golden_dir, transform):
    ''' oracle.collate_inference (symbol-id sequences in, padded tensors out) against the reference's collate_tensors on
        the raw driver inputs of tests/golden/inference_collate.npz '''
    import json
    from tests.util import COLLATE_NAMES, load_driver_fixture
    hp = make_hparams()
    sentences, dur_f, en_f, pi_f, refs, spk, names, fx = load_driver_fixture(golden_dir, transform)
    ids = [[hp.symbols.index(p) for it in s for p in (it if isinstance(it, list) else [it])] for s in sentences]
    col = O.collate_inference(ids, dur_f, en_f, pi_f, transform, refs, spk, names)
    for nm, got in zip(COLLATE_NAMES, col[:-1]):
        want = fx[f'{transform}_col_{nm}']
        assert got.numpy().dtype == want.dtype and np.array_equal(got.numpy(), want), nm
    assert list(col[-1]) == json.loads(str(fx[f'{transform}_col_file_names_json']))


def test_duration_to_integer_kats(golden_dir):
    fx = np.load(os.path.join(golden_dir, 'duration_to_integer.npz'))
    durs, d_off, ints, i_off = fx['durs'], fx['durs_off'], fx['ints'], fx['ints_off']
    n_err = 0
    for c in range(len(d_off) - 1):
        spans, end_prev = [], 0.
        for d in durs[d_off[c]: d_off[c + 1]].tolist():
            spans.append([end_prev, end_prev + d])
            end_prev += d
        want = ints[i_off[c]: i_off[c + 1]].tolist()
        try:
            got = O.duration_to_integer(spans)
        except IndexError:
            got = [-1]
            n_err += 1
        assert got == want, (c, got, want)
    assert n_err == 19


def test_get_int_durations(golden_dir):
    fx = np.load(os.path.join(golden_dir, 'get_int_durations.npz'))
    hp = make_hparams()
    p, ints = O.get_int_durations(torch.from_numpy(fx['preds'].copy()), hp)
    assert np.array_equal(p.numpy(), fx['thresholded'])
    assert np.array_equal(ints.numpy(), fx['ints'])


def test_schedules(golden_dir):
    fx = np.load(os.path.join(golden_dir, 'schedules.npz'))
    hp = make_hparams()
    for it, lr, adv in zip(fx['iterations'], fx['lr'], fx['adv']):
        assert O.learning_rate(hp, int(it)) == pytest.approx(float(lr), rel=1e-12)
        assert O.adversarial_weight(hp, int(it)) == pytest.approx(float(adv), rel=1e-12)
    # the KATs quoted in SURVEY 8c
    assert O.learning_rate(hp, 5000) == pytest.approx(5.5e-4)
    assert O.learning_rate(hp, 40000) == pytest.approx(5e-4)
    assert O.adversarial_weight(hp, 5000) == pytest.approx(5e-3)
