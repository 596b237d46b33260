"""End-to-end parity of the HIP model against the golden fixtures captured from the imported reference
and against the CPU oracle.  Tolerances (stated per mode):
  compute_dtype='fp32' (exact-fp32 MFMA operands): 2e-4 relative to the tensor's max magnitude
  compute_dtype='bf16' (bf16 MFMA operands, fp32 accumulate): 3e-2 relative to the tensor's max magnitude
Integer tensors (durations_int, output_lengths) are bit-exact in both modes where the float inputs to the
integer path agree (fp32 mode)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import daft_exprt_cpu as O
from oracle.fill import fill_params
from tests.util import make_hparams, load_inputs, no_dropout

DEV = 'cuda:0'
TOL = {'fp32': 2e-4, 'bf16': 3e-2}


def _rel(a, b):
    a = a.detach().float().cpu().numpy() if torch.is_tensor(a) else np.asarray(a)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-12))


def _model(hp):
    from daft_exprt.model import DaftExprt
    m = DaftExprt(hp)
    m.load_state_dict(fill_params(O.param_shapes(hp)))
    return m.to(DEV)


@pytest.mark.parametrize('mode', ['fp32', 'bf16'])
def test_forward_eval_matches_reference_golden(golden_dir, mode):
    fx = np.load(os.path.join(golden_dir, 'forward_eval.npz'))
    hp = make_hparams(compute_dtype=mode)
    m = _model(hp).eval()
    inputs = load_inputs(fx, DEV)
    spk, film, enc, dec, weights = m(inputs)
    torch.cuda.synchronize()
    tol = TOL[mode]
    errs = {'speaker_preds': _rel(spk, fx['out_speaker_preds']), 'encoder_film': _rel(film[1], fx['out_encoder_film']),
            'pp_film': _rel(film[2], fx['out_prosody_pred_film']), 'decoder_film': _rel(film[3], fx['out_decoder_film']),
            'duration': _rel(enc[0], fx['out_duration']), 'energy': _rel(enc[1], fx['out_energy']),
            'pitch': _rel(enc[2], fx['out_pitch']), 'mel': _rel(dec[0], fx['out_mel']), 'weights': _rel(weights, fx['out_weights'])}
    print(mode, errs)
    assert dec[0].shape == fx['out_mel'].shape
    # the alignment weights are the output bf16 operand rounding moves most (~10 % in the reference itself under
    # bf16 autocast; 2.6-3.9 % here depending on the summation order of the phoneme-level GEMMs): 6e-2 in bf16 mode, tol elsewhere
    bad = {k: v for k, v in errs.items() if not v <= (6e-2 if (k == 'weights' and mode == 'bf16') else tol)}
    assert not bad, f'{mode}: {bad} (tol {tol})'
    # padded positions are exactly zero, like the reference's masked_fill
    out_len = fx['in_output_lengths']
    mel = dec[0].cpu().numpy()
    for b, t in enumerate(out_len):
        assert not mel[b, :, t:].any()


@pytest.mark.parametrize('mode', ['fp32', 'bf16'])
def test_loss_terms_match_reference_golden(golden_dir, mode):
    from daft_exprt.loss import DaftExprtLoss, KEYS
    fx = np.load(os.path.join(golden_dir, 'forward_eval.npz'))
    hp = make_hparams(compute_dtype=mode)
    m = _model(hp).eval()
    inputs = load_inputs(fx, DEV)
    targets = (inputs[1], inputs[3], inputs[4], inputs[8], inputs[10])
    out = m(inputs)
    crit = DaftExprtLoss(0, hp)
    for it in (0, 1, 5000, 10000, 20000):
        total, indiv = crit(out, targets, it)
        ref_terms = fx[f'loss_terms_it{it}']
        got = np.array([indiv[k] for k in KEYS])
        tol = 1e-4 if mode == 'fp32' else 2e-2
        assert np.abs(got - ref_terms).max() <= tol * np.abs(ref_terms).max(), (it, got, ref_terms)
        assert abs(float(total) - float(fx[f'loss_total_it{it}'])) <= tol * abs(float(fx[f'loss_total_it{it}']))


@pytest.mark.parametrize('mode', ['fp32', 'bf16'])
def test_inference_matches_reference_golden(golden_dir, mode):
    fx = np.load(os.path.join(golden_dir, 'inference.npz'))
    hp = make_hparams(compute_dtype=mode)
    hp.stats = {f'spk {i}': {'pitch': {'mean': float(fx['stats_pitch_mean'][i]), 'std': float(fx['stats_pitch_std'][i])}}
                for i in range(11)}
    m = _model(hp).eval()
    names = ['symbols', 'dur_factors', 'energy_factors', 'pitch_factors', 'input_lengths', 'energy_refs', 'pitch_refs',
             'mel_spec_refs', 'ref_lengths', 'speaker_ids']
    for transform in ('add', 'multiply'):
        inputs = tuple(torch.from_numpy(fx[f'{transform}_in_{n}']).to(DEV) for n in names)
        enc, dec, weights = m.inference(inputs, transform, hp)
        torch.cuda.synchronize()
        tol = TOL[mode]
        assert _rel(enc[0], fx[f'{transform}_out_duration']) <= tol
        if mode == 'fp32':
            # the integer path is bit-exact given matching float durations
            assert np.array_equal(enc[1].cpu().numpy(), fx[f'{transform}_out_durations_int'])
            assert np.array_equal(dec[1].cpu().numpy(), fx[f'{transform}_out_output_lengths'])
            assert _rel(enc[2], fx[f'{transform}_out_energy']) <= tol
            assert _rel(enc[3], fx[f'{transform}_out_pitch']) <= 5 * tol
            assert _rel(dec[0], fx[f'{transform}_out_mel']) <= tol
            assert _rel(weights, fx[f'{transform}_out_weights']) <= tol
        else:
            # bf16 operand rounding may move a duration across a frame boundary: allow +-1 frame on a few symbols
            d = np.abs(enc[1].cpu().numpy() - fx[f'{transform}_out_durations_int'])
            assert d.max() <= 1
            assert np.abs(dec[1].cpu().numpy() - fx[f'{transform}_out_output_lengths']).max() <= 2
    with pytest.raises(NotImplementedError):
        m.inference(inputs, 'nope', hp)


def test_int_durations_kats_bit_exact(golden_dir):
    ''' K16 on the device vs the 3000 reference KATs of duration_to_integer (incl. the IndexError cases) '''
    from daft_exprt import ops
    fx = np.load(os.path.join(golden_dir, 'duration_to_integer.npz'))
    durs, d_off, ints, i_off = fx['durs'], fx['durs_off'], fx['ints'], fx['ints_off']
    n = len(d_off) - 1
    L = int(np.diff(d_off).max())
    pred = np.zeros((n, L), dtype=np.float32)
    for c in range(n):
        pred[c, : d_off[c + 1] - d_off[c]] = durs[d_off[c]: d_off[c + 1]]
    hp = make_hparams()
    p = torch.from_numpy(pred).to(DEV)
    dint, totals, status = ops.int_durations(p, hp)
    dint, status = dint.cpu().numpy(), status.cpu().numpy()
    n_err = 0
    for c in range(n):
        want = ints[i_off[c]: i_off[c + 1]]
        if want[0] == -1:
            assert status[c] == 1, c
            n_err += 1
        else:
            assert status[c] == 0, (c, status[c])
            assert np.array_equal(dint[c, : len(want)], want), (c, dint[c, : len(want)], want)
            assert not dint[c, len(want):].any()
    assert n_err == 19
    fx2 = np.load(os.path.join(golden_dir, 'get_int_durations.npz'))
    p = torch.from_numpy(fx2['preds'].copy()).to(DEV)
    dint, _, status = ops.int_durations(p, hp)
    assert np.array_equal(p.cpu().numpy(), fx2['thresholded'])
    assert np.array_equal(dint.cpu().numpy(), fx2['ints'])


GRAD_TOL = {'fp32': 3e-3, 'bf16': 6e-2}


@pytest.mark.parametrize('mode', ['fp32', 'bf16'])
def test_gradients_match_reference_golden(golden_dir, mode):
    ''' hand-written backward pass vs autograd of the reference (train mode, dropout 0, iteration 20000) '''
    from daft_exprt.loss import DaftExprtLoss
    fx = np.load(os.path.join(golden_dir, 'train_nodrop.npz'))
    hp = no_dropout(make_hparams(compute_dtype=mode))
    m = _model(hp).train()
    inputs = load_inputs(fx, DEV)
    targets = (inputs[1], inputs[3], inputs[4], inputs[8], inputs[10])
    crit = DaftExprtLoss(0, hp)
    m.zero_grad()
    total, _ = crit(m(inputs), targets, 20000)
    total.backward()
    torch.cuda.synchronize()
    assert abs(float(total) - float(fx['loss_total'])) <= (1e-4 if mode == 'fp32' else 2e-2) * float(fx['loss_total'])
    names = list(fx['param_names'])
    grads = {n: p.grad.detach().cpu().numpy() for n, p in m.named_parameters()}
    assert list(grads.keys()) == names
    tol = GRAD_TOL[mode]
    norms = np.array([np.linalg.norm(grads[n]) for n in names])
    rel = np.abs(norms - fx['grad_norms']) / (fx['grad_norms'] + 1e-8 * fx['grad_norms'].max())
    # the four sigma-path parameters of the upsampler get gradient only through the alignments, the tensor bf16 moves most
    # (SURVEY App. B item 9): twice the norm tolerance, like the element-wise check below
    sigma = ('gaussian_upsampling.projection.0.linear_layer', 'gaussian_upsampling.duration_projection.conv')
    rel = rel / np.array([2. if str(n).startswith(sigma) else 1. for n in names])
    worst = sorted(zip(rel, names), reverse=True)[:5]
    print(mode, 'worst grad-norm errors (sigma-path halved)', worst)
    assert rel.max() <= tol, worst
    for key in fx.files:
        if key.startswith('grad_full__'):
            n = key[len('grad_full__'):]
            ref = fx[key]
            err = np.abs(grads[n] - ref).max() / (np.abs(ref).max() + 1e-12)
            # gradients that flow through the Gaussian ranges (sigma) amplify bf16 operand rounding of the encoder
            # output: the reference itself moves by ~10 % there under bf16 autocast
            # bf16 operands: element-wise noise of the deepest gradients (prenet conv 0, seven GEMM layers of rounding
            # below the loss) reaches ~6 % of the tensor max while their norms stay within 3 %: element-wise tolerance 10 %
            t = 0.2 if (mode == 'bf16' and n.startswith('gaussian_upsampling.')) else (0.1 if mode == 'bf16' else tol)
            print(mode, 'full-grad err', n, float(err))
            assert err <= t, (n, err)
    heads = np.stack([np.pad(grads[n].reshape(-1)[:32], (0, max(0, 32 - grads[n].size))) for n in names])
    # element-wise check on the first 32 entries of every gradient, relative to the larger of their own max and
    # 3x the RMS of the whole tensor (so that a locally tiny slice does not turn rounding noise into a relative error)
    rms = (fx['grad_norms'] / np.sqrt(np.array([grads[n].size for n in names])))[:, None]
    scale = np.maximum(np.abs(fx['grad_heads']).max(axis=1, keepdims=True), 3 * rms) + 1e-12
    herr = (np.abs(heads - fx['grad_heads']) / scale).max(axis=1)
    gu = np.array([n.startswith('gaussian_upsampling.') for n in names])
    print(mode, 'worst head errors', sorted(zip(herr, names), reverse=True)[:5])
    assert herr[~gu].max() <= (0.15 if mode == 'bf16' else 2 * tol), sorted(zip(herr, names), reverse=True)[:5]
    assert herr[gu].max() <= (0.25 if mode == 'bf16' else 2 * tol)
    gn = float(np.sqrt(sum((g.astype(np.float64) ** 2).sum() for g in grads.values())))
    assert abs(gn - float(fx['grad_total_norm'])) <= tol * float(fx['grad_total_norm'])


def test_adam_steps_match_reference_golden(golden_dir):
    ''' 3 optimizer steps (fused Adam + LR schedule) vs torch.optim.Adam on the reference, fp32 operand mode '''
    from daft_exprt.loss import DaftExprtLoss
    from daft_exprt.optim import FusedAdam
    from daft_exprt.train import update_learning_rate
    fx = np.load(os.path.join(golden_dir, 'train_nodrop.npz'))
    hp = no_dropout(make_hparams(compute_dtype='fp32'))
    m = _model(hp).train()
    inputs = load_inputs(fx, DEV)
    targets = (inputs[1], inputs[3], inputs[4], inputs[8], inputs[10])
    crit = DaftExprtLoss(0, hp)
    opt = FusedAdam(m, betas=hp.betas, eps=hp.epsilon, weight_decay=hp.weight_decay, grad_clip_thresh=hp.grad_clip_thresh)
    before = m.flat_parameters().clone()
    losses, gnorms = [], []
    m.zero_grad()
    for step in range(3):
        opt.param_groups[0]['lr'] = update_learning_rate(hp, 1 + step)
        total, _ = crit(m(inputs), targets, 1 + step)
        total.backward()
        gnorms.append(float(opt.step().sqrt()))
        m.zero_grad()
        losses.append(float(total))
    print('adam losses', losses, fx['adam_losses'], 'gnorms', gnorms, fx['adam_grad_norms'])
    assert np.abs(np.array(losses) - fx['adam_losses']).max() <= 1e-2 * np.abs(fx['adam_losses']).max()
    assert np.abs(np.array(gnorms) - fx['adam_grad_norms']).max() <= 3e-2 * np.abs(fx['adam_grad_norms']).max()
    delta = (m.flat_parameters() - before).cpu().numpy()
    off = 0
    dn = []
    for n, p in m.named_parameters():
        dn.append(np.linalg.norm(delta[off: off + p.numel()]))
        off += p.numel()
    dn = np.array(dn)
    assert np.abs(dn - fx['adam_delta_norms']).max() <= 3e-2 * fx['adam_delta_norms'].max()


def test_weight_copies_follow_the_parameters(golden_dir):
    ''' the bf16 MFMA-operand copies of the weights are refreshed whenever the parameters changed through torch (in-place op,
        load_state_dict, a torch optimizer): the model watches the version counter of its flat parameter buffer '''
    fx = np.load(os.path.join(golden_dir, 'forward_eval.npz'))
    hp = make_hparams(compute_dtype='bf16')
    m = _model(hp).eval()
    assert m.always_repack          # the safe default; the version tracking below is what Trainer runs on
    m.always_repack = False
    inputs = load_inputs(fx, DEV)
    with torch.no_grad():
        mel0 = m(inputs)[3][0].clone()
        mel0b = m(inputs)[3][0].clone()
        assert torch.equal(mel0, mel0b)
        w = dict(m.named_parameters())['frame_decoder.projection.linear_layer.weight']
        w.mul_(2.)                                        # in-place under no_grad
        mel1 = m(inputs)[3][0].clone()
        assert float((mel1 - mel0).abs().max()) > 1e-3
        sd = {k: v.clone() for k, v in m.state_dict().items()}
        sd['frame_decoder.projection.linear_layer.weight'] *= 0.5
        m.load_state_dict(sd)                             # back to the original weights
        mel2 = m(inputs)[3][0].clone()
    assert torch.equal(mel2, mel0)
