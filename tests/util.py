"""Helpers shared by the tests (hparams factory, fixture loading, synthetic batches)."""
import numpy as np
import torch

from daft_exprt.hparams import HyperParams

INPUT_NAMES = ['symbols', 'durations_float', 'durations_int', 'symbols_energy', 'symbols_pitch', 'input_lengths',
               'frames_energy', 'frames_pitch', 'mel_specs', 'output_lengths', 'speaker_ids']
SPEAKERS = [f'spk{i:02d}' for i in range(11)]


def make_hparams(**extra):
    kw = dict(training_files='none', validation_files='none', output_directory='/nonexistent_daft_exprt_out',
              language='english', speakers=list(SPEAKERS))
    kw.update(extra)
    return HyperParams(verbose=False, **kw)


def no_dropout(hp):
    for cfg in (hp.prosody_encoder, hp.phoneme_encoder, hp.frame_decoder):
        cfg['attn_dropout'] = 0.
        cfg['conv_dropout'] = 0.
    hp.local_prosody_predictor['conv_dropout'] = 0.
    return hp


def load_inputs(fx, device=None):
    out = tuple(torch.from_numpy(np.asarray(fx[f'in_{n}'])) for n in INPUT_NAMES)
    return tuple(t.to(device) for t in out) if device is not None else out


def load_driver_fixture(golden_dir, transform, ref_dir=None):
    ''' RAW inputs of the reference's inference driver from tests/golden/inference_collate.npz (tools/gen_goldens.py, C2):
        (sentences, dur_factors, energy_factors, pitch_factors, refs, speaker_ids, file_names, fx).  refs are
        (energy, pitch, mel_spec) triples, or -- with ref_dir -- `.npz` paths written there under the reference's basenames. '''
    import json
    import os
    fx = np.load(os.path.join(golden_dir, 'inference_collate.npz'))
    sentences = json.loads(str(fx['sentences_json']))
    factors = json.loads(str(fx[f'{transform}_factors_json']))
    names = json.loads(str(fx['file_names_json']))
    base = json.loads(str(fx['ref_basenames_json']))
    refs = []
    for i in range(int(fx['n_sentences'])):
        triple = (fx[f'ref{i}_energy'], fx[f'ref{i}_pitch'], fx[f'ref{i}_mel_spec'])
        if ref_dir is not None:
            path = os.path.join(ref_dir, base[i])
            np.savez(path, energy=triple[0], pitch=triple[1], mel_spec=triple[2])
            refs.append(path)
        else:
            refs.append(triple)
    return sentences, factors['dur'], factors['energy'], factors['pitch'], refs, fx['speaker_ids'].tolist(), names, fx


COLLATE_NAMES = ['symbols', 'dur_factors', 'energy_factors', 'pitch_factors', 'input_lengths', 'energy_refs', 'pitch_refs',
                 'mel_spec_refs', 'ref_lengths', 'speaker_ids']


# parameters whose gradient passes DIRECTLY through a ReLU' gate: a pre-activation within rounding distance of 0 takes the other
# branch in one of the two computations, which moves the whole dW row (and the db element) of that channel by one position's
# contribution -- sparse outliers of ~1e-3 in otherwise 1e-5-exact tensors (measured at B = 8, T = 500, fp32 operand mode)
RELU_GATED = ('feed_forward.convs.0.conv.weight', 'feed_forward.convs.0.conv.bias', 'prosody_encoder.convs.0.conv.weight',
              'prosody_encoder.convs.0.conv.bias', 'prosody_encoder.convs.4.conv.weight', 'prosody_encoder.convs.4.conv.bias',
              'prosody_encoder.convs.8.conv.weight', 'prosody_encoder.convs.8.conv.bias', 'prosody_predictor.blocks.0.0.conv.weight',
              'prosody_predictor.blocks.0.0.conv.bias', 'prosody_predictor.blocks.0.4.conv.weight', 'prosody_predictor.blocks.0.4.conv.bias')
# tensors whose gradient flows through sigma = softplus(Linear(x + E + P + Dur)) of the upsampler (model.py:618-640): the range
# parameters' gradient sums exp(-d^2 / 2 sigma^2)-weighted terms over every (phoneme, frame) pair and amplifies operand rounding
SIGMA_PATH = ('gaussian_upsampling.projection.0.linear_layer.weight', 'gaussian_upsampling.projection.0.linear_layer.bias',
              'gaussian_upsampling.duration_projection.conv.weight', 'gaussian_upsampling.duration_projection.conv.bias',
              'gaussian_upsampling.energy_projection.conv.weight', 'gaussian_upsampling.energy_projection.conv.bias',
              'gaussian_upsampling.pitch_projection.conv.weight', 'gaussian_upsampling.pitch_projection.conv.bias')


def gradient_report(got, ref, rel, floor, sigma_factor=2.):
    ''' element-wise comparison of two {name: gradient tensor} dicts.  Bound per element: rel * max|ref tensor| + floor * (largest
        gradient element of the model); sigma-path tensors get sigma_factor * rel; ReLU-gated tensors may have 0.5 % of their
        elements outside the bound but must agree to 5 * rel in norm.  Returns [(score, name, a, b)] sorted worst first;
        score <= 1 passes. '''
    gmax = max(float(g.abs().max()) for g in ref.values())
    worst = []
    for name, r in ref.items():
        g = got[name].detach().float().cpu()
        r = r.detach().float()
        assert g.shape == r.shape, name
        tol = rel * (sigma_factor if name in SIGMA_PATH else 1.)
        bound = tol * float(r.abs().max()) + floor * gmax
        err = (g - r).abs()
        if name.endswith(RELU_GATED):
            bad = float((err > bound).float().mean())
            nrm = float((g - r).norm() / (r.norm() + 1e-30))
            worst.append((max(bad / 5e-3, nrm / (5. * tol)), name + ' [relu-gated: outlier share, norm]', bad, nrm))
        else:
            worst.append((float(err.max()) / bound, name, float(err.max()), float(r.abs().max())))
    worst.sort(reverse=True)
    return worst


def fill_end(lengths, N):
    ''' csrc/dx_common.h `dx_fill_end`: dead rows (past length + conv halo) are written as zeros only below this row index;
        rows at or past it are never read by any kernel and stay unwritten '''
    lengths = torch.as_tensor(lengths).long().clamp(min=0)
    return torch.clamp(((lengths + 4 + 255) // 256) * 256 + 1, max=N)


def drop_unwritten(t, lengths):
    ''' a copy of the (B, N, ...) kernel output `t` with the rows past `fill_end` set to zero (their content is unspecified):
        what is left is the contract -- live rows computed, dead rows below the fill end exactly zero '''
    t = t.clone()
    N = t.shape[1]
    keep = torch.arange(N, device=t.device)[None, :] < fill_end(lengths, N).to(t.device)[:, None]
    t[~keep] = 0
    return t


def install_unwritten_shim(monkeypatch):
    ''' op-level tests compare whole (B, N, C) outputs: wrap the `ops` entry points that take `lengths` / `skip_lengths` so that the
        rows past `fill_end` (unwritten by contract, content unspecified) come back as zeros.  Everything the tests then assert
        about padding rows -- exact zeros -- is the contract for the dead rows BELOW the fill end; what lies past it is dropped. '''
    from daft_exprt import ops

    def clean(t, lengths, N=None):
        if lengths is None or not torch.is_tensor(t):
            return t
        B = lengths.shape[0]
        if t.dim() == 3 and t.shape[0] == B and (N is None or t.shape[1] == N):
            return t.copy_(drop_unwritten(t, lengths))
        if t.dim() == 3:
            return t                                # (e.g. the (B, H, N) log-sum-exp of the attention forward: not row-major over N)
        if t.dim() == 1 and t.numel() % B == 0 and t.numel() > B:     # mean / rstd: (B * N,)
            return t.copy_(drop_unwritten(t.view(B, -1), lengths).view(-1))
        return t

    def wrap(fn, pick):
        def f(*a, **kw):
            out = fn(*a, **kw)
            lengths = pick(a, kw)
            if kw.get('out') is not None or kw.get('transposed_out'):
                return out
            N = a[0].shape[1] if (a and torch.is_tensor(a[0]) and a[0].dim() == 3) else None
            if isinstance(out, tuple):
                return tuple(clean(t, lengths, N) for t in out)
            return clean(out, lengths, N)
        return f
    monkeypatch.setattr(ops, 'conv1d', wrap(ops.conv1d, lambda a, kw: kw.get('skip_lengths')))
    monkeypatch.setattr(ops, 'conv1d_ln', wrap(ops.conv1d_ln, lambda a, kw: kw.get('lengths', a[6] if len(a) > 6 else None)))
    monkeypatch.setattr(ops, 'conv1d_lnbwd', wrap(ops.conv1d_lnbwd, lambda a, kw: kw.get('lengths', a[8] if len(a) > 8 else None)))
    monkeypatch.setattr(ops, 'layernorm_fwd', wrap(ops.layernorm_fwd, lambda a, kw: kw.get('skip_lengths')))
    monkeypatch.setattr(ops, 'layernorm_bwd', wrap(ops.layernorm_bwd, lambda a, kw: kw.get('skip_lengths')))
    monkeypatch.setattr(ops, 'attention_fwd', wrap(ops.attention_fwd, lambda a, kw: kw.get('lengths', a[1] if len(a) > 1 else None)))
    monkeypatch.setattr(ops, 'attention_bwd', wrap(ops.attention_bwd, lambda a, kw: kw.get('lengths', a[4] if len(a) > 4 else None)))
