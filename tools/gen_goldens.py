#!/usr/bin/env python
"""Generate the golden fixtures under tests/golden/ by IMPORTING the reference
(`/root/reference/src/daft_exprt`) in the build container.  The reference never travels to
the GPU box: only the data written here (inputs + expected outputs) does.

Shims, all outside the reference tree:
  1. stub modules for absent third-party imports (librosa, unidecode, inflect, tgt, tensorboard);
  2. `Tensor.cuda` / `Module.cuda` -> identity (the reference calls `.cuda(device)` everywhere);
  3. `HyperParams.update_mfa_paths` -> no-op (it asserts on MFA model files).

Weights come from `oracle/fill.py` (closed form), so fixtures carry no weight blobs.

Run:  python tools/gen_goldens.py          (writes tests/golden/*.npz, ~1 MB total)
"""
import os
import sys
import tempfile
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_SRC = '/root/reference/src'
OUT = os.path.join(ROOT, 'tests', 'golden')
sys.path.insert(0, ROOT)


def install_shims():
    def stub(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m
    librosa = stub('librosa')
    librosa.filters = stub('librosa.filters', mel=lambda *a, **k: None)
    stub('unidecode', unidecode=lambda s: s)
    stub('inflect', engine=lambda: None)
    stub('tgt')
    import importlib.util
    if importlib.util.find_spec('tensorboard') is None:
        stub('torch.utils.tensorboard', SummaryWriter=object)
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self
    sys.path.insert(0, REF_SRC)
    import daft_exprt.hparams as ref_hparams
    ref_hparams.HyperParams.update_mfa_paths = lambda self: None


SPEAKERS = [f'spk{i:02d}' for i in range(11)]


def make_hparams(ref_hparams, no_dropout=False, **extra):
    kw = dict(training_files='none', validation_files='none', output_directory='/nonexistent_daft_exprt_out',
              language='english', speakers=list(SPEAKERS))
    hp = ref_hparams.HyperParams(verbose=False, **kw, **extra)
    if no_dropout:
        for cfg in (hp.prosody_encoder, hp.phoneme_encoder, hp.frame_decoder):
            cfg['attn_dropout'] = 0.
            cfg['conv_dropout'] = 0.
        hp.local_prosody_predictor['conv_dropout'] = 0.
    return hp


def synth_items(rng, lens_L, lens_T, n_mel=80, n_speakers=11):
    ''' per-utterance training items in the order DaftExprtDataLoader yields them (data_loader.py:120-137) '''
    items = []
    for L, T in zip(lens_L, lens_T):
        d = np.zeros(L, dtype=np.int64)
        # spread T frames over L symbols, leave some zero-duration symbols
        cut = np.sort(rng.randint(0, T + 1, size=L - 1))
        d[:] = np.diff(np.concatenate(([0], cut, [T])))
        assert d.sum() == T
        dur_f = (d * 256. / 22050.).astype(np.float32)
        sym = rng.randint(1, 76, size=L).astype(np.int64)
        s_en = rng.randn(L).astype(np.float32) * (d > 0)
        s_pi = rng.randn(L).astype(np.float32) * (d > 0)
        f_en = rng.uniform(0, 3, size=T).astype(np.float32)
        f_pi = (rng.randn(T) * (rng.rand(T) > 0.3)).astype(np.float32)
        mel = np.clip(rng.randn(n_mel, T) * 1.2 - 1.0, np.log(1e-5), 2.).astype(np.float32)
        spk = int(rng.randint(0, n_speakers))
        items.append([torch.from_numpy(sym), torch.from_numpy(dur_f), torch.from_numpy(d), torch.from_numpy(s_en),
                      torch.from_numpy(s_pi), torch.from_numpy(f_en), torch.from_numpy(f_pi), torch.from_numpy(mel),
                      spk, f'dir{len(items)}', f'file{len(items)}'])
    return items


INPUT_NAMES = ['symbols', 'durations_float', 'durations_int', 'symbols_energy', 'symbols_pitch', 'input_lengths',
               'frames_energy', 'frames_pitch', 'mel_specs', 'output_lengths', 'speaker_ids']


def load_fill(model, hp):
    from oracle import daft_exprt_cpu as O
    from oracle.fill import fill_params
    shapes = O.param_shapes(hp)
    sd = model.state_dict()
    assert list(sd.keys()) == list(shapes.keys()), 'oracle.param_shapes order/name mismatch with reference state_dict'
    for k, v in sd.items():
        assert tuple(v.shape) == tuple(shapes[k]), (k, v.shape, shapes[k])
    model.load_state_dict(fill_params(shapes))


def np_(t):
    return t.detach().cpu().numpy() if torch.is_tensor(t) else np.asarray(t)


def main():
    install_shims()
    import daft_exprt.hparams as ref_hparams
    from daft_exprt.model import DaftExprt
    from daft_exprt.loss import DaftExprtLoss
    from daft_exprt.data_loader import DaftExprtDataCollate
    from daft_exprt.extract_features import duration_to_integer
    from daft_exprt.train import update_learning_rate
    from daft_exprt.generate import collate_tensors
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)
    torch.set_num_threads(8)

    # ------------------------------------------------------------------ A. eval forward + collate KAT
    rng = np.random.RandomState(1234)
    hp = make_hparams(ref_hparams)
    model = DaftExprt(hp)
    load_fill(model, hp)
    model.eval()
    # pad extents 0 / 1 / >=2 on both axes; items arrive unsorted
    lens_L, lens_T = [9, 12, 5, 11], [39, 37, 20, 40]
    items = synth_items(rng, lens_L, lens_T)
    batch = DaftExprtDataCollate(hp)(items)
    inputs, targets, _ = model.parse_batch(0, batch)
    with torch.no_grad():
        out = model(inputs)
    spk, film, enc_preds, dec_preds, weights = out
    fx = {f'in_{n}': np_(t) for n, t in zip(INPUT_NAMES, inputs)}
    # raw items so that the collate contract can be replayed
    for i, it in enumerate(items):
        for j, nm in enumerate(['symbols', 'dur_float', 'dur_int', 'sym_energy', 'sym_pitch', 'frames_energy',
                                'frames_pitch', 'mel']):
            fx[f'item{i}_{nm}'] = np_(it[j])
        fx[f'item{i}_speaker'] = np.int64(it[8])
    fx['collate_dirs'] = np.array(batch[11])
    fx['collate_files'] = np.array(batch[12])
    fx.update(out_speaker_preds=np_(spk), out_post_multipliers=np_(film[0]), out_encoder_film=np_(film[1]),
              out_prosody_pred_film=np_(film[2]), out_decoder_film=np_(film[3]), out_duration=np_(enc_preds[0]),
              out_energy=np_(enc_preds[1]), out_pitch=np_(enc_preds[2]), out_mel=np_(dec_preds[0].contiguous()),
              out_weights=np_(weights))
    # per-module intermediates (eval mode) for finer-grained kernel tests
    with torch.no_grad():
        symbols, dur_f, dur_i, s_en, s_pi, in_len, f_en, f_pi, mel, out_len, spk_ids = inputs
        emb, ef, pf, df = model.prosody_encoder(f_en, f_pi, mel, spk_ids, out_len)
        enc = model.phoneme_encoder(symbols, ef, in_len)
        x_up, _ = model.gaussian_upsampling(enc, dur_f, dur_i, s_en, s_pi, in_len)
    fx.update(mid_prosody_embeddings=np_(emb), mid_enc_outputs=np_(enc), mid_symbols_upsamp=np_(x_up))
    # eval-mode loss terms at several iterations
    crit = DaftExprtLoss(0, hp)
    for it_ in (0, 1, 5000, 10000, 20000):
        total, indiv = crit(out, targets, it_)
        fx[f'loss_total_it{it_}'] = np.float64(total.item())
        fx[f'loss_terms_it{it_}'] = np.array([indiv[k] for k in ('speaker_loss', 'post_mult_loss', 'duration_loss',
                                              'energy_loss', 'pitch_loss', 'mel_spec_l1_loss', 'mel_spec_l2_loss')])
    np.savez_compressed(os.path.join(OUT, 'forward_eval.npz'), **fx)
    print('forward_eval: mel', fx['out_mel'].shape, 'loss@20000', fx['loss_total_it20000'])

    # ------------------------------------------------------------------ B. train mode, dropout 0: grads + 3 Adam steps
    rng = np.random.RandomState(4321)
    hp0 = make_hparams(ref_hparams, no_dropout=True)
    model = DaftExprt(hp0)
    load_fill(model, hp0)
    model.train()
    items = synth_items(rng, [7, 10, 10], [33, 30, 31])
    batch = DaftExprtDataCollate(hp0)(items)
    inputs, targets, _ = model.parse_batch(0, batch)
    crit = DaftExprtLoss(0, hp0)
    fx = {f'in_{n}': np_(t) for n, t in zip(INPUT_NAMES, inputs)}
    iteration = 20000
    out = model(inputs)
    total, indiv = crit(out, targets, iteration)
    model.zero_grad()
    total.backward()
    names = [n for n, _ in model.named_parameters()]
    fx['param_names'] = np.array(names)
    fx['loss_total'] = np.float64(total.item())
    fx['grad_norms'] = np.array([p.grad.norm().item() for _, p in model.named_parameters()])
    fx['grad_heads'] = np.stack([np.pad(np_(p.grad).reshape(-1)[:32], (0, max(0, 32 - p.grad.numel())))
                                 for _, p in model.named_parameters()])
    fx['grad_total_norm'] = np.float64(torch.nn.utils.clip_grad_norm_(model.parameters(), float('inf')).item())
    for nm in ('prosody_encoder.post_multipliers', 'prosody_predictor.projection.linear_layer.weight',
               'gaussian_upsampling.projection.0.linear_layer.weight', 'gaussian_upsampling.duration_projection.conv.weight',
               'speaker_classifier.classifier.5.linear_layer.weight', 'frame_decoder.projection.linear_layer.bias',
               'prosody_encoder.spk_embedding.weight', 'phoneme_encoder.blocks.0.attention.multi_head_attention.in_proj_bias',
               'prosody_encoder.blocks.3.attention.multi_head_attention.out_proj.weight',
               'prosody_encoder.convs.0.conv.bias', 'frame_decoder.blocks.3.feed_forward.layer_norm.weight'):
        fx['grad_full__' + nm] = np_(dict(model.named_parameters())[nm].grad)
    # 3 optimizer steps exactly as train.py:299-301,391-401,486-494 (accumulation_steps=1 here)
    opt = torch.optim.Adam(model.parameters(), betas=hp0.betas, eps=hp0.epsilon, weight_decay=hp0.weight_decay, amsgrad=False)
    before = {n: p.detach().clone() for n, p in model.named_parameters()}
    losses, gnorms = [], []
    model.zero_grad()
    iteration = 1  # first optimizer steps of a run: lr ~1e-4 (train.py:306,315)
    for step in range(3):
        lr = update_learning_rate(hp0, iteration + step)
        for g in opt.param_groups:
            g['lr'] = lr
        out = model(inputs)
        total, _ = crit(out, targets, iteration + step)
        total.backward()
        gnorms.append(torch.nn.utils.clip_grad_norm_(model.parameters(), hp0.grad_clip_thresh).item())
        opt.step()
        model.zero_grad()
        losses.append(total.item())
    fx['adam_losses'] = np.array(losses)
    fx['adam_grad_norms'] = np.array(gnorms)
    fx['adam_delta_norms'] = np.array([(p.detach() - before[n]).norm().item() for n, p in model.named_parameters()])
    fx['adam_delta_heads'] = np.stack([np.pad(np_(p.detach() - before[n]).reshape(-1)[:8], (0, max(0, 8 - p.numel())))
                                       for n, p in model.named_parameters()])
    np.savez_compressed(os.path.join(OUT, 'train_nodrop.npz'), **fx)
    print('train_nodrop: losses', losses, 'gnorm', gnorms)

    # ------------------------------------------------------------------ C. inference (add / multiply)
    rng = np.random.RandomState(777)
    hp = make_hparams(ref_hparams)
    hp.stats = {f'spk {i}': {'pitch': {'mean': 5.0 + 0.05 * i, 'std': 0.25 + 0.01 * i},
                             'energy': {'mean': 20. + i, 'std': 5.}} for i in range(11)}
    model = DaftExprt(hp)
    load_fill(model, hp)
    model.eval()
    tmp = tempfile.mkdtemp()
    sentences, refs, spk_ids, names = [], [], [], []
    for i, (L, Tref) in enumerate(zip([8, 14, 11], [45, 30, 52])):
        # sentence structure: list of words (lists of phones) and boundary symbols (generate.py:150-156)
        phones = [hp.symbols[int(s)] for s in rng.randint(7, 76, size=L - 2)]
        sentences.append([phones[: L // 2], ' ', phones[L // 2:], '.'])
        ref = os.path.join(tmp, f'ref{i}.npz')
        np.savez(ref, energy=rng.uniform(0, 3, size=Tref).astype(np.float32),
                 pitch=(rng.randn(Tref) * (rng.rand(Tref) > 0.3)).astype(np.float32),
                 mel_spec=np.clip(rng.randn(80, Tref) * 1.2 - 1.0, np.log(1e-5), 2.).astype(np.float32))
        refs.append(ref)
        spk_ids.append(int(rng.randint(0, 11)))
        names.append(f'utt{i}')
    fx = {}
    for transform in ('add', 'multiply'):
        dur_factors = [None, list(rng.uniform(0.8, 1.3, size=14)), None]
        energy_factors = [list(rng.uniform(0.5, 1.5, size=8)), None, None]
        if transform == 'add':
            pitch_factors = [None, None, list(rng.uniform(-30, 50, size=11))]
        else:
            pitch_factors = [list(rng.uniform(-1.5, 1.0, size=8)), None, None]
        col = collate_tensors(sentences, dur_factors, energy_factors, pitch_factors, transform, refs, spk_ids, names, hp)
        inputs = tuple(t for t in col[:-1])
        with torch.no_grad():
            enc_preds, dec_preds, weights = model.inference(tuple(t.clone() for t in inputs), transform, hp)
        for nm, t in zip(['symbols', 'dur_factors', 'energy_factors', 'pitch_factors', 'input_lengths', 'energy_refs',
                          'pitch_refs', 'mel_spec_refs', 'ref_lengths', 'speaker_ids'], inputs):
            fx[f'{transform}_in_{nm}'] = np_(t)
        fx[f'{transform}_file_names'] = np.array(col[-1])
        for nm, t in zip(['duration', 'durations_int', 'energy', 'pitch', 'input_lengths'], enc_preds):
            fx[f'{transform}_out_{nm}'] = np_(t)
        fx[f'{transform}_out_mel'] = np_(dec_preds[0].contiguous())
        fx[f'{transform}_out_output_lengths'] = np_(dec_preds[1])
        fx[f'{transform}_out_weights'] = np_(weights)
        print('inference', transform, 'durations_int', np_(enc_preds[1]).tolist(), 'T', np_(dec_preds[1]).tolist())
    fx['stats_pitch_mean'] = np.array([hp.stats[f'spk {i}']['pitch']['mean'] for i in range(11)])
    fx['stats_pitch_std'] = np.array([hp.stats[f'spk {i}']['pitch']['std'] for i in range(11)])
    np.savez_compressed(os.path.join(OUT, 'inference.npz'), **fx)

    # ------------------------------------------------------------------ C2. inference collate + driver contract (generate.py:140-437)
    # RAW driver inputs (nested sentences, per-symbol factor lists, reference .npz contents, speaker ids, file names) next to
    # what the reference's collate_tensors / generate_mel_specs make of them: pins the product's collate_tensors, the oracle's
    # collate_inference, the `_spk_<id>_ref_<name>` file naming, the list-of-6 predictions and the .npz contents.
    import json
    from daft_exprt.generate import generate_mel_specs
    rng = np.random.RandomState(4321)       # own stream: sections C and D above/below keep their draws
    c2 = {'n_sentences': np.array(5)}
    sentences, refs, spk_ids, names = [], [], [], []
    ref_dir = os.path.join(tmp, 'refs2')
    os.makedirs(ref_dir, exist_ok=True)
    for i, (L, Tref) in enumerate(zip([9, 17, 17, 5, 12], [40, 61, 33, 61, 20])):    # ties in L and in T_ref
        phones = [hp.symbols[int(s)] for s in rng.randint(7, 76, size=L - 3)]
        cut = max(1, (L - 3) // 3)
        sentences.append([phones[:cut], ' ', phones[cut:2 * cut], ',', phones[2 * cut:], '~'] if L - 3 - 2 * cut > 0
                         else [phones[:cut], ' ', phones[cut:], '.', '~'])
        ref = os.path.join(ref_dir, f'prosody_ref_{i}.npz')
        arrs = dict(energy=rng.uniform(0, 3, size=Tref).astype(np.float32),
                    pitch=(rng.randn(Tref) * (rng.rand(Tref) > 0.3)).astype(np.float32),
                    mel_spec=np.clip(rng.randn(80, Tref) * 1.2 - 1.0, np.log(1e-5), 2.).astype(np.float32))
        np.savez(ref, **arrs)
        for k, v in arrs.items():
            c2[f'ref{i}_{k}'] = v
        refs.append(ref)
        spk_ids.append(int(rng.randint(0, 11)))
        names.append(f'sent{i}')
    n_sym = []
    for sent in sentences:
        n_sym.append(sum(len(it) if isinstance(it, list) else 1 for it in sent))
    c2['sentences_json'] = np.array(json.dumps(sentences))
    c2['ref_basenames_json'] = np.array(json.dumps([os.path.basename(r) for r in refs]))
    c2['speaker_ids'] = np.array(spk_ids)
    c2['file_names_json'] = np.array(json.dumps(names))
    for transform in ('add', 'multiply'):
        dur_factors = [None if i % 2 == 0 else [float(np.float32(v)) for v in rng.uniform(0.8, 1.3, size=n)] for i, n in enumerate(n_sym)]
        energy_factors = [[float(np.float32(v)) for v in rng.uniform(0.5, 1.5, size=n)] if i in (0, 3) else None for i, n in enumerate(n_sym)]
        lo, hi = (-30, 50) if transform == 'add' else (-1.5, 1.0)
        pitch_factors = [[float(np.float32(v)) for v in rng.uniform(lo, hi, size=n)] if i in (1, 4) else None for i, n in enumerate(n_sym)]
        c2[f'{transform}_factors_json'] = np.array(json.dumps({'dur': dur_factors, 'energy': energy_factors, 'pitch': pitch_factors}))
        col = collate_tensors(sentences, dur_factors, energy_factors, pitch_factors, transform, refs, spk_ids, list(names), hp)
        for nm, t in zip(['symbols', 'dur_factors', 'energy_factors', 'pitch_factors', 'input_lengths', 'energy_refs',
                          'pitch_refs', 'mel_spec_refs', 'ref_lengths', 'speaker_ids'], col[:-1]):
            c2[f'{transform}_col_{nm}'] = np_(t)
        c2[f'{transform}_col_file_names_json'] = np.array(json.dumps(list(col[-1])))
        # the whole driver: 5 sentences in chunks of 2 (last chunk of 1), reference weights from the closed-form fill
        out_dir = os.path.join(tmp, f'gen_{transform}')
        drv_names = list(names)
        preds = generate_mel_specs(model, sentences, drv_names, spk_ids, refs, out_dir, hp, dur_factors=dur_factors,
                                   energy_factors=energy_factors, pitch_factors=[transform.upper(), pitch_factors], batch_size=2,
                                   n_jobs=1, use_griffin_lim=False, get_time_perf=True)
        c2[f'{transform}_drv_keys_json'] = np.array(json.dumps(list(preds.keys())))
        c2[f'{transform}_drv_names_after_json'] = np.array(json.dumps(drv_names))     # the caller's list after the call
        files = sorted(os.listdir(out_dir))
        c2[f'{transform}_drv_files_json'] = np.array(json.dumps(files))
        c2[f'{transform}_drv_npz_keys_json'] = np.array(json.dumps(sorted(np.load(os.path.join(out_dir, files[0])).files)))
        for k, (key, vals) in enumerate(preds.items()):
            for nm, v in zip(['duration', 'duration_int', 'energy', 'pitch', 'mel_spec', 'alignment'], vals):
                c2[f'{transform}_drv{k}_{nm}'] = np.asarray(v)
            assert np.array_equal(np.load(os.path.join(out_dir, f'{key}.npz'))['mel_spec'], vals[4])
        print('driver', transform, list(preds.keys()), 'names after:', drv_names)
    np.savez_compressed(os.path.join(OUT, 'inference_collate.npz'), **c2)

    # ------------------------------------------------------------------ D. duration_to_integer KATs
    rng = np.random.RandomState(99)
    flat_in, off_in, flat_out, off_out = [], [0], [], [0]
    n_err = 0
    for case in range(3000):
        n = int(rng.randint(1, 41))
        if case % 25 == 0:
            n = int(rng.randint(1, 3))  # short utterances -> IndexError region
        durs = rng.uniform(0.0233, 0.30, size=n).astype(np.float32)
        if case % 7 == 0:
            durs[rng.randint(0, n)] = np.float32(512.5 / 22050.)  # near the threshold
        spans, end_prev = [], 0.
        for d in durs.tolist():
            spans.append([end_prev, end_prev + d])
            end_prev += d
        try:
            res = duration_to_integer([list(s) for s in spans], hp)
        except IndexError:
            res = [-1]
            n_err += 1
        flat_in.extend(durs.tolist())
        off_in.append(len(flat_in))
        flat_out.extend(res)
        off_out.append(len(flat_out))
    np.savez_compressed(os.path.join(OUT, 'duration_to_integer.npz'), durs=np.array(flat_in, dtype=np.float32),
                        durs_off=np.array(off_in), ints=np.array(flat_out, dtype=np.int64), ints_off=np.array(off_out))
    print('duration_to_integer: 3000 cases,', n_err, 'IndexError cases')

    # get_int_durations on a padded batch (thresholding + scatter), model.py:789-812
    rng = np.random.RandomState(5)
    preds = rng.uniform(-0.02, 0.2, size=(6, 20)).astype(np.float32)
    for b, l in enumerate([20, 17, 12, 9, 6, 5]):
        preds[b, l:] = 0.
    p = torch.from_numpy(preds.copy())
    p2, ints = model.get_int_durations(p, hp)
    np.savez_compressed(os.path.join(OUT, 'get_int_durations.npz'), preds=preds, thresholded=np_(p2), ints=np_(ints))

    # ------------------------------------------------------------------ F. on-disk feature reader (data_loader.py:11-137)
    # a tiny pre-processed data set in the reference's file formats (written here, read by the reference loader)
    from daft_exprt.data_loader import DaftExprtDataLoader
    rng = np.random.RandomState(2024)
    feat_root = os.path.join(OUT, 'features')
    lines = []
    hp = make_hparams(ref_hparams)
    hp.stats = {f'spk {i}': {'energy': {'mean': 20. + i, 'std': 5. + 0.5 * i}, 'pitch': {'mean': 5.0 + 0.05 * i, 'std': 0.25 + 0.01 * i}}
                for i in range(11)}
    for u, (spk, L, T) in enumerate([(0, 6, 21), (3, 9, 30), (3, 4, 11), (7, 7, 25), (0, 5, 17)]):
        d = os.path.join(feat_root, f'spk{spk:02d}')
        os.makedirs(d, exist_ok=True)
        name = f'utt{u:03d}'
        cut = np.sort(rng.randint(0, T + 1, size=L - 1))
        dur = np.diff(np.concatenate(([0], cut, [T])))
        np.save(os.path.join(d, name + '.npy'), np.clip(rng.randn(80, T) * 1.2 - 1., np.log(1e-5), 2.).astype(np.float32))
        t = 0.
        with open(os.path.join(d, name + '.markers'), 'w') as f:
            for l in range(L):
                end = t + dur[l] * 256. / 22050. + (0.001 if dur[l] == 0 else 0.)
                sym = hp.symbols[int(rng.randint(1, 76))]
                f.write(f'{t:.6f}\t{end:.6f}\t{int(dur[l])}\t{sym}\tword{l}\t{l}\n')
                t = end
        for ext, n, zero_p in (('symbols_nrg', L, 0.2), ('frames_nrg', T, 0.1), ('symbols_f0', L, 0.3), ('frames_f0', T, 0.3)):
            vals = rng.uniform(10., 40., size=n) if 'nrg' in ext else rng.uniform(4.5, 5.6, size=n)
            vals[rng.rand(n) < zero_p] = 0.
            with open(os.path.join(d, name + '.' + ext), 'w') as f:
                f.write('\n'.join(f'{v:.5f}' for v in vals) + '\n')
        lines.append(f'{os.path.join("features", f"spk{spk:02d}")}|{name}|{spk}')
    list_file = os.path.join(OUT, 'train_list.txt')
    with open(list_file, 'w') as f:
        f.write('\n'.join(lines) + '\n')
    cwd = os.getcwd()
    os.chdir(OUT)   # list entries are relative to tests/golden
    ds = DaftExprtDataLoader(list_file, hp, shuffle=True)
    fx = {'n_items': np.int64(len(ds)), 'stats_energy_mean': np.array([hp.stats[f'spk {i}']['energy']['mean'] for i in range(11)]),
          'stats_energy_std': np.array([hp.stats[f'spk {i}']['energy']['std'] for i in range(11)]),
          'stats_pitch_mean': np.array([hp.stats[f'spk {i}']['pitch']['mean'] for i in range(11)]),
          'stats_pitch_std': np.array([hp.stats[f'spk {i}']['pitch']['std'] for i in range(11)])}
    for i in range(len(ds)):
        item = ds[i]
        for j, nm in enumerate(['symbols', 'dur_float', 'dur_int', 'sym_energy', 'sym_pitch', 'frames_energy', 'frames_pitch', 'mel']):
            fx[f'item{i}_{nm}'] = np_(item[j])
        fx[f'item{i}_speaker'] = np.int64(item[8])
        fx[f'item{i}_file'] = np.array(item[10])
    os.chdir(cwd)
    np.savez_compressed(os.path.join(OUT, 'data_loader.npz'), **fx)
    print('data_loader: order', [str(fx[f'item{i}_file']) for i in range(len(ds))])

    # ------------------------------------------------------------------ E. schedules
    its = np.array([0, 1, 2, 100, 4999, 5000, 9999, 10000, 10001, 20000, 40000, 123457, 370000])
    lr = np.array([update_learning_rate(hp, int(i)) if i > 0 else update_learning_rate(hp, 0) for i in its])
    adv = np.array([DaftExprtLoss(0, hp).update_adversarial_weight(int(i)) for i in its])
    np.savez_compressed(os.path.join(OUT, 'schedules.npz'), iterations=its, lr=lr, adv=adv)
    print('schedules ok; total fixture bytes:', sum(os.path.getsize(os.path.join(OUT, f)) for f in os.listdir(OUT)))


if __name__ == '__main__':
    main()
