"""Batched prosody-transfer synthesis driver.

Keeps the hot-path part of the reference driver (`src/daft_exprt/generate.py`): `collate_tensors` (140-239: symbol ids,
per-symbol duration / energy / pitch control factors, reference `.npz` prosody, sort by symbol count, padding rules),
`generate_batch_mel_specs` (242-317: one `model.inference` call per batch, crop per item, `.npz` with the same keys)
and `generate_mel_specs` (320-437: chunking + real-time-factor accounting).  Text phonemisation (MFA g2p), plots and
Griffin-Lim preview audio are outside the accelerated path: sentences arrive phonemised --
a list of words (lists of phone symbols) and boundary symbols, exactly what `prepare_sentences_for_inference` returns.
"""
import logging
import os
import time

import numpy as np
import torch

_logger = logging.getLogger(__name__)


def chunker(seq, size):
    ''' successive chunks of `size` items (`utils.py:92`) '''
    return [seq[pos: pos + size] for pos in range(0, len(seq), size)]


def _symbol_ids(sentence, hparams):
    ids = []
    for item in sentence:
        if isinstance(item, (list, tuple)):      # phones of one word
            ids.extend(hparams.symbols.index(phone) for phone in item)
        else:                                     # word boundary / punctuation / eos
            ids.append(hparams.symbols.index(item))
    return ids


def collate_tensors(batch_sentences, batch_dur_factors, batch_energy_factors, batch_pitch_factors, pitch_transform,
                    batch_refs, batch_speaker_ids, batch_file_names, hparams):
    ''' same contract as `generate.py:140-239`; `batch_refs` are `.npz` paths (keys energy, pitch, mel_spec) or
        already-loaded (energy, pitch, mel_spec) triples '''
    assert pitch_transform in ('add', 'multiply')
    neutral_pitch = 0. if pitch_transform == 'add' else 1.
    rows = []
    for sentence, dur_f, en_f, pi_f, ref in zip(batch_sentences, batch_dur_factors, batch_energy_factors, batch_pitch_factors, batch_refs):
        ids = _symbol_ids(sentence, hparams)
        n = len(ids)
        dur_f = [1.] * n if dur_f is None else list(dur_f)
        en_f = [1.] * n if en_f is None else list(en_f)
        pi_f = [neutral_pitch] * n if pi_f is None else list(pi_f)
        assert len(dur_f) == n, _logger.error(f'{len(dur_f)} duration factors whereas there a {n} symbols')
        assert len(en_f) == n, _logger.error(f'{len(en_f)} energy factors whereas there a {n} symbols')
        assert len(pi_f) == n, _logger.error(f'{len(pi_f)} pitch factors whereas there a {n} symbols')
        if isinstance(ref, (str, os.PathLike)):
            data = np.load(ref)
            ref = (data['energy'], data['pitch'], data['mel_spec'])
        energy, pitch, mel = (torch.as_tensor(np.asarray(a)).float() for a in ref)
        rows.append((torch.tensor(ids, dtype=torch.long), torch.tensor(dur_f), torch.tensor(en_f), torch.tensor(pi_f), energy, pitch, mel))
    n = len(rows)
    input_lengths, order = torch.sort(torch.LongTensor([len(r[0]) for r in rows]), dim=0, descending=True)
    L, T = int(input_lengths[0]), max(r[6].size(1) for r in rows)
    symbols = torch.zeros(n, L, dtype=torch.long)
    dur_factors, energy_factors = torch.ones(n, L), torch.ones(n, L)
    pitch_factors = torch.full((n, L), neutral_pitch)
    energy_refs, pitch_refs = torch.zeros(n, T), torch.zeros(n, T)
    mel_spec_refs = torch.zeros(n, hparams.n_mel_channels, T)
    ref_lengths, speaker_ids, file_names = torch.zeros(n, dtype=torch.long), torch.zeros(n, dtype=torch.long), []
    for row, src in enumerate(order.tolist()):
        ids, dur_f, en_f, pi_f, energy, pitch, mel = rows[src]
        l, t = len(ids), mel.size(1)
        symbols[row, :l], dur_factors[row, :l], energy_factors[row, :l], pitch_factors[row, :l] = ids, dur_f, en_f, pi_f
        energy_refs[row, :t], pitch_refs[row, :t], mel_spec_refs[row, :, :t] = energy, pitch, mel
        ref_lengths[row], speaker_ids[row] = t, batch_speaker_ids[src]
        file_names.append(batch_file_names[src])
    return symbols, dur_factors, energy_factors, pitch_factors, input_lengths, energy_refs, pitch_refs, mel_spec_refs, \
        ref_lengths, speaker_ids, file_names


def generate_batch_mel_specs(model, batch_sentences, batch_refs, batch_dur_factors, batch_energy_factors, batch_pitch_factors,
                             pitch_transform, batch_speaker_ids, batch_file_names, output_dir, hparams, n_jobs=1,
                             use_griffin_lim=False, get_time_perf=False):
    ''' `generate.py:242-317`: returns {file_name: prediction dict}; writes `<output_dir>/<file_name>.npz` when output_dir '''
    col = collate_tensors(batch_sentences, batch_dur_factors, batch_energy_factors, batch_pitch_factors, pitch_transform,
                          batch_refs, batch_speaker_ids, batch_file_names, hparams)
    file_names = col[-1]
    dev = model.flat_parameters().device
    inputs = tuple(t.to(dev, non_blocking=True) for t in col[:-1])
    if get_time_perf:
        torch.cuda.synchronize()
        start = time.time()
    inference = model.module.inference if hasattr(model, 'module') else model.inference
    encoder_preds, decoder_preds, alignments = inference(inputs, pitch_transform, hparams)
    if get_time_perf:
        torch.cuda.synchronize()
        elapsed = time.time() - start
    duration, duration_int, energy, pitch, input_lengths = (t.detach().cpu().numpy() for t in encoder_preds)
    mel_spec, output_lengths = (t.detach().cpu().numpy() for t in decoder_preds)
    weights = alignments.detach().cpu().numpy()
    predictions = {}
    for i, name in enumerate(file_names):
        l, t = int(input_lengths[i]), int(output_lengths[i])
        predictions[name] = {'duration': duration[i, :l], 'duration_int': duration_int[i, :l], 'energy': energy[i, :l],
                             'pitch': pitch[i, :l], 'mel_spec': mel_spec[i, :, :t], 'alignments': weights[i, :l, :t]}
        if output_dir:
            os.makedirs(output_dir, exist_ok=True)
            np.savez(os.path.join(output_dir, f'{name}.npz'), **predictions[name])
    if use_griffin_lim:
        _logger.warning('Griffin-Lim preview audio is outside the accelerated path; use a neural vocoder on the saved mel-specs')
    if get_time_perf:
        return predictions, elapsed
    return predictions


def generate_mel_specs(model, sentences, file_names, speaker_ids, refs, output_dir, hparams, dur_factors=None,
                       energy_factors=None, pitch_factors=None, batch_size=1, n_jobs=1, use_griffin_lim=False,
                       get_time_perf=False):
    ''' `generate.py:320-437`: eval mode, no grad, chunks of `batch_size`; real-time factor = audio seconds / wall seconds '''
    n = len(sentences)
    dur_factors = dur_factors or [None] * n
    energy_factors = energy_factors or [None] * n
    pitch_transform = 'add'
    if pitch_factors is None:
        pitch_factors = [None] * n
    elif isinstance(pitch_factors, (tuple, list)) and len(pitch_factors) == 2 and isinstance(pitch_factors[0], str):
        pitch_transform, pitch_factors = pitch_factors
    model.eval()
    predictions, total_time, audio_seconds = {}, 0., 0.
    with torch.no_grad():
        idx = list(range(n))
        for chunk in chunker(idx, batch_size):
            pick = lambda seq: [seq[i] for i in chunk]
            out = generate_batch_mel_specs(model, pick(sentences), pick(refs), pick(dur_factors), pick(energy_factors),
                                           pick(pitch_factors), pitch_transform, pick(speaker_ids), pick(file_names),
                                           output_dir, hparams, n_jobs, use_griffin_lim, get_time_perf)
            if get_time_perf:
                out, elapsed = out
                total_time += elapsed
                audio_seconds += sum(p['mel_spec'].shape[1] for p in out.values()) * hparams.hop_length / hparams.sampling_rate
            predictions.update(out)
    if get_time_perf:
        _logger.info(f'DaftExprt RTF: {audio_seconds / max(total_time, 1e-9):.2f}')
        return predictions, audio_seconds / max(total_time, 1e-9)
    return predictions
