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


def _ref_name(ref, idx):
    ''' `_ref_<basename without .npz>` part of the output file name (`generate.py:250-251`); references handed over as
        already-loaded triples (an extension, the reference only takes paths) are named by their position '''
    if isinstance(ref, (str, os.PathLike)):
        return os.path.basename(str(ref)).replace('.npz', '')
    return f'mem{idx}'


def generate_batch_mel_specs(model, batch_sentences, batch_refs, batch_dur_factors, batch_energy_factors, batch_pitch_factors,
                             pitch_transform, batch_speaker_ids, batch_file_names, output_dir, hparams, n_jobs=1,
                             use_griffin_lim=True):
    ''' `generate.py:242-317`, same contract: every file name gets the `_spk_<id>_ref_<reference>` suffix IN PLACE
        (the caller's list is updated like the reference does, 248-253), one `model.inference` call on the collated
        batch, `<output_dir>/<file_name>.npz` holding `mel_spec` (298), and the return value
        `{file_name: [duration, duration_int, energy, pitch, mel_spec, alignment]}` cropped per item (300).
        Plots / Griffin-Lim preview audio (303-307) are outside the accelerated path: `use_griffin_lim` only warns. '''
    for idx, file_name in enumerate(batch_file_names):
        file_name += f'_spk_{batch_speaker_ids[idx]}'
        file_name += f'_ref_{_ref_name(batch_refs[idx], idx)}'
        batch_file_names[idx] = file_name
        _logger.info(f'Generating "{batch_sentences[idx]}" as "{file_name}"')
    col = collate_tensors(batch_sentences, batch_dur_factors, batch_energy_factors, batch_pitch_factors, pitch_transform,
                          batch_refs, batch_speaker_ids, batch_file_names, hparams)
    file_names = col[-1]
    gpu = next(model.parameters()).device
    core = model if hasattr(model, 'check_ids') else getattr(model, 'module', None)
    if core is not None and hasattr(core, 'check_ids'):
        core.check_ids(col[0], col[9], training=False)       # the reference's nn.Embedding would raise IndexError here
    if pitch_transform == 'add':
        for spk in col[9].tolist():
            hparams.stats[f'spk {spk}']['pitch']             # KeyError like `model.py:824-825` when a speaker has no statistics
    inputs = tuple(t.to(gpu, non_blocking=True) for t in col[:-1])
    inference = model.inference if hasattr(model, 'inference') else model.module.inference   # DDP-wrapped callers (270-278)
    encoder_preds, decoder_preds, alignments = inference(inputs, pitch_transform, hparams)
    duration, duration_int, energy, pitch, input_lengths = (t.detach().cpu().numpy() for t in encoder_preds)
    mel_spec, output_lengths = (t.detach().cpu().numpy() for t in decoder_preds)
    weights = alignments.detach().cpu().numpy()
    predictions = {}
    for i in range(mel_spec.shape[0]):
        l, t = int(input_lengths[i]), int(output_lengths[i])
        name = file_names[i]
        np.savez(os.path.join(output_dir, f'{name}.npz'), mel_spec=mel_spec[i, :, :t])
        predictions[f'{name}'] = [duration[i, :l], duration_int[i, :l], energy[i, :l], pitch[i, :l], mel_spec[i, :, :t],
                                  weights[i, :l, :t]]
    if use_griffin_lim:
        _logger.warning('Griffin-Lim preview audio / plots are outside the accelerated path; use a vocoder on the saved mel-specs')
    return predictions


LAST_TIME_PERF = {}   # filled by generate_mel_specs(get_time_perf=True): what the reference only logs (generate.py:433-435)


def generate_mel_specs(model, sentences, file_names, speaker_ids, refs, output_dir, hparams, dur_factors=None,
                       energy_factors=None, pitch_factors=None, batch_size=1, n_jobs=1, use_griffin_lim=False,
                       get_time_perf=False):
    ''' `generate.py:320-437`, same contract: `pitch_factors = [transform, [per-sentence factor lists]]`, the list-length
        asserts, eval mode + no grad, chunks of `batch_size`, returns the predictions dict only.  With `get_time_perf`
        the real-time factor is logged exactly like the reference: wall time of the whole per-batch function (collate,
        H2D, inference, D2H, file writes) against `((n_frames - 1) * hop + n_fft - 2 * (n_fft // 2)) / sr` seconds of
        audio per sentence (413-435); the numbers are also left in `LAST_TIME_PERF`. '''
    n = len(sentences)
    dur_factors = [None for _ in range(n)] if dur_factors is None else dur_factors
    energy_factors = [None for _ in range(n)] if energy_factors is None else energy_factors
    pitch_factors = ['add', [None for _ in range(n)]] if pitch_factors is None else pitch_factors
    pitch_transform = pitch_factors[0].lower()
    pitch_factors = pitch_factors[1]
    assert pitch_transform in ['add', 'multiply'], _logger.error(f'Pitch transform "{pitch_transform}" is not currently supported')
    for what, seq in (('filenames', file_names), ('speaker IDs', speaker_ids), ('references', refs),
                      ('duration factors', dur_factors), ('energy factors', energy_factors), ('pitch factors', pitch_factors)):
        assert len(seq) == n, _logger.error(f'{len(seq)} {what} but there are {n} sentences to generate')
    model.eval()
    os.makedirs(output_dir, exist_ok=True)
    predictions, time_per_batch = {}, []
    with torch.no_grad():
        for chunk in zip(chunker(sentences, batch_size), chunker(refs, batch_size), chunker(dur_factors, batch_size),
                         chunker(energy_factors, batch_size), chunker(pitch_factors, batch_size),
                         chunker(speaker_ids, batch_size), chunker(file_names, batch_size)):
            b_sent, b_refs, b_dur, b_en, b_pi, b_spk, b_names = chunk
            begin = time.time() if get_time_perf else None
            predictions.update(generate_batch_mel_specs(model, b_sent, b_refs, b_dur, b_en, b_pi, pitch_transform, b_spk,
                                                        b_names, output_dir, hparams, n_jobs, use_griffin_lim))
            time_per_batch += [time.time() - begin] if get_time_perf else []
    if get_time_perf:
        durations = []
        for prediction in predictions.values():
            nb_frames = prediction[4].shape[1]
            nb_wav_samples = (nb_frames - 1) * hparams.hop_length + hparams.filter_length
            if hparams.centered:
                nb_wav_samples -= 2 * int(hparams.filter_length / 2)
            durations.append(nb_wav_samples / hparams.sampling_rate)
        LAST_TIME_PERF.clear()
        LAST_TIME_PERF.update({'sentences': len(predictions), 'audio_seconds': sum(durations), 'wall_seconds': sum(time_per_batch),
                               'rtf': sum(durations) / sum(time_per_batch)})
        _logger.info('')
        _logger.info(f'{len(predictions)} sentences ({sum(durations):.2f}s) generated in {sum(time_per_batch):.2f}s')
        _logger.info(f'DaftExprt RTF: {sum(durations) / sum(time_per_batch):.2f}')
    return predictions
