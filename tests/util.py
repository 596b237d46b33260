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
