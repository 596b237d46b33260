"""Hyper-parameter object of the MI355X-native Daft-Exprt build.

Keeps the surface of the reference `HyperParams` (`src/daft_exprt/hparams.py:19-244`):
`HyperParams(verbose=True, **kwargs)`, the same attribute names and default values
(`hparams.py:36-147`), the same derived attributes (`n_symbols`, `speakers_id`,
`n_speakers = #speakers + 1`, `hparams.py:189-203`), the same consistency checks
(`hparams.py:158-159, 192, 206-217`) and `save_hyper_params(json_file)`
(`hparams.py:232-244`), so `config.json` files and the `config_params` dict stored in
checkpoints (`train.py:73-78`) round-trip between the two code bases.

Differences, all additive:
  * `update_mfa_paths` only fills the three MFA path attributes; it does not assert that
    the Montreal-Forced-Aligner files exist (the aligner is outside the accelerated path).
  * `compute_dtype` ('bf16' | 'fp32'): arithmetic type of the MFMA operands in the HIP
    kernels.  Accumulation, LayerNorm statistics, softmax, Gaussian upsampling, losses and
    Adam are fp32 in both modes.
"""
import json
import logging
import os
import sys
from pathlib import Path

from daft_exprt.symbols import pad, symbols_english

_logger = logging.getLogger(__name__)


def _fft_block_cfg(heads, with_hidden=True):
    cfg = {'nb_blocks': 4}
    if with_hidden:
        cfg['hidden_embed_dim'] = 128
    cfg.update({'attn_nb_heads': heads, 'attn_dropout': 0.1, 'conv_kernel': 3,
                'conv_channels': 1024, 'conv_dropout': 0.1})
    return cfg


def _defaults():
    ''' (name, default) pairs; `None` marks values the caller has to provide. '''
    return [
        # misc
        ('minimum_wav_duration', 1000),
        # mel-spectrogram extraction
        ('centered', True), ('min_clipping', 1e-5), ('sampling_rate', 22050), ('mel_fmin', 0),
        ('mel_fmax', 8000), ('n_mel_channels', 80), ('filter_length', 1024), ('hop_length', 256),
        # REAPER pitch extraction
        ('f0_interval', 0.005), ('min_f0', 40), ('max_f0', 500), ('uv_interval', 0.01),
        ('uv_cost', 0.9), ('order', 1), ('cutoff', 25),
        # training
        ('seed', 1234), ('cudnn_enabled', True), ('cudnn_benchmark', False), ('cudnn_deterministic', True),
        ('dist_backend', 'nccl'),  # "nccl" is RCCL on ROCm
        ('nb_iterations', 370000), ('iters_per_checkpoint', 10000),
        ('iters_check_for_model_improvement', 5000), ('batch_size', 16), ('accumulation_steps', 3),
        ('checkpoint', ''),
        # loss weights
        ('lambda_reversal', 1.), ('adv_max_weight', 1e-2), ('post_mult_weight', 1e-3), ('dur_weight', 1.),
        ('energy_weight', 1.), ('pitch_weight', 1.), ('mel_spec_weight', 1.),
        # optimizer
        ('optimizer', 'adam'), ('betas', (0.9, 0.98)), ('epsilon', 1e-9), ('weight_decay', 1e-6),
        ('initial_learning_rate', 1e-4), ('max_learning_rate', 1e-3), ('warmup_steps', 10000),
        ('grad_clip_thresh', float('inf')),
        # modules
        ('prosody_encoder', _fft_block_cfg(8)),
        ('phoneme_encoder', _fft_block_cfg(2)),
        ('local_prosody_predictor', {'nb_blocks': 1, 'conv_kernel': 3, 'conv_channels': 256, 'conv_dropout': 0.1}),
        ('gaussian_upsampling_module', {'conv_kernel': 3}),
        ('frame_decoder', _fft_block_cfg(2, with_hidden=False)),
        # MI355X build additions
        ('compute_dtype', 'bf16'),
        # the micro-batches of an optimizer step as ONE pass over their concatenation, each utterance keeping its own micro-batch's
        # padded length as a hard sequence end: same gradients as `accumulation_steps` separate passes (train.py:379-401)
        ('group_micro_batches', True),
        # must come from the caller
        ('training_files', None), ('validation_files', None), ('output_directory', None),
        ('language', None), ('speakers', None),
        # inferred when left empty
        ('stats', {}), ('symbols', []), ('n_speakers', 0), ('speakers_id', []),
    ]


class HyperParams(object):
    def __init__(self, verbose=True, **kwargs):
        if verbose:
            for line in ('--' * 30, 'SETTING HYPER-PARAMETERS', '--' * 30):
                _logger.info(line)
        for name, value in _defaults():
            setattr(self, name, value)
        # caller overrides
        for name, value in kwargs.items():
            previous = getattr(self, name, None)
            if verbose and previous is not None and previous != value:
                _logger.warning(f'Changing parameter "{name}" = {value} (was {previous})')
            setattr(self, name, value)
        missing = [name for name, value in vars(self).items() if value is None]
        assert not missing, _logger.error(f'Hyper-parameter(s) {missing} are None -- please specify a value')
        self._set_default_hyper_params(verbose=verbose)

    def _set_default_hyper_params(self, verbose):
        self.update_mfa_paths()
        # features stats written by the pre-processing step, when present
        stats_file = os.path.join(self.output_directory, 'stats.json')
        if not self.stats and os.path.isfile(stats_file):
            with open(stats_file) as f:
                self.stats = json.load(f)
        # symbols
        if not self.symbols:
            if self.language != 'english':
                _logger.error(f'Language: {self.language} -- No default value for "symbols" -- please specify a value')
                sys.exit(1)
            self.symbols = symbols_english
            if verbose:
                _logger.info(f'Language: {self.language} -- {len(self.symbols)} symbols used')
        self.n_symbols = len(self.symbols)
        # index 0 is what the collate pads with
        assert self.symbols.index(pad) == 0, _logger.error(f'Padding symbol "{pad}" must be at index 0')
        # speakers
        if not self.speakers_id:
            self.speakers_id = list(range(len(self.speakers)))
            if verbose:
                _logger.info(f'Nb speakers: {len(self.speakers)} -- Changed "speakers_id" to {self.speakers_id}')
        nb_ids = len(set(self.speakers_id))
        if self.n_speakers == 0:
            self.n_speakers = nb_ids + 1
            if verbose:
                _logger.info(f'Nb speakers: {nb_ids} -- Changed "n_speakers" to {self.n_speakers}\n')
        assert self.n_speakers >= nb_ids, _logger.error(
            f'Parameter "n_speakers" must be superior or equal to the number of speakers -- '
            f'"n_speakers" = {self.n_speakers} -- Number of speakers = {nb_ids}')
        assert len(self.speakers) == len(set(self.speakers)), _logger.error(
            f'Speakers are not unique: {len(self.speakers)} -- {len(set(self.speakers))}')
        assert len(self.speakers) == len(self.speakers_id), _logger.error(
            f'Parameters "speakers" and "speakers_id" don\'t have the same length: '
            f'{len(self.speakers)} -- {len(self.speakers_id)}')
        assert self.filter_length % self.hop_length == 0, _logger.error('filter_length must be a multiple of hop_length')
        assert self.compute_dtype in ('bf16', 'fp32'), _logger.error('compute_dtype must be "bf16" or "fp32"')

    def update_mfa_paths(self):
        ''' Point the MFA attributes at the current home directory (existence is not checked here). '''
        root = os.path.join(str(Path.home()), 'Documents', 'MFA', 'pretrained_models')
        self.mfa_dictionary = os.path.join(root, 'dictionary', f'{self.language}.dict')
        self.mfa_g2p_model = os.path.join(root, 'g2p', f'{self.language}_g2p.zip')
        self.mfa_acoustic_model = os.path.join(root, 'acoustic', f'{self.language}.zip')

    def save_hyper_params(self, json_file):
        os.makedirs(os.path.dirname(json_file), exist_ok=True)
        with open(json_file, 'w') as f:
            json.dump(dict(vars(self)), f, indent=4, sort_keys=True)
