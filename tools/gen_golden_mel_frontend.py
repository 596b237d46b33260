#!/usr/bin/env python
"""Fixture for the mel / energy front-end: runs the REFERENCE's `mel_spectrogram_HiFi` and
`extract_energy` (`/root/reference/src/daft_exprt/extract_features.py`) on seeded synthetic waveforms in the build
container and stores inputs + outputs in tests/golden/mel_frontend.npz.

`librosa` is not installed here, so `librosa.filters.mel` -- the one third-party function on this path -- is replaced by
the restatement `oracle.mel_frontend_cpu.mel_filterbank` (positional signature (sr, n_fft, n_mels, fmin, fmax) as the
reference calls it).  Everything else (windowing, STFT, magnitude, log-clamp, energy) is the reference's own code.
Run:  python tools/gen_golden_mel_frontend.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import gen_goldens  # noqa: E402  (shims + hparams helper)
from oracle.mel_frontend_cpu import mel_filterbank  # noqa: E402


def synth_wav(rng, n):
    t = np.arange(n) / 22050.
    f0 = rng.uniform(90., 300.)
    wav = sum(rng.uniform(0.05, 0.3) / (h + 1) * np.sin(2 * np.pi * f0 * (h + 1) * t + rng.uniform(0, 6.28)) for h in range(12))
    wav = wav * (0.5 + 0.5 * np.sin(2 * np.pi * 3. * t)) + 0.01 * rng.randn(n)
    if n > 4000:
        wav[1000:2500] = 0.   # digital silence: exercises the 1e-9 / min_clipping floors
    return np.clip(wav, -1., 1.).astype(np.float32)


def main():
    gen_goldens.install_shims()
    sys.modules['librosa.filters'].mel = lambda sr, n_fft, n_mels, fmin, fmax: mel_filterbank(sr, n_fft, n_mels, fmin, fmax)
    import daft_exprt.hparams as ref_hparams
    import daft_exprt.extract_features as ref_fx
    ref_fx.librosa_mel_fn = sys.modules['librosa.filters'].mel
    rng = np.random.RandomState(4321)
    fx = {}
    lengths = [1024, 1279, 5000, 12345, 22050]
    for centered in (True, False):
        hp = gen_goldens.make_hparams(ref_hparams, centered=centered)
        for i, n in enumerate(lengths):
            wav = synth_wav(rng, n)
            mel = ref_fx.mel_spectrogram_HiFi(wav, hp)
            mel = mel.reshape(hp.n_mel_channels, -1)
            energy = ref_fx.extract_energy(np.exp(mel))
            key = f'c{int(centered)}_{i}'
            fx[f'{key}_wav'], fx[f'{key}_mel'], fx[f'{key}_energy'] = wav, mel.astype(np.float32), energy.astype(np.float32)
            print(key, n, mel.shape, float(mel.min()), float(mel.max()))
    np.savez_compressed(os.path.join(gen_goldens.OUT, 'mel_frontend.npz'), **fx)


if __name__ == '__main__':
    main()
