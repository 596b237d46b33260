"""Mel / energy front-end.  CPU: the oracle restatement against the fixture produced by the reference's
own `mel_spectrogram_HiFi` (tools/gen_golden_mel_frontend.py); GPU: the HIP kernels against the same fixture.
Tolerances: oracle vs fixture 1e-5 (same torch.stft); GPU vs fixture 2e-3 absolute on the log-mel values above the
clamp floor region (DFT as an fp32 GEMM vs torch's FFT: both fp32, different summation order; log amplifies relative
error of near-silent bins) and 1e-3 relative on the frame energies."""
import os

import numpy as np
import pytest
import torch

from tests.util import make_hparams

GOLD = os.path.join(os.path.dirname(__file__), 'golden', 'mel_frontend.npz')


def _cases():
    fx = np.load(GOLD)
    for centered in (1, 0):
        for i in range(5):
            k = f'c{centered}_{i}'
            yield centered, fx[f'{k}_wav'], fx[f'{k}_mel'], fx[f'{k}_energy']


def test_oracle_matches_reference_fixture():
    from oracle import mel_frontend_cpu as M
    for centered, wav, mel, energy in _cases():
        hp = make_hparams(centered=bool(centered))
        out = M.mel_spectrogram(wav, hp)
        assert out.shape == mel.shape
        assert np.abs(out - mel).max() <= 1e-5
        assert np.abs(M.frames_energy(out) - energy).max() <= 1e-5 * max(1., energy.max())


def test_filterbank_properties_and_product_copy():
    ''' the product's filterbank is the oracle's (two independent restatements of the Slaney formula), unit-area triangles '''
    from oracle import mel_frontend_cpu as M
    import importlib
    fe = importlib.import_module('daft_exprt.extract_features')
    a, b = M.mel_filterbank(22050, 1024, 80, 0, 8000), fe.mel_filter_bank(22050, 1024, 80, 0, 8000)
    assert a.shape == (80, 513) and np.array_equal(a, b)
    assert (a >= 0).all() and (a.sum(1) > 0).all()
    hz = np.linspace(0, 11025, 513)
    area = np.trapz(a, hz, axis=1)
    assert np.abs(area - 1.).max() < 0.1           # Slaney normalisation: unit area up to the 21.5 Hz bin discretisation
    assert a[:, hz > 8000].max() == 0.             # nothing above fmax
    hp = make_hparams(centered=True)
    assert fe.nb_frames(22050, hp) == 87 and fe.nb_frames(1024, make_hparams(centered=False)) == 1


def test_mel_scale_known_answers_from_librosa_documentation():
    ''' `librosa` is absent from the build image, so `librosa.filters.mel` (extract_features.py:346) cannot be run here.  What CAN be
        pinned independently of this repository are the worked examples in librosa's public documentation of the functions the
        filterbank is built from (Slaney scale, htk=False): hz_to_mel(60) = 0.9, hz_to_mel([110, 220, 440]) = [1.65, 3.3, 6.6],
        mel_to_hz(3) = 200, mel_to_hz([1..5]) = [66.667, 133.333, 200, 266.667, 333.333], and the head of
        mel_frequencies(n_mels=40) = [0, 85.317, 170.635, 255.952, 341.269, 426.586, 511.904, 597.221, 682.538, 767.855, 853.173,
        938.49, 1024.856, ...] up to fmax = 11025 -- the entry 1024.856 is the first one above the 1 kHz break and fixes both the
        break point and the logarithmic step ln(6.4) / 27 (it moves to 1024.1 / 1025.6 for a step 3 % off).  The remaining freedom
        of `filters.mel` -- triangles between consecutive centre frequencies, `norm='slaney'` = 2 / (f[i+2] - f[i]) -- is checked
        by construction below (peak position, linear flanks, area). '''
    from oracle import mel_frontend_cpu as M
    import importlib
    fe = importlib.import_module('daft_exprt.extract_features')
    assert abs(float(M._hz_to_mel(60.)) - 0.9) < 1e-12
    assert np.allclose(M._hz_to_mel([110., 220., 440.]), [1.65, 3.3, 6.6], atol=1e-12)
    assert abs(float(M._mel_to_hz(3.)) - 200.) < 1e-9
    assert np.allclose(M._mel_to_hz([1., 2., 3., 4., 5.]), [66.667, 133.333, 200., 266.667, 333.333], atol=5e-4)
    head = [0., 85.317, 170.635, 255.952, 341.269, 426.586, 511.904, 597.221, 682.538, 767.855, 853.173, 938.49, 1024.856]
    f40 = M._mel_to_hz(np.linspace(M._hz_to_mel(0.), M._hz_to_mel(11025.), 40))
    assert np.abs(f40[:13] - np.array(head)).max() < 6e-4 and abs(f40[-1] - 11025.) < 1e-6
    # the filterbank of the model's configuration, both implementations: triangle i peaks at centre i + 1 of the 82 mel-spaced
    # frequencies with height 2 / (f[i+2] - f[i]) and falls linearly to zero at centres i and i + 2
    ctr = M._mel_to_hz(np.linspace(M._hz_to_mel(0.), M._hz_to_mel(8000.), 82))
    hz = np.linspace(0., 11025., 513)
    for fb in (M.mel_filterbank(22050, 1024, 80, 0, 8000), fe.mel_filter_bank(22050, 1024, 80, 0, 8000)):
        for i in (0, 1, 17, 40, 63, 79):
            lo, mid, hi = ctr[i], ctr[i + 1], ctr[i + 2]
            peak = 2. / (hi - lo)
            want = np.where(hz <= mid, (hz - lo) / (mid - lo), (hi - hz) / (hi - mid)).clip(min=0.) * peak
            assert np.abs(fb[i] - want).max() <= 1e-6 * max(1., peak), i


@pytest.mark.gpu
def test_gpu_front_end_matches_reference_fixture():
    from daft_exprt import extract_features as fe
    dev = torch.device('cuda:0')
    for centered in (1, 0):
        hp = make_hparams(centered=bool(centered))
        cases = [c for c in _cases() if c[0] == centered]
        S = max(len(c[1]) for c in cases)
        wavs = torch.zeros(len(cases), S)
        for i, c in enumerate(cases):
            wavs[i, :len(c[1])] = torch.from_numpy(c[1])
        n = torch.tensor([len(c[1]) for c in cases], dtype=torch.int64)
        mel, energy, nfr = fe.mel_spectrogram_batch(wavs.to(dev), n.to(dev), hp)
        mel, energy, nfr = mel.cpu().numpy(), energy.cpu().numpy(), nfr.cpu().numpy()
        for i, (_, wav, ref_mel, ref_en) in enumerate(cases):
            T = ref_mel.shape[1]
            assert nfr[i] == T
            assert np.abs(mel[i, :, :T] - ref_mel).max() <= 2e-3, (centered, i, np.abs(mel[i, :, :T] - ref_mel).max())
            assert np.abs(energy[i, :T] - ref_en).max() <= 1e-3 * max(1e-3, ref_en.max())
            assert (mel[i, :, T:] == 0).all() and (energy[i, T:] == 0).all()
        # single-utterance reference signature
        one = fe.mel_spectrogram_HiFi(cases[2][1], hp)
        assert one.shape == cases[2][2].shape and np.abs(one - cases[2][2]).max() <= 2e-3
