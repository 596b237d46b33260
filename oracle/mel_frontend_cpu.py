"""CPU restatement of the reference's mel / energy front-end (TEST INFRASTRUCTURE ONLY, like the rest of oracle/).

Follows `extract_features.py:330-359` (`mel_spectrogram_HiFi`: Hann-windowed `torch.stft` 1024/256, magnitude
`sqrt(re^2 + im^2 + 1e-9)`, mel filterbank matmul, `log(clamp(., min_clipping))`) and `extract_features.py:299-304`
+ `generate.py:457` (`frames energy = || exp(log-mel) ||_2` over the mel channels).

Pin: `tests/golden/mel_frontend.npz` holds outputs of the reference's own `mel_spectrogram_HiFi` run in the build
container (tools/gen_goldens.py) -- with ONE substitution: `librosa.filters.mel` is absent there, so the fixture
generator injects `mel_filterbank()` below in its place.  Everything except the filterbank constants is therefore pinned
by the reference; the filterbank itself is a restatement of librosa 0.8 `filters.mel(sr, n_fft, n_mels, fmin, fmax,
htk=False, norm='slaney')` (the version range `setup.py` allows) and is "parity unpinned" (DESIGN.md section 0, row f4).
"""
import numpy as np
import torch


def _hz_to_mel(f):
    ''' Slaney (Auditory Toolbox) mel scale: linear below 1 kHz, logarithmic above '''
    f = np.asarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz, min_log_mel, logstep = 1000.0, 1000.0 / f_sp, np.log(6.4) / 27.0
    return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-30) / min_log_hz) / logstep, mels)


def _mel_to_hz(m):
    m = np.asarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    min_log_hz, min_log_mel, logstep = 1000.0, 1000.0 / f_sp, np.log(6.4) / 27.0
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)


def mel_filterbank(sr, n_fft, n_mels, fmin, fmax):
    ''' (n_mels, 1 + n_fft // 2) float32 triangular filters, Slaney-normalised (area 1 per filter) '''
    fmax = float(sr) / 2 if fmax is None else float(fmax)
    fftfreqs = np.linspace(0., float(sr) / 2, 1 + n_fft // 2)
    mel_f = _mel_to_hz(np.linspace(_hz_to_mel(fmin), _hz_to_mel(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fftfreqs[None, :]
    weights = np.zeros((n_mels, 1 + n_fft // 2))
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        weights[i] = np.maximum(0., np.minimum(lower, upper))
    weights *= (2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels]))[:, None]
    return weights.astype(np.float32)


def mel_spectrogram(wav, hparams):
    ''' wav (n_samples,) float32 in [-1, 1] -> log-mel (n_mel_channels, n_frames) float32; `extract_features.py:330-359` '''
    wav = torch.as_tensor(np.asarray(wav, dtype=np.float32))
    n_fft, hop = int(hparams.filter_length), int(hparams.hop_length)
    fb = torch.from_numpy(mel_filterbank(hparams.sampling_rate, n_fft, hparams.n_mel_channels, hparams.mel_fmin, hparams.mel_fmax))
    spec = torch.stft(wav, n_fft, hop_length=hop, win_length=n_fft, window=torch.hann_window(n_fft), center=bool(hparams.centered),
                      pad_mode='reflect', normalized=False, onesided=True, return_complex=True)
    mag = torch.sqrt(spec.real.pow(2) + spec.imag.pow(2) + 1e-9)
    return torch.log(torch.clamp(torch.matmul(fb, mag), min=float(hparams.min_clipping))).numpy()


def frames_energy(log_mel):
    ''' `extract_features.py:299-304` applied to `np.exp(mel_spec)` (`generate.py:457`, `extract_features.py:465-466`) '''
    return np.linalg.norm(np.exp(log_mel), axis=0)
