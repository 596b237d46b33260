"""Mel-spectrogram / frame-energy front-end of the synthesis path on the GPU (reference: `extract_features.py:299-304,
330-359`; callers `generate.py:455-457`, `extract_features.py:429, 465-466`, `fine_tune.py:102`).

`mel_spectrogram_HiFi(wav, hparams)` and `extract_energy(mel_spec)` keep the reference's signatures (NumPy in, NumPy
out, one utterance); `mel_spectrogram_batch` is the batched device entry.  The mel filterbank is built here from the
published Slaney formula (what `librosa.filters.mel(sr, n_fft, n_mels, fmin, fmax)` computes with its defaults
htk=False, norm='slaney'); librosa itself is not a dependency.  No CPU fallback: without the HIP library / a GPU these
functions raise.
"""
import numpy as np
import torch

from daft_exprt import _hip as H

_TABLES = {}


def _hz_to_mel(f):
    f = np.asarray(f, dtype=np.float64)
    f_sp, min_log_hz = 200.0 / 3, 1000.0
    min_log_mel, logstep = min_log_hz / f_sp, np.log(6.4) / 27.0
    return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-30) / min_log_hz) / logstep, f / f_sp)


def _mel_to_hz(m):
    m = np.asarray(m, dtype=np.float64)
    f_sp, min_log_hz = 200.0 / 3, 1000.0
    min_log_mel, logstep = min_log_hz / f_sp, np.log(6.4) / 27.0
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)


def mel_filter_bank(sr, n_fft, n_mels, fmin, fmax):
    ''' (n_mels, 1 + n_fft // 2) float32: triangular filters on the Slaney mel scale, each normalised to unit area '''
    fmax = float(sr) / 2 if fmax is None else float(fmax)
    freqs = np.linspace(0., float(sr) / 2, 1 + n_fft // 2)
    edges = _mel_to_hz(np.linspace(_hz_to_mel(fmin), _hz_to_mel(fmax), n_mels + 2))
    width = np.diff(edges)
    ramps = edges[:, None] - freqs[None, :]
    fb = np.maximum(0., np.minimum(-ramps[:-2] / width[:-1, None], ramps[2:] / width[1:, None]))
    fb *= (2.0 / (edges[2:] - edges[:-2]))[:, None]
    return fb.astype(np.float32)


def _tables(hparams, device):
    key = (str(device), hparams.sampling_rate, hparams.filter_length, hparams.n_mel_channels, hparams.mel_fmin, hparams.mel_fmax)
    if key not in _TABLES:
        n_fft = int(hparams.filter_length)
        basis = torch.empty(2 * n_fft, dtype=torch.float32, device=device)      # FFT twiddles exp(-2 pi i t / n_fft)
        window = torch.empty(n_fft, dtype=torch.float32, device=device)
        H.check(H.lib().dx_mel_tables(H.ptr(basis), H.ptr(window), n_fft, H.stream()))
        fb = mel_filter_bank(hparams.sampling_rate, n_fft, hparams.n_mel_channels, hparams.mel_fmin, hparams.mel_fmax)
        nz = fb > 0
        lo = np.where(nz.any(1), nz.argmax(1), 0).astype(np.int32)
        hi = np.where(nz.any(1), fb.shape[1] - nz[:, ::-1].argmax(1), 0).astype(np.int32)
        _TABLES[key] = (basis, window, torch.from_numpy(fb).to(device), torch.from_numpy(lo).to(device), torch.from_numpy(hi).to(device))
    return _TABLES[key]


def nb_frames(n_samples, hparams):
    ''' frames torch.stft produces for a waveform of n_samples (`extract_features.py:347-348`) '''
    n_fft, hop = int(hparams.filter_length), int(hparams.hop_length)
    if hparams.centered:
        return 1 + n_samples // hop
    return 1 + (n_samples - n_fft) // hop if n_samples >= n_fft else 0


def mel_spectrogram_batch(wavs, n_samples, hparams):
    ''' wavs (B, S) float32 device tensor (right-padded), n_samples (B,) int64 device tensor.
        Returns (log-mel (B, n_mel, T) fp32, frame energies (B, T) fp32, n_frames (B,) int64); frames >= n_frames[b] are 0. '''
    H.require_gpu(wavs, n_samples)
    assert wavs.dtype == torch.float32 and wavs.stride(1) == 1 and n_samples.dtype == torch.int64
    B, S = wavs.shape
    dev = wavs.device
    n_fft, hop, n_mel = int(hparams.filter_length), int(hparams.hop_length), int(hparams.n_mel_channels)
    if hparams.centered and int(n_samples.min()) <= n_fft // 2:
        raise ValueError('mel_spectrogram: reflect padding needs more than filter_length / 2 samples')
    T = max(1, nb_frames(S, hparams))
    basis, window, fb, lo, hi = _tables(hparams, dev)
    mel = torch.empty((B, n_mel, T), dtype=torch.float32, device=dev)
    energy = torch.empty((B, T), dtype=torch.float32, device=dev)
    n_frames = torch.empty((B,), dtype=torch.int64, device=dev)
    H.check(H.lib().dx_mel_spectrogram(H.ptr(wavs), wavs.stride(0), H.ptr(n_samples), H.ptr(basis), H.ptr(window), H.ptr(fb),
                                       H.ptr(lo), H.ptr(hi), H.ptr(mel), H.ptr(energy), H.ptr(n_frames), B, T, n_fft,
                                       hop, n_mel, int(bool(hparams.centered)), float(hparams.min_clipping), H.stream()))
    return mel, energy, n_frames


def mel_spectrogram_HiFi(wav, hparams, device='cuda:0'):
    ''' reference signature (`extract_features.py:330`): wav (T,) in [-1, 1] -> log-mel (n_mel_channels, n_frames) NumPy '''
    w = torch.as_tensor(np.asarray(wav, dtype=np.float32)).reshape(1, -1).to(device)
    n = torch.tensor([w.shape[1]], dtype=torch.int64, device=device)
    mel, _, nfr = mel_spectrogram_batch(w, n, hparams)
    return mel[0, :, :int(nfr[0])].cpu().numpy()


def extract_energy(mel_spec):
    ''' `extract_features.py:299-304`: L2 norm over the mel channels (callers pass np.exp(log-mel)); tiny, host side '''
    return np.linalg.norm(mel_spec, axis=0)
