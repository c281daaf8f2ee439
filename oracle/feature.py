"""Oracle: log-mel + intensity-vector features (test infrastructure, CPU only).

Follows reference utils/feature.py:20-56 (LogmelIV_Extractor), :59-91 (Logmel_Extractor), :93-117
(intensityvector). torchaudio==2.2.1 (requirements.txt:11) is a third-party dependency absent here; its
transforms are restated from their published algorithm:
  Spectrogram(n_fft, hop, win_length=n_fft, window_fn, power=None) = torch.stft(center=True,
      pad_mode='reflect', normalized=False, onesided=True, return_complex=True)
  MelScale(n_mels, sample_rate, f_min, f_max, n_stft, norm='slaney', mel_scale='htk'):
      fb = melscale_fbanks(...); out = (spec^T @ fb)^T
  AmplitudeToDB('power', top_db=None): 10*log10(clamp(x, 1e-10)) - 10*log10(max(1e-10, 1.0))
"""
import math

import numpy as np
import torch

EPS = float(torch.finfo(torch.float32).eps)   # feature.py:8
WINDOWS = {'hann': torch.hann_window, 'hamming': torch.hamming_window,
           'blackman': torch.blackman_window, 'bartlett': torch.bartlett_window}   # feature.py:9-14


def hz_to_mel_htk(f):
    return 2595.0 * math.log10(1.0 + f / 700.0)


def melscale_fbanks(n_freqs, f_min, f_max, n_mels, sample_rate):
    """torchaudio.functional.melscale_fbanks(norm='slaney', mel_scale='htk') restated."""
    all_freqs = torch.linspace(0, sample_rate // 2, n_freqs)
    m_pts = torch.linspace(hz_to_mel_htk(f_min), hz_to_mel_htk(f_max), n_mels + 2)
    f_pts = 700.0 * (10.0 ** (m_pts / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)
    down = (-1.0 * slopes[:, :-2]) / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    fb = torch.max(torch.zeros(1), torch.min(down, up))
    enorm = 2.0 / (f_pts[2:n_mels + 2] - f_pts[:n_mels])
    return fb * enorm.unsqueeze(0)


def spectrogram_complex(x, n_fft, hop, window):
    """x [..., L] -> complex [..., n_fft//2+1, T] (torchaudio Spectrogram(power=None))."""
    shape = x.shape
    spec = torch.stft(x.reshape(-1, shape[-1]), n_fft=n_fft, hop_length=hop, win_length=n_fft, window=window,
                      center=True, pad_mode='reflect', normalized=False, onesided=True, return_complex=True)
    return spec.reshape(shape[:-1] + spec.shape[-2:])


def amplitude_to_db_power(x):
    return 10.0 * torch.log10(torch.clamp(x, min=1e-10))


def intensity_vector(real, imag, fb):
    """reference feature.py:93-117; real/imag [B, 4, T, F] -> [B, 3, T, n_mels]."""
    w_re, w_im = real[:, 0], imag[:, 0]
    comps = [w_re * real[:, c] + w_im * imag[:, c] for c in (1, 2, 3)]
    norm = torch.sqrt(comps[0] ** 2 + comps[1] ** 2 + comps[2] ** 2) + EPS
    return torch.stack([torch.matmul(c / norm, fb) for c in comps], dim=1)


def logmel(x, n_fft=1024, hop=240, window='hann', n_mels=64, sample_rate=24000):
    """reference feature.py:78-91."""
    if x.ndim != 3:
        raise ValueError("x shape must be (batch_size, num_channels, data_length)")
    fb = melscale_fbanks(n_fft // 2 + 1, 20.0, sample_rate / 2, n_mels, sample_rate)
    spec = spectrogram_complex(x, n_fft, hop, WINDOWS[window](n_fft))
    mel = torch.matmul((torch.abs(spec) ** 2).transpose(-1, -2), fb).transpose(-1, -2)
    return amplitude_to_db_power(mel).transpose(-1, -2)


def logmel_iv(x, n_fft=1024, hop=240, window='hann', n_mels=64, sample_rate=24000):
    """reference feature.py:39-56: [B, 4, L] -> [B, 7, T, n_mels]."""
    if x.ndim != 3:
        raise ValueError("x shape must be (batch_size, num_channels, data_length)")
    fb = melscale_fbanks(n_fft // 2 + 1, 20.0, sample_rate / 2, n_mels, sample_rate)
    spec = spectrogram_complex(x, n_fft, hop, WINDOWS[window](n_fft))
    mel = torch.matmul((torch.abs(spec) ** 2).transpose(-1, -2), fb).transpose(-1, -2)
    lm = amplitude_to_db_power(mel).transpose(-1, -2)
    iv = intensity_vector(spec.real.transpose(-1, -2), spec.imag.transpose(-1, -2), fb)
    return torch.cat((lm, iv), dim=1)


def logmel_iv_f64(x, n_fft=1024, hop=240, n_mels=64, sample_rate=24000):
    """Independent float64 numpy restatement (explicit reflect-pad framing + rFFT, periodic Hann) used to
    cross-check the torch.stft-based restatement above. x: numpy [B, 4, L]."""
    x = np.asarray(x, dtype=np.float64)
    B, C, L = x.shape
    T = 1 + L // hop
    pad = n_fft // 2
    xp = np.pad(x, ((0, 0), (0, 0), (pad, pad)), mode='reflect')
    n = np.arange(n_fft)
    win = 0.5 - 0.5 * np.cos(2.0 * np.pi * n / n_fft)
    idx = hop * np.arange(T)[:, None] + n[None, :]
    spec = np.fft.rfft(xp[:, :, idx] * win, axis=-1)            # [B, C, T, F]
    fb = melscale_fbanks(n_fft // 2 + 1, 20.0, sample_rate / 2, n_mels, sample_rate).double().numpy()
    lm = 10.0 * np.log10(np.maximum((np.abs(spec) ** 2) @ fb, 1e-10))
    re, im = spec.real, spec.imag
    comps = [re[:, 0] * re[:, c] + im[:, 0] * im[:, c] for c in (1, 2, 3)]
    norm = np.sqrt(comps[0] ** 2 + comps[1] ** 2 + comps[2] ** 2) + EPS
    iv = np.stack([(c / norm) @ fb for c in comps], axis=1)
    return np.concatenate([lm, iv], axis=1)
