"""On-the-fly audio features on MI355X: log-mel (+ FOA intensity vector).

Host-side mirror of the reference seam `utils/feature.py` (reference: LogmelIV_Extractor :20-56,
Logmel_Extractor :59-91, intensityvector :93-117; selected by utils/config.py:24-32). Same constructor
(`cfg['data'][nfft, hoplen, window, n_mels, sample_rate]`), same forward contract
(`f32[B, C, L] -> f32[B, C(+3), 1 + L // hoplen, n_mels]`, ValueError unless x.ndim == 3).
All arithmetic is the fused HIP kernel `pseld_logmel_iv_fwd` (csrc/feature.hip); this file only builds the
immutable tables (window, FFT twiddles, mel filter bank in compact column form).
"""
import ctypes
import math

import torch
import torch.nn as nn

from .. import _lib

eps = torch.finfo(torch.float32).eps
window_fn_dict = {
    'hann': torch.hann_window,
    'hamming': torch.hamming_window,
    'blackman': torch.blackman_window,
    'bartlett': torch.bartlett_window,
}


def mel_filter_bank(n_freqs, f_min, f_max, n_mels, sample_rate):
    """Triangular HTK-scale, slaney-normalised filter bank f32[n_freqs, n_mels] — the table torchaudio 2.2.1's
    `MelScale(norm='slaney', mel_scale='htk')` holds as `.fb` (reference call site utils/feature.py:32-34)."""
    bins_hz = torch.linspace(0, sample_rate // 2, n_freqs)
    mel_lo = 2595.0 * math.log10(1.0 + f_min / 700.0)
    mel_hi = 2595.0 * math.log10(1.0 + f_max / 700.0)
    edges_hz = 700.0 * (10.0 ** (torch.linspace(mel_lo, mel_hi, n_mels + 2) / 2595.0) - 1.0)
    width = edges_hz[1:] - edges_hz[:-1]
    dist = edges_hz.unsqueeze(0) - bins_hz.unsqueeze(1)           # [n_freqs, n_mels + 2]
    falling = -dist[:, :-2] / width[:-1]
    rising = dist[:, 2:] / width[1:]
    fb = torch.clamp(torch.minimum(falling, rising), min=0.0)
    fb = fb * (2.0 / (edges_hz[2:n_mels + 2] - edges_hz[:n_mels])).unsqueeze(0)
    return fb


def compact_filter_bank(fb):
    """Column-compact form of a filter bank whose columns have contiguous support:
    (lo[m], cnt[m], off[m], weights) with filter m = weights[off[m] : off[m]+cnt[m]] over bins lo[m]..."""
    n_mels = fb.shape[1]
    lo, cnt, off, w = [], [], [], []
    for m in range(n_mels):
        nz = torch.nonzero(fb[:, m]).flatten()
        if nz.numel() == 0:
            lo.append(0); cnt.append(0); off.append(len(w)); continue
        a, b = int(nz[0]), int(nz[-1]) + 1
        lo.append(a); cnt.append(b - a); off.append(len(w))
        w.extend(fb[a:b, m].tolist())
    return (torch.tensor(lo, dtype=torch.int32), torch.tensor(cnt, dtype=torch.int32),
            torch.tensor(off, dtype=torch.int32), torch.tensor(w, dtype=torch.float32))


class _HipSpectralFrontEnd(nn.Module):
    with_iv = False

    def __init__(self, cfg):
        super().__init__()
        data = cfg['data']
        assert data['window'] in window_fn_dict.keys(), \
            "window must be in {}, but got {}".format(window_fn_dict.keys(), data['window'])
        self.n_fft = int(data['nfft'])
        self.hop = int(data['hoplen'])
        self.n_mels = int(data['n_mels'])
        self.sample_rate = data['sample_rate']
        window = window_fn_dict[data['window']](self.n_fft)
        n = torch.arange(self.n_fft, dtype=torch.float64)
        ang = -2.0 * math.pi * n / self.n_fft
        twiddle = torch.stack([torch.cos(ang), torch.sin(ang)], dim=1).to(torch.float32)
        fb = mel_filter_bank(self.n_fft // 2 + 1, 20.0, self.sample_rate / 2, self.n_mels, self.sample_rate)
        lo, cnt, off, w = compact_filter_bank(fb)
        self.register_buffer('window', window, persistent=False)
        self.register_buffer('twiddle', twiddle, persistent=False)
        self.register_buffer('fb', fb, persistent=False)
        self.register_buffer('mel_lo', lo, persistent=False)
        self.register_buffer('mel_cnt', cnt, persistent=False)
        self.register_buffer('mel_off', off, persistent=False)
        self.register_buffer('mel_w', w, persistent=False)

    def forward(self, x):
        """
        input:
            (batch_size, channels, data_length)
        output:
            (batch_size, channels(+3), time_steps, mel_bins)
        """
        if x.ndim != 3:
            raise ValueError("x shape must be (batch_size, num_channels, data_length)\n \
                            Now it is {}".format(x.shape))
        if not x.is_cuda:
            raise _lib.PseldError("feature extractor input must live on the MI355X (no CPU fallback)")
        if self.window.device != x.device:
            self.to(x.device)
        x = x.contiguous().float()
        B, C, L = x.shape
        T = 1 + L // self.hop
        n_out = C + (3 if self.with_iv else 0)
        out = torch.empty((B, n_out, T, self.n_mels), dtype=torch.float32, device=x.device)
        rc = _lib.lib().pseld_logmel_iv_fwd(
            _lib.ptr(x), _lib.ptr(out), B, C, L, self.hop, self.n_fft, self.n_mels,
            _lib.ptr(self.window), _lib.ptr(self.twiddle), _lib.ptr(self.mel_lo), _lib.ptr(self.mel_cnt),
            _lib.ptr(self.mel_off), _lib.ptr(self.mel_w), int(self.mel_w.numel()), int(self.with_iv),
            1e-10, eps, _lib.stream_ptr())
        _lib.check(rc, "pseld_logmel_iv_fwd")
        return out


class LogmelIV_Extractor(_HipSpectralFrontEnd):
    """4-ch FOA waveform -> 4 log-mel + 3 mel-projected, L2-normalised intensity-vector channels."""
    with_iv = True


class Logmel_Extractor(_HipSpectralFrontEnd):
    """Per-channel log-mel only."""
    with_iv = False
