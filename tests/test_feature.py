"""K1 feature kernel vs the CPU oracle (oracle/feature.py). Tolerances: log-mel in dB 2e-3 abs (1e-3 rel of the
~20-80 dB range is far looser; fp32 FFT round-off gives ~1e-5), intensity-vector mel 1e-4 abs (values in [-1, 1])."""
import numpy as np
import pytest
import torch

from oracle import feature as of

CFG = {'data': {'nfft': 1024, 'hoplen': 240, 'window': 'hann', 'n_mels': 64, 'sample_rate': 24000,
                'audio_feature': 'logmelIV'}}


def _wave(B, C, L, seed=0):
    g = torch.Generator().manual_seed(seed)
    x = 0.1 * torch.randn(B, C, L, generator=g)
    t = torch.arange(L) / 24000.0
    x[:, 0] += 0.05 * torch.sin(2 * np.pi * 440.0 * t)      # a tone so some bins dominate
    if C > 1:
        x[:, 1] += 0.03 * torch.sin(2 * np.pi * 3000.0 * t + 0.3)
    return x


def test_oracle_matches_independent_f64_restatement():
    x = _wave(2, 4, 9600, seed=1)
    a = of.logmel_iv(x).numpy()
    b = of.logmel_iv_f64(x.numpy())
    assert a.shape == (2, 7, 41, 64)
    assert np.abs(a[:, :4] - b[:, :4]).max() < 5e-4     # dB
    assert np.abs(a[:, 4:] - b[:, 4:]).max() < 5e-6


def test_oracle_rejects_bad_rank():
    with pytest.raises(ValueError):
        of.logmel_iv(torch.zeros(4, 4800))


@pytest.mark.gpu
@pytest.mark.parametrize("B,L", [(1, 4800), (3, 24000), (2, 240000)])
def test_logmel_iv_kernel_matches_oracle(dev, B, L):
    from pseldnets_amd.utils.config import get_afextractor
    x = _wave(B, 4, L, seed=B)
    ref = of.logmel_iv(x)
    ext = get_afextractor(CFG).to(dev)
    out = ext(x.to(dev)).cpu()
    assert out.shape == ref.shape == (B, 7, 1 + L // 240, 64)
    err_db = (out[:, :4] - ref[:, :4]).abs().max().item()
    err_iv = (out[:, 4:] - ref[:, 4:]).abs().max().item()
    print(f"logmel max|d| = {err_db:.3e} dB, iv max|d| = {err_iv:.3e}")
    assert err_db < 2e-3
    assert err_iv < 1e-4


@pytest.mark.gpu
def test_logmel_only_and_edge_cases(dev):
    from pseldnets_amd.utils.feature import Logmel_Extractor, LogmelIV_Extractor
    x = _wave(2, 1, 7200, seed=5)
    ext = Logmel_Extractor(CFG).to(dev)
    out = ext(x.to(dev)).cpu()
    ref = of.logmel(x)
    assert out.shape == ref.shape
    assert (out - ref).abs().max().item() < 2e-3
    # silence: every bin clamps to amin -> exactly -100 dB, IV = 0
    z = torch.zeros(1, 4, 4800, device=dev)
    o = LogmelIV_Extractor(CFG).to(dev)(z).cpu()
    zr = of.logmel_iv(torch.zeros(1, 4, 4800))
    assert (o[:, :4] - zr[:, :4]).abs().max().item() < 1e-4 and (o[:, :4] + 100.0).abs().max().item() < 1e-4
    assert torch.all(o[:, 4:] == 0.0)
    with pytest.raises(ValueError):
        ext(torch.zeros(4, 4800, device=dev))


@pytest.mark.gpu
def test_chunked_equals_whole_clip_chunks(dev):
    """A 60 s clip is 6 independent 10 s chunks: feature(chunked) has 6x1001 frames."""
    from pseldnets_amd.utils.feature import LogmelIV_Extractor
    x = _wave(1, 4, 48000, seed=9)
    ext = LogmelIV_Extractor(CFG).to(dev)
    chunks = x.reshape(1, 4, 2, 24000).permute(0, 2, 1, 3).reshape(2, 4, 24000).contiguous()
    out = ext(chunks.to(dev)).cpu()
    ref = of.logmel_iv(chunks)
    assert (out[:, :4] - ref[:, :4]).abs().max().item() < 2e-3


@pytest.mark.gpu
def test_near_silent_bins_are_bounded_by_fp32_round_off(dev):
    """The closed-form test wave leaves mel bins at the -100 dB floor, where fp32 round-off alone moves the dB value by
    a few hundredths and the per-bin normalised intensity vectors of near-silent FFT bins are pure round-off (the fp32
    oracle differs from the float64 evaluation by as much): the kernel must be as close to float64 as the fp32 oracle
    is, and exact as log-mel power."""
    from oracle import synth
    from pseldnets_amd.utils.feature import LogmelIV_Extractor
    wave = synth.formula_wave(1, 4, 240000)
    out = LogmelIV_Extractor(CFG).to(dev)(wave.to(dev)).cpu().double()
    ref32 = of.logmel_iv(wave).double()
    ref64 = torch.from_numpy(of.logmel_iv_f64(wave.numpy())).double()
    noise = (ref32 - ref64).abs().max().item()
    assert (out - ref64).abs().max().item() <= 2 * noise + 1e-3
    p, pr = 10.0 ** (out[:, :4] / 10), 10.0 ** (ref64[:, :4] / 10)
    assert ((p - pr).abs() <= 2e-3 * pr + 1e-9).all()


def test_oracle_stft_matches_scipy_a_third_implementation():
    """torchaudio's Spectrogram is not in /root/reference (a pinned pip dependency), so the STFT convention the oracle restates
    (centred frames, reflect padding, periodic Hann, one-sided, no normalisation) is checked against a third, unrelated
    implementation: scipy.signal.stft on the reflect-padded signal, with scipy's 1/sum(window) scaling undone."""
    import scipy.signal as ss
    x = _wave(1, 2, 7200, seed=3)
    n_fft, hop = 1024, 240
    win_t = torch.hann_window(n_fft, periodic=True)
    spec = of.spectrogram_complex(x, n_fft, hop, win_t).numpy()                 # [1, 2, 513, T]
    xp = np.pad(x.numpy().astype(np.float64), ((0, 0), (0, 0), (n_fft // 2, n_fft // 2)), mode='reflect')
    win = ss.get_window('hann', n_fft, fftbins=True)                            # periodic
    f, t, Z = ss.stft(xp, fs=24000, window=win, nperseg=n_fft, noverlap=n_fft - hop, nfft=n_fft, boundary=None, padded=False,
                      return_onesided=True, detrend=False)
    Z = Z * win.sum()                                                           # scipy divides by sum(window)
    assert Z.shape[-2] == 513 and Z.shape[-1] == spec.shape[-1] == 1 + 7200 // hop
    scale = np.abs(Z).max()
    assert np.abs(spec - Z).max() < 2e-5 * scale
    assert np.allclose(win, win_t.numpy(), atol=1e-7)


def test_oracle_mel_filterbank_properties():
    """melscale_fbanks(mel_scale='htk', norm='slaney') as torchaudio documents it and as the reference asks for it
    (utils/feature.py:32-34): 64 triangles over 513 bins whose corner frequencies are equally spaced on the HTK mel scale
    mel = 2595 log10(1 + f / 700) between 0 and sample_rate / 2, rising and falling linearly IN HERTZ, each divided by half its base
    width (Slaney's area normalisation: 2 / (f[m + 2] - f[m])); at most two filters cover a bin."""
    fb = of.melscale_fbanks(513, 0.0, 12000.0, 64, 24000).numpy().astype(np.float64)      # [513, 64]
    assert fb.shape == (513, 64) and fb.min() >= 0.0
    mel = lambda f: 2595.0 * np.log10(1.0 + f / 700.0)
    imel = lambda m: 700.0 * (10.0 ** (m / 2595.0) - 1.0)
    corners = imel(np.linspace(mel(0.0), mel(12000.0), 66))
    freqs = np.linspace(0.0, 12000.0, 513)
    for m in (0, 1, 17, 40, 63):
        lo, c, hi = corners[m], corners[m + 1], corners[m + 2]
        tri = np.maximum(0.0, np.minimum((freqs - lo) / (c - lo), (hi - freqs) / (hi - c))) * 2.0 / (hi - lo)
        assert np.abs(fb[:, m] - tri).max() < 2e-5 * tri.max(), m
    assert ((fb > 0).sum(1) <= 2).all()
