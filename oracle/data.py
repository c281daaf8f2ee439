"""TEST INFRASTRUCTURE ONLY — numpy restatement of the chunk loading and label synthesis of the reference's datasets
(/root/reference/src/data/data.py:7-15,17-59,75-77,87-93,198-213). Imported only by tests/. soundfile / h5py are absent from this
image: `load_chunk` restates soundfile's documented PCM16 -> float32 read (sample / 32768) — "parity unpinned" for that
third-party step; segment_index is pinned by tests/golden/data.npz and generate_spatial_samples by tests/golden/spatial.npz (the reference's own
functions)."""
import numpy as np


def load_chunk(pcm, begin, end, pad_before, pad_after):
    """sf.read(path, dtype='float32', start=begin, stop=end)[0].T, then np.pad(x, ((0, 0), (pad_before, pad_after)))."""
    x = (pcm[begin:end].astype(np.float32) / np.float32(32768.0)).T
    return np.pad(x, ((0, 0), (pad_before, pad_after)), mode='constant')


def adpit_label(se, azi, ele):
    """data.py:207-213: se / azi / ele [T, 6, C] -> [T, 6, 4, C] float32."""
    se, azi, ele = se.astype(np.float32), azi.astype(np.float32), ele.astype(np.float32)
    lx = np.cos(np.deg2rad(azi)) * np.cos(np.deg2rad(ele)) * se
    ly = np.sin(np.deg2rad(azi)) * np.cos(np.deg2rad(ele)) * se
    lz = np.sin(np.deg2rad(ele)) * se
    return np.stack((se, lx, ly, lz), axis=2, dtype=np.float32)


def accdoa_label(se, azi, ele):
    """data.py:87-93: [T, C] each -> [T, 4C] float32 (se | x | y | z)."""
    se, azi, ele = se.astype(np.float32), azi.astype(np.float32), ele.astype(np.float32)
    lx = np.cos(np.deg2rad(azi)) * np.cos(np.deg2rad(ele)) * se
    ly = np.sin(np.deg2rad(azi)) * np.cos(np.deg2rad(ele)) * se
    lz = np.sin(np.deg2rad(ele)) * se
    return np.concatenate((se, lx, ly, lz), axis=1, dtype=np.float32)


def generate_spatial_samples(audio, method, rng=np.random, **kw):
    """data/data.py:17-59 (mono_adapter recipe, single-source targets): one mono clip -> FOA at a random direction, label rewritten.
    audio [L] or [ch, L] (channel 0 is used). Products are float64 scalar x float32 array = float64 (NEP 50), as in the reference."""
    if audio.ndim == 2:
        audio = audio[0]
    azi = rng.randint(-180, 180)
    ele = rng.randint(-90, 90)
    x = np.cos(np.deg2rad(azi)) * np.cos(np.deg2rad(ele))
    y = np.sin(np.deg2rad(azi)) * np.cos(np.deg2rad(ele))
    z = np.sin(np.deg2rad(ele))
    foa = np.stack((audio, y * audio, z * audio, x * audio), axis=0)
    if method == 'einv2':
        sed, doa = kw['sed_label'], np.zeros_like(kw['doa_label'])
        act = sed.sum(axis=(-1, -2))
        doa[..., 0, 0], doa[..., 0, 1], doa[..., 0, 2] = act * x, act * y, act * z
        return foa, sed, doa
    if method == 'accdoa':
        lab = kw['accdoa_label']
        C = lab.shape[-1] // 4
        se, out = lab[:, :C], np.zeros_like(lab)
        out[..., C:2 * C], out[..., 2 * C:3 * C], out[..., 3 * C:] = x * se, y * se, z * se
        return foa, out
    if method == 'multi_accdoa':
        lab = kw['adpit_label']
        se, out = lab[:, :, 0, :], np.zeros_like(lab)
        out[:, :, 0, :], out[:, :, 1, :], out[:, :, 2, :], out[:, :, 3, :] = se, x * se, y * se, z * se
        return foa, out
    raise NotImplementedError(method)
