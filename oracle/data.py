"""TEST INFRASTRUCTURE ONLY — numpy restatement of the chunk loading and label synthesis of the reference's datasets
(/root/reference/src/data/data.py:7-15,75-77,87-93,198-213). Imported only by tests/. soundfile / h5py are absent from this
image: `load_chunk` restates soundfile's documented PCM16 -> float32 read (sample / 32768) — "parity unpinned" for that
third-party step; segment_index is pinned by tests/golden/data.npz (the reference's own function)."""
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
