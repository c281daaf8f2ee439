"""FLAC recordings -> PCM arrays (the reference: `sf.read(path, dtype='float32', start, stop)[0].T`, data/data.py:9-13; its synthetic
datasets are stored as FLAC, data/components/data.py:81). The decoding is the C++ library `libpseld_host.so` (csrc/host/flac.cpp,
include/pseld_host.h: own code - soundfile / libFLAC are not in this image; PARITY UNPINNED, see the header); this wrapper adds the last
of the format's self-checks: the MD5 signature of the unencoded audio that the encoder stored in STREAMINFO is recomputed over the
decoded samples and must match."""
import ctypes
import hashlib
import os

import numpy as np

LIB_PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'libpseld_host.so')
_lib = None


class FlacError(ValueError):
    pass


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise FlacError(f'{LIB_PATH} not found: build it first (python -c "import __graft_entry__ as g; g.build()")')
        L = ctypes.CDLL(LIB_PATH)
        L.pseld_host_last_error.restype = ctypes.c_char_p
        L.pseld_flac_info.restype = ctypes.c_int
        L.pseld_flac_info.argtypes = [ctypes.c_char_p, ctypes.c_long] + [ctypes.c_void_p] * 5
        L.pseld_flac_decode.restype = ctypes.c_long
        L.pseld_flac_decode.argtypes = [ctypes.c_char_p, ctypes.c_long, ctypes.c_void_p, ctypes.c_long]
        _lib = L
    return _lib


def flac_info(data):
    """bytes -> dict(sample_rate, channels, bits_per_sample, total_samples, md5)."""
    L = lib()
    sr, ch, bps, tot = ctypes.c_int(), ctypes.c_int(), ctypes.c_int(), ctypes.c_long()
    md5 = (ctypes.c_uint8 * 16)()
    if L.pseld_flac_info(data, len(data), ctypes.byref(sr), ctypes.byref(ch), ctypes.byref(bps), ctypes.byref(tot), md5) != 0:
        raise FlacError(L.pseld_host_last_error().decode())
    return dict(sample_rate=sr.value, channels=ch.value, bits_per_sample=bps.value, total_samples=tot.value, md5=bytes(md5))


def decode_flac(data, verify_md5=True):
    """bytes of a FLAC stream -> (int32 array [samples, channels], info). Frame CRCs are checked by the library, the MD5 signature here."""
    L = lib()
    info = flac_info(data)
    # STREAMINFO may leave the length open (total_samples = 0: a streamed encode). The buffer then starts at what an ordinary compression
    # ratio gives (~1 byte per sample and channel) and grows geometrically while the library reports it too small - never the len(data) * 8
    # samples x channels of int32 a one-shot worst case would take (tens of GB for an ordinary file)
    cap = info['total_samples'] if info['total_samples'] > 0 else max(4096, len(data) // max(1, info['channels']))
    while True:
        out = np.empty((cap, info['channels']), dtype=np.int32)
        n = L.pseld_flac_decode(data, len(data), out.ctypes.data_as(ctypes.c_void_p), cap)
        if n >= 0:
            break
        msg = L.pseld_host_last_error().decode()
        if info['total_samples'] > 0 or not msg.startswith('output buffer too small') or cap >= len(data) * 8:
            raise FlacError(msg)
        cap = min(cap * 4, len(data) * 8)
    out = out[:n]
    if verify_md5 and info['md5'] != bytes(16):
        nbytes = (info['bits_per_sample'] + 7) // 8
        if nbytes == 2:
            raw = out.astype('<i2').tobytes()
        elif nbytes == 4:
            raw = out.astype('<i4').tobytes()
        elif nbytes == 1:
            raw = out.astype('i1').tobytes()
        else:                                          # 3 bytes per sample: the low three bytes of each little-endian int32
            raw = out.astype('<i4').view(np.uint8).reshape(-1, 4)[:, :3].tobytes()
        if hashlib.md5(raw).digest() != info['md5']:
            raise FlacError('decoded audio does not match the MD5 signature in STREAMINFO')
    return out, info


def read_flac(path, dtype='int'):
    """path -> (array [samples, channels], sample_rate). dtype 'int': the stored sample values (int32); 'float32': scaled by
    2^-(bits-1) as soundfile's `dtype='float32'` does."""
    with open(path, 'rb') as f:
        data = f.read()
    pcm, info = decode_flac(data)
    if dtype == 'float32':
        return pcm.astype(np.float32) / np.float32(1 << (info['bits_per_sample'] - 1)), info['sample_rate']
    return pcm, info['sample_rate']
