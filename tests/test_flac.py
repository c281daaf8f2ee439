"""pseldnets_amd/data/flac.py + csrc/host/flac.cpp (the decoder standing in for `sf.read` on the reference's FLAC recordings,
data/data.py:9-13, data/components/data.py:81) against streams built by the independent test encoder tests/flac_testenc.py: every
subframe type, residual coding, channel assignment and header form must decode to the PCM that went in; the format's own checks (header
CRC-8, frame CRC-16, MD5 signature) must reject damaged streams. PARITY UNPINNED (no FLAC file / encoder in the image): see the header."""
import os

import numpy as np
import pytest

from tests import flac_testenc as E


def _F():
    from pseldnets_amd.data import flac
    if not os.path.exists(flac.LIB_PATH):
        pytest.skip('libpseld_host.so not built')
    return flac


def _audio(n, ch, bps, seed, tone=True):
    rng = np.random.default_rng(seed)
    t = np.arange(n)[:, None]
    amp = (1 << (bps - 1)) * 0.3
    x = amp * np.sin(2 * np.pi * (0.01 + 0.007 * np.arange(ch)[None]) * t + rng.uniform(0, 6, (1, ch))) if tone else 0
    x = x + rng.standard_normal((n, ch)) * amp * 0.02
    lim = (1 << (bps - 1)) - 1
    return np.clip(np.round(x), -lim - 1, lim).astype(np.int64)


def test_library_exports_what_the_header_declares():
    F = _F()
    import re
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'include', 'pseld_host.h')).read()
    names = re.findall(r'\b(pseld_\w+)\s*\(', hdr)
    assert {'pseld_flac_info', 'pseld_flac_decode', 'pseld_host_last_error'} <= set(names)
    for n in names:
        assert hasattr(F.lib(), n), n


@pytest.mark.parametrize("kinds,porder", [(('fixed0', 'fixed1', 'fixed2', 'fixed3'), 0), (('fixed4', 'fixed2'), 3), (('verbatim', 'fixed2', 'constant', 'fixed1'), 2),
                                          (('lpc1', 'lpc8p12s9', 'lpc12p15s12', 'lpc32p14s10'), 2)])
def test_four_channel_16_bit_streams_every_subframe_type(kinds, porder):
    F = _F()
    pcm = _audio(4096 * 2 + 1000, 4, 16, 1)
    if 'constant' in kinds:
        pcm[:, kinds.index('constant')] = -1234
    data = E.encode(pcm, 24000, 16, 4096, kinds, porder)
    got, info = F.decode_flac(data)
    assert info['sample_rate'] == 24000 and info['channels'] == 4 and info['bits_per_sample'] == 16 and info['total_samples'] == len(pcm)
    assert got.dtype == np.int32 and np.array_equal(got, pcm)


@pytest.mark.parametrize("stereo", ['ls', 'rs', 'ms', 'indep'])
@pytest.mark.parametrize("bps", [16, 24, 8])
def test_stereo_decorrelation_and_sample_sizes(stereo, bps):
    F = _F()
    pcm = _audio(3000, 2, bps, 2 + bps)
    pcm[:, 1] = np.clip(pcm[:, 0] + _audio(3000, 1, max(bps - 4, 4), 9)[:, 0], -(1 << (bps - 1)), (1 << (bps - 1)) - 1)      # correlated pair, odd sums
    data = E.encode(pcm, 44100, bps, 1152, ('fixed2', 'lpc6p12s9'), 1, stereo=stereo, streaminfo_ss=(bps == 8))
    got, info = F.decode_flac(data)
    assert info['bits_per_sample'] == bps and np.array_equal(got, pcm)


def test_wasted_bits_rice2_escapes_and_header_forms():
    F = _F()
    pcm = _audio(5000, 2, 16, 3)
    pcm[:, 0] = (pcm[:, 0] >> 3) << 3                                    # three wasted bits in channel 0
    for kw in (dict(rice2=True), dict(escape_parts=(0, 2)), dict(explicit_bs=True), dict(rate_in_header='streaminfo'), dict(rate_in_header='hz'),
               dict(rate_in_header='khz'), dict(rate_in_header='tens'), dict(variable=True), dict(extra_metadata=True, id3=True), dict(md5=False),
               dict(first_frame_number=5_000_000)):
        data = E.encode(pcm, 24000, 16, 1024, ('fixed3', 'lpc4p10s8'), 2, **kw)
        got, info = F.decode_flac(data)
        assert np.array_equal(got, pcm), kw
    noisy = (np.random.default_rng(4).integers(-30000, 30000, (2048, 1))).astype(np.int64)          # residuals that need Rice parameters > 14
    data = E.encode(noisy, 16000, 16, 2048, ('fixed0',), 0, rice2=True)
    assert np.array_equal(F.decode_flac(data)[0], noisy)
    tiny = _audio(16 * 300, 1, 16, 5)                                    # 300 frames of 16 samples: frame numbers past 127 (two-byte coding)
    assert np.array_equal(F.decode_flac(E.encode(tiny, 8000, 16, 16, ('verbatim',), 0))[0], tiny)


def test_damaged_streams_are_errors(tmp_path):
    F = _F()
    pcm = _audio(4096 + 500, 4, 16, 6)
    data = bytearray(E.encode(pcm, 24000, 16, 4096, ('fixed2',), 2))
    good = bytes(data)
    body = good.index(b'\xff\xf8', 42)
    for off, what in ((body + 40, 'CRC-16'), (body + 2, 'CRC-8|sync|reserved|block size'), (len(good) - 30, 'CRC-16')):
        bad = bytearray(good); bad[off] ^= 0x10
        with pytest.raises(F.FlacError, match=what):
            F.decode_flac(bytes(bad))
    bad = bytearray(good); bad[30] ^= 0xFF                               # a byte of the MD5 signature
    with pytest.raises(F.FlacError, match='MD5'):
        F.decode_flac(bytes(bad))
    with pytest.raises(F.FlacError):
        F.decode_flac(good[:len(good) // 2])
    with pytest.raises(F.FlacError, match='fLaC'):
        F.decode_flac(b'RIFF' + bytes(100))
    p = tmp_path / 'a.flac'
    p.write_bytes(good)
    x, sr = F.read_flac(str(p), dtype='float32')
    assert sr == 24000 and x.dtype == np.float32 and np.array_equal(x, (pcm / 32768.0).astype(np.float32))          # sf.read(dtype='float32') scaling


def test_unknown_length_streams_decode_without_a_worst_case_buffer():
    """STREAMINFO with total_samples = 0 (a streamed encode): the decoder's output buffer starts small and grows (it used to be
    len(data) * 8 samples x channels of int32); the samples and the MD5 check are those of the stream with the length filled in."""
    F = _F()
    pcm = _audio(40000, 4, 16, 7)
    data = bytearray(E.encode(pcm, 24000, 16, 4096, ('fixed2', 'lpc8p12s9'), 2))
    data[21] &= 0xF0                              # the 36-bit total-samples field: low nibble of byte 21 + bytes 22-25
    data[22:26] = bytes(4)
    got, info = F.decode_flac(bytes(data))
    assert info['total_samples'] == 0 and np.array_equal(got, pcm)
