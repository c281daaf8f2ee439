"""A small FLAC ENCODER for the tests of pseldnets_amd/data/flac.py (test infrastructure, never shipped): written from the format
description independently of the decoder's code paths (bit WRITER, bitwise CRCs, hashlib MD5), so that every syntax element the decoder
reads is produced by other code than the code that reads it: CONSTANT / VERBATIM / FIXED / LPC subframes, wasted bits, Rice / Rice2
partitions incl. escapes, the four channel assignments, all block size / sample size codes, UTF-8-like frame numbers, fixed and
variable block size streams, extra metadata blocks and an ID3v2 tag. No FLAC encoder or file exists in the image - see the decoder's header
for what that means ("parity unpinned"; the format's CRCs and MD5 are what a real file brings along)."""
import hashlib

import numpy as np


class BitWriter:
    def __init__(self):
        self.acc, self.n, self.out = 0, 0, bytearray()

    def put(self, v, k):
        if k == 0:
            return
        v &= (1 << k) - 1
        self.acc = (self.acc << k) | v
        self.n += k
        while self.n >= 8:
            self.n -= 8
            self.out.append((self.acc >> self.n) & 0xFF)
        self.acc &= (1 << self.n) - 1

    def unary(self, q):
        while q >= 32:
            self.put(0, 32); q -= 32
        self.put(1, q + 1)

    def align(self):
        if self.n:
            self.put(0, 8 - self.n)

    def bytes(self):
        assert self.n == 0
        return bytes(self.out)


def crc8(data):
    c = 0
    for b in data:
        c ^= b
        for _ in range(8):
            c = ((c << 1) ^ 0x07) & 0xFF if c & 0x80 else (c << 1) & 0xFF
    return c


def crc16(data):
    c = 0
    for b in data:
        c ^= b << 8
        for _ in range(8):
            c = ((c << 1) ^ 0x8005) & 0xFFFF if c & 0x8000 else (c << 1) & 0xFFFF
    return c


def utf8_number(v):
    if v < 0x80:
        return bytes([v])
    n = 2
    while v >= 1 << (5 * n + 1):                      # n bytes carry 5 n + 1 bits (7 - n in the lead, 6 per continuation byte)
        n += 1
    out = [((0xFF << (8 - n)) & 0xFF) | (v >> (6 * (n - 1)))]
    for i in range(n - 2, -1, -1):
        out.append(0x80 | ((v >> (6 * i)) & 0x3F))
    return bytes(out)


def _zigzag(e):
    return 2 * e if e >= 0 else -2 * e - 1


def _residual(w, res, blocksize, order, porder, rice2=False, escape_parts=()):
    w.put(1 if rice2 else 0, 2)
    w.put(porder, 4)
    pbits, esc = (5, 31) if rice2 else (4, 15)
    i = 0
    for p in range(1 << porder):
        cnt = (blocksize >> porder) - (order if p == 0 else 0)
        part = res[i:i + cnt]
        i += cnt
        if p in escape_parts:
            nb = max([1] + [int(abs(int(e))).bit_length() + 1 for e in part])
            w.put(esc, pbits); w.put(nb, 5)
            for e in part:
                w.put(int(e), nb)
            continue
        us = [_zigzag(int(e)) for e in part]
        best_k, best = 0, None
        for k in range(0, esc):
            size = sum((u >> k) + 1 + k for u in us)
            if best is None or size < best:
                best_k, best = k, size
        w.put(best_k, pbits)
        for u in us:
            w.unary(u >> best_k)
            w.put(u, best_k)


def _subframe(w, s, bps, kind, porder=0, rice2=False, escape_parts=(), lpc=None):
    """s: list of ints (one channel of one block). kind: 'constant' | 'verbatim' | ('fixed', order) | ('lpc', order, precision, shift)."""
    s = [int(v) for v in s]
    n = len(s)
    wasted = 0
    if any(s) and kind != 'constant':
        while all((v >> wasted) & 1 == 0 for v in s):
            wasted += 1
    if wasted:
        s = [v >> wasted for v in s]
    bps -= wasted
    w.put(0, 1)
    if kind == 'constant':
        assert len(set(s)) == 1
        w.put(0, 6); w.put(0, 1); w.put(s[0], bps)
        return
    if kind == 'verbatim':
        w.put(1, 6)
    elif kind[0] == 'fixed':
        w.put(8 + kind[1], 6)
    else:
        w.put(32 + kind[1] - 1, 6)
    if wasted:
        w.put(1, 1); w.unary(wasted - 1)
    else:
        w.put(0, 1)
    if kind == 'verbatim':
        for v in s:
            w.put(v, bps)
        return
    order = kind[1]
    for v in s[:order]:
        w.put(v, bps)
    if kind[0] == 'fixed':
        C = {0: [], 1: [1], 2: [2, -1], 3: [3, -3, 1], 4: [4, -6, 4, -1]}[order]
        res = [s[i] - sum(c * s[i - 1 - j] for j, c in enumerate(C)) for i in range(order, n)]
    else:
        _, order, prec, shift = kind
        coef = lpc
        w.put(prec - 1, 4); w.put(shift, 5)
        for c in coef:
            w.put(int(c), prec)
        res = [s[i] - (sum(int(c) * s[i - 1 - j] for j, c in enumerate(coef)) >> shift) for i in range(order, n)]
    _residual(w, res, n, order, porder, rice2, escape_parts)


def lpc_coefficients(s, order, prec, shift):
    """Least-squares predictor of the block, quantised to `prec`-bit integers at 2^-shift."""
    x = np.asarray(s, np.float64)
    A = np.stack([x[order - 1 - j:len(x) - 1 - j] for j in range(order)], 1)
    c = np.linalg.lstsq(A, x[order:], rcond=None)[0]
    q = np.round(c * (1 << shift)).astype(np.int64)
    lim = (1 << (prec - 1)) - 1
    return np.clip(q, -lim - 1, lim)


BS_CODES = {192: 1, 576: 2, 1152: 3, 2304: 4, 4608: 5, 256: 8, 512: 9, 1024: 10, 2048: 11, 4096: 12, 8192: 13, 16384: 14, 32768: 15}
SS_CODES = {8: 1, 12: 2, 16: 4, 20: 5, 24: 6, 32: 7}
SR_CODES = {88200: 1, 176400: 2, 192000: 3, 8000: 4, 16000: 5, 22050: 6, 24000: 7, 32000: 8, 44100: 9, 48000: 10, 96000: 11}


def encode(pcm, rate=24000, bps=16, blocksize=4096, kinds=('fixed2',), porder=2, stereo='indep', rice2=False, escape_parts=(), variable=False,
           explicit_bs=False, rate_in_header='code', first_frame_number=0, extra_metadata=False, id3=False, md5=True, streaminfo_ss=True):
    """pcm: int array [samples, channels]. kinds: per channel (cycled): 'constant' | 'verbatim' | 'fixedK' | 'lpcK'. Returns bytes."""
    pcm = np.asarray(pcm)
    n, ch = pcm.shape
    frames = []
    pos, fno = 0, first_frame_number
    while pos < n:
        bs = min(blocksize, n - pos)
        blk = pcm[pos:pos + bs].astype(np.int64)
        w = BitWriter()
        w.put(0x3FFE, 14); w.put(0, 1); w.put(1 if variable else 0, 1)
        if bs in BS_CODES and not explicit_bs:
            bsc = BS_CODES[bs]
        else:
            bsc = 6 if bs <= 256 else 7
        w.put(bsc, 4)
        if rate_in_header == 'code' and rate in SR_CODES:
            src = SR_CODES[rate]
        elif rate_in_header == 'streaminfo':
            src = 0
        elif rate % 1000 == 0 and rate // 1000 < 256 and rate_in_header == 'khz':
            src = 12
        elif rate_in_header == 'tens' and rate % 10 == 0:
            src = 14
        else:
            src = 13
        w.put(src, 4)
        if stereo == 'indep':
            w.put(ch - 1, 4)
        else:
            assert ch == 2
            w.put({'ls': 8, 'rs': 9, 'ms': 10}[stereo], 4)
        w.put(0 if (streaminfo_ss or bps not in SS_CODES) else SS_CODES[bps], 3)
        w.put(0, 1)
        for b in utf8_number(pos if variable else fno):
            w.put(b, 8)
        if bsc == 6:
            w.put(bs - 1, 8)
        elif bsc == 7:
            w.put(bs - 1, 16)
        if src == 12:
            w.put(rate // 1000, 8)
        elif src == 13:
            w.put(rate, 16)
        elif src == 14:
            w.put(rate // 10, 16)
        w.align()
        w.put(crc8(w.bytes()), 8)
        chans = [blk[:, c] for c in range(ch)]
        widths = [bps] * ch
        if stereo == 'ls':
            chans, widths = [blk[:, 0], blk[:, 0] - blk[:, 1]], [bps, bps + 1]
        elif stereo == 'rs':
            chans, widths = [blk[:, 0] - blk[:, 1], blk[:, 1]], [bps + 1, bps]
        elif stereo == 'ms':
            chans, widths = [(blk[:, 0] + blk[:, 1]) >> 1, blk[:, 0] - blk[:, 1]], [bps, bps + 1]
        for c, (s, wd) in enumerate(zip(chans, widths)):
            kind = kinds[c % len(kinds)]
            po = porder
            while po > 0 and ((bs >> po) << po != bs or (bs >> po) <= 32):
                po -= 1
            if kind == 'constant':
                _subframe(w, s, wd, 'constant')
            elif kind == 'verbatim' or bs <= 32:
                _subframe(w, s, wd, 'verbatim')
            elif kind.startswith('fixed'):
                _subframe(w, s, wd, ('fixed', int(kind[5:])), po, rice2, escape_parts)
            else:
                import re
                m = re.fullmatch(r'lpc(\d+)(?:p(\d+))?(?:s(\d+))?', kind)
                order, prec, shift = int(m.group(1)), int(m.group(2) or 12), int(m.group(3) or 9)
                sv = [int(v) for v in s]
                wz = 0
                if any(sv):
                    while all((v >> wz) & 1 == 0 for v in sv):
                        wz += 1
                coef = lpc_coefficients([v >> wz for v in sv], order, prec, shift)
                _subframe(w, s, wd, ('lpc', order, prec, shift), po, rice2, escape_parts, lpc=coef)
        w.align()
        body = w.bytes()
        frames.append(body + crc16(body).to_bytes(2, 'big'))
        pos += bs
        fno += 1
    nbytes = (bps + 7) // 8
    if nbytes == 3:
        raw = pcm.astype('<i4').view(np.uint8).reshape(-1, 4)[:, :3].tobytes()
    else:
        raw = pcm.astype({1: 'i1', 2: '<i2', 4: '<i4'}[nbytes]).tobytes()
    digest = hashlib.md5(raw).digest() if md5 else bytes(16)
    si = BitWriter()
    si.put(blocksize if n >= blocksize else max(n, 16), 16); si.put(blocksize, 16)
    si.put(min(len(f) for f in frames), 24); si.put(max(len(f) for f in frames), 24)
    si.put(rate, 20); si.put(ch - 1, 3); si.put(bps - 1, 5); si.put(n, 36)
    streaminfo = si.bytes() + digest
    blocks = [(0, streaminfo)]
    if extra_metadata:
        blocks += [(1, bytes(37)), (4, (4).to_bytes(4, 'little') + b'test' + (0).to_bytes(4, 'little'))]
    out = bytearray()
    if id3:
        out += b'ID3\x04\x00\x00' + bytes([0, 0, 0, 20]) + bytes(20)
    out += b'fLaC'
    for i, (t, body) in enumerate(blocks):
        out += bytes([(0x80 if i == len(blocks) - 1 else 0) | t]) + len(body).to_bytes(3, 'big') + body
    for f in frames:
        out += f
    return bytes(out)
