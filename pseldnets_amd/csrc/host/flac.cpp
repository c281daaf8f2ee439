// FLAC stream decoder (host side, plain C++): the reference reads its synthetic datasets' recordings as FLAC through soundfile / libsndfile
// (data/components/data.py:81 swaps '.wav' for '.flac'; data/data.py:9-13 `sf.read(path, dtype='float32', start, stop)`), neither of which
// is in this image - and no FLAC file or encoder either: PARITY UNPINNED. What stands in for a pin: the format carries its own checks, and
// this decoder enforces all of them - CRC-8 of every frame header, CRC-16 of every frame, and (in the Python wrapper) the MD5 signature of
// the decoded audio that the ENCODER stored in STREAMINFO. A misread bit stream is an error, never silently wrong samples.
// Covered: the whole "subset" and non-subset stream syntax of FLAC 1.x for 4-32 bit samples, 1-8 channels: CONSTANT / VERBATIM / FIXED
// (orders 0-4) / LPC (orders 1-32) subframes, wasted bits, Rice and Rice2 partitioned residuals with escape partitions, independent /
// left-side / right-side / mid-side channel assignments, fixed and variable block size streams, metadata blocks skipped (an ID3v2 tag in
// front too). Not covered: Ogg encapsulation.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <cstdarg>
#include <vector>

namespace {
thread_local char g_err[256] = "";
int fail(const char* fmt, ...) {
    va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof g_err, fmt, ap); va_end(ap);
    return -1;
}

struct Bits {
    const uint8_t* p; long n; long pos = 0;          // pos in BITS
    bool bad = false;
    Bits(const uint8_t* d, long len) : p(d), n(len) {}
    uint64_t get(int k) {                             // k <= 57
        if (k == 0) return 0;
        if (pos + k > n * 8) { bad = true; pos = n * 8; return 0; }
        uint64_t v = 0;
        long byte = pos >> 3; int off = (int)(pos & 7), need = k + off, got = 0;
        while (got < need) { v = (v << 8) | p[byte++]; got += 8; }
        v >>= (got - need);
        pos += k;
        return v & ((k == 64) ? ~0ull : ((1ull << k) - 1));
    }
    int64_t gets(int k) {                             // signed, two's complement
        if (k == 0) return 0;
        const uint64_t v = get(k);
        return (int64_t)(v << (64 - k)) >> (64 - k);
    }
    uint32_t unary() {                                // number of 0 bits in front of the next 1
        uint32_t q = 0;
        while (true) {
            if (pos >= n * 8) { bad = true; return q; }
            const long byte = pos >> 3; const int off = (int)(pos & 7);
            const uint8_t b = (uint8_t)(p[byte] << off);              // remaining bits of this byte, MSB first
            if (b) { const int z = __builtin_clz((unsigned)b) - 24; pos += z + 1; return q + z; }
            q += 8 - off; pos += 8 - off;
        }
    }
    void align() { pos = (pos + 7) & ~7L; }
};

uint8_t crc8(const uint8_t* d, long n) {
    uint8_t c = 0;
    for (long i = 0; i < n; ++i) { c ^= d[i]; for (int b = 0; b < 8; ++b) c = (uint8_t)((c & 0x80) ? (c << 1) ^ 0x07 : (c << 1)); }
    return c;
}
uint16_t crc16(const uint8_t* d, long n) {
    static uint16_t tab[256]; static bool init = false;
    if (!init) {
        for (int i = 0; i < 256; ++i) { uint16_t c = (uint16_t)(i << 8); for (int b = 0; b < 8; ++b) c = (uint16_t)((c & 0x8000) ? (c << 1) ^ 0x8005 : (c << 1)); tab[i] = c; }
        init = true;
    }
    uint16_t c = 0;
    for (long i = 0; i < n; ++i) c = (uint16_t)((c << 8) ^ tab[(c >> 8) ^ d[i]]);
    return c;
}

struct Info { int rate = 0, channels = 0, bps = 0, min_block = 0, max_block = 0; long total = 0; uint8_t md5[16] = {0}; long audio_off = 0; };

int parse_header(const uint8_t* d, long n, Info& s) {
    long p = 0;
    if (n >= 10 && d[0] == 'I' && d[1] == 'D' && d[2] == '3')                       // an ID3v2 tag in front of the stream
        p = 10 + (((long)(d[6] & 0x7f) << 21) | ((long)(d[7] & 0x7f) << 14) | ((long)(d[8] & 0x7f) << 7) | (d[9] & 0x7f));
    if (p + 4 > n || memcmp(d + p, "fLaC", 4) != 0) return fail("not a FLAC stream (no fLaC marker)");
    p += 4;
    bool have = false;
    while (true) {
        if (p + 4 > n) return fail("truncated metadata");
        const bool last = d[p] & 0x80; const int type = d[p] & 0x7f;
        const long len = ((long)d[p + 1] << 16) | ((long)d[p + 2] << 8) | d[p + 3];
        p += 4;
        if (p + len > n) return fail("truncated metadata block");
        if (type == 0) {
            if (len < 34) return fail("short STREAMINFO");
            const uint8_t* q = d + p;
            s.min_block = (q[0] << 8) | q[1]; s.max_block = (q[2] << 8) | q[3];
            s.rate = (q[10] << 12) | (q[11] << 4) | (q[12] >> 4);
            s.channels = ((q[12] >> 1) & 7) + 1;
            s.bps = (((q[12] & 1) << 4) | (q[13] >> 4)) + 1;
            s.total = ((long)(q[13] & 0x0f) << 32) | ((long)q[14] << 24) | ((long)q[15] << 16) | ((long)q[16] << 8) | q[17];
            memcpy(s.md5, q + 18, 16);
            have = true;
        }
        p += len;
        if (last) break;
    }
    if (!have) return fail("no STREAMINFO block");
    if (s.rate == 0 || s.bps < 4 || s.bps > 32) return fail("bad STREAMINFO (rate %d, %d bits)", s.rate, s.bps);
    s.audio_off = p;
    return 0;
}

int residual(Bits& b, int blocksize, int order, int64_t* r) {
    const int method = (int)b.get(2);
    if (method > 1) return fail("reserved residual coding method");
    const int pbits = method == 0 ? 4 : 5, esc = method == 0 ? 15 : 31;
    const int porder = (int)b.get(4), parts = 1 << porder;
    if ((blocksize >> porder) << porder != blocksize && porder > 0) return fail("block size %d not divisible into %d partitions", blocksize, parts);
    int i = 0;
    for (int p = 0; p < parts; ++p) {
        int cnt = (blocksize >> porder) - (p == 0 ? order : 0);
        if (cnt < 0) return fail("partition shorter than the predictor order");
        const int k = (int)b.get(pbits);
        if (k == esc) {
            const int nb = (int)b.get(5);
            for (int j = 0; j < cnt; ++j) r[i++] = b.gets(nb);
        } else {
            for (int j = 0; j < cnt; ++j) {
                const uint64_t u = ((uint64_t)b.unary() << k) | b.get(k);
                r[i++] = (int64_t)(u >> 1) ^ -(int64_t)(u & 1);
            }
        }
        if (b.bad) return fail("frame data ends inside a residual partition");
    }
    return 0;
}

int subframe(Bits& b, int blocksize, int bps, int64_t* s) {
    if (b.get(1)) return fail("subframe padding bit set");
    const int type = (int)b.get(6);
    int wasted = 0;
    if (b.get(1)) { wasted = (int)b.unary() + 1; }
    bps -= wasted;
    if (bps <= 0) return fail("wasted bits >= sample size");
    if (type == 0) {                                   // CONSTANT
        const int64_t v = b.gets(bps);
        for (int i = 0; i < blocksize; ++i) s[i] = v;
    } else if (type == 1) {                            // VERBATIM
        for (int i = 0; i < blocksize; ++i) s[i] = b.gets(bps);
    } else if (type >= 8 && type <= 12) {              // FIXED, order type - 8
        const int order = type - 8;
        if (order > blocksize) return fail("fixed predictor order > block size");
        for (int i = 0; i < order; ++i) s[i] = b.gets(bps);
        if (residual(b, blocksize, order, s + order) < 0) return -1;
        for (int i = order; i < blocksize; ++i) {
            const int64_t r = s[i];
            switch (order) {
                case 0: s[i] = r; break;
                case 1: s[i] = r + s[i - 1]; break;
                case 2: s[i] = r + 2 * s[i - 1] - s[i - 2]; break;
                case 3: s[i] = r + 3 * s[i - 1] - 3 * s[i - 2] + s[i - 3]; break;
                default: s[i] = r + 4 * s[i - 1] - 6 * s[i - 2] + 4 * s[i - 3] - s[i - 4]; break;
            }
        }
    } else if (type >= 32) {                           // LPC, order (type & 31) + 1
        const int order = (type & 31) + 1;
        if (order > blocksize) return fail("LPC order > block size");
        for (int i = 0; i < order; ++i) s[i] = b.gets(bps);
        const int prec = (int)b.get(4) + 1;
        if (prec == 16) return fail("invalid LPC precision");
        const int shift = (int)b.gets(5);
        if (shift < 0) return fail("negative LPC shift");
        int64_t coef[32];
        for (int j = 0; j < order; ++j) coef[j] = b.gets(prec);
        if (residual(b, blocksize, order, s + order) < 0) return -1;
        for (int i = order; i < blocksize; ++i) {
            int64_t acc = 0;
            for (int j = 0; j < order; ++j) acc += coef[j] * s[i - 1 - j];
            s[i] += acc >> shift;
        }
    } else {
        return fail("reserved subframe type %d", type);
    }
    if (b.bad) return fail("frame data ends inside a subframe");
    if (wasted) for (int i = 0; i < blocksize; ++i) s[i] = (int64_t)((uint64_t)s[i] << wasted);
    return 0;
}
}  // namespace

extern "C" const char* pseld_host_last_error(void) { return g_err; }

// STREAMINFO of a FLAC stream held in memory. md5: the encoder's signature of the unencoded audio (all zero = not stored).
extern "C" int pseld_flac_info(const uint8_t* data, long n, int* sample_rate, int* channels, int* bits_per_sample, long* total_samples, uint8_t* md5) {
    Info s;
    if (!data || n < 42) return fail("flac_info: no data");
    if (parse_header(data, n, s) < 0) return -1;
    if (sample_rate) *sample_rate = s.rate;
    if (channels) *channels = s.channels;
    if (bits_per_sample) *bits_per_sample = s.bps;
    if (total_samples) *total_samples = s.total;
    if (md5) memcpy(md5, s.md5, 16);
    return 0;
}

// Decodes the stream into out[sample][channel] (int32, interleaved; room for `capacity` samples per channel). Returns the number of samples
// per channel decoded, or -1 (pseld_host_last_error). Every frame's header CRC-8 and frame CRC-16 are checked.
extern "C" long pseld_flac_decode(const uint8_t* data, long n, int32_t* out, long capacity) {
    Info s;
    if (!data || !out) return fail("flac_decode: null pointer");
    if (parse_header(data, n, s) < 0) return -1;
    long p = s.audio_off, done = 0;
    std::vector<int64_t> buf;
    while (p < n) {
        if (s.total && done >= s.total) break;
        if (n - p < 6) break;                                          // trailing bytes shorter than any frame
        if (!(data[p] == 0xFF && (data[p + 1] & 0xFE) == 0xF8)) return fail("lost frame sync at byte %ld", p);
        Bits b(data + p, n - p);
        b.get(14); if (b.get(1)) return fail("reserved header bit set");
        b.get(1);                                                      // blocking strategy: only the meaning of the coded number changes
        const int bs_code = (int)b.get(4), sr_code = (int)b.get(4), ch_code = (int)b.get(4), ss_code = (int)b.get(3);
        if (b.get(1)) return fail("reserved header bit set");
        int lead = (int)b.get(8), extra = 0;                           // UTF-8-like coded frame / sample number
        if (lead & 0x80) { while (lead & (0x40 >> extra)) ++extra; ++extra; if (extra < 2 || extra > 7) return fail("bad coded number"); --extra; }
        for (int i = 0; i < extra; ++i) if ((b.get(8) & 0xC0) != 0x80) return fail("bad coded number continuation");
        int blocksize;
        if (bs_code == 0) return fail("reserved block size code");
        else if (bs_code == 1) blocksize = 192;
        else if (bs_code <= 5) blocksize = 576 << (bs_code - 2);
        else if (bs_code == 6) blocksize = (int)b.get(8) + 1;
        else if (bs_code == 7) blocksize = (int)b.get(16) + 1;
        else blocksize = 256 << (bs_code - 8);
        if (sr_code == 12) b.get(8); else if (sr_code == 13 || sr_code == 14) b.get(16); else if (sr_code == 15) return fail("invalid sample rate code");
        static const int SS[8] = {0, 8, 12, -1, 16, 20, 24, 32};
        const int bps = ss_code == 0 ? s.bps : SS[ss_code];
        if (bps < 0) return fail("reserved sample size code");
        const long hdr_bytes = b.pos >> 3;
        const uint8_t c8 = (uint8_t)b.get(8);
        if (b.bad) return fail("truncated frame header");
        if (crc8(data + p, hdr_bytes) != c8) return fail("frame header CRC-8 mismatch at byte %ld", p);
        int nch;
        if (ch_code < 8) nch = ch_code + 1; else if (ch_code <= 10) nch = 2; else return fail("reserved channel assignment");
        if (nch != s.channels) return fail("frame with %d channels in a %d-channel stream", nch, s.channels);
        if (done + blocksize > capacity) return fail("output buffer too small (%ld samples)", capacity);
        buf.resize((size_t)nch * blocksize);
        for (int c = 0; c < nch; ++c) {
            const bool side = (ch_code == 8 && c == 1) || (ch_code == 9 && c == 0) || (ch_code == 10 && c == 1);
            if (subframe(b, blocksize, bps + (side ? 1 : 0), buf.data() + (size_t)c * blocksize) < 0) return -1;
        }
        b.align();
        const long body = b.pos >> 3;
        const uint16_t c16 = (uint16_t)b.get(16);
        if (b.bad) return fail("truncated frame at byte %ld", p);
        if (crc16(data + p, body) != c16) return fail("frame CRC-16 mismatch at byte %ld", p);
        int64_t* c0 = buf.data(); int64_t* c1 = buf.data() + blocksize;
        if (ch_code == 8) { for (int i = 0; i < blocksize; ++i) c1[i] = c0[i] - c1[i]; }                       // left, side -> right = left - side
        else if (ch_code == 9) { for (int i = 0; i < blocksize; ++i) c0[i] = c0[i] + c1[i]; }                  // side, right -> left = side + right
        else if (ch_code == 10) {
            for (int i = 0; i < blocksize; ++i) {
                const int64_t sd = c1[i], m = (int64_t)(((uint64_t)c0[i] << 1) | (uint64_t)(sd & 1));
                c0[i] = (m + sd) >> 1; c1[i] = (m - sd) >> 1;
            }
        }
        int32_t* o = out + done * nch;
        for (int i = 0; i < blocksize; ++i) for (int c = 0; c < nch; ++c) o[(long)i * nch + c] = (int32_t)buf[(size_t)c * blocksize + i];
        done += blocksize;
        p += body + 2;
    }
    if (s.total && done < s.total) return fail("stream ends after %ld of %ld samples", done, s.total);
    return (s.total && done > s.total) ? s.total : done;
}
