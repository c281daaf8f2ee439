// K1: fused framed STFT -> power -> mel -> dB  +  FOA intensity vector -> L2-normalise -> mel.
//
// Replaces (reference, /root/reference/src): utils/feature.py:39-56 (LogmelIV_Extractor.forward),
// utils/feature.py:78-91 (Logmel_Extractor.forward), utils/feature.py:93-117 (intensityvector) and the
// torchaudio 2.2.1 transforms they call (Spectrogram: reflect-padded centred STFT, periodic window,
// onesided; MelScale: spec^T @ fb; AmplitudeToDB('power', top_db=None)).
//
// One wave transforms one frame: the 4 real channels are packed into two complex signals (ch0 + i*ch1, ch2 + i*ch3),
// each transformed by a 1024-point FFT held in the registers of 32 lanes (see "Register FFT" below), then split back
// into the four one-sided spectra. Nothing but the waveform is read from HBM and nothing but the [7, T, n_mels]
// features is written: the complex STFT, the power spectrogram and the three 513-bin intensity maps of the unfused
// reference never exist in memory.
//
// Roofline: the HBM floor (5.634 MB algorithmic bytes per 10 s chunk: 3.84 MB in + 1.794 MB out -> 0.2 ms per 192 chunks) is NOT
// what binds this kernel: it is VALU-issue-bound. rocprofv3 --pmc (profiles/r02_pmc_feature.json): 3 590 vector instructions per
// frame (two 1024-point register FFTs in packed fp32, inter-stage twiddles, spectrum split, compact mel) = 690 M wave-instructions
// per 192 chunks, SQ_ACTIVE_INST_VALU = 81 % of the kernel's wave-cycles (2 waves per SIMD, 4 cycles per wave64 instruction),
// 1.2x the algorithmic bytes fetched. Getting closer to the HBM floor means fewer instructions per frame, not better memory access.
#include "common.h"

namespace {

constexpr int NFFT = 1024;
constexpr int NBIN = NFFT / 2 + 1;  // 513
constexpr int MAX_NNZ = 2048;       // capacity of the compact mel filter bank held in LDS
constexpr int MAX_MELS = 128;

struct FeatArgs {
    const float* wave;   // [B, n_ch, L]
    float* feat;         // [B, n_out, T, n_mels]
    const float* window; // [NFFT]
    const float2* twid;  // [NFFT] exp(-2*pi*i*n/NFFT)
    const int* mel_lo;   // [n_mels] first bin of each filter's support
    const int* mel_cnt;  // [n_mels] number of bins in the support
    const int* mel_off;  // [n_mels] offset of the filter's weights in mel_w
    const float* mel_w;  // [nnz]
    long L;
    int T, hop, n_ch, n_out, n_mels, nnz, with_iv;
    float amin, iv_eps;
};

__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
    return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}

// Register FFT. A 1024-point complex FFT is a 32 x 32 Cooley-Tukey split run by 32 lanes that each hold 32 points:
//   x[t + 32 m] (m = 0..31) -> 32-point DFT over m in registers -> twiddle W_1024^(t k1) -> ONE transpose through LDS
//   -> 32-point DFT over t in registers -> X[k1 + 32 k2].
// A wave is two such FFTs side by side (lanes 0-31: channels 0+i*1, lanes 32-63: channels 2+i*3), i.e. one whole frame,
// and then carries the frame through the spectrum split, the intensity vector and the mel projection on its own: there
// is no workgroup barrier in the frame loop, LDS traffic per FFT is one 8 KB transpose + the spectrum, and the per-bin
// values are stored IN PLACE over the spectrum (17 KB of LDS per wave, 8 waves per CU). Measured on 192 ten-second
// chunks: 1.75 ms, against 2.59 ms for the five-pass radix-4 Stockham FFT in LDS this replaces (bank-conflicted strided
// writes, eight workgroup barriers per frame).
constexpr int EX_LD = 33;                  // transpose row stride (elements): lane k1' reads row k1' conflict-free
constexpr int FFT_LDS = 32 * EX_LD;        // elements per FFT region (also holds the 1024-point spectrum afterwards)
constexpr int WAVES2 = 8;                  // waves (= frames in flight) per workgroup
constexpr int FPB2 = 32;                   // frames per workgroup (the 14 KB of tables are re-read per workgroup: amortised over 4 frames per wave)
constexpr int V2_WAVE_BYTES = 2 * FFT_LDS * 8 + 64;

__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 mul_mi(float2 a) { return make_float2(a.y, -a.x); }      // a * (-i)

// forward 4-point DFT, in place
__device__ __forceinline__ void dft4(float2& x0, float2& x1, float2& x2, float2& x3) {
    const float2 a0 = cadd(x0, x2), a1 = csub(x0, x2), a2 = cadd(x1, x3), a3 = mul_mi(csub(x1, x3));
    x0 = cadd(a0, a2); x1 = cadd(a1, a3); x2 = csub(a0, a2); x3 = csub(a1, a3);
}
// forward 8-point DFT: in a[0..7] (natural), out X[k] (natural) written to o[0..7]
__device__ __forceinline__ void dft8(const float2 (&a)[8], float2 (&o)[8]) {
    float2 e0 = a[0], e1 = a[2], e2 = a[4], e3 = a[6];
    float2 q0 = a[1], q1 = a[3], q2 = a[5], q3 = a[7];
    dft4(e0, e1, e2, e3);
    dft4(q0, q1, q2, q3);
    constexpr float R = 0.70710678118654752f;
    const float2 t1 = make_float2((q1.x + q1.y) * R, (q1.y - q1.x) * R);      // q1 * (1 - i)/sqrt2
    const float2 t2 = mul_mi(q2);                                              // q2 * (-i)
    const float2 t3 = make_float2((q3.y - q3.x) * R, -(q3.x + q3.y) * R);     // q3 * (-1 - i)/sqrt2
    o[0] = cadd(e0, q0); o[4] = csub(e0, q0);
    o[1] = cadd(e1, t1); o[5] = csub(e1, t1);
    o[2] = cadd(e2, t2); o[6] = csub(e2, t2);
    o[3] = cadd(e3, t3); o[7] = csub(e3, t3);
}
// exp(-2 pi i j / 32), j = n2 * k1 <= 21
__device__ __forceinline__ float2 w32(int j) {
    constexpr float C[22] = {1.f, 0.98078528f, 0.923879533f, 0.831469612f, 0.707106781f, 0.555570233f, 0.382683432f, 0.195090322f,
                             0.f, -0.195090322f, -0.382683432f, -0.555570233f, -0.707106781f, -0.831469612f, -0.923879533f,
                             -0.98078528f, -1.f, -0.98078528f, -0.923879533f, -0.831469612f, -0.707106781f, -0.555570233f};
    constexpr float S[22] = {0.f, -0.195090322f, -0.382683432f, -0.555570233f, -0.707106781f, -0.831469612f, -0.923879533f,
                             -0.98078528f, -1.f, -0.98078528f, -0.923879533f, -0.831469612f, -0.707106781f, -0.555570233f,
                             -0.382683432f, -0.195090322f, 0.f, 0.195090322f, 0.382683432f, 0.555570233f, 0.707106781f, 0.831469612f};
    return make_float2(C[j], S[j]);
}
// forward 32-point DFT in registers (32 = 4 x 8: n = 8 n1 + n2, k = k1 + 4 k2). In place: natural order in, and
// X[k] is left in slot fslot(k) = 8 (k & 3) + (k >> 2) — a compile-time renaming instead of a 32-register copy.
__device__ __forceinline__ constexpr int fslot(int k) { return 8 * (k & 3) + (k >> 2); }
__device__ __forceinline__ void fft32(float2 (&x)[32]) {
#pragma unroll
    for (int n2 = 0; n2 < 8; ++n2) dft4(x[n2], x[8 + n2], x[16 + n2], x[24 + n2]);     // x[8 k1 + n2] = y[n2][k1]
#pragma unroll
    for (int k1 = 0; k1 < 4; ++k1) {
        float2 a[8], r[8];
#pragma unroll
        for (int n2 = 0; n2 < 8; ++n2) {
            const float2 v = x[8 * k1 + n2];
            const int j = n2 * k1;
            if (j == 0) a[n2] = v;
            else if (j == 8) a[n2] = mul_mi(v);
            else a[n2] = cmul(v, w32(j));
        }
        dft8(a, r);
#pragma unroll
        for (int k2 = 0; k2 < 8; ++k2) x[8 * k1 + k2] = r[k2];                        // X[k1 + 4 k2]
    }
}

__global__ __launch_bounds__(WAVES2 * 64) void logmel_iv_kernel(FeatArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* melw = (float*)smem;                       // [MAX_NNZ]
    int* mlo = (int*)(melw + MAX_NNZ);                // [MAX_MELS]
    int* mcnt = mlo + MAX_MELS;
    int* moff = mcnt + MAX_MELS;
    float* win_s = (float*)(moff + MAX_MELS);         // [NFFT]
    char* wave_base = (char*)(win_s + NFFT);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float2* spec = (float2*)(wave_base + wave * V2_WAVE_BYTES);   // [2 pairs][FFT_LDS]: transpose, spectrum, then values
    float2* spc = spec + 2 * FFT_LDS;                             // [2][4]: the self-mirrored bins 0 and 512

    for (int n = tid; n < a.nnz; n += WAVES2 * 64) melw[n] = a.mel_w[n];
    for (int n = tid; n < a.n_mels; n += WAVES2 * 64) { mlo[n] = a.mel_lo[n]; mcnt[n] = a.mel_cnt[n]; moff[n] = a.mel_off[n]; }
    for (int n = tid; n < NFFT; n += WAVES2 * 64) win_s[n] = a.window[n];
    __syncthreads();

    const int b = blockIdx.y;
    const int pr = lane >> 5, t = lane & 31;
    const int c0 = 2 * pr, c1 = 2 * pr + 1;
    const bool has0 = c0 < a.n_ch, has1 = c1 < a.n_ch;
    const float* wv = a.wave + (long)b * a.n_ch * a.L;
    const float* w0 = wv + (long)(has0 ? c0 : 0) * a.L;
    const float* w1 = wv + (long)(has1 ? c1 : 0) * a.L;
    const float k0 = has0 ? 1.f : 0.f, k1m = has1 ? 1.f : 0.f;    // missing channels: load channel 0, scale by zero
    float2* ex = spec + pr * FFT_LDS;
    float2* sp0 = spec;
    float2* sp1 = spec + FFT_LDS;
    // twiddle bases W^(t * 2^j): every W_1024^(t k1) is a product of at most five of them
    const float2 P1 = a.twid[t], P2 = a.twid[(2 * t) & (NFFT - 1)], P4 = a.twid[(4 * t) & (NFFT - 1)],
                 P8 = a.twid[(8 * t) & (NFFT - 1)], P16 = a.twid[(16 * t) & (NFFT - 1)];

    for (int f = wave; f < FPB2; f += WAVES2) {
        const int frame = blockIdx.x * FPB2 + f;
        if (frame >= a.T) break;                                   // uniform across the wave
        const long s0 = (long)frame * a.hop - NFFT / 2;
        const bool interior = s0 >= 0 && s0 + NFFT <= a.L;
        float2 x[32];
        if (interior) {
            // all 64 loads are issued before anything consumes them
            const float* p0 = w0 + s0 + t;
            const float* p1 = w1 + s0 + t;
#pragma unroll
            for (int m = 0; m < 32; ++m) x[m] = make_float2(p0[32 * m], p1[32 * m]);
        } else {
#pragma unroll
            for (int m = 0; m < 32; ++m) {
                long sx = s0 + t + 32 * m;
                if (sx < 0) sx = -sx;
                if (sx >= a.L) sx = 2 * (a.L - 1) - sx;
                x[m] = make_float2(w0[sx], w1[sx]);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < 32; ++m) {
            const float wn = win_s[t + 32 * m];
            x[m] = make_float2(x[m].x * (wn * k0), x[m].y * (wn * k1m));
        }
        fft32(x);                                                  // slot fslot(k1) = sum_m x[t + 32 m] W_32^(m k1)
        // twiddle W_1024^(t k1) = product over the set bits of k1 of W^(t 2^j), applied factor by factor so that no
        // table of 32 powers is ever live
#pragma unroll
        for (int k1 = 1; k1 < 32; ++k1) {
            float2 v = x[fslot(k1)];
            if (k1 & 1) v = cmul(v, P1);
            if (k1 & 2) v = cmul(v, P2);
            if (k1 & 4) v = cmul(v, P4);
            if (k1 & 8) v = cmul(v, P8);
            if (k1 & 16) v = cmul(v, P16);
            ex[k1 * EX_LD + t] = v;
        }
        ex[t] = x[0];
        // transpose: lane k1' = t gathers row k1' (the 32 t-values of that k1)
#pragma unroll
        for (int tt = 0; tt < 32; ++tt) x[tt] = ex[t * EX_LD + tt];
        fft32(x);                                                  // slot fslot(k2) = X[t + 32 k2]
#pragma unroll
        for (int k2 = 0; k2 < 32; ++k2) ex[t + 32 * k2] = x[fslot(k2)];   // natural-order spectrum over the transpose buffer

        // ---- split the packed spectra, power + intensity per bin (wave-local: this wave owns both pairs) ----------
        // IN PLACE: bin k's seven values overwrite exactly the four complex slots they were computed from
        // (pair0[k] = (P0, P1), pair0[N-k] = (P2, P3), pair1[k] = (I1, I2), pair1[N-k] = (I3, -)); bins 0 and 512 mirror
        // onto themselves and go to a 64-byte side area.
#pragma unroll 3
        for (int k = lane; k < NBIN; k += 64) {
            const int kn = (NFFT - k) & (NFFT - 1);
            const float2 z0 = sp0[k], zn0 = sp0[kn], z1 = sp1[k], zn1 = sp1[kn];
            float re[4], im[4], v[7];
            re[0] = 0.5f * (z0.x + zn0.x); im[0] = 0.5f * (z0.y - zn0.y);
            re[1] = 0.5f * (z0.y + zn0.y); im[1] = 0.5f * (zn0.x - z0.x);
            re[2] = 0.5f * (z1.x + zn1.x); im[2] = 0.5f * (z1.y - zn1.y);
            re[3] = 0.5f * (z1.y + zn1.y); im[3] = 0.5f * (zn1.x - z1.x);
#pragma unroll
            for (int c = 0; c < 4; ++c) v[c] = (c < a.n_ch) ? re[c] * re[c] + im[c] * im[c] : 0.f;
            v[4] = v[5] = v[6] = 0.f;
            if (a.with_iv) {
                const float i1 = re[0] * re[1] + im[0] * im[1];
                const float i2 = re[0] * re[2] + im[0] * im[2];
                const float i3 = re[0] * re[3] + im[0] * im[3];
                const float inv = 1.f / (sqrtf(i1 * i1 + i2 * i2 + i3 * i3) + a.iv_eps);
                v[4] = i1 * inv; v[5] = i2 * inv; v[6] = i3 * inv;
            }
            const bool self = kn == k;                              // k = 0 or 512
            float2* d0 = self ? spc + (k ? 4 : 0) : sp0 + k;
            float2* d1 = self ? spc + (k ? 5 : 1) : sp0 + kn;
            float2* d2 = self ? spc + (k ? 6 : 2) : sp1 + k;
            float2* d3 = self ? spc + (k ? 7 : 3) : sp1 + kn;
            *d0 = make_float2(v[0], v[1]);
            *d1 = make_float2(v[2], v[3]);
            *d2 = make_float2(v[4], v[5]);
            *d3 = make_float2(v[6], 0.f);
        }
        // ---- mel projection: 4 lanes per mel filter interleave its compact support (16 filters per pass), all 7
        // channels per weight read (4 ds_read_b64), then two quad shuffles combine the quarters ------------------------
        for (int mb = 0; mb < a.n_mels; mb += 16) {
            const int m = mb + (lane >> 2), q = lane & 3;
            const bool live = m < a.n_mels;
            float acc[7];
#pragma unroll
            for (int c = 0; c < 7; ++c) acc[c] = 0.f;
            const int lo = live ? mlo[m] : 0, cnt = live ? mcnt[m] : 0;
            const float* wr = melw + (live ? moff[m] : 0);
            for (int i0 = q; i0 < cnt; i0 += 4) {
                const float w = wr[i0];
                const int kb = lo + i0, kn = (NFFT - kb) & (NFFT - 1);
                const bool self = kn == kb;
                const float2 A = *(self ? spc + (kb ? 4 : 0) : sp0 + kb);
                const float2 Bv = *(self ? spc + (kb ? 5 : 1) : sp0 + kn);
                const float2 Cv = *(self ? spc + (kb ? 6 : 2) : sp1 + kb);
                const float2 Dv = *(self ? spc + (kb ? 7 : 3) : sp1 + kn);
                acc[0] = fmaf(A.x, w, acc[0]); acc[1] = fmaf(A.y, w, acc[1]);
                acc[2] = fmaf(Bv.x, w, acc[2]); acc[3] = fmaf(Bv.y, w, acc[3]);
                acc[4] = fmaf(Cv.x, w, acc[4]); acc[5] = fmaf(Cv.y, w, acc[5]);
                acc[6] = fmaf(Dv.x, w, acc[6]);
            }
#pragma unroll
            for (int c = 0; c < 7; ++c) {
                acc[c] += __shfl_xor(acc[c], 1, 64);
                acc[c] += __shfl_xor(acc[c], 2, 64);
            }
            if (live) {
                // lane q writes output channels q and q + 4 (value rows: power 0..n_ch-1, IV 4..6)
#pragma unroll
                for (int rep = 0; rep < 2; ++rep) {
                    const int oc = q + 4 * rep;
                    if (oc < a.n_out) {
                        const int vc = (oc < a.n_ch) ? oc : (4 + oc - a.n_ch);
                        float v = 0.f;
#pragma unroll
                        for (int c = 0; c < 7; ++c) v = (c == vc) ? acc[c] : v;
                        if (oc < a.n_ch) v = 10.0f * log10f(fmaxf(v, a.amin));
                        a.feat[(((long)b * a.n_out + oc) * a.T + frame) * a.n_mels + m] = v;
                    }
                }
            }
        }
    }
}

}  // namespace

extern "C" int pseld_logmel_iv_fwd(const float* wave, float* feat, int B, int n_ch, long L, int hop, int n_fft,
                                   int n_mels, const float* window, const float* twiddle, const int* mel_lo,
                                   const int* mel_cnt, const int* mel_off, const float* mel_w, int nnz,
                                   int with_iv, float amin, float iv_eps, void* stream) {
    PSELD_CHECK_ARG(wave && feat && window && twiddle && mel_lo && mel_cnt && mel_off && mel_w,
                    "logmel_iv_fwd: null pointer");
    PSELD_CHECK_ARG(n_fft == NFFT, "logmel_iv_fwd: only n_fft=1024 is built (got %d)", n_fft);
    PSELD_CHECK_ARG(B > 0 && L > NFFT / 2 && hop > 0, "logmel_iv_fwd: bad B/L/hop (%d, %ld, %d)", B, L, hop);
    PSELD_CHECK_ARG(n_ch >= 1 && n_ch <= 4, "logmel_iv_fwd: n_ch must be 1..4 (got %d)", n_ch);
    PSELD_CHECK_ARG(!with_iv || n_ch == 4, "logmel_iv_fwd: intensity vector needs 4 FOA channels");
    PSELD_CHECK_ARG(n_mels >= 1 && n_mels <= MAX_MELS, "logmel_iv_fwd: n_mels must be 1..%d", MAX_MELS);
    PSELD_CHECK_ARG(nnz >= 0 && nnz <= MAX_NNZ, "logmel_iv_fwd: mel filter bank has %d taps (max %d)", nnz, MAX_NNZ);
    FeatArgs a;
    a.wave = wave; a.feat = feat; a.window = window; a.twid = (const float2*)twiddle;
    a.mel_lo = mel_lo; a.mel_cnt = mel_cnt; a.mel_off = mel_off; a.mel_w = mel_w;
    a.L = L; a.T = (int)(1 + L / hop); a.hop = hop; a.n_ch = n_ch; a.n_out = n_ch + (with_iv ? 3 : 0);
    a.n_mels = n_mels; a.nnz = nnz; a.with_iv = with_iv; a.amin = amin; a.iv_eps = iv_eps;
    const size_t lds = MAX_NNZ * sizeof(float) + 3 * MAX_MELS * sizeof(int) + NFFT * sizeof(float) + (size_t)WAVES2 * V2_WAVE_BYTES;
    static bool attr_set = false;
    if (!attr_set) { (void)hipFuncSetAttribute((const void*)logmel_iv_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr_set = true; }
    dim3 grid(pseld_cdiv(a.T, FPB2), B, 1);
    hipLaunchKernelGGL(logmel_iv_kernel, grid, dim3(WAVES2 * 64), lds, (hipStream_t)stream, a);
    PSELD_LAUNCH_CHECK("logmel_iv_fwd");
    return PSELD_OK;
}
