// K1: fused framed STFT -> power -> mel -> dB  +  FOA intensity vector -> L2-normalise -> mel.
//
// Replaces (reference, /root/reference/src): utils/feature.py:39-56 (LogmelIV_Extractor.forward),
// utils/feature.py:78-91 (Logmel_Extractor.forward), utils/feature.py:93-117 (intensityvector) and the
// torchaudio 2.2.1 transforms they call (Spectrogram: reflect-padded centred STFT, periodic window,
// onesided; MelScale: spec^T @ fb; AmplitudeToDB('power', top_db=None)).
//
// One 256-thread workgroup transforms FPB consecutive frames of one clip. Per frame the 4 real channels
// are packed into two complex signals (ch0 + i*ch1, ch2 + i*ch3), each transformed by a 1024-point
// radix-4 Stockham FFT that lives entirely in LDS (5 passes, ping-pong buffers, twiddles from an LDS
// table), then split back into the four one-sided spectra. Nothing but the waveform is read from HBM and
// nothing but the [7, T, n_mels] features is written: the complex STFT, the power spectrogram and the
// three 513-bin intensity maps of the unfused reference never exist in memory.
//
// Roofline: HBM-bound by design (5.634 MB algorithmic bytes per 10 s chunk: 3.84 MB in + 1.794 MB out).
#include "common.h"

namespace {

constexpr int NFFT = 1024;
constexpr int NBIN = NFFT / 2 + 1;  // 513
constexpr int NT = 256;             // threads per workgroup = NFFT / 4
constexpr int FPB = 8;              // frames per workgroup
constexpr int VAL_LD = 520;         // padded row length of the per-bin value rows
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

// One radix-4 Stockham pass for sub-transform length P (inputs already have length-P DFTs interleaved).
// Thread i combines src[i + t*NT], t = 0..3, and scatters to dst[j + t*P].
template <int P>
__device__ __forceinline__ void radix4_pass(const float2* __restrict__ src, float2* __restrict__ dst,
                                            const float2* __restrict__ tw, int i) {
    const int k = i & (P - 1);
    const int j = ((i - k) << 2) + k;
    constexpr int STEP = NFFT / (4 * P);
    float2 u0 = src[i];
    float2 u1 = src[i + NT];
    float2 u2 = src[i + 2 * NT];
    float2 u3 = src[i + 3 * NT];
    if (P > 1) {
        u1 = cmul(u1, tw[(k * STEP) & (NFFT - 1)]);
        u2 = cmul(u2, tw[(2 * k * STEP) & (NFFT - 1)]);
        u3 = cmul(u3, tw[(3 * k * STEP) & (NFFT - 1)]);
    }
    const float2 v0 = make_float2(u0.x + u2.x, u0.y + u2.y);
    const float2 v1 = make_float2(u0.x - u2.x, u0.y - u2.y);
    const float2 v2 = make_float2(u1.x + u3.x, u1.y + u3.y);
    const float2 d = make_float2(u1.x - u3.x, u1.y - u3.y);
    const float2 v3 = make_float2(d.y, -d.x);  // (u1 - u3) * (-i)
    dst[j] = make_float2(v0.x + v2.x, v0.y + v2.y);
    dst[j + P] = make_float2(v1.x + v3.x, v1.y + v3.y);
    dst[j + 2 * P] = make_float2(v0.x - v2.x, v0.y - v2.y);
    dst[j + 3 * P] = make_float2(v1.x - v3.x, v1.y - v3.y);
}

__global__ __launch_bounds__(NT) void logmel_iv_kernel(FeatArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2* buf0 = (float2*)smem;                    // [2][NFFT]  16 KB
    float2* buf1 = buf0 + 2 * NFFT;                  // [2][NFFT]  16 KB (re-used as val[7][VAL_LD])
    float2* tw = buf1 + 2 * NFFT;                    // [NFFT]      8 KB
    float* melw = (float*)(tw + NFFT);               // [MAX_NNZ]   8 KB
    int* mlo = (int*)(melw + MAX_NNZ);               // [MAX_MELS]
    int* mcnt = mlo + MAX_MELS;
    int* moff = mcnt + MAX_MELS;
    float* val = (float*)buf1;

    const int tid = threadIdx.x;
    const int b = blockIdx.y;
    const int frame0 = blockIdx.x * FPB;

    for (int n = tid; n < NFFT; n += NT) tw[n] = a.twid[n];
    for (int n = tid; n < a.nnz; n += NT) melw[n] = a.mel_w[n];
    for (int n = tid; n < a.n_mels; n += NT) { mlo[n] = a.mel_lo[n]; mcnt[n] = a.mel_cnt[n]; moff[n] = a.mel_off[n]; }

    // thread i owns the 4 CONSECUTIVE samples 4i..4i+3 of a frame: one 16-byte load per channel (frame starts are
    // multiples of 16 samples when hop % 16 == 0), instead of 16 scalar loads of a 4.27x-overlapped stream
    const f32x4 win4 = *(const f32x4*)(a.window + 4 * tid);
    __syncthreads();

    const float* wv = a.wave + (long)b * a.n_ch * a.L;
    const int n_pairs = (a.n_ch + 1) >> 1;
    const bool vec_ok = (a.hop % 4 == 0) && (a.L % 4 == 0) && (((unsigned long)a.wave & 15) == 0);

    for (int f = 0; f < FPB; ++f) {
        const int frame = frame0 + f;
        if (frame >= a.T) break;  // uniform across the workgroup

        // ---- windowed, reflect-padded frame from HBM/L2 into LDS in natural order (buf1) ------------------
        const long s0 = (long)frame * a.hop - NFFT / 2;
        const bool interior = vec_ok && s0 >= 0 && s0 + NFFT <= a.L;   // uniform: no reflection in this frame
        for (int pr = 0; pr < n_pairs; ++pr) {
            const int c0 = 2 * pr, c1 = 2 * pr + 1;
            const float* w0 = wv + (long)c0 * a.L;
            const float* w1 = wv + (long)c1 * a.L;
            const bool has1 = c1 < a.n_ch;
            f32x4 x0, x1 = {0.f, 0.f, 0.f, 0.f};
            if (interior) {
                x0 = *(const f32x4*)(w0 + s0 + 4 * tid);
                if (has1) x1 = *(const f32x4*)(w1 + s0 + 4 * tid);
            } else {
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    long sx = s0 + 4 * tid + t;
                    if (sx < 0) sx = -sx;
                    if (sx >= a.L) sx = 2 * (a.L - 1) - sx;
                    x0[t] = w0[sx];
                    if (has1) x1[t] = w1[sx];
                }
            }
            float2* dst = buf1 + pr * NFFT + 4 * tid;
#pragma unroll
            for (int t = 0; t < 4; ++t) dst[t] = make_float2(x0[t] * win4[t], x1[t] * win4[t]);
        }
        __syncthreads();
        // ---- pass 1 (P = 1, no twiddles) ------------------------------------------------------------------
        for (int pr = 0; pr < n_pairs; ++pr) radix4_pass<1>(buf1 + pr * NFFT, buf0 + pr * NFFT, tw, tid);
        __syncthreads();
        // ---- passes 2..5 in LDS ------------------------------------------------------------------------
        for (int pr = 0; pr < n_pairs; ++pr) radix4_pass<4>(buf0 + pr * NFFT, buf1 + pr * NFFT, tw, tid);
        __syncthreads();
        for (int pr = 0; pr < n_pairs; ++pr) radix4_pass<16>(buf1 + pr * NFFT, buf0 + pr * NFFT, tw, tid);
        __syncthreads();
        for (int pr = 0; pr < n_pairs; ++pr) radix4_pass<64>(buf0 + pr * NFFT, buf1 + pr * NFFT, tw, tid);
        __syncthreads();
        for (int pr = 0; pr < n_pairs; ++pr) radix4_pass<256>(buf1 + pr * NFFT, buf0 + pr * NFFT, tw, tid);
        __syncthreads();

        // ---- split the packed spectra, power + intensity per bin (buf0 -> val, which aliases buf1) ---
        for (int k = tid; k < NBIN; k += NT) {
            const int kn = (NFFT - k) & (NFFT - 1);
            float re[4], im[4];
#pragma unroll
            for (int pr = 0; pr < 2; ++pr) {
                if (pr < n_pairs) {
                    const float2 z = buf0[pr * NFFT + k];
                    const float2 zn = buf0[pr * NFFT + kn];
                    re[2 * pr] = 0.5f * (z.x + zn.x);
                    im[2 * pr] = 0.5f * (z.y - zn.y);
                    re[2 * pr + 1] = 0.5f * (z.y + zn.y);
                    im[2 * pr + 1] = 0.5f * (zn.x - z.x);
                } else {
                    re[2 * pr] = im[2 * pr] = re[2 * pr + 1] = im[2 * pr + 1] = 0.f;
                }
            }
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (c < a.n_ch) val[c * VAL_LD + k] = re[c] * re[c] + im[c] * im[c];
            if (a.with_iv) {
                const float i1 = re[0] * re[1] + im[0] * im[1];
                const float i2 = re[0] * re[2] + im[0] * im[2];
                const float i3 = re[0] * re[3] + im[0] * im[3];
                const float nrm = sqrtf(i1 * i1 + i2 * i2 + i3 * i3) + a.iv_eps;
                val[4 * VAL_LD + k] = i1 / nrm;
                val[5 * VAL_LD + k] = i2 / nrm;
                val[6 * VAL_LD + k] = i3 / nrm;
            }
        }
        __syncthreads();

        // ---- mel projection over each filter's compact support, dB for the power channels -----------
        // thread = (mel m, quarter q): the 4 lanes of a mel interleave its bins and carry all output channels at
        // once (7 independent FMA chains per LDS weight read), then combine with two shuffles.
        for (int mb = 0; mb < a.n_mels; mb += NT / 4) {
            const int m = mb + (tid >> 2), q = tid & 3;
            float acc[7];
#pragma unroll
            for (int c = 0; c < 7; ++c) acc[c] = 0.f;
            if (m < a.n_mels) {
                const int lo = mlo[m], cnt = mcnt[m];
                const float* wr = melw + moff[m];
                for (int i = q; i < cnt; i += 4) {
                    const float w = wr[i];
                    const float* vr = val + lo + i;
#pragma unroll
                    for (int c = 0; c < 7; ++c)
                        if (c < a.n_ch || (a.with_iv && c >= 4)) acc[c] = fmaf(vr[c * VAL_LD], w, acc[c]);
                }
            }
#pragma unroll
            for (int c = 0; c < 7; ++c) {
                acc[c] += __shfl_xor(acc[c], 1, 64);
                acc[c] += __shfl_xor(acc[c], 2, 64);
            }
            if (m < a.n_mels) {
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
        __syncthreads();
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
    const size_t lds = (size_t)(5 * NFFT) * sizeof(float2) + MAX_NNZ * sizeof(float) + 3 * MAX_MELS * sizeof(int);
    dim3 grid(pseld_cdiv(a.T, FPB), B, 1);
    hipLaunchKernelGGL(logmel_iv_kernel, grid, dim3(NT), lds, (hipStream_t)stream, a);
    PSELD_LAUNCH_CHECK("logmel_iv_fwd");
    return PSELD_OK;
}
