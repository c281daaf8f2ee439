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
// what binds this kernel. Rounds 1-4 it was VALU-issue-bound (3 590, then 1 993 vector instructions per frame; profiles/r02_pmc_feature.json,
// r04 / r05_pmc_feature.json). Round 5: the complex arithmetic is written in packed-fp32 instructions with operand selects (below) - the frame
// loop's 1 455 static vector instructions became 829 - and the workgroups are persistent (tables built once per CU): 1.22 -> 0.88 ms per 192
// chunks. What is left is latency: 2 waves per SIMD (150 KB of LDS per workgroup = one workgroup per CU) walk a chain of four LDS exchanges
// per frame; neither the VALU (~0.40 ms of issue) nor the LDS pipe (~0.43 ms incl. bank conflicts) is saturated. Inside the step the kernel
// is NOT hidden by the second stream it runs on: with the features cached the step is 0.9 ms shorter (tools/experiments/feature_cost.py).
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
    int T, hop, n_ch, n_out, n_mels, nnz, with_iv, nb;
    float amin, iv_eps;
};


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
constexpr int FPB2 = 64;                   // frames per work item (a persistent workgroup walks the (chunk, 64-frame block) items round-robin)
constexpr int V2_WAVE_BYTES = 2 * FFT_LDS * 8;

// Complex arithmetic on packed fp32 (round 5). A complex number is one 64-bit register pair; v_pk_add / v_pk_mul / v_pk_fma_f32 pick, per
// result half, which half of each source they read (op_sel / op_sel_hi) and whether it is negated (neg_lo / neg_hi) - so a multiplication by
// -i, a conjugate, a broadcast of the real part are operand modifiers, not instructions. Written as inline asm: the compiler's own packing
// of the scalar formulation spent 466 of the frame loop's 1 455 vector instructions on v_mov (pairing unrelated scalars).
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define PSELD_PK2(name, op, mods)                                                                                  \
    __device__ __forceinline__ f32x2 name(f32x2 a, f32x2 b) { f32x2 d; asm(op " %0, %1, %2 " mods : "=v"(d) : "v"(a), "v"(b)); return d; }
PSELD_PK2(cadd, "v_pk_add_f32", "")                                                          // a + b
PSELD_PK2(csub, "v_pk_add_f32", "neg_lo:[0,1] neg_hi:[0,1]")                               // a - b
PSELD_PK2(cadd_mi, "v_pk_add_f32", "op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]")            // a + (-i) b = (a.x + b.y, a.y - b.x)
PSELD_PK2(csub_mi, "v_pk_add_f32", "op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]")            // a - (-i) b = (a.x - b.y, a.y + b.x)
PSELD_PK2(pmul, "v_pk_mul_f32", "")                                                          // (a.x b.x, a.y b.y)
PSELD_PK2(pmul_lo, "v_pk_mul_f32", "op_sel:[0,0] op_sel_hi:[1,0]")                          // a * b.x
PSELD_PK2(pmul_hi, "v_pk_mul_f32", "op_sel:[0,1] op_sel_hi:[1,1]")                          // a * b.y
#undef PSELD_PK2
__device__ __forceinline__ f32x2 pfma(f32x2 a, f32x2 b, f32x2 c) { f32x2 d; asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c)); return d; }
__device__ __forceinline__ f32x2 pfma_lo(f32x2 a, f32x2 b, f32x2 c) {      // a * b.x + c
    f32x2 d; asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "=v"(d) : "v"(a), "v"(b), "v"(c)); return d; }
__device__ __forceinline__ f32x2 pfma_hi(f32x2 a, f32x2 b, f32x2 c) {      // a * b.y + c
    f32x2 d; asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "=v"(d) : "v"(a), "v"(b), "v"(c)); return d; }
// c + r a / c - r a, r = (R, R) in scalar registers
__device__ __forceinline__ f32x2 pfma_s(f32x2 a, f32x2 r, f32x2 c) { f32x2 d; asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "s"(r), "v"(c)); return d; }
__device__ __forceinline__ f32x2 pfnma_s(f32x2 a, f32x2 r, f32x2 c) { f32x2 d; asm("v_pk_fma_f32 %0, %1, %2, %3 neg_lo:[1,0,0] neg_hi:[1,0,0]" : "=v"(d) : "v"(a), "s"(r), "v"(c)); return d; }
// complex product v w = (v.x w.x - v.y w.y, v.x w.y + v.y w.x): two instructions; w in vector or in scalar registers
__device__ __forceinline__ f32x2 cmul(f32x2 v, f32x2 w) {
    f32x2 t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=v"(t) : "v"(v), "v"(w));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "=v"(r) : "v"(v), "v"(w), "v"(t));
    return r;
}
__device__ __forceinline__ f32x2 cmul_s(f32x2 v, f32x2 w) {
    f32x2 t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=v"(t) : "v"(v), "s"(w));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "=v"(r) : "v"(v), "s"(w), "v"(t));
    return r;
}
__device__ __forceinline__ f32x2 mul_mi(f32x2 a) {                          // a * (-i) = (a.y, -a.x)
    const f32x2 c = {1.f, -1.f};
    f32x2 d; asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1]" : "=v"(d) : "v"(a), "s"(c)); return d;
}

// forward 4-point DFT, in place (8 packed additions: the -i of the odd butterfly is an operand select)
__device__ __forceinline__ void dft4(f32x2& x0, f32x2& x1, f32x2& x2, f32x2& x3) {
    const f32x2 a0 = cadd(x0, x2), a1 = csub(x0, x2), a2 = cadd(x1, x3), a3 = csub(x1, x3);
    x0 = cadd(a0, a2); x2 = csub(a0, a2); x1 = cadd_mi(a1, a3); x3 = csub_mi(a1, a3);
}
// forward 8-point DFT: in a[0..7] (natural), out X[k] (natural) written to o[0..7] (26 packed instructions)
__device__ __forceinline__ void dft8(const f32x2 (&a)[8], f32x2 (&o)[8]) {
    f32x2 e0 = a[0], e1 = a[2], e2 = a[4], e3 = a[6];
    f32x2 q0 = a[1], q1 = a[3], q2 = a[5], q3 = a[7];
    dft4(e0, e1, e2, e3);
    dft4(q0, q1, q2, q3);
    const f32x2 RR = {0.70710678118654752f, 0.70710678118654752f};
    const f32x2 s1 = cadd_mi(q1, q1);            // (q1.x + q1.y, q1.y - q1.x):  q1 (1 - i) / sqrt2 = R s1
    const f32x2 s3 = csub_mi(q3, q3);            // (q3.x - q3.y, q3.y + q3.x):  q3 (-1 - i) / sqrt2 = -R s3
    o[0] = cadd(e0, q0); o[4] = csub(e0, q0);
    o[1] = pfma_s(s1, RR, e1); o[5] = pfnma_s(s1, RR, e1);
    o[2] = cadd_mi(e2, q2); o[6] = csub_mi(e2, q2);          // q2 (-i)
    o[3] = pfnma_s(s3, RR, e3); o[7] = pfma_s(s3, RR, e3);
}
// exp(-2 pi i j / 32), j = n2 * k1 <= 21
__device__ __forceinline__ constexpr float w32c(int j) {
    constexpr float C[22] = {1.f, 0.98078528f, 0.923879533f, 0.831469612f, 0.707106781f, 0.555570233f, 0.382683432f, 0.195090322f,
                             0.f, -0.195090322f, -0.382683432f, -0.555570233f, -0.707106781f, -0.831469612f, -0.923879533f,
                             -0.98078528f, -1.f, -0.98078528f, -0.923879533f, -0.831469612f, -0.707106781f, -0.555570233f};
    return C[j];
}
__device__ __forceinline__ constexpr float w32s(int j) {
    constexpr float S[22] = {0.f, -0.195090322f, -0.382683432f, -0.555570233f, -0.707106781f, -0.831469612f, -0.923879533f,
                             -0.98078528f, -1.f, -0.98078528f, -0.923879533f, -0.831469612f, -0.707106781f, -0.555570233f,
                             -0.382683432f, -0.195090322f, 0.f, 0.195090322f, 0.382683432f, 0.555570233f, 0.707106781f, 0.831469612f};
    return S[j];
}
// forward 32-point DFT in registers (32 = 4 x 8: n = 8 n1 + n2, k = k1 + 4 k2). In place: natural order in, and
// X[k] is left in slot fslot(k) = 8 (k & 3) + (k >> 2) — a compile-time renaming instead of a 32-register copy.
__device__ __forceinline__ constexpr int fslot(int k) { return 8 * (k & 3) + (k >> 2); }
__device__ __forceinline__ void fft32(f32x2 (&x)[32]) {
#pragma unroll
    for (int n2 = 0; n2 < 8; ++n2) dft4(x[n2], x[8 + n2], x[16 + n2], x[24 + n2]);     // x[8 k1 + n2] = y[n2][k1]
#pragma unroll
    for (int k1 = 0; k1 < 4; ++k1) {
        f32x2 a[8], r[8];
#pragma unroll
        for (int n2 = 0; n2 < 8; ++n2) {
            const f32x2 v = x[8 * k1 + n2];
            const int j = n2 * k1;
            if (j == 0) a[n2] = v;
            else if (j == 8) a[n2] = mul_mi(v);
            else { const f32x2 w = {w32c(j), w32s(j)}; a[n2] = cmul_s(v, w); }
        }
        dft8(a, r);
#pragma unroll
        for (int k2 = 0; k2 < 8; ++k2) x[8 * k1 + k2] = r[k2];                        // X[k1 + 4 k2]
    }
}

// ---- round 4: frame-invariant tables in registers / LDS, one pass over the spectrum, balanced mel projection -------------------
// What changed against the round-1..3 kernel (3 590 vector instructions per frame, stamps: mel projection 43 %, spectrum split 24 %):
//  * the 31 inter-stage twiddles W_1024^(t k1) are a per-workgroup LDS table (one ds_read_b64 + one complex multiply each; they were
//    rebuilt per frame from five base powers: 80 complex multiplies); the window is held in 32 registers across the frame loop;
//  * the spectrum split reads every Z[k], Z[N-k] it needs FIRST (one wave owns the region: its LDS operations execute in program
//    order), then writes the seven per-bin values as two dense planes VP[k] = (P0..P3), VI[k] = (I1..I3, -) indexed by the bin - no
//    mirrored in-place slots, no per-bin pointer selects; raw v_sqrt / v_rcp instead of the IEEE sequences, the factors 1/2 of the
//    real / imaginary split folded into one 1/4 on the mel sums and into the IV epsilon;
//  * mel projection: the 998 (filter, bin) weights are cut into runs of <= MEL_CAP consecutive bins of one filter (115 runs at 12, 124 at 11 for the
//    64-mel HTK bank); a lane owns one run per pass (2 passes), its 12 weights live in registers across the frame loop, a bin's
//    seven values arrive as two ds_read_b128 (was: five dependent LDS reads per weight, the widest filter setting the trip count
//    of all 16 filters of its pass: 26 iterations of 25 instructions), the runs of one filter meet by two lane shuffles;
//  * the next frame's samples are requested before the split / mel phase of the current one.
#ifndef PSELD_MEL_CAP
#define PSELD_MEL_CAP 12
#endif
constexpr int MEL_CAP = PSELD_MEL_CAP;     // bins per run
constexpr int CAP_LD = (MEL_CAP + 1) & ~1;     // weights per run in the LDS table (pairs)
constexpr int VBINS = 528;                 // bins per value plane (513 + the overhang of a zero-weighted run tail)
constexpr int VI_OFF = VBINS * 16;         // byte offset of the IV plane inside a wave's region (2 * 8448 = 16896 = 2 * FFT_LDS * 8)
static_assert(2 * VBINS * 16 == 2 * FFT_LDS * 8, "value planes must overlay the two spectrum regions exactly");

__global__ __launch_bounds__(WAVES2 * 64) void logmel_iv_kernel(FeatArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f32x2* tws = (f32x2*)smem;                        // [32 k1][32 t]: W_1024^(t k1)
    int* sched = (int*)(tws + 32 * 32);               // [128 runs][2]: {k0 | band << 16 | first << 24 | following runs << 25, weight offset | n << 16}
    int* sched_ok = sched + 256;                      // [4]
    float* mws = (float*)(sched_ok + 4);              // [128 runs][12]: zero-padded weights of each run
    float* wins = mws + 128 * CAP_LD;                // [32 t][36]: window[t + 32 m], rows padded to 144 B (conflict-free ds_read_b128)
    char* wave_base = (char*)(wins + 32 * 36);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    char* region = wave_base + wave * V2_WAVE_BYTES;
    f32x2* spec = (f32x2*)region;                     // [2 pairs][FFT_LDS]: transpose buffer, then the spectrum, then the value planes

    const int pr = lane >> 5, t = lane & 31;
    for (int n = tid; n < 32 * 32; n += WAVES2 * 64) { const float2 w = a.twid[((n & 31) * (n >> 5)) & (NFFT - 1)]; tws[n] = f32x2{w.x, w.y}; }
    for (int n = tid; n < 256; n += WAVES2 * 64) sched[n] = 0;
    for (int n = tid; n < NFFT; n += WAVES2 * 64) wins[(n & 31) * 36 + (n >> 5)] = a.window[n];
    __syncthreads();
    if (wave == 0) {
        // runs of filter m = lane: ceil(cnt / 12), laid out in filter order; a filter's runs never straddle the pass boundary (run 64)
        const int m = lane;
        const int cnt = m < a.n_mels ? a.mel_cnt[m] : 0, lo = m < a.n_mels ? a.mel_lo[m] : 0, off = m < a.n_mels ? a.mel_off[m] : 0;
        const int parts = (cnt + MEL_CAP - 1) / MEL_CAP;
        int incl = parts;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(incl, o, 64); if (lane >= o) incl += v; }
        int start = incl - parts;
        const unsigned long long strad = __ballot(start < 64 && start + parts > 64);
        int pad = 0;
        if (strad) {
            const int bs = __builtin_ctzll(strad);
            pad = 64 - __shfl(start, bs, 64);
            if (m >= bs) start += pad;
        }
        const int total = __shfl(incl, 63, 64) + pad;
        const unsigned long long toolong = __ballot(parts > 8);
        const bool ok = a.n_mels <= 64 && total <= 128 && !toolong;
        if (ok)
            for (int p = 0; p < parts; ++p) {
                const int n = min(MEL_CAP, cnt - p * MEL_CAP);
                sched[2 * (start + p)] = (lo + p * MEL_CAP) | (m << 16) | ((p == 0) << 24) | ((parts - 1 - p) << 25);
                sched[2 * (start + p) + 1] = (off + p * MEL_CAP) | (n << 16);
            }
        if (lane == 0) sched_ok[0] = ok;
    }
    __syncthreads();
    const bool fast_mel = sched_ok[0] != 0;
    // this lane's two runs (frame-invariant): first bin, filter, combine role; the runs' zero-padded weights go to the LDS table
    int mk0[2], mband[2], mfirst[2], mfollow[2];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int e0 = sched[2 * (p * 64 + lane)];
        mk0[p] = e0 & 0xffff; mband[p] = (e0 >> 16) & 0xff; mfirst[p] = (e0 >> 24) & 1; mfollow[p] = (e0 >> 25) & 15;
    }
    for (int n = tid; n < 128 * CAP_LD; n += WAVES2 * 64) {
        const int r = n / CAP_LD, i = n - r * CAP_LD, e1 = sched[2 * r + 1];
        mws[n] = (fast_mel && i < (e1 >> 16)) ? a.mel_w[(e1 & 0xffff) + i] : 0.f;
    }
    __syncthreads();

    // Persistent workgroups (round 5): the 14 KB of tables above are built once per workgroup, not once per 32 frames (the prologue - the
    // table gathers, wave 0's serial run schedule, three barriers - was 12 % of the kernel); a workgroup then walks the (chunk, 32-frame block)
    // items round-robin. No barrier below this line: the eight waves drift apart freely.
    const int c0 = 2 * pr, c1 = 2 * pr + 1;
    const bool has0 = c0 < a.n_ch, has1 = c1 < a.n_ch;
    const int nfb = (a.T + FPB2 - 1) / FPB2;
  for (int item = blockIdx.x; item < nfb * a.nb; item += gridDim.x) {
    const int b = item / nfb;
    const float* wv = a.wave + (long)b * a.n_ch * a.L;
    const float* w0 = wv + (long)(has0 ? c0 : 0) * a.L;
    const float* w1 = wv + (long)(has1 ? c1 : 0) * a.L;
    const float k0f = has0 ? 1.f : 0.f, k1f = has1 ? 1.f : 0.f;   // missing channels: load channel 0, scale by zero
    f32x2* ex = spec + pr * FFT_LDS;
    const f32x2* sp0 = spec;
    const f32x2* sp1 = spec + FFT_LDS;
    const float eps4 = 4.f * a.iv_eps;

    auto interior = [&](int frame) { const long s0 = (long)frame * a.hop - NFFT / 2; return s0 >= 0 && s0 + NFFT <= a.L; };
    auto load_interior = [&](int frame, f32x2 (&x)[32]) {        // 64 loads at immediate offsets from two base pointers
        const long s0 = (long)frame * a.hop - NFFT / 2;
        const float* p0 = w0 + s0 + t;
        const float* p1 = w1 + s0 + t;
#pragma unroll
        for (int m = 0; m < 32; ++m) x[m] = f32x2{p0[32 * m], p1[32 * m]};
    };
    // the 6 frames of a chunk that reach over its ends (reflect padding): a ROLLED loop stages the samples in the wave's (idle)
    // transpose buffer - unrolled, the 64 reflected 64-bit addresses cost the whole kernel 80 registers
    auto load_edge = [&](int frame, f32x2 (&x)[32]) {
        const long s0 = (long)frame * a.hop - NFFT / 2;
#pragma unroll 1
        for (int m = 0; m < 32; ++m) {
            long sx = s0 + t + 32 * m;
            if (sx < 0) sx = -sx;
            if (sx >= a.L) sx = 2 * (a.L - 1) - sx;
            ex[m * 32 + t] = f32x2{w0[sx], w1[sx]};
        }
#pragma unroll
        for (int m = 0; m < 32; ++m) x[m] = ex[m * 32 + t];
    };

    const int frame0 = (item - b * nfb) * FPB2;
    f32x2 xr[32];
    if (frame0 + wave < a.T) { if (interior(frame0 + wave)) load_interior(frame0 + wave, xr); else load_edge(frame0 + wave, xr); }
    for (int f = wave; f < FPB2; f += WAVES2) {
        const int frame = frame0 + f;
        if (frame >= a.T) break;                                   // uniform across the wave
        f32x2 x[32];
#pragma unroll
        for (int m2 = 0; m2 < 16; ++m2) {
            const f32x2 wn = *(const f32x2*)(wins + t * 36 + 2 * m2);      // window[t + 32 m], m = 2 m2, 2 m2 + 1
            x[2 * m2] = pmul_lo(xr[2 * m2], wn);
            x[2 * m2 + 1] = pmul_hi(xr[2 * m2 + 1], wn);
        }
        if (a.n_ch < 4) {
            const f32x2 kf = {k0f, k1f};
#pragma unroll
            for (int m = 0; m < 32; ++m) x[m] = pmul(x[m], kf);
        }
        fft32(x);                                                  // slot fslot(k1) = sum_m x[t + 32 m] W_32^(m k1)
#pragma unroll
        for (int k1 = 1; k1 < 32; ++k1) ex[k1 * EX_LD + t] = cmul(x[fslot(k1)], tws[k1 * 32 + t]);
        ex[t] = x[0];
        asm volatile("" ::: "memory");   // (lanes exchange data through LDS: the compiler must keep the order it can prove irrelevant for ONE lane)
        // transpose: lane k1' = t gathers row k1' (the 32 t-values of that k1)
#pragma unroll
        for (int tt = 0; tt < 32; ++tt) x[tt] = ex[t * EX_LD + tt];
        asm volatile("" ::: "memory");   // (lanes exchange data through LDS: the compiler must keep the order it can prove irrelevant for ONE lane)
        fft32(x);                                                  // slot fslot(k2) = X[t + 32 k2]
#pragma unroll
        for (int k2 = 0; k2 < 32; ++k2) ex[t + 32 * k2] = x[fslot(k2)];   // natural-order spectrum over the transpose buffer
        asm volatile("" ::: "memory");   // (lanes exchange data through LDS: the compiler must keep the order it can prove irrelevant for ONE lane)
        // ---- split the packed spectra, power + intensity per bin: ALL reads, then the value planes over the same region ----
        f32x2 z0[9], zn0[9], z1[9], zn1[9];
#pragma unroll
        for (int j = 0; j < 9; ++j) {
            const int k = min(lane + 64 * j, NBIN - 1), kn = (NFFT - k) & (NFFT - 1);
            z0[j] = sp0[k]; zn0[j] = sp0[kn]; z1[j] = sp1[k]; zn1[j] = sp1[kn];
        }
        asm volatile("" ::: "memory");       // no read of the spectrum may be re-issued behind the plane stores below
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 9; ++j) {
            const int k = lane + 64 * j;
            // ch0 = a / 2, ch1 = b / (2i), ch2 = c / 2, ch3 = d / (2i)  with  a = z + conj(zn), b = z - conj(zn). Packed as the pairs the products
            // below want: X = (ax, bx), Y = (ay, by), P = (cx, dy), Q = (cy, dx) - four instructions
            f32x2 X, Y, P, Q, t, pw01, pw23;
            asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,0] neg_hi:[0,1]" : "=v"(X) : "v"(z0[j]), "v"(zn0[j]));       // (z.x + zn.x, z.x - zn.x)
            asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,1] neg_lo:[0,1]" : "=v"(Y) : "v"(z0[j]), "v"(zn0[j]));       // (z.y - zn.y, z.y + zn.y)
            P = cadd(z1[j], zn1[j]);                                                                                              // (z.x + zn.x, z.y + zn.y)
            asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,0] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(Q) : "v"(z1[j]), "v"(zn1[j]));   // (z.y - zn.y, z.x - zn.x)
            pw01 = pfma(Y, Y, pmul(X, X));                                   // 4 x power of channels 0, 1
            pw23 = pfma(Q, Q, pmul(P, P));                                   // (cx^2 + cy^2, dy^2 + dx^2): channels 2, 3
            f32x4 iv = {0.f, 0.f, 0.f, 0.f};
            if (a.with_iv) {
                // 4 x intensity: i1 = ax by - ay bx, (i2, i3) = (ax cx + ay cy, ax dy - ay dx) = ax P + ay (Q.x, -Q.y)
                f32x2 i23;
                asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=v"(t) : "v"(X), "v"(P));
                asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_hi:[0,1,0]" : "=v"(i23) : "v"(Y), "v"(Q), "v"(t));
                const float i1 = fmaf(X[0], Y[1], -Y[0] * X[1]);
                const float inv = __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(fmaf(i1, i1, fmaf(i23[0], i23[0], i23[1] * i23[1]))) + eps4);
                iv[0] = i1 * inv; iv[1] = i23[0] * inv; iv[2] = i23[1] * inv;
            }
            if (k < NBIN) {
                *(f32x4*)(region + k * 16) = __builtin_shufflevector(pw01, pw23, 0, 1, 2, 3);     // one 16-byte write per plane: conflict-free
                *(f32x4*)(region + VI_OFF + k * 16) = iv;
            }
        }
        asm volatile("" ::: "memory");   // (lanes exchange data through LDS: the compiler must keep the order it can prove irrelevant for ONE lane)
        // the next frame's samples travel while this frame is projected onto the mel bands
        const bool more = f + WAVES2 < FPB2 && frame + WAVES2 < a.T;
        const bool more_in = more && interior(frame + WAVES2);
        if (more_in) load_interior(frame + WAVES2, xr);
        if (fast_mel) {
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const char* vp = region + mk0[p] * 16;
                f32x2 w2[CAP_LD / 2];
#pragma unroll
                for (int i = 0; i < CAP_LD / 2; ++i) w2[i] = *(const f32x2*)(mws + (p * 64 + lane) * CAP_LD + 2 * i);
                f32x2 a01 = {0.f, 0.f}, a23 = a01, a45 = a01;
                float a6 = 0.f;
#pragma unroll
                for (int i = 0; i < MEL_CAP; ++i) {                 // a bin's seven values times its weight: three packed FMAs + one
                    const f32x4 P = *(const f32x4*)(vp + i * 16), I = *(const f32x4*)(vp + VI_OFF + i * 16);
                    const f32x2 P01 = __builtin_shufflevector(P, P, 0, 1), P23 = __builtin_shufflevector(P, P, 2, 3);
                    const f32x2 I01 = __builtin_shufflevector(I, I, 0, 1);
                    const float I2 = I[2];
                    if (i & 1) { a01 = pfma_hi(P01, w2[i >> 1], a01); a23 = pfma_hi(P23, w2[i >> 1], a23); a45 = pfma_hi(I01, w2[i >> 1], a45); }
                    else { a01 = pfma_lo(P01, w2[i >> 1], a01); a23 = pfma_lo(P23, w2[i >> 1], a23); a45 = pfma_lo(I01, w2[i >> 1], a45); }
                    a6 = fmaf(I2, w2[i >> 1][i & 1], a6);
                }
                float acc[7] = {a01[0], a01[1], a23[0], a23[1], a45[0], a45[1], a6};
                // the runs of one filter sit on consecutive lanes: run r collects the runs behind it (up to 7)
#pragma unroll
                for (int o = 1; o <= 4; o <<= 1) {
#pragma unroll
                    for (int c = 0; c < 7; ++c) {
                        const float v = __shfl_down(acc[c], o, 64);
                        acc[c] += (mfollow[p] >= o) ? v : 0.f;
                    }
                }
                if (mfirst[p]) {
                    const int m = mband[p];
#pragma unroll
                    for (int oc = 0; oc < 7; ++oc) {
                        if (oc < a.n_out) {
                            float v;
                            if (oc < a.n_ch) {
                                float pv = 0.f;
#pragma unroll
                                for (int c = 0; c < 4; ++c) pv = (c == oc) ? acc[c] : pv;
                                v = 3.01029995663981f * __builtin_amdgcn_logf(fmaxf(0.25f * pv, a.amin));      // 10 log10(x) = 10 log10(2) log2(x)
                            } else {
                                v = 0.f;
#pragma unroll
                                for (int c = 0; c < 3; ++c) v = (c == oc - a.n_ch) ? acc[4 + c] : v;
                            }
                            a.feat[(((long)b * a.n_out + oc) * a.T + frame) * a.n_mels + m] = v;
                        }
                    }
                }
            }
        } else {
            // any other filter bank: 4 lanes per filter interleave its support (16 filters per pass)
            for (int mb = 0; mb < a.n_mels; mb += 16) {
                const int m = mb + (lane >> 2), q = lane & 3;
                const bool live = m < a.n_mels;
                float acc[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                const int lo = live ? a.mel_lo[m] : 0, cnt = live ? a.mel_cnt[m] : 0;
                const float* wr = a.mel_w + (live ? a.mel_off[m] : 0);
                for (int i0 = q; i0 < cnt; i0 += 4) {
                    const float w = wr[i0];
                    const f32x4 P = *(const f32x4*)(region + (lo + i0) * 16);
                    const f32x4 I = *(const f32x4*)(region + VI_OFF + (lo + i0) * 16);
                    acc[0] = fmaf(P[0], w, acc[0]); acc[1] = fmaf(P[1], w, acc[1]); acc[2] = fmaf(P[2], w, acc[2]); acc[3] = fmaf(P[3], w, acc[3]);
                    acc[4] = fmaf(I[0], w, acc[4]); acc[5] = fmaf(I[1], w, acc[5]); acc[6] = fmaf(I[2], w, acc[6]);
                }
#pragma unroll
                for (int c = 0; c < 7; ++c) {
                    acc[c] += __shfl_xor(acc[c], 1, 64);
                    acc[c] += __shfl_xor(acc[c], 2, 64);
                }
                if (live && q == 0) {
                    for (int oc = 0; oc < a.n_out; ++oc) {
                        float v = 0.f;
                        const int vc = (oc < a.n_ch) ? oc : (4 + oc - a.n_ch);
#pragma unroll
                        for (int c = 0; c < 7; ++c) v = (c == vc) ? acc[c] : v;
                        if (oc < a.n_ch) v = 3.01029995663981f * __builtin_amdgcn_logf(fmaxf(0.25f * v, a.amin));
                        a.feat[(((long)b * a.n_out + oc) * a.T + frame) * a.n_mels + m] = v;
                    }
                }
            }
        }
        asm volatile("" ::: "memory");   // (lanes exchange data through LDS: the compiler must keep the order it can prove irrelevant for ONE lane)
        if (more && !more_in) load_edge(frame + WAVES2, xr);     // (the region is idle again)
        asm volatile("" ::: "memory");   // (lanes exchange data through LDS: the compiler must keep the order it can prove irrelevant for ONE lane)
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
    a.n_mels = n_mels; a.nnz = nnz; a.with_iv = with_iv; a.amin = amin; a.iv_eps = iv_eps; a.nb = B;
    const size_t lds = 32 * 32 * sizeof(float2) + 256 * sizeof(int) + 4 * sizeof(int) + 128 * CAP_LD * sizeof(float) + 32 * 36 * sizeof(float) +
                       (size_t)WAVES2 * V2_WAVE_BYTES;
    static bool attr_set = false;
    if (!attr_set) { (void)hipFuncSetAttribute((const void*)logmel_iv_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr_set = true; }
    static int n_cu = 0;
    if (n_cu == 0) {
        int dev = 0; hipDeviceProp_t pr;
        n_cu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0) ? pr.multiProcessorCount : 256;
    }
    const long items = (long)pseld_cdiv(a.T, FPB2) * B;          // one workgroup per CU (150 KB of LDS), walking the items round-robin
    PSELD_CHECK_ARG(items < (1L << 31), "logmel_iv_fwd: B x T too large");
    dim3 grid((unsigned)(items < n_cu ? items : n_cu), 1, 1);
    hipLaunchKernelGGL(logmel_iv_kernel, grid, dim3(WAVES2 * 64), lds, (hipStream_t)stream, a);
    PSELD_LAUNCH_CHECK("logmel_iv_fwd");
    return PSELD_OK;
}
