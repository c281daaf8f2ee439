// On-device data augmentations of the training step (gfx950). All HBM-bound elementwise / gather kernels.
//
// Replaces (reference, /root/reference/src/augment): specaug.py:14-63 (time masks on data AND labels, iid frequency masks),
// crop.py:10-32 (random rectangles per sample and channel), freqshift.py:17-38 (reflect-padded shift along frequency),
// rotate.py:10-101 (FOA channel swap / sign flips and the same on the DOA labels), trackmix.py:15-75 and
// wavmix.py:16-116 (pairwise mixing of samples with the ADPIT / ACCDOA / track-wise label surgery). The random
// parameters (mask positions, shifts, permutations, pairings, Beta weights) are drawn by the host-side mirror
// (pseldnets_amd/augment) with the reference's own generator calls; these kernels apply them to the whole batch at once
// instead of the reference's Python loops over samples.
#include "common.h"

namespace {

// x [N, C, T, F]; rects [N*C][R][4] = (t0, t1, f0, f1): x[n,c,t,f] = value inside any rectangle
__global__ void rect_fill_kernel(float* __restrict__ x, const int* __restrict__ rects, int T, int F, int R, float value, long total) {
    const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= total) return;
    const int f = (int)(id % F);
    const long rest = id / F;
    const int t = (int)(rest % T);
    const long nc = rest / T;
    const int* rc = rects + nc * R * 4;
    bool hit = false;
    for (int i = 0; i < R; ++i) hit |= (t >= rc[4 * i] && t < rc[4 * i + 1] && f >= rc[4 * i + 2] && f < rc[4 * i + 3]);
    if (hit) x[id] = value;
}
// y [N, Ty, inner]; spans [N][R][2]: y[n,t,:] = value for t inside any span
__global__ void time_fill_kernel(float* __restrict__ y, const int* __restrict__ spans, int Ty, long inner, int R, float value, long total) {
    const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= total) return;
    const long rest = id / inner;
    const int t = (int)(rest % Ty);
    const long n = rest / Ty;
    const int* sp = spans + n * R * 2;
    bool hit = false;
    for (int i = 0; i < R; ++i) hit |= (t >= sp[2 * i] && t < sp[2 * i + 1]);
    if (hit) y[id] = value;
}
// freqshift.py:26-38: s > 0 'up' = F.pad(x, (s, 0), 'reflect')[..., :F]; s < 0 'down' = F.pad(x, (0, -s), 'reflect')[..., -s:]
__global__ void freqshift_kernel(const float* __restrict__ x, float* __restrict__ y, const int* __restrict__ shift, int F, long per_sample,
                                 long total) {
    const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= total) return;
    const int f = (int)(id % F);
    const int s = shift[id / per_sample];
    int src;
    if (s >= 0) src = f < s ? s - f : f - s;
    else src = f - s < F ? f - s : 2 * F - 2 - (f - s);
    y[id] = x[id - f + src];
}
// rotate.py:71-73: y[n,0] = x[n,0]; y[n,1+j] = sign[n][j] * x[n, src[n][j]]
__global__ void rotate_wave_kernel(const float* __restrict__ x, float* __restrict__ y, const int* __restrict__ src, const float* __restrict__ sign,
                                   long L4, long total4) {
    const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= total4) return;
    const long l = id % L4;
    const long nc = id / L4;
    const int ch = (int)(nc & 3);
    const long n = nc >> 2;
    f32x4 v;
    if (ch == 0) v = ((const f32x4*)x)[id];
    else {
        const int sc = src[n * 3 + ch - 1];
        v = ((const f32x4*)x)[(n * 4 + sc) * L4 + l] * sign[n * 3 + ch - 1];
    }
    ((f32x4*)y)[id] = v;
}
// label [N, outer, A, inner]: y[n,o,a0+j,i] = sign[n][j] * x[n,o,a0+src[n][j],i]; other axis positions copied
__global__ void rotate_label_kernel(const float* __restrict__ x, float* __restrict__ y, const int* __restrict__ src, const float* __restrict__ sign,
                                    long outer, int A, long inner, int a0, long total) {
    const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= total) return;
    const long i = id % inner;
    long rest = id / inner;
    const int a = (int)(rest % A);
    rest /= A;
    const long n = rest / outer;
    const int j = a - a0;
    float v;
    if (j < 0 || j > 2) v = x[id];
    else v = sign[n * 3 + j] * x[id + (long)(a0 + src[n * 3 + j] - a) * inner];
    y[id] = v;
}
// trackmix.py:42 / wavmix.py:51: y[dst[p]] = lam*x[dst[p]] + (1 - lam)*x[src[p]] (x = the state BEFORE the assignment)
__global__ void mix_kernel(const float* __restrict__ x, float* __restrict__ y, const int* __restrict__ dst, const int* __restrict__ src,
                           const float* __restrict__ lam, long elems, long total) {
    const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= total) return;
    const long p = id / elems, e = id - p * elems;
    const float l = lam[p];
    // two rounded products and one rounded sum, as the reference's tensor expression (no fused multiply-add)
    y[(long)dst[p] * elems + e] = __fadd_rn(__fmul_rn(l, x[(long)dst[p] * elems + e]), __fmul_rn(__fsub_rn(1.f, l), x[(long)src[p] * elems + e]));
}

// ADPIT label surgery (trackmix.py:61-72, wavmix.py:85-113). lab [N, T, 6, 4, C]; one workgroup per (pair, frame), one thread
// per class. mode 1: the partner has one source (TrackMix; WavMix add_ov '1'); mode 2: up to two (WavMix add_ov '2').
__global__ __launch_bounds__(256) void mix_adpit_kernel(const float* __restrict__ lab, float* __restrict__ out, const int* __restrict__ dst,
                                                        const int* __restrict__ src, const float* __restrict__ lam, int T, int C, int mode) {
    const int p = blockIdx.x / T, t = blockIdx.x % T;
    const float l = lam[p], l1 = __fsub_rn(1.f, l);
    const float* a = lab + ((long)dst[p] * T + t) * 24 * C;     // [6][4][C]
    const float* b = lab + ((long)src[p] * T + t) * 24 * C;
    float* o = out + ((long)dst[p] * T + t) * 24 * C;
    bool any_col = false;
    for (int c0 = 0; c0 < C; c0 += 256) {                       // mode 1 zeroes the WHOLE frame when any class collides
        const int c = c0 + threadIdx.x;
        const bool col = c < C && __fmul_rn(a[c], b[c]) != 0.f;
        any_col |= __syncthreads_or(col) != 0;
    }
    for (int c = threadIdx.x; c < C; c += 256) {
        float va[24], vb[24], v[24];
#pragma unroll
        for (int i = 0; i < 24; ++i) { va[i] = a[i * C + c]; vb[i] = b[i * C + c]; }
#pragma unroll
        for (int tr = 0; tr < 6; ++tr) {
            v[tr * 4] = __fadd_rn(__fmul_rn(l, va[tr * 4]), __fmul_rn(l1, vb[tr * 4]));
#pragma unroll
            for (int k = 1; k < 4; ++k) v[tr * 4 + k] = __fadd_rn(va[tr * 4 + k], vb[tr * 4 + k]);
        }
        const bool col0 = __fmul_rn(va[0], vb[0]) != 0.f;       // A0 of both, same class
        if (mode == 1) {
            if (any_col) {
#pragma unroll
                for (int i = 0; i < 24; ++i) v[i] = 0.f;
            }
            if (col0) {
                v[4] = __fmul_rn(l, va[0]); v[5] = va[1]; v[6] = va[2]; v[7] = va[3];
                v[8] = __fmul_rn(l1, vb[0]); v[9] = vb[1]; v[10] = vb[2]; v[11] = vb[3];
            }
        } else {
            if (col0) {
#pragma unroll
                for (int i = 0; i < 24; ++i) v[i] = 0.f;
                v[4] = __fmul_rn(l, va[0]); v[5] = va[1]; v[6] = va[2]; v[7] = va[3];
                v[8] = __fmul_rn(l1, vb[0]); v[9] = vb[1]; v[10] = vb[2]; v[11] = vb[3];
            }
            if (__fmul_rn(va[0], vb[4]) != 0.f) {               // A0 with the partner's B0: three sources of one class
#pragma unroll
                for (int i = 0; i < 24; ++i) v[i] = 0.f;
                v[12] = __fmul_rn(l, va[0]); v[13] = va[1]; v[14] = va[2]; v[15] = va[3];
                v[16] = __fmul_rn(l1, vb[4]); v[17] = vb[5]; v[18] = vb[6]; v[19] = vb[7];
                v[20] = __fmul_rn(l1, vb[8]); v[21] = vb[9]; v[22] = vb[10]; v[23] = vb[11];
            }
        }
#pragma unroll
        for (int i = 0; i < 24; ++i) o[i * C + c] = v[i];
    }
}
// track-wise labels (trackmix.py:43-58, wavmix.py:52-72): sed [N, T, 3, C], doa [N, T, 3, 3].
// new sed tracks = (lam * a0, (1-lam) * b0, third), new doa tracks = (a0, b0, third); third = 0 (TrackMix) or (1-lam)*b1 / b1 (WavMix)
__global__ void mix_tracks_kernel(const float* __restrict__ sed, const float* __restrict__ doa, float* __restrict__ sed_o, float* __restrict__ doa_o,
                                  const int* __restrict__ dst, const int* __restrict__ src, const float* __restrict__ lam, int T, int C, int wavmix,
                                  long total) {
    const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= total) return;
    const int W = C + 3;                                         // per (pair, frame): C sed columns then 3 doa columns
    const int col = (int)(id % W);
    const long pt = id / W;
    const int t = (int)(pt % T);
    const long p = pt / T;
    const float l = lam[p], l1 = __fsub_rn(1.f, l);
    const long ra = (long)dst[p] * T + t, rb = (long)src[p] * T + t;
    if (col < C) {
        const float* a = sed + ra * 3 * C; const float* b = sed + rb * 3 * C;
        float* o = sed_o + ra * 3 * C;
        o[col] = __fmul_rn(l, a[col]);
        o[C + col] = __fmul_rn(l1, b[col]);
        o[2 * C + col] = wavmix ? __fmul_rn(l1, b[C + col]) : 0.f;
    } else {
        const int k = col - C;
        const float* a = doa + ra * 9; const float* b = doa + rb * 9;
        float* o = doa_o + ra * 9;
        o[k] = a[k];
        o[3 + k] = b[k];
        o[6 + k] = wavmix ? b[3 + k] : 0.f;
    }
}

// ---- mono -> FOA spatialisation of the mono_adapter recipe (data/data.py:17-59) -----------------------------------------------------
// foa[n] = (w, y*w, z*w, x*w) with the products formed in double and rounded once (numpy: float64 scalar * float32 array)
__global__ void spatialize_mono_kernel(const float* __restrict__ mono, long mono_stride, const double* __restrict__ xyz, float* __restrict__ foa,
                                       long L, long total) {
    const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= total) return;
    const long l = id % L, n = id / L;
    const double w = (double)mono[n * mono_stride + l];
    float* o = foa + n * 4 * L + l;
    o[0] = (float)w; o[L] = (float)(xyz[n * 3 + 1] * w); o[2 * L] = (float)(xyz[n * 3 + 2] * w); o[3 * L] = (float)(xyz[n * 3] * w);
}
// out[n, o, a, i] = coef[n][a] * lab[n, o, 0, i] (A = 4): ADPIT labels [N, T*6, 4, C] with coef (1, x, y, z); ACCDOA labels
// [N, T, 4, C] with coef (0, x, y, z) (the reference leaves the activity block of the rewritten ACCDOA label at zero)
__global__ void spatial_label_kernel(const float* __restrict__ lab, float* __restrict__ out, const double* __restrict__ coef, long outer, long inner,
                                     long total) {
    const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= total) return;
    const long i = id % inner;
    long rest = id / inner;
    const int a = (int)(rest & 3);
    rest >>= 2;
    const long n = rest / outer;
    out[id] = (float)(coef[n * 4 + a] * (double)lab[(rest * 4) * inner + i]);
}
// EINV2: doa_out[n,t,0,:] = (sum over tracks and classes of sed[n,t]) * (x, y, z), tracks 1-2 zero; one wave per (n, t)
__global__ __launch_bounds__(256) void spatial_doa_label_kernel(const float* __restrict__ sed, const double* __restrict__ xyz, float* __restrict__ doa,
                                                                long T, int K, long rows) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float s = 0.f;
    for (int k = lane; k < K; k += 64) s += sed[row * K + k];
    s = wave_sum(s);
    if (lane < 9) doa[row * 9 + lane] = lane < 3 ? (float)((double)s * xyz[(row / T) * 3 + lane]) : 0.f;
}
}  // namespace

extern "C" int pseld_aug_rect_fill(float* x, const int* rects, int N, int C, int T, int F, int R, float value, void* stream) {
    PSELD_CHECK_ARG(x && rects && N > 0 && C > 0 && T > 0 && F > 0 && R > 0, "aug_rect_fill: bad argument");
    const long total = (long)N * C * T * F;
    hipLaunchKernelGGL(rect_fill_kernel, dim3(pseld_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, x, rects, T, F, R, value, total);
    PSELD_LAUNCH_CHECK("aug_rect_fill");
    return PSELD_OK;
}
extern "C" int pseld_aug_time_fill(float* y, const int* spans, int N, int Ty, long inner, int R, float value, void* stream) {
    PSELD_CHECK_ARG(y && spans && N > 0 && Ty > 0 && inner > 0 && R > 0, "aug_time_fill: bad argument");
    const long total = (long)N * Ty * inner;
    hipLaunchKernelGGL(time_fill_kernel, dim3(pseld_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, y, spans, Ty, inner, R, value, total);
    PSELD_LAUNCH_CHECK("aug_time_fill");
    return PSELD_OK;
}
extern "C" int pseld_aug_freqshift(const float* x, float* y, const int* shift, int N, int C, int T, int F, void* stream) {
    PSELD_CHECK_ARG(x && y && shift && x != y && N > 0 && C > 0 && T > 0 && F > 1, "aug_freqshift: bad argument");
    const long per = (long)C * T * F, total = per * N;
    hipLaunchKernelGGL(freqshift_kernel, dim3(pseld_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, x, y, shift, F, per, total);
    PSELD_LAUNCH_CHECK("aug_freqshift");
    return PSELD_OK;
}
extern "C" int pseld_spatialize_mono(const float* mono, long mono_stride, const double* xyz, float* foa, int N, long L, void* stream) {
    PSELD_CHECK_ARG(mono && xyz && foa && N > 0 && L > 0 && mono_stride >= L, "spatialize_mono: bad argument");
    const long total = (long)N * L;
    hipLaunchKernelGGL(spatialize_mono_kernel, dim3(pseld_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, mono, mono_stride, xyz, foa, L, total);
    PSELD_LAUNCH_CHECK("spatialize_mono");
    return PSELD_OK;
}
extern "C" int pseld_spatial_label(const float* lab, float* out, const double* coef, int N, long outer, long inner, void* stream) {
    PSELD_CHECK_ARG(lab && out && coef && lab != out && N > 0 && outer > 0 && inner > 0, "spatial_label: bad argument");
    const long total = (long)N * outer * 4 * inner;
    hipLaunchKernelGGL(spatial_label_kernel, dim3(pseld_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, lab, out, coef, outer, inner, total);
    PSELD_LAUNCH_CHECK("spatial_label");
    return PSELD_OK;
}
extern "C" int pseld_spatial_doa_label(const float* sed, const double* xyz, float* doa, int N, long T, int tracks, int C, void* stream) {
    PSELD_CHECK_ARG(sed && xyz && doa && N > 0 && T > 0 && tracks == 3 && C > 0, "spatial_doa_label: bad argument (3 tracks)");
    const long rows = (long)N * T;
    hipLaunchKernelGGL(spatial_doa_label_kernel, dim3(pseld_cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, sed, xyz, doa, T, tracks * C, rows);
    PSELD_LAUNCH_CHECK("spatial_doa_label");
    return PSELD_OK;
}
extern "C" int pseld_aug_rotate_wave(const float* x, float* y, const int* src, const float* sign, int N, long L, void* stream) {
    PSELD_CHECK_ARG(x && y && src && sign && x != y && N > 0 && L > 0 && L % 4 == 0, "aug_rotate_wave: bad argument (L % 4 == 0)");
    const long total4 = (long)N * 4 * (L / 4);
    hipLaunchKernelGGL(rotate_wave_kernel, dim3(pseld_cdiv(total4, 256)), dim3(256), 0, (hipStream_t)stream, x, y, src, sign, L / 4, total4);
    PSELD_LAUNCH_CHECK("aug_rotate_wave");
    return PSELD_OK;
}
extern "C" int pseld_aug_rotate_label(const float* x, float* y, const int* src, const float* sign, int N, long outer, int A, long inner,
                                      int a0, void* stream) {
    PSELD_CHECK_ARG(x && y && src && sign && x != y && N > 0 && outer > 0 && A >= a0 + 3 && a0 >= 0 && inner > 0, "aug_rotate_label: bad argument");
    const long total = (long)N * outer * A * inner;
    hipLaunchKernelGGL(rotate_label_kernel, dim3(pseld_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, x, y, src, sign, outer, A, inner, a0, total);
    PSELD_LAUNCH_CHECK("aug_rotate_label");
    return PSELD_OK;
}
extern "C" int pseld_aug_mix(const float* x, float* y, const int* dst, const int* src, const float* lam, int P, long elems, void* stream) {
    PSELD_CHECK_ARG(x && y && dst && src && lam && x != y && P > 0 && elems > 0, "aug_mix: bad argument");
    const long total = (long)P * elems;
    hipLaunchKernelGGL(mix_kernel, dim3(pseld_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, x, y, dst, src, lam, elems, total);
    PSELD_LAUNCH_CHECK("aug_mix");
    return PSELD_OK;
}
extern "C" int pseld_aug_mix_adpit(const float* lab, float* out, const int* dst, const int* src, const float* lam, int P, int T, int C, int mode,
                                   void* stream) {
    PSELD_CHECK_ARG(lab && out && dst && src && lam && lab != out && P > 0 && T > 0 && C > 0 && (mode == 1 || mode == 2), "aug_mix_adpit: bad argument");
    hipLaunchKernelGGL(mix_adpit_kernel, dim3(P * T), dim3(256), 0, (hipStream_t)stream, lab, out, dst, src, lam, T, C, mode);
    PSELD_LAUNCH_CHECK("aug_mix_adpit");
    return PSELD_OK;
}
extern "C" int pseld_aug_mix_tracks(const float* sed, const float* doa, float* sed_out, float* doa_out, const int* dst, const int* src,
                                    const float* lam, int P, int T, int C, int wavmix, void* stream) {
    PSELD_CHECK_ARG(sed && doa && sed_out && doa_out && dst && src && lam && sed != sed_out && doa != doa_out && P > 0 && T > 0 && C > 0,
                    "aug_mix_tracks: bad argument");
    const long total = (long)P * T * (C + 3);
    hipLaunchKernelGGL(mix_tracks_kernel, dim3(pseld_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, sed, doa, sed_out, doa_out, dst, src, lam,
                       T, C, wavmix, total);
    PSELD_LAUNCH_CHECK("aug_mix_tracks");
    return PSELD_OK;
}
