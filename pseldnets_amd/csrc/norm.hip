// LayerNorm (plain and patch-merging gather form) forward/backward, and the 7 "scalar" BatchNorms fused with
// the HTS-AT time->frequency fold and the 4x4 patch extraction.
//
// Replaces (reference, /root/reference/src):
//   nn.LayerNorm calls at models/components/htsat.py:234 (norm1), :261 (norm2), :525 (final norm),
//   model_utilities.py:212 (PatchEmbed.norm); htsat.py:290-311 PatchMerging.forward (2x2 gather + LN(4C));
//   models/accdoa.py:223-227 (per-input-channel BatchNorm2d written in place), htsat.py:493-511
//   (reshape_wav2img: zero-pad T->1024, fold time into frequency) and the im2col of PatchEmbed's Conv2d(k=s=4).
// All of them are HBM-bound streaming kernels: 16-byte vector loads/stores, wave-shuffle reductions, fp32 math.
#include "common.h"

void pseld_reduce_slabs(const float* slabs, float* out, long n, int splits, long slab_stride, int accumulate,
                        hipStream_t stream);

namespace {

// ---------------------------------------------------------------------------------------------------------
// LayerNorm. A row of C elements (C % 8 == 0, C <= 64*8*NV) is owned by G lanes; lane g holds the 8-element
// vectors v*G + g. MERGE: the row is the concatenation of 4 tokens of a [B, R, R, Cs] grid (Cs = C/4):
// segment s -> token (2i + (s&1), 2j + (s>>1))  (x0,x1,x2,x3 of htsat.py:302-306).
struct LnArgs {
    const void* x; void* y; const float* gamma; const float* beta;
    float* mean; float* rstd;            // optional [M]
    const void* dy; void* dx; const void* dres; float* partial;  // backward
    long M; int C; int res;              // res: side of the INPUT token grid in MERGE mode
    float eps;
};

template <bool MERGE>
__device__ __forceinline__ long row_elem_offset(const LnArgs& a, long row, int e) {
    if (!MERGE) return row * a.C + e;
    const int Cs = a.C >> 2;
    const int half = a.res >> 1;
    const long b = row / (half * half);
    const int ij = (int)(row - b * half * half);
    const int i = ij / half, j = ij - i * half;
    const int s = e / Cs, off = e - s * Cs;
    const long tok = (b * a.res + (2 * i + (s & 1))) * a.res + (2 * j + (s >> 1));
    return tok * Cs + off;
}

template <typename T, int G, int NV, bool MERGE>
__global__ __launch_bounds__(256) void ln_fwd_kernel(LnArgs a) {
    constexpr int GROUPS = 256 / G;
    const int g = threadIdx.x % G, grp = threadIdx.x / G;
    const long row = (long)blockIdx.x * GROUPS + grp;
    if (row >= a.M) return;
    const T* x = (const T*)a.x;
    float v[NV][8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int e = (i * G + g) * 8;
        if (e < a.C) {
            load8<T>(x + row_elem_offset<MERGE>(a, row, e), v[i]);
#pragma unroll
            for (int k = 0; k < 8; ++k) s += v[i][k];
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) v[i][k] = 0.f;
        }
    }
    const float mean = group_sum<G>(s) / a.C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int e = (i * G + g) * 8;
        if (e < a.C)
#pragma unroll
            for (int k = 0; k < 8; ++k) { const float d = v[i][k] - mean; q += d * d; }
    }
    const float rstd = rsqrtf(group_sum<G>(q) / a.C + a.eps);
    if (g == 0 && a.mean) { a.mean[row] = mean; a.rstd[row] = rstd; }
    T* y = (T*)a.y;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int e = (i * G + g) * 8;
        if (e < a.C) {
            float o[8], ga[8], be[8];
            load8<float>(a.gamma + e, ga);
            load8<float>(a.beta + e, be);
#pragma unroll
            for (int k = 0; k < 8; ++k) o[k] = (v[i][k] - mean) * rstd * ga[k] + be[k];
            store8<T>(y + row * a.C + e, o);
        }
    }
}

// Backward: dx = rstd * (dyh - mean(dyh) - xh * mean(dyh*xh)) (+ dres), dyh = dy*gamma; per-block partial
// sums of dgamma = sum dy*xh and dbeta = sum dy go to partial[block][2][C] (reduced by pseld_reduce_slabs).
template <typename T, int G, int NV, bool MERGE>
__global__ __launch_bounds__(256) void ln_bwd_kernel(LnArgs a, int rows_per_block) {
    constexpr int GROUPS = 256 / G;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* red = (float*)smem;  // [GROUPS][2][C]
    const int g = threadIdx.x % G, grp = threadIdx.x / G;
    const T* x = (const T*)a.x;
    const T* dy = (const T*)a.dy;
    const T* dres = (const T*)a.dres;
    T* dx = (T*)a.dx;
    float dg[NV][8], db[NV][8], ga[NV][8];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int e = (i * G + g) * 8;
#pragma unroll
        for (int k = 0; k < 8; ++k) { dg[i][k] = 0.f; db[i][k] = 0.f; ga[i][k] = 0.f; }
        if (e < a.C) load8<float>(a.gamma + e, ga[i]);
    }
    const long rbeg = (long)blockIdx.x * rows_per_block;
    const long rend = min(a.M, rbeg + rows_per_block);
    // The rows a group owns form one dependent chain each (load -> mean -> variance -> two projections -> store), so
    // the NEXT row's x / dy / dres vectors are requested (still packed) before the current row is reduced.
    Vec8<T> nx[NV] = {}, nd[NV] = {}, nr[NV] = {};
    auto fetch = [&](long row) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int e = (i * G + g) * 8;
            if (e < a.C) {
                const long off = row_elem_offset<MERGE>(a, row, e);
                nx[i] = *(const Vec8<T>*)(x + off);
                nd[i] = *(const Vec8<T>*)(dy + row * a.C + e);
                if (dres) nr[i] = *(const Vec8<T>*)(dres + off);
            }
        }
    };
    if (rbeg + grp < rend) fetch(rbeg + grp);
    for (long row = rbeg + grp; row < rend; row += GROUPS) {
        float v[NV][8], d[NV][8], rs[NV][8];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int e = (i * G + g) * 8;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const bool in = e < a.C;
                v[i][k] = in ? nx[i].get(k) : 0.f;
                d[i][k] = in ? nd[i].get(k) : 0.f;
                rs[i][k] = (in && dres) ? nr[i].get(k) : 0.f;
                s += v[i][k];
            }
        }
        if (row + GROUPS < rend) fetch(row + GROUPS);
        const float mean = group_sum<G>(s) / a.C;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int e = (i * G + g) * 8;
            if (e < a.C)
#pragma unroll
                for (int k = 0; k < 8; ++k) { const float t = v[i][k] - mean; q += t * t; }
        }
        const float rstd = rsqrtf(group_sum<G>(q) / a.C + a.eps);
        float c1 = 0.f, c2 = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int e = (i * G + g) * 8;
            if (e < a.C)
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const float xh = (v[i][k] - mean) * rstd;
                    const float dyh = d[i][k] * ga[i][k];
                    dg[i][k] += d[i][k] * xh;
                    db[i][k] += d[i][k];
                    c1 += dyh;
                    c2 += dyh * xh;
                    v[i][k] = xh;
                    d[i][k] = dyh;
                }
        }
        c1 = group_sum<G>(c1) / a.C;
        c2 = group_sum<G>(c2) / a.C;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int e = (i * G + g) * 8;
            if (e < a.C) {
                float o[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) o[k] = rstd * (d[i][k] - c1 - v[i][k] * c2);
                const long off = row_elem_offset<MERGE>(a, row, e);
#pragma unroll
                for (int k = 0; k < 8; ++k) o[k] += rs[i][k];
                store8<T>(dx + off, o);
            }
        }
    }
    // cross-group reduction of dgamma / dbeta through LDS
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int e = (i * G + g) * 8;
        if (e < a.C)
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                red[(grp * 2 + 0) * a.C + e + k] = dg[i][k];
                red[(grp * 2 + 1) * a.C + e + k] = db[i][k];
            }
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < 2 * a.C; idx += 256) {
        float s = 0.f;
        for (int gi = 0; gi < GROUPS; ++gi) s += red[gi * 2 * a.C + idx];
        a.partial[(long)blockIdx.x * 2 * a.C + idx] = s;
    }
}

template <typename T, bool MERGE>
int launch_ln(const LnArgs& a, bool bwd, int rows_per_block, int nblocks, hipStream_t s) {
#define LN_CASE(G, NV)                                                                                          \
    do {                                                                                                        \
        if (!bwd) {                                                                                             \
            hipLaunchKernelGGL((ln_fwd_kernel<T, G, NV, MERGE>), dim3(pseld_cdiv(a.M, 256 / G)), dim3(256), 0, s, a); \
        } else {                                                                                                \
            const size_t lds = (size_t)(256 / G) * 2 * a.C * sizeof(float);                                     \
            hipLaunchKernelGGL((ln_bwd_kernel<T, G, NV, MERGE>), dim3(nblocks), dim3(256), lds, s, a, rows_per_block); \
        }                                                                                                       \
    } while (0)
    // widths that are 24 x a power of two (every HTS-AT stage: 96 / 192 / 384 / 768, PaSST's 768): G = C / 24 lanes x 3 vectors of 8 cover the
    // row exactly - the power-of-two groups below leave a quarter of their lanes without a vector at these widths. Round 5 (tools/ln_shapes.py,
    // same box): forward 70 / 35.5 / 19.8 / 11.7 -> 63 / 31.4 / 17.9 / 10.5 us at stages 0-3, PaSST 76.5 -> 68; backward 167 -> 150 us at PaSST's
    // 115 584 rows, unchanged at stages 0-2, 27.8 -> 29.6 at stage 3 (12 288 rows of 768: stays on 64 lanes); the step 17.71 -> 17.58 ms.
    // Knob LN_EXACT = 0: the old mapping (A/B).
    const int g24 = a.C % 24 == 0 ? a.C / 24 : 0;
    if (pseld_knob(KNOB_LN_EXACT, 1) != 0 && (g24 == 4 || g24 == 8 || g24 == 16 || g24 == 32) && !(bwd && g24 == 32 && a.M < 65536)) {
        if (g24 == 4) LN_CASE(4, 3); else if (g24 == 8) LN_CASE(8, 3); else if (g24 == 16) LN_CASE(16, 3); else LN_CASE(32, 3);
    } else
    if (a.C <= 128) LN_CASE(16, 1);
    else if (a.C <= 256) LN_CASE(32, 1);
    else if (a.C <= 512) LN_CASE(64, 1);
    else if (a.C <= 1024) LN_CASE(64, 2);
    else if (a.C <= 1536) LN_CASE(64, 3);
    else if (a.C <= 2048) LN_CASE(64, 4);                 // the CNN14-Conformer decoder width
    else { pseld_set_error("layernorm: C=%d > 2048 not built", a.C); return PSELD_ERR_UNSUPPORTED; }
#undef LN_CASE
    PSELD_LAUNCH_CHECK("layernorm");
    return PSELD_OK;
}

// rows per workgroup of the backward: each row group walks its rows serially (load -> two reductions -> store is one dependent chain per
// row), and every workgroup ends with the same fixed work (its d(gamma) / d(beta) sums through LDS, one partial row written). Alone, fewer
// and longer workgroups win (tools/ln_shapes.py with a grid knob, round 5: 256 workgroups 130 / 64.7 / 37.3 us at stages 0-2 against 132 /
// 68.2 / 42.2 with 1024); INSIDE the step they lose (17.72 / 17.75 -> 17.88 / 17.86 ms, alternating on one box): the weight gradients of
// the second stream hold CUs, and a grid of exactly one long workgroup per CU waits for them. 1024 it stays.
constexpr int LN_BWD_MAX_BLOCKS = 1024;
static inline int ln_bwd_rows(long M, int) { long r = (M + LN_BWD_MAX_BLOCKS - 1) / LN_BWD_MAX_BLOCKS; if (r < 16) r = 16; return (int)r; }

// ---------------------------------------------------------------------------------------------------------
// Scalar BatchNorm statistics: sums[c][f][0..1] = (sum x, sum x^2) over (b, t) of feat[B, Cin, T, F]
// 16 lanes cover one 64-bin row with 16-byte loads, so a workgroup iteration moves 16 rows (4 KiB); 4 iterations are
// kept in flight. MODE 0: (sum x, sum x^2); MODE 1: centred second pass, sum (x - mean)^2.
template <int MODE>
__global__ __launch_bounds__(256) void bn_rows_kernel(const float* __restrict__ feat, const float* __restrict__ sums, float count,
                                                      float* __restrict__ part, int B, int Cin, int T, int rows_per_block) {
    __shared__ float red[16][64][2];
    const int fq = threadIdx.x & 15, rr = threadIdx.x >> 4, c = blockIdx.y;
    const long rows = (long)B * T;
    const long rbeg = (long)blockIdx.x * rows_per_block;
    const long rend = min(rows, rbeg + rows_per_block);
    f32x4 mean = {0.f, 0.f, 0.f, 0.f};
    if (MODE == 1) {
#pragma unroll
        for (int k = 0; k < 4; ++k) mean[k] = sums[2 * (c * 64 + 4 * fq + k)] / count;
    }
    f32x4 s = {0.f, 0.f, 0.f, 0.f}, q = {0.f, 0.f, 0.f, 0.f};
    // row -> (sample, frame) through the reciprocal (exact after one fix-up step for rows < 2^24; a 64-bit division per load cost more
    // than the load: 64 us for the 57 MB of a 32-chunk batch, latency of the division chain, not of memory)
    const float rT = 1.0f / (float)T;
    for (long r0 = rbeg + rr; r0 < rend; r0 += 64) {
        f32x4 v[4];
        bool live[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long r = r0 + 16 * u;
            live[u] = r < rend;
            const int rc = (int)(live[u] ? r : rbeg);
            int b = (int)((float)rc * rT);
            int t = rc - b * T;
            b += (t >= T) - (t < 0);
            t = rc - b * T;
            v[u] = *(const f32x4*)(feat + (((long)b * Cin + c) * T + t) * 64 + 4 * fq);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (!live[u]) continue;
            if (MODE == 0) { s += v[u]; q += v[u] * v[u]; }
            else { const f32x4 d = v[u] - mean; q += d * d; }
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) { red[rr][4 * fq + k][0] = s[k]; red[rr][4 * fq + k][1] = q[k]; }
    __syncthreads();
    if (threadIdx.x < 64) {
        const int f = threadIdx.x;
        float a0 = 0.f, a1 = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) { a0 += red[i][f][0]; a1 += red[i][f][1]; }
        if (MODE == 0) {
            float* o = part + (((long)blockIdx.x * Cin + c) * 64 + f) * 2;
            o[0] = a0; o[1] = a1;
        } else {
            part[((long)blockIdx.x * Cin + c) * 64 + f] = a1;
        }
    }
}

// sums [Cin*F][2] (already reduced over every rank when sync-BN) -> scale/shift for the forward, saved
// mean/rstd for the backward, running statistics (momentum, unbiased variance; nn.BatchNorm2d semantics).
__global__ void bn_finalize_kernel(const float* __restrict__ sums, float count, const float* __restrict__ weight,
                                   const float* __restrict__ bias, float* running_mean, float* running_var,
                                   long long* num_batches, float* __restrict__ mean_rstd, float* __restrict__ scale_shift,
                                   int n, int F, float momentum, float eps, int training) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float mean, var;
    if (training) {
        mean = sums[2 * i] / count;
        var = fmaxf(sums[2 * i + 1] / count - mean * mean, 0.f);
        running_mean[i] = (1.f - momentum) * running_mean[i] + momentum * mean;
        running_var[i] = (1.f - momentum) * running_var[i] + momentum * var * (count / (count - 1.f));
        if (i % F == 0) num_batches[i / F] += 1;
    } else {
        mean = running_mean[i]; var = running_var[i];
    }
    const float rstd = rsqrtf(var + eps);
    mean_rstd[2 * i] = mean; mean_rstd[2 * i + 1] = rstd;
    scale_shift[2 * i] = weight[i] * rstd;
    scale_shift[2 * i + 1] = bias[i] - mean * rstd * weight[i];
}

// BN apply + zero-pad + fold + 4x4 patch extraction: A[(b, ph, pw)][c*16 + i*4 + j] = bn(feat[b,c,t,f]) with
// r = ph/16, f = 4*(ph%16)+i, t = 256*r + 4*pw + j (0 when t >= T). One wave handles (b, r, pw, c): lane = f.
template <typename T>
__global__ __launch_bounds__(256) void patchify_kernel(const float* __restrict__ feat, const float* __restrict__ scale_shift,
                                                       T* __restrict__ A, int B, int Cin, int c_first, int Cuse, int Tn,
                                                       long total_waves) {
    const long wid = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wid >= total_waves) return;
    const int f = threadIdx.x & 63;
    const int c = (int)(wid % Cuse);
    long rest = wid / Cuse;
    const int pw = (int)(rest % 64); rest /= 64;
    const int r = (int)(rest % 4);
    const long b = rest / 4;
    const int cs = c_first + c;
    const float sc = scale_shift[2 * (cs * 64 + f)], sh = scale_shift[2 * (cs * 64 + f) + 1];
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int t = 256 * r + 4 * pw + j;
        v[j] = (t < Tn) ? feat[((b * Cin + cs) * Tn + t) * 64 + f] * sc + sh : 0.f;
    }
    const int ph = 16 * r + (f >> 2), i = f & 3;
    T* dst = A + ((b * 64 + ph) * 64 + pw) * (long)(Cuse * 16) + c * 16 + i * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) dst[j] = from_f32<T>(v[j]);
}

// BN parameter gradients from dA (gradient of the patch matrix): dweight[c,f] = sum dy*xhat, dbias = sum dy.
// A thread owns one 16-byte chunk position of the dA rows (k -> channel c = k/2 and the patch-row pair i0 = 2*(k&1))
// for one frequency group fh = ph % 16, i.e. the two mel bins f = 4*fh + i0 + {0,1} of channel c: every (c, f) has
// exactly one owner, so the sums stay in registers and no cross-thread reduction is needed. Lanes with equal fh read
// one whole contiguous dA row; lanes with equal c read whole 256-byte feature rows.
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_kernel(const float* __restrict__ feat, const float* __restrict__ mean_rstd,
                                                     const T* __restrict__ dA, float* __restrict__ part, int B, int Cin,
                                                     int c_first, int Cuse, int Tn, int cols_per_block) {
    const int cpr = Cuse * 2;                                  // 8-element chunks per dA row
    const int fh = threadIdx.x & 15, k = threadIdx.x >> 4;     // 16 chunk slots per workgroup pass
    const long total = (long)B * 4 * 64;                      // (b, r, pw) triples
    const long beg = (long)blockIdx.x * cols_per_block, end = min(total, beg + cols_per_block);
    for (int kk = k; kk < cpr; kk += 16) {
        const int c = kk >> 1, i0 = (kk & 1) * 2, cs = c_first + c, f0 = 4 * fh + i0;
        const float m0 = mean_rstd[2 * (cs * 64 + f0)], r0 = mean_rstd[2 * (cs * 64 + f0) + 1];
        const float m1 = mean_rstd[2 * (cs * 64 + f0 + 1)], r1 = mean_rstd[2 * (cs * 64 + f0 + 1) + 1];
        float dw0 = 0.f, dw1 = 0.f, db0 = 0.f, db1 = 0.f;
#pragma unroll 4
        for (long q = beg; q < end; ++q) {
            const int pw = (int)(q & 63), r = (int)((q >> 6) & 3);
            const long b = q >> 8;
            float g[8];
            load8<T>(dA + ((b * 64 + 16 * r + fh) * 64 + pw) * (long)(Cuse * 16) + kk * 8, g);
            const int t0 = 256 * r + 4 * pw;
            const float* src = feat + ((b * Cin + cs) * Tn) * 64 + f0;
            f32x2 x[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) x[j] = *(const f32x2*)(src + (long)min(t0 + j, Tn - 1) * 64);   // unconditional
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float live = (t0 + j < Tn) ? 1.f : 0.f;      // rows past T are the zero padding of the fold
                const float ga = g[j] * live, gb = g[4 + j] * live;
                dw0 += ga * (x[j][0] - m0); db0 += ga;
                dw1 += gb * (x[j][1] - m1); db1 += gb;
            }
        }
        float* o = part + (((long)blockIdx.x * Cuse + c) * 64 + f0) * 2;
        o[0] = dw0 * r0; o[1] = db0; o[2] = dw1 * r1; o[3] = db1;
    }
}

__global__ void bn_bwd_finish_kernel(const float* __restrict__ part, int nblocks, int n, float* dweight, float* dbias,
                                     int accumulate) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float dw = 0.f, db = 0.f;
    for (int k = 0; k < nblocks; ++k) { dw += part[((long)k * n + i) * 2]; db += part[((long)k * n + i) * 2 + 1]; }
    if (accumulate) { dweight[i] += dw; dbias[i] += db; } else { dweight[i] = dw; dbias[i] = db; }
}

__global__ void reduce_slabs_kernel(const float* __restrict__ slabs, float* __restrict__ out, long n, int splits,
                                    long slab_stride, int accumulate) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s = accumulate ? out[i] : 0.f;
    for (int z = 0; z < splits; ++z) s += slabs[z * slab_stride + i];
    out[i] = s;
}
// many slabs, few columns: CW columns x (256 / CW) slab groups per workgroup, fixed-order sum in LDS (deterministic).
// CW = 16 for up to a few dozen slabs; CW = 4 (64 slab groups, four times the workgroups) for the per-row-block partials of
// the LayerNorm / BatchNorm backward passes, where a launch used to be 12 workgroups each walking 64 slabs serially.
template <int CW>
__global__ __launch_bounds__(256) void reduce_slabs_wide_kernel(const float* __restrict__ slabs, float* __restrict__ out,
                                                                long n, int splits, long slab_stride, int accumulate) {
    constexpr int SG = 256 / CW;
    __shared__ float red[SG][CW + 1];
    const int c = threadIdx.x % CW, sg = threadIdx.x / CW;
    const long i = (long)blockIdx.x * CW + c;
    float s = 0.f;
    if (i < n)
        for (int z = sg; z < splits; z += SG) s += slabs[z * slab_stride + i];
    red[sg][c] = s;
    __syncthreads();
    if (sg == 0 && i < n) {
        float t = accumulate ? out[i] : 0.f;
#pragma unroll 16
        for (int k = 0; k < SG; ++k) t += red[k][c];
        out[i] = t;
    }
}

// many slabs, many columns (split-K weight gradients): 64 float4 columns x 4 slab groups per workgroup, so every
// wave-load is 1 KiB contiguous; the 4 partial sums are combined in fixed order (deterministic)
__global__ __launch_bounds__(256) void reduce_slabs_vec4_kernel(const float* __restrict__ slabs, float* __restrict__ out,
                                                                long n4, int splits, long slab_stride, int accumulate) {
    __shared__ f32x4 red[4][64];
    const int c = threadIdx.x & 63, sg = threadIdx.x >> 6;
    const long i = (long)blockIdx.x * 64 + c;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (i < n4) {
        const f32x4* p = (const f32x4*)slabs + i;
        const long st4 = slab_stride >> 2;
#pragma unroll 4
        for (int z = sg; z < splits; z += 4) s += p[z * st4];
    }
    red[sg][c] = s;
    __syncthreads();
    if (sg == 0 && i < n4) {
        f32x4 t = red[0][c];
        t += red[1][c]; t += red[2][c]; t += red[3][c];
        f32x4* o = (f32x4*)out + i;
        if (accumulate) t += *o;
        *o = t;
    }
}

template <typename T>
__global__ void add_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ y, long n8) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n8) return;
    float u[8], v[8];
    load8<T>(a + i * 8, u);
    load8<T>(b + i * 8, v);
#pragma unroll
    for (int k = 0; k < 8; ++k) u[k] += v[k];
    store8<T>(y + i * 8, u);
}

template <typename T>
__global__ void rowscale_kernel(const T* __restrict__ x, const float* __restrict__ scale, T* __restrict__ y, long n8,
                                long elems_per_scale) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n8) return;
    float v[8];
    load8<T>(x + i * 8, v);
    const float s = scale[(i * 8) / elems_per_scale];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] *= s;
    store8<T>(y + i * 8, v);
}

// ---------------------------------------------------------------------------------------------------------
// CrossStitch (model_utilities.py:35-54): x' = w00*x + w01*y ; y' = w10*x' + w11*y  (y' uses the UPDATED x').
// w f32[C,2,2]; tokens [M, C].
template <typename T>
__global__ void cross_stitch_fwd_kernel(const T* __restrict__ x, const T* __restrict__ y, const float* __restrict__ w,
                                        T* __restrict__ xo, T* __restrict__ yo, long n8, int C) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n8) return;
    const int c0 = (int)((i * 8) % C);
    float a[8], b[8], oa[8], ob[8];
    load8<T>(x + i * 8, a);
    load8<T>(y + i * 8, b);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const f32x4 ww = *(const f32x4*)(w + (c0 + k) * 4);
        oa[k] = ww[0] * a[k] + ww[1] * b[k];
        ob[k] = ww[2] * oa[k] + ww[3] * b[k];
    }
    store8<T>(xo + i * 8, oa);
    store8<T>(yo + i * 8, ob);
}
// backward: dx, dy and per-block partial sums of dw [C][4] -> partial[block][C*4]
template <typename T>
__global__ __launch_bounds__(256) void cross_stitch_bwd_kernel(const T* __restrict__ x, const T* __restrict__ y,
                                                               const float* __restrict__ w, const T* __restrict__ dxo,
                                                               const T* __restrict__ dyo, T* __restrict__ dx, T* __restrict__ dy,
                                                               float* __restrict__ partial, long M, int C, int rows_per_block) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* red = (float*)smem;                       // [RPI][C*4]
    const int CG = C >> 3;                           // 8-channel groups per row
    const int RPI = 256 / CG;                        // rows handled per iteration
    const int cg = threadIdx.x % CG, r0 = threadIdx.x / CG;
    float acc[8][4];
#pragma unroll
    for (int k = 0; k < 8; ++k)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[k][j] = 0.f;
    if (r0 < RPI) {
        f32x4 ww[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) ww[k] = *(const f32x4*)(w + (cg * 8 + k) * 4);
        const long rbeg = (long)blockIdx.x * rows_per_block;
        const long rend = min(M, rbeg + rows_per_block);
        for (long row = rbeg + r0; row < rend; row += RPI) {
            const long off = row * C + cg * 8;
            float a[8], b[8], ga[8], gb[8], oa[8], ob[8];
            load8<T>(x + off, a); load8<T>(y + off, b); load8<T>(dxo + off, ga); load8<T>(dyo + off, gb);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float xp = ww[k][0] * a[k] + ww[k][1] * b[k];     // x'
                const float gxp = ga[k] + ww[k][2] * gb[k];             // total gradient reaching x'
                oa[k] = ww[k][0] * gxp;
                ob[k] = ww[k][1] * gxp + ww[k][3] * gb[k];
                acc[k][0] += gxp * a[k]; acc[k][1] += gxp * b[k];
                acc[k][2] += gb[k] * xp; acc[k][3] += gb[k] * b[k];
            }
            store8<T>(dx + off, oa);
            store8<T>(dy + off, ob);
        }
#pragma unroll
        for (int k = 0; k < 8; ++k)
#pragma unroll
            for (int j = 0; j < 4; ++j) red[(r0 * C + cg * 8 + k) * 4 + j] = acc[k][j];
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < C * 4; idx += 256) {
        float s = 0.f;
        for (int rr = 0; rr < RPI; ++rr) s += red[rr * C * 4 + idx];
        partial[(long)blockIdx.x * C * 4 + idx] = s;
    }
}

// up to 32 independent slab reductions in one launch (blockIdx.y = entry): the column-sum partials of several LayerNorm backward
// passes are reduced together at the end of a stage instead of by one small launch each on the dependent chain. Same algorithm
// (and summation order) as reduce_slabs_wide_kernel<4>.
struct ReduceBatch {
    const float* src[32];
    float* dst[32];
    int n[32], splits[32], stride[32];
    int accumulate;
};
__global__ __launch_bounds__(256) void reduce_slabs_batched_kernel(ReduceBatch b) {
    constexpr int CW = 4, SG = 256 / CW;
    __shared__ float red[SG][CW + 1];
    const int e = blockIdx.y;
    const int n = b.n[e];
    if ((int)blockIdx.x * CW >= n) return;
    const float* __restrict__ slabs = b.src[e];
    float* __restrict__ out = b.dst[e];
    const int splits = b.splits[e];
    const long slab_stride = b.stride[e];
    const int c = threadIdx.x % CW, sg = threadIdx.x / CW;
    const long i = (long)blockIdx.x * CW + c;
    float s = 0.f;
    if (i < n)
        for (int z = sg; z < splits; z += SG) s += slabs[z * slab_stride + i];
    red[sg][c] = s;
    __syncthreads();
    if (sg == 0 && i < n) {
        float t = b.accumulate ? out[i] : 0.f;
#pragma unroll 16
        for (int k = 0; k < SG; ++k) t += red[k][c];
        out[i] = t;
    }
}

}  // namespace

// count <= 32 reductions dst[e][0..n[e]) (+)= sum over splits[e] slabs of src[e] (slab stride stride[e] floats); host arrays.
extern "C" int pseld_reduce_slabs_batched(const void* const* src, void* const* dst, const int* n, const int* splits, const int* stride,
                                          int count, int accumulate, void* stream) {
    PSELD_CHECK_ARG(src && dst && n && splits && stride && count > 0 && count <= 32, "reduce_slabs_batched: bad arguments (count %d)", count);
    ReduceBatch b;
    memset(&b, 0, sizeof(b));
    int max_n = 0;
    for (int e = 0; e < count; ++e) {
        PSELD_CHECK_ARG(src[e] && dst[e] && n[e] > 0 && splits[e] > 0, "reduce_slabs_batched: bad entry %d", e);
        b.src[e] = (const float*)src[e]; b.dst[e] = (float*)dst[e]; b.n[e] = n[e]; b.splits[e] = splits[e]; b.stride[e] = stride[e];
        if (n[e] > max_n) max_n = n[e];
    }
    b.accumulate = accumulate;
    hipLaunchKernelGGL(reduce_slabs_batched_kernel, dim3(pseld_cdiv(max_n, 4), count), dim3(256), 0, (hipStream_t)stream, b);
    PSELD_LAUNCH_CHECK("reduce_slabs_batched");
    return PSELD_OK;
}

void pseld_reduce_slabs(const float* slabs, float* out, long n, int splits, long slab_stride, int accumulate,
                        hipStream_t stream) {
    if (splits >= 4 && n >= 4096 && n % 4 == 0 && slab_stride % 4 == 0 && ((unsigned long)slabs & 15) == 0 && ((unsigned long)out & 15) == 0)
        hipLaunchKernelGGL(reduce_slabs_vec4_kernel, dim3(pseld_cdiv(n / 4, 64)), dim3(256), 0, stream, slabs, out, n / 4, splits,
                           slab_stride, accumulate);
    else if (splits > 128)
        hipLaunchKernelGGL(reduce_slabs_wide_kernel<4>, dim3(pseld_cdiv(n, 4)), dim3(256), 0, stream, slabs, out, n, splits,
                           slab_stride, accumulate);
    else if (splits > 32)
        hipLaunchKernelGGL(reduce_slabs_wide_kernel<16>, dim3(pseld_cdiv(n, 16)), dim3(256), 0, stream, slabs, out, n, splits,
                           slab_stride, accumulate);
    else
        hipLaunchKernelGGL(reduce_slabs_kernel, dim3(pseld_cdiv(n, 256)), dim3(256), 0, stream, slabs, out, n, splits,
                           slab_stride, accumulate);
}

extern "C" int pseld_layernorm_fwd(int dtype, const void* x, const float* gamma, const float* beta, void* y,
                                   float* mean, float* rstd, long M, int C, int merge_res, float eps, void* stream) {
    PSELD_CHECK_ARG(x && gamma && beta && y, "layernorm_fwd: null pointer");
    PSELD_CHECK_ARG(M > 0 && C > 0 && C % 8 == 0, "layernorm_fwd: bad M/C (%ld, %d)", M, C);
    PSELD_CHECK_ARG(merge_res == 0 || (merge_res % 2 == 0 && C % 32 == 0), "layernorm_fwd: bad merge geometry");
    LnArgs a; memset(&a, 0, sizeof(a));
    a.x = x; a.y = y; a.gamma = gamma; a.beta = beta; a.mean = mean; a.rstd = rstd; a.M = M; a.C = C; a.res = merge_res; a.eps = eps;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == PSELD_BF16) return merge_res ? launch_ln<bf16_t, true>(a, false, 0, 0, s) : launch_ln<bf16_t, false>(a, false, 0, 0, s);
    if (dtype == PSELD_F32) return merge_res ? launch_ln<float, true>(a, false, 0, 0, s) : launch_ln<float, false>(a, false, 0, 0, s);
    pseld_set_error("layernorm_fwd: unknown dtype %d", dtype);
    return PSELD_ERR_BAD_ARG;
}

extern "C" long pseld_layernorm_bwd_workspace(long M, int C) {
    return (long)pseld_cdiv(M, ln_bwd_rows(M, C)) * 2 * C * (long)sizeof(float);
}

// dx = LN'(dy) (+ dres); dgamma/dbeta (fp32, overwritten or accumulated). In merge mode x/dx use the un-merged
// token grid [B, res, res, C/4] and dy the merged rows [M, C].
extern "C" int pseld_layernorm_bwd(int dtype, const void* dy, const void* x, const float* gamma, const void* dres,
                                   void* dx, float* dgamma, float* dbeta, long M, int C, int merge_res, float eps,
                                   int accumulate, float* workspace, long workspace_bytes, void* stream) {
    PSELD_CHECK_ARG(dy && x && gamma && dx && dgamma && dbeta && workspace, "layernorm_bwd: null pointer");
    PSELD_CHECK_ARG(M > 0 && C > 0 && C % 8 == 0, "layernorm_bwd: bad M/C");
    PSELD_CHECK_ARG(!(merge_res && dres), "layernorm_bwd: merge mode has no residual gradient");
    PSELD_CHECK_ARG(workspace_bytes >= pseld_layernorm_bwd_workspace(M, C), "layernorm_bwd: workspace too small");
    LnArgs a; memset(&a, 0, sizeof(a));
    a.x = x; a.dy = dy; a.dx = dx; a.dres = dres; a.gamma = gamma; a.partial = workspace; a.M = M; a.C = C; a.res = merge_res; a.eps = eps;
    const int rows = ln_bwd_rows(M, C);
    const int nb = pseld_cdiv(M, rows);
    hipStream_t s = (hipStream_t)stream;
    int rc;
    if (dtype == PSELD_BF16) rc = merge_res ? launch_ln<bf16_t, true>(a, true, rows, nb, s) : launch_ln<bf16_t, false>(a, true, rows, nb, s);
    else if (dtype == PSELD_F32) rc = merge_res ? launch_ln<float, true>(a, true, rows, nb, s) : launch_ln<float, false>(a, true, rows, nb, s);
    else { pseld_set_error("layernorm_bwd: unknown dtype %d", dtype); return PSELD_ERR_BAD_ARG; }
    if (rc != PSELD_OK) return rc;
    if (accumulate & 2) return PSELD_OK;      // deferred: the caller reduces the [nb][2][C] partials (pseld_reduce_slabs_batched)
    // partial layout [nb][2][C]: reduce the two halves separately
    if (dbeta == dgamma + C) {   // adjacent in the parameter arena: one reduction for both
        pseld_reduce_slabs(workspace, dgamma, (long)2 * C, nb, (long)2 * C, accumulate, s);
    } else {
        pseld_reduce_slabs(workspace, dgamma, (long)C, nb, (long)2 * C, accumulate, s);
        pseld_reduce_slabs(workspace + C, dbeta, (long)C, nb, (long)2 * C, accumulate, s);
    }
    PSELD_LAUNCH_CHECK("layernorm_bwd reduce");
    return PSELD_OK;
}

// ---- scalar BatchNorm --------------------------------------------------------------------------------------
static const int BN_ROWS = 512;

extern "C" long pseld_bn_scalar_workspace(int B, int Cin, int T) {
    return (long)pseld_cdiv((long)B * T, BN_ROWS) * Cin * 64 * 2 * (long)sizeof(float);
}

// Step 1 (training): per-rank sums[Cin*64][2] = (sum x, sum (x - local_mean)^2 ... see below).
// To keep torch's two-pass numerics AND allow sync-BN, the statistics are produced as (sum x, sum x^2) when
// `centered` = 0 (cheap, what a cross-rank all-reduce needs), or as (sum x, sum (x-mean)^2) with the mean taken
// from `sums` itself when `centered` = 1 (single-rank, bit-closer to torch). Both feed pseld_bn_scalar_finalize.
extern "C" int pseld_bn_scalar_stats(const float* feat, float* sums, int B, int Cin, int T, int F, int centered,
                                     float* workspace, long workspace_bytes, void* stream) {
    PSELD_CHECK_ARG(feat && sums && workspace, "bn_scalar_stats: null pointer");
    PSELD_CHECK_ARG(F == 64, "bn_scalar_stats: mel_bins must be 64 (got %d)", F);
    PSELD_CHECK_ARG(workspace_bytes >= pseld_bn_scalar_workspace(B, Cin, T), "bn_scalar_stats: workspace too small");
    PSELD_CHECK_ARG((long)B * T < (1L << 24), "bn_scalar_stats: B * T = %ld rows: the row -> (sample, frame) map is exact below 2^24", (long)B * T);
    hipStream_t s = (hipStream_t)stream;
    const int nb = pseld_cdiv((long)B * T, BN_ROWS);
    const int n = Cin * F;
    hipLaunchKernelGGL(bn_rows_kernel<0>, dim3(nb, Cin), dim3(256), 0, s, feat, (const float*)nullptr, 0.f, workspace, B, Cin, T, BN_ROWS);
    pseld_reduce_slabs(workspace, sums, (long)2 * n, nb, (long)2 * n, 0, s);
    if (centered) {
        const float count = (float)((long)B * T);
        hipLaunchKernelGGL(bn_rows_kernel<1>, dim3(nb, Cin), dim3(256), 0, s, feat, (const float*)sums, count, workspace, B, Cin, T, BN_ROWS);
        // overwrite sums[i][1] with count*var_centered + count*mean^2 so finalize's E[x^2]-mean^2 recovers it
        // exactly: done in finalize via the `centered` flag instead (keeps this buffer all-reducible).
        pseld_reduce_slabs(workspace, sums + 2 * n, (long)n, nb, (long)n, 0, s);
    }
    PSELD_LAUNCH_CHECK("bn_scalar_stats");
    return PSELD_OK;
}

__global__ void bn_fix_centered_kernel(float* sums, int n, float count) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float mean = sums[2 * i] / count;
    sums[2 * i + 1] = sums[2 * n + i] + count * mean * mean;   // so that E[x^2] - mean^2 == centred variance
}

// Step 2: sums (length 2*n, or 3*n when centered) + element count -> scale_shift[n][2], mean_rstd[n][2],
// running statistics updated in place (training) or read (eval).
extern "C" int pseld_bn_scalar_finalize(float* sums, float count, int centered, const float* weight, const float* bias,
                                        float* running_mean, float* running_var, long long* num_batches,
                                        float* mean_rstd, float* scale_shift, int Cin, int F, float momentum, float eps,
                                        int training, void* stream) {
    PSELD_CHECK_ARG(weight && bias && running_mean && running_var && mean_rstd && scale_shift, "bn_scalar_finalize: null pointer");
    PSELD_CHECK_ARG(!training || (sums && num_batches && count > 1.f), "bn_scalar_finalize: training needs sums/count");
    hipStream_t s = (hipStream_t)stream;
    const int n = Cin * F;
    if (training && centered)
        hipLaunchKernelGGL(bn_fix_centered_kernel, dim3(pseld_cdiv(n, 256)), dim3(256), 0, s, sums, n, count);
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(pseld_cdiv(n, 256)), dim3(256), 0, s, sums, count, weight, bias, running_mean,
                       running_var, num_batches, mean_rstd, scale_shift, n, F, momentum, eps, training);
    PSELD_LAUNCH_CHECK("bn_scalar_finalize");
    return PSELD_OK;
}

// Step 3: A[B*4096, Cuse*16] (dtype) = patches of fold(pad(bn(feat[:, c_first : c_first+Cuse])))
extern "C" int pseld_bn_fold_patchify(int dtype, const float* feat, const float* scale_shift, void* A, int B, int Cin,
                                      int c_first, int Cuse, int T, void* stream) {
    PSELD_CHECK_ARG(feat && scale_shift && A, "bn_fold_patchify: null pointer");
    PSELD_CHECK_ARG(T <= 1024 && c_first >= 0 && c_first + Cuse <= Cin, "bn_fold_patchify: bad geometry");
    const long waves = (long)B * 4 * 64 * Cuse;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == PSELD_BF16)
        hipLaunchKernelGGL(patchify_kernel<bf16_t>, dim3(pseld_cdiv(waves, 4)), dim3(256), 0, s, feat, scale_shift, (bf16_t*)A, B, Cin, c_first, Cuse, T, waves);
    else if (dtype == PSELD_F32)
        hipLaunchKernelGGL(patchify_kernel<float>, dim3(pseld_cdiv(waves, 4)), dim3(256), 0, s, feat, scale_shift, (float*)A, B, Cin, c_first, Cuse, T, waves);
    else { pseld_set_error("bn_fold_patchify: unknown dtype"); return PSELD_ERR_BAD_ARG; }
    PSELD_LAUNCH_CHECK("bn_fold_patchify");
    return PSELD_OK;
}

static const int BN_BWD_COLS = 64;
extern "C" long pseld_bn_scalar_bwd_workspace(int B, int Cuse) {
    return ((long)pseld_cdiv((long)B * 256, BN_BWD_COLS) + 1) * Cuse * 64 * 2 * (long)sizeof(float);   // slabs + their sum
}
// Step 4 (backward): dweight/dbias [Cin*64] rows c_first.. (+)= from dA [B*4096, Cuse*16]
extern "C" int pseld_bn_scalar_bwd(int dtype, const float* feat, const float* mean_rstd, const void* dA, float* dweight,
                                   float* dbias, int B, int Cin, int c_first, int Cuse, int T, int accumulate,
                                   float* workspace, long workspace_bytes, void* stream) {
    PSELD_CHECK_ARG(feat && mean_rstd && dA && dweight && dbias && workspace, "bn_scalar_bwd: null pointer");
    PSELD_CHECK_ARG(workspace_bytes >= pseld_bn_scalar_bwd_workspace(B, Cuse), "bn_scalar_bwd: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    const int nb = pseld_cdiv((long)B * 256, BN_BWD_COLS);
    if (dtype == PSELD_BF16)
        hipLaunchKernelGGL(bn_bwd_kernel<bf16_t>, dim3(nb), dim3(256), 0, s, feat, mean_rstd, (const bf16_t*)dA, workspace, B, Cin, c_first, Cuse, T, BN_BWD_COLS);
    else if (dtype == PSELD_F32)
        hipLaunchKernelGGL(bn_bwd_kernel<float>, dim3(nb), dim3(256), 0, s, feat, mean_rstd, (const float*)dA, workspace, B, Cin, c_first, Cuse, T, BN_BWD_COLS);
    else { pseld_set_error("bn_scalar_bwd: unknown dtype"); return PSELD_ERR_BAD_ARG; }
    const int n = Cuse * 64;
    float* total = workspace + (long)nb * n * 2;
    pseld_reduce_slabs(workspace, total, (long)n * 2, nb, (long)n * 2, 0, s);
    hipLaunchKernelGGL(bn_bwd_finish_kernel, dim3(pseld_cdiv(n, 256)), dim3(256), 0, s, total, 1, n, dweight + c_first * 64, dbias + c_first * 64, accumulate);
    PSELD_LAUNCH_CHECK("bn_scalar_bwd");
    return PSELD_OK;
}

// y = x * scale[row_block]: the DropPath backward factor (model_utilities.py:216-232) applied per sample.
extern "C" int pseld_rowscale(int dtype, const void* x, const float* scale, void* y, long n, long elems_per_scale, void* stream) {
    PSELD_CHECK_ARG(x && scale && y && n % 8 == 0 && elems_per_scale % 8 == 0, "rowscale: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    const long n8 = n / 8;
    if (dtype == PSELD_BF16) hipLaunchKernelGGL(rowscale_kernel<bf16_t>, dim3(pseld_cdiv(n8, 256)), dim3(256), 0, s, (const bf16_t*)x, scale, (bf16_t*)y, n8, elems_per_scale);
    else hipLaunchKernelGGL(rowscale_kernel<float>, dim3(pseld_cdiv(n8, 256)), dim3(256), 0, s, (const float*)x, scale, (float*)y, n8, elems_per_scale);
    PSELD_LAUNCH_CHECK("rowscale");
    return PSELD_OK;
}

// ---- CrossStitch ---------------------------------------------------------------------------------------------------
extern "C" int pseld_cross_stitch_fwd(int dtype, const void* x, const void* y, const float* w, void* x_out, void* y_out,
                                      long M, int C, void* stream) {
    PSELD_CHECK_ARG(x && y && w && x_out && y_out && M > 0 && C % 8 == 0, "cross_stitch_fwd: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    const long n8 = M * C / 8;
    if (dtype == PSELD_BF16) hipLaunchKernelGGL(cross_stitch_fwd_kernel<bf16_t>, dim3(pseld_cdiv(n8, 256)), dim3(256), 0, s, (const bf16_t*)x, (const bf16_t*)y, w, (bf16_t*)x_out, (bf16_t*)y_out, n8, C);
    else if (dtype == PSELD_F32) hipLaunchKernelGGL(cross_stitch_fwd_kernel<float>, dim3(pseld_cdiv(n8, 256)), dim3(256), 0, s, (const float*)x, (const float*)y, w, (float*)x_out, (float*)y_out, n8, C);
    else { pseld_set_error("cross_stitch_fwd: unknown dtype"); return PSELD_ERR_BAD_ARG; }
    PSELD_LAUNCH_CHECK("cross_stitch_fwd");
    return PSELD_OK;
}
static inline int cs_rows(long M) { long r = (M + 511) / 512; if (r < 64) r = 64; return (int)r; }
extern "C" long pseld_cross_stitch_bwd_workspace(long M, int C) { return (long)pseld_cdiv(M, cs_rows(M)) * C * 4 * (long)sizeof(float); }
// dx, dy from (dx', dy'); dw f32[C,2,2] overwritten or accumulated.
extern "C" int pseld_cross_stitch_bwd(int dtype, const void* x, const void* y, const float* w, const void* dx_out,
                                      const void* dy_out, void* dx, void* dy, float* dw, long M, int C, int accumulate,
                                      float* workspace, long workspace_bytes, void* stream) {
    PSELD_CHECK_ARG(x && y && w && dx_out && dy_out && dx && dy && dw && workspace, "cross_stitch_bwd: null pointer");
    PSELD_CHECK_ARG(M > 0 && C % 8 == 0 && C <= 2048, "cross_stitch_bwd: bad M/C");
    PSELD_CHECK_ARG(workspace_bytes >= pseld_cross_stitch_bwd_workspace(M, C), "cross_stitch_bwd: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    const int rows = cs_rows(M), nb = pseld_cdiv(M, rows);
    const size_t lds = (size_t)(256 / (C / 8)) * C * 4 * sizeof(float);
    if (dtype == PSELD_BF16) hipLaunchKernelGGL(cross_stitch_bwd_kernel<bf16_t>, dim3(nb), dim3(256), lds, s, (const bf16_t*)x, (const bf16_t*)y, w, (const bf16_t*)dx_out, (const bf16_t*)dy_out, (bf16_t*)dx, (bf16_t*)dy, workspace, M, C, rows);
    else if (dtype == PSELD_F32) hipLaunchKernelGGL(cross_stitch_bwd_kernel<float>, dim3(nb), dim3(256), lds, s, (const float*)x, (const float*)y, w, (const float*)dx_out, (const float*)dy_out, (float*)dx, (float*)dy, workspace, M, C, rows);
    else { pseld_set_error("cross_stitch_bwd: unknown dtype"); return PSELD_ERR_BAD_ARG; }
    PSELD_LAUNCH_CHECK("cross_stitch_bwd");
    pseld_reduce_slabs(workspace, dw, (long)C * 4, nb, (long)C * 4, accumulate, s);
    return PSELD_OK;
}

// y = a + b (gradient fan-in of the shared final tokens in HTSAT_SEDDOA, einv2.py:420-421)
extern "C" int pseld_add(int dtype, const void* a, const void* b, void* y, long n, void* stream) {
    PSELD_CHECK_ARG(a && b && y && n % 8 == 0, "add: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    const long n8 = n / 8;
    if (dtype == PSELD_BF16) hipLaunchKernelGGL(add_kernel<bf16_t>, dim3(pseld_cdiv(n8, 256)), dim3(256), 0, s, (const bf16_t*)a, (const bf16_t*)b, (bf16_t*)y, n8);
    else hipLaunchKernelGGL(add_kernel<float>, dim3(pseld_cdiv(n8, 256)), dim3(256), 0, s, (const float*)a, (const float*)b, (float*)y, n8);
    PSELD_LAUNCH_CHECK("add");
    return PSELD_OK;
}
