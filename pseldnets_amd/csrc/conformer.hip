// Conformer decoder of the CRNN networks (gfx950): the pieces around the GEMMs and LayerNorms.
//
// Replaces (reference, /root/reference/src/models/components/conformer): modules.py:23-35 (ResidualConnectionModule:
// module(x) * factor + x), activation.py (Swish, GLU), feed_forward.py (Swish between the two Linears, Dropout),
// convolution.py:94-151 (GLU, depthwise Conv1d k31 'same', Swish), attention.py:28-147 (RelativeMultiHeadAttention:
// content score (q + u) k^T, positional score (q + v) p^T with the Transformer-XL relative shift, / sqrt(d_model),
// softmax, dropout, @ v) — and the autograd of each. The sequence is 125 frames, so everything here is small next to
// the conv stack: simple LDS-resident kernels, fp32 arithmetic inside.
#include "common.h"

void pseld_reduce_slabs(const float* slabs, float* out, long n, int splits, long slab_stride, int accumulate,
                        hipStream_t stream);

namespace {

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + __expf(-x)); }

template <typename T>
__global__ void axpby_kernel(const T* __restrict__ x, const T* __restrict__ y, T* __restrict__ out, float a, float b, long n8) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n8) return;
    float u[8], v[8];
    load8<T>(x + i * 8, u);
    load8<T>(y + i * 8, v);
#pragma unroll
    for (int k = 0; k < 8; ++k) u[k] = a * u[k] + b * v[k];
    store8<T>(out + i * 8, u);
}
template <typename T>
__global__ void mul_kernel(const T* __restrict__ x, const T* __restrict__ m, T* __restrict__ y, float scale, long n8) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n8) return;
    float u[8], v[8];
    load8<T>(x + i * 8, u);
    load8<T>(m + i * 8, v);
#pragma unroll
    for (int k = 0; k < 8; ++k) u[k] *= v[k] * scale;
    store8<T>(y + i * 8, u);
}
// BWD = false: y = u * sigmoid(u); BWD = true: y = dy * d/du (u * sigmoid(u)) with dy in `g`
template <typename T, bool BWD>
__global__ void swish_kernel(const T* __restrict__ u, const T* __restrict__ g, T* __restrict__ y, long n8) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n8) return;
    float a[8], d[8];
    load8<T>(u + i * 8, a);
    if (BWD) load8<T>(g + i * 8, d);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const float s = sigmoidf_(a[k]);
        a[k] = BWD ? d[k] * (s + a[k] * s * (1.f - s)) : a[k] * s;
    }
    store8<T>(y + i * 8, a);
}
// x [M, 2D] = (a | g): y = a * sigmoid(g); backward: dx = (dy * sigmoid(g) | dy * a * sigmoid(g) * (1 - sigmoid(g)))
template <typename T, bool BWD>
__global__ void glu_kernel(const T* __restrict__ x, const T* __restrict__ dy, T* __restrict__ out, int D, long total8) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total8) return;
    const int d8 = D >> 3;
    const long row = i / d8;
    const int c = (int)(i - row * d8) * 8;
    float a[8], g[8], o[8];
    load8<T>(x + row * 2 * D + c, a);
    load8<T>(x + row * 2 * D + D + c, g);
    if (!BWD) {
#pragma unroll
        for (int k = 0; k < 8; ++k) o[k] = a[k] * sigmoidf_(g[k]);
        store8<T>(out + row * D + c, o);
    } else {
        float d[8], o2[8];
        load8<T>(dy + row * D + c, d);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float s = sigmoidf_(g[k]);
            o[k] = d[k] * s;
            o2[k] = d[k] * a[k] * s * (1.f - s);
        }
        store8<T>(out + row * 2 * D + c, o);
        store8<T>(out + row * 2 * D + D + c, o2);
    }
}

// depthwise Conv1d over time, 'same' padding: y[b,t,d] = sum_k w[d][k] * x[b, t + k - K/2, d]  (FLIP: t - k + K/2 = the
// input gradient). One thread = (b, t, 8 channels).
template <typename T, bool FLIP>
__global__ void dwconv_kernel(const T* __restrict__ x, const float* __restrict__ w, T* __restrict__ y, int Tn, int D, int K,
                              long total8) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total8) return;
    const int d8 = D >> 3;
    const long row = i / d8;
    const int c = (int)(i - row * d8) * 8;
    const int t = (int)(row % Tn);
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < K; ++k) {
        const int ts = FLIP ? t - k + K / 2 : t + k - K / 2;
        if (ts < 0 || ts >= Tn) continue;
        float v[8];
        load8<T>(x + (row + ts - t) * D + c, v);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = fmaf(w[(c + j) * K + k], v[j], acc[j]);
    }
    store8<T>(y + row * D + c, acc);
}
// dw[d][k] = sum_{b,t} dy[b,t,d] * x[b, t + k - K/2, d]: grid (D / 64, K, row blocks), lane = channel; fp32 partial slabs
template <typename T>
__global__ __launch_bounds__(256) void dwconv_wgrad_kernel(const T* __restrict__ x, const T* __restrict__ dy, float* __restrict__ part,
                                                           int B, int Tn, int D, int K, int rows_per_block) {
    __shared__ float red[4][64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int d = blockIdx.x * 64 + lane, k = blockIdx.y;
    const long rows = (long)B * Tn;
    const long beg = (long)blockIdx.z * rows_per_block, end = min(rows, beg + rows_per_block);
    float s = 0.f;
    if (d < D)
        for (long r = beg + wv; r < end; r += 4) {
            const int t = (int)(r % Tn), ts = t + k - K / 2;
            if (ts >= 0 && ts < Tn) s += to_f32<T>(dy[r * D + d]) * to_f32<T>(x[(r + ts - t) * D + d]);
        }
    red[wv][lane] = s;
    __syncthreads();
    if (wv == 0 && d < D) part[((long)blockIdx.z * D + d) * K + k] = red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane];
}

}  // namespace

#define CF_DISPATCH(name, CALL)                                                   \
    if (dtype == PSELD_BF16) { using T = bf16_t; CALL; }                          \
    else if (dtype == PSELD_F32) { using T = float; CALL; }                       \
    else { pseld_set_error(name ": unknown dtype"); return PSELD_ERR_BAD_ARG; }   \
    PSELD_LAUNCH_CHECK(name);                                                     \
    return PSELD_OK

extern "C" int pseld_axpby(int dtype, const void* x, const void* y, void* out, float a, float b, long n, void* stream) {
    PSELD_CHECK_ARG(x && y && out && n > 0 && n % 8 == 0, "axpby: bad argument");
    hipStream_t s = (hipStream_t)stream;
    CF_DISPATCH("axpby", hipLaunchKernelGGL(axpby_kernel<T>, dim3(pseld_cdiv(n / 8, 256)), dim3(256), 0, s, (const T*)x, (const T*)y, (T*)out,
                                            a, b, n / 8));
}
extern "C" int pseld_mul(int dtype, const void* x, const void* m, void* y, float scale, long n, void* stream) {
    PSELD_CHECK_ARG(x && m && y && n > 0 && n % 8 == 0, "mul: bad argument");
    hipStream_t s = (hipStream_t)stream;
    CF_DISPATCH("mul", hipLaunchKernelGGL(mul_kernel<T>, dim3(pseld_cdiv(n / 8, 256)), dim3(256), 0, s, (const T*)x, (const T*)m, (T*)y, scale, n / 8));
}
extern "C" int pseld_swish_fwd(int dtype, const void* u, void* y, long n, void* stream) {
    PSELD_CHECK_ARG(u && y && n > 0 && n % 8 == 0, "swish_fwd: bad argument");
    hipStream_t s = (hipStream_t)stream;
    CF_DISPATCH("swish_fwd", hipLaunchKernelGGL((swish_kernel<T, false>), dim3(pseld_cdiv(n / 8, 256)), dim3(256), 0, s, (const T*)u,
                                                (const T*)nullptr, (T*)y, n / 8));
}
extern "C" int pseld_swish_bwd(int dtype, const void* u, const void* dy, void* du, long n, void* stream) {
    PSELD_CHECK_ARG(u && dy && du && n > 0 && n % 8 == 0, "swish_bwd: bad argument");
    hipStream_t s = (hipStream_t)stream;
    CF_DISPATCH("swish_bwd", hipLaunchKernelGGL((swish_kernel<T, true>), dim3(pseld_cdiv(n / 8, 256)), dim3(256), 0, s, (const T*)u,
                                                (const T*)dy, (T*)du, n / 8));
}
extern "C" int pseld_glu_fwd(int dtype, const void* x, void* y, long M, int D, void* stream) {
    PSELD_CHECK_ARG(x && y && M > 0 && D > 0 && D % 8 == 0, "glu_fwd: bad argument");
    hipStream_t s = (hipStream_t)stream;
    const long total = M * (D / 8);
    CF_DISPATCH("glu_fwd", hipLaunchKernelGGL((glu_kernel<T, false>), dim3(pseld_cdiv(total, 256)), dim3(256), 0, s, (const T*)x,
                                              (const T*)nullptr, (T*)y, D, total));
}
extern "C" int pseld_glu_bwd(int dtype, const void* x, const void* dy, void* dx, long M, int D, void* stream) {
    PSELD_CHECK_ARG(x && dy && dx && M > 0 && D > 0 && D % 8 == 0, "glu_bwd: bad argument");
    hipStream_t s = (hipStream_t)stream;
    const long total = M * (D / 8);
    CF_DISPATCH("glu_bwd", hipLaunchKernelGGL((glu_kernel<T, true>), dim3(pseld_cdiv(total, 256)), dim3(256), 0, s, (const T*)x, (const T*)dy,
                                              (T*)dx, D, total));
}
extern "C" int pseld_dwconv_fwd(int dtype, const void* x, const float* w, void* y, int B, int Tn, int D, int K, int flip,
                                void* stream) {
    PSELD_CHECK_ARG(x && w && y && B > 0 && Tn > 0 && D % 8 == 0 && K % 2 == 1, "dwconv_fwd: bad argument");
    hipStream_t s = (hipStream_t)stream;
    const long total = (long)B * Tn * (D / 8);
    if (flip) { CF_DISPATCH("dwconv", hipLaunchKernelGGL((dwconv_kernel<T, true>), dim3(pseld_cdiv(total, 256)), dim3(256), 0, s, (const T*)x, w, (T*)y, Tn, D, K, total)); }
    CF_DISPATCH("dwconv", hipLaunchKernelGGL((dwconv_kernel<T, false>), dim3(pseld_cdiv(total, 256)), dim3(256), 0, s, (const T*)x, w, (T*)y, Tn, D, K, total));
}
static const int DW_ROWS = 2048;
extern "C" long pseld_dwconv_wgrad_workspace(int B, int Tn, int D, int K) {
    return (long)pseld_cdiv((long)B * Tn, DW_ROWS) * D * K * (long)sizeof(float);
}
extern "C" int pseld_dwconv_wgrad(int dtype, const void* x, const void* dy, float* dw, int B, int Tn, int D, int K,
                                  float* workspace, long workspace_bytes, void* stream) {
    PSELD_CHECK_ARG(x && dy && dw && workspace && B > 0 && Tn > 0 && D > 0 && K % 2 == 1, "dwconv_wgrad: bad argument");
    PSELD_CHECK_ARG(workspace_bytes >= pseld_dwconv_wgrad_workspace(B, Tn, D, K), "dwconv_wgrad: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    const int nb = pseld_cdiv((long)B * Tn, DW_ROWS);
    const dim3 grid(pseld_cdiv(D, 64), K, nb);
    if (dtype == PSELD_BF16) hipLaunchKernelGGL(dwconv_wgrad_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)x, (const bf16_t*)dy, workspace, B, Tn, D, K, DW_ROWS);
    else if (dtype == PSELD_F32) hipLaunchKernelGGL(dwconv_wgrad_kernel<float>, grid, dim3(256), 0, s, (const float*)x, (const float*)dy, workspace, B, Tn, D, K, DW_ROWS);
    else { pseld_set_error("dwconv_wgrad: unknown dtype"); return PSELD_ERR_BAD_ARG; }
    pseld_reduce_slabs(workspace, dw, (long)D * K, nb, (long)D * K, 0, s);
    PSELD_LAUNCH_CHECK("dwconv_wgrad");
    return PSELD_OK;
}

// ---- relative-positional multi-head attention (attention.py:28-112) ------------------------------------------------------------
namespace {

constexpr int RA_TMAX = 128;       // sequence length limit (125 frames here)
constexpr int RA_DC = 16;          // head-dim chunk staged in LDS
constexpr int RA_NP = RA_TMAX * RA_TMAX / 256;

struct RelAttnArgs {
    const void *q, *k, *v, *mask, *dout;   // [B*T, D] (mask: [B, heads, T, T] 0/1 keep-mask, or null)
    const float* pos;                      // [T, D] projected positional encodings (batch independent)
    const float *u_bias, *v_bias;          // [heads * hd]
    void *out, *dq, *dk, *dv;
    float* attn;                           // [B, heads, T, T] softmax output (before dropout)
    float *dpos_part, *dbias_part;         // bwd: [B, T, D] and [B, 2, D] partials (summed over the batch afterwards)
    int B, T, D, heads;
    float scale;                           // 1 / sqrt(d_model)
    float mask_scale;                      // 1 / (1 - p) of the attention dropout
};

// the Transformer-XL "relative shift" of attention.py:104-112: shifted[i][j] = padded_flat[(i + 1) * T + j], padded being the raw
// [T, T] score matrix with a zero column in front. Returns false where the shifted entry is that zero.
__device__ __forceinline__ bool rel_shift_src(int i, int j, int T, int& r, int& c) {
    const int flat = (i + 1) * T + j;
    r = flat / (T + 1);
    c = flat - r * (T + 1) - 1;
    return c >= 0;
}

template <typename T_>
__global__ __launch_bounds__(256) void relattn_fwd_kernel(RelAttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int T = a.T, hd = a.D / a.heads, LD = T + 1;
    float* S = (float*)smem;                 // [T][LD] content, then probabilities
    float* PR = S + T * LD;                  // [T][LD] raw positional scores
    float* Qs = PR + T * LD;                 // [T][RA_DC + 1] x 3
    float* Ks = Qs + T * (RA_DC + 1);
    float* Ps = Ks + T * (RA_DC + 1);
    const int tid = threadIdx.x, b = blockIdx.x / a.heads, h = blockIdx.x % a.heads;
    const T_* q = (const T_*)a.q + (long)b * T * a.D + h * hd;
    const T_* k = (const T_*)a.k + (long)b * T * a.D + h * hd;
    const T_* v = (const T_*)a.v + (long)b * T * a.D + h * hd;
    const float* p = a.pos + h * hd;
    float accc[RA_NP], accp[RA_NP];
#pragma unroll
    for (int n = 0; n < RA_NP; ++n) { accc[n] = 0.f; accp[n] = 0.f; }
    for (int d0 = 0; d0 < hd; d0 += RA_DC) {
        __syncthreads();
        for (int e = tid; e < T * RA_DC; e += 256) {
            const int row = e / RA_DC, dd = e - row * RA_DC;
            const bool in = d0 + dd < hd;
            Qs[row * (RA_DC + 1) + dd] = in ? to_f32<T_>(q[(long)row * a.D + d0 + dd]) : 0.f;
            Ks[row * (RA_DC + 1) + dd] = in ? to_f32<T_>(k[(long)row * a.D + d0 + dd]) : 0.f;
            Ps[row * (RA_DC + 1) + dd] = in ? p[(long)row * a.D + d0 + dd] : 0.f;
        }
        __syncthreads();
        float ub[RA_DC], vb[RA_DC];
#pragma unroll
        for (int dd = 0; dd < RA_DC; ++dd) {
            const bool in = d0 + dd < hd;
            ub[dd] = in ? a.u_bias[h * hd + d0 + dd] : 0.f;
            vb[dd] = in ? a.v_bias[h * hd + d0 + dd] : 0.f;
        }
#pragma unroll
        for (int n = 0; n < RA_NP; ++n) {
            const int idx = tid + 256 * n;
            if (idx < T * T) {
                const int i = idx / T, j = idx - i * T;
                float c = 0.f, pp = 0.f;
#pragma unroll
                for (int dd = 0; dd < RA_DC; ++dd) {
                    const float qv = Qs[i * (RA_DC + 1) + dd];
                    c = fmaf(qv + ub[dd], Ks[j * (RA_DC + 1) + dd], c);
                    pp = fmaf(qv + vb[dd], Ps[j * (RA_DC + 1) + dd], pp);
                }
                accc[n] += c; accp[n] += pp;
            }
        }
    }
#pragma unroll
    for (int n = 0; n < RA_NP; ++n) {
        const int idx = tid + 256 * n;
        if (idx < T * T) { const int i = idx / T, j = idx - i * T; S[i * LD + j] = accc[n]; PR[i * LD + j] = accp[n]; }
    }
    __syncthreads();
    // row-wise: add the shifted positional score, scale, softmax, keep the probabilities, apply the dropout mask
    if (tid < T) {
        const int i = tid;
        float m = -1e30f;
        for (int j = 0; j < T; ++j) {
            int r, c;
            const float ps = rel_shift_src(i, j, T, r, c) ? PR[r * LD + c] : 0.f;
            const float s = (S[i * LD + j] + ps) * a.scale;
            S[i * LD + j] = s;
            m = fmaxf(m, s);
        }
        float l = 0.f;
        for (int j = 0; j < T; ++j) { const float e = __expf(S[i * LD + j] - m); S[i * LD + j] = e; l += e; }
        const float il = 1.f / l;
        float* arow = a.attn + (((long)b * a.heads + h) * T + i) * T;
        const T_* mrow = a.mask ? (const T_*)a.mask + (((long)b * a.heads + h) * T + i) * T : nullptr;
        for (int j = 0; j < T; ++j) {
            const float pr = S[i * LD + j] * il;
            arow[j] = pr;
            S[i * LD + j] = mrow ? pr * to_f32<T_>(mrow[j]) * a.mask_scale : pr;
        }
    }
    __syncthreads();
    T_* o = (T_*)a.out + (long)b * T * a.D + h * hd;
    for (int idx = tid; idx < T * hd; idx += 256) {
        const int i = idx / hd, c = idx - i * hd;
        float acc = 0.f;
        for (int j = 0; j < T; ++j) acc = fmaf(S[i * LD + j], to_f32<T_>(v[(long)j * a.D + c]), acc);
        o[(long)i * a.D + c] = from_f32<T_>(acc);
    }
}

template <typename T_>
__global__ __launch_bounds__(256) void relattn_bwd_kernel(RelAttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int T = a.T, hd = a.D / a.heads, LD = T + 1;
    float* M1 = (float*)smem;                // [T][LD]: A*mask, later the un-shifted positional-score gradient
    float* M2 = M1 + T * LD;                 // [T][LD]: G = (dO V^T)*mask, later dS*scale
    float* As = M2 + T * LD;                 // [T][RA_DC + 1] x 2 tiles
    float* Bs = As + T * (RA_DC + 1);
    const int tid = threadIdx.x, b = blockIdx.x / a.heads, h = blockIdx.x % a.heads;
    const long base = (long)b * T * a.D + h * hd;
    const T_* q = (const T_*)a.q + base;
    const T_* k = (const T_*)a.k + base;
    const T_* v = (const T_*)a.v + base;
    const T_* dO = (const T_*)a.dout + base;
    const float* p = a.pos + h * hd;
    const float* attn = a.attn + ((long)b * a.heads + h) * T * T;
    const T_* mask = a.mask ? (const T_*)a.mask + ((long)b * a.heads + h) * T * T : nullptr;
    // G = (dO V^T) * mask (pairs in registers, head-dim chunks through LDS); M1 = A * mask
    float acc[RA_NP];
#pragma unroll
    for (int n = 0; n < RA_NP; ++n) acc[n] = 0.f;
    for (int d0 = 0; d0 < hd; d0 += RA_DC) {
        __syncthreads();
        for (int e = tid; e < T * RA_DC; e += 256) {
            const int row = e / RA_DC, dd = e - row * RA_DC;
            const bool in = d0 + dd < hd;
            As[row * (RA_DC + 1) + dd] = in ? to_f32<T_>(dO[(long)row * a.D + d0 + dd]) : 0.f;
            Bs[row * (RA_DC + 1) + dd] = in ? to_f32<T_>(v[(long)row * a.D + d0 + dd]) : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int n = 0; n < RA_NP; ++n) {
            const int idx = tid + 256 * n;
            if (idx < T * T) {
                const int i = idx / T, j = idx - i * T;
                float c = 0.f;
#pragma unroll
                for (int dd = 0; dd < RA_DC; ++dd) c = fmaf(As[i * (RA_DC + 1) + dd], Bs[j * (RA_DC + 1) + dd], c);
                acc[n] += c;
            }
        }
    }
#pragma unroll
    for (int n = 0; n < RA_NP; ++n) {
        const int idx = tid + 256 * n;
        if (idx < T * T) {
            const int i = idx / T, j = idx - i * T;
            const float mk = mask ? to_f32<T_>(mask[idx]) * a.mask_scale : 1.f;
            M2[i * LD + j] = acc[n] * mk;
            M1[i * LD + j] = attn[idx] * mk;
        }
    }
    __syncthreads();
    // dV[j][c] = sum_i (A*mask)[i][j] dO[i][c]
    T_* dv = (T_*)a.dv + base;
    for (int idx = tid; idx < T * hd; idx += 256) {
        const int j = idx / hd, c = idx - j * hd;
        float s = 0.f;
        for (int i = 0; i < T; ++i) s = fmaf(M1[i * LD + j], to_f32<T_>(dO[(long)i * a.D + c]), s);
        dv[(long)j * a.D + c] = from_f32<T_>(s);
    }
    __syncthreads();
    for (int e = tid; e < T * LD; e += 256) M1[e] = 0.f;
    __syncthreads();
    // softmax backward per row, scaled; scatter the positional part back through the relative shift
    if (tid < T) {
        const int i = tid;
        float dot = 0.f;
        for (int j = 0; j < T; ++j) dot = fmaf(M2[i * LD + j], attn[i * T + j], dot);
        for (int j = 0; j < T; ++j) {
            const float ds = attn[i * T + j] * (M2[i * LD + j] - dot) * a.scale;
            M2[i * LD + j] = ds;
            int r, c;
            if (rel_shift_src(i, j, T, r, c)) M1[r * LD + c] = ds;     // the shift is a reshape: every source has one reader
        }
    }
    __syncthreads();
    // dq = DS k + DPR p ; du = sum_i DS k ; dvb = sum_i DPR p  (per batch partials) ; dk = DS^T (q + u) ; dp = DPR^T (q + vb)
    T_* dq = (T_*)a.dq + base;
    T_* dk = (T_*)a.dk + base;
    float* dpos = a.dpos_part + (long)b * T * a.D + h * hd;
    float* dub = a.dbias_part + (long)b * 2 * a.D + h * hd;
    for (int idx = tid; idx < T * hd; idx += 256) {
        const int i = idx / hd, c = idx - i * hd;
        float sc = 0.f, sp = 0.f;
        for (int j = 0; j < T; ++j) {
            sc = fmaf(M2[i * LD + j], to_f32<T_>(k[(long)j * a.D + c]), sc);
            sp = fmaf(M1[i * LD + j], p[(long)j * a.D + c], sp);
        }
        dq[(long)i * a.D + c] = from_f32<T_>(sc + sp);
        // reuse the loop index as (j, c) for the transposed products
        const int j2 = i;
        float sk = 0.f, spp = 0.f;
        const float ub = a.u_bias[h * hd + c], vb = a.v_bias[h * hd + c];
        for (int i2 = 0; i2 < T; ++i2) {
            const float qv = to_f32<T_>(q[(long)i2 * a.D + c]);
            sk = fmaf(M2[i2 * LD + j2], qv + ub, sk);
            spp = fmaf(M1[i2 * LD + j2], qv + vb, spp);
        }
        dk[(long)j2 * a.D + c] = from_f32<T_>(sk);
        dpos[(long)j2 * a.D + c] = spp;
    }
    // bias gradients: column sums over i of the two dq parts (recomputed per column by hd threads)
    for (int c = tid; c < hd; c += 256) {
        float su = 0.f, sv = 0.f;
        // sum_i sum_j DS[i][j] k[j][c] = sum_j (sum_i DS[i][j]) k[j][c]
        for (int j = 0; j < T; ++j) {
            float cs = 0.f, cp = 0.f;
            for (int i = 0; i < T; ++i) { cs += M2[i * LD + j]; cp += M1[i * LD + j]; }
            su = fmaf(cs, to_f32<T_>(k[(long)j * a.D + c]), su);
            sv = fmaf(cp, p[(long)j * a.D + c], sv);
        }
        dub[c] = su;
        dub[a.D + c] = sv;
    }
}

size_t relattn_lds(int T, bool bwd) {
    return (size_t)(2 * T * (T + 1) + (bwd ? 2 : 3) * T * (RA_DC + 1)) * sizeof(float);
}

}  // namespace

extern "C" int pseld_relattn_fwd(int dtype, const void* q, const void* k, const void* v, const float* pos, const float* u_bias,
                                 const float* v_bias, const void* mask, float mask_scale, void* out, float* attn, int B, int T, int D,
                                 int heads, void* stream) {
    PSELD_CHECK_ARG(q && k && v && pos && u_bias && v_bias && out && attn, "relattn_fwd: null pointer");
    PSELD_CHECK_ARG(B > 0 && T > 0 && T <= RA_TMAX && heads > 0 && D % heads == 0, "relattn_fwd: bad geometry (T <= 128)");
    RelAttnArgs a; memset(&a, 0, sizeof(a));
    a.q = q; a.k = k; a.v = v; a.pos = pos; a.u_bias = u_bias; a.v_bias = v_bias; a.mask = mask; a.out = out; a.attn = attn;
    a.B = B; a.T = T; a.D = D; a.heads = heads; a.scale = 1.0f / sqrtf((float)D); a.mask_scale = mask_scale;
    hipStream_t s = (hipStream_t)stream;
    const size_t lds = relattn_lds(T, false);
    if (dtype == PSELD_BF16) {
        static bool set = false;
        if (!set) { (void)hipFuncSetAttribute((const void*)relattn_fwd_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); set = true; }
        hipLaunchKernelGGL(relattn_fwd_kernel<bf16_t>, dim3(B * heads), dim3(256), lds, s, a);
    } else if (dtype == PSELD_F32) {
        static bool set = false;
        if (!set) { (void)hipFuncSetAttribute((const void*)relattn_fwd_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); set = true; }
        hipLaunchKernelGGL(relattn_fwd_kernel<float>, dim3(B * heads), dim3(256), lds, s, a);
    } else { pseld_set_error("relattn_fwd: unknown dtype"); return PSELD_ERR_BAD_ARG; }
    PSELD_LAUNCH_CHECK("relattn_fwd");
    return PSELD_OK;
}

extern "C" long pseld_relattn_bwd_workspace(int B, int T, int D) { return ((long)B * T * D + (long)B * 2 * D) * (long)sizeof(float); }

/* dq, dk, dv [B*T, D]; dpos f32[T, D], du_bias / dv_bias f32[heads*hd] (overwritten; summed over the batch) */
extern "C" int pseld_relattn_bwd(int dtype, const void* q, const void* k, const void* v, const float* pos, const float* u_bias,
                                 const float* v_bias, const void* mask, float mask_scale, const float* attn, const void* dout, void* dq,
                                 void* dk, void* dv, float* dpos, float* du_bias, float* dv_bias, int B, int T, int D, int heads,
                                 float* workspace, long workspace_bytes, void* stream) {
    PSELD_CHECK_ARG(q && k && v && pos && u_bias && v_bias && attn && dout && dq && dk && dv && dpos && du_bias && dv_bias && workspace,
                    "relattn_bwd: null pointer");
    PSELD_CHECK_ARG(B > 0 && T > 0 && T <= RA_TMAX && heads > 0 && D % heads == 0, "relattn_bwd: bad geometry (T <= 128)");
    PSELD_CHECK_ARG(workspace_bytes >= pseld_relattn_bwd_workspace(B, T, D), "relattn_bwd: workspace too small");
    RelAttnArgs a; memset(&a, 0, sizeof(a));
    a.q = q; a.k = k; a.v = v; a.pos = pos; a.u_bias = u_bias; a.v_bias = v_bias; a.mask = mask; a.attn = const_cast<float*>(attn);
    a.dout = dout; a.dq = dq; a.dk = dk; a.dv = dv; a.dpos_part = workspace; a.dbias_part = workspace + (long)B * T * D;
    a.B = B; a.T = T; a.D = D; a.heads = heads; a.scale = 1.0f / sqrtf((float)D); a.mask_scale = mask_scale;
    hipStream_t s = (hipStream_t)stream;
    const size_t lds = relattn_lds(T, true);
    if (dtype == PSELD_BF16) {
        static bool set = false;
        if (!set) { (void)hipFuncSetAttribute((const void*)relattn_bwd_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); set = true; }
        hipLaunchKernelGGL(relattn_bwd_kernel<bf16_t>, dim3(B * heads), dim3(256), lds, s, a);
    } else if (dtype == PSELD_F32) {
        static bool set = false;
        if (!set) { (void)hipFuncSetAttribute((const void*)relattn_bwd_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); set = true; }
        hipLaunchKernelGGL(relattn_bwd_kernel<float>, dim3(B * heads), dim3(256), lds, s, a);
    } else { pseld_set_error("relattn_bwd: unknown dtype"); return PSELD_ERR_BAD_ARG; }
    pseld_reduce_slabs(a.dpos_part, dpos, (long)T * D, B, (long)T * D, 0, s);
    float* tmp = a.dbias_part;                                   // [B][2][D]
    pseld_reduce_slabs(tmp, du_bias, (long)D, B, (long)2 * D, 0, s);
    pseld_reduce_slabs(tmp + D, dv_bias, (long)D, B, (long)2 * D, 0, s);
    PSELD_LAUNCH_CHECK("relattn_bwd");
    return PSELD_OK;
}
