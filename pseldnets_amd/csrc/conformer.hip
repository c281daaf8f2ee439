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
__global__ void axpby_kernel(const T* __restrict__ x, const T* __restrict__ y, T* __restrict__ out, float a, float b, long n8, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i == n8) {                                       // scalar tail (n % 8 elements)
        for (long e = n8 * 8; e < n; ++e) out[e] = from_f32<T>(a * to_f32<T>(x[e]) + b * to_f32<T>(y[e]));
        return;
    }
    if (i > n8) return;
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
// BWD = false: y = max(u, 0); BWD = true: y = dy * (u > 0)
template <typename T, bool BWD>
__global__ void relu_kernel(const T* __restrict__ u, const T* __restrict__ g, T* __restrict__ y, long n8) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n8) return;
    float a[8], d[8];
    load8<T>(u + i * 8, a);
    if (BWD) load8<T>(g + i * 8, d);
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] = BWD ? (a[k] > 0.f ? d[k] : 0.f) : fmaxf(a[k], 0.f);
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


// ---- the kernel-31 case (ConformerBlocks' default): one workgroup per (sample, 128 channels) keeps the whole padded
// sequence tile in LDS and the 31 taps of a thread's two channels in registers ----------------------------------------------
constexpr int DW_K = 31, DW_CH = 128;
template <typename T> __device__ __forceinline__ void pair_read(const T* p, float& a, float& b);
template <> __device__ __forceinline__ void pair_read<float>(const float* p, float& a, float& b) { const float2 v = *(const float2*)p; a = v.x; b = v.y; }
template <> __device__ __forceinline__ void pair_read<bf16_t>(const bf16_t* p, float& a, float& b) {
    const unsigned u = *(const unsigned*)p;
    a = __uint_as_float(u << 16); b = __uint_as_float(u & 0xffff0000u);
}
template <typename T> __device__ __forceinline__ void pair_write(T* p, float a, float b);
template <> __device__ __forceinline__ void pair_write<float>(float* p, float a, float b) { *(float2*)p = make_float2(a, b); }
template <> __device__ __forceinline__ void pair_write<bf16_t>(bf16_t* p, float a, float b) {
    typedef bf16_t bf16x2 __attribute__((ext_vector_type(2)));
    bf16x2 v; v[0] = (bf16_t)a; v[1] = (bf16_t)b;
    *(bf16x2*)p = v;
}
// rows r <-> time r - lead of sample b, channels c0..c0+127, zero outside the sequence / channel range
template <typename T>
__device__ __forceinline__ void dw_stage(T* dst, const T* src, int b, int Tn, int D, int c0, int lead, int rows, int tid) {
    for (int e = tid; e < rows * (DW_CH / 8); e += 256) {
        const int r = e >> 4, o = e & 15, t = r - lead;
        Vec8<T> v = {};
        if (t >= 0 && t < Tn && c0 + o * 8 < D) v = *(const Vec8<T>*)(src + ((long)b * Tn + t) * D + c0 + o * 8);
        *(Vec8<T>*)(dst + r * DW_CH + o * 8) = v;
    }
}
template <typename T, bool FLIP>
__global__ __launch_bounds__(256) void dwconv31_kernel(const T* __restrict__ x, const float* __restrict__ w, T* __restrict__ y, int Tn, int D) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* xs = (T*)smem;                                   // [Tn + 30][128]
    const int tid = threadIdx.x, b = blockIdx.y, c0 = blockIdx.x * DW_CH;
    dw_stage<T>(xs, x, b, Tn, D, c0, DW_K / 2, Tn + DW_K - 1, tid);
    const int cp = tid & 63, lane = tid >> 6, ch = c0 + 2 * cp;
    float w0[DW_K], w1[DW_K];
#pragma unroll
    for (int k = 0; k < DW_K; ++k) {
        const int kk = FLIP ? DW_K - 1 - k : k;
        w0[k] = ch < D ? w[ch * DW_K + kk] : 0.f;
        w1[k] = ch + 1 < D ? w[(ch + 1) * DW_K + kk] : 0.f;
    }
    __syncthreads();
    if (ch >= D) return;
    for (int t = lane; t < Tn; t += 4) {
        float a0 = 0.f, a1 = 0.f;
#pragma unroll
        for (int k = 0; k < DW_K; ++k) {
            float x0, x1;
            pair_read<T>(xs + (t + k) * DW_CH + 2 * cp, x0, x1);
            a0 = fmaf(w0[k], x0, a0); a1 = fmaf(w1[k], x1, a1);
        }
        pair_write<T>(y + ((long)b * Tn + t) * D + ch, a0, a1);
    }
}
// dw partial of one sample: part[b][d][k] = sum_t dy[b,t,d] * x[b, t + k - 15, d]
template <typename T>
__global__ __launch_bounds__(256) void dwconv31_wgrad_kernel(const T* __restrict__ x, const T* __restrict__ dy, float* __restrict__ part,
                                                             int Tn, int D) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* xs = (T*)smem;                                   // [Tn + 30][128]
    T* ds = xs + (Tn + DW_K - 1) * DW_CH;               // [Tn][128]
    const int tid = threadIdx.x, b = blockIdx.y, c0 = blockIdx.x * DW_CH;
    dw_stage<T>(xs, x, b, Tn, D, c0, DW_K / 2, Tn + DW_K - 1, tid);
    dw_stage<T>(ds, dy, b, Tn, D, c0, 0, Tn, tid);
    __syncthreads();
    const int cp = tid & 63, lane = tid >> 6;
    float a0[DW_K], a1[DW_K];
#pragma unroll
    for (int k = 0; k < DW_K; ++k) { a0[k] = 0.f; a1[k] = 0.f; }
    for (int t = lane; t < Tn; t += 4) {
        float d0, d1;
        pair_read<T>(ds + t * DW_CH + 2 * cp, d0, d1);
#pragma unroll
        for (int k = 0; k < DW_K; ++k) {
            float x0, x1;
            pair_read<T>(xs + (t + k) * DW_CH + 2 * cp, x0, x1);
            a0[k] = fmaf(d0, x0, a0[k]); a1[k] = fmaf(d1, x1, a1[k]);
        }
    }
    __syncthreads();
    float* red = (float*)smem;                          // [4][128][31]
#pragma unroll
    for (int k = 0; k < DW_K; ++k) {
        red[(lane * DW_CH + 2 * cp) * DW_K + k] = a0[k];
        red[(lane * DW_CH + 2 * cp + 1) * DW_K + k] = a1[k];
    }
    __syncthreads();
    for (int e = tid; e < DW_CH * DW_K; e += 256) {
        const int c = e / DW_K;
        if (c0 + c < D)
            part[((long)b * D + c0) * DW_K + e] = red[e] + red[DW_CH * DW_K + e] + red[2 * DW_CH * DW_K + e] + red[3 * DW_CH * DW_K + e];
    }
}
template <typename T> size_t dw31_lds(int Tn, bool wgrad) {
    const size_t tiles = (size_t)((Tn + DW_K - 1) + (wgrad ? Tn : 0)) * DW_CH * sizeof(T);
    const size_t red = wgrad ? (size_t)4 * DW_CH * DW_K * sizeof(float) : 0;
    return tiles > red ? tiles : red;
}

}  // namespace

#define CF_DISPATCH(name, CALL)                                                   \
    if (dtype == PSELD_BF16) { using T = bf16_t; CALL; }                          \
    else if (dtype == PSELD_F32) { using T = float; CALL; }                       \
    else { pseld_set_error(name ": unknown dtype"); return PSELD_ERR_BAD_ARG; }   \
    PSELD_LAUNCH_CHECK(name);                                                     \
    return PSELD_OK

extern "C" int pseld_axpby(int dtype, const void* x, const void* y, void* out, float a, float b, long n, void* stream) {
    PSELD_CHECK_ARG(x && y && out && n > 0, "axpby: bad argument");
    hipStream_t s = (hipStream_t)stream;
    CF_DISPATCH("axpby", hipLaunchKernelGGL(axpby_kernel<T>, dim3(pseld_cdiv(n / 8 + 1, 256)), dim3(256), 0, s, (const T*)x, (const T*)y, (T*)out,
                                            a, b, n / 8, n));
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
extern "C" int pseld_relu_fwd(int dtype, const void* u, void* y, long n, void* stream) {
    PSELD_CHECK_ARG(u && y && n > 0 && n % 8 == 0, "relu_fwd: bad argument");
    hipStream_t s = (hipStream_t)stream;
    CF_DISPATCH("relu_fwd", hipLaunchKernelGGL((relu_kernel<T, false>), dim3(pseld_cdiv(n / 8, 256)), dim3(256), 0, s, (const T*)u,
                                               (const T*)nullptr, (T*)y, n / 8));
}
extern "C" int pseld_relu_bwd(int dtype, const void* u, const void* dy, void* du, long n, void* stream) {
    PSELD_CHECK_ARG(u && dy && du && n > 0 && n % 8 == 0, "relu_bwd: bad argument");
    hipStream_t s = (hipStream_t)stream;
    CF_DISPATCH("relu_bwd", hipLaunchKernelGGL((relu_kernel<T, true>), dim3(pseld_cdiv(n / 8, 256)), dim3(256), 0, s, (const T*)u, (const T*)dy,
                                               (T*)du, n / 8));
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
    if (K == DW_K && D % 2 == 0) {
        const dim3 grid(pseld_cdiv(D, DW_CH), B);
        if (dtype == PSELD_BF16 && dw31_lds<bf16_t>(Tn, false) <= 160 * 1024) {
            const size_t lds = dw31_lds<bf16_t>(Tn, false);
            static bool set = false;
            if (!set) {
                (void)hipFuncSetAttribute((const void*)dwconv31_kernel<bf16_t, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                (void)hipFuncSetAttribute((const void*)dwconv31_kernel<bf16_t, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                set = true;
            }
            if (flip) hipLaunchKernelGGL((dwconv31_kernel<bf16_t, true>), grid, dim3(256), lds, s, (const bf16_t*)x, w, (bf16_t*)y, Tn, D);
            else hipLaunchKernelGGL((dwconv31_kernel<bf16_t, false>), grid, dim3(256), lds, s, (const bf16_t*)x, w, (bf16_t*)y, Tn, D);
            PSELD_LAUNCH_CHECK("dwconv");
            return PSELD_OK;
        }
        if (dtype == PSELD_F32 && dw31_lds<float>(Tn, false) <= 160 * 1024) {
            const size_t lds = dw31_lds<float>(Tn, false);
            static bool set = false;
            if (!set) {
                (void)hipFuncSetAttribute((const void*)dwconv31_kernel<float, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                (void)hipFuncSetAttribute((const void*)dwconv31_kernel<float, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                set = true;
            }
            if (flip) hipLaunchKernelGGL((dwconv31_kernel<float, true>), grid, dim3(256), lds, s, (const float*)x, w, (float*)y, Tn, D);
            else hipLaunchKernelGGL((dwconv31_kernel<float, false>), grid, dim3(256), lds, s, (const float*)x, w, (float*)y, Tn, D);
            PSELD_LAUNCH_CHECK("dwconv");
            return PSELD_OK;
        }
    }
    if (flip) { CF_DISPATCH("dwconv", hipLaunchKernelGGL((dwconv_kernel<T, true>), dim3(pseld_cdiv(total, 256)), dim3(256), 0, s, (const T*)x, w, (T*)y, Tn, D, K, total)); }
    CF_DISPATCH("dwconv", hipLaunchKernelGGL((dwconv_kernel<T, false>), dim3(pseld_cdiv(total, 256)), dim3(256), 0, s, (const T*)x, w, (T*)y, Tn, D, K, total));
}
static const int DW_ROWS = 2048;
extern "C" long pseld_dwconv_wgrad_workspace(int B, int Tn, int D, int K) {
    const long slabs = pseld_cdiv((long)B * Tn, DW_ROWS) > B ? pseld_cdiv((long)B * Tn, DW_ROWS) : B;      // tiled path: one slab per sample
    return slabs * D * K * (long)sizeof(float);
}
extern "C" int pseld_dwconv_wgrad(int dtype, const void* x, const void* dy, float* dw, int B, int Tn, int D, int K,
                                  float* workspace, long workspace_bytes, void* stream) {
    PSELD_CHECK_ARG(x && dy && dw && workspace && B > 0 && Tn > 0 && D > 0 && K % 2 == 1, "dwconv_wgrad: bad argument");
    PSELD_CHECK_ARG(workspace_bytes >= pseld_dwconv_wgrad_workspace(B, Tn, D, K), "dwconv_wgrad: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    if (K == DW_K && D % 2 == 0 && (dtype == PSELD_BF16 ? dw31_lds<bf16_t>(Tn, true) : dw31_lds<float>(Tn, true)) <= 160 * 1024) {
        const dim3 grid31(pseld_cdiv(D, DW_CH), B);
        if (dtype == PSELD_BF16) {
            static bool set = false;
            if (!set) { (void)hipFuncSetAttribute((const void*)dwconv31_wgrad_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); set = true; }
            hipLaunchKernelGGL(dwconv31_wgrad_kernel<bf16_t>, grid31, dim3(256), dw31_lds<bf16_t>(Tn, true), s, (const bf16_t*)x, (const bf16_t*)dy, workspace, Tn, D);
        } else if (dtype == PSELD_F32) {
            static bool set = false;
            if (!set) { (void)hipFuncSetAttribute((const void*)dwconv31_wgrad_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); set = true; }
            hipLaunchKernelGGL(dwconv31_wgrad_kernel<float>, grid31, dim3(256), dw31_lds<float>(Tn, true), s, (const float*)x, (const float*)dy, workspace, Tn, D);
        } else { pseld_set_error("dwconv_wgrad: unknown dtype"); return PSELD_ERR_BAD_ARG; }
        pseld_reduce_slabs(workspace, dw, (long)D * K, B, (long)D * K, 0, s);
        PSELD_LAUNCH_CHECK("dwconv_wgrad");
        return PSELD_OK;
    }
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
// One workgroup (256 threads) per (sample, head); the sequence (<= 128 frames) is one 128 x 128 score tile. Every product
// is register-tiled: thread (ti, tj) = (tid >> 4, tid & 15) owns the 8 x 8 outputs {ti + 16 r} x {tj + 16 c} (interleaved
// rows / columns keep the LDS reads conflict-free), operand chunks come through LDS, and the two [128][129] fp32
// matrices (scores / probabilities, raw positional scores and their gradients) stay in LDS for the whole kernel.
namespace {

constexpr int RA_TMAX = 128;       // sequence length limit (125 frames here)
constexpr int RA_LD = RA_TMAX + 1; // row stride of the LDS-resident matrices
constexpr int RA_DC = 16;          // reduction chunk staged in LDS
constexpr int RA_SLD = RA_DC + 1;  // row stride of a [128][16] staged chunk
constexpr int RA_MAT = RA_TMAX * RA_LD;
constexpr int RA_STG = RA_TMAX * RA_SLD;    // floats per staged chunk ([128][17]; a [16][128] chunk fits too)

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
// its inverse: raw[r][c] is read by shifted[i][j] (false: by nobody — the first T - 1 entries fall off the front)
__device__ __forceinline__ bool rel_shift_dst(int r, int c, int T, int& i, int& j) {
    const int flat = r * (T + 1) + c + 1;
    i = flat / T - 1;
    j = flat - (i + 1) * T;
    return i >= 0;
}

// rows [0, 128) x 16 columns d0.. of a [T][ld] matrix -> registers (zero outside T x ncols) -> S[128][17]
template <typename T_>
__device__ __forceinline__ void fetch_rows16(float (&r)[8], const T_* g, int ld, int T, int d0, int ncols, int tid) {
#pragma unroll
    for (int n = 0; n < 8; ++n) {
        const int e = tid + 256 * n, row = e >> 4, dd = e & 15;
        r[n] = (row < T && d0 + dd < ncols) ? to_f32<T_>(g[(long)row * ld + d0 + dd]) : 0.f;
    }
}
__device__ __forceinline__ void put_rows16(float* S, const float (&r)[8], int tid) {
#pragma unroll
    for (int n = 0; n < 8; ++n) { const int e = tid + 256 * n; S[(e >> 4) * RA_SLD + (e & 15)] = r[n]; }
}
// rows k0..k0+15 x columns c0..c0+127 of a [T][ld] matrix (+ a per-column bias) -> registers -> S[16][128]
template <typename T_>
__device__ __forceinline__ void fetch_cols128(float (&r)[8], const T_* g, int ld, int T, int k0, int c0, int ncols, const float* colbias,
                                              int tid) {
#pragma unroll
    for (int n = 0; n < 8; ++n) {
        const int e = tid + 256 * n, kk = e >> 7, col = e & 127;
        const bool ok = k0 + kk < T && c0 + col < ncols;
        r[n] = ok ? to_f32<T_>(g[(long)(k0 + kk) * ld + c0 + col]) + (colbias ? colbias[c0 + col] : 0.f) : 0.f;
    }
}
__device__ __forceinline__ void put_cols128(float* S, const float (&r)[8], int tid) {
#pragma unroll
    for (int n = 0; n < 8; ++n) S[tid + 256 * n] = r[n];
}
__device__ __forceinline__ void zero_tile(float (&acc)[8][8]) {
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int c = 0; c < 8; ++c) acc[r][c] = 0.f;
}
// acc[r][c] += sum_kk A(ti + 16 r, kk) * Bst[kk][tj + 16 c]; A(i, kk) = Mat[i][k0 + kk] or (TRANS) Mat[k0 + kk][i]
template <bool TRANS>
__device__ __forceinline__ void mat_times_chunk(float (&acc)[8][8], const float* Mat, const float* Bst, int k0, int ti, int tj) {
#pragma unroll 4
    for (int kk = 0; kk < RA_DC; ++kk) {
        float a[8], b[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) a[r] = TRANS ? Mat[(k0 + kk) * RA_LD + ti + 16 * r] : Mat[(ti + 16 * r) * RA_LD + k0 + kk];
#pragma unroll
        for (int c = 0; c < 8; ++c) b[c] = Bst[kk * 128 + tj + 16 * c];
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int c = 0; c < 8; ++c) acc[r][c] = fmaf(a[r], b[c], acc[r][c]);
    }
}
// out[ti + 16 r][c0 + tj + 16 c] = acc (rows < T, columns < ncols)
template <typename TO>
__device__ __forceinline__ void store_tile(TO* out, int ld, const float (&acc)[8][8], int T, int c0, int ncols, int ti, int tj) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int i = ti + 16 * r;
        if (i >= T) continue;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const int col = c0 + tj + 16 * c;
            if (col < ncols) out[(long)i * ld + col] = from_f32<TO>(acc[r][c]);
        }
    }
}

template <typename T_>
__global__ __launch_bounds__(256) void relattn_fwd_kernel(RelAttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int T = a.T, hd = a.D / a.heads;
    float* S = (float*)smem;                 // [128][129] content scores, then probabilities (x dropout mask)
    float* PR = S + RA_MAT;                  // [128][129] raw positional scores
    float* Qs = PR + RA_MAT;                 // staged chunks
    float* Ks = Qs + RA_STG;
    float* Ps = Ks + RA_STG;
    const int tid = threadIdx.x, ti = tid >> 4, tj = tid & 15;
    const int b = blockIdx.x / a.heads, h = blockIdx.x % a.heads;
    const long base = (long)b * T * a.D + h * hd;
    const T_* q = (const T_*)a.q + base;
    const T_* k = (const T_*)a.k + base;
    const T_* v = (const T_*)a.v + base;
    const float* p = a.pos + h * hd;
    // content = (q + u) k^T and raw positional = (q + v) p^T over head-dim chunks
    {
        float accc[8][8], accp[8][8], rq[8], rk[8], rp[8];
        zero_tile(accc); zero_tile(accp);
        fetch_rows16<T_>(rq, q, a.D, T, 0, hd, tid);
        fetch_rows16<T_>(rk, k, a.D, T, 0, hd, tid);
        fetch_rows16<float>(rp, p, a.D, T, 0, hd, tid);
        for (int d0 = 0; d0 < hd; d0 += RA_DC) {
            __syncthreads();
            put_rows16(Qs, rq, tid); put_rows16(Ks, rk, tid); put_rows16(Ps, rp, tid);
            __syncthreads();
            if (d0 + RA_DC < hd) {
                fetch_rows16<T_>(rq, q, a.D, T, d0 + RA_DC, hd, tid);
                fetch_rows16<T_>(rk, k, a.D, T, d0 + RA_DC, hd, tid);
                fetch_rows16<float>(rp, p, a.D, T, d0 + RA_DC, hd, tid);
            }
#pragma unroll 2
            for (int dd = 0; dd < RA_DC; ++dd) {
                const float ub = d0 + dd < hd ? a.u_bias[h * hd + d0 + dd] : 0.f;
                const float vb = d0 + dd < hd ? a.v_bias[h * hd + d0 + dd] : 0.f;
                float qa[8], kb[8], pb[8];
#pragma unroll
                for (int r = 0; r < 8; ++r) qa[r] = Qs[(ti + 16 * r) * RA_SLD + dd];
#pragma unroll
                for (int c = 0; c < 8; ++c) { kb[c] = Ks[(tj + 16 * c) * RA_SLD + dd]; pb[c] = Ps[(tj + 16 * c) * RA_SLD + dd]; }
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    const float qu = qa[r] + ub, qv = qa[r] + vb;
#pragma unroll
                    for (int c = 0; c < 8; ++c) { accc[r][c] = fmaf(qu, kb[c], accc[r][c]); accp[r][c] = fmaf(qv, pb[c], accp[r][c]); }
                }
            }
        }
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                S[(ti + 16 * r) * RA_LD + tj + 16 * c] = accc[r][c];
                PR[(ti + 16 * r) * RA_LD + tj + 16 * c] = accp[r][c];
            }
    }
    __syncthreads();
    // rows: add the shifted positional score, scale, softmax. Two threads per row (even / odd columns).
    {
        const int i = tid >> 1, half = tid & 1;
        if (i < T) {
            float m = -1e30f;
            for (int j = half; j < T; j += 2) {
                int r, c;
                const float ps = rel_shift_src(i, j, T, r, c) ? PR[r * RA_LD + c] : 0.f;
                const float s = (S[i * RA_LD + j] + ps) * a.scale;
                S[i * RA_LD + j] = s;
                m = fmaxf(m, s);
            }
            m = fmaxf(m, __shfl_xor(m, 1));
            float l = 0.f;
            for (int j = half; j < T; j += 2) { const float e = __expf(S[i * RA_LD + j] - m); S[i * RA_LD + j] = e; l += e; }
            l += __shfl_xor(l, 1);
            const float il = 1.f / l;
            for (int j = half; j < T; j += 2) S[i * RA_LD + j] *= il;
        }
    }
    __syncthreads();
    // keep the probabilities for the backward, apply the dropout mask (coalesced over the T x T block); zero the padding
    {
        float* arow = a.attn + ((long)b * a.heads + h) * T * T;
        const T_* mrow = a.mask ? (const T_*)a.mask + ((long)b * a.heads + h) * T * T : nullptr;
        for (int idx = tid; idx < RA_TMAX * RA_TMAX; idx += 256) {
            const int i = idx >> 7, j = idx & 127;
            if (i < T && j < T) {
                const float pr = S[i * RA_LD + j];
                arow[i * T + j] = pr;
                if (mrow) S[i * RA_LD + j] = pr * to_f32<T_>(mrow[i * T + j]) * a.mask_scale;
            } else {
                S[i * RA_LD + j] = 0.f;
            }
        }
    }
    // context = P v, 128 head-dim columns per pass
    T_* o = (T_*)a.out + base;
    float* Bst = Qs;
    for (int c0 = 0; c0 < hd; c0 += 128) {
        float acc[8][8], rv[8];
        zero_tile(acc);
        fetch_cols128<T_>(rv, v, a.D, T, 0, c0, hd, nullptr, tid);
        for (int k0 = 0; k0 < T; k0 += RA_DC) {
            __syncthreads();
            put_cols128(Bst, rv, tid);
            __syncthreads();
            if (k0 + RA_DC < T) fetch_cols128<T_>(rv, v, a.D, T, k0 + RA_DC, c0, hd, nullptr, tid);
            mat_times_chunk<false>(acc, S, Bst, k0, ti, tj);
        }
        store_tile<T_>(o, a.D, acc, T, c0, hd, ti, tj);
    }
}

template <typename T_>
__global__ __launch_bounds__(256) void relattn_bwd_kernel(RelAttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int T = a.T, hd = a.D / a.heads;
    float* M1 = (float*)smem;                // [128][129]: A*mask -> A -> un-shifted positional-score gradient
    float* M2 = M1 + RA_MAT;                 // [128][129]: G = (dO V^T)*mask -> dS*scale
    float* St0 = M2 + RA_MAT;                // staged chunks
    float* St1 = St0 + RA_STG;
    float* colsum = St1 + RA_STG;            // [2][hd]: du_bias, dv_bias of this (sample, head)
    const int tid = threadIdx.x, ti = tid >> 4, tj = tid & 15;
    const int b = blockIdx.x / a.heads, h = blockIdx.x % a.heads;
    const long base = (long)b * T * a.D + h * hd;
    const T_* q = (const T_*)a.q + base;
    const T_* k = (const T_*)a.k + base;
    const T_* v = (const T_*)a.v + base;
    const T_* dO = (const T_*)a.dout + base;
    const float* p = a.pos + h * hd;
    const float* attn = a.attn + ((long)b * a.heads + h) * T * T;
    const T_* mask = a.mask ? (const T_*)a.mask + ((long)b * a.heads + h) * T * T : nullptr;
    for (int c = tid; c < 2 * hd; c += 256) colsum[c] = 0.f;
    // M1 = A * mask (zero padding)
    for (int idx = tid; idx < RA_TMAX * RA_TMAX; idx += 256) {
        const int i = idx >> 7, j = idx & 127;
        float val = 0.f;
        if (i < T && j < T) val = attn[i * T + j] * (mask ? to_f32<T_>(mask[i * T + j]) * a.mask_scale : 1.f);
        M1[i * RA_LD + j] = val;
    }
    // dV = (A*mask)^T dO
    T_* dv = (T_*)a.dv + base;
    for (int c0 = 0; c0 < hd; c0 += 128) {
        float acc[8][8], rr[8];
        zero_tile(acc);
        fetch_cols128<T_>(rr, dO, a.D, T, 0, c0, hd, nullptr, tid);
        for (int k0 = 0; k0 < T; k0 += RA_DC) {
            __syncthreads();
            put_cols128(St0, rr, tid);
            __syncthreads();
            if (k0 + RA_DC < T) fetch_cols128<T_>(rr, dO, a.D, T, k0 + RA_DC, c0, hd, nullptr, tid);
            mat_times_chunk<true>(acc, M1, St0, k0, ti, tj);
        }
        store_tile<T_>(dv, a.D, acc, T, c0, hd, ti, tj);
    }
    // G = dO V^T over head-dim chunks; then M2 = G * mask, M1 = A
    {
        float acc[8][8], ra[8], rb[8];
        zero_tile(acc);
        fetch_rows16<T_>(ra, dO, a.D, T, 0, hd, tid);
        fetch_rows16<T_>(rb, v, a.D, T, 0, hd, tid);
        for (int d0 = 0; d0 < hd; d0 += RA_DC) {
            __syncthreads();
            put_rows16(St0, ra, tid); put_rows16(St1, rb, tid);
            __syncthreads();
            if (d0 + RA_DC < hd) {
                fetch_rows16<T_>(ra, dO, a.D, T, d0 + RA_DC, hd, tid);
                fetch_rows16<T_>(rb, v, a.D, T, d0 + RA_DC, hd, tid);
            }
#pragma unroll 4
            for (int dd = 0; dd < RA_DC; ++dd) {
                float x[8], y[8];
#pragma unroll
                for (int r = 0; r < 8; ++r) x[r] = St0[(ti + 16 * r) * RA_SLD + dd];
#pragma unroll
                for (int c = 0; c < 8; ++c) y[c] = St1[(tj + 16 * c) * RA_SLD + dd];
#pragma unroll
                for (int r = 0; r < 8; ++r)
#pragma unroll
                    for (int c = 0; c < 8; ++c) acc[r][c] = fmaf(x[r], y[c], acc[r][c]);
            }
        }
        __syncthreads();                       // every reader of M1 = A*mask (dV) is done
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const int i = ti + 16 * r, j = tj + 16 * c;
                float g = 0.f, pa = 0.f;
                if (i < T && j < T) {
                    pa = attn[i * T + j];
                    g = acc[r][c] * (mask ? to_f32<T_>(mask[i * T + j]) * a.mask_scale : 1.f);
                }
                M2[i * RA_LD + j] = g;
                M1[i * RA_LD + j] = pa;
            }
    }
    __syncthreads();
    // softmax backward per row (two threads per row), scaled: M2 = dS
    {
        const int i = tid >> 1, half = tid & 1;
        if (i < T) {
            float dot = 0.f;
            for (int j = half; j < T; j += 2) dot = fmaf(M2[i * RA_LD + j], M1[i * RA_LD + j], dot);
            dot += __shfl_xor(dot, 1);
            for (int j = half; j < T; j += 2) M2[i * RA_LD + j] = M1[i * RA_LD + j] * (M2[i * RA_LD + j] - dot) * a.scale;
        }
    }
    __syncthreads();
    // M1 = gradient of the RAW positional scores: the relative shift is a reshape, so this is a gather from dS
    for (int idx = tid; idx < RA_TMAX * RA_TMAX; idx += 256) {
        const int r = idx >> 7, c = idx & 127;
        float val = 0.f;
        int i, j;
        if (r < T && c < T && rel_shift_dst(r, c, T, i, j)) val = M2[i * RA_LD + j];
        M1[r * RA_LD + c] = val;
    }
    __syncthreads();
    // dq = dS k + dPR p (their column sums are the u / v bias gradients); dk = dS^T (q + u); dpos = dPR^T (q + v)
    T_* dq = (T_*)a.dq + base;
    T_* dk = (T_*)a.dk + base;
    float* dpos = a.dpos_part + (long)b * T * a.D + h * hd;
    for (int c0 = 0; c0 < hd; c0 += 128) {
        {
            float accc[8][8], accp[8][8], rk[8], rp[8];
            zero_tile(accc); zero_tile(accp);
            fetch_cols128<T_>(rk, k, a.D, T, 0, c0, hd, nullptr, tid);
            fetch_cols128<float>(rp, p, a.D, T, 0, c0, hd, nullptr, tid);
            for (int k0 = 0; k0 < T; k0 += RA_DC) {
                __syncthreads();
                put_cols128(St0, rk, tid); put_cols128(St1, rp, tid);
                __syncthreads();
                if (k0 + RA_DC < T) {
                    fetch_cols128<T_>(rk, k, a.D, T, k0 + RA_DC, c0, hd, nullptr, tid);
                    fetch_cols128<float>(rp, p, a.D, T, k0 + RA_DC, c0, hd, nullptr, tid);
                }
                mat_times_chunk<false>(accc, M2, St0, k0, ti, tj);
                mat_times_chunk<false>(accp, M1, St1, k0, ti, tj);
            }
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                float su = 0.f, sv = 0.f;
#pragma unroll
                for (int r = 0; r < 8; ++r) { su += accc[r][c]; sv += accp[r][c]; accc[r][c] += accp[r][c]; }
                const int col = c0 + tj + 16 * c;
                if (col < hd) { atomicAdd(&colsum[col], su); atomicAdd(&colsum[hd + col], sv); }
            }
            store_tile<T_>(dq, a.D, accc, T, c0, hd, ti, tj);
        }
        {
            float acck[8][8], accq[8][8], ru[8], rv[8];
            zero_tile(acck); zero_tile(accq);
            fetch_cols128<T_>(ru, q, a.D, T, 0, c0, hd, a.u_bias + h * hd, tid);
            fetch_cols128<T_>(rv, q, a.D, T, 0, c0, hd, a.v_bias + h * hd, tid);
            for (int k0 = 0; k0 < T; k0 += RA_DC) {
                __syncthreads();
                put_cols128(St0, ru, tid); put_cols128(St1, rv, tid);
                __syncthreads();
                if (k0 + RA_DC < T) {
                    fetch_cols128<T_>(ru, q, a.D, T, k0 + RA_DC, c0, hd, a.u_bias + h * hd, tid);
                    fetch_cols128<T_>(rv, q, a.D, T, k0 + RA_DC, c0, hd, a.v_bias + h * hd, tid);
                }
                mat_times_chunk<true>(acck, M2, St0, k0, ti, tj);
                mat_times_chunk<true>(accq, M1, St1, k0, ti, tj);
            }
            store_tile<T_>(dk, a.D, acck, T, c0, hd, ti, tj);
            store_tile<float>(dpos, a.D, accq, T, c0, hd, ti, tj);
        }
    }
    __syncthreads();
    float* dub = a.dbias_part + (long)b * 2 * a.D + h * hd;
    for (int c = tid; c < hd; c += 256) { dub[c] = colsum[c]; dub[a.D + c] = colsum[hd + c]; }
}

size_t relattn_lds(int hd, bool bwd) {
    return (size_t)(2 * RA_MAT + (bwd ? 2 : 3) * RA_STG + (bwd ? 2 * hd : 0)) * sizeof(float);
}

}  // namespace

// plain scaled-dot-product attention of short sequences (nn.MultiheadAttention inside the Transformer decoder): the same kernels with a
// zero positional table / zero biases and scale = 1 / sqrt(head_dim)
static thread_local float g_relattn_scale_override = 0.f;
extern "C" int pseld_relattn_fwd(int dtype, const void* q, const void* k, const void* v, const float* pos, const float* u_bias,
                                 const float* v_bias, const void* mask, float mask_scale, void* out, float* attn, int B, int T, int D,
                                 int heads, void* stream) {
    PSELD_CHECK_ARG(q && k && v && pos && u_bias && v_bias && out && attn, "relattn_fwd: null pointer");
    PSELD_CHECK_ARG(B > 0 && T > 0 && T <= RA_TMAX && heads > 0 && D % heads == 0, "relattn_fwd: bad geometry (T <= 128)");
    RelAttnArgs a; memset(&a, 0, sizeof(a));
    a.q = q; a.k = k; a.v = v; a.pos = pos; a.u_bias = u_bias; a.v_bias = v_bias; a.mask = mask; a.out = out; a.attn = attn;
    a.B = B; a.T = T; a.D = D; a.heads = heads; a.scale = g_relattn_scale_override > 0.f ? g_relattn_scale_override : 1.0f / sqrtf((float)D); a.mask_scale = mask_scale;
    hipStream_t s = (hipStream_t)stream;
    const size_t lds = relattn_lds(D / heads, false);
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
    PSELD_CHECK_ARG(B > 0 && T > 0 && T <= RA_TMAX && heads > 0 && D % heads == 0 && relattn_lds(D / heads, true) <= 160 * 1024,
                    "relattn_bwd: bad geometry (T <= 128, head_dim <= 512)");
    PSELD_CHECK_ARG(workspace_bytes >= pseld_relattn_bwd_workspace(B, T, D), "relattn_bwd: workspace too small");
    RelAttnArgs a; memset(&a, 0, sizeof(a));
    a.q = q; a.k = k; a.v = v; a.pos = pos; a.u_bias = u_bias; a.v_bias = v_bias; a.mask = mask; a.attn = const_cast<float*>(attn);
    a.dout = dout; a.dq = dq; a.dk = dk; a.dv = dv; a.dpos_part = workspace; a.dbias_part = workspace + (long)B * T * D;
    a.B = B; a.T = T; a.D = D; a.heads = heads; a.scale = g_relattn_scale_override > 0.f ? g_relattn_scale_override : 1.0f / sqrtf((float)D); a.mask_scale = mask_scale;
    hipStream_t s = (hipStream_t)stream;
    const size_t lds = relattn_lds(D / heads, true);
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

// ---- GRU decoder cell (CRNN decoder='gru', models/components/model_utilities.py:249-252 nn.GRU) ------------------------------------
// The input projections of ALL timesteps and the weight gradients are ordinary GEMMs; per timestep only the recurrent
// product h_{t-1} W_hh^T (a GEMM with B rows) and this gate kernel run. Gate order r | z | n (torch.nn.GRU):
//   r = sigmoid(gi_r + gh_r), z = sigmoid(gi_z + gh_z), n = tanh(gi_n + r * gh_n), h = (1 - z) * n + z * h_prev
namespace {

template <typename T>
__global__ void gru_gate_fwd_kernel(const T* __restrict__ gi, long gi_stride, const T* __restrict__ gh, const T* __restrict__ hprev, long hp_stride,
                                    T* __restrict__ h, long h_stride, T* __restrict__ gates, int H, long total) {
    const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;        // over (b, j < H)
    if (id >= total) return;
    const int j = (int)(id % H);
    const long b = id / H;
    const T* gib = gi + b * gi_stride;
    const T* ghb = gh + b * 3 * H;
    const float r = sigmoidf_(to_f32<T>(gib[j]) + to_f32<T>(ghb[j]));
    const float z = sigmoidf_(to_f32<T>(gib[H + j]) + to_f32<T>(ghb[H + j]));
    const float ghn = to_f32<T>(ghb[2 * H + j]);
    const float n = tanhf(to_f32<T>(gib[2 * H + j]) + r * ghn);
    const float hp = hprev ? to_f32<T>(hprev[b * hp_stride + j]) : 0.f;
    h[b * h_stride + j] = from_f32<T>((1.f - z) * n + z * hp);
    T* g = gates + b * 4 * H;                                            // kept for the backward: r | z | n | gh_n
    g[j] = from_f32<T>(r); g[H + j] = from_f32<T>(z); g[2 * H + j] = from_f32<T>(n); g[3 * H + j] = from_f32<T>(ghn);
}
// dh: gradient wrt h_t (output gradient + the carry from t+1). Writes dgi_t = (dr_pre | dz_pre | dn_pre), dgh_t =
// (dr_pre | dz_pre | dn_pre * r) and the direct part of the carry, dh * z (the W_hh part is added by the caller's GEMM).
template <typename T>
__global__ void gru_gate_bwd_kernel(const T* __restrict__ dh, long dh_stride, const T* __restrict__ carry, const T* __restrict__ gates,
                                    const T* __restrict__ hprev, long hp_stride, T* __restrict__ dgi, long dgi_stride, T* __restrict__ dgh,
                                    T* __restrict__ dhprev, T* __restrict__ hprev_out, int H, long total) {
    const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= total) return;
    const int j = (int)(id % H);
    const long b = id / H;
    const T* g = gates + b * 4 * H;
    const float r = to_f32<T>(g[j]), z = to_f32<T>(g[H + j]), n = to_f32<T>(g[2 * H + j]), ghn = to_f32<T>(g[3 * H + j]);
    const float hp = hprev ? to_f32<T>(hprev[b * hp_stride + j]) : 0.f;
    const float d = to_f32<T>(dh[b * dh_stride + j]) + (carry ? to_f32<T>(carry[b * H + j]) : 0.f);
    const float dn = d * (1.f - z) * (1.f - n * n);
    const float dz = d * (hp - n) * z * (1.f - z);
    const float dr = dn * ghn * r * (1.f - r);
    T* a = dgi + b * dgi_stride;
    a[j] = from_f32<T>(dr); a[H + j] = from_f32<T>(dz); a[2 * H + j] = from_f32<T>(dn);
    T* c = dgh + b * 3 * H;
    c[j] = from_f32<T>(dr); c[H + j] = from_f32<T>(dz); c[2 * H + j] = from_f32<T>(dn * r);
    dhprev[b * H + j] = from_f32<T>(d * z);
    if (hprev_out) hprev_out[b * H + j] = from_f32<T>(hp);            // contiguous copy of h_{t-1} for the W_hh weight-gradient GEMM
}

}  // namespace

/* gi rows b at gi + b*gi_stride (3H wide: the slice of timestep t of the all-timestep input projection), gh [B, 3H] =
 * h_prev W_hh^T + b_hh, hprev rows at hp_stride (NULL: zeros), h rows at h_stride (the output sequence), gates [B, 4H] */
extern "C" int pseld_gru_gate_fwd(int dtype, const void* gi, long gi_stride, const void* gh, const void* hprev, long hp_stride, void* h,
                                  long h_stride, void* gates, int B, int H, void* stream) {
    PSELD_CHECK_ARG(gi && gh && h && gates && B > 0 && H > 0, "gru_gate_fwd: bad argument");
    hipStream_t s = (hipStream_t)stream;
    const long total = (long)B * H;
    CF_DISPATCH("gru_gate_fwd", hipLaunchKernelGGL(gru_gate_fwd_kernel<T>, dim3(pseld_cdiv(total, 256)), dim3(256), 0, s, (const T*)gi, gi_stride,
                                                   (const T*)gh, (const T*)hprev, hp_stride, (T*)h, h_stride, (T*)gates, H, total));
}
static int gru_gate_bwd_impl(int dtype, const void* dh, long dh_stride, const void* carry, const void* gates, const void* hprev,
                             long hp_stride, void* dgi, long dgi_stride, void* dgh, void* dhprev, void* hprev_out, int B, int H, void* stream) {
    PSELD_CHECK_ARG(dh && gates && dgi && dgh && dhprev && B > 0 && H > 0, "gru_gate_bwd: bad argument");
    hipStream_t s = (hipStream_t)stream;
    const long total = (long)B * H;
    CF_DISPATCH("gru_gate_bwd", hipLaunchKernelGGL(gru_gate_bwd_kernel<T>, dim3(pseld_cdiv(total, 256)), dim3(256), 0, s, (const T*)dh, dh_stride,
                                                   (const T*)carry, (const T*)gates, (const T*)hprev, hp_stride, (T*)dgi, dgi_stride, (T*)dgh,
                                                   (T*)dhprev, (T*)hprev_out, H, total));
}
extern "C" int pseld_gru_gate_bwd(int dtype, const void* dh, long dh_stride, const void* carry, const void* gates, const void* hprev,
                                  long hp_stride, void* dgi, long dgi_stride, void* dgh, void* dhprev, int B, int H, void* stream) {
    return gru_gate_bwd_impl(dtype, dh, dh_stride, carry, gates, hprev, hp_stride, dgi, dgi_stride, dgh, dhprev, nullptr, B, H, stream);
}

namespace {
template <typename T>
__global__ void gru_bias_rows_kernel(const float* __restrict__ b, T* __restrict__ out, int N, long total) {
    const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (id < total) out[id] = from_f32<T>(b[id % N]);
}
}  // namespace
static int pseld_gru_bias_rows(int dtype, const float* b, void* out, int B, int N, void* stream) {
    const long total = (long)B * N;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == PSELD_BF16) hipLaunchKernelGGL(gru_bias_rows_kernel<bf16_t>, dim3(pseld_cdiv(total, 256)), dim3(256), 0, s, b, (bf16_t*)out, N, total);
    else hipLaunchKernelGGL(gru_bias_rows_kernel<float>, dim3(pseld_cdiv(total, 256)), dim3(256), 0, s, b, (float*)out, N, total);
    PSELD_LAUNCH_CHECK("gru_bias_rows");
    return PSELD_OK;
}
// ---- fused GRU timestep kernels (bf16): recurrent product + gates in ONE launch per step -------------------------------------------------
// Dependent tiny launches cost ~8 us each here, so the step count, not the arithmetic, sets the time. Forward: a workgroup owns 8 hidden
// units = the rows (j, H+j, 2H+j) of W_hh; the MFMA tile is [32 weight rows (gate-major, 8 padded)] x [64 samples], K split over the
// four waves, operands straight from global; after the LDS reduction lane (sample, half) holds r/z/n pre-activations of four units and
// finishes the cell. Backward: a workgroup owns 32 hidden units = rows of W_hh^T; it forms the carry dh_{t} += dgh_{t+1} W_hh for its
// units and immediately runs the gate backward of step t for them (everything it needs is unit-local).
namespace {

typedef __bf16 bf16x8_ __attribute__((ext_vector_type(8)));
typedef float f32x16_ __attribute__((ext_vector_type(16)));

struct GruStepArgs {
    const bf16_t *gi, *w, *hprev, *gates_in, *dseq, *dgh_next, *direct_in;   // w: W_hh [3H,H] (fwd) or W_hh^T [H,3H] (bwd)
    const float* b_hh;
    bf16_t *h, *gates, *dgi, *dgh, *direct_out, *hprev_out;
    long gi_stride, row;                 // sample strides of gi / seq-like tensors
    int B, H, first;                     // first: no previous hidden state (fwd) / no carry yet (bwd)
};

__device__ __forceinline__ void skinny_accumulate(f32x16_ (&acc)[2], const bf16_t* wrow, const bf16_t* a0, const bf16_t* a1, bool two, int ksteps,
                                                  int wave, int waves) {
    for (int ks = wave; ks < ksteps; ks += 8 * waves) {
        bf16x8_ fb[8], fa0[8], fa1[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const bool in = ks + waves * u < ksteps;
            const int kc = in ? (ks + waves * u) << 4 : 0;
            fb[u] = *(const bf16x8_*)(wrow + kc);
            fa0[u] = *(const bf16x8_*)(a0 + kc);
            if (two) fa1[u] = *(const bf16x8_*)(a1 + kc);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (ks + waves * u < ksteps) {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[u], fa0[u], acc[0], 0, 0, 0);
                if (two) acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[u], fa1[u], acc[1], 0, 0, 0);
            }
    }
}

// Up to GRU_MAX_SEQS independent recurrences of one geometry (the two directions of a layer, the six decoders of the EINV2 tail)
// advance in the SAME launch: blockIdx.y picks the descriptor. The step count, not the work per step, sets the time.
constexpr int GRU_MAX_SEQS = 12;
struct GruMultiArgs { GruStepArgs s[GRU_MAX_SEQS]; };

__global__ __launch_bounds__(256) void gru_step_fwd_kernel(GruMultiArgs m) {
    const GruStepArgs& a = m.s[blockIdx.y];
    __shared__ float red[4][2][16][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, h2 = lane >> 5;
    const int H = a.H, j0 = blockIdx.x * 8;
    // the cell operands of this thread's (unit, sample) pairs are fetched first: their latency hides behind the recurrent product
    float pgi[2][3], pbh[2][3], php[2];
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int idx = threadIdx.x + 256 * it, u = idx & 7, b = idx >> 3, j = j0 + u;
        if (idx < 8 * a.B) {
            const bf16_t* gib = a.gi + b * a.gi_stride;
#pragma unroll
            for (int gt = 0; gt < 3; ++gt) { pgi[it][gt] = (float)gib[gt * H + j]; pbh[it][gt] = a.b_hh[gt * H + j]; }
            php[it] = a.first ? 0.f : (float)a.hprev[b * a.row + j];
        }
    }
    if (!a.first) {
        f32x16_ acc[2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
        const int gate = min(r >> 3, 2), unit = r & 7;                         // fragment row r -> W_hh row gate*H + j0 + unit
        const bf16_t* wrow = a.w + (long)(gate * H + j0 + unit) * H + 8 * h2;
        const bf16_t* a0 = a.hprev + (long)min(r, a.B - 1) * a.row + 8 * h2;
        const bf16_t* a1 = a.hprev + (long)min(32 + r, a.B - 1) * a.row + 8 * h2;
        skinny_accumulate(acc, wrow, a0, a1, a.B > 32, H >> 4, wave, 4);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) red[wave][t][e][lane] = acc[t][e];
        __syncthreads();
    }
    // the cell update of the 8 units x B samples of this workgroup, spread over all 256 threads (unit fastest)
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int idx = threadIdx.x + 256 * it;
        if (idx >= 8 * a.B) break;
        const int u = idx & 7, b = idx >> 3, j = j0 + u;
        float g3[3];
#pragma unroll
        for (int gt = 0; gt < 3; ++gt) {
            float v = pbh[it][gt];
            if (!a.first) {
                const int t = b >> 5, e = 4 * gt + (u & 3), ln = (b & 31) + 32 * (u >> 2);          // tile row gt*8 + u
#pragma unroll
                for (int w = 0; w < 4; ++w) v += red[w][t][e][ln];
            }
            g3[gt] = v;
        }
        const float rr = sigmoidf_(pgi[it][0] + g3[0]);
        const float zz = sigmoidf_(pgi[it][1] + g3[1]);
        const float nn = tanhf(pgi[it][2] + rr * g3[2]);
        const float hp = php[it];
        a.h[b * a.row + j] = (bf16_t)((1.f - zz) * nn + zz * hp);
        bf16_t* gs = a.gates + (long)b * 4 * H;
        gs[j] = (bf16_t)rr; gs[H + j] = (bf16_t)zz; gs[2 * H + j] = (bf16_t)nn; gs[3 * H + j] = (bf16_t)g3[2];
    }
}

// UB hidden units per workgroup: the 32-row MFMA tile is only partly used for UB < 32, but the launch is latency-bound and more,
// shorter workgroups finish sooner (each reads UB rows of W_hh^T = UB x 3H weights)
template <int UB>
__global__ __launch_bounds__(512) void gru_step_bwd_kernel(GruMultiArgs m) {
    const GruStepArgs& a = m.s[blockIdx.y];
    __shared__ float red[8][2][16][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, h2 = lane >> 5;
    const int H = a.H, j0 = blockIdx.x * UB;
    // gate values, h_{t-1}, the output gradient and the direct carry term of this thread's (unit, sample) pairs: fetched first
    constexpr int ITERS = UB / 8;                                               // UB * 64 samples / 512 threads
    float pg[ITERS][4], php[ITERS], pds[ITERS], pdir[ITERS];
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int idx = threadIdx.x + 512 * it, jl = idx % UB, b = idx / UB, j = j0 + jl;
        if (idx < UB * a.B) {
            const bf16_t* g = a.gates_in + (long)b * 4 * H;
#pragma unroll
            for (int q = 0; q < 4; ++q) pg[it][q] = (float)g[q * H + j];
            php[it] = a.hprev ? (float)a.hprev[b * a.row + j] : 0.f;
            pds[it] = (float)a.dseq[b * a.row + j];
            pdir[it] = a.first ? 0.f : (float)a.direct_in[(long)b * H + j];
        }
    }
    if (!a.first) {                                                             // carry = direct(t+1) + dgh(t+1) W_hh for units j0..j0+31
        f32x16_ acc[2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
        const bf16_t* wrow = a.w + (long)(j0 + (r % UB)) * 3 * H + 8 * h2;       // W_hh^T row j (rows >= UB of the tile are unused)
        const bf16_t* a0 = a.dgh_next + (long)min(r, a.B - 1) * 3 * H + 8 * h2;
        const bf16_t* a1 = a.dgh_next + (long)min(32 + r, a.B - 1) * 3 * H + 8 * h2;
        skinny_accumulate(acc, wrow, a0, a1, a.B > 32, (3 * H) >> 4, wave, 8);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) red[wave][t][e][lane] = acc[t][e];
        __syncthreads();
    }
    // the gate backward of step t for the 32 units x B samples of this workgroup, spread over all 512 threads (unit fastest)
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int idx = threadIdx.x + 512 * it;
        if (idx >= UB * a.B) break;
        const int jl = idx % UB, b = idx / UB, j = j0 + jl;
        float carry = 0.f;
        if (!a.first) {
            const int t = b >> 5, e = (jl & 3) + 4 * (jl >> 3), ln = (b & 31) + 32 * ((jl >> 2) & 1);   // where the MFMA tile keeps (j, b)
            carry = pdir[it];
#pragma unroll
            for (int w = 0; w < 8; ++w) carry += red[w][t][e][ln];
            carry = (float)(bf16_t)carry;                                      // the unfused path stores the carry in bf16
        }
        const float rr = pg[it][0], zz = pg[it][1], nn = pg[it][2], ghn = pg[it][3];
        const float hp = php[it];
        const float d = pds[it] + carry;
        const float dn = d * (1.f - zz) * (1.f - nn * nn);
        const float dz = d * (hp - nn) * zz * (1.f - zz);
        const float dr = dn * ghn * rr * (1.f - rr);
        bf16_t* gi_ = a.dgi + b * a.gi_stride;
        gi_[j] = (bf16_t)dr; gi_[H + j] = (bf16_t)dz; gi_[2 * H + j] = (bf16_t)dn;
        bf16_t* gh_ = a.dgh + (long)b * 3 * H;
        gh_[j] = (bf16_t)dr; gh_[H + j] = (bf16_t)dz; gh_[2 * H + j] = (bf16_t)(dn * rr);
        a.direct_out[(long)b * H + j] = (bf16_t)(d * zz);
        if (a.hprev_out) a.hprev_out[(long)b * H + j] = (bf16_t)hp;
    }
}

}  // namespace

// ---- the whole recurrence of one GRU layer and direction in ONE C-ABI call: the T x (B-row GEMM + gate kernel) launches are
// issued from here instead of from a Python loop (the host dispatch, not the GPU, was the limit: 2 000 launches per step) ------------
extern "C" int pseld_gemm(int dtype, int trans_a, int trans_b, const void* A, const void* B, void* C, int M, int N, int K, int lda, int ldb,
                          int ldc, const float* bias, const void* resid, int ldr, const float* rowscale, int rows_per_scale, const void* aux,
                          int ldaux, int epi, int pro, void* c2, void* stream);

/* gi [B, T, 3H] (input projections incl. b_ih), w_hh [3H, H] in the compute dtype, b_hh f32 [3H]; seq = the layer's output
 * [B, T, ld_seq] with this direction's H columns starting at seq (caller offsets the pointer); gates [T, B, 4H]; gh scratch
 * [B, 3H]; reverse != 0 walks t = T-1 .. 0. */
static void gru_bwd_launch(const GruMultiArgs& m, int H, int cnt, hipStream_t s) {
    // measured: 8 units per workgroup win while the launch is short of workgroups (one or two recurrences), 32 once many recurrences
    // share it (every workgroup re-reads the [B, 3H] gate-gradient block: 12 x 128 workgroups of 8 units are slower than 12 x 32 of 32)
    const int forced = pseld_knob(KNOB_GRU_BWD_UNITS, 0);
    const int ub = forced ? forced : (cnt <= 2 ? 8 : 32);
    if (ub == 32) hipLaunchKernelGGL(gru_step_bwd_kernel<32>, dim3(H / 32, cnt), dim3(512), 0, s, m);
    else if (ub == 16) hipLaunchKernelGGL(gru_step_bwd_kernel<16>, dim3(H / 16, cnt), dim3(512), 0, s, m);
    else hipLaunchKernelGGL(gru_step_bwd_kernel<8>, dim3(H / 8, cnt), dim3(512), 0, s, m);
}

static void gru_fwd_step_args(GruStepArgs& a, const void* gi, const void* w_hh, const float* b_hh, void* seq, long ld_seq, void* gates, int B, int T,
                              int H, int reverse, int k) {
    const int t = reverse ? T - 1 - k : k;
    const int prev = k > 0 ? (reverse ? T - k : k - 1) : -1;
    memset(&a, 0, sizeof(a));
    a.gi = (const bf16_t*)gi + (size_t)t * 3 * H; a.gi_stride = (long)T * 3 * H; a.w = (const bf16_t*)w_hh; a.b_hh = b_hh;
    a.hprev = prev >= 0 ? (const bf16_t*)seq + (size_t)prev * ld_seq : nullptr; a.h = (bf16_t*)seq + (size_t)t * ld_seq; a.row = (long)T * ld_seq;
    a.gates = (bf16_t*)gates + (size_t)t * B * 4 * H; a.B = B; a.H = H; a.first = prev < 0;
}

static void gru_bwd_step_args(GruStepArgs& a, const void* dseq, const void* seq, long ld_seq, const void* gates, const void* w_hh_t, void* dgi,
                              void* dgh, void* hprev_all, void* carry, void* direct, int B, int T, int H, int reverse, int k) {
    // step k consumes the dgh / direct of step k+1 (the time index processed just before in the loop) and ping-pongs `direct`
    const int t = reverse ? T - 1 - k : k;
    const int tp = k > 0 ? (reverse ? T - k : k - 1) : -1;
    const int tn = k < T - 1 ? (reverse ? T - 2 - k : k + 1) : -1;
    memset(&a, 0, sizeof(a));
    a.w = (const bf16_t*)w_hh_t; a.first = tn < 0; a.B = B; a.H = H; a.row = (long)T * ld_seq; a.gi_stride = (long)T * 3 * H;
    a.dgh_next = tn >= 0 ? (const bf16_t*)dgh + (size_t)tn * B * 3 * H : nullptr;
    a.direct_in = (const bf16_t*)((k & 1) ? carry : direct); a.direct_out = (bf16_t*)((k & 1) ? direct : carry);
    a.gates_in = (const bf16_t*)gates + (size_t)t * B * 4 * H; a.dseq = (const bf16_t*)dseq + (size_t)t * ld_seq;
    a.hprev = tp >= 0 ? (const bf16_t*)seq + (size_t)tp * ld_seq : nullptr; a.dgi = (bf16_t*)dgi + (size_t)t * 3 * H;
    a.dgh = (bf16_t*)dgh + (size_t)t * B * 3 * H;
    a.hprev_out = tp >= 0 ? (bf16_t*)hprev_all + (size_t)t * B * H : nullptr;
}

extern "C" int pseld_gru_seq_fwd(int dtype, const void* gi, const void* w_hh, const float* b_hh, void* seq, long ld_seq, void* gates, void* gh,
                                 int B, int T, int H, int reverse, void* stream) {
    PSELD_CHECK_ARG(gi && w_hh && b_hh && seq && gates && gh && B > 0 && T > 0 && H > 0 && H % 8 == 0 && ld_seq % 8 == 0, "gru_seq_fwd: bad argument");
    PSELD_CHECK_ARG(dtype == PSELD_BF16 || dtype == PSELD_F32, "gru_seq_fwd: unknown dtype");
    const size_t es = dtype == PSELD_BF16 ? 2 : 4;
    const long row = (long)T * ld_seq;                      // elements between consecutive samples of seq
    int prev = -1;
    const bool fused = dtype == PSELD_BF16 && B <= 64 && H % 16 == 0;
    for (int k = 0; k < T; ++k) {
        const int t = reverse ? T - 1 - k : k;
        const char* hprev = prev >= 0 ? (const char*)seq + (size_t)prev * ld_seq * es : nullptr;
        int rc;
        if (fused) {
            GruMultiArgs m;
            gru_fwd_step_args(m.s[0], gi, w_hh, b_hh, seq, ld_seq, gates, B, T, H, reverse, k);
            hipLaunchKernelGGL(gru_step_fwd_kernel, dim3(H / 8, 1), dim3(256), 0, (hipStream_t)stream, m);
            prev = t;
            continue;
        }
        if (hprev) {
            rc = pseld_gemm(dtype, 0, 0, hprev, w_hh, gh, B, 3 * H, H, (int)row, H, 3 * H, b_hh, nullptr, 0, nullptr, 1, nullptr, 0, 1 /*EPI_BIAS*/,
                            0, nullptr, stream);
            if (rc != PSELD_OK) return rc;
        } else {
            // h_0 = 0: gh = b_hh (a GEMM against a zero row would do; the gate kernel reads gh, so broadcast the bias instead)
            rc = pseld_gru_bias_rows(dtype, b_hh, gh, B, 3 * H, stream);
            if (rc != PSELD_OK) return rc;
        }
        rc = pseld_gru_gate_fwd(dtype, (const char*)gi + (size_t)t * 3 * H * es, (long)T * 3 * H, gh, hprev, row, (char*)seq + (size_t)t * ld_seq * es, row,
                                (char*)gates + (size_t)t * B * 4 * H * es, B, H, stream);
        if (rc != PSELD_OK) return rc;
        prev = t;
    }
    return PSELD_OK;
}

/* BPTT of the same recurrence. dseq: gradient wrt this direction's slice of the output [B, T, ld_seq]; w_hh_t [H, 3H] (the
 * transposed copy: dh_prev = dgh W_hh is then an NT product) or NULL with w_hh [3H, H]; writes dgi [B, T, 3H], dgh [T, B, 3H] and
 * hprev_all [T, B, H] (the h_{t-1} of every step, zeros for the first) for the weight-gradient GEMMs; carry / direct scratch [B, H]. */
extern "C" int pseld_gru_seq_bwd(int dtype, const void* dseq, const void* seq, long ld_seq, const void* gates, const void* w_hh,
                                 const void* w_hh_t, void* dgi, void* dgh, void* hprev_all, void* carry, void* direct, int B, int T, int H,
                                 int reverse, void* stream) {
    PSELD_CHECK_ARG(dseq && seq && gates && w_hh && dgi && dgh && hprev_all && carry && direct && B > 0 && T > 0 && H > 0 && H % 8 == 0,
                    "gru_seq_bwd: bad argument");
    PSELD_CHECK_ARG(dtype == PSELD_BF16 || dtype == PSELD_F32, "gru_seq_bwd: unknown dtype");
    const size_t es = dtype == PSELD_BF16 ? 2 : 4;
    const long row = (long)T * ld_seq;
    bool have_carry = false;
    const bool fused = dtype == PSELD_BF16 && B <= 64 && H % 32 == 0 && w_hh_t != nullptr;
    for (int k = T - 1; k >= 0; --k) {                       // reverse of the processing order
        const int t = reverse ? T - 1 - k : k;
        const int tp = k > 0 ? (reverse ? T - k : k - 1) : -1;
        const char* hprev = tp >= 0 ? (const char*)seq + (size_t)tp * ld_seq * es : nullptr;
        char* dgh_t = (char*)dgh + (size_t)t * B * 3 * H * es;
        if (fused) {
            GruMultiArgs m;
            gru_bwd_step_args(m.s[0], dseq, seq, ld_seq, gates, w_hh_t, dgi, dgh, hprev_all, carry, direct, B, T, H, reverse, k);
            gru_bwd_launch(m, H, 1, (hipStream_t)stream);
            continue;
        }
        int rc = gru_gate_bwd_impl(dtype, (const char*)dseq + (size_t)t * ld_seq * es, row, have_carry ? carry : nullptr,
                                   (const char*)gates + (size_t)t * B * 4 * H * es, hprev, row, (char*)dgi + (size_t)t * 3 * H * es, (long)T * 3 * H,
                                   dgh_t, direct, tp >= 0 ? (char*)hprev_all + (size_t)t * B * H * es : nullptr, B, H, stream);
        if (rc != PSELD_OK) return rc;
        if (tp >= 0) {
            // carry = direct + dgh_t W_hh
            if (w_hh_t) rc = pseld_gemm(dtype, 0, 0, dgh_t, w_hh_t, carry, B, H, 3 * H, 3 * H, 3 * H, H, nullptr, direct, H, nullptr, 1, nullptr, 0,
                                        2 /*EPI_RESID*/, 0, nullptr, stream);
            else rc = pseld_gemm(dtype, 0, 1, dgh_t, w_hh, carry, B, H, 3 * H, 3 * H, H, H, nullptr, direct, H, nullptr, 1, nullptr, 0, 2, 0, nullptr,
                                 stream);
            if (rc != PSELD_OK) return rc;
            have_carry = true;
        }
    }
    return PSELD_OK;
}



/* nn.MultiheadAttention core for sequences of up to 128 frames (components/model_utilities.py:256-259 nn.TransformerEncoderLayer):
 * softmax(q k^T / sqrt(head_dim)) (x dropout mask) v on the relative-attention kernels with a zero positional table. zeros: f32
 * buffer of at least T*D zeros; scratch (backward): f32 [T*D + 2*D] that receives the (meaningless) positional gradients. */
extern "C" int pseld_sdpa_small_fwd(int dtype, const void* q, const void* k, const void* v, const float* zeros, const void* mask, float mask_scale,
                                    void* out, float* attn, int B, int T, int D, int heads, void* stream) {
    PSELD_CHECK_ARG(zeros && heads > 0 && D % heads == 0, "sdpa_small_fwd: bad argument");
    g_relattn_scale_override = 1.0f / sqrtf((float)(D / heads));
    const int rc = pseld_relattn_fwd(dtype, q, k, v, zeros, zeros, zeros, mask, mask_scale, out, attn, B, T, D, heads, stream);
    g_relattn_scale_override = 0.f;
    return rc;
}
extern "C" int pseld_sdpa_small_bwd(int dtype, const void* q, const void* k, const void* v, const float* zeros, const void* mask, float mask_scale,
                                    const float* attn, const void* dout, void* dq, void* dk, void* dv, float* scratch, int B, int T, int D,
                                    int heads, float* workspace, long workspace_bytes, void* stream) {
    PSELD_CHECK_ARG(zeros && scratch && heads > 0 && D % heads == 0, "sdpa_small_bwd: bad argument");
    g_relattn_scale_override = 1.0f / sqrtf((float)(D / heads));
    const int rc = pseld_relattn_bwd(dtype, q, k, v, zeros, zeros, zeros, mask, mask_scale, attn, dout, dq, dk, dv, scratch, scratch + (long)T * D,
                                     scratch + (long)T * D + D, B, T, D, heads, workspace, workspace_bytes, stream);
    g_relattn_scale_override = 0.f;
    return rc;
}

// ---- n independent recurrences of one geometry advanced together (host arrays of n device pointers) --------------------------------------
extern "C" int pseld_gru_multi_fwd(int dtype, int n, const void* const* gi, const void* const* w_hh, const float* const* b_hh, void* const* seq,
                                   long ld_seq, void* const* gates, void* const* gh, const int* reverse, int B, int T, int H, void* stream) {
    PSELD_CHECK_ARG(n > 0 && gi && w_hh && b_hh && seq && gates && gh && reverse, "gru_multi_fwd: bad argument");
    const bool fused = dtype == PSELD_BF16 && B <= 64 && H % 16 == 0 && ld_seq % 8 == 0;
    if (!fused) {
        for (int i = 0; i < n; ++i) {
            const int rc = pseld_gru_seq_fwd(dtype, gi[i], w_hh[i], b_hh[i], seq[i], ld_seq, gates[i], gh[i], B, T, H, reverse[i], stream);
            if (rc != PSELD_OK) return rc;
        }
        return PSELD_OK;
    }
    for (int i = 0; i < n; ++i) PSELD_CHECK_ARG(gi[i] && w_hh[i] && b_hh[i] && seq[i] && gates[i], "gru_multi_fwd: null pointer");
    PSELD_CHECK_ARG(B > 0 && T > 0 && H > 0, "gru_multi_fwd: bad geometry");
    for (int i0 = 0; i0 < n; i0 += GRU_MAX_SEQS) {
        const int cnt = n - i0 < GRU_MAX_SEQS ? n - i0 : GRU_MAX_SEQS;
        for (int k = 0; k < T; ++k) {
            GruMultiArgs m;
            for (int i = 0; i < cnt; ++i)
                gru_fwd_step_args(m.s[i], gi[i0 + i], w_hh[i0 + i], b_hh[i0 + i], seq[i0 + i], ld_seq, gates[i0 + i], B, T, H, reverse[i0 + i], k);
            hipLaunchKernelGGL(gru_step_fwd_kernel, dim3(H / 8, cnt), dim3(256), 0, (hipStream_t)stream, m);
        }
    }
    PSELD_LAUNCH_CHECK("gru_multi_fwd");
    return PSELD_OK;
}

extern "C" int pseld_gru_multi_bwd(int dtype, int n, const void* const* dseq, const void* const* seq, long ld_seq, const void* const* gates,
                                   const void* const* w_hh, const void* const* w_hh_t, void* const* dgi, void* const* dgh, void* const* hprev_all,
                                   void* const* carry, void* const* direct, const int* reverse, int B, int T, int H, void* stream) {
    PSELD_CHECK_ARG(n > 0 && dseq && seq && gates && w_hh && dgi && dgh && hprev_all && carry && direct && reverse, "gru_multi_bwd: bad argument");
    bool fused = dtype == PSELD_BF16 && B <= 64 && H % 32 == 0 && w_hh_t != nullptr;
    for (int i = 0; fused && i < n; ++i) fused = w_hh_t[i] != nullptr;
    if (!fused) {
        for (int i = 0; i < n; ++i) {
            const int rc = pseld_gru_seq_bwd(dtype, dseq[i], seq[i], ld_seq, gates[i], w_hh[i], w_hh_t ? w_hh_t[i] : nullptr, dgi[i], dgh[i],
                                             hprev_all[i], carry[i], direct[i], B, T, H, reverse[i], stream);
            if (rc != PSELD_OK) return rc;
        }
        return PSELD_OK;
    }
    for (int i = 0; i < n; ++i)
        PSELD_CHECK_ARG(dseq[i] && seq[i] && gates[i] && dgi[i] && dgh[i] && hprev_all[i] && carry[i] && direct[i], "gru_multi_bwd: null pointer");
    PSELD_CHECK_ARG(B > 0 && T > 0 && H > 0, "gru_multi_bwd: bad geometry");
    for (int i0 = 0; i0 < n; i0 += GRU_MAX_SEQS) {
        const int cnt = n - i0 < GRU_MAX_SEQS ? n - i0 : GRU_MAX_SEQS;
        for (int k = T - 1; k >= 0; --k) {
            GruMultiArgs m;
            for (int i = 0; i < cnt; ++i)
                gru_bwd_step_args(m.s[i], dseq[i0 + i], seq[i0 + i], ld_seq, gates[i0 + i], w_hh_t[i0 + i], dgi[i0 + i], dgh[i0 + i],
                                  hprev_all[i0 + i], carry[i0 + i], direct[i0 + i], B, T, H, reverse[i0 + i], k);
            gru_bwd_launch(m, H, cnt, (hipStream_t)stream);
        }
    }
    PSELD_LAUNCH_CHECK("gru_multi_bwd");
    return PSELD_OK;
}
