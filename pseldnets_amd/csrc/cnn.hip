// Convolutional encoder of the CRNN networks (CNN8 / CNN12 = the PANNs CNN14 conv stack) on gfx950: everything around
// the MFMA GEMMs of the 3x3 convolutions. Activations live as NHWC rows [B*T*F, C] so that every convolution is
// im2col -> gemm.hip -> (col2im for the input gradient) and every BatchNorm2d is a per-column operation.
//
// Replaces (reference, /root/reference/src): models/accdoa.py:72-90 (scalar BN, conv stack, frequency mean, 'repeat'
// interpolation + 10-frame mean), models/components/backbone.py:6-60 (CNN8, CNN12), models/components/
// model_utilities.py:92-126 (ConvBlock: conv3x3 -> BatchNorm2d -> ReLU, twice, AvgPool2d), models/components/
// utils.py:25-52 (interpolate, method 'repeat') — and the autograd of each. All HBM-bound address maps / reductions;
// first version: correctness and parity first (the im2col matrix is materialised, k = c*9 + tap so that the
// reference's [Cout, Cin, 3, 3] weights are used as they are).
#include "common.h"

void pseld_reduce_slabs(const float* slabs, float* out, long n, int splits, long slab_stride, int accumulate,
                        hipStream_t stream);

namespace {

// X[(b,t,f), c] = feat[b,c,t,f] * scale[c,f] + shift[c,f] for c < Cin, 0 for the pad channels (Cp = padded count)
template <typename T>
__global__ void cnn_input_kernel(const float* __restrict__ feat, const float* __restrict__ ss, T* __restrict__ X, int Cin, int Tn,
                                 int Cp, long total) {
    const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= total) return;
    const int c = (int)(id % Cp);
    const long row = id / Cp;
    const int f = (int)(row & 63);
    const long bt = row >> 6;
    const long b = bt / Tn;
    const int t = (int)(bt - b * Tn);
    float v = 0.f;
    if (c < Cin) v = feat[((b * Cin + c) * Tn + t) * 64 + f] * ss[2 * (c * 64 + f)] + ss[2 * (c * 64 + f) + 1];
    X[id] = from_f32<T>(v);
}
// scalar-BN parameter gradients from dX[(b,t,f), c]: grid (blocks over (b,t), Cin), lane = f
template <typename T>
__global__ __launch_bounds__(256) void cnn_input_bwd_kernel(const float* __restrict__ feat, const float* __restrict__ mean_rstd,
                                                            const T* __restrict__ dX, float* __restrict__ part, int B, int Cin,
                                                            int Tn, int Cp, int rows_per_block) {
    __shared__ float red[4][64][2];
    const int f = threadIdx.x & 63, w = threadIdx.x >> 6, c = blockIdx.y;
    const long total = (long)B * Tn;
    const long beg = (long)blockIdx.x * rows_per_block, end = min(total, beg + rows_per_block);
    const float mean = mean_rstd[2 * (c * 64 + f)], rstd = mean_rstd[2 * (c * 64 + f) + 1];
    float dw = 0.f, db = 0.f;
    for (long q = beg + w; q < end; q += 4) {
        const long b = q / Tn;
        const int t = (int)(q - b * Tn);
        const float g = to_f32<T>(dX[(q * 64 + f) * Cp + c]);
        const float xh = (feat[((b * Cin + c) * Tn + t) * 64 + f] - mean) * rstd;
        dw += g * xh; db += g;
    }
    red[w][f][0] = dw; red[w][f][1] = db;
    __syncthreads();
    if (w == 0) {
        float* o = part + (((long)blockIdx.x * Cin + c) * 64 + f) * 2;
        o[0] = red[0][f][0] + red[1][f][0] + red[2][f][0] + red[3][f][0];
        o[1] = red[0][f][1] + red[1][f][1] + red[2][f][1] + red[3][f][1];
    }
}
__global__ void deinterleave2_kernel(const float* __restrict__ tot, int n, float* a, float* b) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { a[i] = tot[2 * i]; b[i] = tot[2 * i + 1]; }
}

// A[(b,t,f)][c*9 + (dt+1)*3 + (df+1)] = X[b, t+dt, f+df, c] (0 outside the map); columns >= 9C (padding up to lda) = 0.
// one thread = one (row, c): 9 taps
template <typename T>
__global__ void im2col3x3_kernel(const T* __restrict__ X, T* __restrict__ A, int Tn, int Fn, int C, int lda, long total) {
    const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= total) return;
    const int cp = lda / 9 + 1;                               // channel slots per row incl. one slot for the pad columns
    const int c = (int)(id % cp);
    const long row = id / cp;
    T* dst = A + row * lda;
    if (c * 9 >= lda) return;
    if (c >= C) {                                             // pad columns
        for (int k = C * 9 + (c - C) * 9; k < min(lda, C * 9 + (c - C + 1) * 9); ++k) dst[k] = from_f32<T>(0.f);
        return;
    }
    const int f = (int)(row % Fn);
    const long bt = row / Fn;
    const int t = (int)(bt % Tn);
#pragma unroll
    for (int dt = -1; dt <= 1; ++dt)
#pragma unroll
        for (int df = -1; df <= 1; ++df) {
            const bool in = t + dt >= 0 && t + dt < Tn && f + df >= 0 && f + df < Fn;
            dst[c * 9 + (dt + 1) * 3 + (df + 1)] = in ? X[(row + (long)dt * Fn + df) * C + c] : from_f32<T>(0.f);
        }
}
// dX[b,t,f,c] = sum over taps of dA[(b, t-dt, f-df)][c*9 + tap]
template <typename T>
__global__ void col2im3x3_kernel(const T* __restrict__ dA, T* __restrict__ dX, int Tn, int Fn, int C, int lda, long total) {
    const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= total) return;
    const int c = (int)(id % C);
    const long row = id / C;
    const int f = (int)(row % Fn);
    const long bt = row / Fn;
    const int t = (int)(bt % Tn);
    float s = 0.f;
#pragma unroll
    for (int dt = -1; dt <= 1; ++dt)
#pragma unroll
        for (int df = -1; df <= 1; ++df) {
            const int ts = t - dt, fs = f - df;
            if (ts >= 0 && ts < Tn && fs >= 0 && fs < Fn)
                s += to_f32<T>(dA[(row - (long)dt * Fn - df) * lda + c * 9 + (dt + 1) * 3 + (df + 1)]);
        }
    dX[id] = from_f32<T>(s);
}

// per-column sums over a block of rows: MODE 0: (sum x, sum x^2); MODE 1: with g = dy * (y > 0): (sum g*xhat, sum g).
// part[block][C][2]; thread = column (C <= blockDim.x * gridDim.y)
template <typename T, int MODE>
__global__ __launch_bounds__(256) void bn2d_sums_kernel(const T* __restrict__ X, const T* __restrict__ Y, const T* __restrict__ dY,
                                                        const float* __restrict__ mean_rstd, float* __restrict__ part, long rows,
                                                        int C, int rows_per_block) {
    const int c = blockIdx.y * 256 + threadIdx.x;
    if (c >= C) return;
    const long beg = (long)blockIdx.x * rows_per_block, end = min(rows, beg + rows_per_block);
    float a = 0.f, b = 0.f;
    float mean = 0.f, rstd = 1.f;
    if (MODE == 1) { mean = mean_rstd[2 * c]; rstd = mean_rstd[2 * c + 1]; }
    for (long r = beg; r < end; ++r) {
        const float x = to_f32<T>(X[r * C + c]);
        if (MODE == 0) { a += x; b += x * x; }
        else {
            const float g = to_f32<T>(Y[r * C + c]) > 0.f ? to_f32<T>(dY[r * C + c]) : 0.f;
            a += g * (x - mean) * rstd; b += g;
        }
    }
    float* o = part + ((long)blockIdx.x * C + c) * 2;
    o[0] = a; o[1] = b;
}
template <typename T>
__global__ void bn_relu_fwd_kernel(const T* __restrict__ X, const float* __restrict__ ss, T* __restrict__ Y, int C, long total) {
    const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= total) return;
    const int c = (int)(id % C);
    Y[id] = from_f32<T>(fmaxf(to_f32<T>(X[id]) * ss[2 * c] + ss[2 * c + 1], 0.f));
}
// dx = gamma * rstd * (g - mean(g) - xhat * mean(g * xhat)), g = dy * (y > 0); sums = [C][2] = (sum g*xhat, sum g)
template <typename T>
__global__ void bn_relu_bwd_kernel(const T* __restrict__ X, const T* __restrict__ Y, const T* __restrict__ dY,
                                   const float* __restrict__ mean_rstd, const float* __restrict__ gamma,
                                   const float* __restrict__ sums, float inv_n, T* __restrict__ dX, int C, long total) {
    const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= total) return;
    const int c = (int)(id % C);
    const float mean = mean_rstd[2 * c], rstd = mean_rstd[2 * c + 1];
    const float g = to_f32<T>(Y[id]) > 0.f ? to_f32<T>(dY[id]) : 0.f;
    const float xh = (to_f32<T>(X[id]) - mean) * rstd;
    dX[id] = from_f32<T>(gamma[c] * rstd * (g - sums[2 * c + 1] * inv_n - xh * sums[2 * c] * inv_n));
}

// AvgPool2d((pt, pf)) on [B, T, F, C] rows (floor mode, stride = kernel)
template <typename T>
__global__ void avgpool_fwd_kernel(const T* __restrict__ X, T* __restrict__ Y, int Tn, int Fn, int C, int pt, int pf, long total) {
    const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= total) return;
    const int To = Tn / pt, Fo = Fn / pf;
    const int c = (int)(id % C);
    long rest = id / C;
    const int fo = (int)(rest % Fo); rest /= Fo;
    const int to = (int)(rest % To);
    const long b = rest / To;
    float s = 0.f;
    for (int i = 0; i < pt; ++i)
        for (int j = 0; j < pf; ++j) s += to_f32<T>(X[((b * Tn + to * pt + i) * Fn + fo * pf + j) * C + c]);
    Y[id] = from_f32<T>(s / (pt * pf));
}
template <typename T>
__global__ void avgpool_bwd_kernel(const T* __restrict__ dY, T* __restrict__ dX, int Tn, int Fn, int C, int pt, int pf, long total) {
    const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= total) return;
    const int To = Tn / pt, Fo = Fn / pf;
    const int c = (int)(id % C);
    long rest = id / C;
    const int f = (int)(rest % Fn); rest /= Fn;
    const int t = (int)(rest % Tn);
    const long b = rest / Tn;
    const int to = t / pt, fo = f / pf;
    float v = 0.f;
    if (to < To && fo < Fo) v = to_f32<T>(dY[((b * To + to) * Fo + fo) * C + c]) / (pt * pf);
    dX[id] = from_f32<T>(v);
}

// y[b, j, :] = sum_{k<3} w[j][k] * x[b, i0[j] + k, :]  (the 'repeat' x ratio + group-mean map of accdoa.py:86-87 in compact form)
template <typename T>
__global__ void rows_pool_fwd_kernel(const T* __restrict__ X, const int* __restrict__ i0, const float* __restrict__ w, T* __restrict__ Y,
                                     int n_in, int n_out, int C, long total) {
    const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= total) return;
    const int c = (int)(id % C);
    long rest = id / C;
    const int j = (int)(rest % n_out);
    const long b = rest / n_out;
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int i = min(i0[j] + k, n_in - 1);
        s += w[j * 3 + k] * to_f32<T>(X[(b * n_in + i) * C + c]);
    }
    Y[id] = from_f32<T>(s);
}
template <typename T>
__global__ void rows_pool_bwd_kernel(const T* __restrict__ dY, const int* __restrict__ i0, const float* __restrict__ w, T* __restrict__ dX,
                                     int n_in, int n_out, int C, long total) {
    const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= total) return;
    const int c = (int)(id % C);
    long rest = id / C;
    const int i = (int)(rest % n_in);
    const long b = rest / n_in;
    float s = 0.f;
    for (int j = 0; j < n_out; ++j) {                        // n_out = 100: the map is tiny
        const int k = i - i0[j];
        if (k >= 0 && k < 3 && w[j * 3 + k] != 0.f) s += w[j * 3 + k] * to_f32<T>(dY[(b * n_out + j) * C + c]);
    }
    dX[id] = from_f32<T>(s);
}

// dst[r, 0:cols] = src[r, 0:cols], dst[r, cols:ldd] = 0 (weight matrices whose row length is not a multiple of 8)
template <typename T>
__global__ void copy2d_kernel(const T* __restrict__ src, int lds_, T* __restrict__ dst, int ldd, int cols, long total) {
    const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= total) return;
    const int c = (int)(id % ldd);
    const long r = id / ldd;
    dst[id] = c < cols ? src[r * lds_ + c] : from_f32<T>(0.f);
}

constexpr int CNN_ROWS_PER_BLOCK = 1024;

}  // namespace

#define CNN_DISPATCH(name, CALL)                                                  \
    if (dtype == PSELD_BF16) { using T = bf16_t; CALL; }                          \
    else if (dtype == PSELD_F32) { using T = float; CALL; }                       \
    else { pseld_set_error(name ": unknown dtype"); return PSELD_ERR_BAD_ARG; }   \
    PSELD_LAUNCH_CHECK(name);                                                     \
    return PSELD_OK

extern "C" int pseld_cnn_input(int dtype, const float* feat, const float* scale_shift, void* X, int B, int Cin, int Tn, int Cp,
                               void* stream) {
    PSELD_CHECK_ARG(feat && scale_shift && X && B > 0 && Cin > 0 && Cp >= Cin && Tn > 0, "cnn_input: bad argument");
    hipStream_t s = (hipStream_t)stream;
    const long total = (long)B * Tn * 64 * Cp;
    CNN_DISPATCH("cnn_input", hipLaunchKernelGGL(cnn_input_kernel<T>, dim3(pseld_cdiv(total, 256)), dim3(256), 0, s, feat, scale_shift,
                                                 (T*)X, Cin, Tn, Cp, total));
}
extern "C" long pseld_cnn_input_bwd_workspace(int B, int Cin, int Tn) {
    return ((long)pseld_cdiv((long)B * Tn, CNN_ROWS_PER_BLOCK) + 1) * Cin * 64 * 2 * (long)sizeof(float);
}
extern "C" int pseld_cnn_input_bwd(int dtype, const float* feat, const float* mean_rstd, const void* dX, float* dweight,
                                   float* dbias, int B, int Cin, int Tn, int Cp, float* workspace, long workspace_bytes,
                                   void* stream) {
    PSELD_CHECK_ARG(feat && mean_rstd && dX && dweight && dbias && workspace, "cnn_input_bwd: null pointer");
    PSELD_CHECK_ARG(workspace_bytes >= pseld_cnn_input_bwd_workspace(B, Cin, Tn), "cnn_input_bwd: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    const int nb = pseld_cdiv((long)B * Tn, CNN_ROWS_PER_BLOCK), n = Cin * 64;
    if (dtype == PSELD_BF16)
        hipLaunchKernelGGL(cnn_input_bwd_kernel<bf16_t>, dim3(nb, Cin), dim3(256), 0, s, feat, mean_rstd, (const bf16_t*)dX, workspace, B, Cin, Tn, Cp, CNN_ROWS_PER_BLOCK);
    else if (dtype == PSELD_F32)
        hipLaunchKernelGGL(cnn_input_bwd_kernel<float>, dim3(nb, Cin), dim3(256), 0, s, feat, mean_rstd, (const float*)dX, workspace, B, Cin, Tn, Cp, CNN_ROWS_PER_BLOCK);
    else { pseld_set_error("cnn_input_bwd: unknown dtype"); return PSELD_ERR_BAD_ARG; }
    float* total = workspace + (long)nb * n * 2;
    pseld_reduce_slabs(workspace, total, (long)n * 2, nb, (long)n * 2, 0, s);
    hipLaunchKernelGGL(deinterleave2_kernel, dim3(pseld_cdiv(n, 256)), dim3(256), 0, s, total, n, dweight, dbias);
    PSELD_LAUNCH_CHECK("cnn_input_bwd");
    return PSELD_OK;
}

extern "C" int pseld_im2col3x3(int dtype, const void* X, void* A, int B, int Tn, int Fn, int C, int lda, void* stream) {
    PSELD_CHECK_ARG(X && A && B > 0 && Tn > 0 && Fn > 0 && C > 0 && lda >= 9 * C, "im2col3x3: bad argument");
    hipStream_t s = (hipStream_t)stream;
    const long total = (long)B * Tn * Fn * (lda / 9 + 1);
    CNN_DISPATCH("im2col3x3", hipLaunchKernelGGL(im2col3x3_kernel<T>, dim3(pseld_cdiv(total, 256)), dim3(256), 0, s, (const T*)X, (T*)A,
                                                 Tn, Fn, C, lda, total));
}
extern "C" int pseld_col2im3x3(int dtype, const void* dA, void* dX, int B, int Tn, int Fn, int C, int lda, void* stream) {
    PSELD_CHECK_ARG(dA && dX && B > 0 && Tn > 0 && Fn > 0 && C > 0 && lda >= 9 * C, "col2im3x3: bad argument");
    hipStream_t s = (hipStream_t)stream;
    const long total = (long)B * Tn * Fn * C;
    CNN_DISPATCH("col2im3x3", hipLaunchKernelGGL(col2im3x3_kernel<T>, dim3(pseld_cdiv(total, 256)), dim3(256), 0, s, (const T*)dA, (T*)dX,
                                                 Tn, Fn, C, lda, total));
}

/* BatchNorm2d statistics over the rows of an NHWC map: sums f32[C][2] = (sum x, sum x^2) */
extern "C" long pseld_bn2d_workspace(long rows, int C) {
    return ((long)pseld_cdiv(rows, CNN_ROWS_PER_BLOCK) + 1) * C * 2 * (long)sizeof(float);
}
extern "C" int pseld_bn2d_stats(int dtype, const void* X, float* sums, long rows, int C, float* workspace, long workspace_bytes,
                                void* stream) {
    PSELD_CHECK_ARG(X && sums && workspace && rows > 0 && C > 0, "bn2d_stats: bad argument");
    PSELD_CHECK_ARG(workspace_bytes >= pseld_bn2d_workspace(rows, C), "bn2d_stats: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    const int nb = pseld_cdiv(rows, CNN_ROWS_PER_BLOCK);
    const dim3 grid(nb, pseld_cdiv(C, 256));
    if (dtype == PSELD_BF16)
        hipLaunchKernelGGL((bn2d_sums_kernel<bf16_t, 0>), grid, dim3(256), 0, s, (const bf16_t*)X, (const bf16_t*)nullptr, (const bf16_t*)nullptr, (const float*)nullptr, workspace, rows, C, CNN_ROWS_PER_BLOCK);
    else if (dtype == PSELD_F32)
        hipLaunchKernelGGL((bn2d_sums_kernel<float, 0>), grid, dim3(256), 0, s, (const float*)X, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, workspace, rows, C, CNN_ROWS_PER_BLOCK);
    else { pseld_set_error("bn2d_stats: unknown dtype"); return PSELD_ERR_BAD_ARG; }
    pseld_reduce_slabs(workspace, sums, (long)C * 2, nb, (long)C * 2, 0, s);
    PSELD_LAUNCH_CHECK("bn2d_stats");
    return PSELD_OK;
}
extern "C" int pseld_bn_relu_fwd(int dtype, const void* X, const float* scale_shift, void* Y, long rows, int C, void* stream) {
    PSELD_CHECK_ARG(X && scale_shift && Y && rows > 0 && C > 0, "bn_relu_fwd: bad argument");
    hipStream_t s = (hipStream_t)stream;
    const long total = rows * C;
    CNN_DISPATCH("bn_relu_fwd", hipLaunchKernelGGL(bn_relu_fwd_kernel<T>, dim3(pseld_cdiv(total, 256)), dim3(256), 0, s, (const T*)X,
                                                   scale_shift, (T*)Y, C, total));
}
/* backward of y = relu(bn(x)): dX, dgamma (+)=, dbeta (+)= (train-mode batch statistics) */
extern "C" int pseld_bn_relu_bwd(int dtype, const void* X, const void* Y, const void* dY, const float* mean_rstd,
                                 const float* gamma, void* dX, float* dgamma, float* dbeta, long rows, int C, float* workspace,
                                 long workspace_bytes, void* stream) {
    PSELD_CHECK_ARG(X && Y && dY && mean_rstd && gamma && dX && dgamma && dbeta && workspace, "bn_relu_bwd: null pointer");
    PSELD_CHECK_ARG(workspace_bytes >= pseld_bn2d_workspace(rows, C), "bn_relu_bwd: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    const int nb = pseld_cdiv(rows, CNN_ROWS_PER_BLOCK);
    const dim3 grid(nb, pseld_cdiv(C, 256));
    float* total = workspace + (long)nb * C * 2;
    const long n = rows * C;
    if (dtype == PSELD_BF16) {
        hipLaunchKernelGGL((bn2d_sums_kernel<bf16_t, 1>), grid, dim3(256), 0, s, (const bf16_t*)X, (const bf16_t*)Y, (const bf16_t*)dY, mean_rstd, workspace, rows, C, CNN_ROWS_PER_BLOCK);
        pseld_reduce_slabs(workspace, total, (long)C * 2, nb, (long)C * 2, 0, s);
        hipLaunchKernelGGL(bn_relu_bwd_kernel<bf16_t>, dim3(pseld_cdiv(n, 256)), dim3(256), 0, s, (const bf16_t*)X, (const bf16_t*)Y, (const bf16_t*)dY, mean_rstd, gamma, total, 1.f / (float)rows, (bf16_t*)dX, C, n);
    } else if (dtype == PSELD_F32) {
        hipLaunchKernelGGL((bn2d_sums_kernel<float, 1>), grid, dim3(256), 0, s, (const float*)X, (const float*)Y, (const float*)dY, mean_rstd, workspace, rows, C, CNN_ROWS_PER_BLOCK);
        pseld_reduce_slabs(workspace, total, (long)C * 2, nb, (long)C * 2, 0, s);
        hipLaunchKernelGGL(bn_relu_bwd_kernel<float>, dim3(pseld_cdiv(n, 256)), dim3(256), 0, s, (const float*)X, (const float*)Y, (const float*)dY, mean_rstd, gamma, total, 1.f / (float)rows, (float*)dX, C, n);
    } else { pseld_set_error("bn_relu_bwd: unknown dtype"); return PSELD_ERR_BAD_ARG; }
    hipLaunchKernelGGL(deinterleave2_kernel, dim3(pseld_cdiv(C, 256)), dim3(256), 0, s, total, C, dgamma, dbeta);
    PSELD_LAUNCH_CHECK("bn_relu_bwd");
    return PSELD_OK;
}

extern "C" int pseld_avgpool_fwd(int dtype, const void* X, void* Y, int B, int Tn, int Fn, int C, int pt, int pf, void* stream) {
    PSELD_CHECK_ARG(X && Y && B > 0 && pt > 0 && pf > 0 && Tn >= pt && Fn >= pf && C > 0, "avgpool_fwd: bad argument");
    hipStream_t s = (hipStream_t)stream;
    const long total = (long)B * (Tn / pt) * (Fn / pf) * C;
    CNN_DISPATCH("avgpool_fwd", hipLaunchKernelGGL(avgpool_fwd_kernel<T>, dim3(pseld_cdiv(total, 256)), dim3(256), 0, s, (const T*)X, (T*)Y,
                                                   Tn, Fn, C, pt, pf, total));
}
extern "C" int pseld_avgpool_bwd(int dtype, const void* dY, void* dX, int B, int Tn, int Fn, int C, int pt, int pf, void* stream) {
    PSELD_CHECK_ARG(dY && dX && B > 0 && pt > 0 && pf > 0 && Tn >= pt && Fn >= pf && C > 0, "avgpool_bwd: bad argument");
    hipStream_t s = (hipStream_t)stream;
    const long total = (long)B * Tn * Fn * C;
    CNN_DISPATCH("avgpool_bwd", hipLaunchKernelGGL(avgpool_bwd_kernel<T>, dim3(pseld_cdiv(total, 256)), dim3(256), 0, s, (const T*)dY, (T*)dX,
                                                   Tn, Fn, C, pt, pf, total));
}
extern "C" int pseld_rows_pool_fwd(int dtype, const void* X, const int* i0, const float* w, void* Y, int B, int n_in, int n_out,
                                   int C, void* stream) {
    PSELD_CHECK_ARG(X && i0 && w && Y && B > 0 && n_in > 0 && n_out > 0 && C > 0, "rows_pool_fwd: bad argument");
    hipStream_t s = (hipStream_t)stream;
    const long total = (long)B * n_out * C;
    CNN_DISPATCH("rows_pool_fwd", hipLaunchKernelGGL(rows_pool_fwd_kernel<T>, dim3(pseld_cdiv(total, 256)), dim3(256), 0, s, (const T*)X, i0,
                                                     w, (T*)Y, n_in, n_out, C, total));
}
extern "C" int pseld_rows_pool_bwd(int dtype, const void* dY, const int* i0, const float* w, void* dX, int B, int n_in, int n_out,
                                   int C, void* stream) {
    PSELD_CHECK_ARG(dY && i0 && w && dX && B > 0 && n_in > 0 && n_out > 0 && C > 0, "rows_pool_bwd: bad argument");
    hipStream_t s = (hipStream_t)stream;
    const long total = (long)B * n_in * C;
    CNN_DISPATCH("rows_pool_bwd", hipLaunchKernelGGL(rows_pool_bwd_kernel<T>, dim3(pseld_cdiv(total, 256)), dim3(256), 0, s, (const T*)dY, i0,
                                                     w, (T*)dX, n_in, n_out, C, total));
}
extern "C" int pseld_copy2d(int dtype, const void* src, int ld_src, void* dst, int ld_dst, long rows, int cols, void* stream) {
    PSELD_CHECK_ARG(src && dst && rows > 0 && cols > 0 && ld_src >= cols && ld_dst >= cols, "copy2d: bad argument");
    hipStream_t s = (hipStream_t)stream;
    const long total = rows * ld_dst;
    CNN_DISPATCH("copy2d", hipLaunchKernelGGL(copy2d_kernel<T>, dim3(pseld_cdiv(total, 256)), dim3(256), 0, s, (const T*)src, ld_src, (T*)dst,
                                              ld_dst, cols, total));
}
