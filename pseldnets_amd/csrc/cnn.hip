// Convolutional encoder of the CRNN networks (CNN8 / CNN12 = the PANNs CNN14 conv stack) on gfx950: everything around
// the MFMA GEMMs of the 3x3 convolutions. Activations live as NHWC rows [B*T*F, C] so that every convolution is
// im2col -> gemm.hip -> (col2im for the input gradient) and every BatchNorm2d is a per-column operation.
//
// Replaces (reference, /root/reference/src): models/accdoa.py:72-90 (scalar BN, conv stack, frequency mean, 'repeat'
// interpolation + 10-frame mean), models/components/backbone.py:6-60 (CNN8, CNN12), models/components/
// model_utilities.py:92-126 (ConvBlock: conv3x3 -> BatchNorm2d -> ReLU, twice, AvgPool2d), models/components/
// utils.py:25-52 (interpolate, method 'repeat') — and the autograd of each. All HBM-bound address maps / reductions;
// the im2col matrix is materialised per batch slice in tap-major column order (k = tap*C + c: 16-byte copies); the
// reference's [Cout, Cin, 3, 3] weights are permuted to that order once per step and their gradient permuted back.
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

// A[(b,t,f)][tap*C + c] = X[b, t+dt, f+df, c] (0 outside the map), tap = (dt+1)*3 + (df+1): TAP-MAJOR columns, so
// that every tap is one contiguous C-vector of the neighbouring pixel and the copy runs in 16-byte chunks (C % 8 == 0).
// The weights are permuted to the same order once per step (conv_weight_kernel). One thread = one 8-channel chunk.
template <typename T>
__global__ void im2col3x3_kernel(const T* __restrict__ X, T* __restrict__ A, int Tn, int Fn, int C, long total) {
    const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= total) return;
    const int c8 = C >> 3;
    const int ch = (int)(id % c8);
    long rest = id / c8;
    const int tap = (int)(rest % 9);
    const long row = rest / 9;
    const int dt = tap / 3 - 1, df = tap % 3 - 1;
    const int f = (int)(row % Fn);
    const int t = (int)((row / Fn) % Tn);
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (t + dt >= 0 && t + dt < Tn && f + df >= 0 && f + df < Fn) load8<T>(X + (row + (long)dt * Fn + df) * C + ch * 8, v);
    store8<T>(A + row * (9L * C) + (long)tap * C + ch * 8, v);
}
// dX[b,t,f,c] = sum over taps of dA[(b, t-dt, f-df)][tap*C + c]; one thread = one 8-channel chunk
template <typename T>
__global__ void col2im3x3_kernel(const T* __restrict__ dA, T* __restrict__ dX, int Tn, int Fn, int C, long total) {
    const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= total) return;
    const int c8 = C >> 3;
    const int ch = (int)(id % c8);
    const long row = id / c8;
    const int f = (int)(row % Fn);
    const int t = (int)((row / Fn) % Tn);
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        const int dt = tap / 3 - 1, df = tap % 3 - 1;
        const int ts = t - dt, fs = f - df;
        if (ts >= 0 && ts < Tn && fs >= 0 && fs < Fn) {
            float v[8];
            load8<T>(dA + (row - (long)dt * Fn - df) * (9L * C) + (long)tap * C + ch * 8, v);
#pragma unroll
            for (int k = 0; k < 8; ++k) s[k] += v[k];
        }
    }
    store8<T>(dX + row * C + ch * 8, s);
}
// conv weight [Cout][Cin][9] (the reference's [Cout, Cin, 3, 3]) -> tap-major [Cout][9][Cp] with zero channels
// Cin..Cp-1 (TO_TAP), or the gradient back: dW[co][ci][tap] = dWp[co][tap][ci] (!TO_TAP)
template <typename TS, typename TD, bool TO_TAP>
__global__ void conv_weight_kernel(const TS* __restrict__ src, TD* __restrict__ dst, int Cin, int Cp, long total) {
    const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= total) return;
    if (TO_TAP) {
        const int ci = (int)(id % Cp);
        const int tap = (int)((id / Cp) % 9);
        const long co = id / (9L * Cp);
        dst[id] = ci < Cin ? (TD)src[(co * Cin + ci) * 9 + tap] : (TD)0.f;
    } else {
        const int tap = (int)(id % 9);
        const int ci = (int)((id / 9) % Cin);
        const long co = id / (9L * Cin);
        dst[id] = (TD)src[(co * 9 + tap) * Cp + ci];
    }
}

// per-column sums over a block of rows: MODE 0: (sum x, sum x^2); MODE 1: with g = dy * (y > 0): (sum g*xhat, sum g).
// part[block][C][2]. A workgroup covers CW = min(C, 256) columns with 256 / CW row lanes striding its rows (coalesced
// row segments, every thread busy also for narrow maps); the row lanes are combined through LDS.
template <typename T, int MODE>
__global__ __launch_bounds__(256) void bn2d_sums_kernel(const T* __restrict__ X, const T* __restrict__ Y, const T* __restrict__ dY,
                                                        const float* __restrict__ mean_rstd, float* __restrict__ part, long rows,
                                                        int C, int rows_per_block) {
    // a thread owns 8 consecutive columns (16-byte loads); a workgroup covers CW8 = min(C/8, 256) column octets with
    // 256 / CW8 row lanes striding its rows; the row lanes are combined through LDS
    __shared__ float red[256][16];
    const int c8 = C >> 3;
    const int cw = c8 < 256 ? c8 : 256;
    const int lanes = 256 / cw;
    const int cl = threadIdx.x % cw, rl = threadIdx.x / cw;
    const int co = blockIdx.y * cw + cl;                   // column octet
    const long beg = (long)blockIdx.x * rows_per_block, end = min(rows, beg + rows_per_block);
    float a[8], b[8], mean[8], rstd[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { a[k] = 0.f; b[k] = 0.f; mean[k] = 0.f; rstd[k] = 1.f; }
    if (co < c8 && rl < lanes) {
        if (MODE == 1) {
#pragma unroll
            for (int k = 0; k < 8; ++k) { mean[k] = mean_rstd[2 * (co * 8 + k)]; rstd[k] = mean_rstd[2 * (co * 8 + k) + 1]; }
        }
        for (long r = beg + rl; r < end; r += lanes) {
            float x[8];
            load8<T>(X + r * C + co * 8, x);
            if (MODE == 0) {
#pragma unroll
                for (int k = 0; k < 8; ++k) { a[k] += x[k]; b[k] += x[k] * x[k]; }
            } else {
                float y[8], dy[8];
                if (Y) load8<T>(Y + r * C + co * 8, y);
                load8<T>(dY + r * C + co * 8, dy);
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const float g = (!Y || y[k] > 0.f) ? dy[k] : 0.f;
                    a[k] += g * (x[k] - mean[k]) * rstd[k]; b[k] += g;
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) { red[threadIdx.x][k] = a[k]; red[threadIdx.x][8 + k] = b[k]; }
    __syncthreads();
    if (rl == 0 && co < c8) {
        for (int i = 1; i < lanes; ++i)
#pragma unroll
            for (int k = 0; k < 8; ++k) { a[k] += red[i * cw + cl][k]; b[k] += red[i * cw + cl][8 + k]; }
        float* o = part + ((long)blockIdx.x * C + co * 8) * 2;
#pragma unroll
        for (int k = 0; k < 8; ++k) { o[2 * k] = a[k]; o[2 * k + 1] = b[k]; }
    }
}
template <typename T>
__global__ void bn_relu_fwd_kernel(const T* __restrict__ X, const float* __restrict__ ss, T* __restrict__ Y, int C, long total8,
                                   float floor_) {
    const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= total8) return;
    const int c0 = (int)((id * 8) % C);
    float x[8];
    load8<T>(X + id * 8, x);
#pragma unroll
    for (int k = 0; k < 8; ++k) x[k] = fmaxf(x[k] * ss[2 * (c0 + k)] + ss[2 * (c0 + k) + 1], floor_);   // floor -inf: plain affine
    store8<T>(Y + id * 8, x);
}
// dx = gamma * rstd * (g - mean(g) - xhat * mean(g * xhat)), g = dy * (y > 0); sums = [C][2] = (sum g*xhat, sum g)
template <typename T>
__global__ void bn_relu_bwd_kernel(const T* __restrict__ X, const T* __restrict__ Y, const T* __restrict__ dY,
                                   const float* __restrict__ mean_rstd, const float* __restrict__ gamma,
                                   const float* __restrict__ sums, float inv_n, T* __restrict__ dX, int C, long total8) {
    const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= total8) return;
    const int c0 = (int)((id * 8) % C);
    float x[8], y[8], dy[8], o[8];
    load8<T>(X + id * 8, x);
    if (Y) load8<T>(Y + id * 8, y);
    load8<T>(dY + id * 8, dy);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int c = c0 + k;
        const float mean = mean_rstd[2 * c], rstd = mean_rstd[2 * c + 1];
        const float g = (!Y || y[k] > 0.f) ? dy[k] : 0.f;
        const float xh = (x[k] - mean) * rstd;
        o[k] = gamma[c] * rstd * (g - sums[2 * c + 1] * inv_n - xh * sums[2 * c] * inv_n);
    }
    store8<T>(dX + id * 8, o);
}

// AvgPool2d((pt, pf)) on [B, T, F, C] rows (floor mode, stride = kernel); a thread moves 8 channels
template <typename T>
__global__ void avgpool_fwd_kernel(const T* __restrict__ X, T* __restrict__ Y, int Tn, int Fn, int C, int pt, int pf, long total8) {
    const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= total8) return;
    const int To = Tn / pt, Fo = Fn / pf, c8 = C >> 3;
    const int c = (int)(id % c8) * 8;
    long rest = id / c8;
    const int fo = (int)(rest % Fo); rest /= Fo;
    const int to = (int)(rest % To);
    const long b = rest / To;
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < pt; ++i)
        for (int j = 0; j < pf; ++j) {
            float v[8];
            load8<T>(X + ((b * Tn + to * pt + i) * Fn + fo * pf + j) * C + c, v);
#pragma unroll
            for (int k = 0; k < 8; ++k) s[k] += v[k];
        }
    const float inv = 1.f / (pt * pf);
#pragma unroll
    for (int k = 0; k < 8; ++k) s[k] *= inv;
    store8<T>(Y + id * 8, s);
}
template <typename T>
__global__ void avgpool_bwd_kernel(const T* __restrict__ dY, T* __restrict__ dX, int Tn, int Fn, int C, int pt, int pf, long total8) {
    const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= total8) return;
    const int To = Tn / pt, Fo = Fn / pf, c8 = C >> 3;
    const int c = (int)(id % c8) * 8;
    long rest = id / c8;
    const int f = (int)(rest % Fn); rest /= Fn;
    const int t = (int)(rest % Tn);
    const long b = rest / Tn;
    const int to = t / pt, fo = f / pf;
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (to < To && fo < Fo) {
        load8<T>(dY + ((b * To + to) * Fo + fo) * C + c, v);
        const float inv = 1.f / (pt * pf);
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] *= inv;
    }
    store8<T>(dX + id * 8, v);
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

constexpr int CNN_ROWS_PER_BLOCK = 1024;
// BatchNorm2d column sums: rows per workgroup such that even the deep, short layers (12 000 rows x 2048 channels) put
// about two workgroups on every CU, without more than 1024 rows per workgroup on the long ones
static inline int bn2d_rows_per_block(long rows, int C) {
    const int c8 = C / 8, colgroups = pseld_cdiv(c8, c8 < 256 ? c8 : 256);
    const long target = 512 / colgroups > 1 ? 512 / colgroups : 1;
    long r = (rows + target - 1) / target;
    if (r < 16) r = 16;
    if (r > CNN_ROWS_PER_BLOCK) r = CNN_ROWS_PER_BLOCK;
    return (int)r;
}

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

extern "C" int pseld_im2col3x3(int dtype, const void* X, void* A, int B, int Tn, int Fn, int C, void* stream) {
    PSELD_CHECK_ARG(X && A && B > 0 && Tn > 0 && Fn > 0 && C > 0 && C % 8 == 0, "im2col3x3: bad argument (C must be a multiple of 8)");
    hipStream_t s = (hipStream_t)stream;
    const long total = (long)B * Tn * Fn * 9 * (C / 8);
    CNN_DISPATCH("im2col3x3", hipLaunchKernelGGL(im2col3x3_kernel<T>, dim3(pseld_cdiv(total, 256)), dim3(256), 0, s, (const T*)X, (T*)A,
                                                 Tn, Fn, C, total));
}
extern "C" int pseld_col2im3x3(int dtype, const void* dA, void* dX, int B, int Tn, int Fn, int C, void* stream) {
    PSELD_CHECK_ARG(dA && dX && B > 0 && Tn > 0 && Fn > 0 && C > 0 && C % 8 == 0, "col2im3x3: bad argument (C must be a multiple of 8)");
    hipStream_t s = (hipStream_t)stream;
    const long total = (long)B * Tn * Fn * (C / 8);
    CNN_DISPATCH("col2im3x3", hipLaunchKernelGGL(col2im3x3_kernel<T>, dim3(pseld_cdiv(total, 256)), dim3(256), 0, s, (const T*)dA, (T*)dX,
                                                 Tn, Fn, C, total));
}
/* weights in the compute dtype, reference layout [Cout, Cin, 3, 3] -> tap-major [Cout, 9, Cp]; fp32 gradient back */
extern "C" int pseld_conv_weight_to_tap(int dtype, const void* W, void* Wp, int Cout, int Cin, int Cp, void* stream) {
    PSELD_CHECK_ARG(W && Wp && Cout > 0 && Cin > 0 && Cp >= Cin && Cp % 8 == 0, "conv_weight_to_tap: bad argument");
    hipStream_t s = (hipStream_t)stream;
    const long total = (long)Cout * 9 * Cp;
    if (dtype == PSELD_BF16) hipLaunchKernelGGL((conv_weight_kernel<bf16_t, bf16_t, true>), dim3(pseld_cdiv(total, 256)), dim3(256), 0, s, (const bf16_t*)W, (bf16_t*)Wp, Cin, Cp, total);
    else if (dtype == PSELD_F32) hipLaunchKernelGGL((conv_weight_kernel<float, float, true>), dim3(pseld_cdiv(total, 256)), dim3(256), 0, s, (const float*)W, (float*)Wp, Cin, Cp, total);
    else { pseld_set_error("conv_weight_to_tap: unknown dtype"); return PSELD_ERR_BAD_ARG; }
    PSELD_LAUNCH_CHECK("conv_weight_to_tap");
    return PSELD_OK;
}
// input-gradient weights of the implicit convolution: Wd[ci][8 - tap][co] = W[co][ci][tap], rows ci >= Cin zero ([Cp, 9*Cout])
template <typename T>
__global__ void conv_weight_t_kernel(const T* __restrict__ W, T* __restrict__ Wd, int Cout, int Cin, long total) {
    const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= total) return;
    const int co = (int)(id % Cout);
    const int tapd = (int)((id / Cout) % 9);
    const long ci = id / (9L * Cout);
    Wd[id] = ci < Cin ? W[((long)co * Cin + ci) * 9 + (8 - tapd)] : (T)0.f;
}
extern "C" int pseld_conv_weight_to_tap_t(int dtype, const void* W, void* Wd, int Cout, int Cin, int Cp, void* stream) {
    PSELD_CHECK_ARG(W && Wd && Cout > 0 && Cin > 0 && Cp >= Cin && Cp % 8 == 0 && Cout % 8 == 0, "conv_weight_to_tap_t: bad argument");
    hipStream_t s = (hipStream_t)stream;
    const long total = (long)Cp * 9 * Cout;
    if (dtype == PSELD_BF16) hipLaunchKernelGGL(conv_weight_t_kernel<bf16_t>, dim3(pseld_cdiv(total, 256)), dim3(256), 0, s, (const bf16_t*)W, (bf16_t*)Wd, Cout, Cin, total);
    else if (dtype == PSELD_F32) hipLaunchKernelGGL(conv_weight_t_kernel<float>, dim3(pseld_cdiv(total, 256)), dim3(256), 0, s, (const float*)W, (float*)Wd, Cout, Cin, total);
    else { pseld_set_error("conv_weight_to_tap_t: unknown dtype"); return PSELD_ERR_BAD_ARG; }
    PSELD_LAUNCH_CHECK("conv_weight_to_tap_t");
    return PSELD_OK;
}
extern "C" int pseld_conv_wgrad_from_tap(const float* dWp, float* dW, int Cout, int Cin, int Cp, void* stream) {
    PSELD_CHECK_ARG(dWp && dW && Cout > 0 && Cin > 0 && Cp >= Cin, "conv_wgrad_from_tap: bad argument");
    const long total = (long)Cout * Cin * 9;
    hipLaunchKernelGGL((conv_weight_kernel<float, float, false>), dim3(pseld_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, dWp, dW, Cin, Cp, total);
    PSELD_LAUNCH_CHECK("conv_wgrad_from_tap");
    return PSELD_OK;
}

/* BatchNorm2d statistics over the rows of an NHWC map: sums f32[C][2] = (sum x, sum x^2) */
extern "C" long pseld_bn2d_workspace(long rows, int C) {
    return ((long)pseld_cdiv(rows, bn2d_rows_per_block(rows, C)) + 1) * C * 2 * (long)sizeof(float);
}
extern "C" int pseld_bn2d_stats(int dtype, const void* X, float* sums, long rows, int C, float* workspace, long workspace_bytes,
                                void* stream) {
    PSELD_CHECK_ARG(X && sums && workspace && rows > 0 && C > 0, "bn2d_stats: bad argument");
    PSELD_CHECK_ARG(workspace_bytes >= pseld_bn2d_workspace(rows, C), "bn2d_stats: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    const int rpb = bn2d_rows_per_block(rows, C), nb = pseld_cdiv(rows, rpb);
    PSELD_CHECK_ARG(C % 8 == 0, "bn2d_stats: C must be a multiple of 8");
    const dim3 grid(nb, pseld_cdiv(C / 8, C / 8 < 256 ? C / 8 : 256));
    if (dtype == PSELD_BF16)
        hipLaunchKernelGGL((bn2d_sums_kernel<bf16_t, 0>), grid, dim3(256), 0, s, (const bf16_t*)X, (const bf16_t*)nullptr, (const bf16_t*)nullptr, (const float*)nullptr, workspace, rows, C, rpb);
    else if (dtype == PSELD_F32)
        hipLaunchKernelGGL((bn2d_sums_kernel<float, 0>), grid, dim3(256), 0, s, (const float*)X, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, workspace, rows, C, rpb);
    else { pseld_set_error("bn2d_stats: unknown dtype"); return PSELD_ERR_BAD_ARG; }
    pseld_reduce_slabs(workspace, sums, (long)C * 2, nb, (long)C * 2, 0, s);
    PSELD_LAUNCH_CHECK("bn2d_stats");
    return PSELD_OK;
}
extern "C" int pseld_bn_relu_fwd(int dtype, const void* X, const float* scale_shift, void* Y, long rows, int C, void* stream) {
    PSELD_CHECK_ARG(X && scale_shift && Y && rows > 0 && C > 0, "bn_relu_fwd: bad argument");
    hipStream_t s = (hipStream_t)stream;
    PSELD_CHECK_ARG(C % 8 == 0, "bn_relu_fwd: C must be a multiple of 8");
    const long total = rows * C / 8;
    CNN_DISPATCH("bn_relu_fwd", hipLaunchKernelGGL(bn_relu_fwd_kernel<T>, dim3(pseld_cdiv(total, 256)), dim3(256), 0, s, (const T*)X,
                                                   scale_shift, (T*)Y, C, total, 0.f));
}
/* y = x*scale + shift without the ReLU (the Conformer's BatchNorm1d, convolution.py:129); its backward is
 * pseld_bn_relu_bwd with Y = NULL */
extern "C" int pseld_bn_affine_fwd(int dtype, const void* X, const float* scale_shift, void* Y, long rows, int C, void* stream) {
    PSELD_CHECK_ARG(X && scale_shift && Y && rows > 0 && C > 0, "bn_affine_fwd: bad argument");
    hipStream_t s = (hipStream_t)stream;
    PSELD_CHECK_ARG(C % 8 == 0, "bn_affine_fwd: C must be a multiple of 8");
    const long total = rows * C / 8;
    CNN_DISPATCH("bn_affine_fwd", hipLaunchKernelGGL(bn_relu_fwd_kernel<T>, dim3(pseld_cdiv(total, 256)), dim3(256), 0, s, (const T*)X,
                                                     scale_shift, (T*)Y, C, total, -INFINITY));
}
/* backward of y = relu(bn(x)) in two halves, so that a data-parallel caller can all-reduce the per-channel sums in between
 * (torch.nn.SyncBatchNorm, which configs/trainer/gpu.yaml:9 `sync_batchnorm: true` installs on every BatchNorm):
 *   sums:  sums f32[C][2] = (sum g * xhat, sum g) over this rank's rows, g = dY * (Y > 0) (Y = NULL: no ReLU); dgamma / dbeta are
 *          written from these rank-local sums (the gradient all-reduce averages them like every other parameter gradient)
 *   apply: dX = gamma * rstd * (g - sums[c][1] * inv_count - xhat * sums[c][0] * inv_count) with the (summed) sums and
 *          inv_count = 1 / (rows over all ranks). */
extern "C" int pseld_bn_relu_bwd_sums(int dtype, const void* X, const void* Y, const void* dY, const float* mean_rstd, float* sums,
                                      float* dgamma, float* dbeta, long rows, int C, float* workspace, long workspace_bytes, void* stream) {
    PSELD_CHECK_ARG(X && dY && mean_rstd && sums && dgamma && dbeta && workspace, "bn_relu_bwd_sums: null pointer");
    PSELD_CHECK_ARG(workspace_bytes >= pseld_bn2d_workspace(rows, C), "bn_relu_bwd_sums: workspace too small");
    PSELD_CHECK_ARG(C % 8 == 0, "bn_relu_bwd_sums: C must be a multiple of 8");
    hipStream_t s = (hipStream_t)stream;
    const int rpb = bn2d_rows_per_block(rows, C), nb = pseld_cdiv(rows, rpb);
    const dim3 grid(nb, pseld_cdiv(C / 8, C / 8 < 256 ? C / 8 : 256));
    if (dtype == PSELD_BF16)
        hipLaunchKernelGGL((bn2d_sums_kernel<bf16_t, 1>), grid, dim3(256), 0, s, (const bf16_t*)X, (const bf16_t*)Y, (const bf16_t*)dY, mean_rstd, workspace, rows, C, rpb);
    else if (dtype == PSELD_F32)
        hipLaunchKernelGGL((bn2d_sums_kernel<float, 1>), grid, dim3(256), 0, s, (const float*)X, (const float*)Y, (const float*)dY, mean_rstd, workspace, rows, C, rpb);
    else { pseld_set_error("bn_relu_bwd_sums: unknown dtype"); return PSELD_ERR_BAD_ARG; }
    pseld_reduce_slabs(workspace, sums, (long)C * 2, nb, (long)C * 2, 0, s);
    hipLaunchKernelGGL(deinterleave2_kernel, dim3(pseld_cdiv(C, 256)), dim3(256), 0, s, sums, C, dgamma, dbeta);
    PSELD_LAUNCH_CHECK("bn_relu_bwd_sums");
    return PSELD_OK;
}
extern "C" int pseld_bn_relu_bwd_apply(int dtype, const void* X, const void* Y, const void* dY, const float* mean_rstd, const float* gamma,
                                       const float* sums, float inv_count, void* dX, long rows, int C, void* stream) {
    PSELD_CHECK_ARG(X && dY && mean_rstd && gamma && sums && dX && rows > 0 && inv_count > 0.f, "bn_relu_bwd_apply: bad argument");
    PSELD_CHECK_ARG(C % 8 == 0, "bn_relu_bwd_apply: C must be a multiple of 8");
    hipStream_t s = (hipStream_t)stream;
    const long n = rows * C / 8;
    CNN_DISPATCH("bn_relu_bwd_apply", hipLaunchKernelGGL(bn_relu_bwd_kernel<T>, dim3(pseld_cdiv(n, 256)), dim3(256), 0, s, (const T*)X, (const T*)Y,
                                                         (const T*)dY, mean_rstd, gamma, sums, inv_count, (T*)dX, C, n));
}
/* both halves on one rank: dX, dgamma, dbeta (train-mode batch statistics) */
extern "C" int pseld_bn_relu_bwd(int dtype, const void* X, const void* Y, const void* dY, const float* mean_rstd,
                                 const float* gamma, void* dX, float* dgamma, float* dbeta, long rows, int C, float* workspace,
                                 long workspace_bytes, void* stream) {
    PSELD_CHECK_ARG(X && dY && mean_rstd && gamma && dX && dgamma && dbeta && workspace, "bn_relu_bwd: null pointer");
    PSELD_CHECK_ARG(workspace_bytes >= pseld_bn2d_workspace(rows, C), "bn_relu_bwd: workspace too small");
    const int rpb = bn2d_rows_per_block(rows, C), nb = pseld_cdiv(rows, rpb);
    float* total = workspace + (long)nb * C * 2;          // (the workspace has one extra [C][2] row for the reduced sums)
    int rc = pseld_bn_relu_bwd_sums(dtype, X, Y, dY, mean_rstd, total, dgamma, dbeta, rows, C, workspace, workspace_bytes, stream);
    if (rc) return rc;
    return pseld_bn_relu_bwd_apply(dtype, X, Y, dY, mean_rstd, gamma, total, 1.f / (float)rows, dX, rows, C, stream);
}

extern "C" int pseld_avgpool_fwd(int dtype, const void* X, void* Y, int B, int Tn, int Fn, int C, int pt, int pf, void* stream) {
    PSELD_CHECK_ARG(X && Y && B > 0 && pt > 0 && pf > 0 && Tn >= pt && Fn >= pf && C > 0, "avgpool_fwd: bad argument");
    hipStream_t s = (hipStream_t)stream;
    PSELD_CHECK_ARG(C % 8 == 0, "avgpool_fwd: C must be a multiple of 8");
    const long total = (long)B * (Tn / pt) * (Fn / pf) * (C / 8);
    CNN_DISPATCH("avgpool_fwd", hipLaunchKernelGGL(avgpool_fwd_kernel<T>, dim3(pseld_cdiv(total, 256)), dim3(256), 0, s, (const T*)X, (T*)Y,
                                                   Tn, Fn, C, pt, pf, total));
}
extern "C" int pseld_avgpool_bwd(int dtype, const void* dY, void* dX, int B, int Tn, int Fn, int C, int pt, int pf, void* stream) {
    PSELD_CHECK_ARG(dY && dX && B > 0 && pt > 0 && pf > 0 && Tn >= pt && Fn >= pf && C > 0, "avgpool_bwd: bad argument");
    hipStream_t s = (hipStream_t)stream;
    PSELD_CHECK_ARG(C % 8 == 0, "avgpool_bwd: C must be a multiple of 8");
    const long total = (long)B * Tn * Fn * (C / 8);
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
