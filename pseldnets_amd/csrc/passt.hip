// PaSST front and back ends around the transformer blocks (gfx950): everything between the scalar BatchNorms and
// the first block, and between the last block and the fc head. All HBM-bound address maps / small reductions.
//
// Replaces (reference, /root/reference/src): models/accdoa.py:320-327 (in-place scalar BN), the im2col of
// models/components/model_utilities.py:174-213 PatchEmbed (Conv2d k16 s10 pad 3 on x.transpose(-1,-2)),
// models/components/passt.py:219-247 (time/freq positional embeddings, cls/dist tokens + new_pos_embed, concat),
// :296-300 (drop the 2 tokens, mean over the frequency rows of the patch grid), models/accdoa.py:328 tanh —
// and the autograd of each.
#include "common.h"

namespace {

constexpr int PS = 16, STR = 10, PAD = 3, MEL = 64, FG = 6;   // patch 16, stride 10, pad (16-10)/2; 64 mel bins -> 6 rows

// A[(b, fg, tg)][c*256 + kf*16 + kt] = bn(feat[b, c, t = 10 tg - 3 + kt, f = 10 fg - 3 + kf]) (0 outside);
// one thread = 8 consecutive kt.
template <typename T>
__global__ __launch_bounds__(256) void passt_patchify_kernel(const float* __restrict__ feat, const float* __restrict__ ss,
                                                             T* __restrict__ A, int Cin, int Ctot, int Tn, int Tg,
                                                             long chunks) {
    const long id = (long)blockIdx.x * 256 + threadIdx.x;
    if (id >= chunks) return;
    const int cpr = Cin * 32;                       // 8-element chunks per row
    const long row = id / cpr;
    const int k = (int)(id - row * cpr);
    const int c = k >> 5, kf = (k >> 1) & 15, kt0 = (k & 1) * 8;
    const int tg = (int)(row % Tg);
    const long rest = row / Tg;
    const int fg = (int)(rest % FG);
    const long b = rest / FG;
    const int f = fg * STR - PAD + kf;
    float v[8];
    if (f >= 0 && f < MEL) {
        const float sc = ss[2 * (c * MEL + f)], sh = ss[2 * (c * MEL + f) + 1];
        const float* src = feat + ((b * Ctot + c) * Tn) * MEL + f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int t = tg * STR - PAD + kt0 + j;
            v[j] = (t >= 0 && t < Tn) ? src[(long)t * MEL] * sc + sh : 0.f;
        }
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = 0.f;
    }
    store8<T>(A + id * 8, v);
}

// BN parameter gradients through the overlapping patches: d(bn out)[b,c,t,f] = sum of the (<= 2x2) dA entries that
// read that pixel. grid (blocks over (b,t), Cin); lane = f.
template <typename T>
__global__ __launch_bounds__(256) void passt_bn_bwd_kernel(const float* __restrict__ feat, const float* __restrict__ mean_rstd,
                                                           const T* __restrict__ dA, float* __restrict__ part, int B, int Cin,
                                                           int Ctot, int Tn, int Tg, int rows_per_block) {
    __shared__ float red[4][64][2];
    const int f = threadIdx.x & 63, w = threadIdx.x >> 6, c = blockIdx.y;
    const long total = (long)B * Tn;
    const long beg = (long)blockIdx.x * rows_per_block, end = min(total, beg + rows_per_block);
    const float mean = mean_rstd[2 * (c * MEL + f)], rstd = mean_rstd[2 * (c * MEL + f) + 1];
    // the (fg, kf) pairs covering mel bin f
    int fgs[2], kfs[2], nf = 0;
    {
        const int u = f + PAD, g = u / STR, kf = u - g * STR;
        if (g < FG) { fgs[nf] = g; kfs[nf] = kf; ++nf; }
        if (kf + STR < PS && g >= 1 && g - 1 < FG) { fgs[nf] = g - 1; kfs[nf] = kf + STR; ++nf; }
    }
    const long lda = (long)Cin * 256;
    float dw = 0.f, db = 0.f;
    for (long q = beg + w; q < end; q += 4) {
        const long b = q / Tn;
        const int t = (int)(q - b * Tn);
        const int u = t + PAD, g = u / STR, kt = u - g * STR;
        float gsum = 0.f;
        for (int i = 0; i < nf; ++i) {
            const long rowbase = (b * FG + fgs[i]) * Tg;
            const int col = c * 256 + kfs[i] * 16;
            if (g < Tg) gsum += to_f32<T>(dA[(rowbase + g) * lda + col + kt]);
            if (kt + STR < PS && g >= 1 && g - 1 < Tg) gsum += to_f32<T>(dA[(rowbase + g - 1) * lda + col + kt + STR]);
        }
        const float xh = (feat[((b * Ctot + c) * Tn + t) * MEL + f] - mean) * rstd;
        dw += gsum * xh;
        db += gsum;
    }
    red[w][f][0] = dw; red[w][f][1] = db;
    __syncthreads();
    if (w == 0) {
        float* o = part + (((long)blockIdx.x * Cin + c) * 64 + f) * 2;
        o[0] = red[0][f][0] + red[1][f][0] + red[2][f][0] + red[3][f][0];
        o[1] = red[0][f][1] + red[1][f][1] + red[2][f][1] + red[3][f][1];
    }
}
__global__ void passt_bn_finish_kernel(const float* __restrict__ part, int nblocks, int n, float* dweight, float* dbias,
                                       int accumulate) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float dw = 0.f, db = 0.f;
    for (int k = 0; k < nblocks; ++k) { dw += part[((long)k * n + i) * 2]; db += part[((long)k * n + i) * 2 + 1]; }
    if (accumulate) { dw += dweight[i]; db += dbias[i]; }
    dweight[i] = dw; dbias[i] = db;
}

// X[b, 0] = cls + npos[0]; X[b, 1] = dist + npos[1]; X[b, 2 + fg*Tg + tg] = P[b, fg*Tg + tg] + tpos[:, tg] + fpos[:, fg]
template <typename T>
__global__ __launch_bounds__(256) void passt_assemble_kernel(const T* __restrict__ P, const float* __restrict__ tpos,
                                                             const float* __restrict__ fpos, const float* __restrict__ cls,
                                                             const float* __restrict__ dist, const float* __restrict__ npos,
                                                             T* __restrict__ X, int E, int Tg, long chunks) {
    const long id = (long)blockIdx.x * 256 + threadIdx.x;
    if (id >= chunks) return;
    const int cpr = E / 8, L = FG * Tg + 2;
    const long row = id / cpr;
    const int e0 = (int)(id - row * cpr) * 8;
    const long b = row / L;
    const int n = (int)(row - b * L);
    float v[8];
    if (n < 2) {
        const float* tok = n == 0 ? cls : dist;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = tok[e0 + j] + npos[n * E + e0 + j];
    } else {
        const int fg = (n - 2) / Tg, tg = (n - 2) - fg * Tg;
        load8<T>(P + (b * (L - 2) + (n - 2)) * E + e0, v);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] += tpos[(long)(e0 + j) * Tg + tg] + fpos[(e0 + j) * FG + fg];
    }
    store8<T>(X + id * 8, v);
}

// S[n, e] = sum_b dX[b, n, e] (fp32) and dP[b, n-2, e] = dX[b, n, e]
template <typename T>
__global__ __launch_bounds__(256) void passt_assemble_bwd_kernel(const T* __restrict__ dX, T* __restrict__ dP, float* __restrict__ S,
                                                                 int B, int E, int L) {
    const long id = (long)blockIdx.x * 256 + threadIdx.x;
    const int cpr = E / 8;
    if (id >= (long)L * cpr) return;
    const int n = (int)(id / cpr), e0 = (int)(id - (long)n * cpr) * 8;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int b = 0; b < B; ++b) {
        float v[8];
        load8<T>(dX + ((long)b * L + n) * E + e0, v);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += v[j];
        if (n >= 2) store8<T>(dP + ((long)b * (L - 2) + (n - 2)) * E + e0, v);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) S[(long)n * E + e0 + j] = acc[j];
}
// positional / token gradients from S: one thread per embedding channel
__global__ void passt_pos_grads_kernel(const float* __restrict__ S, float* __restrict__ dtpos, float* __restrict__ dfpos,
                                       float* __restrict__ dcls, float* __restrict__ ddist, float* __restrict__ dnpos, int E,
                                       int Tg) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    dcls[e] = S[e]; dnpos[e] = S[e];
    ddist[e] = S[E + e]; dnpos[E + e] = S[E + e];
    float ft[FG];
#pragma unroll
    for (int f = 0; f < FG; ++f) ft[f] = 0.f;
    for (int t = 0; t < Tg; ++t) {
        float s = 0.f;
#pragma unroll
        for (int f = 0; f < FG; ++f) {
            const float v = S[(long)(2 + f * Tg + t) * E + e];
            s += v;
            ft[f] += v;
        }
        dtpos[(long)e * Tg + t] = s;
    }
#pragma unroll
    for (int f = 0; f < FG; ++f) dfpos[e * FG + f] = ft[f];
}

// Structured patch-out (passt.py:250-258,333-338): Y[b, j] = X[b, src[j]] keeps a subset of the token rows; its adjoint writes
// dX[b, i] = dY[b, inv[i]] (zeros where inv[i] < 0). One thread = 8 channels.
template <typename T>
__global__ __launch_bounds__(256) void rows_select_kernel(const T* __restrict__ X, const int* __restrict__ map, T* __restrict__ Y, int n_src,
                                                          int n_dst, int E, long chunks) {
    const long id = (long)blockIdx.x * 256 + threadIdx.x;
    if (id >= chunks) return;
    const int cpr = E / 8;
    const long row = id / cpr;
    const int e0 = (int)(id - row * cpr) * 8;
    const long b = row / n_dst;
    const int j = (int)(row - b * n_dst), i = map[j];
    float v[8];
    if (i >= 0) load8<T>(X + (b * n_src + i) * E + e0, v);
    else {
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = 0.f;
    }
    store8<T>(Y + id * 8, v);
}

// Y[b, tg] = mean_fg X[b, 2 + fg*Tg + tg]      (passt.py:296-300)
template <typename T>
__global__ __launch_bounds__(256) void passt_pool_fwd_kernel(const T* __restrict__ X, T* __restrict__ Y, int E, int Tg, long chunks) {
    const long id = (long)blockIdx.x * 256 + threadIdx.x;
    if (id >= chunks) return;
    const int cpr = E / 8, L = FG * Tg + 2;
    const long row = id / cpr;
    const int e0 = (int)(id - row * cpr) * 8;
    const long b = row / Tg;
    const int tg = (int)(row - b * Tg);
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int f = 0; f < FG; ++f) {
        float v[8];
        load8<T>(X + (b * L + 2 + f * Tg + tg) * E + e0, v);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += v[j];
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] *= (1.f / FG);
    store8<T>(Y + id * 8, acc);
}
template <typename T>
__global__ __launch_bounds__(256) void passt_pool_bwd_kernel(const T* __restrict__ dY, T* __restrict__ dX, int E, int Tg, long chunks) {
    const long id = (long)blockIdx.x * 256 + threadIdx.x;
    if (id >= chunks) return;
    const int cpr = E / 8, L = FG * Tg + 2;
    const long row = id / cpr;
    const int e0 = (int)(id - row * cpr) * 8;
    const long b = row / L;
    const int n = (int)(row - b * L);
    float v[8];
    if (n < 2) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = 0.f;
    } else {
        const int tg = (n - 2) % Tg;
        load8<T>(dY + (b * Tg + tg) * E + e0, v);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] *= (1.f / FG);
    }
    store8<T>(dX + id * 8, v);
}

// y f32[rows, D] = act(z[rows, :D]) out of the padded GEMM output (ldz >= D); TANH=false: the plain f32 copy (SED logits)
template <typename T, bool TANH = true>
__global__ void tanh_fwd_kernel(const T* __restrict__ z, int ldz, float* __restrict__ y, int ldy, long rows, int D) {
    const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= rows * D) return;
    const long r = id / D;
    const int d = (int)(id - r * D);
    const float v = to_f32<T>(z[r * ldz + d]);
    y[r * ldy + d] = TANH ? tanhf(v) : v;
}
template <typename T, bool TANH = true>
__global__ void tanh_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, int ldy, T* __restrict__ dz, int ldz,
                                long rows, int D) {
    const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= rows * ldz) return;
    const long r = id / ldz;
    const int d = (int)(id - r * ldz);
    float g = 0.f;
    if (d < D) {
        g = dy[r * ldy + d];
        if (TANH) { const float t = y[r * ldy + d]; g *= 1.f - t * t; }
    }
    dz[id] = from_f32<T>(g);
}

constexpr int BN_ROWS_PER_BLOCK = 512;

}  // namespace

#define PASST_DISPATCH(name, CALL)                                            \
    if (dtype == PSELD_BF16) { using T = bf16_t; CALL; }                          \
    else if (dtype == PSELD_F32) { using T = float; CALL; }                       \
    else { pseld_set_error(name ": unknown dtype"); return PSELD_ERR_BAD_ARG; }   \
    PSELD_LAUNCH_CHECK(name);                                                     \
    return PSELD_OK

extern "C" int pseld_passt_grid_t(int T) { return (T + 2 * PAD - PS) / STR + 1; }

extern "C" int pseld_passt_patchify(int dtype, const float* feat, const float* scale_shift, void* A, int B, int Cin, int Ctot,
                                    int Tn, void* stream) {
    PSELD_CHECK_ARG(feat && scale_shift && A, "passt_patchify: null pointer");
    PSELD_CHECK_ARG(B > 0 && Cin > 0 && Ctot >= Cin && Tn >= PS, "passt_patchify: bad geometry");
    hipStream_t s = (hipStream_t)stream;
    const int Tg = pseld_passt_grid_t(Tn);
    const long chunks = (long)B * FG * Tg * Cin * 32;
    PASST_DISPATCH("passt_patchify", hipLaunchKernelGGL(passt_patchify_kernel<T>, dim3(pseld_cdiv(chunks, 256)), dim3(256), 0, s,
                                                         feat, scale_shift, (T*)A, Cin, Ctot, Tn, Tg, chunks));
}

extern "C" long pseld_passt_bn_bwd_workspace(int B, int Cin, int T) {
    return (long)pseld_cdiv((long)B * T, BN_ROWS_PER_BLOCK) * Cin * 64 * 2 * (long)sizeof(float);
}
extern "C" int pseld_passt_bn_bwd(int dtype, const float* feat, const float* mean_rstd, const void* dA, float* dweight,
                                  float* dbias, int B, int Cin, int Ctot, int T, int accumulate, float* workspace,
                                  long workspace_bytes, void* stream) {
    PSELD_CHECK_ARG(feat && mean_rstd && dA && dweight && dbias && workspace, "passt_bn_bwd: null pointer");
    PSELD_CHECK_ARG(Ctot >= Cin, "passt_bn_bwd: Ctot < Cin");
    PSELD_CHECK_ARG(workspace_bytes >= pseld_passt_bn_bwd_workspace(B, Cin, T), "passt_bn_bwd: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    const int Tg = pseld_passt_grid_t(T);
    const int nb = pseld_cdiv((long)B * T, BN_ROWS_PER_BLOCK);
    if (dtype == PSELD_BF16)
        hipLaunchKernelGGL(passt_bn_bwd_kernel<bf16_t>, dim3(nb, Cin), dim3(256), 0, s, feat, mean_rstd, (const bf16_t*)dA, workspace, B, Cin, Ctot, T, Tg, BN_ROWS_PER_BLOCK);
    else if (dtype == PSELD_F32)
        hipLaunchKernelGGL(passt_bn_bwd_kernel<float>, dim3(nb, Cin), dim3(256), 0, s, feat, mean_rstd, (const float*)dA, workspace, B, Cin, Ctot, T, Tg, BN_ROWS_PER_BLOCK);
    else { pseld_set_error("passt_bn_bwd: unknown dtype"); return PSELD_ERR_BAD_ARG; }
    const int n = Cin * 64;
    hipLaunchKernelGGL(passt_bn_finish_kernel, dim3(pseld_cdiv(n, 256)), dim3(256), 0, s, workspace, nb, n, dweight, dbias, accumulate);
    PSELD_LAUNCH_CHECK("passt_bn_bwd");
    return PSELD_OK;
}

extern "C" int pseld_passt_assemble_fwd(int dtype, const void* P, const float* tpos, const float* fpos, const float* cls,
                                        const float* dist, const float* npos, void* X, int B, int E, int Tg, void* stream) {
    PSELD_CHECK_ARG(P && tpos && fpos && cls && dist && npos && X, "passt_assemble_fwd: null pointer");
    PSELD_CHECK_ARG(B > 0 && E % 8 == 0 && Tg > 0, "passt_assemble_fwd: bad geometry");
    hipStream_t s = (hipStream_t)stream;
    const long chunks = (long)B * (FG * Tg + 2) * (E / 8);
    PASST_DISPATCH("passt_assemble_fwd", hipLaunchKernelGGL(passt_assemble_kernel<T>, dim3(pseld_cdiv(chunks, 256)), dim3(256), 0,
                                                             s, (const T*)P, tpos, fpos, cls, dist, npos, (T*)X, E, Tg, chunks));
}

extern "C" long pseld_passt_assemble_bwd_workspace(int E, int Tg) { return (long)(FG * Tg + 2) * E * (long)sizeof(float); }
extern "C" int pseld_passt_assemble_bwd(int dtype, const void* dX, void* dP, float* dtpos, float* dfpos, float* dcls,
                                        float* ddist, float* dnpos, int B, int E, int Tg, float* workspace,
                                        long workspace_bytes, void* stream) {
    PSELD_CHECK_ARG(dX && dP && dtpos && dfpos && dcls && ddist && dnpos && workspace, "passt_assemble_bwd: null pointer");
    PSELD_CHECK_ARG(B > 0 && E % 8 == 0 && Tg > 0, "passt_assemble_bwd: bad geometry");
    PSELD_CHECK_ARG(workspace_bytes >= pseld_passt_assemble_bwd_workspace(E, Tg), "passt_assemble_bwd: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    const int L = FG * Tg + 2;
    const long n = (long)L * (E / 8);
    if (dtype == PSELD_BF16)
        hipLaunchKernelGGL(passt_assemble_bwd_kernel<bf16_t>, dim3(pseld_cdiv(n, 256)), dim3(256), 0, s, (const bf16_t*)dX, (bf16_t*)dP, workspace, B, E, L);
    else if (dtype == PSELD_F32)
        hipLaunchKernelGGL(passt_assemble_bwd_kernel<float>, dim3(pseld_cdiv(n, 256)), dim3(256), 0, s, (const float*)dX, (float*)dP, workspace, B, E, L);
    else { pseld_set_error("passt_assemble_bwd: unknown dtype"); return PSELD_ERR_BAD_ARG; }
    hipLaunchKernelGGL(passt_pos_grads_kernel, dim3(pseld_cdiv(E, 128)), dim3(128), 0, s, workspace, dtpos, dfpos, dcls, ddist, dnpos, E, Tg);
    PSELD_LAUNCH_CHECK("passt_assemble_bwd");
    return PSELD_OK;
}

extern "C" int pseld_rows_select(int dtype, const void* X, const int* map, void* Y, int B, int n_src, int n_dst, int E, void* stream) {
    PSELD_CHECK_ARG(X && map && Y && X != Y && B > 0 && n_src > 0 && n_dst > 0 && E % 8 == 0, "rows_select: bad argument");
    hipStream_t s = (hipStream_t)stream;
    const long chunks = (long)B * n_dst * (E / 8);
    PASST_DISPATCH("rows_select", hipLaunchKernelGGL(rows_select_kernel<T>, dim3(pseld_cdiv(chunks, 256)), dim3(256), 0, s, (const T*)X, map,
                                                      (T*)Y, n_src, n_dst, E, chunks));
}

extern "C" int pseld_passt_pool_fwd(int dtype, const void* X, void* Y, int B, int E, int Tg, void* stream) {
    PSELD_CHECK_ARG(X && Y && B > 0 && E % 8 == 0 && Tg > 0, "passt_pool_fwd: bad argument");
    hipStream_t s = (hipStream_t)stream;
    const long chunks = (long)B * Tg * (E / 8);
    PASST_DISPATCH("passt_pool_fwd", hipLaunchKernelGGL(passt_pool_fwd_kernel<T>, dim3(pseld_cdiv(chunks, 256)), dim3(256), 0, s,
                                                         (const T*)X, (T*)Y, E, Tg, chunks));
}
extern "C" int pseld_passt_pool_bwd(int dtype, const void* dY, void* dX, int B, int E, int Tg, void* stream) {
    PSELD_CHECK_ARG(dY && dX && B > 0 && E % 8 == 0 && Tg > 0, "passt_pool_bwd: bad argument");
    hipStream_t s = (hipStream_t)stream;
    const long chunks = (long)B * (FG * Tg + 2) * (E / 8);
    PASST_DISPATCH("passt_pool_bwd", hipLaunchKernelGGL(passt_pool_bwd_kernel<T>, dim3(pseld_cdiv(chunks, 256)), dim3(256), 0, s,
                                                         (const T*)dY, (T*)dX, E, Tg, chunks));
}

extern "C" int pseld_fc_out_fwd(int dtype, const void* z, int ldz, float* y, int ldy, long rows, int D, int act, void* stream) {
    PSELD_CHECK_ARG(z && y && rows > 0 && D > 0 && ldz >= D && ldy >= D, "fc_out_fwd: bad argument");
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid(pseld_cdiv(rows * D, 256));
    if (act) {
        PASST_DISPATCH("fc_out_fwd", hipLaunchKernelGGL((tanh_fwd_kernel<T, true>), grid, dim3(256), 0, s, (const T*)z, ldz, y, ldy, rows, D));
    }
    PASST_DISPATCH("fc_out_fwd", hipLaunchKernelGGL((tanh_fwd_kernel<T, false>), grid, dim3(256), 0, s, (const T*)z, ldz, y, ldy, rows, D));
}
extern "C" int pseld_fc_out_bwd(int dtype, const float* dy, const float* y, int ldy, void* dz, int ldz, long rows, int D, int act,
                                void* stream) {
    PSELD_CHECK_ARG(dy && dz && (y || !act) && rows > 0 && D > 0 && ldz >= D && ldy >= D, "fc_out_bwd: bad argument");
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid(pseld_cdiv(rows * ldz, 256));
    if (act) {
        PASST_DISPATCH("fc_out_bwd", hipLaunchKernelGGL((tanh_bwd_kernel<T, true>), grid, dim3(256), 0, s, dy, y, ldy, (T*)dz, ldz, rows, D));
    }
    PASST_DISPATCH("fc_out_bwd", hipLaunchKernelGGL((tanh_bwd_kernel<T, false>), grid, dim3(256), 0, s, dy, y, ldy, (T*)dz, ldz, rows, D));
}
extern "C" int pseld_tanh_fwd(int dtype, const void* z, int ldz, float* y, long rows, int D, void* stream) {
    return pseld_fc_out_fwd(dtype, z, ldz, y, D, rows, D, 1, stream);
}
extern "C" int pseld_tanh_bwd(int dtype, const float* dy, const float* y, void* dz, int ldz, long rows, int D, void* stream) {
    return pseld_fc_out_bwd(dtype, dy, y, D, dz, ldz, rows, D, 1, stream);
}
