// MFMA GEMM for the Swin/HTS-AT linear layers (forward, input-gradient, weight-gradient) on gfx950.
//
// Replaces the nn.Linear / Conv2d-as-GEMM calls of the reference hot path and their autograd:
//   htsat.py:112-145 (qkv, proj), model_utilities.py:159-171 (Mlp fc1/fc2), htsat.py:290-311 (PatchMerging
//   reduction), model_utilities.py:205-213 (PatchEmbed proj), accdoa.py:230 (tscam_conv).
//
// One kernel template, three operand layouts (all row-major, leading dimensions in elements):
//   TA=0,TB=0  C[M,N] = A[M,K] * B[N,K]^T          forward  (x @ W^T)
//   TA=0,TB=1  C[M,N] = A[M,K] * B[K,N]            dX = dY @ W
//   TA=1,TB=1  C[M,N] = A[K,M]^T * B[K,N]          dW = dY^T @ X   (split-K over the token dimension,
//                                                  fp32 partial slabs reduced by splitk_reduce)
// Element type T is bf16 (v_mfma_f32_32x32x16_bf16) or f32 (v_mfma_f32_32x32x2_f32, exact f32 FMA chain,
// used as the parity mode); both share one fragment layout: lane (r = l&31, h = l>>5) holds the 8
// contraction elements k = 16*kk + 8*h + j of row/column r, so every loader is dtype-agnostic.
//
// Workgroup = 4 waves (8 for the 256x192 weight-gradient tile), each owning a 64x96 output tile (2x3 MFMA tiles, 96
// accumulators): WM=4,WN=1 -> 256x96 block tile (N == 96-ish layers), WM=2,WN=2 -> 128x192, WM=4,WN=2 -> 256x192.
// Two feeders share the tiles, fragment maps and epilogues:
//   * gemm_kernel: K in 128-byte slices (64 bf16 / 32 f32) staged HBM -> registers -> LDS with the next slice's loads in
//     flight during the MFMAs. Non-transposed LDS rows are 128 B + 16 B pad (conflict-free ds_read_b128 over a
//     16-lane group); transposed operands keep their [k][m] image and are read with ds_read_b64_tr_b16 (bf16), so no
//     transpose pass ever runs. Every f32 product, every weight gradient, shapes the DMA feeder does not take.
//   * gemm_dma_kernel: bf16 products with both operands k-contiguous (forward; input gradient through the arena's
//     pre-transposed weight copies) fed by LDS-DMA, no staging registers -> three workgroups per CU.
// bf16 epilogue: the tile is computed transposed, staged through LDS and written as 16-byte coalesced rows by one
// straight-line instance per fused-option set (bias / residual / DropPath factor / GELU pair / x aux).
// Launches are 1-D with an XCD-aware tile order (tile_coords). The weight gradient is split-K over the tokens with
// one resident round of workgroups (wgrad_splits_for), fp32 slabs and a deterministic reduce.
//
// Roofline: HBM-bound for stages 0-2 of HTS-AT (arithmetic intensity <= 307 flop/B, ridge 314), MFMA-bound for stage 3
// and PaSST. Measured per shape: tools/gemm_shapes.py; phase stamps: tools/gemm_stamps.py.
#include "common.h"
#include "gemm8.h"
#include <stdlib.h>

void pseld_reduce_slabs(const float* slabs, float* out, long n, int splits, long slab_stride, int accumulate,
                        hipStream_t stream);

// name of the kernel the last pseld_gemm / pseld_gemm_wgrad call launched (measurement aid: bench.py attributes its HIP-event
// launch times to kernel symbols with it)
static const char* g_last_gemm_kernel = "";
extern "C" const char* pseld_gemm_last_kernel(void) { return g_last_gemm_kernel; }

namespace {

constexpr int ROWB = 144;  // bytes per non-transposed LDS row: 128 B of K + 16 B pad

enum { EPI_NONE = 0, EPI_BIAS = 1, EPI_RESID = 2, EPI_MULGELUGRAD = 4, EPI_ACCUM = 8, EPI_GELU_DUAL = 16, EPI_MULAUX = 32 };
enum { PRO_NONE = 0, PRO_GELU_A = 1, PRO_GELU_B = 2, PRO_ROWSCALE_A = 4 };   // ROWSCALE_A: weight gradient only

struct GemmArgs {
    const void* A;
    const void* B;
    void* C;
    const float* bias;      // [N]
    const void* resid;      // [M, ldr] (T)
    const float* rowscale;  // [ceil(M / rows_per_scale)] or null
    const void* aux;        // [M, ldaux] (T): pre-activation for EPI_MULGELUGRAD / multiplier for EPI_MULAUX
    void* C2;               // EPI_GELU_DUAL: C = gelu(v), C2 = gelu'(v), both [M, ldc]
    int M, N, K;
    int lda, ldb, ldc, ldr, ldaux;
    int rows_per_scale;
    int kchunk;             // contraction elements per split (multiple of BK); == K when no split
    long slab_stride;       // elements between split-K output slabs
    float* colsum;          // TA only: per-split column sums of A (= bias gradient), split z at colsum + z*colsum_stride
    long colsum_stride;
    int epi, pro;
    float ln_eps;             // gemm_dma_kernel<.., LNBWD>: epsilon of the LayerNorm whose backward runs in the epilogue
    unsigned long long* dbg;  // diagnostic build only: per-workgroup s_memtime stamps
    int nx, ny, nz;           // tile grid (N tiles, M tiles, K splits); the launch is 1-D and XCD-swizzled
    int xcd_swizzle;          // 0: plain x-fastest order (experiment knob PSELD_GEMM_XCD=0)
    // CONV kernels: the non-weight operand is the 3x3 / pad 1 im2col view [B*T*F, 9*C] (column = tap*C + c) of an NHWC
    // activation X[B*T*F, C]; it is never materialised, the loaders read X with shifted rows and zero the border
    int cv_T, cv_F, cv_C;
    float cv_rF, cv_rT;       // reciprocals for the row -> (t, f) decode
};

// x / d for 0 <= x < 2^24 through the reciprocal (exact after one fix-up step)
__device__ __forceinline__ int div_by(int x, int d, float rd) {
    int q = (int)((float)x * rd);
    const int r = x - q * d;
    q += (r >= d) - (r < 0);
    return q;
}

template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
    static constexpr int BK = 64;
    static constexpr int KSTEPS = 4;
    using Frag = bf16x8;
    static __device__ __forceinline__ void mma(const Frag& a, const Frag& b, f32x16& c) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
};
struct FragF32 { float v[8]; };
template <> struct Mma<float> {
    static constexpr int BK = 32;
    static constexpr int KSTEPS = 2;
    using Frag = FragF32;
    static __device__ __forceinline__ void mma(const Frag& a, const Frag& b, f32x16& c) {
#pragma unroll
        for (int j = 0; j < 8; ++j) c = __builtin_amdgcn_mfma_f32_32x32x2f32(a.v[j], b.v[j], c, 0, 0, 0);
    }
};

typedef __attribute__((address_space(3))) short4v* lds_s4_ptr;

// ---- fragment loaders ------------------------------------------------------------------------------
// Non-transposed image: [rows][ROWB bytes]; lane reads 8 consecutive k of its row.
template <typename T>
__device__ __forceinline__ typename Mma<T>::Frag ld_frag_n(const char* tile, int row, int kk, int h);
template <>
__device__ __forceinline__ bf16x8 ld_frag_n<bf16_t>(const char* tile, int row, int kk, int h) {
    return *(const bf16x8*)(tile + row * ROWB + (kk * 16 + 8 * h) * 2);
}
template <>
__device__ __forceinline__ FragF32 ld_frag_n<float>(const char* tile, int row, int kk, int h) {
    const f32x4* p = (const f32x4*)(tile + row * ROWB + (kk * 16 + 8 * h) * 4);
    const f32x4 a = p[0], b = p[1];
    FragF32 f;
#pragma unroll
    for (int j = 0; j < 4; ++j) { f.v[j] = a[j]; f.v[4 + j] = b[j]; }
    return f;
}
// Transposed image: [k rows][strideB bytes], columns = the operand's row/column index.
// bf16: two ds_read_b64_tr_b16, each delivering 4 consecutive k of this lane's column. Within a 16-lane
// group, lane 4q+p supplies the address of k-row q, columns 4p..4p+3 of the group's 16-column block.
template <typename T>
__device__ __forceinline__ typename Mma<T>::Frag ld_frag_t(const char* tile, int strideB, int col0, int kk, int lane);
template <>
__device__ __forceinline__ bf16x8 ld_frag_t<bf16_t>(const char* tile, int strideB, int col0, int kk, int lane) {
    const int i = lane & 15, q = i >> 2, p = i & 3, h = lane >> 5, gsel = (lane >> 4) & 1;
    const char* addr = tile + (kk * 16 + 8 * h + q) * strideB + (col0 + 16 * gsel + 4 * p) * 2;
    const short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_ptr)(addr));
    const short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_ptr)(addr + 4 * strideB));
    typedef __attribute__((ext_vector_type(8))) short short8v;
    short8v s;
    s[0] = lo[0]; s[1] = lo[1]; s[2] = lo[2]; s[3] = lo[3];
    s[4] = hi[0]; s[5] = hi[1]; s[6] = hi[2]; s[7] = hi[3];
    return __builtin_bit_cast(bf16x8, s);
}
template <>
__device__ __forceinline__ FragF32 ld_frag_t<float>(const char* tile, int strideB, int col0, int kk, int lane) {
    const int r = lane & 31, h = lane >> 5;
    FragF32 f;
#pragma unroll
    for (int j = 0; j < 8; ++j) f.v[j] = *(const float*)(tile + (kk * 16 + 8 * h + j) * strideB + (col0 + r) * 4);
    return f;
}

// transposed-image row stride in bytes: X columns of T, forced to 64 (mod 128) so the four k-rows a
// 32-lane half reads with ds_read_b64_tr_b16 fall in four different 64-byte bank groups.
template <typename T, int X> struct TStride {
    static constexpr int raw = X * (int)sizeof(T);
    static constexpr int value = (raw % 128 == 64) ? raw : raw + 64;
};

template <typename T> __device__ __forceinline__ float frag_sum(const typename Mma<T>::Frag& f);
template <> __device__ __forceinline__ float frag_sum<bf16_t>(const bf16x8& f) {
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) s += (float)f[j];
    return s;
}
template <> __device__ __forceinline__ float frag_sum<float>(const FragF32& f) {
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) s += f.v[j];
    return s;
}

__device__ __forceinline__ f32x4 gelu4_bf16(f32x4 v) {
    // v carries 8 packed bf16; apply exact GELU elementwise
    bf16x8 x = __builtin_bit_cast(bf16x8, v);
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = (bf16_t)gelu_f((float)x[i]);
    return __builtin_bit_cast(f32x4, x);
}
__device__ __forceinline__ f32x4 gelu4_f32(f32x4 v) {
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = gelu_f(v[i]);
    return v;
}
template <typename T> __device__ __forceinline__ f32x4 scale_chunk(f32x4 v, float s);
template <> __device__ __forceinline__ f32x4 scale_chunk<bf16_t>(f32x4 v, float s) {
    bf16x8 x = __builtin_bit_cast(bf16x8, v);
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = (bf16_t)((float)x[i] * s);
    return __builtin_bit_cast(f32x4, x);
}
template <> __device__ __forceinline__ f32x4 scale_chunk<float>(f32x4 v, float s) { return v * s; }
template <typename T> __device__ __forceinline__ f32x4 gelu_chunk(f32x4 v);
template <> __device__ __forceinline__ f32x4 gelu_chunk<bf16_t>(f32x4 v) { return gelu4_bf16(v); }
template <> __device__ __forceinline__ f32x4 gelu_chunk<float>(f32x4 v) { return gelu4_f32(v); }

// ---- bf16 epilogue, second half: C tile in LDS -> fused options -> 16-byte coalesced global stores -----------------
enum { EM_PLAIN, EM_RESID, EM_RESID_SCALE, EM_GELU_DUAL, EM_MULAUX, EM_MULAUX_SCALE, EM_SCALE, EM_GENERIC };

__device__ __forceinline__ void unpack8(const f32x4& p, float (&v)[8]) {
    const bf16x8 x = __builtin_bit_cast(bf16x8, p);
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = (float)x[k];
}
__device__ __forceinline__ f32x4 pack8(const float (&v)[8]) {
    bf16x8 x;
#pragma unroll
    for (int k = 0; k < 8; ++k) x[k] = (bf16_t)v[k];
    return __builtin_bit_cast(f32x4, x);
}

// The whole bf16 epilogue for one fused-option combination (straight-line code: no per-lane branches, every load
// unconditional at a clamped address). Order matters for latency: the residual / aux rows and the DropPath factors
// this thread will need are requested from HBM FIRST, so their latency hides behind the barrier and the
// accumulator -> LDS staging. bias_s must be a shared object of its own: when it shared the array with Cs every
// bias read between the C-tile writes was ordered after the previous write (24 serial LDS round trips, 4k cycles per
// tile measured with s_memtime).
template <int MODE, int WM, int WN>
__device__ __forceinline__ void staged_epilogue(const GemmArgs& g, char* smem, const float* bias_s, const f32x16 (&acc)[2][3],
                                                int m0, int n0, bf16_t* Cg, unsigned long long& t4) {
    constexpr int THREADS = WM * WN * 64, BM = WM * 64, BN = WN * 96;
    constexpr int CS_STRIDE = BN * 2 + 16;
    constexpr int CPR = BN / 8;
    constexpr int NCHUNK = BM * CPR / THREADS;        // 16-byte chunks per thread (12 for every tile shape)
    constexpr bool HAS_X = MODE == EM_RESID || MODE == EM_RESID_SCALE || MODE == EM_MULAUX || MODE == EM_MULAUX_SCALE;
    constexpr bool SCALED = MODE == EM_RESID_SCALE || MODE == EM_MULAUX_SCALE || MODE == EM_SCALE;
    constexpr int NCAND = BM / 64 + 2;   // DropPath factors a BM-row tile can meet when a sample has >= 64 rows
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN, r = lane & 31, h = lane >> 5;
    const bf16_t* Rg = (const bf16_t*)g.resid;
    const bf16_t* Ug = (const bf16_t*)g.aux;
    const int mlast = g.M - 1, nlast = max(g.N - 8, 0);

    f32x4 xv[HAS_X ? NCHUNK : 1];
    if constexpr (HAS_X) {
#pragma unroll
        for (int j = 0; j < NCHUNK; ++j) {
            const int c = tid + j * THREADS;
            const int row = c / CPR, cb = c - row * CPR;
            const int mr = min(m0 + row, mlast), nc = min(n0 + cb * 8, nlast);   // clamped: loads never branch
            if constexpr (MODE == EM_MULAUX || MODE == EM_MULAUX_SCALE) xv[j] = *(const f32x4*)(Ug + (long)mr * g.ldaux + nc);
            else xv[j] = *(const f32x4*)(Rg + (long)mr * g.ldr + nc);
        }
    }
    // DropPath: the (at most NCAND) per-sample factors this tile's rows can meet, fetched through uniform addresses
    float cand[SCALED ? NCAND : 1];
    const int s_first = SCALED ? m0 / g.rows_per_scale : 0;
    if constexpr (SCALED) {
        const int s_last = mlast / g.rows_per_scale;
#pragma unroll
        for (int i = 0; i < NCAND; ++i) cand[i] = g.rowscale[min(s_first + i, s_last)];
    }
    __syncthreads();  // every wave is done with the operand images
    char* Cs = smem;
#pragma unroll
    for (int ni = 0; ni < 3; ++ni)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
            __builtin_amdgcn_sched_barrier(0);   // one 32x32 tile at a time: 16 accumulators leave the AGPRs, not 96
            const int ml = wm * 64 + mi * 32 + r;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int nl = wn * 96 + ni * 32 + 8 * q + 4 * h;
                bf16x4 pk;
                const f32x4 bv = *(const f32x4*)(bias_s + nl);   // zero when there is no bias / past N
#pragma unroll
                for (int j = 0; j < 4; ++j) pk[j] = (bf16_t)(acc[mi][ni][4 * q + j] + bv[j]);
                *(bf16x4*)(Cs + ml * CS_STRIDE + nl * 2) = pk;
            }
        }
    __syncthreads();
    if (g.dbg) t4 = __builtin_amdgcn_s_memtime();
    // the chunk coordinates are RE-derived from a laundered thread id: otherwise the compiler keeps all 12 chunks'
    // rows / columns / addresses from the prefetch above alive across the staging (+100 VGPRs, one wave per SIMD less)
    int tid2 = tid;
    asm volatile("" : "+v"(tid2));
#pragma unroll
    for (int j = 0; j < NCHUNK; ++j) {
        if (j % 2 == 0) __builtin_amdgcn_sched_barrier(0);   // two chunks in flight: bounds the live registers
        const int c = tid2 + j * THREADS;
        const int row = c / CPR, cb = c - row * CPR;
        const int mrow = m0 + row, ncol = n0 + cb * 8;
        const bool okj = mrow < g.M && ncol < g.N;
        const long orowj = (long)min(mrow, mlast) * g.ldc + min(ncol, nlast);
        float v[8];
        unpack8(*(const f32x4*)(Cs + row * CS_STRIDE + cb * 16), v);
        float scj = 1.f;
        if constexpr (SCALED) {
            // sample index relative to the tile's first sample = number of sample boundaries at or below this row
            const int rel = mrow - s_first * g.rows_per_scale;
            scj = cand[0];
#pragma unroll
            for (int i = 1; i < NCAND; ++i) scj = (rel >= i * g.rows_per_scale) ? cand[i] : scj;
        }
        if constexpr (MODE == EM_RESID) {
            float x[8];
            unpack8(xv[j], x);
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] += x[k];
        } else if constexpr (MODE == EM_RESID_SCALE) {
            float x[8];
            unpack8(xv[j], x);
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = fmaf(v[k], scj, x[k]);
        } else if constexpr (MODE == EM_SCALE) {
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] *= scj;
        } else if constexpr (MODE == EM_MULAUX || MODE == EM_MULAUX_SCALE) {
            float x[8];
            unpack8(xv[j], x);
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] *= x[k] * scj;
        } else if constexpr (MODE == EM_GELU_DUAL) {
            float dv[8];
#pragma unroll
            for (int k = 0; k < 8; k += 2) {
                f32x2 xx = {v[k], v[k + 1]}, yy, dd;
                gelu_both2(xx, yy, dd);
                v[k] = yy[0]; v[k + 1] = yy[1]; dv[k] = dd[0]; dv[k + 1] = dd[1];
            }
            if (okj) *(f32x4*)((bf16_t*)g.C2 + orowj) = pack8(dv);
        } else if constexpr (MODE == EM_GENERIC) {
            float x[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (g.epi & (EPI_MULGELUGRAD | EPI_MULAUX))
                unpack8(*(const f32x4*)(Ug + (long)min(mrow, mlast) * g.ldaux + min(ncol, nlast)), x);
            if (g.rowscale) {
                const float scj = g.rowscale[min(mrow, mlast) / g.rows_per_scale];
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] *= scj;
            }
            if (g.epi & EPI_MULAUX) {
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] *= x[k];
            } else if (g.epi & EPI_MULGELUGRAD) {
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] *= gelu_grad_f(x[k]);
            }
            if (g.epi & EPI_RESID) {
                float rr[8];
                unpack8(*(const f32x4*)(Rg + (long)min(mrow, mlast) * g.ldr + min(ncol, nlast)), rr);
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] += rr[k];
            }
            if (g.epi & EPI_GELU_DUAL) {
                float dv[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) gelu_both(v[k], v[k], dv[k]);
                if (okj) *(f32x4*)((bf16_t*)g.C2 + orowj) = pack8(dv);
            }
        }
        if (okj) *(f32x4*)(Cg + orowj) = pack8(v);
    }
}

// Input-gradient product whose tile spans the LayerNorm row (BN == N = C): the LayerNorm BACKWARD runs in the epilogue. The accumulators are
// dxh = d(LN output); staged as bf16 (the value the separate LayerNorm kernel would read back), then TPR threads per token row (24 channels
// each) recompute mean / rstd from the LN input x (g.aux), form dx = rstd (dxh gamma - mean(dxh gamma) - xh mean(dxh gamma xh)) + dres
// (g.resid, optional) exactly as norm.hip:ln_bwd_kernel, and leave this row tile's partial sums of d(gamma) = sum dxh xh, d(beta) = sum dxh
// in g.C2 as [tile][2][N] fp32 (the layout pseld_reduce_slabs_batched takes from the stand-alone kernel). g.bias = gamma.
// (the packed row chunks are laundered between the passes: otherwise hipcc keeps the 24 unpacked floats of the first pass alive)
#define LAUNDER_XP asm volatile("" : "+v"(xp[0]), "+v"(xp[1]), "+v"(xp[2]))
template <int WM, int WN>
__device__ __forceinline__ void lnbwd_epilogue(const GemmArgs& g, char* smem, const float* gamma_s, const f32x16 (&acc)[2][3], int m0, int by) {
    constexpr int THREADS = WM * WN * 64, BM = WM * 64, BN = WN * 96;
    constexpr int CS_STRIDE = BN * 2 + 16;
    constexpr int TPR = BN / 24, RPP = THREADS / TPR, PASSES = BM / RPP;
    static_assert(PASSES * RPP == BM && (TPR == 4 || TPR == 8), "row passes must tile the block");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN, r = lane & 31, h = lane >> 5;
    const bf16_t* Xg = (const bf16_t*)g.aux;
    const bf16_t* Rg = (const bf16_t*)g.resid;
    bf16_t* Cg = (bf16_t*)g.C;
    __syncthreads();  // every wave is done with the operand images
    char* Cs = smem;
#pragma unroll
    for (int ni = 0; ni < 3; ++ni)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
            __builtin_amdgcn_sched_barrier(0);
            const int ml = wm * 64 + mi * 32 + r;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int nl = wn * 96 + ni * 32 + 8 * q + 4 * h;
                bf16x4 pk;
#pragma unroll
                for (int j = 0; j < 4; ++j) pk[j] = (bf16_t)acc[mi][ni][4 * q + j];
                *(bf16x4*)(Cs + ml * CS_STRIDE + nl * 2) = pk;
            }
        }
    __syncthreads();
    int tid2 = tid;
    asm volatile("" : "+v"(tid2));
    const int part = tid2 % TPR, rsub = tid2 / TPR;
    // register budget (3 workgroups per CU: 168): the row's x / dres chunks stay PACKED (12 + 12 registers) and are unpacked per use, the
    // staged dxh and gamma are re-read from LDS per use; only the 48 running sums are fp32 arrays
    float dgs[24], dbs[24];
#pragma unroll
    for (int k = 0; k < 24; ++k) { dgs[k] = 0.f; dbs[k] = 0.f; }
    const float invn = 1.f / (float)BN;
    const int mlast = g.M - 1;
    const float* gam = gamma_s + part * 24;
#pragma unroll 1
    for (int p = 0; p < PASSES; ++p) {
        const int row = p * RPP + rsub, m = m0 + row;
        const bool ok = m < g.M;
        const long mr = (long)min(m, mlast);
        f32x4 xp[3], rp[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            xp[c] = *(const f32x4*)(Xg + mr * g.ldaux + part * 24 + c * 8);
            rp[c] = Rg ? *(const f32x4*)(Rg + mr * g.ldr + part * 24 + c * 8) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        const char* drow = Cs + row * CS_STRIDE + part * 48;
        float sx = 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float t[8];
            unpack8(xp[c], t);
#pragma unroll
            for (int k = 0; k < 8; ++k) sx += t[k];
        }
#pragma unroll
        for (int o = 1; o < TPR; o <<= 1) sx += __shfl_xor(sx, o, 64);
        const float mean = sx * invn;
        float q = 0.f;
        LAUNDER_XP;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float t[8];
            unpack8(xp[c], t);
#pragma unroll
            for (int k = 0; k < 8; ++k) { const float u = t[k] - mean; q += u * u; }
        }
#pragma unroll
        for (int o = 1; o < TPR; o <<= 1) q += __shfl_xor(q, o, 64);
        const float rstd = rsqrtf(q * invn + g.ln_eps);
        float c1 = 0.f, c2 = 0.f;
        LAUNDER_XP;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float t[8], d[8];
            unpack8(xp[c], t);
            unpack8(*(const f32x4*)(drow + c * 16), d);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float xh = (t[k] - mean) * rstd, dyh = d[k] * gam[c * 8 + k];
                if (ok) { dgs[c * 8 + k] += d[k] * xh; dbs[c * 8 + k] += d[k]; }
                c1 += dyh; c2 += dyh * xh;
            }
        }
#pragma unroll
        for (int o = 1; o < TPR; o <<= 1) { c1 += __shfl_xor(c1, o, 64); c2 += __shfl_xor(c2, o, 64); }
        c1 *= invn; c2 *= invn;
        LAUNDER_XP;
        if (ok) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float t[8], d[8], rr[8], o8[8];
                unpack8(xp[c], t);
                unpack8(*(const f32x4*)(drow + c * 16), d);
                unpack8(rp[c], rr);
#pragma unroll
                for (int k = 0; k < 8; ++k) o8[k] = rstd * (d[k] * gam[c * 8 + k] - c1 - (t[k] - mean) * rstd * c2) + rr[k];
                *(f32x4*)(Cg + (long)m * g.ldc + part * 24 + c * 8) = pack8(o8);
            }
        }
    }
    // this tile's d(gamma) / d(beta): the RPP threads that share a channel group meet through LDS (the staged tile is dead)
    __syncthreads();
    float* red = (float*)smem;                                  // [RPP][TPR][48]
#pragma unroll
    for (int k = 0; k < 24; ++k) { red[(rsub * TPR + part) * 48 + k] = dgs[k]; red[(rsub * TPR + part) * 48 + 24 + k] = dbs[k]; }
    __syncthreads();
    for (int j = tid; j < 2 * BN; j += THREADS) {
        const int which = j / BN, c = j - which * BN, pp = c / 24, idx = c - pp * 24;
        float sum = 0.f;
        for (int rr = 0; rr < RPP; ++rr) sum += red[(rr * TPR + pp) * 48 + which * 24 + idx];
        ((float*)g.C2)[((long)by * 2 + which) * BN + c] = sum;
    }
}

#undef LAUNDER_XP

// linear workgroup id -> (N tile, M tile, K split); false = padding workgroup of the swizzled launch
__device__ __forceinline__ bool tile_coords(const GemmArgs& g, int& bx, int& by, int& bz) {
    const int L = blockIdx.x, xcd = L & 7, j = L >> 3;
    if (!g.xcd_swizzle) {
        const int nxy = g.nx * g.ny, t = L % nxy;
        bz = L / nxy; bx = t % g.nx; by = t / g.nx;
        return bz < g.nz;
    }
    if (g.nz == 1) {
        by = (j / g.nx) * 8 + xcd; bx = j % g.nx; bz = 0;
        return by < g.ny;
    }
    const int nxy = g.nx * g.ny, t = j % nxy;
    bz = (j / nxy) * 8 + xcd; bx = t % g.nx; by = t / g.nx;
    return bz < g.nz;
}

// ---- the kernel ---------------------------------------------------------------------------------------
template <typename T, typename OutT, int WM, int WN, bool TA, bool TB, bool CONV = false>
__global__ __launch_bounds__(WM * WN * 64, (sizeof(T) == 2 && WM * WN == 4) ? 2 : 1) void gemm_kernel(GemmArgs g) {
    constexpr int GEMM_THREADS = WM * WN * 64;
    constexpr int BM = WM * 64, BN = WN * 96;
    constexpr int BK = Mma<T>::BK, KSTEPS = Mma<T>::KSTEPS;
    constexpr int ES = (int)sizeof(T);
    constexpr int EPC = 16 / ES;  // elements per 16-byte chunk
    constexpr int A_STRIDE = TA ? TStride<T, BM>::value : ROWB;
    constexpr int B_STRIDE = TB ? TStride<T, BN>::value : ROWB;
    constexpr int A_BYTES = TA ? BK * A_STRIDE : BM * ROWB;
    constexpr int B_BYTES = TB ? BK * B_STRIDE : BN * ROWB;
    constexpr int NCH_A = BM * 8 / GEMM_THREADS, NCH_B = BN * 8 / GEMM_THREADS;  // 16-byte chunks per thread per K slice
    constexpr int A_CPR = TA ? BM / EPC : 8;         // chunks per image row
    constexpr int B_CPR = TB ? BN / EPC : 8;
    constexpr bool STAGED = (sizeof(OutT) == 2);      // bf16 output: LDS-staged, 16-byte coalesced epilogue
    constexpr int CS_STRIDE = BN * 2 + 16;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* As = smem;
    char* Bs = smem + A_BYTES;
    // the tile's BN bias values sit in LDS: read once per workgroup instead of 24 dependent global loads per lane in
    // the epilogue (measured: 20k of a 42k-cycle workgroup lifetime). A SEPARATE shared object, so that the compiler can
    // prove the epilogue's C-tile writes do not alias it and batch the bias reads.
    __shared__ __attribute__((aligned(16))) float bias_s[BN];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int r = lane & 31, h = lane >> 5;
    // XCD-aware tile order. Workgroups are handed to the 8 XCDs round-robin by linear id, and each XCD has its own L2:
    // tiles that read the same operand slice are therefore given ids that are congruent mod 8. Forward / input
    // gradient (nz == 1): the nx column tiles of one row block share the A rows. Weight gradient: the nx*ny tiles of
    // one K split share the same token range of dY and X (otherwise every slice is fetched by up to 8 L2s).
    int bx, by, bz;
    if (!tile_coords(g, bx, by, bz)) return;
    const int m0 = by * BM, n0 = bx * BN;
    const int kbeg = bz * g.kchunk;
    const int kend = min(g.K, kbeg + g.kchunk);

    const char* Ag = (const char*)g.A;
    const char* Bg = (const char*)g.B;

    f32x4 ra[NCH_A], rb[NCH_B];
    float rsa[TA ? NCH_A : 1];       // TA + PRO_ROWSCALE_A: DropPath factor of the token each A chunk belongs to
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    if (STAGED && tid < BN) bias_s[tid] = ((g.epi & EPI_BIAS) && n0 + tid < g.N) ? g.bias[n0 + tid] : 0.f;

    // CONV, forward layout (A = im2col view): the (t, f) position of each of this thread's A rows, decoded once.
    // CONV, weight-gradient layout (B = im2col view): the (tap, channel) of each of this thread's B column chunks.
    int cva[CONV ? (TA ? NCH_B : NCH_A) : 1], cvb[CONV ? (TA ? NCH_B : NCH_A) : 1];
    if constexpr (CONV && !TA) {
#pragma unroll
        for (int i = 0; i < NCH_A; ++i) {
            const int m = min(m0 + (tid + GEMM_THREADS * i) / A_CPR, g.M - 1);
            const int q = div_by(m, g.cv_F, g.cv_rF);
            cvb[i] = m - q * g.cv_F;                                  // f
            cva[i] = q - div_by(q, g.cv_T, g.cv_rT) * g.cv_T;         // t
        }
    }
    if constexpr (CONV && TA) {
#pragma unroll
        for (int i = 0; i < NCH_B; ++i) {
            const int n = n0 + ((tid + GEMM_THREADS * i) % B_CPR) * EPC;
            const int tap = n / g.cv_C;
            cva[i] = tap;                                             // tap (>= 9: beyond N)
            cvb[i] = n - tap * g.cv_C;                                // channel
        }
    }
    // a K slice (BK tokens) lies inside one sample when samples are whole multiples of BK tokens: one uniform factor
    const bool slice_uniform = TA && (g.rows_per_scale % BK == 0) && (g.kchunk % BK == 0);
    auto load_regs = [&](int k0) {
#pragma unroll
        for (int i = 0; i < NCH_A; ++i) {
            const int c = tid + GEMM_THREADS * i;
            const int row = c / A_CPR, cb = c % A_CPR;
            bool ok;
            long off;
            if constexpr (CONV && !TA) {
                const int k = k0 + cb * EPC, tap = k / g.cv_C, cc = k - tap * g.cv_C;
                const int dt = tap / 3 - 1, df = tap - (tap / 3) * 3 - 1;
                ok = (m0 + row < g.M) && (k < kend) && ((unsigned)(cva[i] + dt) < (unsigned)g.cv_T) &&
                     ((unsigned)(cvb[i] + df) < (unsigned)g.cv_F);
                off = ((long)(m0 + row + dt * g.cv_F + df) * g.cv_C + cc) * ES;
            } else
            if (TA) {  // image row = contraction index, chunk runs along M
                ok = (k0 + row < kend) && (m0 + cb * EPC < g.M);
                off = ((long)(k0 + row) * g.lda + m0 + cb * EPC) * ES;
                if (g.pro & PRO_ROWSCALE_A)
                    rsa[i] = g.rowscale[slice_uniform ? k0 / g.rows_per_scale : min(k0 + row, g.K - 1) / g.rows_per_scale];
            } else {
                ok = (m0 + row < g.M) && (k0 + cb * EPC < kend);
                off = ((long)(m0 + row) * g.lda + k0 + cb * EPC) * ES;
            }
            ra[i] = ok ? *(const f32x4*)(Ag + off) : zero4;
        }
#pragma unroll
        for (int i = 0; i < NCH_B; ++i) {
            const int c = tid + GEMM_THREADS * i;
            const int row = c / B_CPR, cb = c % B_CPR;
            bool ok;
            long off;
            if constexpr (CONV && TA) {
                const int tok = k0 + row, tap = cva[i];
                const int q = div_by(tok, g.cv_F, g.cv_rF);
                const int f = tok - q * g.cv_F, t = q - div_by(q, g.cv_T, g.cv_rT) * g.cv_T;
                const int dt = tap / 3 - 1, df = tap - (tap / 3) * 3 - 1;
                ok = (tok < kend) && (tap < 9) && ((unsigned)(t + dt) < (unsigned)g.cv_T) && ((unsigned)(f + df) < (unsigned)g.cv_F);
                off = ((long)(tok + dt * g.cv_F + df) * g.cv_C + cvb[i]) * ES;
            } else
            if (TB) {
                ok = (k0 + row < kend) && (n0 + cb * EPC < g.N);
                off = ((long)(k0 + row) * g.ldb + n0 + cb * EPC) * ES;
            } else {
                ok = (n0 + row < g.N) && (k0 + cb * EPC < kend);
                off = ((long)(n0 + row) * g.ldb + k0 + cb * EPC) * ES;
            }
            rb[i] = ok ? *(const f32x4*)(Bg + off) : zero4;
        }
    };
    auto store_lds = [&]() {
#pragma unroll
        for (int i = 0; i < NCH_A; ++i) {
            const int c = tid + GEMM_THREADS * i;
            const int row = c / A_CPR, cb = c % A_CPR;
            f32x4 v = ra[i];
            if (g.pro & PRO_GELU_A) v = gelu_chunk<T>(v);
            if (TA && (g.pro & PRO_ROWSCALE_A)) v = scale_chunk<T>(v, rsa[i]);
            *(f32x4*)(As + row * A_STRIDE + cb * 16) = v;
        }
#pragma unroll
        for (int i = 0; i < NCH_B; ++i) {
            const int c = tid + GEMM_THREADS * i;
            const int row = c / B_CPR, cb = c % B_CPR;
            f32x4 v = rb[i];
            if (g.pro & PRO_GELU_B) v = gelu_chunk<T>(v);
            *(f32x4*)(Bs + row * B_STRIDE + cb * 16) = v;
        }
    };

    f32x16 acc[2][3];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 3; ++ni)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.f;

    // bias gradient for free (TA): column sums of dY taken from the A fragments already in registers — lane (row, h)
    // holds 8 tokens of its column per k-step; halves and k-steps are summed at the end (no extra LDS traffic)
    float colacc[2] = {0.f, 0.f};
    const bool do_colsum = TA && g.colsum && bx == 0 && wn == 0;
    unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0, t5 = 0;
    if (g.dbg) t0 = __builtin_amdgcn_s_memtime();
    if (kbeg < kend) load_regs(kbeg);
    for (int k0 = kbeg; k0 < kend; k0 += BK) {
        __syncthreads();
        store_lds();
        if (g.dbg && k0 == kbeg) t1 = __builtin_amdgcn_s_memtime();
        __syncthreads();
        if (k0 + BK < kend) load_regs(k0 + BK);
#pragma unroll
        for (int kk = 0; kk < KSTEPS; ++kk) {
            if (k0 + kk * 16 < kend) {  // uniform: skip MFMA steps that are pure K padding
                typename Mma<T>::Frag fa[2], fb[3];
#pragma unroll
                for (int mi = 0; mi < 2; ++mi) {
                    const int base = wm * 64 + mi * 32;
                    fa[mi] = TA ? ld_frag_t<T>(As, A_STRIDE, base, kk, lane) : ld_frag_n<T>(As, base + r, kk, h);
                    if (TA && do_colsum) colacc[mi] += frag_sum<T>(fa[mi]);
                }
#pragma unroll
                for (int ni = 0; ni < 3; ++ni) {
                    const int base = wn * 96 + ni * 32;
                    fb[ni] = TB ? ld_frag_t<T>(Bs, B_STRIDE, base, kk, lane) : ld_frag_n<T>(Bs, base + r, kk, h);
                }
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                    for (int ni = 0; ni < 3; ++ni) {
                        // STAGED: compute the tile transposed (lane = output row, registers = 4-column groups)
                        // so the epilogue can move 8-byte row segments instead of single elements
                        if (STAGED) Mma<T>::mma(fb[ni], fa[mi], acc[mi][ni]);
                        else Mma<T>::mma(fa[mi], fb[ni], acc[mi][ni]);
                    }
            }
        }
    }

    if (g.dbg) t2 = t4 = __builtin_amdgcn_s_memtime();
    if (TA && do_colsum) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
            const float cs = colacc[mi] + __shfl_xor(colacc[mi], 32, 64);
            const int m = m0 + wm * 64 + mi * 32 + r;
            if (h == 0 && m < g.M) g.colsum[(long)bz * g.colsum_stride + m] = cs;
        }
    }
    OutT* Cg = (OutT*)g.C + (long)bz * g.slab_stride;
    const T* Rg = (const T*)g.resid;
    const T* Ug = (const T*)g.aux;
    if constexpr (STAGED) {
        // ---- bf16 epilogue: tile^T -> LDS (8-byte row segments) -> 16-byte coalesced rows with the fused ops ----
        // the fused-option combinations the networks use get their own straight-line instance; the rest is generic
        const int e = g.epi & ~EPI_BIAS;
        bf16_t* Cb = (bf16_t*)Cg;
        if (e == 0 && !g.rowscale) staged_epilogue<EM_PLAIN, WM, WN>(g, smem, bias_s, acc, m0, n0, Cb, t4);
        else if (e == EPI_RESID && !g.rowscale) staged_epilogue<EM_RESID, WM, WN>(g, smem, bias_s, acc, m0, n0, Cb, t4);
        else if (e == EPI_RESID && g.rows_per_scale >= 64) staged_epilogue<EM_RESID_SCALE, WM, WN>(g, smem, bias_s, acc, m0, n0, Cb, t4);
        else if (e == EPI_GELU_DUAL && !g.rowscale) staged_epilogue<EM_GELU_DUAL, WM, WN>(g, smem, bias_s, acc, m0, n0, Cb, t4);
        else if (e == EPI_MULAUX && !g.rowscale) staged_epilogue<EM_MULAUX, WM, WN>(g, smem, bias_s, acc, m0, n0, Cb, t4);
        else if (e == EPI_MULAUX && g.rows_per_scale >= 64) staged_epilogue<EM_MULAUX_SCALE, WM, WN>(g, smem, bias_s, acc, m0, n0, Cb, t4);
        else if (e == 0 && g.rows_per_scale >= 64) staged_epilogue<EM_SCALE, WM, WN>(g, smem, bias_s, acc, m0, n0, Cb, t4);
        else staged_epilogue<EM_GENERIC, WM, WN>(g, smem, bias_s, acc, m0, n0, Cb, t4);
    } else {
        // ---- f32 epilogue: C layout of the 32x32 tile is col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5) ----
#pragma unroll
        for (int ni = 0; ni < 3; ++ni) {
            const int n = n0 + wn * 96 + ni * 32 + r;
            if (n >= g.N) continue;
            if constexpr (TA) {
                // weight-gradient slabs: plain fp32 stores, no fused options (keeps this variant at 2 waves/SIMD)
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int m = m0 + wm * 64 + mi * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                        if (m < g.M) Cg[(long)m * g.ldc + n] = from_f32<OutT>(acc[mi][ni][e]);
                    }
            } else {
                const float bv = (g.epi & EPI_BIAS) ? g.bias[n] : 0.f;
#pragma unroll
                for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int m = m0 + wm * 64 + mi * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                        if (m >= g.M) continue;
                        float v = acc[mi][ni][e] + bv;
                        if (g.rowscale) v *= g.rowscale[m / g.rows_per_scale];
                        if (g.epi & EPI_MULGELUGRAD) v *= gelu_grad_f(to_f32<T>(Ug[(long)m * g.ldaux + n]));
                        if (g.epi & EPI_MULAUX) v *= to_f32<T>(Ug[(long)m * g.ldaux + n]);
                        if (g.epi & EPI_RESID) v += to_f32<T>(Rg[(long)m * g.ldr + n]);
                        if (g.epi & EPI_GELU_DUAL) {
                            float dv;
                            gelu_both(v, v, dv);
                            ((OutT*)g.C2)[(long)m * g.ldc + n] = from_f32<OutT>(dv);
                        }
                        OutT* dst = Cg + (long)m * g.ldc + n;
                        if (g.epi & EPI_ACCUM) v += to_f32<OutT>(*dst);
                        *dst = from_f32<OutT>(v);
                    }
                }
            }
        }
    }
    if (g.dbg) {
        t5 = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        t3 = __builtin_amdgcn_s_memtime();
        if (tid == 0) {
            const long b = ((long)bz * g.ny + by) * g.nx + bx;
            g.dbg[b * 6 + 0] = t0; g.dbg[b * 6 + 1] = t1; g.dbg[b * 6 + 2] = t2; g.dbg[b * 6 + 3] = t3; g.dbg[b * 6 + 4] = t4; g.dbg[b * 6 + 5] = t5;
        }
    }
}

// ---- forward GEMM (both operands k-contiguous, bf16) fed by LDS-DMA --------------------------------------------------
// Same tiles, fragment maps and epilogue as gemm_kernel; operands reach LDS by global_load_lds_dwordx4: no VGPR
// staging (165 registers instead of 200) and no ds_write, two 32-element K slices in LDS (20 KB each; the C-tile
// staging of the epilogue is the larger user), so THREE workgroups fit per CU instead of two — that occupancy, not
// the DMA itself, is what pays (a 3-slice / 2-workgroup version measured equal to the register-staged kernel;
// this one is 7 % faster over the HTS-AT forward shapes: tools/gemm_shapes.py, one session, 4.06 -> 3.77 ms).
// STAGES > 2 turns the two slices into a ring (STAGES - 1 slices in flight, vmcnt distance kept constant by dummy
// loads in the tail). Measured (tools/gemm_deep.py, tools/experiments/loop_ceiling.hip): ring depth 2..6 changes nothing and
// a 256x192 / 8-wave / 4-slice variant is no faster — the slice loop is bound by the THROUGHPUT of global_load_lds
// (about 30 B/clk/CU even on cache hits: 1.37 PFLOP/s for this tile with fragment reads, barrier and MFMAs in place,
// 2.1 PFLOP/s without the DMA), not by its latency, the LDS reads or the barrier. Only STAGES = 2 is instantiated.
// One raw s_barrier per slice; the next slice's DMA is issued right after it. Images are lane-linear 64-byte rows; the
// 16-byte chunk each lane FETCHES is XOR-ed with (row>>2)&3 so every ds_read_b128 lane group is bank-conflict free.
// PSELD_GEMM_DMA=0 falls back to gemm_kernel (A/B knob).
typedef __attribute__((address_space(3))) void* lds_void_ptr;
typedef const __attribute__((address_space(1))) void* gbl_void_ptr;

template <int WM, int WN, int STAGES = 2, bool LNBWD = false>
__global__ __launch_bounds__(WM * WN * 64, STAGES == 2 ? 3 : 2) void gemm_dma_kernel(GemmArgs g) {
    constexpr int THREADS = WM * WN * 64, WAVES = WM * WN, BM = WM * 64, BN = WN * 96;
    constexpr int BKD = 32;
    constexpr int A_BYTES = BM * 64, B_BYTES = BN * 64, STAGE = A_BYTES + B_BYTES;
    constexpr int A_INSTR = A_BYTES / 1024, TOTAL = STAGE / 1024;
    constexpr int LPW = (TOTAL + WAVES - 1) / WAVES;          // DMA instructions per wave per slice (padded with dummies)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ __attribute__((aligned(16))) float bias_s[BN];
    char* dummy = smem + STAGES * STAGE;                     // 1 KiB sink for the padding instructions
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN, r = lane & 31, h = lane >> 5;
    int bx, by, bz;
    if (!tile_coords(g, bx, by, bz)) return;
    const int m0 = by * BM, n0 = bx * BN;
    if (tid < BN) bias_s[tid] = (((g.epi & EPI_BIAS) || LNBWD) && n0 + tid < g.N) ? g.bias[n0 + tid] : 0.f;   // (LNBWD: the LayerNorm's gamma)
    const bf16_t* Ag = (const bf16_t*)g.A;
    const bf16_t* Bg = (const bf16_t*)g.B;
    const int nslices = g.K / BKD;

    auto issue = [&](int s) {
        char* st = smem + (s % STAGES) * STAGE;
        const int k0 = s * BKD;
#pragma unroll
        for (int j = 0; j < LPW; ++j) {
            const int i = wave + WAVES * j;                  // wave-uniform instruction index inside the slice
            if (i < TOTAL && s < nslices) {                  // (deep pipeline: slices past the end are dummies, so that
                                                             //  the vmcnt distance stays constant in the tail)
                const bool isA = i < A_INSTR;
                const int ii = isA ? i : i - A_INSTR;
                const int row = ii * 16 + (lane >> 2), chunk = (lane & 3) ^ ((row >> 2) & 3);
                const bf16_t* src = isA ? Ag + (long)min(m0 + row, g.M - 1) * g.lda + k0 + chunk * 8
                                        : Bg + (long)min(n0 + row, g.N - 1) * g.ldb + k0 + chunk * 8;
                if constexpr (STAGES > 2) {
                    // ring: inline asm, so that hipcc does not drain the slices in flight with vmcnt(0) before the next ds_read
                    const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long)(lds_void_ptr)(st + i * 1024));
                    unsigned keep;
                    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                                 : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
                } else {
                    __builtin_amdgcn_global_load_lds((gbl_void_ptr)src, (lds_void_ptr)(st + i * 1024), 16, 0, 0);
                }
            } else {
                if constexpr (STAGES > 2) {
                    const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long)(lds_void_ptr)dummy);
                    unsigned keep;
                    const bf16_t* src = Ag + lane * 8;
                    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                                 : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
                } else {
                    __builtin_amdgcn_global_load_lds((gbl_void_ptr)(Ag + lane * 8), (lds_void_ptr)dummy, 16, 0, 0);
                }
            }
        }
    };
    auto frag = [&](const char* img, int row, int kk) -> bf16x8 {
        return *(const bf16x8*)(img + row * 64 + (((2 * kk + h) ^ ((row >> 2) & 3)) << 4));
    };

    f32x16 acc[2][3];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 3; ++ni)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.f;

    // STAGES - 1 slices are in flight; the wait leaves the STAGES - 2 younger ones outstanding
#pragma unroll
    for (int p = 0; p < STAGES - 1; ++p) issue(p);
    for (int s = 0; s < nslices; ++s) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((STAGES - 2) * LPW) : "memory");
        __builtin_amdgcn_s_barrier();
        if (STAGES > 2 || s + 1 < nslices) issue(s + STAGES - 1);
        const char* As = smem + (s % STAGES) * STAGE;
        const char* Bs = As + A_BYTES;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 fa[2], fb[3];
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) fa[mi] = frag(As, wm * 64 + mi * 32 + r, kk);
#pragma unroll
            for (int ni = 0; ni < 3; ++ni) fb[ni] = frag(Bs, wn * 96 + ni * 32 + r, kk);
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 3; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[ni], fa[mi], acc[mi][ni], 0, 0, 0);
        }
    }
    if constexpr (LNBWD) {
        lnbwd_epilogue<WM, WN>(g, smem, bias_s, acc, m0, by);
        return;
    }
    unsigned long long t4 = 0;
    bf16_t* Cb = (bf16_t*)g.C;
    const int e = g.epi & ~EPI_BIAS;
    if (e == 0 && !g.rowscale) staged_epilogue<EM_PLAIN, WM, WN>(g, smem, bias_s, acc, m0, n0, Cb, t4);
    else if (e == EPI_RESID && !g.rowscale) staged_epilogue<EM_RESID, WM, WN>(g, smem, bias_s, acc, m0, n0, Cb, t4);
    else if (e == EPI_RESID && g.rows_per_scale >= 64) staged_epilogue<EM_RESID_SCALE, WM, WN>(g, smem, bias_s, acc, m0, n0, Cb, t4);
    else if (e == EPI_GELU_DUAL && !g.rowscale) staged_epilogue<EM_GELU_DUAL, WM, WN>(g, smem, bias_s, acc, m0, n0, Cb, t4);
    else if (e == EPI_MULAUX && !g.rowscale) staged_epilogue<EM_MULAUX, WM, WN>(g, smem, bias_s, acc, m0, n0, Cb, t4);
    else if (e == EPI_MULAUX && g.rows_per_scale >= 64) staged_epilogue<EM_MULAUX_SCALE, WM, WN>(g, smem, bias_s, acc, m0, n0, Cb, t4);
    else if (e == 0 && g.rows_per_scale >= 64) staged_epilogue<EM_SCALE, WM, WN>(g, smem, bias_s, acc, m0, n0, Cb, t4);
    else staged_epilogue<EM_GENERIC, WM, WN>(g, smem, bias_s, acc, m0, n0, Cb, t4);
}

template <int WM, int WN, int STAGES = 2>
int launch_gemm_dma(const GemmArgs& g, hipStream_t stream) {
    constexpr int BM = WM * 64, BN = WN * 96;
    constexpr int STAGE = (BM + BN) * 64;
    constexpr int CS_BYTES = BM * (BN * 2 + 16);
    constexpr int LDS = (STAGES * STAGE + 1024 > CS_BYTES) ? STAGES * STAGE + 1024 : CS_BYTES;
    GemmArgs ga = g;
    ga.nx = pseld_cdiv(g.N, BN); ga.ny = pseld_cdiv(g.M, BM); ga.nz = 1;
    ga.xcd_swizzle = 1;
    const long nblocks = (long)8 * pseld_cdiv(ga.ny, 8) * ga.nx;
    static bool attr_set = false;
    if (!attr_set) { (void)hipFuncSetAttribute((const void*)gemm_dma_kernel<WM, WN, STAGES>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS); attr_set = true; }
    // (the names rocprofv3 prints: the LNBWD template flag is part of the symbol)
    g_last_gemm_kernel = (WM == 2 && STAGES == 2) ? "gemm_dma_kernel<2, 2, 2, false>" : (WM == 4 ? "gemm_dma_kernel<4, 1, 2, false>" : "gemm_dma_kernel<2, 2, 3, false>");
    hipLaunchKernelGGL((gemm_dma_kernel<WM, WN, STAGES>), dim3((unsigned)nblocks), dim3(WM * WN * 64), LDS, stream, ga);
    PSELD_LAUNCH_CHECK("gemm_dma");
    return PSELD_OK;
}


// ---- forward / input-gradient GEMM with a long contraction: 256 x 192 tile, 8 waves, 4-stage LDS-DMA ring -----------------------
// Same images, fragment maps and staged epilogue as gemm_dma_kernel, but ONE workgroup of 8 waves per CU on a 256 x 192 tile
// (110 flop per LDS-DMA byte instead of 77: the loop of the 128 x 192 kernel is bound by the ~28 B/clk a CU's vector-memory
// path moves into LDS, tools/gemm_ab.py) with three 28 KB slices in flight (inline-asm DMA + counted vmcnt: with the builtin
// hipcc drains the ring before every slice's ds_reads). The epilogue of a tile is not overlapped with anything (one workgroup
// per CU): measured against the 128 x 192 kernel (tools/gemm_ab.py, in-process) it wins 9-10 % at K >= 3072 and loses 8-15 % at
// K <= 1536, so it only takes K >= 3072 (dispatch in pseld_gemm; PSELD_GEMM_FWD_RING=<K threshold>).
__global__ __launch_bounds__(512, 2) void gemm_fwd_ring_kernel(GemmArgs g) {
    constexpr int WM = 4, WN = 2, WAVES = 8, BM = 256, BN = 192;
    constexpr int BKD = 32, STAGES = 4;
    constexpr int A_BYTES = BM * 64, B_BYTES = BN * 64, STAGE = A_BYTES + B_BYTES;
    constexpr int A_INSTR = A_BYTES / 1024, TOTAL = STAGE / 1024;              // 16 + 12 DMA instructions per slice
    constexpr int LPW_HI = (TOTAL + WAVES - 1) / WAVES, LPW_LO = TOTAL / WAVES, NHI = TOTAL - LPW_LO * WAVES;   // 4, 3, waves 0-3
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ __attribute__((aligned(16))) float bias_s[BN];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN, r = lane & 31, h = lane >> 5;
    int bx, by, bz;
    if (!tile_coords(g, bx, by, bz)) return;
    const int m0 = by * BM, n0 = bx * BN;
    if (tid < BN) bias_s[tid] = ((g.epi & EPI_BIAS) && n0 + tid < g.N) ? g.bias[n0 + tid] : 0.f;
    const bf16_t* Ag = (const bf16_t*)g.A;
    const bf16_t* Bg = (const bf16_t*)g.B;
    const int nslices = g.K / BKD;
    const bool hi = wave < NHI;

    const bf16_t* srcp[LPW_HI];                   // this wave's DMA rows (slice 0), computed once
#pragma unroll
    for (int j = 0; j < LPW_HI; ++j) {
        const int i = min(wave + WAVES * j, TOTAL - 1);
        const bool isA = i < A_INSTR;
        const int ii = isA ? i : i - A_INSTR;
        const int row = ii * 16 + (lane >> 2), chunk = (lane & 3) ^ ((row >> 2) & 3);
        srcp[j] = isA ? Ag + (long)min(m0 + row, g.M - 1) * g.lda + chunk * 8 : Bg + (long)min(n0 + row, g.N - 1) * g.ldb + chunk * 8;
    }
    auto issue = [&](int s) {                     // slices past the end re-read slice 0 into a slot nobody reads: constant vmcnt distance
        char* st = smem + (s % STAGES) * STAGE;
        const int k0 = (s < nslices ? s : 0) * BKD;
#pragma unroll
        for (int j = 0; j < LPW_HI; ++j) {
            if (j < LPW_LO || hi) {
                const int i = wave + WAVES * j;
                const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long)(lds_void_ptr)(st + i * 1024));
                const bf16_t* src = srcp[j] + k0;
                unsigned keep;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
            }
        }
    };
    auto frag = [&](const char* img, int row, int kk) -> bf16x8 {
        return *(const bf16x8*)(img + row * 64 + (((2 * kk + h) ^ ((row >> 2) & 3)) << 4));
    };

    f32x16 acc[2][3];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 3; ++ni)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.f;

#pragma unroll
    for (int p = 0; p < STAGES - 1; ++p) issue(p);
    for (int s = 0; s < nslices; ++s) {
        if (hi) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((STAGES - 2) * LPW_HI) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((STAGES - 2) * LPW_LO) : "memory");
        __builtin_amdgcn_s_barrier();
        issue(s + STAGES - 1);
        const char* As = smem + (s % STAGES) * STAGE;
        const char* Bs = As + A_BYTES;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 fa[2], fb[3];
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) fa[mi] = frag(As, wm * 64 + mi * 32 + r, kk);
#pragma unroll
            for (int ni = 0; ni < 3; ++ni) fb[ni] = frag(Bs, wn * 96 + ni * 32 + r, kk);
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 3; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[ni], fa[mi], acc[mi][ni], 0, 0, 0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the tail's surplus DMAs land before the C tile is staged over the ring
    unsigned long long t4 = 0;
    bf16_t* Cb = (bf16_t*)g.C;
    const int e = g.epi & ~EPI_BIAS;
    if (e == 0 && !g.rowscale) staged_epilogue<EM_PLAIN, WM, WN>(g, smem, bias_s, acc, m0, n0, Cb, t4);
    else if (e == EPI_RESID && !g.rowscale) staged_epilogue<EM_RESID, WM, WN>(g, smem, bias_s, acc, m0, n0, Cb, t4);
    else if (e == EPI_RESID && g.rows_per_scale >= 64) staged_epilogue<EM_RESID_SCALE, WM, WN>(g, smem, bias_s, acc, m0, n0, Cb, t4);
    else if (e == EPI_GELU_DUAL && !g.rowscale) staged_epilogue<EM_GELU_DUAL, WM, WN>(g, smem, bias_s, acc, m0, n0, Cb, t4);
    else if (e == EPI_MULAUX && !g.rowscale) staged_epilogue<EM_MULAUX, WM, WN>(g, smem, bias_s, acc, m0, n0, Cb, t4);
    else if (e == EPI_MULAUX && g.rows_per_scale >= 64) staged_epilogue<EM_MULAUX_SCALE, WM, WN>(g, smem, bias_s, acc, m0, n0, Cb, t4);
    else if (e == 0 && g.rows_per_scale >= 64) staged_epilogue<EM_SCALE, WM, WN>(g, smem, bias_s, acc, m0, n0, Cb, t4);
    else staged_epilogue<EM_GENERIC, WM, WN>(g, smem, bias_s, acc, m0, n0, Cb, t4);
}

static int launch_gemm_fwd_ring(const GemmArgs& g, hipStream_t stream) {
    constexpr int BM = 256, BN = 192;
    constexpr int LDS = 4 * (BM + BN) * 64;               // 112 KB ring; the 256 x (192 * 2 + 16) C staging (100 KB) re-uses it
    GemmArgs ga = g;
    ga.nx = pseld_cdiv(g.N, BN); ga.ny = pseld_cdiv(g.M, BM); ga.nz = 1;
    ga.xcd_swizzle = 1;
    const long nblocks = (long)8 * pseld_cdiv(ga.ny, 8) * ga.nx;
    static bool attr_set = false;
    if (!attr_set) { (void)hipFuncSetAttribute((const void*)gemm_fwd_ring_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS); attr_set = true; }
    g_last_gemm_kernel = "gemm_fwd_ring_kernel";
    hipLaunchKernelGGL(gemm_fwd_ring_kernel, dim3((unsigned)nblocks), dim3(512), LDS, stream, ga);
    PSELD_LAUNCH_CHECK("gemm_fwd_ring");
    return PSELD_OK;
}

// PSELD_GEMM_XCD: bit 0 = swizzle forward / input-gradient launches, bit 1 = swizzle weight-gradient launches
static int gemm_xcd_mode(bool wgrad) {
    const int v = pseld_knob(KNOB_GEMM_XCD, 3);
    return wgrad ? (v >> 1) & 1 : v & 1;
}

// ---- weight gradient fed by a deep LDS-DMA ring (bf16, large matrices) ---------------------------------------------------
// dW[N_out, K_in] (one fp32 slab per token split) = dY^T X with BOTH operands token-major, i.e. k-major LDS images: a slice is
// 32 token rows of dY (BM columns) and of X (BN columns), written lane-linear by the DMA and read back as MFMA operands with
// ds_read_b64_tr_b16. Why a second weight-gradient kernel: s_memtime stamps on the register-staged one (tools/gemm_stamps.py,
// stage-2 fc1: 4.0k cycles per 64-token slice against 1.5k of MFMA time) show a loop that waits for memory with ONE slice in
// flight per CU, and the A/B of tools/gemm_ab.py shows what bounds an LDS-fed loop on this chip once latency is covered: the
// CU's vector-memory path moves about 28 B/clk into LDS wherever the DMA instructions are placed, so the tile must be large
// enough in flop per operand byte. Here: ONE workgroup of 8 waves per CU, a 384 x 192 tile (wave tile 96 x 96 = 3 x 3 MFMA
// tiles, 128 flop per LDS-DMA byte), 32-token slices of 36 KB in a 4-stage ring (three slices = 108 KB in flight per CU,
// counted vmcnt, one raw s_barrier per slice), no staging registers and no ds_write.
// The 16-byte chunk a lane fetches is ROTATED inside its row by 4 * (k & 3) chunks so that the four token rows one transposed
// read touches fall in four different 64-byte bank groups (row lengths 768 and 384 bytes are multiples of 128).
// DropPath (PRO_ROWSCALE_A): a slice never straddles samples (rows_per_scale % 32 == 0), its factor is wave-uniform: slices of
// dropped samples (factor 0) are skipped by the MFMAs, other factors scale the dY fragments.
template <int MT, int NT>
__global__ __launch_bounds__(512, 2) void gemm_wgrad_ring_kernel(GemmArgs g) {
    constexpr int WM = 4, WN = 2, WAVES = 8, BM = WM * MT * 32, BN = WN * NT * 32;
    constexpr int BKD = 32, STAGES = 4;
    constexpr int RA = BM * 2, RB = BN * 2, CA = BM / 8, CB = BN / 8;        // row bytes / 16-byte chunks per row
    constexpr int ROTA = ((RA / 64) & 1) ? 0 : 4, ROTB = ((RB / 64) & 1) ? 0 : 4;
    constexpr int A_BYTES = BKD * RA, B_BYTES = BKD * RB, STAGE = A_BYTES + B_BYTES;
    constexpr int A_INSTR = A_BYTES / 1024, TOTAL = STAGE / 1024;
    constexpr int LPW_HI = (TOTAL + WAVES - 1) / WAVES, LPW_LO = TOTAL / WAVES;   // DMA instructions per slice: waves < NHI issue LPW_HI
    constexpr int NHI = TOTAL - LPW_LO * WAVES;                                   // (0: every wave issues LPW_LO)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN, r = lane & 31, h = lane >> 5;
    int bx, by, bz;
    if (!tile_coords(g, bx, by, bz)) return;
    const int m0 = by * BM, n0 = bx * BN;
    const int kbeg = bz * g.kchunk, kend = min(g.K, kbeg + g.kchunk);
    const int nslices = (kend - kbeg) / BKD;
    const bf16_t* Ag = (const bf16_t*)g.A;
    const bf16_t* Bg = (const bf16_t*)g.B;
    const bool hi = NHI > 0 && wave < NHI;

    // per-lane source offsets of this wave's DMA instructions (slice-independent part), computed once
    long srcoff[LPW_HI];
    bool srcA[LPW_HI];
#pragma unroll
    for (int j = 0; j < LPW_HI; ++j) {
        const int i = wave + WAVES * j;
        const bool isA = i < A_INSTR;
        const int off = (isA ? i : i - A_INSTR) * 1024 + lane * 16;
        const int R = isA ? RA : RB, C = isA ? CA : CB, ROT = isA ? ROTA : ROTB;
        const int krow = off / R, pc = (off - krow * R) >> 4;
        int c = pc - ROT * (krow & 3);
        c += c < 0 ? C : 0;
        srcA[j] = isA;
        srcoff[j] = isA ? (long)krow * g.lda + min(m0 + c * 8, g.M - 8) : (long)krow * g.ldb + min(n0 + c * 8, g.N - 8);
    }
    auto issue = [&](int s) {                 // slices past the end are issued too (into their ring slot, from slice 0's rows):
        char* st = smem + (s % STAGES) * STAGE;   // the vmcnt distance stays constant in the tail and nobody reads them
        const long k0 = kbeg + (long)(s < nslices ? s : 0) * BKD;
#pragma unroll
        for (int j = 0; j < LPW_HI; ++j) {
            const int i = wave + WAVES * j;
            if (j < LPW_LO || hi) {
                const bf16_t* src = srcA[j] ? Ag + k0 * g.lda + srcoff[j] : Bg + k0 * g.ldb + srcoff[j];
                // inline asm, not __builtin_amdgcn_global_load_lds: hipcc cannot prove that the ring slot being filled and the
                // slot the next ds_read touches differ, and drains the ring with s_waitcnt vmcnt(0) before every slice's reads
                const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long)(lds_void_ptr)(st + i * 1024));
                unsigned keep;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
            }
        }
    };
    // MFMA operand for image columns col0 + (lane&31), k-step kk: two transposed 8-byte reads (k rows q and q + 4)
    auto frag_t = [&](const char* img, int R, int C, int ROT, int col0, int kk) -> bf16x8 {
        const int i = lane & 15, q = i >> 2, p = i & 3, gsel = (lane >> 4) & 1;
        const int col = col0 + 16 * gsel + 4 * p;
        const int krow = kk * 16 + 8 * h + q;
        int pc = (col >> 3) + ROT * (krow & 3);
        pc -= pc >= C ? C : 0;
        const char* addr = img + krow * R + pc * 16 + (col & 7) * 2;
        const short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_ptr)(addr));
        const short4v hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_ptr)(addr + 4 * R));
        typedef __attribute__((ext_vector_type(8))) short short8v;
        short8v v;
        v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3]; v[4] = hi4[0]; v[5] = hi4[1]; v[6] = hi4[2]; v[7] = hi4[3];
        return __builtin_bit_cast(bf16x8, v);
    };

    f32x16 acc[MT][NT];
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
#pragma unroll
        for (int ni = 0; ni < NT; ++ni)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.f;
    float colacc[MT];
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) colacc[mi] = 0.f;
    const bool do_colsum = g.colsum && bx == 0 && wn == 0;
    const bool scaled = (g.pro & PRO_ROWSCALE_A) != 0;

    const long dbgi = (((long)bz * g.ny + by) * g.nx + bx) * 6;
    if (g.dbg && tid == 0) g.dbg[dbgi + 0] = __builtin_amdgcn_s_memtime();
    // One phase per slice: retire the oldest slice (counted vmcnt: the two younger ones stay in flight), barrier, refill the
    // slot everybody has just finished reading, then fragments + MFMAs. (A two-group ping-pong version of this loop - one wave
    // of each SIMD computing while its partner loads - measured slower: tools/experiments/gemm_round2_experiments.hip.txt.)
#pragma unroll
    for (int p = 0; p < STAGES - 1; ++p) issue(p);
    for (int s = 0; s < nslices; ++s) {
        if (hi) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((STAGES - 2) * LPW_HI) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((STAGES - 2) * LPW_LO) : "memory");
        __builtin_amdgcn_s_barrier();
        if (g.dbg && tid == 0 && (s == 0 || s == 8)) g.dbg[dbgi + (s == 0 ? 1 : 4)] = __builtin_amdgcn_s_memtime();
        issue(s + STAGES - 1);
        const char* As = smem + (s % STAGES) * STAGE;
        const char* Bs = As + A_BYTES;
        // DropPath factor of this slice's sample through the SCALAR cache (inline asm: hipcc would fetch it with a VMEM load -
        // the pointer is not provably read-only - and then drain the whole DMA ring with vmcnt(0) before the first use)
        float sc = 1.f;
        if (scaled) {
            const float* sp = g.rowscale + __builtin_amdgcn_readfirstlane((kbeg + s * BKD) / g.rows_per_scale);
            asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(sc) : "s"(sp) : "memory");
        }
        if (sc == 0.f) continue;        // uniform: a dropped sample contributes nothing
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 fa[MT], fb[NT];
#pragma unroll
            for (int mi = 0; mi < MT; ++mi) {
                fa[mi] = frag_t(As, RA, CA, ROTA, (wm * MT + mi) * 32, kk);
                if (scaled && sc != 1.f) fa[mi] = __builtin_bit_cast(bf16x8, scale_chunk<bf16_t>(__builtin_bit_cast(f32x4, fa[mi]), sc));
                if (do_colsum) colacc[mi] += frag_sum<bf16_t>(fa[mi]);
            }
#pragma unroll
            for (int ni = 0; ni < NT; ++ni) fb[ni] = frag_t(Bs, RB, CB, ROTB, (wn * NT + ni) * 32, kk);
#pragma unroll
            for (int mi = 0; mi < MT; ++mi)
#pragma unroll
                for (int ni = 0; ni < NT; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[mi], fb[ni], acc[mi][ni], 0, 0, 0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the tail's surplus DMAs must not outlive the workgroup's LDS
    if (g.dbg && tid == 0) g.dbg[dbgi + 2] = __builtin_amdgcn_s_memtime();
    if (do_colsum) {
#pragma unroll
        for (int mi = 0; mi < MT; ++mi) {
            const float cs = colacc[mi] + __shfl_xor(colacc[mi], 32, 64);
            const int m = m0 + (wm * MT + mi) * 32 + r;
            if (h == 0 && m < g.M) g.colsum[(long)bz * g.colsum_stride + m] = cs;
        }
    }
    float* Cg = (float*)g.C + (long)bz * g.slab_stride;
#pragma unroll
    for (int ni = 0; ni < NT; ++ni) {
        const int n = n0 + (wn * NT + ni) * 32 + r;
        if (n >= g.N) continue;
#pragma unroll
        for (int mi = 0; mi < MT; ++mi)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + (wm * MT + mi) * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (m < g.M) Cg[(long)m * g.ldc + n] = acc[mi][ni][e];
            }
    }
    if (g.dbg) {
        const unsigned long long t5 = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (tid == 0) { g.dbg[dbgi + 5] = t5; g.dbg[dbgi + 3] = __builtin_amdgcn_s_memtime(); }
    }
}

template <int MT, int NT>
int launch_wgrad_ring(const GemmArgs& g, int splits, hipStream_t stream) {
    constexpr int BM = 4 * MT * 32, BN = 2 * NT * 32;
    constexpr int LDS = 4 * 32 * (BM + BN) * 2;
    GemmArgs ga = g;
    ga.nx = pseld_cdiv(g.N, BN); ga.ny = pseld_cdiv(g.M, BM); ga.nz = splits;
    ga.xcd_swizzle = gemm_xcd_mode(true) && (splits == 1 || splits % 8 == 0);
    const long nblocks = !ga.xcd_swizzle ? (long)ga.nx * ga.ny * splits
                         : splits == 1 ? (long)8 * pseld_cdiv(ga.ny, 8) * ga.nx : (long)8 * pseld_cdiv(splits, 8) * ga.nx * ga.ny;
    static bool attr_set = false;
    if (!attr_set) { (void)hipFuncSetAttribute((const void*)gemm_wgrad_ring_kernel<MT, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS); attr_set = true; }
    g_last_gemm_kernel = MT == 3 ? "gemm_wgrad_ring_kernel<3, 3>" : "gemm_wgrad_ring_kernel<2, 3>";
    hipLaunchKernelGGL((gemm_wgrad_ring_kernel<MT, NT>), dim3((unsigned)nblocks), dim3(512), LDS, stream, ga);
    PSELD_LAUNCH_CHECK("gemm_wgrad_ring");
    return PSELD_OK;
}


template <typename T, typename OutT, int WM, int WN, bool TA, bool TB, bool CONV = false>
int launch_gemm(const GemmArgs& g, int splits, hipStream_t stream) {
    constexpr int BM = WM * 64, BN = WN * 96, BK = Mma<T>::BK;
    constexpr int A_STRIDE = TA ? TStride<T, BM>::value : ROWB;
    constexpr int B_STRIDE = TB ? TStride<T, BN>::value : ROWB;
    constexpr int A_BYTES = TA ? BK * A_STRIDE : BM * ROWB;
    constexpr int B_BYTES = TB ? BK * B_STRIDE : BN * ROWB;
    constexpr int CS_BYTES = (sizeof(OutT) == 2) ? BM * (BN * 2 + 16) : 0;
    constexpr int LDS = (A_BYTES + B_BYTES > CS_BYTES) ? (A_BYTES + B_BYTES) : CS_BYTES;
    GemmArgs ga = g;
    ga.nx = pseld_cdiv(g.N, BN); ga.ny = pseld_cdiv(g.M, BM); ga.nz = splits;
    // split launches are swizzled only when every XCD gets the same number of whole splits
    ga.xcd_swizzle = gemm_xcd_mode(TA) && (splits == 1 || splits % 8 == 0);
    const long nblocks = !ga.xcd_swizzle ? (long)ga.nx * ga.ny * splits
                         : splits == 1 ? (long)8 * pseld_cdiv(ga.ny, 8) * ga.nx : (long)8 * pseld_cdiv(splits, 8) * ga.nx * ga.ny;
    dim3 grid((unsigned)nblocks);
    if (LDS > 64 * 1024) {
        static bool attr_set = false;
        if (!attr_set) { (void)hipFuncSetAttribute((const void*)gemm_kernel<T, OutT, WM, WN, TA, TB, CONV>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS); attr_set = true; }
    }
    g_last_gemm_kernel = TA ? (WM == 4 && WN == 2 ? "gemm_kernel<.., 4, 2, TA>" : (WM == 4 ? "gemm_kernel<.., 4, 1, TA>" : "gemm_kernel<.., 2, 2, TA>"))
                            : (WM == 4 ? "gemm_kernel<.., 4, 1>" : "gemm_kernel<.., 2, 2>");
    hipLaunchKernelGGL((gemm_kernel<T, OutT, WM, WN, TA, TB, CONV>), grid, dim3(WM * WN * 64), LDS, stream, ga);
    PSELD_LAUNCH_CHECK("gemm");
    return PSELD_OK;
}

static int gemm_big_tiles() {
    return pseld_knob(KNOB_GEMM_BIG, 1);
}

template <typename T, typename OutT, bool TA, bool TB, bool CONV = false>
int dispatch_tile(const GemmArgs& g, int splits, hipStream_t stream) {
    // 256x96 when one 96-column tile covers N (or N is not worth a 192 tile), else 128x192
    if (g.N <= 96 || (g.N % 192 != 0 && g.N % 96 == 0 && g.N <= 288))
        return launch_gemm<T, OutT, 4, 1, TA, TB, CONV>(g, splits, stream);
    if (sizeof(T) == 2 && TA && g.M >= 256) {
        const int force = pseld_knob(KNOB_WGRAD_TILE, 0);          // experiment knob: 22 / 42
        if (force == 42 || (force == 0 && gemm_big_tiles()))
            return launch_gemm<T, OutT, 4, 2, TA, TB, CONV>(g, splits, stream);   // weight gradient: 256x192, 8 waves (measured +4 %)
    }
    return launch_gemm<T, OutT, 2, 2, TA, TB, CONV>(g, splits, stream);
}

// ---- split-K slab reduction: out[i] (+)= sum_s slabs[s][i] ----------------------------------------------
__global__ void splitk_reduce_kernel(const float* __restrict__ slabs, float* __restrict__ out, long n, int splits,
                                     long slab_stride, int accumulate) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s = accumulate ? out[i] : 0.f;
    for (int z = 0; z < splits; ++z) s += slabs[z * slab_stride + i];
    out[i] = s;
}

// ---- column sums (bias gradients): out[n] = sum_m X[m, n] * rowscale ------------------------------------
template <typename T>
__global__ void colsum_partial_kernel(const T* __restrict__ X, float* __restrict__ part, int M, int N, int ld,
                                      int rows_per_block) {
    // block handles rows [blockIdx.y*rows_per_block, ...), 64 columns starting at blockIdx.x*64
    __shared__ float red[4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int n = blockIdx.x * 64 + lane;
    const int mbeg = blockIdx.y * rows_per_block;
    const int mend = min(M, mbeg + rows_per_block);
    float s = 0.f;
    if (n < N)
        for (int m = mbeg + w; m < mend; m += 4) s += to_f32<T>(X[(long)m * ld + n]);
    red[w][lane] = s;
    __syncthreads();
    if (w == 0 && n < N) part[(long)blockIdx.y * N + n] = red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane];
}

// the same with 16-byte loads: a thread owns 8 consecutive columns, 256 / (N/8) row lanes stride the block's rows and are
// combined through LDS (the scalar kernel above moves 2 bytes per lane: 1 TB/s on the bias gradients of a frozen backbone)
template <typename T>
__global__ __launch_bounds__(256) void colsum_vec_kernel(const T* __restrict__ X, float* __restrict__ part, int M, int N, int ld,
                                                         int rows_per_block) {
    __shared__ float red[256][8];
    const int c8 = N >> 3, cw = c8 < 256 ? c8 : 256, lanes = 256 / cw;
    const int cl = threadIdx.x % cw, rl = threadIdx.x / cw;
    const int co = blockIdx.x * cw + cl;
    const int mbeg = blockIdx.y * rows_per_block, mend = min(M, mbeg + rows_per_block);
    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (co < c8 && rl < lanes)
        for (int m = mbeg + rl; m < mend; m += lanes) {
            float v[8];
            load8<T>(X + (long)m * ld + co * 8, v);
#pragma unroll
            for (int k = 0; k < 8; ++k) a[k] += v[k];
        }
#pragma unroll
    for (int k = 0; k < 8; ++k) red[threadIdx.x][k] = a[k];
    __syncthreads();
    if (rl == 0 && co < c8) {
        for (int i = 1; i < lanes; ++i)
#pragma unroll
            for (int k = 0; k < 8; ++k) a[k] += red[i * cw + cl][k];
#pragma unroll
        for (int k = 0; k < 8; ++k) part[(long)blockIdx.y * N + co * 8 + k] = a[k];
    }
}

// ---- skinny forward GEMM: C[M <= 64, N] = A[M, K] B[N, K]^T (+ bias) (+ resid), bf16 ------------------------------------------------
// The recurrent products of the GRU decoder (48 rows x 3072 x 1024, a thousand of them per step) put ONE row tile = 16
// workgroups on the chip with the tiled kernels. Here a workgroup owns 32 output columns (N / 32 workgroups), its four waves
// split K, and the MFMA operands are read straight from global memory (a lane's fragment is 8 consecutive k of one row =
// one 16-byte load; B rows are the weight rows, A rows the few activations rows, which every workgroup re-reads from L2);
// the four partial tiles meet in LDS. C^T tiles are computed (weight rows in registers, activation rows in lanes).
template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void gemm_skinny_kernel(GemmArgs g) {
    __shared__ float red[WAVES - 1][2][16][64];
    const bf16_t* A = (const bf16_t*)g.A;        // [M, lda]
    const bf16_t* B = (const bf16_t*)g.B;        // [N, ldb]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, h2 = lane >> 5;
    const int n0 = blockIdx.x * 32;
    const bf16_t* brow = B + (long)min(n0 + r, g.N - 1) * g.ldb + 8 * h2;
    const bf16_t* arow0 = A + (long)min(r, g.M - 1) * g.lda + 8 * h2;
    const bf16_t* arow1 = A + (long)min(32 + r, g.M - 1) * g.lda + 8 * h2;
    const bool two = g.M > 32;
    f32x16 acc[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
    const int ksteps = g.K >> 4;
    for (int ks = wave; ks < ksteps; ks += 8 * WAVES) {      // 8 steps of this wave per batch: their 24 loads are in flight together
        bf16x8 fb[8], fa0[8], fa1[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int k = (ks + WAVES * u) << 4;
            const bool in = ks + WAVES * u < ksteps;
            const int kc = in ? k : 0;
            fb[u] = *(const bf16x8*)(brow + kc);
            fa0[u] = *(const bf16x8*)(arow0 + kc);
            if (two) fa1[u] = *(const bf16x8*)(arow1 + kc);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (ks + WAVES * u < ksteps) {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[u], fa0[u], acc[0], 0, 0, 0);
                if (two) acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[u], fa1[u], acc[1], 0, 0, 0);
            }
        }
    }
    if (wave > 0) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) red[wave - 1][t][e][lane] = acc[t][e];
    }
    __syncthreads();
    if (wave == 0) {
        bf16_t* C = (bf16_t*)g.C;
        const bf16_t* R = (const bf16_t*)g.resid;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int m = t * 32 + r;                        // activation row of this lane
            if (m >= g.M) continue;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int n = n0 + (e & 3) + 8 * (e >> 2) + 4 * h2;
                if (n >= g.N) continue;
                float v = acc[t][e];
#pragma unroll
                for (int w = 0; w < WAVES - 1; ++w) v += red[w][t][e][lane];
                if (g.epi & EPI_BIAS) v += g.bias[n];
                if (g.epi & EPI_RESID) v += (float)R[(long)m * g.ldr + n];
                C[(long)m * g.ldc + n] = (bf16_t)v;
            }
        }
    }
}

}  // namespace

static unsigned long long* g_gemm_dbg = nullptr;
extern "C" void pseld_gemm_set_debug_buffer(void* p) { g_gemm_dbg = (unsigned long long*)p; }

// C[M,N] = op(A) op(B) with fused prologue/epilogue; see include/pseld_hip.h for the contract.
extern "C" int pseld_gemm(int dtype, int trans_a, int trans_b, const void* A, const void* B, void* C, int M, int N,
                          int K, int lda, int ldb, int ldc, const float* bias, const void* resid, int ldr,
                          const float* rowscale, int rows_per_scale, const void* aux, int ldaux, int epi, int pro,
                          void* c2, void* stream) {
    PSELD_CHECK_ARG(A && B && C, "gemm: null operand");
    PSELD_CHECK_ARG(M > 0 && N > 0 && K > 0, "gemm: bad shape %dx%dx%d", M, N, K);
    PSELD_CHECK_ARG(lda % 8 == 0 && ldb % 8 == 0, "gemm: lda/ldb must be multiples of 8 (%d,%d)", lda, ldb);
    // K %% 8 != 0 is allowed only for B = [K,N] (rows >= K are zero-filled) with A rows padded (finite) to 8
    PSELD_CHECK_ARG(K % 8 == 0 || (trans_b && lda >= (K + 7) / 8 * 8), "gemm: K=%d must be a multiple of 8", K);
    PSELD_CHECK_ARG(!(trans_a && !trans_b), "gemm: layout A^T*B^T is not built");
    PSELD_CHECK_ARG(!trans_a, "gemm: use pseld_gemm_wgrad for the split-K weight-gradient layout");
    PSELD_CHECK_ARG(!(epi & EPI_BIAS) || bias, "gemm: EPI_BIAS without bias");
    PSELD_CHECK_ARG(!(epi & EPI_RESID) || resid, "gemm: EPI_RESID without resid");
    PSELD_CHECK_ARG(!(epi & (EPI_MULGELUGRAD | EPI_MULAUX)) || aux, "gemm: aux-multiply epilogue without aux");
    PSELD_CHECK_ARG(!(epi & EPI_GELU_DUAL) || c2, "gemm: EPI_GELU_DUAL without the second output");
    PSELD_CHECK_ARG(!trans_b || N % 8 == 0, "gemm: N must be a multiple of 8 when B is [K,N]");
    PSELD_CHECK_ARG(dtype != PSELD_BF16 || (N % 8 == 0 && ldc % 8 == 0 && (!resid || ldr % 8 == 0) && (!aux || ldaux % 8 == 0)),
                    "gemm(bf16): N and ldc/ldr/ldaux must be multiples of 8 (%d,%d)", N, ldc);
    GemmArgs g;
    g.A = A; g.B = B; g.C = C; g.bias = bias; g.resid = resid; g.rowscale = rowscale; g.aux = aux; g.C2 = c2;
    g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.ldr = ldr; g.ldaux = ldaux;
    g.rows_per_scale = rows_per_scale > 0 ? rows_per_scale : 1;
    g.kchunk = K; g.slab_stride = 0; g.epi = epi; g.pro = pro; g.colsum = nullptr; g.dbg = g_gemm_dbg;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == PSELD_BF16 && !trans_a && !trans_b && pro == 0 && M <= 64 && K % 16 == 0 && N >= 512 && !rowscale &&
        (epi & ~(EPI_BIAS | EPI_RESID)) == 0) {
        if (K >= 2048) hipLaunchKernelGGL(gemm_skinny_kernel<8>, dim3(pseld_cdiv(N, 32)), dim3(512), 0, s, g);   // long K, few column tiles
        else hipLaunchKernelGGL(gemm_skinny_kernel<4>, dim3(pseld_cdiv(N, 32)), dim3(256), 0, s, g);
        g_last_gemm_kernel = "gemm_skinny_kernel";
        PSELD_LAUNCH_CHECK("gemm_skinny");
        return PSELD_OK;
    }
    // Products with K >= 192 (stages 1-3, merges, head): the persistent eight-phase kernel of gemm8.hip (at K = 192 its whole-line epilogue
    // is what wins: stage-1 qkv 96 -> 84 us, fc1 + GELU pair 200 -> 170 us against the 128 x 192 kernel).
    // PSELD_GEMM8=0 disables it, PSELD_GEMM8_MINK=<K> moves the threshold (knobs, common.h)
    if (dtype == PSELD_BF16 && !trans_a && !trans_b && pro == 0 && (epi & ~(EPI_BIAS | EPI_RESID | EPI_MULAUX | EPI_GELU_DUAL)) == 0) {
        if (pseld_knob(KNOB_GEMM8, 1) != 0 && K >= pseld_knob(KNOB_GEMM8_MINK, 192)) {
            Gemm8Desc d{};
            d.A = A; d.B = B; d.C = C; d.C2 = (epi & EPI_GELU_DUAL) ? c2 : nullptr;
            d.bias = (epi & EPI_BIAS) ? bias : nullptr; d.resid = (epi & EPI_RESID) ? resid : nullptr;
            d.aux = (epi & EPI_MULAUX) ? aux : nullptr; d.rowscale = rowscale;
            d.M = M; d.N = N; d.K = K; d.lda = lda; d.ldb = ldb; d.ldc = ldc; d.ldr = ldr; d.ldaux = ldaux;
            d.rows_per_scale = g.rows_per_scale; d.gelu_dual = (epi & EPI_GELU_DUAL) ? 1 : 0;
            // K = 192 / 384 on enough rows to fill the chip with 192-row panels: the row-panel-stationary kernel (gemm8p.hip; same bits, so the
            // choice may depend on M). OFF by default (knob GEMM8P = 1 switches it on, GEMM8P_MINM moves the row threshold, GEMM8P_MODES / _K
            // pick epilogue kinds / K): isolated on cold operands it takes 12 % off the stage-2 qkv and fc1 products (47.6 against 53.8 us, 93.6
            // against 107.0) and loses on the residual / aux epilogues; inside the step every routing measured 18.51-18.73 ms against
            // 18.46-18.58 without it (profiles/r06_gemm8p_ab.txt) - the A rows are read once, but the per-tile time is then bound by the matrix
            // pipe and the exposed A-panel load (stamps: docs/EXPERIMENTS.md, round 6)
            if (pseld_knob(KNOB_GEMM8P, 0) != 0 && M >= pseld_knob(KNOB_GEMM8P_MINM, 36864) && pseld_gemm8p_supported(d) && pseld_gemm8p_wanted(d)) {
                const int rcp = pseld_gemm8p_launch(d, s);
                g_last_gemm_kernel = pseld_gemm8p_last_symbol();
                return rcp;
            }
            if (pseld_gemm8_supported(d)) {
                const int rc8 = pseld_gemm8_launch(d, s);
                g_last_gemm_kernel = pseld_gemm8_last_symbol();
                return rc8;
            }
        }
    }
    if (dtype == PSELD_BF16 && !trans_a && !trans_b && pro == 0 && K % 32 == 0 && lda % 8 == 0 && ldb % 8 == 0 && M >= 128) {
        if (pseld_knob(KNOB_GEMM_DMA, 1) != 0) {
            const bool narrow = g.N <= 96 || (g.N % 192 != 0 && g.N % 96 == 0 && g.N <= 288);
            if (narrow && K < 384) return launch_gemm_dma<4, 1>(g, s);     // 256x96 with a long K loop: gemm_kernel is faster
            if (!narrow) {
                // long contractions: the 256 x 192 ring kernel (PSELD_GEMM_FWD_RING=<K threshold>, 0 = never)
                { const int kth = pseld_knob(KNOB_GEMM_FWD_RING, 3072);
                  if (kth > 0 && K >= kth && M >= 2048) return launch_gemm_fwd_ring(g, s); }
                // PSELD_GEMM_RING3=<K>: 3-stage ring at two workgroups per CU (4 slices in flight per CU instead of 3) for K >= <K>
                const int ring_k = pseld_knob(KNOB_GEMM_RING3, 0);
                if (ring_k > 0 && K >= ring_k) return launch_gemm_dma<2, 2, 3>(g, s);
                return launch_gemm_dma<2, 2>(g, s);
            }
        }
    }
    if (dtype == PSELD_BF16) {
        return trans_b ? dispatch_tile<bf16_t, bf16_t, false, true>(g, 1, s)
                       : dispatch_tile<bf16_t, bf16_t, false, false>(g, 1, s);
    } else if (dtype == PSELD_F32) {
        return trans_b ? dispatch_tile<float, float, false, true>(g, 1, s)
                       : dispatch_tile<float, float, false, false>(g, 1, s);
    }
    pseld_set_error("gemm: unknown dtype %d", dtype);
    return PSELD_ERR_BAD_ARG;
}

// Input gradient of a Linear that follows a LayerNorm, with that LayerNorm's backward in the epilogue (bf16; C = 96 or 192: the output
// tile spans the row): dx[M, C] = LN'(dY[M, K] Wt[C, K]^T; x, gamma) (+ dres), i.e. pseld_gemm (input gradient through the transposed weight
// copy Wt) + pseld_layernorm_bwd in one launch. partial: fp32 [pseld_gemm_dgrad_lnbwd_parts(M, C)][2][C], this launch's row-tile sums of
// d(gamma) = sum dxh xh and d(beta) = sum dxh (reduce with pseld_reduce_slabs / pseld_reduce_slabs_batched, as the stand-alone kernel's).
// (C = 384 - a 128 x 384 row-spanning tile of the eight-phase kernel with this epilogue - was built in round 5, measured equal to two launches
// in the step and removed in round 6: docs/EXPERIMENTS.md.)
extern "C" int pseld_gemm_dgrad_lnbwd_supported(int dtype, long M, int C, int K) {
    return dtype == PSELD_BF16 && (C == 96 || C == 192) && K % 32 == 0 && K >= 64 && M >= 128 ? 1 : 0;
}
extern "C" long pseld_gemm_dgrad_lnbwd_parts(long M, int C) { return pseld_cdiv(M, C == 96 ? 256 : 128); }
extern "C" int pseld_gemm_dgrad_lnbwd(int dtype, const void* dY, const void* Wt, const void* x, const float* gamma, const void* dres, void* dx,
                                      float* partial, long M, int C, int K, int lddy, int ldwt, float eps, void* stream) {
    PSELD_CHECK_ARG(dY && Wt && x && gamma && dx && partial, "gemm_dgrad_lnbwd: null pointer");
    PSELD_CHECK_ARG(pseld_gemm_dgrad_lnbwd_supported(dtype, M, C, K), "gemm_dgrad_lnbwd: built for bf16, C = 96 / 192, K %% 32 == 0 (got dtype %d C %d K %d)", dtype, C, K);
    PSELD_CHECK_ARG(lddy % 8 == 0 && ldwt % 8 == 0 && M < (1L << 31), "gemm_dgrad_lnbwd: bad leading dimension / M");
    GemmArgs g;
    memset(&g, 0, sizeof(g));
    g.A = dY; g.B = Wt; g.C = dx; g.bias = gamma; g.resid = dres; g.aux = x; g.C2 = partial;
    g.M = (int)M; g.N = C; g.K = K; g.lda = lddy; g.ldb = ldwt; g.ldc = C; g.ldr = C; g.ldaux = C;
    g.rows_per_scale = 1; g.kchunk = K; g.ln_eps = eps; g.dbg = nullptr;
    hipStream_t s = (hipStream_t)stream;
    g.nx = 1; g.nz = 1; g.xcd_swizzle = 1;
    if (C == 96) {
        constexpr int BM = 256, BN = 96, STAGE = (BM + BN) * 64, CS = BM * (BN * 2 + 16), RED = (256 / 4) * 4 * 48 * 4;
        constexpr int LDS = (2 * STAGE + 1024 > CS ? 2 * STAGE + 1024 : CS) > RED ? (2 * STAGE + 1024 > CS ? 2 * STAGE + 1024 : CS) : RED;
        g.ny = pseld_cdiv(g.M, BM);
        static bool attr = false;
        if (!attr) { (void)hipFuncSetAttribute((const void*)gemm_dma_kernel<4, 1, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS); attr = true; }
        hipLaunchKernelGGL((gemm_dma_kernel<4, 1, 2, true>), dim3((unsigned)(8 * pseld_cdiv(g.ny, 8))), dim3(256), LDS, s, g);
    } else {
        constexpr int BM = 128, BN = 192, STAGE = (BM + BN) * 64, CS = BM * (BN * 2 + 16), RED = (256 / 8) * 8 * 48 * 4;
        constexpr int LDS = (2 * STAGE + 1024 > CS ? 2 * STAGE + 1024 : CS) > RED ? (2 * STAGE + 1024 > CS ? 2 * STAGE + 1024 : CS) : RED;
        g.ny = pseld_cdiv(g.M, BM);
        static bool attr = false;
        if (!attr) { (void)hipFuncSetAttribute((const void*)gemm_dma_kernel<2, 2, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS); attr = true; }
        hipLaunchKernelGGL((gemm_dma_kernel<2, 2, 2, true>), dim3((unsigned)(8 * pseld_cdiv(g.ny, 8))), dim3(256), LDS, s, g);
    }
    g_last_gemm_kernel = C == 96 ? "gemm_dma_kernel<4, 1, 2, true>" : "gemm_dma_kernel<2, 2, 2, true>";
    PSELD_LAUNCH_CHECK("gemm_dgrad_lnbwd");
    return PSELD_OK;
}

// Weight gradient dW[N,K] (fp32) = dY[Mtok,N]^T @ X[Mtok,K], optionally with GELU applied to X on load
// (dW2 = dY^T gelu(u)). Split over the token dimension into `splits` fp32 slabs in `workspace`
// (needs splits*N*K floats), then reduced into dW (overwrite, or accumulate when `accumulate`).
// Split planning (tools/wgrad_sweep.py, s_memtime stamps): ONE resident round of workgroups — the number of splits is
// the number of workgroup slots of the chip (256 CUs x workgroups resident per CU for the tile the dispatcher will use)
// divided by the number of output tiles, so no CU idles and there is no tail round; at least 512 tokens per split
// keeps the fp32 slab traffic small.
static int wgrad_fill_percent() {
    const int t = pseld_knob(KNOB_WGRAD_FILL, 100);
    return t < 10 ? 100 : t;
}
static int wgrad_min_tokens() {
    const int t = pseld_knob(KNOB_WGRAD_MINTOK, 512);
    return t < 64 ? 512 : t;
}
static int wgrad_splits_for(int dtype, int Mtok, int N, int K) {
    int bm, bn, slots;
    if (K <= 96 || (K % 192 != 0 && K % 96 == 0 && K <= 288)) { bm = 256; bn = 96; slots = 512; }
    else if (dtype == PSELD_BF16 && N >= 256 && gemm_big_tiles()) { bm = 256; bn = 192; slots = 256; }
    else { bm = 128; bn = 192; slots = dtype == PSELD_BF16 ? 512 : 256; }
    const int tiles = pseld_cdiv(N, bm) * pseld_cdiv(K, bn);
    int splits = (int)((long)slots * wgrad_fill_percent() / 100 / tiles);
    // a weight with more than half a round of output tiles (the CNN14 / Conformer matrices) would get ONE split and leave
    // the tail of its single round idle: three rounds of shorter workgroups instead (tools/wgrad_sweep_crnn.py:
    // 24000 x 1024 x 9216 1498 -> 685 us; the small weights stay with one resident round)
    if (2 * tiles > slots) splits = (int)((long)slots * 3 / tiles);
    // XCD-swizzled launches give whole splits to one XCD: keep the 8 XCDs evenly loaded
    if (gemm_xcd_mode(true) && splits >= 8) splits = splits / 8 * 8;
    const int max_splits = pseld_cdiv(Mtok, wgrad_min_tokens());
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    return splits;
}
// The ring kernel (gemm_wgrad_ring_kernel) takes the large bf16 weight gradients: returns its M-tile count in 32-row units
// (3: 384 x 192 tile, 2: 256 x 192) and the split count, or 0 when the register-staged kernel keeps the shape.
static int wgrad_ring_plan(int dtype, int Mtok, int N, int K, int gelu_on_x, int rows_per_scale, bool has_rowscale, int* splits_out) {
    const int enabled = pseld_knob(KNOB_WGRAD_RING, 1);
    if (!enabled || dtype != PSELD_BF16 || gelu_on_x || Mtok % 32 != 0 || N % 8 != 0 || K % 8 != 0) return 0;
    if (has_rowscale && rows_per_scale % 32 != 0) return 0;
    if (N < 256 || K < 192 || (long)N * K < 384L * 384) return 0;          // small matrices: the 256x96 / 128x192 tiles waste less
    auto tiles_of = [&](int bm) { return (long)pseld_cdiv(N, bm) * pseld_cdiv(K, 192); };
    const long pad3 = tiles_of(384) * 384 * 192, pad2 = tiles_of(256) * 256 * 192;
    const int mt = pad3 <= pad2 ? 3 : 2;
    if ((double)(mt == 3 ? pad3 : pad2) > 1.34 * (double)N * K) return 0;   // more than a third of the tile area would be padding
    const int tiles = (int)tiles_of(mt == 3 ? 384 : 256);
    int splits = 256 / tiles;                                               // one workgroup per CU, one resident round
    if (splits >= 8 && gemm_xcd_mode(true) && (splits / 8 * 8) * 10 >= splits * 9) splits = splits / 8 * 8;   // XCD-local splits if that keeps >= 90 % of the CUs busy
    const int max_splits = Mtok / 512;
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    int kchunk = pseld_cdiv(pseld_cdiv(Mtok, splits), 32) * 32;
    // short token ranges per workgroup (< 32 slices) leave the ring's prologue and the 384 x 192 fp32 slab store un-amortised:
    // tools/wgrad_ab.py, 12288 x 768 x 768 and 49152 x 384 x 384 (512 tokens per split) are 5-7 % slower than the 256 x 192 kernel
    if (kchunk < 1024 && enabled != 2) return 0;      // (PSELD_WGRAD_RING=2: take every eligible shape - the parity tests)
    *splits_out = pseld_cdiv(Mtok, kchunk);
    return mt;
}
// ---- measurement aid: the weight-gradient entry point launches a GEMM kernel AND a slab reduction, so HIP events around the call do not
// time one kernel. With pseld_gemm_wgrad_timing(1) every pseld_gemm_wgrad call brackets its GEMM kernel alone with two library-owned
// events; pseld_gemm_wgrad_timing_read(i) returns the i-th call's kernel time in ms (bench.py: the per-symbol roofline over ALL symbols).
#include <vector>
namespace {
struct WgradStamp { hipEvent_t a, b; const char* sym; };
std::vector<WgradStamp> g_wgrad_stamps;
size_t g_wgrad_stamp_n = 0;
bool g_wgrad_timing = false;
inline WgradStamp* wgrad_stamp_begin(hipStream_t s) {
    if (!g_wgrad_timing) return nullptr;
    if (g_wgrad_stamp_n == g_wgrad_stamps.size()) {
        WgradStamp w{};
        if (hipEventCreate(&w.a) != hipSuccess || hipEventCreate(&w.b) != hipSuccess) return nullptr;
        g_wgrad_stamps.push_back(w);
    }
    WgradStamp* w = &g_wgrad_stamps[g_wgrad_stamp_n++];
    (void)hipEventRecord(w->a, s);
    return w;
}
inline void wgrad_stamp_end(WgradStamp* w, hipStream_t s, const char* sym) { if (w) { (void)hipEventRecord(w->b, s); w->sym = sym; } }
}  // namespace
extern "C" int pseld_gemm_wgrad_timing(int enable) { g_wgrad_timing = enable != 0; if (enable) g_wgrad_stamp_n = 0; return PSELD_OK; }     // (switching off keeps the records readable)
extern "C" int pseld_gemm_wgrad_timing_count(void) { return (int)g_wgrad_stamp_n; }
extern "C" float pseld_gemm_wgrad_timing_read(int i) {
    if (i < 0 || (size_t)i >= g_wgrad_stamp_n) return -1.f;
    float ms = -1.f;
    if (hipEventSynchronize(g_wgrad_stamps[i].b) != hipSuccess || hipEventElapsedTime(&ms, g_wgrad_stamps[i].a, g_wgrad_stamps[i].b) != hipSuccess) return -1.f;
    return ms;
}
extern "C" const char* pseld_gemm_wgrad_timing_symbol(int i) { return (i < 0 || (size_t)i >= g_wgrad_stamp_n || !g_wgrad_stamps[i].sym) ? "" : g_wgrad_stamps[i].sym; }

static int gemm8w_min_n() { return pseld_knob(KNOB_WGRAD8_MINN, 192); }     // (dW[192, 768] over 196 608 tokens: 102 against 127 us)
extern "C" long pseld_gemm_wgrad_workspace(int Mtok, int N, int K, int* splits_out) {
    const int sa = wgrad_splits_for(PSELD_BF16, Mtok, N, K), sb = wgrad_splits_for(PSELD_F32, Mtok, N, K);
    int sr = 0;
    {   // the ring kernel's plan (always counted, whatever PSELD_WGRAD_RING says now: workspaces are sized once)
        const long t3 = (long)pseld_cdiv(N, 384) * pseld_cdiv(K, 192), t2 = (long)pseld_cdiv(N, 256) * pseld_cdiv(K, 192);
        const long tiles = t3 < t2 ? t3 : t2;
        sr = (int)(256 / (tiles > 0 ? tiles : 1));
        if (sr > Mtok / 512) sr = Mtok / 512;
        if (sr < 1) sr = 1;
    }
    int splits = (sa > sb ? sa : sb);
    if (sr > splits) splits = sr;
    splits += 1;                                       // +1: rounding kchunk to the slice size can add one split
    if (splits_out) *splits_out = splits;
    return (long)splits * ((long)N * K + N) * (long)sizeof(float);   // dW slabs + bias-gradient slabs
}

extern "C" int pseld_gemm_wgrad(int dtype, const void* dY, const void* X, float* dW, float* dbias, int Mtok, int N, int K,
                                int lddy, int ldx, int lddw, int gelu_on_x, int accumulate, float* workspace,
                                long workspace_bytes, const float* rowscale, int rows_per_scale, void* stream) {
    PSELD_CHECK_ARG(dY && X && dW && workspace, "gemm_wgrad: null pointer");
    PSELD_CHECK_ARG(Mtok > 0 && N > 0 && K > 0, "gemm_wgrad: bad shape");
    PSELD_CHECK_ARG(lddy % 8 == 0 && ldx % 8 == 0 && K % 8 == 0,
                    "gemm_wgrad: K/ld must be multiples of 8 (%d,%d,%d)", K, lddy, ldx);
    PSELD_CHECK_ARG(N % 8 == 0 || lddy >= (N + 7) / 8 * 8, "gemm_wgrad: N=%d needs dY rows padded to 8", N);
    PSELD_CHECK_ARG(lddw == K, "gemm_wgrad: dW must be dense [N,K]");
    const long need = pseld_gemm_wgrad_workspace(Mtok, N, K, nullptr);
    PSELD_CHECK_ARG(workspace_bytes >= need, "gemm_wgrad: workspace %ld < %ld bytes", workspace_bytes, need);
    // MFMA-bound weight gradients: the eight-phase kernel of gemm8w.hip (PSELD_WGRAD8=0 disables it: knob, common.h)
    if (dtype == PSELD_BF16 && !gelu_on_x && N >= gemm8w_min_n() && K >= 192 && Mtok >= 4096) {
        if (pseld_knob(KNOB_WGRAD8, 1) != 0) {
            int bn8 = 0, kchunk8 = 0;
            const long per_split = ((long)N * K + N) * (long)sizeof(float);
            const int max_splits = (int)(workspace_bytes / per_split);
            const int s8 = pseld_gemm8w_plan(Mtok, N, K, lddy, ldx, rows_per_scale > 0 ? rows_per_scale : 1, rowscale != nullptr, max_splits, &bn8, &kchunk8);
            if (s8 > 0) {
                const bool fused = dbias && dbias == dW + (long)N * K;
                const long stride = fused ? (long)N * K + N : (long)N * K;
                float* cs = fused ? workspace + (long)N * K : (dbias ? workspace + (long)s8 * N * K : nullptr);
                g_last_gemm_kernel = bn8 == 192 ? "gemm8w_kernel<3, (anonymous namespace)::G8WOne>" : "gemm8w_kernel<4, (anonymous namespace)::G8WOne>";   // as rocprofv3 prints them
                WgradStamp* st8 = wgrad_stamp_begin((hipStream_t)stream);
                const int rc8 = pseld_gemm8w_launch(dY, X, workspace, cs, stride, fused ? stride : (long)N, Mtok, N, K, lddy, ldx, bn8, kchunk8, s8,
                                                    rowscale, rows_per_scale > 0 ? rows_per_scale : 1, (hipStream_t)stream);
                wgrad_stamp_end(st8, (hipStream_t)stream, g_last_gemm_kernel);
                if (rc8 != PSELD_OK) return rc8;
                const long n8 = (long)N * K;
                if (fused) pseld_reduce_slabs(workspace, dW, n8 + N, s8, stride, accumulate, (hipStream_t)stream);
                else {
                    pseld_reduce_slabs(workspace, dW, n8, s8, stride, accumulate, (hipStream_t)stream);
                    if (dbias) pseld_reduce_slabs(cs, dbias, (long)N, s8, (long)N, accumulate, (hipStream_t)stream);
                }
                PSELD_LAUNCH_CHECK("splitk_reduce");
                return PSELD_OK;
            }
        }
    }
    int splits = wgrad_splits_for(dtype, Mtok, N, K);
    int ring_splits = 0;
    const int ring_mt = wgrad_ring_plan(dtype, Mtok, N, K, gelu_on_x, rows_per_scale > 0 ? rows_per_scale : 1, rowscale != nullptr, &ring_splits);
    GemmArgs g;
    memset(&g, 0, sizeof(g));
    g.A = dY; g.B = X; g.C = workspace;
    g.M = N; g.N = K; g.K = Mtok; g.lda = lddy; g.ldb = ldx; g.ldc = K;
    g.rows_per_scale = rows_per_scale > 0 ? rows_per_scale : 1;
    g.rowscale = rowscale;
    const int bk = ring_mt ? 32 : (dtype == PSELD_BF16) ? 64 : 32;
    if (ring_mt) splits = ring_splits;
    int kchunk = pseld_cdiv(Mtok, splits);
    kchunk = pseld_cdiv(kchunk, bk) * bk;
    splits = pseld_cdiv(Mtok, kchunk);
    g.dbg = g_gemm_dbg;
    g.kchunk = kchunk; g.slab_stride = (long)N * K; g.epi = EPI_NONE; g.pro = (gelu_on_x ? PRO_GELU_B : PRO_NONE) | (rowscale ? PRO_ROWSCALE_A : PRO_NONE);
    // when the bias gradient sits right behind the weight gradient (as in the parameter arena) the bias slabs are
    // interleaved with the dW slabs and ONE reduction covers both
    const bool fused_bias = dbias && dbias == dW + (long)N * K;
    if (fused_bias) { g.slab_stride = (long)N * K + N; g.colsum = workspace + (long)N * K; g.colsum_stride = g.slab_stride; }
    else { g.colsum = dbias ? workspace + (long)splits * N * K : nullptr; g.colsum_stride = N; }
    hipStream_t s = (hipStream_t)stream;
    int rc;
    WgradStamp* st = wgrad_stamp_begin(s);
    if (ring_mt == 3) rc = launch_wgrad_ring<3, 3>(g, splits, s);
    else if (ring_mt == 2) rc = launch_wgrad_ring<2, 3>(g, splits, s);
    else if (dtype == PSELD_BF16) rc = dispatch_tile<bf16_t, float, true, true>(g, splits, s);
    else if (dtype == PSELD_F32) rc = dispatch_tile<float, float, true, true>(g, splits, s);
    else { pseld_set_error("gemm_wgrad: unknown dtype %d", dtype); return PSELD_ERR_BAD_ARG; }
    wgrad_stamp_end(st, s, g_last_gemm_kernel);
    if (rc != PSELD_OK) return rc;
    const long n = (long)N * K;
    if (fused_bias) {
        pseld_reduce_slabs(workspace, dW, n + N, splits, g.slab_stride, accumulate, s);
    } else {
        pseld_reduce_slabs(workspace, dW, n, splits, g.slab_stride, accumulate, s);
        if (dbias) pseld_reduce_slabs(g.colsum, dbias, (long)N, splits, (long)N, accumulate, s);
    }
    PSELD_LAUNCH_CHECK("splitk_reduce");
    return PSELD_OK;
}

// out[n] (fp32) = sum_m X[m,n]; two deterministic passes through `workspace` (>= blocks*N floats).
static inline int colsum_rows(int M) { int r = pseld_cdiv(M, 1024); return r < 256 ? 256 : r; }   // about four workgroups per CU
extern "C" long pseld_colsum_workspace(int M, int N) {
    return (long)pseld_cdiv(M, colsum_rows(M)) * N * (long)sizeof(float);
}
extern "C" int pseld_colsum(int dtype, const void* X, float* out, int M, int N, int ld, int accumulate,
                            float* workspace, long workspace_bytes, void* stream) {
    PSELD_CHECK_ARG(X && out && workspace, "colsum: null pointer");
    const int rows_per_block = colsum_rows(M);
    const int nb = pseld_cdiv(M, rows_per_block);
    PSELD_CHECK_ARG(workspace_bytes >= (long)nb * N * 4, "colsum: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    dim3 grid(pseld_cdiv(N, 64), nb);
    const bool vec = N % 8 == 0 && ld % 8 == 0 && ((unsigned long)X & 15) == 0;
    const dim3 gridv(pseld_cdiv(N / 8 > 0 ? N / 8 : 1, 256), nb);
    if (dtype == PSELD_BF16 && vec)
        hipLaunchKernelGGL(colsum_vec_kernel<bf16_t>, gridv, dim3(256), 0, s, (const bf16_t*)X, workspace, M, N, ld, rows_per_block);
    else if (dtype == PSELD_F32 && vec)
        hipLaunchKernelGGL(colsum_vec_kernel<float>, gridv, dim3(256), 0, s, (const float*)X, workspace, M, N, ld, rows_per_block);
    else if (dtype == PSELD_BF16)
        hipLaunchKernelGGL(colsum_partial_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)X, workspace, M, N, ld, rows_per_block);
    else
        hipLaunchKernelGGL(colsum_partial_kernel<float>, grid, dim3(256), 0, s, (const float*)X, workspace, M, N, ld, rows_per_block);
    PSELD_LAUNCH_CHECK("colsum_partial");
    pseld_reduce_slabs(workspace, out, (long)N, nb, (long)N, accumulate, s);
    PSELD_LAUNCH_CHECK("colsum_reduce");
    return PSELD_OK;
}


// ---- implicit 3x3 convolution (pad 1) on NHWC rows: the im2col matrix of cnn.hip is never built ------------------------------
static void conv_geometry(GemmArgs& g, int T, int F, int C) {
    g.cv_T = T; g.cv_F = F; g.cv_C = C; g.cv_rF = 1.0f / (float)F; g.cv_rT = 1.0f / (float)T;
}
// Y[B*T*F, N] = im2col(X)[B*T*F, 9*C] @ Wp[N, 9*C]^T. Forward: Wp = conv_weight_to_tap(W); input gradient: X = dY
// (C = Cout) and Wp = conv_weight_to_tap_t(W) (N = Cin padded).
extern "C" int pseld_conv3x3_fwd(int dtype, const void* X, const void* Wp, void* Y, int B, int T, int F, int C, int N, void* stream) {
    PSELD_CHECK_ARG(X && Wp && Y && B > 0 && T > 0 && F > 0, "conv3x3_fwd: bad argument");
    PSELD_CHECK_ARG(C % 8 == 0 && N % 8 == 0 && (long)B * T * F < (1L << 24), "conv3x3_fwd: C, N multiples of 8; rows < 2^24");
    GemmArgs g;
    memset(&g, 0, sizeof(g));
    g.A = X; g.B = Wp; g.C = Y;
    g.M = B * T * F; g.N = N; g.K = 9 * C; g.lda = C; g.ldb = 9 * C; g.ldc = N;
    g.rows_per_scale = 1; g.kchunk = g.K; g.epi = EPI_NONE; g.pro = PRO_NONE; g.dbg = nullptr;
    conv_geometry(g, T, F, C);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == PSELD_BF16) return dispatch_tile<bf16_t, bf16_t, false, false, true>(g, 1, s);
    if (dtype == PSELD_F32) return dispatch_tile<float, float, false, false, true>(g, 1, s);
    pseld_set_error("conv3x3_fwd: unknown dtype %d", dtype);
    return PSELD_ERR_BAD_ARG;
}
extern "C" long pseld_conv3x3_wgrad_workspace(int B, int T, int F, int C, int N) { return pseld_gemm_wgrad_workspace(B * T * F, N, 9 * C, nullptr); }
// dWp[N, 9*C] (fp32, tap-major) (+)= dY[B*T*F, N]^T @ im2col(X)
extern "C" int pseld_conv3x3_wgrad(int dtype, const void* dY, const void* X, float* dWp, int B, int T, int F, int C, int N,
                                   int accumulate, float* workspace, long workspace_bytes, void* stream) {
    PSELD_CHECK_ARG(dY && X && dWp && workspace && B > 0 && T > 0 && F > 0, "conv3x3_wgrad: bad argument");
    PSELD_CHECK_ARG(C % 8 == 0 && N % 8 == 0 && (long)B * T * F < (1L << 24), "conv3x3_wgrad: C, N multiples of 8; rows < 2^24");
    const int Mtok = B * T * F, K = 9 * C;
    PSELD_CHECK_ARG(workspace_bytes >= pseld_gemm_wgrad_workspace(Mtok, N, K, nullptr), "conv3x3_wgrad: workspace too small");
    int splits = wgrad_splits_for(dtype, Mtok, N, K);
    GemmArgs g;
    memset(&g, 0, sizeof(g));
    g.A = dY; g.B = X; g.C = workspace;
    g.M = N; g.N = K; g.K = Mtok; g.lda = N; g.ldb = C; g.ldc = K;
    g.rows_per_scale = 1;
    const int bk = (dtype == PSELD_BF16) ? 64 : 32;
    int kchunk = pseld_cdiv(Mtok, splits);
    kchunk = pseld_cdiv(kchunk, bk) * bk;
    splits = pseld_cdiv(Mtok, kchunk);
    g.kchunk = kchunk; g.slab_stride = (long)N * K; g.epi = EPI_NONE; g.pro = PRO_NONE;
    conv_geometry(g, T, F, C);
    hipStream_t s = (hipStream_t)stream;
    int rc;
    if (dtype == PSELD_BF16) rc = dispatch_tile<bf16_t, float, true, true, true>(g, splits, s);
    else if (dtype == PSELD_F32) rc = dispatch_tile<float, float, true, true, true>(g, splits, s);
    else { pseld_set_error("conv3x3_wgrad: unknown dtype %d", dtype); return PSELD_ERR_BAD_ARG; }
    if (rc != PSELD_OK) return rc;
    pseld_reduce_slabs(workspace, dWp, (long)N * K, splits, g.slab_stride, accumulate, s);
    PSELD_LAUNCH_CHECK("conv3x3_wgrad");
    return PSELD_OK;
}
