// Persistent eight-phase bf16 GEMM for the MFMA-bound Linear layers of HTS-AT (stages 2-3, merges, head) on gfx950.
//
// C[M,N] = A[M,K] B[N,K]^T with the fused epilogues of pseld_gemm (include/pseld_hip.h): + bias, x DropPath factor, x aux, + residual,
// GELU pair. Replaces, for products with K >= 384 (htsat.py:118,140 qkv / proj, model_utilities.py:166-170 fc1 / fc2, htsat.py:309
// PatchMerging.reduction, accdoa.py:230 head), the 128 x 192 three-workgroups-per-CU kernel of gemm.hip, whose slice loop is capped
// by the CU's LDS-DMA rate.
//
// Structure (cdna_hip_programming.md, "The 256^2 8-phase template"):
//  * ONE workgroup of 8 waves per CU, persistent over its list of 256 x 256 output tiles; waves 2 (M) x 4 (N), 128 x 64 per wave as
//    8 x 4 accumulator blocks of v_mfma_f32_16x16x32_bf16 computed TRANSPOSED (weight rows on the accumulator rows, tokens on the
//    lanes), so that a lane ends up with 8 consecutive output columns of one token row: the epilogue stores 16-byte pieces straight
//    from the accumulators (64 contiguous bytes per row and instruction) and needs neither LDS nor a barrier.
//  * 128 KiB of LDS = 2 K-tiles (BK = 64) x {A-h0, A-h1, B-h0, B-h1} half-tile images of 128 rows x 128 B, filled by
//    global_load_lds_dwordx4 (8 whole 128-byte rows per wave-instruction) with the 16-byte chunk XOR-swizzled on the SOURCE address
//    by (row >> 1) & 7: every ds_read_b128 fragment read is bank-conflict free.
//  * A K-tile is four phases {fragment reads + one half-tile of LDS-DMA | s_barrier | 16 MFMAs | s_barrier}; the two wave groups
//    (wr = 0 / 1, the two waves of each SIMD) run staggered by one barrier, so one computes while its partner loads. Loads are
//    issued in consumption order and three half-tiles stay in flight across every barrier (ONE counted vmcnt(6) per K-tile).
//  * The K-tiles of ALL the tiles a workgroup owns form one stream: while a tile's epilogue runs, the next tile's first K-tiles are
//    already in LDS / in flight. Accumulators start from the bias (an LDS copy of the whole bias vector), so the epilogue adds nothing.
//  * Tile order: each XCD owns a contiguous range of tile ids (N fastest), its 32 workgroups walk it round-robin: the N tiles that
//    share an A row block run together on one L2.
#include "gemm8.h"
#include <stdlib.h>

namespace {

typedef __attribute__((address_space(3))) void* lds_vptr8;

constexpr int HALF_B = 16384;             // one half-tile image: 128 rows x 128 B (64 bf16 of K)
constexpr int BUF_B = 4 * HALF_B;         // one K-tile: A-h0 | A-h1 | B-h0 | B-h1
constexpr int RING_B = 2 * BUF_B;         // 128 KiB
constexpr int BIAS_FLOATS = 4096;         // the product's whole bias vector (padded to the tile grid) lives in LDS
constexpr int LDS_B = RING_B + BIAS_FLOATS * 4;

enum { G8_PLAIN = 0, G8_RESID = 1, G8_MULAUX = 2, G8_GELU_DUAL = 3 };   // x SCALED (DropPath factor per token row)

// s_waitcnt vmcnt(0) the compiler's own wait bookkeeping sees (gfx9 encoding: vmcnt 0, expcnt 7, lgkmcnt 15)
#define G8_WAIT_VM0() __builtin_amdgcn_s_waitcnt(0x0F70)

struct G8Args {
    const char* A; const char* B; bf16_t* C; bf16_t* C2;
    const float* bias; const bf16_t* resid; const bf16_t* aux; const float* rowscale;
    int M, N, K, lda, ldb, ldc, ldr, ldaux, rows_per_scale;
    float inv_rps;
    int nx, ntiles, nk;
    unsigned long long* dbg;   // diagnostic instantiation only: s_memtime stamps per (workgroup, wave group, tile)
};

// x / d for 0 <= x < 2^24 through the reciprocal (exact after one fix-up step)
__device__ __forceinline__ int div_by8(int x, int d, float rd) {
    int q = (int)((float)x * rd);
    const int r = x - q * d;
    q += (r >= d) - (r < 0);
    return q;
}

// one LDS-DMA wave-instruction: 64 lanes x 16 B from (uniform base + per-lane 32-bit offset) to 1 KiB of LDS at a uniform address.
// Inline asm: the waits are counted by hand (hipcc would drain the prefetch with vmcnt(0) in front of every fragment read).
__device__ __forceinline__ void g8_dma(unsigned lds_dst, const void* sbase, unsigned voff) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(lds_dst), "s"(sbase) : "memory");
}

__device__ __forceinline__ f32x4 pack8f(const float (&v)[8]) {
    bf16x8 x;
#pragma unroll
    for (int k = 0; k < 8; ++k) x[k] = (bf16_t)v[k];
    return __builtin_bit_cast(f32x4, x);
}
__device__ __forceinline__ void unpack8f(const f32x4& p, float (&v)[8]) {
    const bf16x8 x = __builtin_bit_cast(bf16x8, p);
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = (float)x[k];
}

#define G8_BAR()                                  \
    do {                                          \
        __builtin_amdgcn_sched_barrier(0);        \
        __builtin_amdgcn_s_barrier();             \
        __builtin_amdgcn_sched_barrier(0);        \
    } while (0)

template <int MODE, bool SCALED, bool DBG = false>
__global__ __launch_bounds__(512, 2) void gemm8_kernel(const G8Args g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* bias_s = (float*)(smem + RING_B);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int l15 = lane & 15, q = lane >> 4;

    // ---- this workgroup's tiles: XCD x owns a contiguous range of tile ids, its workgroups take them round-robin ----
    const int nwg = gridDim.x, wg = blockIdx.x;
    const int xcd = wg & 7, slot = wg >> 3, per = nwg >> 3;
    const int Qt = g.ntiles >> 3, Rt = g.ntiles & 7;
    const int chunk0 = xcd * Qt + min(xcd, Rt), chunkn = Qt + (xcd < Rt ? 1 : 0);
    if (slot >= chunkn) return;
    const int my_n = (chunkn - slot + per - 1) / per;
    const int first = chunk0 + slot;

    for (int i = tid; i < g.nx * 256; i += 512) bias_s[i] = (g.bias && i < g.N) ? g.bias[i] : 0.f;
    __syncthreads();

    // ---- fragment read addresses (buffer 0): lane reads row l15 of a 16-row block, 16-byte chunk (4 kk + q) ^ ((row >> 1) & 7) ----
    const int sw = (lane >> 1) & 7;
    const unsigned c0 = (unsigned)((q ^ sw) << 4);
    unsigned ra0 = (unsigned)(wr * 8192 + l15 * 128) + c0, ra1 = ra0 ^ 64;
    unsigned rb0 = (unsigned)(2 * HALF_B + wc * 4096 + l15 * 128) + c0, rb1 = rb0 ^ 64;

    // ---- LDS-DMA source offsets of the load cursor's tile: [half][instruction] ----
    unsigned offA[2][2], offB[2][2];
    auto set_tile = [&](int T) {
        const int mblk = T / g.nx, nblk = T - mblk * g.nx;
        const int m0 = mblk * 256, n0 = nblk * 256;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int rho = wave * 16 + j * 8 + (lane >> 3);                  // image row this lane fills
            const int ch = (lane & 7) ^ ((rho >> 1) & 7);                     // source chunk (swizzle on the source)
            const int i = rho & 15;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int tok = min(m0 + (rho >> 6) * 128 + h * 64 + (rho & 63), g.M - 1);
                offA[h][j] = (unsigned)tok * (unsigned)(g.lda * 2) + (unsigned)(ch * 16);
                const int n = min(n0 + (rho >> 5) * 64 + 32 * h + 8 * (i >> 2) + 4 * ((rho >> 4) & 1) + (i & 3), g.N - 1);
                offB[h][j] = (unsigned)n * (unsigned)(g.ldb * 2) + (unsigned)(ch * 16);
            }
        }
    };
    const unsigned lds_base = (unsigned)(unsigned long)(lds_vptr8)smem;
    const unsigned dst_w = lds_base + (unsigned)wave * 2048u;
    int ld_i = 0, ld_kt = 0;
    unsigned ld_buf = 0;
    auto dmaA = [&](int h) {
        const char* sb = g.A + ld_kt * 128;
#pragma unroll
        for (int j = 0; j < 2; ++j) g8_dma(dst_w + ld_buf + (unsigned)(h * HALF_B + j * 1024), sb, offA[h][j]);
    };
    auto dmaB = [&](int h) {
        const char* sb = g.B + ld_kt * 128;
#pragma unroll
        for (int j = 0; j < 2; ++j) g8_dma(dst_w + ld_buf + (unsigned)((2 + h) * HALF_B + j * 1024), sb, offB[h][j]);
    };
    auto advance = [&]() {       // past the end of the list the cursor re-reads the last tile (nobody reads those images): the
        ld_buf ^= BUF_B;         // vmcnt distance stays constant in the tail
        if (++ld_kt == g.nk) {
            ld_kt = 0;
            if (ld_i + 1 < my_n) { ++ld_i; set_tile(first + ld_i * per); }
        }
    };

    bf16x8 fa[4][2], fb0[2][2], fb1[2][2];
    f32x4 acc[8][4];

    auto init_acc = [&](int n0) {
        const float* bp = bias_s + n0 + wc * 64 + 8 * q;
        f32x4 b[4];
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) b[nb] = *(const f32x4*)(bp + 32 * (nb >> 1) + 4 * (nb & 1));
#pragma unroll
        for (int mb = 0; mb < 8; ++mb)
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) acc[mb][nb] = b[nb];
    };

    // lane: token row m0 + wr*128 + mb*16 + l15, columns n0 + wc*64 + 32 nq + 8 q + {4 nbl + k} = acc[mb][2 nq + nbl][k]
    auto epilogue = [&](int m0, int n0) {
        constexpr bool HAS_X = MODE == G8_RESID || MODE == G8_MULAUX;
        const int rowb = m0 + wr * 128 + l15, colb = n0 + wc * 64 + 8 * q;
        const int mlast = g.M - 1, nlast = g.N - 8;
        const bf16_t* X = MODE == G8_RESID ? g.resid : g.aux;
        const int ldx = MODE == G8_RESID ? g.ldr : g.ldaux;
        f32x4 xv[HAS_X ? 16 : 1];
        float sc[SCALED ? 8 : 1];
        // every load of the epilogue is issued first, unconditionally (clamped addresses), and waited for by ONE wait the compiler
        // knows about: no load of its own is then pending at the loop's back edge, where it would otherwise drain the LDS-DMA
        // prefetch with a vmcnt(0) in front of the next K-tile's fragment reads
        if constexpr (SCALED || HAS_X) {
#pragma unroll
            for (int mb = 0; mb < 8; ++mb) {
                const int rowc = min(rowb + mb * 16, mlast);
                if constexpr (SCALED) sc[mb] = g.rowscale[div_by8(rowc, g.rows_per_scale, g.inv_rps)];
                if constexpr (HAS_X) {
#pragma unroll
                    for (int nq = 0; nq < 2; ++nq)
                        xv[mb * 2 + nq] = *(const f32x4*)(X + (long)rowc * ldx + min(colb + 32 * nq, nlast));
                }
            }
            G8_WAIT_VM0();
        }
#pragma unroll
        for (int mb = 0; mb < 8; ++mb) {
            const int row = rowb + mb * 16;
#pragma unroll
            for (int nq = 0; nq < 2; ++nq) {
                const int col = colb + 32 * nq;
                const bool ok = row < g.M && col < g.N;
                const long o = (long)min(row, mlast) * g.ldc + min(col, nlast);
                float v[8];
#pragma unroll
                for (int k = 0; k < 4; ++k) { v[k] = acc[mb][2 * nq][k]; v[4 + k] = acc[mb][2 * nq + 1][k]; }
                const float scm = SCALED ? sc[SCALED ? mb : 0] : 1.f;
                if constexpr (MODE == G8_PLAIN) {
                    if constexpr (SCALED) {
#pragma unroll
                        for (int k = 0; k < 8; ++k) v[k] *= scm;
                    }
                } else if constexpr (MODE == G8_RESID) {
                    float x[8];
                    unpack8f(xv[mb * 2 + nq], x);
#pragma unroll
                    for (int k = 0; k < 8; ++k) v[k] = SCALED ? fmaf(v[k], scm, x[k]) : v[k] + x[k];
                } else if constexpr (MODE == G8_MULAUX) {
                    float x[8];
                    unpack8f(xv[mb * 2 + nq], x);
#pragma unroll
                    for (int k = 0; k < 8; ++k) v[k] *= SCALED ? x[k] * scm : x[k];
                } else {
                    float dv[8];
#pragma unroll
                    for (int k = 0; k < 8; k += 2) {
                        f32x2 xx = {v[k], v[k + 1]}, yy, dd;
                        gelu_both2(xx, yy, dd);
                        v[k] = yy[0]; v[k + 1] = yy[1]; dv[k] = dd[0]; dv[k + 1] = dd[1];
                    }
                    if (ok) *(f32x4*)(g.C2 + o) = pack8f(dv);
                }
                if (ok) *(f32x4*)(g.C + o) = pack8f(v);
            }
        }
    };

#define G8_LD_A(mq)                                                                                              \
    _Pragma("unroll") for (int mbl = 0; mbl < 4; ++mbl) {                                                        \
        fa[mbl][0] = *(const bf16x8*)(smem + ra0 + (mq) * HALF_B + mbl * 2048);                                  \
        fa[mbl][1] = *(const bf16x8*)(smem + ra1 + (mq) * HALF_B + mbl * 2048);                                  \
    }
#define G8_LD_B(fb, nq)                                                                                          \
    _Pragma("unroll") for (int nbl = 0; nbl < 2; ++nbl) {                                                        \
        fb[nbl][0] = *(const bf16x8*)(smem + rb0 + (nq) * HALF_B + nbl * 2048);                                  \
        fb[nbl][1] = *(const bf16x8*)(smem + rb1 + (nq) * HALF_B + nbl * 2048);                                  \
    }
#define G8_MMA(mq, nq, fb)                                                                                       \
    __builtin_amdgcn_s_setprio(1);                                                                               \
    _Pragma("unroll") for (int kk = 0; kk < 2; ++kk)                                                             \
        _Pragma("unroll") for (int mbl = 0; mbl < 4; ++mbl)                                                      \
            _Pragma("unroll") for (int nbl = 0; nbl < 2; ++nbl)                                                  \
                acc[(mq) * 4 + mbl][(nq) * 2 + nbl] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(                   \
                    fb[nbl][kk], fa[mbl][kk], acc[(mq) * 4 + mbl][(nq) * 2 + nbl], 0, 0, 0);                     \
    __builtin_amdgcn_s_setprio(0);

    // ---- prologue: stream K-tile 0 complete, the first three half-tiles of K-tile 1 in flight ----
    int cp_i = 0, cp_kt = 0;
    int T = first;
    int m0c = (T / g.nx) * 256, n0c = (T - (T / g.nx) * g.nx) * 256;
    set_tile(first);
    dmaB(0); dmaA(0); dmaB(1); dmaA(1); advance();
    dmaB(0); dmaA(0); dmaB(1);
    init_acc(n0c);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    G8_BAR();
    if (wr == 1) G8_BAR();                       // the stagger: waves 4-7 run one barrier behind waves 0-3

    const int total_kt = my_n * g.nk;
    unsigned long long t_start = 0;
    if constexpr (DBG) t_start = __builtin_amdgcn_s_memtime();
    for (int s = 0; s < total_kt; ++s) {
        // phase 1: B-h0 + A-h0 fragments | A-h1 of the next K-tile | quadrant (m 0-63, n 0-31)
        G8_LD_B(fb0, 0);
        __builtin_amdgcn_sched_barrier(0);
        G8_LD_A(0);
        dmaA(1); advance();
        asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");       // the B-h0 reads have left LDS: B-h0 may be refilled next phase
        G8_BAR();
        G8_MMA(0, 0, fb0);
        G8_BAR();
        // phase 2: B-h1 fragments | B-h0 of K-tile + 2 | quadrant (m 0-63, n 32-63)
        G8_LD_B(fb1, 1);
        dmaB(0);
        G8_BAR();
        G8_MMA(0, 1, fb1);
        G8_BAR();
        // phase 3: A-h1 fragments | A-h0 of K-tile + 2 | quadrant (m 64-127, n 32-63)
        G8_LD_A(1);
        dmaA(0);
        G8_BAR();
        G8_MMA(1, 1, fb1);
        G8_BAR();
        // phase 4: B-h1 of K-tile + 2 | everything but the three youngest half-tiles has landed | quadrant (m 64-127, n 0-31)
        dmaB(1);
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        G8_BAR();
        G8_MMA(1, 0, fb0);
        G8_BAR();
        ra0 ^= BUF_B; ra1 ^= BUF_B; rb0 ^= BUF_B; rb1 ^= BUF_B;
        if (++cp_kt == g.nk) {
            unsigned long long t_loop = 0;
            if constexpr (DBG) t_loop = __builtin_amdgcn_s_memtime();
            epilogue(m0c, n0c);
            if constexpr (DBG) {
                const unsigned long long t_epi = __builtin_amdgcn_s_memtime();
                if ((tid & 255) == 0 && cp_i < 16) {
                    unsigned long long* d = g.dbg + ((long)(blockIdx.x * 2 + wr) * 16 + cp_i) * 4;
                    d[0] = t_start; d[1] = t_loop; d[2] = t_epi; d[3] = __builtin_amdgcn_s_memrealtime();
                }
                t_start = t_epi;
            }
            cp_kt = 0;
            if (++cp_i < my_n) {
                T = first + cp_i * per;
                const int mblk = T / g.nx;
                m0c = mblk * 256; n0c = (T - mblk * g.nx) * 256;
                init_acc(n0c);
            }
        }
    }
    if (wr == 0) G8_BAR();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the cursor's surplus DMAs must not outlive the workgroup's LDS
}

template <int MODE, bool SCALED>
int launch8(const G8Args& a, int nwg, hipStream_t stream) {
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute((const void*)gemm8_kernel<MODE, SCALED>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_B); attr = true; }
    hipLaunchKernelGGL((gemm8_kernel<MODE, SCALED>), dim3((unsigned)nwg), dim3(512), LDS_B, stream, a);
    PSELD_LAUNCH_CHECK("gemm8");
    return PSELD_OK;
}

unsigned long long* g_gemm8_dbg = nullptr;
}  // namespace

extern "C" void pseld_gemm8_set_debug_buffer(void* p) { g_gemm8_dbg = (unsigned long long*)p; }

int pseld_gemm8_supported(const Gemm8Desc& d) {
    if (d.K % 64 != 0 || d.K < 128 || d.M < 256 || d.N < 128 || d.N % 8 != 0) return 0;
    if (d.lda % 8 != 0 || d.ldb % 8 != 0 || d.ldc % 8 != 0 || (d.resid && d.ldr % 8 != 0) || (d.aux && d.ldaux % 8 != 0)) return 0;
    if ((long)d.M * d.lda * 2 >= (1L << 32) || (long)d.N * d.ldb * 2 >= (1L << 32) || d.M >= (1 << 24)) return 0;
    if (pseld_cdiv(d.N, 256) * 256 > BIAS_FLOATS) return 0;
    if ((((unsigned long)d.A | (unsigned long)d.B | (unsigned long)d.C | (unsigned long)d.C2 | (unsigned long)d.resid | (unsigned long)d.aux) & 15) != 0) return 0;
    if (d.resid && d.aux) return 0;
    if (d.gelu_dual && (d.resid || d.aux || d.rowscale || !d.C2)) return 0;
    return 1;
}

int pseld_gemm8_launch(const Gemm8Desc& d, hipStream_t stream) {
    G8Args a;
    a.A = (const char*)d.A; a.B = (const char*)d.B; a.C = (bf16_t*)d.C; a.C2 = (bf16_t*)d.C2;
    a.bias = d.bias; a.resid = (const bf16_t*)d.resid; a.aux = (const bf16_t*)d.aux; a.rowscale = d.rowscale;
    a.M = d.M; a.N = d.N; a.K = d.K; a.lda = d.lda; a.ldb = d.ldb; a.ldc = d.ldc; a.ldr = d.ldr; a.ldaux = d.ldaux;
    a.rows_per_scale = d.rows_per_scale > 0 ? d.rows_per_scale : 1;
    a.inv_rps = 1.0f / (float)a.rows_per_scale;
    a.nx = pseld_cdiv(d.N, 256);
    a.ntiles = a.nx * pseld_cdiv(d.M, 256);
    a.nk = d.K / 64;
    int nwg = (a.ntiles + 7) / 8 * 8;
    if (nwg > 256) nwg = 256;
    a.dbg = g_gemm8_dbg;
    if (a.dbg) {          // diagnostic build of three epilogue kinds: stamps to [workgroup][wave group][tile < 16][4]
        auto go = [&](auto kern) {
            (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_B);
            hipLaunchKernelGGL(kern, dim3((unsigned)nwg), dim3(512), LDS_B, stream, a);
            return PSELD_OK;
        };
        if (d.gelu_dual) return go(gemm8_kernel<G8_GELU_DUAL, false, true>);
        if (d.resid && d.rowscale) return go(gemm8_kernel<G8_RESID, true, true>);
        if (!d.resid && !d.aux && !d.rowscale) return go(gemm8_kernel<G8_PLAIN, false, true>);
    }
    const bool sc = d.rowscale != nullptr;
    if (d.gelu_dual) return launch8<G8_GELU_DUAL, false>(a, nwg, stream);
    if (d.resid) return sc ? launch8<G8_RESID, true>(a, nwg, stream) : launch8<G8_RESID, false>(a, nwg, stream);
    if (d.aux) return sc ? launch8<G8_MULAUX, true>(a, nwg, stream) : launch8<G8_MULAUX, false>(a, nwg, stream);
    return sc ? launch8<G8_PLAIN, true>(a, nwg, stream) : launch8<G8_PLAIN, false>(a, nwg, stream);
}
