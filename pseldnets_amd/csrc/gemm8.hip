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
constexpr int BIAS_FLOATS = 4672;         // the product's whole bias vector (padded to the tile grid) lives in LDS
constexpr int LDS_B = RING_B + BIAS_FLOATS * 4;

// G8_TOUCH=1 (build-time A/B): pull the next tile's A block into L2 a whole epilogue ahead of its LDS-DMA. Measured NOT to help (in-process
// against the untouched 128 x 192 kernel on the same box: stage-2 qkv forward 53.0 -> 58.1 us while the old kernel read 60.8 -> 64.7, K loop
// 3 512 -> 3 657 cycles per K-tile, epilogue +600 cycles): the short-K loop is not waiting for HBM-served A lines. Off.
#ifndef G8_TOUCH
#define G8_TOUCH 0
#endif
enum { G8_PLAIN = 0, G8_RESID = 1, G8_MULAUX = 2, G8_GELU_DUAL = 3 };   // x SCALED (DropPath factor per token row)

// s_waitcnt vmcnt(0) the compiler's own wait bookkeeping sees (gfx9 encoding: vmcnt 0, expcnt 7, lgkmcnt 15)
#define G8_WAIT_VM0() __builtin_amdgcn_s_waitcnt(0x0F70)

struct G8Args {
    const char* A; const char* B; bf16_t* C; bf16_t* C2;
    const float* bias; const bf16_t* resid; const bf16_t* aux; const float* rowscale;
    int M, N, K, lda, ldb, ldc, ldr, ldaux, rows_per_scale;
    float inv_rps;
    int nx, ntiles, nk;
    int store_m;              // rows that are stored (== M; 0 in the PSELD_GEMM8_NOSTORE timing experiment)
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

// L2 "touch": 64 lanes x 4 B from 64 different 128-byte lines, dumped into a scratch corner of LDS - brings the lines into this XCD's L2
// without a destination register (an ordinary load whose result nobody reads could land in a register the compiler has reused)
__device__ __forceinline__ void g8_touch(unsigned lds_dst, const void* sbase, unsigned voff) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, %3\n\ts_mov_b32 m0, %0"
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

// NB = 16-column accumulator blocks per wave: 4 -> 256 x 256 tile (wave 128 x 64), 3 -> 256 x 192 (wave 128 x 48: the second column
// quadrant is one block wide, B-h1 is a 64-row image, and every N of HTS-AT is a multiple of 192)
template <int MODE, bool SCALED, int NB, bool DBG = false>
__global__ __launch_bounds__(512, 2) void gemm8_kernel(const G8Args g) {
    constexpr int WN = NB * 16, BN = 4 * WN;
    constexpr int NB1 = NB - 2;                  // blocks in the second column quadrant
    // Just-in-time waits: a half-tile is waited for in the phase BEFORE the one that reads it, so the five youngest half-tiles stay in
    // flight at every wait (B-h1 is NB1 instructions per wave, the others two): every load has five phases to land (three with the
    // single wait per K-tile of the guide's template - too few for operands that come from HBM rather than L2)
    constexpr int VM_P4 = 6 + 2 * NB1, VM_P1 = 8 + NB1, VM_P2 = 8 + NB1;
    constexpr int NST = MODE == G8_GELU_DUAL ? 32 : 16;      // stores of one epilogue per wave
    constexpr int NTOUCH = G8_TOUCH ? 2 : 0;                 // L2 touches of the next tile's A block per wave (issued in front of the epilogue)
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

    for (int i = tid; i < g.nx * BN; i += 512) bias_s[i] = (g.bias && i < g.N) ? g.bias[i] : 0.f;
    __syncthreads();

    // ---- fragment read addresses (buffer 0): lane reads row l15 of a 16-row block, 16-byte chunk (4 kk + q) ^ ((row >> 1) & 7) ----
    const int sw = (lane >> 1) & 7;
    const unsigned c0 = (unsigned)((q ^ sw) << 4);
    unsigned ra0 = (unsigned)(wr * 8192 + l15 * 128) + c0, ra1 = ra0 ^ 64;
    unsigned rb0 = (unsigned)(2 * HALF_B + wc * 4096 + l15 * 128) + c0, rb1 = rb0 ^ 64;                       // B-h0: rows wc*32 + nbl*16 + l15
    unsigned rc0 = (unsigned)(3 * HALF_B + wc * (NB1 * 2048) + l15 * 128) + c0, rc1 = rc0 ^ 64;               // B-h1: rows wc*(16 NB1) + nbl*16 + l15

    // ---- LDS-DMA source offsets of the load cursor's tile: [half][instruction] ----
    unsigned offA[2][2], offB[2][2];
    auto set_tile = [&](int T) {
        const int mblk = T / g.nx, nblk = T - mblk * g.nx;
        const int m0 = mblk * 256, n0 = nblk * BN;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int rho = wave * 16 + j * 8 + (lane >> 3);                  // image row this lane fills
            const int ch = (lane & 7) ^ ((rho >> 1) & 7);                     // source chunk (swizzle on the source)
            const int i = rho & 15;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int tok = min(m0 + (rho >> 6) * 128 + h * 64 + (rho & 63), g.M - 1);
                offA[h][j] = (unsigned)tok * (unsigned)(g.lda * 2) + (unsigned)(ch * 16);
            }
            // B-h0 (and B-h1 of the 64-column wave tile): image row wc*32 + nbl*16 + i <- weight row wc*WN + 32 h + 8 (i>>2) + 4 nbl + (i&3)
            const int nb0 = n0 + (rho >> 5) * WN + 8 * (i >> 2) + 4 * ((rho >> 4) & 1) + (i & 3);
            offB[0][j] = (unsigned)min(nb0, g.N - 1) * (unsigned)(g.ldb * 2) + (unsigned)(ch * 16);
            if constexpr (NB == 4) offB[1][j] = (unsigned)min(nb0 + 32, g.N - 1) * (unsigned)(g.ldb * 2) + (unsigned)(ch * 16);
        }
        if constexpr (NB == 3) {      // B-h1, one block per wave column: 64 image rows wc*16 + i <- weight row wc*48 + 32 + i; one instruction per wave
            const int rho = wave * 8 + (lane >> 3);
            const int ch = (lane & 7) ^ ((rho >> 1) & 7);
            offB[1][0] = (unsigned)min(n0 + (rho >> 4) * WN + 32 + (rho & 15), g.N - 1) * (unsigned)(g.ldb * 2) + (unsigned)(ch * 16);
            offB[1][1] = 0;
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
        if (NB == 3 && h == 1) { g8_dma(lds_base + (unsigned)wave * 1024u + ld_buf + (unsigned)(3 * HALF_B), sb, offB[1][0]); return; }
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

    bf16x8 fa[4][2], fb0[2][2], fb1[NB1][2];
    f32x4 acc[8][NB];

    auto init_acc = [&](int n0) {
        const float* bp = bias_s + n0 + wc * WN;
        f32x4 b[NB];
        b[0] = *(const f32x4*)(bp + 8 * q); b[1] = *(const f32x4*)(bp + 8 * q + 4);
        if constexpr (NB == 4) { b[2] = *(const f32x4*)(bp + 32 + 8 * q); b[3] = *(const f32x4*)(bp + 32 + 8 * q + 4); }
        else b[2] = *(const f32x4*)(bp + 32 + 4 * q);
#pragma unroll
        for (int mb = 0; mb < 8; ++mb)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) acc[mb][nb] = b[nb];
    };

    // lane: token row m0 + wr*128 + mb*16 + l15; columns n0 + wc*WN + 8 q + {4 nbl + k} = acc[mb][nbl][k] (first quadrant, 16 bytes) and
    // n0 + wc*WN + 32 + 8 q + {4 nbl + k} = acc[mb][2 + nbl][k] (NB = 4) or n0 + wc*WN + 32 + 4 q + k = acc[mb][2][k] (NB = 3, 8 bytes)
    auto epilogue = [&](int m0, int n0) {
        constexpr bool HAS_X = MODE == G8_RESID || MODE == G8_MULAUX;
        constexpr int W1 = NB == 4 ? 8 : 4;                   // columns per lane in the second quadrant
        const int rowb = m0 + wr * 128 + l15;
        const int col0 = n0 + wc * WN + 8 * q, col1 = n0 + wc * WN + 32 + W1 * q;
        const int mlast = g.M - 1;
        const int c0c = min(col0, g.N - 8), c1c = min(col1, g.N - W1);
        const bf16_t* X = MODE == G8_RESID ? g.resid : g.aux;
        const int ldx = MODE == G8_RESID ? g.ldr : g.ldaux;
        f32x4 xv0[HAS_X ? 8 : 1];
        f32x4 xv1[HAS_X && NB == 4 ? 8 : 1];
        f32x2 xw1[HAS_X && NB == 3 ? 8 : 1];
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
                    xv0[mb] = *(const f32x4*)(X + (long)rowc * ldx + c0c);
                    if constexpr (NB == 4) xv1[mb] = *(const f32x4*)(X + (long)rowc * ldx + c1c);
                    else xw1[mb] = *(const f32x2*)(X + (long)rowc * ldx + c1c);
                }
            }
            G8_WAIT_VM0();
        }
        // v (W values): the fused options on fp32, one rounding to bf16
        auto fuse = [&](float* v, const float* x, float scm, int W) {
            if constexpr (MODE == G8_PLAIN) {
                if constexpr (SCALED) for (int k = 0; k < W; ++k) v[k] *= scm;
            } else if constexpr (MODE == G8_RESID) {
                for (int k = 0; k < W; ++k) v[k] = SCALED ? fmaf(v[k], scm, x[k]) : v[k] + x[k];
            } else if constexpr (MODE == G8_MULAUX) {
                for (int k = 0; k < W; ++k) v[k] *= SCALED ? x[k] * scm : x[k];
            }
        };
#pragma unroll
        for (int mb = 0; mb < 8; ++mb) {
            const int row = rowb + mb * 16;
            const long orow = (long)min(row, mlast) * g.ldc;
            const float scm = SCALED ? sc[SCALED ? mb : 0] : 1.f;
            {   // first quadrant: 8 columns
                const bool ok = row < g.store_m && col0 < g.N;
                float v[8], x[8];
#pragma unroll
                for (int k = 0; k < 4; ++k) { v[k] = acc[mb][0][k]; v[4 + k] = acc[mb][1][k]; }
                if constexpr (HAS_X) unpack8f(xv0[mb], x);
                if constexpr (MODE == G8_GELU_DUAL) {
                    float dv[8];
#pragma unroll
                    for (int k = 0; k < 8; k += 2) {
                        f32x2 xx = {v[k], v[k + 1]}, yy, dd;
                        gelu_both2(xx, yy, dd);
                        v[k] = yy[0]; v[k + 1] = yy[1]; dv[k] = dd[0]; dv[k + 1] = dd[1];
                    }
                    if (ok) *(f32x4*)(g.C2 + orow + c0c) = pack8f(dv);
                } else {
#pragma unroll
                    for (int once = 0; once < 1; ++once) fuse(v, x, scm, 8);
                }
                if (ok) *(f32x4*)(g.C + orow + c0c) = pack8f(v);
            }
            {   // second quadrant: 8 (NB = 4) or 4 (NB = 3) columns
                const bool ok = row < g.store_m && col1 < g.N;
                float v[8], x[8];
#pragma unroll
                for (int k = 0; k < 4; ++k) { v[k] = acc[mb][2][k]; v[4 + k] = NB == 4 ? acc[mb][NB - 1][k] : 0.f; }
                if constexpr (HAS_X) {
                    if constexpr (NB == 4) unpack8f(xv1[mb], x);
                    else {
                        const bf16x4 xb = __builtin_bit_cast(bf16x4, xw1[mb]);
#pragma unroll
                        for (int k = 0; k < 4; ++k) x[k] = (float)xb[k];
                    }
                }
                float dv[8];
                if constexpr (MODE == G8_GELU_DUAL) {
#pragma unroll
                    for (int k = 0; k < W1; k += 2) {
                        f32x2 xx = {v[k], v[k + 1]}, yy, dd;
                        gelu_both2(xx, yy, dd);
                        v[k] = yy[0]; v[k + 1] = yy[1]; dv[k] = dd[0]; dv[k + 1] = dd[1];
                    }
                } else {
#pragma unroll
                    for (int once = 0; once < 1; ++once) fuse(v, x, scm, W1);
                }
                if constexpr (NB == 4) {
                    if constexpr (MODE == G8_GELU_DUAL) { if (ok) *(f32x4*)(g.C2 + orow + c1c) = pack8f(dv); }
                    if (ok) *(f32x4*)(g.C + orow + c1c) = pack8f(v);
                } else {
                    bf16x4 pv, pd;
#pragma unroll
                    for (int k = 0; k < 4; ++k) { pv[k] = (bf16_t)v[k]; pd[k] = (bf16_t)dv[k]; }
                    if constexpr (MODE == G8_GELU_DUAL) { if (ok) *(bf16x4*)(g.C2 + orow + c1c) = pd; }
                    if (ok) *(bf16x4*)(g.C + orow + c1c) = pv;
                }
            }
        }
    };

#define G8_LD_A(mq)                                                                                              \
    _Pragma("unroll") for (int mbl = 0; mbl < 4; ++mbl) {                                                        \
        fa[mbl][0] = *(const bf16x8*)(smem + ra0 + (mq) * HALF_B + mbl * 2048);                                  \
        fa[mbl][1] = *(const bf16x8*)(smem + ra1 + (mq) * HALF_B + mbl * 2048);                                  \
    }
#define G8_LD_B0()                                                                                               \
    _Pragma("unroll") for (int nbl = 0; nbl < 2; ++nbl) {                                                        \
        fb0[nbl][0] = *(const bf16x8*)(smem + rb0 + nbl * 2048);                                                 \
        fb0[nbl][1] = *(const bf16x8*)(smem + rb1 + nbl * 2048);                                                 \
    }
#define G8_LD_B1()                                                                                               \
    _Pragma("unroll") for (int nbl = 0; nbl < NB1; ++nbl) {                                                      \
        fb1[nbl][0] = *(const bf16x8*)(smem + rc0 + nbl * 2048);                                                 \
        fb1[nbl][1] = *(const bf16x8*)(smem + rc1 + nbl * 2048);                                                 \
    }
#define G8_MMA(mq, nq, fb)                                                                                       \
    __builtin_amdgcn_s_setprio(1);                                                                               \
    _Pragma("unroll") for (int kk = 0; kk < 2; ++kk)                                                             \
        _Pragma("unroll") for (int mbl = 0; mbl < 4; ++mbl)                                                      \
            _Pragma("unroll") for (int nbl = 0; nbl < ((nq) == 0 ? 2 : NB1); ++nbl)                              \
                acc[(mq) * 4 + mbl][(nq) * 2 + nbl] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(                   \
                    fb[nbl][kk], fa[mbl][kk], acc[(mq) * 4 + mbl][(nq) * 2 + nbl], 0, 0, 0);                     \
    __builtin_amdgcn_s_setprio(0);

    // ---- prologue: stream K-tile 0 complete, the first three half-tiles of K-tile 1 in flight ----
    int cp_i = 0, cp_kt = 0;
    int T = first;
    int m0c = (T / g.nx) * 256, n0c = (T - (T / g.nx) * g.nx) * BN;
    set_tile(first);
    dmaB(0); dmaA(0); dmaB(1); dmaA(1); advance();
    dmaB(0); dmaA(0); dmaB(1);
    init_acc(n0c);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VM_P4) : "memory");
    G8_BAR();
    if (wr == 1) G8_BAR();                       // the stagger: waves 4-7 run one barrier behind waves 0-3

    const int total_kt = my_n * g.nk;
    unsigned long long t_start = 0;
    if constexpr (DBG) t_start = __builtin_amdgcn_s_memtime();
    for (int s = 0; s < total_kt; ++s) {
        // phase 1: B-h0 + A-h0 fragments | A-h1 of the next K-tile | quadrant (m 0-63, n 0-31)
        G8_LD_B0();
        __builtin_amdgcn_sched_barrier(0);
        G8_LD_A(0);
        dmaA(1); advance();
        asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");       // the B-h0 reads have left LDS: B-h0 may be refilled next phase
        // B-h1 of this K-tile has landed (read next phase); behind an epilogue its stores sit in the queue too and may stay there
        if (cp_kt == 0 && cp_i > 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VM_P1 + NST + NTOUCH) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VM_P1) : "memory");
        G8_BAR();
        G8_MMA(0, 0, fb0);
        G8_BAR();
        // phase 2: B-h1 fragments | B-h0 of K-tile + 2 | quadrant (m 0-63, n 32-63)
        G8_LD_B1();
        dmaB(0);
        if (cp_kt == 0 && cp_i > 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VM_P2 + NST + NTOUCH) : "memory");   // A-h1 of this K-tile has landed
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VM_P2) : "memory");
        G8_BAR();
        G8_MMA(0, 1, fb1);
        G8_BAR();
        // phase 3: A-h1 fragments | A-h0 of K-tile + 2 | quadrant (m 64-127, n 32-63)
        G8_LD_A(1);
        dmaA(0);
        G8_BAR();
        G8_MMA(1, 1, fb1);
        G8_BAR();
        // phase 4: B-h1 of K-tile + 2 | B-h0 and A-h0 of the next K-tile have landed | quadrant (m 64-127, n 0-31)
        dmaB(1);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VM_P4) : "memory");
        G8_BAR();
        G8_MMA(1, 0, fb0);
        G8_BAR();
        ra0 ^= BUF_B; ra1 ^= BUF_B; rb0 ^= BUF_B; rb1 ^= BUF_B; rc0 ^= BUF_B; rc1 ^= BUF_B;
        if (++cp_kt == g.nk) {
            unsigned long long t_loop = 0;
            if constexpr (DBG) t_loop = __builtin_amdgcn_s_memtime();
            if constexpr (NTOUCH > 0) {
                // The K-tiles 2 .. 5 of the NEXT tile's A block (its first two are already in flight) are pulled into this XCD's L2 now,
                // a whole epilogue ahead of their LDS-DMA: a lone workgroup keeps <= 80 KB in flight (the ring), and at the ~4k cycles an
                // HBM-served line takes under load that is 17 B/clk per CU, half of what the loop needs - from L2 the same ring feeds it.
                // One 4-byte request per 128-byte line; thread = (row tid & 255, K-tile 2 + (tid >> 8) + 2 j); lines past K wrap to K-tile 0.
                const int Tn = first + min(cp_i + 1, my_n - 1) * per;
                const int rown = min((Tn / g.nx) * 256 + (tid & 255), g.M - 1);
#pragma unroll
                for (int j = 0; j < NTOUCH; ++j) {
                    int kt = 2 + (tid >> 8) + 2 * j;
                    kt = kt < g.nk ? kt : 0;
                    g8_touch(lds_base + (unsigned)(RING_B + (BIAS_FLOATS - 64) * 4), g.A, (unsigned)rown * (unsigned)(g.lda * 2) + (unsigned)(kt * 128));
                }
            }
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
                m0c = mblk * 256; n0c = (T - mblk * g.nx) * BN;
                init_acc(n0c);
            }
        }
    }
    if (wr == 0) G8_BAR();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the cursor's surplus DMAs must not outlive the workgroup's LDS
}

const char* g_gemm8_symbol = "";     // the instantiation the last launch ran, as rocprofv3 prints it (measurement aid)

template <int MODE, bool SCALED, int NB>
int launch8(const G8Args& a, int nwg, hipStream_t stream) {
    static const char* const names[4][2][2] = {
        {{"gemm8_kernel<0, false, 3, false>", "gemm8_kernel<0, false, 4, false>"}, {"gemm8_kernel<0, true, 3, false>", "gemm8_kernel<0, true, 4, false>"}},
        {{"gemm8_kernel<1, false, 3, false>", "gemm8_kernel<1, false, 4, false>"}, {"gemm8_kernel<1, true, 3, false>", "gemm8_kernel<1, true, 4, false>"}},
        {{"gemm8_kernel<2, false, 3, false>", "gemm8_kernel<2, false, 4, false>"}, {"gemm8_kernel<2, true, 3, false>", "gemm8_kernel<2, true, 4, false>"}},
        {{"gemm8_kernel<3, false, 3, false>", "gemm8_kernel<3, false, 4, false>"}, {"gemm8_kernel<3, true, 3, false>", "gemm8_kernel<3, true, 4, false>"}}};
    g_gemm8_symbol = names[MODE][SCALED ? 1 : 0][NB - 3];
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute((const void*)gemm8_kernel<MODE, SCALED, NB>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_B); attr = true; }
    hipLaunchKernelGGL((gemm8_kernel<MODE, SCALED, NB>), dim3((unsigned)nwg), dim3(512), LDS_B, stream, a);
    PSELD_LAUNCH_CHECK("gemm8");
    return PSELD_OK;
}
template <int NB>
int launch8_mode(const Gemm8Desc& d, const G8Args& a, int nwg, hipStream_t stream) {
    const bool sc = d.rowscale != nullptr;
    if (a.dbg) {          // diagnostic build of three epilogue kinds: stamps to [workgroup][wave group][tile < 16][4]
        auto go = [&](auto kern) {
            (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_B);
            hipLaunchKernelGGL(kern, dim3((unsigned)nwg), dim3(512), LDS_B, stream, a);
            return PSELD_OK;
        };
        if (d.gelu_dual) return go(gemm8_kernel<G8_GELU_DUAL, false, NB, true>);
        if (d.resid && sc) return go(gemm8_kernel<G8_RESID, true, NB, true>);
        if (!d.resid && !d.aux && !sc) return go(gemm8_kernel<G8_PLAIN, false, NB, true>);
    }
    if (d.gelu_dual) return launch8<G8_GELU_DUAL, false, NB>(a, nwg, stream);
    if (d.resid) return sc ? launch8<G8_RESID, true, NB>(a, nwg, stream) : launch8<G8_RESID, false, NB>(a, nwg, stream);
    if (d.aux) return sc ? launch8<G8_MULAUX, true, NB>(a, nwg, stream) : launch8<G8_MULAUX, false, NB>(a, nwg, stream);
    return sc ? launch8<G8_PLAIN, true, NB>(a, nwg, stream) : launch8<G8_PLAIN, false, NB>(a, nwg, stream);
}

unsigned long long* g_gemm8_dbg = nullptr;
}  // namespace

extern "C" void pseld_gemm8_set_debug_buffer(void* p) { g_gemm8_dbg = (unsigned long long*)p; }
const char* pseld_gemm8_last_symbol() { return g_gemm8_symbol; }

int pseld_gemm8_supported(const Gemm8Desc& d) {
    // (any M: a layer must take the same kernel at every batch size - the bit-exact batch-independence and additivity tests; rows past M are
    //  clamped on load and never stored. The recurrent products with M <= 64 keep their skinny kernel: pseld_gemm asks it first)
    if (d.K % 64 != 0 || d.K < 128 || d.M < 1 || d.N < 128 || d.N % 8 != 0) return 0;
    if (d.lda % 8 != 0 || d.ldb % 8 != 0 || d.ldc % 8 != 0 || (d.resid && d.ldr % 8 != 0) || (d.aux && d.ldaux % 8 != 0)) return 0;
    if ((long)d.M * d.lda * 2 >= (1L << 32) || (long)d.N * d.ldb * 2 >= (1L << 32) || d.M >= (1 << 24)) return 0;
    if (pseld_cdiv(d.N, 192) * 192 > BIAS_FLOATS - 64 || pseld_cdiv(d.N, 256) * 256 > BIAS_FLOATS - 64) return 0;     // (the last 256 B: sink of the L2 touches)
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
    a.nk = d.K / 64;
    a.dbg = g_gemm8_dbg;
    { const char* en = getenv("PSELD_GEMM8_NOSTORE"); a.store_m = (en && en[0] == '1') ? 0 : d.M; }     // (timing experiment: results are not written)
    // tile width: 192 when that wastes fewer columns / fills the rounds better (PSELD_GEMM8_BN=256 / 192 forces one: A/B knob)
    const char* eb = getenv("PSELD_GEMM8_BN");
    int bn = eb ? atoi(eb) : 0;
    if (bn != 256 && bn != 192) {
        auto cost = [&](int w) {                      // rounds x (loop cost of one tile ~ DMA bytes per K-tile)
            const long tiles = (long)pseld_cdiv(d.N, w) * pseld_cdiv(d.M, 256);
            return (double)((tiles + 255) / 256) * (256 + w);
        };
        bn = cost(192) < cost(256) ? 192 : 256;
    }
    a.nx = pseld_cdiv(d.N, bn);
    a.ntiles = a.nx * pseld_cdiv(d.M, 256);
    int nwg = (a.ntiles + 7) / 8 * 8;
    if (nwg > 256) nwg = 256;
    return bn == 192 ? launch8_mode<3>(d, a, nwg, stream) : launch8_mode<4>(d, a, nwg, stream);
}
