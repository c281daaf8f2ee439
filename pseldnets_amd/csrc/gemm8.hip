// Persistent eight-phase bf16 GEMM for the Linear layers of HTS-AT with K >= 192 (stages 1-3, merges, head) on gfx950.
//
// C[M,N] = A[M,K] B[N,K]^T with the fused epilogues of pseld_gemm (include/pseld_hip.h): + bias, x DropPath factor, x aux, + residual,
// GELU pair. Replaces, for products with K >= 192 (htsat.py:118,140 qkv / proj, model_utilities.py:166-170 fc1 / fc2, htsat.py:309
// PatchMerging.reduction, accdoa.py:230 head), the 128 x 192 three-workgroups-per-CU kernel of gemm.hip, whose slice loop is capped
// by the CU's LDS-DMA rate.
//
// Structure (cdna_hip_programming.md, "The 256^2 8-phase template"):
//  * ONE workgroup of 8 waves per CU, persistent over its list of 256 x 256 output tiles; waves 2 (M) x 4 (N), 128 x 64 per wave as
//    8 x 4 accumulator blocks of v_mfma_f32_16x16x32_bf16 computed TRANSPOSED (weight rows on the accumulator rows, tokens on the
//    lanes), and the B image -> weight row map gives a lane 16 (12) consecutive output columns of one token row. Adjacent token rows
//    (lanes) trade halves through one DPP quad_perm, so every epilogue store / residual load instruction covers 8 rows x 128 (96)
//    contiguous bytes - whole cache lines straight from the accumulators, no LDS, no barrier (the CU's vector-memory path is paced
//    by the lines an instruction touches: 16 rows x 64 B per instruction measured 4.1k cycles per epilogue, 8 x 128 B 2.6k).
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
#include <type_traits>
#include <stdio.h>

namespace {

typedef __attribute__((address_space(3))) void* lds_vptr8;

constexpr int HALF_B = 16384;             // one half-tile image: 128 rows x 128 B (64 bf16 of K)
constexpr int BUF_B = 4 * HALF_B;         // one K-tile: A-h0 | A-h1 | B-h0 | B-h1
constexpr int RING_B = 2 * BUF_B;         // 128 KiB
constexpr int BIAS_FLOATS = 4672;         // the product's whole bias vector (padded to the tile grid) lives in LDS
constexpr int LDS_B = RING_B + BIAS_FLOATS * 4;

enum { G8_PLAIN = 0, G8_RESID = 1, G8_MULAUX = 2, G8_GELU_DUAL = 3 };     // x SCALED (DropPath factor per token row)

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

__device__ __forceinline__ unsigned g8_pack2(float a, float b) {
    bf16x2 t;
    t[0] = (bf16_t)a; t[1] = (bf16_t)b;
    return __builtin_bit_cast(unsigned, t);
}
__device__ __forceinline__ void g8_unpack2(unsigned p, float& a, float& b) {
    const bf16x2 t = __builtin_bit_cast(bf16x2, p);
    a = (float)t[0]; b = (float)t[1];
}
// the value the neighbouring lane (lane ^ 1) holds: DPP quad_perm [1, 0, 3, 2]
__device__ __forceinline__ unsigned g8_swap1(unsigned v) { return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true); }
// 16- / 12-byte pieces (PD = 4 / 3 dwords) at 4-byte aligned addresses
template <int PD> struct G8Piece { unsigned d[PD]; };
template <int PD>
__device__ __forceinline__ void g8_store_nt(bf16_t* p, const G8Piece<PD>& v) {
    static_assert(PD == 4 || PD == 3, "non-temporal pieces exist for the 256- and 192-column tiles");
    if constexpr (PD == 4) {
        typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
        __builtin_nontemporal_store(u32x4_t{v.d[0], v.d[1], v.d[2], v.d[3]}, (u32x4_t*)p);
    } else {
        typedef __attribute__((ext_vector_type(3))) unsigned u32x3_t;
        const u32x3_t x = {v.d[0], v.d[1], v.d[2]};
        // (inline asm: the compiler's hazard recognizer does not see a store here - the wait states a >64-bit store needs before its data
        //  registers may be overwritten are spelled out)
        asm volatile("global_store_dwordx3 %0, %1, off nt\n\ts_nop 1" ::"v"(p), "v"(x) : "memory");
    }
}

#define G8_BAR()                                  \
    do {                                          \
        __builtin_amdgcn_sched_barrier(0);        \
        __builtin_amdgcn_s_barrier();             \
        __builtin_amdgcn_sched_barrier(0);        \
    } while (0)

// NB = 16-column accumulator blocks per wave: 4 -> 256-column tile (wave 128 x 64), 3 -> 192 (wave 128 x 48: the second column
// quadrant is one block wide, B-h1 is a 64-row image, and every N of HTS-AT is a multiple of 192).
// MBQ = 16-row accumulator blocks per wave and row quadrant: 4 -> 256-row tile (wave 128 rows), 2 -> 128-row tile (wave 64 rows, the A
// half images are 64 rows = ONE LDS-DMA instruction per wave). Same K order per output element at every (NB, MBQ): the four tile
// shapes give the same bits, so the launch may pick by grid fill (under-filled launches: the 32-chunk step, stage 3).
// (Round 6 removed three built-and-measured-equal variants of this kernel - the 128 x 192 tile packed for two workgroups per CU, the
//  128 x 384 row-spanning tile and its LayerNorm forward / backward epilogues: docs/EXPERIMENTS.md, rounds 4-5, has their numbers.)
template <int MODE, bool SCALED, int NB, int MBQ = 4, bool DBG = false>
__global__ __launch_bounds__(512, 2) void gemm8_kernel(const G8Args g) {
    constexpr int WN = NB * 16, BN = 4 * WN;
    constexpr int NB0 = 2;                       // blocks in the first column quadrant = LDS-DMA instructions per wave of B-h0 (128 rows)
    constexpr int NB1 = NB - NB0;                // ... in the second = instructions per wave of B-h1
    constexpr int OFF_A1 = HALF_B;               // A-h1 image (A-h0 at 0)
    constexpr int OFF_B0 = 2 * HALF_B, OFF_B1 = 3 * HALF_B;
    constexpr int BM = 64 * MBQ;                 // tile rows
    constexpr int MBN = 2 * MBQ;                 // 16-row blocks per wave
    constexpr int MA = MBQ / 2;                  // LDS-DMA instructions per wave and A half image (32 MBQ rows, 8 per instruction and wave)
    // Just-in-time waits: a half-tile is waited for in the phase BEFORE the one that reads it, so the five youngest half-tiles stay in
    // flight at every wait (B-h1 is NB1 instructions per wave, the others two): every load has five phases to land (three with the
    // single wait per K-tile of the guide's template - too few for operands that come from HBM rather than L2).
    // Measured and NOT adopted: L2 touches of the next tile's A block a whole epilogue ahead of its LDS-DMA (stage-2 qkv forward 53.0 -> 58.1 us
    // against 60.8 -> 64.7 for the 128 x 192 kernel in the same process, K loop 3 512 -> 3 657 cycles per K-tile), and a touch cursor running 2 / 4
    // K-tiles ahead of the LDS-DMA cursor through the whole stream, measured with COLD operands (tools/gemm8_check.py COLD=1: every launch reads
    // buffers the memory-side cache has not seen, as in the step): stage 2 + 3 forward + input gradient 4.13 -> 4.45 / 4.51 ms per step; letting the stores of an
    // epilogue stay outstanding over two more waits (no change); one extra barrier per wave group so that the two groups' epilogues run side
    // by side instead of one after the other (no change: the epilogue is paced by the CU's vector-memory path, which the groups share).
    // (instructions per wave behind the awaited half-tile in the in-order queue - A halves MA, B-h0 two, B-h1 NB1:
    //  phase 4 waits for A-h0(k+1): B-h1(k+1) A-h1(k+1) B-h0(k+2) A-h0(k+2) B-h1(k+2) stay; phase 1 for B-h1(k): A-h1(k) B-h0(k+1) A-h0(k+1)
    //  B-h1(k+1) A-h1(k+1); phase 2 for A-h1(k): B-h0(k+1) A-h0(k+1) B-h1(k+1) A-h1(k+1) B-h0(k+2))
    constexpr int VM_P4 = 2 * NB1 + 2 * MA + NB0, VM_P1 = 3 * MA + NB0 + NB1, VM_P2 = 2 * NB0 + 2 * MA + NB1;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* bias_s = (float*)(smem + RING_B);
    // (diagnostic instantiation) real-time stamps of the start-up pieces, and the end of K-tiles 0..7 of this workgroup's SECOND tile
    unsigned long long r_entry = 0, r_bias = 0, r_loop = 0, kt0 = 0, kt1 = 0, kt2 = 0, kt3 = 0, kt4 = 0, kt5 = 0, kt6 = 0, kt7 = 0;
    if constexpr (DBG) r_entry = __builtin_amdgcn_s_memrealtime();
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
    if constexpr (DBG) r_bias = __builtin_amdgcn_s_memrealtime();

    // ---- fragment read addresses (buffer 0): lane reads row l15 of a 16-row block, 16-byte chunk (4 kk + q) ^ ((row >> 1) & 7) ----
    const int lane_m = lane;
    unsigned ra0, ra1, rb0, rb1, rc0, rc1;
    auto set_frag_addrs = [&](unsigned par) {
        const int l15m = lane_m & 15, qm = lane_m >> 4, sw = (lane_m >> 1) & 7;
        const unsigned c0 = (unsigned)((qm ^ sw) << 4);
        ra0 = ((unsigned)(wr * (MBQ * 2048) + l15m * 128) + c0) ^ par; ra1 = ra0 ^ 64;                      // A halves: rows wr*16 MBQ + mbl*16 + l15
        rb0 = ((unsigned)(OFF_B0 + wc * (NB0 * 2048) + l15m * 128) + c0) ^ par; rb1 = rb0 ^ 64;           // B-h0: rows wc*(16 NB0) + nbl*16 + l15
        rc0 = ((unsigned)(OFF_B1 + wc * (NB1 * 2048) + l15m * 128) + c0) ^ par; rc1 = rc0 ^ 64;           // B-h1: rows wc*(16 NB1) + nbl*16 + l15
    };
    set_frag_addrs(0u);

    // ---- LDS-DMA source offsets of the load cursor's tile: [half][instruction] ----
    unsigned offA[2][MA], offB[2][NB0];
    auto set_tile = [&](int T) {
        const int mblk = T / g.nx, nblk = T - mblk * g.nx;
        const int m0 = mblk * BM, n0 = nblk * BN;
#pragma unroll
        for (int j = 0; j < MA; ++j) {
            const int rho = wave * (8 * MA) + j * 8 + (lane_m >> 3);            // A image row this lane fills (of 32 MBQ per half)
            const int ch = (lane_m & 7) ^ ((rho >> 1) & 7);                   // source chunk (swizzle on the source)
            const int wq = rho / (16 * MBQ), within = rho - wq * (16 * MBQ);  // wave row group, row inside its half
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int tok = min(m0 + wq * (32 * MBQ) + h * (16 * MBQ) + within, g.M - 1);
                offA[h][j] = (unsigned)tok * (unsigned)(g.lda * 2) + (unsigned)(ch * 16);
            }
        }
        // B-h0 (blocks b = nbl < NB0) and B-h1 (b = NB0 + nbl) of a wave's WN-column strip: image row wc*(16 NBh) + nbl*16 + i <- weight row
        // wc*WN + 4 NB (i>>2) + 4 b + (i&3): accumulator row i = 4 q + k of block b is column 4 NB q + 4 b + k of the strip. A half image is
        // 64 NBh rows = NBh instructions per wave (8 rows each); instruction (wave, j) fills rows (wave NBh + j) 8 .. + 7
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int NBH = h ? NB1 : NB0;
#pragma unroll
            for (int j = 0; j < (h ? NB1 : NB0); ++j) {
                const int rho = (wave * NBH + j) * 8 + (lane_m >> 3);
                const int ch = (lane_m & 7) ^ ((rho >> 1) & 7);
                const int wcc = rho / (16 * NBH), within = rho - wcc * (16 * NBH);
                const int nbl = within >> 4, i = within & 15;
                const int row = n0 + wcc * WN + 4 * NB * (i >> 2) + 4 * ((h ? NB0 : 0) + nbl) + (i & 3);
                offB[h][j] = (unsigned)min(row, g.N - 1) * (unsigned)(g.ldb * 2) + (unsigned)(ch * 16);
            }
        }
    };
    const unsigned lds_base = (unsigned)(unsigned long)(lds_vptr8)smem;
    int ld_i = 0, ld_kt = 0;
    unsigned ld_buf = 0;
    const unsigned dst_a = lds_base + (unsigned)wave * (unsigned)(1024 * MA);
    auto dmaA = [&](int h) {
        const char* sb = g.A + ld_kt * 128;
#pragma unroll
        for (int j = 0; j < MA; ++j) g8_dma(dst_a + ld_buf + (unsigned)(h * OFF_A1 + j * 1024), sb, offA[h][j]);
    };
    auto dmaB = [&](int h) {
        const char* sb = g.B + ld_kt * 128;
        if (h == 0) {
#pragma unroll
            for (int j = 0; j < NB0; ++j) g8_dma(lds_base + ld_buf + (unsigned)(OFF_B0 + (wave * NB0 + j) * 1024), sb, offB[0][j]);
        } else {
#pragma unroll
            for (int j = 0; j < NB1; ++j) g8_dma(lds_base + ld_buf + (unsigned)(OFF_B1 + (wave * NB1 + j) * 1024), sb, offB[1][j]);
        }
    };
    auto advance = [&]() {       // past the end of the list the cursor re-reads the last tile (nobody reads those images): the
        ld_buf = (unsigned)BUF_B - ld_buf;         // vmcnt distance stays constant in the tail
        if (++ld_kt == g.nk) {
            ld_kt = 0;
            if (ld_i + 1 < my_n) { ++ld_i; set_tile(first + ld_i * per); }
        }
    };

    bf16x8 fa[MBQ][2], fb0[NB0][2], fb1[NB1][2];
    f32x4 acc[MBN][NB];

    auto init_acc = [&](int n0) {
        f32x4 b[NB];
        const float* bp = bias_s + n0 + wc * WN + 4 * NB * q;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) b[nb] = *(const f32x4*)(bp + 4 * nb);
#pragma unroll
        for (int mb = 0; mb < MBN; ++mb)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) acc[mb][nb] = b[nb];
    };

    // Lane (q, l15) holds, for token row m0 + wr*(32 MBQ) + 16 mb + l15, the W = 4 NB consecutive columns n0 + wc*WN + W q + (4 b + k) = acc[mb][b][k].
    // The lanes of two adjacent token rows (l15 even / odd: "A" / "B") trade halves: A ends up with columns [W q, W q + W/2) of BOTH rows,
    // B with [W q + W/2, W q + W) of both, so one store instruction writes the even rows of the pairs and one the odd rows - 8 rows x
    // (WN x 2) contiguous bytes each. The residual / aux operand is loaded in the same two-row pattern and traded back before the fp32 math.
    auto epilogue = [&](int m0, int n0) {
        constexpr bool HAS_X = MODE == G8_RESID || MODE == G8_MULAUX;
        constexpr int W = 4 * NB, HW = W / 2, PD = HW / 2;        // columns per lane, per piece; dwords per piece
        typedef G8Piece<PD> piece_t;
        const bool isB = (l15 & 1) != 0;
        const int rown = m0 + wr * (32 * MBQ) + l15;              // own token row (+ 16 mb)
        const int r1b = rown - (isB ? 1 : 0);                      // the pair's even row (+ 16 mb); the odd one is r1 + 1
        const int cL = n0 + wc * WN + W * q + (isB ? HW : 0);      // this lane's piece of both rows
        const bool colok = cL < g.N;                               // (N % 8 == 0, and N % 192 == 0 where HW = 6: pieces are whole or absent)
        const int cLc = min(cL, g.N - HW);
        const int mlast = g.M - 1;
        const bf16_t* X = MODE == G8_RESID ? g.resid : g.aux;
        const int ldx = MODE == G8_RESID ? g.ldr : g.ldaux;
        piece_t x1[HAS_X ? MBN : 1], x2[HAS_X ? MBN : 1];
        float sc[SCALED ? MBN : 1];
        // The loads of the epilogue (DropPath factor, residual / aux pieces) are unconditional (clamped addresses) and all consumed inside it:
        // no load of its own is pending at the loop's back edge, where the compiler would otherwise drain the LDS-DMA prefetch with a vmcnt(0)
        // in front of the next K-tile's fragment reads. 192-wide tile: all of them first, ONE wait. 256-wide tile: the 64 registers of all
        // eight row blocks do not fit beside 128 accumulators - four row blocks ahead, the compiler counts the waits.
        constexpr int PF = (NB == 4 && HAS_X && MBN > 4) ? 4 : MBN;
        auto load_mb = [&](int mb) {
            if constexpr (SCALED) sc[SCALED ? mb : 0] = g.rowscale[div_by8(min(rown + mb * 16, mlast), g.rows_per_scale, g.inv_rps)];
            if constexpr (HAS_X) {
                x1[HAS_X ? mb : 0] = *(const piece_t*)(X + (long)min(r1b + mb * 16, mlast) * ldx + cLc);
                x2[HAS_X ? mb : 0] = *(const piece_t*)(X + (long)min(r1b + mb * 16 + 1, mlast) * ldx + cLc);
            }
        };
        if constexpr (SCALED || HAS_X) {
#pragma unroll
            for (int mb = 0; mb < PF; ++mb) load_mb(mb);
            if constexpr (PF == MBN) G8_WAIT_VM0();
        }
        // pk = the own row's W values as bf16 pairs -> the two pieces this lane stores
        auto trade = [&](const unsigned (&pk)[W / 2], piece_t& d1, piece_t& d2) {
#pragma unroll
            for (int k = 0; k < PD; ++k) {
                const unsigned got = g8_swap1(isB ? pk[k] : pk[PD + k]);        // A gives its second half, B its first
                d1.d[k] = isB ? got : pk[k];
                d2.d[k] = isB ? pk[PD + k] : got;
            }
        };
#pragma unroll
        for (int mb = 0; mb < MBN; ++mb) {
            if constexpr (PF < MBN) {
                __builtin_amdgcn_sched_barrier(0);
                if (mb + PF < MBN) load_mb(mb + PF);
                __builtin_amdgcn_sched_barrier(0);
            }
            const int r1 = r1b + mb * 16;
            const float scm = SCALED ? sc[SCALED ? mb : 0] : 1.f;
            unsigned pk[W / 2], pd[W / 2];
            unsigned xo[HAS_X ? W / 2 : 1];             // the own row's residual / aux values: A loaded B's first half (x2), B loaded A's second (x1)
            if constexpr (HAS_X) {
#pragma unroll
                for (int k = 0; k < PD; ++k) {
                    const unsigned got = g8_swap1(isB ? x1[mb].d[k] : x2[mb].d[k]);
                    xo[k] = isB ? got : x1[mb].d[k];
                    xo[PD + k] = isB ? x2[mb].d[k] : got;
                }
            }
#pragma unroll
            for (int h = 0; h < 2; ++h) {               // the fused options on fp32, one rounding to bf16; half a strip at a time (registers)
#pragma unroll
                for (int k = 0; k < HW; k += 2) {
                    const int c = h * HW + k;            // column of the strip: acc[mb][c >> 2][c & 3]
                    float v0 = acc[mb][c >> 2][c & 3], v1 = acc[mb][(c + 1) >> 2][(c + 1) & 3];
                    if constexpr (MODE == G8_GELU_DUAL) {
                        // (the LDS interpolation table of common.h, which pays in the fused MLP kernels, does not here: 512 x 16 B entries read at
                        //  data-dependent addresses by all eight waves at once - epilogue 11.7k -> 23.5k cycles, measured)
                        f32x2 xx = {v0, v1}, yy, dd;
                        gelu_both2(xx, yy, dd);
                        pk[c / 2] = g8_pack2(yy[0], yy[1]); pd[c / 2] = g8_pack2(dd[0], dd[1]);
                    } else {
                        float xa = 0.f, xb = 0.f;
                        if constexpr (HAS_X) g8_unpack2(xo[c / 2], xa, xb);
                        if constexpr (MODE == G8_PLAIN) { if constexpr (SCALED) { v0 *= scm; v1 *= scm; } }
                        else if constexpr (MODE == G8_RESID) { v0 = SCALED ? fmaf(v0, scm, xa) : v0 + xa; v1 = SCALED ? fmaf(v1, scm, xb) : v1 + xb; }
                        else { v0 *= SCALED ? xa * scm : xa; v1 *= SCALED ? xb * scm : xb; }
                        pk[c / 2] = g8_pack2(v0, v1);
                    }
                }
            }
            piece_t d1, d2;
            trade(pk, d1, d2);
            const bool ok1 = colok && r1 < g.M, ok2 = colok && r1 + 1 < g.M;
            const long o1 = (long)min(r1, mlast) * g.ldc + cLc, o2 = (long)min(r1 + 1, mlast) * g.ldc + cLc;
            if (ok1) *(piece_t*)(g.C + o1) = d1;
            if (ok2) *(piece_t*)(g.C + o2) = d2;
            if constexpr (MODE == G8_GELU_DUAL) {
                trade(pd, d1, d2);
                // gelu'(u) is read again only in the backward pass, milliseconds later: written non-temporally it does not push the other half
                // of the pair - a = gelu(u), which fc2 reads NEXT - out of the 256 MB memory-side cache (fc2 forward inside the step
                // 102 -> 87 us, the step -0.08 ms; the store instruction itself is no faster)
                if (ok1) g8_store_nt<PD>(g.C2 + o1, d1);
                if (ok2) g8_store_nt<PD>(g.C2 + o2, d2);
            }
        }
    };

#define G8_LD_A(mq)                                                                                              \
    _Pragma("unroll") for (int mbl = 0; mbl < MBQ; ++mbl) {                                                      \
        fa[mbl][0] = *(const bf16x8*)(smem + ra0 + (mq) * OFF_A1 + mbl * 2048);                                  \
        fa[mbl][1] = *(const bf16x8*)(smem + ra1 + (mq) * OFF_A1 + mbl * 2048);                                  \
    }
#define G8_LD_B0()                                                                                               \
    _Pragma("unroll") for (int nbl = 0; nbl < NB0; ++nbl) {                                                      \
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
        _Pragma("unroll") for (int mbl = 0; mbl < MBQ; ++mbl)                                                    \
            _Pragma("unroll") for (int nbl = 0; nbl < ((nq) == 0 ? NB0 : NB1); ++nbl)                            \
                acc[(mq) * MBQ + mbl][(nq) * NB0 + nbl] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(               \
                    fb[nbl][kk], fa[mbl][kk], acc[(mq) * MBQ + mbl][(nq) * NB0 + nbl], 0, 0, 0);                 \
    __builtin_amdgcn_s_setprio(0);

    // ---- prologue: stream K-tile 0 complete, the first three half-tiles of K-tile 1 in flight ----
    int cp_i = 0, cp_kt = 0;
    int T = first;
    int m0c = (T / g.nx) * BM, n0c = (T - (T / g.nx) * g.nx) * BN;
    set_tile(first);
    dmaB(0); dmaA(0); dmaB(1); dmaA(1); advance();
    dmaB(0); dmaA(0); dmaB(1);
    init_acc(n0c);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VM_P4) : "memory");
    G8_BAR();
    if (wr == 1) G8_BAR();                       // the stagger: waves 4-7 run one barrier behind waves 0-3

    const int total_kt = my_n * g.nk;
    unsigned long long t_start = 0;
    if constexpr (DBG) { t_start = __builtin_amdgcn_s_memtime(); r_loop = __builtin_amdgcn_s_memrealtime(); }
    for (int s = 0; s < total_kt; ++s) {
        // phase 1: B-h0 + A-h0 fragments | A-h1 of the next K-tile | quadrant (m 0-63, n 0-31)
        G8_LD_B0();
        __builtin_amdgcn_sched_barrier(0);
        G8_LD_A(0);
        dmaA(1); advance();
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(2 * MBQ) : "memory");       // the B-h0 reads (issued first) have left LDS: B-h0 may be refilled next phase
        // B-h1 of this K-tile has landed (read next phase). Behind an epilogue its stores sit in the same in-order queue and are waited for
        // too: a count relaxed by the epilogue's stores would be wrong for a wave whose predicated stores were all skipped (ragged M / N edge),
        // and letting them stay outstanding measured no gain
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VM_P1) : "memory");
        G8_BAR();
        G8_MMA(0, 0, fb0);
        G8_BAR();
        // phase 2: B-h1 fragments | B-h0 of K-tile + 2 | quadrant (m 0-63, n 32-63)
        G8_LD_B1();
        dmaB(0);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VM_P2) : "memory");   // A-h1 of this K-tile has landed
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
        if constexpr (DBG) {
            if (cp_i == 1) {
                const unsigned long long now = __builtin_amdgcn_s_memtime();
                if (cp_kt == 0) kt0 = now; else if (cp_kt == 1) kt1 = now; else if (cp_kt == 2) kt2 = now; else if (cp_kt == 3) kt3 = now;
                else if (cp_kt == 4) kt4 = now; else if (cp_kt == 5) kt5 = now; else if (cp_kt == 6) kt6 = now; else if (cp_kt == 7) kt7 = now;
            }
        }
        if (++cp_kt == g.nk) {
            unsigned long long t_loop = 0;
            if constexpr (DBG) t_loop = __builtin_amdgcn_s_memtime();
            // The wave groups run one barrier apart and an epilogue has no barrier inside: left alone, group 1 sits at its last loop barrier for
            // the whole of group 0's epilogue and group 0 at its first barrier of the next tile for the whole of group 1's. One extra barrier
            // each - group 0 in front of its epilogue, group 1 behind its own - lets the two run side by side: their load latencies (residual /
            // aux / DropPath factor) and GELU arithmetic overlap (GELU-pair product 120.6 -> 112.0 us; nothing for the plain store-only epilogue,
            // which is paced by the vector-memory path the groups share).
            if (wr == 0) G8_BAR();
            epilogue(m0c, n0c);
            if (wr == 1) G8_BAR();
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
                m0c = mblk * BM; n0c = (T - mblk * g.nx) * BN;
                init_acc(n0c);
            }
        }
    }
    if (wr == 0) G8_BAR();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the cursor's surplus DMAs must not outlive the workgroup's LDS
    if constexpr (DBG) {
        if ((tid & 255) == 0) {       // [wg][group]: entry, bias copy done, loop start, exit (100 MHz); then the second tile's K-tile ends (cycles)
            unsigned long long* d = g.dbg + (long)256 * 2 * 16 * 4 + (long)(blockIdx.x * 2 + wr) * 4;
            d[0] = r_entry; d[1] = r_bias; d[2] = r_loop; d[3] = __builtin_amdgcn_s_memrealtime();
            unsigned long long* e = g.dbg + (long)256 * 2 * 16 * 4 + 256 * 2 * 4 + (long)(blockIdx.x * 2 + wr) * 8;
            e[0] = kt0; e[1] = kt1; e[2] = kt2; e[3] = kt3; e[4] = kt4; e[5] = kt5; e[6] = kt6; e[7] = kt7;
        }
    }
}

const char* g_gemm8_symbol = "";     // the instantiation the last launch ran, as rocprofv3 prints it (measurement aid)
char g_gemm8_symbuf[64];

template <int MODE, bool SCALED, int NB, int MBQ>
int launch8(const G8Args& a, int nwg, hipStream_t stream) {
    snprintf(g_gemm8_symbuf, sizeof g_gemm8_symbuf, "gemm8_kernel<%d, %s, %d, %d, false>", MODE, SCALED ? "true" : "false", NB, MBQ);
    g_gemm8_symbol = g_gemm8_symbuf;
    constexpr int lds = LDS_B;
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute((const void*)gemm8_kernel<MODE, SCALED, NB, MBQ, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); attr = true; }
    hipLaunchKernelGGL((gemm8_kernel<MODE, SCALED, NB, MBQ, false>), dim3((unsigned)nwg), dim3(512), lds, stream, a);
    PSELD_LAUNCH_CHECK("gemm8");
    return PSELD_OK;
}
template <int NB, int MBQ>
int launch8_mode(const Gemm8Desc& d, const G8Args& a, int nwg, hipStream_t stream) {
    const bool sc = d.rowscale != nullptr;
    if constexpr (MBQ == 4) {
        if (a.dbg) {          // diagnostic build of three epilogue kinds: stamps to [workgroup][wave group][tile < 16][4]
            auto go = [&](auto kern) {
                (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_B);
                hipLaunchKernelGGL(kern, dim3((unsigned)nwg), dim3(512), LDS_B, stream, a);
                return PSELD_OK;
            };
            if (d.gelu_dual) return go(gemm8_kernel<G8_GELU_DUAL, false, NB, 4, true>);
            if (d.resid && sc) return go(gemm8_kernel<G8_RESID, true, NB, 4, true>);
            if (!d.resid && !d.aux && !sc) return go(gemm8_kernel<G8_PLAIN, false, NB, 4, true>);
        }
    }
    if (d.gelu_dual) return launch8<G8_GELU_DUAL, false, NB, MBQ>(a, nwg, stream);
    if (d.resid) return sc ? launch8<G8_RESID, true, NB, MBQ>(a, nwg, stream) : launch8<G8_RESID, false, NB, MBQ>(a, nwg, stream);
    if (d.aux) return sc ? launch8<G8_MULAUX, true, NB, MBQ>(a, nwg, stream) : launch8<G8_MULAUX, false, NB, MBQ>(a, nwg, stream);
    return sc ? launch8<G8_PLAIN, true, NB, MBQ>(a, nwg, stream) : launch8<G8_PLAIN, false, NB, MBQ>(a, nwg, stream);
}

unsigned long long* g_gemm8_dbg = nullptr;
int g_gemm8_force_bm = 0, g_gemm8_force_bn = 0;
// cost of a 128-row tile per staged byte against the 256-row tile's. Isolated, on cold operands, the stage-2 N = 384 products (two rounds
// of 256 x 192 or three of 128 x 192) say 0.93 (74.0 against 72.5 us); inside the step that choice LOSES (three alternating same-box
// runs: 18.67 / 18.72 / 18.75 ms against 18.57 / 18.50 with 256 rows there), and the stage-3 products (one round against two: 60.5
// against 80.2 us) and the whole 32-chunk table (1.80 -> 1.53 ms) are indifferent between 0.93 and 1: so 1
constexpr double G8_ROW128_COST = 1.0;
}  // namespace

extern "C" void pseld_gemm8_set_debug_buffer(void* p) { g_gemm8_dbg = (unsigned long long*)p; }
// measurement aid (tools/gemm8_check.py, tests): force the tile shape of the following launches (0 = the launch's own choice)
extern "C" void pseld_gemm8_force_tile(int bm, int bn) { g_gemm8_force_bm = bm; g_gemm8_force_bn = bn; }
const char* pseld_gemm8_last_symbol() { return g_gemm8_symbol; }

int pseld_gemm8_supported(const Gemm8Desc& d) {
    // (any M: a layer must take the same kernel at every batch size - the bit-exact batch-independence and additivity tests; rows past M are
    //  clamped on load and never stored. The recurrent products with M <= 64 keep their skinny kernel: pseld_gemm asks it first)
    if (d.K % 64 != 0 || d.K < 128 || d.M < 1 || d.N < 128 || d.N % 8 != 0) return 0;
    if (d.lda % 8 != 0 || d.ldb % 8 != 0 || d.ldc % 8 != 0 || (d.resid && d.ldr % 8 != 0) || (d.aux && d.ldaux % 8 != 0)) return 0;
    if ((long)d.M * d.lda * 2 >= (1L << 32) || (long)d.N * d.ldb * 2 >= (1L << 32) || d.M >= (1 << 24)) return 0;
    if (pseld_cdiv(d.N, 192) * 192 > BIAS_FLOATS || pseld_cdiv(d.N, 256) * 256 > BIAS_FLOATS) return 0;
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
    // Tile shape: the cheapest of {256, 128} rows x {256, 192} columns by rounds x (bytes a workgroup stages per K-tile ~ rows + columns).
    // 192 columns need N % 192 == 0 (12-byte store pieces are whole only when the strips are); short K (<= 4 K-tiles) takes 256 columns
    // (the epilogue dominates and the 256 tile writes whole 128-byte lines per wave). A 128-row tile pays where the 256-row grid leaves
    // CUs idle in its last round: the 32-chunk step (stages 1-3: 1.80 -> 1.53 ms per step for forward + input gradients,
    // tools/gemm8_check.py CHUNKS=32; the step 6.66 -> 6.30 ms). All four shapes give the same bits, so the choice may depend on M (the batch size). PSELD_GEMM8_BN / PSELD_GEMM8_BM
    // (knobs, common.h) and pseld_gemm8_force_tile (tools, tests) force a shape.
    const int want_bn = g_gemm8_force_bn ? g_gemm8_force_bn : pseld_knob(KNOB_GEMM8_BN, 0);
    const int want_bm = g_gemm8_force_bm ? g_gemm8_force_bm : pseld_knob(KNOB_GEMM8_BM, 0);
    int bn = 0, bm = 0;
    double best = 0;
    for (int rows = 256; rows >= 128; rows -= 128)
        for (int w = 256; w >= 192; w -= 64) {
            if (w == 192 && d.N % 192 != 0) continue;
            if (want_bm == 128 || want_bm == 256) { if (rows != want_bm) continue; }
            if (want_bn == 192 || want_bn == 256) { if (w != want_bn && !(w == 256 && d.N % 192 != 0)) continue; }
            else if (w == 192 && a.nk <= 4) continue;
            const long tiles = (long)pseld_cdiv(d.N, w) * pseld_cdiv(d.M, rows);
            const double c = (double)((tiles + 255) / 256) * (rows + w) * (rows == 128 ? G8_ROW128_COST : 1.0);
            if (bn == 0 || c < best) { best = c; bn = w; bm = rows; }
        }
    a.nx = pseld_cdiv(d.N, bn);
    a.ntiles = a.nx * pseld_cdiv(d.M, bm);
    int nwg = (a.ntiles + 7) / 8 * 8;
    if (nwg > 256) nwg = 256;
    if (bm == 128) return bn == 192 ? launch8_mode<3, 2>(d, a, nwg, stream) : launch8_mode<4, 2>(d, a, nwg, stream);
    return bn == 192 ? launch8_mode<3, 4>(d, a, nwg, stream) : launch8_mode<4, 4>(d, a, nwg, stream);
}
