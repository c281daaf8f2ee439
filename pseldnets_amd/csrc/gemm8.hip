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
constexpr int LN_SCRATCH_B = 3 * 2 * 4 * 64 * 4;      // LayerNorm epilogues: three exchanges of [2 row groups][4 wave columns][64 rows] floats
constexpr int LDS_LN_B = LDS_B + LN_SCRATCH_B;

enum { G8_PLAIN = 0, G8_RESID = 1, G8_MULAUX = 2, G8_GELU_DUAL = 3,     // x SCALED (DropPath factor per token row)
       G8_RESID_LN = 4,      // (128 x 384 tile, N = 384) residual epilogue + LayerNorm of the row it has just formed: C = y, C2 = LN(y)
       G8_LNBWD = 5 };       // (128 x 384 tile, N = 384) input gradient whose epilogue is the backward of the LayerNorm in front of the layer

// s_waitcnt vmcnt(0) the compiler's own wait bookkeeping sees (gfx9 encoding: vmcnt 0, expcnt 7, lgkmcnt 15)
#define G8_WAIT_VM0() __builtin_amdgcn_s_waitcnt(0x0F70)
#define G8_FRESH_ROW(r) _Pragma("unroll") for (int k_ = 0; k_ < (int)(sizeof(r.d) / 4); ++k_) asm volatile("" : "+v"(r.d[k_]))   // (the unpacked floats of a packed row are not shared between passes)
#define G8_OPAQUE(v) asm volatile("" : "+v"(v))       // the value is the same, the compiler no longer knows: nothing derived from it is carried across

struct G8Args {
    const char* A; const char* B; bf16_t* C; bf16_t* C2;
    const float* bias; const bf16_t* resid; const bf16_t* aux; const float* rowscale;
    int M, N, K, lda, ldb, ldc, ldr, ldaux, rows_per_scale;
    float inv_rps;
    int nx, ntiles, nk;
    // LayerNorm epilogues (G8_RESID_LN, G8_LNBWD): gamma / beta of the norm (fp32 [N]); G8_LNBWD: ln_x = the norm's INPUT rows (bf16 [M, ldx]),
    // resid = the gradient that bypasses the norm (added to dx; may be null), ln_partial = per-tile partial sums [tile][2][N] of d(gamma), d(beta)
    const float* ln_gamma; const float* ln_beta; const bf16_t* ln_x; float* ln_partial;
    int ln_ldx; float ln_eps;
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
// PACK (built for NB = 3, MBQ = 2: the 128 x 192 tile): the four half-tile images are packed to their real sizes (8 + 8 + 16 + 8 KB per K-tile,
// 80 KB for the ring) and the bias comes from global memory instead of an LDS copy, so that TWO workgroups fit a CU (16 waves, <= 128
// registers each): twice the LDS-DMA bytes in flight per CU and a second workgroup's phases under every barrier of the first.
template <int MODE, bool SCALED, int NB, int MBQ = 4, bool DBG = false, bool PACK = false>
__global__ __launch_bounds__(512, PACK ? 4 : 2) void gemm8_kernel(const G8Args g) {
    constexpr int WN = NB * 16, BN = 4 * WN;
    // NB = 6 ("WIDE", MBQ = 2 only): the 128 x 384 tile - a tile spans the whole row of a C = 384 layer (stage 2 of HTS-AT), the A operand
    // is staged once per row block instead of once per column tile, and row-wise epilogues (LayerNorm) become possible. Column quadrants
    // of 3 + 3 blocks; images packed: A-h0 8 | A-h1 8 | B-h0 24 | B-h1 24 KB = the same 64 KB per K-tile.
    constexpr bool WIDE = NB == 6;
    static_assert(!WIDE || MBQ == 2, "the 384-column tile is built for 128 rows");
    constexpr int NB0 = WIDE ? 3 : 2;            // blocks in the first column quadrant = LDS-DMA instructions per wave of B-h0 (64 NB0 rows)
    constexpr int NB1 = NB - NB0;                // ... in the second = instructions per wave of B-h1
    constexpr int OFF_A1 = (PACK || WIDE) ? 32 * MBQ * 128 : HALF_B;        // A-h1 image (A-h0 at 0)
    constexpr int OFF_B0 = (PACK || WIDE) ? 2 * OFF_A1 : 2 * HALF_B;
    constexpr int OFF_B1 = (PACK || WIDE) ? OFF_B0 + NB0 * 8192 : 3 * HALF_B;
    constexpr int BUF = (PACK || WIDE) ? OFF_B1 + NB1 * 8192 : BUF_B;       // one K-tile
    static_assert(PACK || BUF == BUF_B, "the unpacked ring toggles its buffers by XOR");
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
    float* bias_s = (float*)(smem + RING_B);      // (not PACK)
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

    if constexpr (!PACK) {
        for (int i = tid; i < g.nx * BN; i += 512) bias_s[i] = (g.bias && i < g.N) ? g.bias[i] : 0.f;
        if constexpr (MODE == G8_RESID_LN || MODE == G8_LNBWD) {       // gamma | beta of the fused LayerNorm behind the bias (one column tile: N <= BN)
            for (int i = tid; i < BN; i += 512) {
                bias_s[BN + i] = i < g.N ? g.ln_gamma[i] : 0.f;
                bias_s[2 * BN + i] = (i < g.N && g.ln_beta) ? g.ln_beta[i] : 0.f;
            }
        }
        __syncthreads();
    }
    if constexpr (DBG) r_bias = __builtin_amdgcn_s_memrealtime();

    // ---- fragment read addresses (buffer 0): lane reads row l15 of a 16-row block, 16-byte chunk (4 kk + q) ^ ((row >> 1) & 7) ----
    // (LN epilogues: everything the main loop derives from the lane id is formed again behind the epilogue from `lane_m`, so that none of it
    //  is live across the epilogue's register peak - the compiler otherwise spills these loop invariants and reloads them INSIDE the K loop,
    //  each reload behind a vmcnt(0) that drains the LDS-DMA prefetch)
    constexpr bool LN_EPI = MODE == G8_RESID_LN || MODE == G8_LNBWD;
    int lane_m = lane;
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
        ld_buf = (unsigned)BUF - ld_buf;         // vmcnt distance stays constant in the tail
        if (++ld_kt == g.nk) {
            ld_kt = 0;
            if (ld_i + 1 < my_n) { ++ld_i; set_tile(first + ld_i * per); }
        }
    };

    bf16x8 fa[MBQ][2], fb0[NB0][2], fb1[NB1][2];
    f32x4 acc[MBN][NB];

    auto init_acc = [&](int n0) {
        f32x4 b[NB];
        if constexpr (PACK) {          // no LDS copy of the bias: this lane's 4 NB values straight from global memory (N % BN == 0 for the packed tile)
            const float* bp = g.bias + n0 + wc * WN + 4 * NB * q;
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) b[nb] = g.bias ? *(const f32x4*)(bp + 4 * nb) : f32x4{0.f, 0.f, 0.f, 0.f};
        } else {
            const float* bp = bias_s + n0 + wc * WN + 4 * NB * q;
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) b[nb] = *(const f32x4*)(bp + 4 * nb);
        }
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

    // ---- LayerNorm epilogues of the 128 x 384 tile (the tile spans the row: N <= 384 = one column tile) ----------------------------------
    // A lane holds W = 24 columns of its own rows (m0 + wr*64 + 16 mb + l15, mb < 4); the row's other columns sit in the lanes l15 + 16 q' of the
    // same wave (two __shfl_xor) and in the other three waves of the row group (one LDS exchange: scratch [exchange][wr][wc][64 rows], a
    // barrier - both wave groups run their epilogues side by side, so every exchange is one s_barrier all eight waves take part in).
    [[maybe_unused]] float* const ln_red = (float*)(smem + LDS_B);
    [[maybe_unused]] auto row_sum4 = [&](float (&v)[MBN], int xchg, int l15, int q) {      // v[mb] <- the sum over the whole row of the four rows this lane works on
#pragma unroll
        for (int mb = 0; mb < MBN; ++mb) {
            v[mb] += __shfl_xor(v[mb], 16);
            v[mb] += __shfl_xor(v[mb], 32);
        }
        float* red = ln_red + xchg * (2 * 4 * 64);
        if (q == 0) {
#pragma unroll
            for (int mb = 0; mb < MBN; ++mb) red[(wr * 4 + wc) * 64 + mb * 16 + l15] = v[mb];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        G8_BAR();
#pragma unroll
        for (int mb = 0; mb < MBN; ++mb) {
            const float* r = red + wr * 4 * 64 + mb * 16 + l15;
            v[mb] = (r[0] + r[64]) + (r[128] + r[192]);             // fixed order: every wave of the row group forms the same bits
        }
    };
    // G8_RESID_LN: y = resid + s (acc) [bias in acc], C = bf16(y), C2 = bf16(LayerNorm(bf16(y)) gamma + beta) - what the stand-alone LayerNorm
    // kernel would read and write (norm2 behind attn.proj, norm1 of the next block behind mlp.fc2: htsat.py:235,262 / model_utilities.py:166-170).
    // The rounded y replaces the accumulators in place (registers: 96 accumulators + 48 residual pieces is the peak)
    [[maybe_unused]] auto epilogue_ln = [&](int m0) {
        constexpr int W = 4 * NB, HW = W / 2, PD = HW / 2;
        typedef G8Piece<PD> piece_t;
        int le = tid & 63;
        asm volatile("" : "+v"(le));                                 // (the epilogue's lane-derived values are formed here, not hoisted over the K loop)
        const int l15 = le & 15, q = le >> 4;
        const bool isB = (l15 & 1) != 0;
        const int rown = m0 + wr * (32 * MBQ) + l15, r1b = rown - (isB ? 1 : 0);
        const int cown = wc * WN + W * q;                          // this lane's 24 own-row columns
        const int cL = cown + (isB ? HW : 0);
        const bool colok = cL < g.N, ownok = cown < g.N;           // (N % 24 == 0: strips and pieces are whole or absent)
        const int cLc = min(cL, g.N - HW), mlast = g.M - 1;
        float s1[MBN], s2[MBN];
        {
            piece_t x1[MBN], x2[MBN];
            float sc[MBN];
#pragma unroll
            for (int mb = 0; mb < MBN; ++mb) {
                sc[mb] = SCALED ? g.rowscale[div_by8(min(rown + mb * 16, mlast), g.rows_per_scale, g.inv_rps)] : 1.f;
                x1[mb] = *(const piece_t*)(g.resid + (long)min(r1b + mb * 16, mlast) * g.ldr + cLc);
                x2[mb] = *(const piece_t*)(g.resid + (long)min(r1b + mb * 16 + 1, mlast) * g.ldr + cLc);
            }
            G8_WAIT_VM0();
#pragma unroll
            for (int mb = 0; mb < MBN; ++mb) {
                __builtin_amdgcn_sched_barrier(0);
                float s = 0.f;
                unsigned pk[W / 2];
#pragma unroll
                for (int c = 0; c < W; c += 2) {
                    const int k = (c / 2) % PD;
                    // the own row's residual dword for columns (c, c + 1): its first half sits in piece 1 of the pair's A lane, its second in piece 2 of B
                    const unsigned got = g8_swap1(isB ? x1[mb].d[k] : x2[mb].d[k]);
                    const unsigned xo = c < HW ? (isB ? got : x1[mb].d[k]) : (isB ? x2[mb].d[k] : got);
                    float xa, xb, ya, yb;
                    g8_unpack2(xo, xa, xb);
                    const float v0 = fmaf(acc[mb][c >> 2][c & 3], sc[mb], xa), v1 = fmaf(acc[mb][(c + 1) >> 2][(c + 1) & 3], sc[mb], xb);
                    pk[c / 2] = g8_pack2(v0, v1);
                    g8_unpack2(pk[c / 2], ya, yb);                 // the statistics (and C2) are those of the STORED, rounded row
                    acc[mb][c >> 2][c & 3] = ya; acc[mb][(c + 1) >> 2][(c + 1) & 3] = yb;
                    s += ya + yb;
                }
                s1[mb] = ownok ? s : 0.f;
                // y leaves here (the packed row is dead behind its stores: kept for the last phase it costs 48 registers the compiler spills)
                piece_t y1, y2;
#pragma unroll
                for (int k = 0; k < PD; ++k) {
                    const unsigned gy = g8_swap1(isB ? pk[k] : pk[PD + k]);       // A gives its second half, B its first
                    y1.d[k] = isB ? gy : pk[k]; y2.d[k] = isB ? pk[PD + k] : gy;
                }
                const int r1 = r1b + mb * 16;
                if (colok && r1 < g.M) *(piece_t*)(g.C + (long)min(r1, mlast) * g.ldc + cLc) = y1;
                if (colok && r1 + 1 < g.M) *(piece_t*)(g.C + (long)min(r1 + 1, mlast) * g.ldc + cLc) = y2;
            }
        }
        row_sum4(s1, 0, l15, q);
        const float rN = 1.0f / (float)g.N;
#pragma unroll
        for (int mb = 0; mb < MBN; ++mb) {
            const float mean = s1[mb] * rN;
            float qv = 0.f;
#pragma unroll
            for (int c = 0; c < W; ++c) { const float dv = acc[mb][c >> 2][c & 3] - mean; qv = fmaf(dv, dv, qv); }
            s2[mb] = ownok ? qv : 0.f;
        }
        row_sum4(s2, 1, l15, q);
        const float* gam = bias_s + BN + cown;
        const float* bet = bias_s + 2 * BN + cown;
#pragma unroll
        for (int mb = 0; mb < MBN; ++mb) {
            __builtin_amdgcn_sched_barrier(0);
            float mean = s1[mb] * rN, rstd = rsqrtf(s2[mb] * rN + g.ln_eps);
            G8_OPAQUE(mean);                                         // (y - mean of the variance pass is not kept for this one: 96 registers)
            const int r1 = r1b + mb * 16;
            const bool ok1 = colok && r1 < g.M, ok2 = colok && r1 + 1 < g.M;
            const long o1 = (long)min(r1, mlast) * g.ldc + cLc, o2 = (long)min(r1 + 1, mlast) * g.ldc + cLc;
            piece_t z1, z2;
#pragma unroll
            for (int k = 0; k < PD; ++k) {                          // dword k of the first half / of the second half of the own row
                const float a0 = acc[mb][(2 * k) >> 2][(2 * k) & 3], a1 = acc[mb][(2 * k + 1) >> 2][(2 * k + 1) & 3];
                const float b0 = acc[mb][(HW + 2 * k) >> 2][(HW + 2 * k) & 3], b1 = acc[mb][(HW + 2 * k + 1) >> 2][(HW + 2 * k + 1) & 3];
                const unsigned zlo = g8_pack2(fmaf((a0 - mean) * rstd, gam[2 * k], bet[2 * k]), fmaf((a1 - mean) * rstd, gam[2 * k + 1], bet[2 * k + 1]));
                const unsigned zhi = g8_pack2(fmaf((b0 - mean) * rstd, gam[HW + 2 * k], bet[HW + 2 * k]), fmaf((b1 - mean) * rstd, gam[HW + 2 * k + 1], bet[HW + 2 * k + 1]));
                const unsigned gz = g8_swap1(isB ? zlo : zhi);     // A gives its second half, B its first
                z1.d[k] = isB ? gz : zlo; z2.d[k] = isB ? zhi : gz;
            }
            if (ok1) *(piece_t*)(g.C2 + o1) = z1;
            if (ok2) *(piece_t*)(g.C2 + o2) = z2;
        }
    };
    // G8_LNBWD: acc = d(LN output) of the rows (the input gradient of the Linear behind the norm). With x = the norm's input rows:
    // xh = (x - mean) rstd, gy = acc gamma, dx = rstd (gy - mean(gy) - xh mean(gy xh)) + dres  (reference: autograd of nn.LayerNorm,
    // htsat.py:235,262); d(gamma) += acc xh, d(beta) += acc per column -> per-tile partial sums [tile][2][N] (summed by the callers' reduction).
    // acc is rounded to bf16 first - exactly what the two-launch path stores between the GEMM and the LayerNorm backward.
    [[maybe_unused]] auto epilogue_lnbwd = [&](int m0, int tile) {
        constexpr int W = 4 * NB, HW = W / 2, PD = HW / 2;
        typedef G8Piece<PD> piece_t;
        typedef G8Piece<W / 2> row_t;                              // a lane's 24 own-row columns: 48 bytes
        int le = tid & 63;
        asm volatile("" : "+v"(le));
        const int l15 = le & 15, q = le >> 4;
        const bool isB = (l15 & 1) != 0;
        const int rown = m0 + wr * (32 * MBQ) + l15, r1b = rown - (isB ? 1 : 0);
        const int cown = wc * WN + W * q;
        const int cL = cown + (isB ? HW : 0);
        const bool colok = cL < g.N, ownok = cown < g.N;
        const int cLc = min(cL, g.N - HW), mlast = g.M - 1, cownc = min(cown, g.N - W);
        row_t xr[MBN];
#pragma unroll
        for (int mb = 0; mb < MBN; ++mb) xr[mb] = *(const row_t*)(g.ln_x + (long)min(rown + mb * 16, mlast) * g.ln_ldx + cownc);
        G8_WAIT_VM0();
        const float rN = 1.0f / (float)g.N;
        float s1[MBN], s2[MBN];
#pragma unroll
        for (int mb = 0; mb < MBN; ++mb) {
            float s = 0.f;
#pragma unroll
            for (int c = 0; c < W; c += 2) { float a, b; g8_unpack2(xr[mb].d[c / 2], a, b); s += a + b; }
            s1[mb] = ownok ? s : 0.f;
        }
        row_sum4(s1, 0, l15, q);
#pragma unroll
        for (int mb = 0; mb < MBN; ++mb) {
            const float mean = s1[mb] * rN;
            float qv = 0.f;
            G8_FRESH_ROW(xr[mb]);
#pragma unroll
            for (int c = 0; c < W; c += 2) { float a, b; g8_unpack2(xr[mb].d[c / 2], a, b); a -= mean; b -= mean; qv = fmaf(a, a, fmaf(b, b, qv)); }
            s2[mb] = ownok ? qv : 0.f;
        }
        row_sum4(s2, 1, l15, q);
        const float* gam = bias_s + BN + cown;
        float c1[MBN], c2[MBN];
#pragma unroll
        for (int mb = 0; mb < MBN; ++mb) {
            __builtin_amdgcn_sched_barrier(0);
            float mean = s1[mb] * rN, rstd = rsqrtf(s2[mb] * rN + g.ln_eps);
            G8_OPAQUE(mean);                                         // (x - mean of the variance pass is not kept: formed again from the packed row)
            const bool live = ownok && rown + mb * 16 < g.M;
            float a1 = 0.f, a2 = 0.f;
            G8_FRESH_ROW(xr[mb]);
#pragma unroll
            for (int c = 0; c < W; c += 2) {
                float xa, xb, da, db;
                g8_unpack2(xr[mb].d[c / 2], xa, xb);
                g8_unpack2(g8_pack2(acc[mb][c >> 2][c & 3], acc[mb][(c + 1) >> 2][(c + 1) & 3]), da, db);
                if (!live) { da = 0.f; db = 0.f; }
                acc[mb][c >> 2][c & 3] = da; acc[mb][(c + 1) >> 2][(c + 1) & 3] = db;       // the rounded d(LN output) stays in the accumulators
                const float ga = da * gam[c], gb = db * gam[c + 1];
                a1 += ga + gb;
                a2 = fmaf(ga, (xa - mean) * rstd, fmaf(gb, (xb - mean) * rstd, a2));
            }
            c1[mb] = a1; c2[mb] = a2;
            s1[mb] = mean; s2[mb] = rstd;
        }
        row_sum4(c1, 2, l15, q);
        row_sum4(c2, 0, l15, q);                                           // (the first exchange's scratch is free again: two barriers lie between)
        // dx, row block by row block; the bypass gradient's pieces (traded layout) are fetched one row block ahead
        piece_t rA, rB, nA, nB;
        auto fetch_res = [&](int mb, piece_t& pa, piece_t& pb) {
            if (g.resid) {
                pa = *(const piece_t*)(g.resid + (long)min(r1b + mb * 16, mlast) * g.ldr + cLc);
                pb = *(const piece_t*)(g.resid + (long)min(r1b + mb * 16 + 1, mlast) * g.ldr + cLc);
            } else {
#pragma unroll
                for (int k = 0; k < PD; ++k) { pa.d[k] = 0u; pb.d[k] = 0u; }
            }
        };
        fetch_res(0, rA, rB);
#pragma unroll
        for (int mb = 0; mb < MBN; ++mb) {
            __builtin_amdgcn_sched_barrier(0);
            if (mb + 1 < MBN) fetch_res(mb + 1, nA, nB);
            const float m1 = c1[mb] * rN, m2 = c2[mb] * rN;
            float mean = s1[mb], rstd = s2[mb];
            G8_OPAQUE(mean); G8_OPAQUE(rstd);                       // ((x - mean) rstd of the pass before is not kept)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) G8_OPAQUE(acc[mb][nb]);   // (nor its products with gamma)
            piece_t d1, d2;
            G8_FRESH_ROW(xr[mb]);
#pragma unroll
            for (int k = 0; k < PD; ++k) {
                float xa, xb, ra, rb;
                g8_unpack2(xr[mb].d[k], xa, xb);
                const unsigned lo = g8_pack2(rstd * (acc[mb][(2 * k) >> 2][(2 * k) & 3] * gam[2 * k] - m1 - (xa - mean) * rstd * m2),
                                             rstd * (acc[mb][(2 * k + 1) >> 2][(2 * k + 1) & 3] * gam[2 * k + 1] - m1 - (xb - mean) * rstd * m2));
                g8_unpack2(xr[mb].d[PD + k], xa, xb);
                const unsigned hi = g8_pack2(rstd * (acc[mb][(HW + 2 * k) >> 2][(HW + 2 * k) & 3] * gam[HW + 2 * k] - m1 - (xa - mean) * rstd * m2),
                                             rstd * (acc[mb][(HW + 2 * k + 1) >> 2][(HW + 2 * k + 1) & 3] * gam[HW + 2 * k + 1] - m1 - (xb - mean) * rstd * m2));
                const unsigned got = g8_swap1(isB ? lo : hi);
                unsigned e1 = isB ? got : lo, e2 = isB ? hi : got;
                // + the bypass gradient, in the traded layout (the two-launch path adds it to the UNROUNDED dx: here dx is rounded once more first -
                // one extra bf16 rounding of the branch gradient only, inside the tolerance of every gradient gate)
                float a, b;
                g8_unpack2(e1, a, b); g8_unpack2(rA.d[k], ra, rb); d1.d[k] = g.resid ? g8_pack2(a + ra, b + rb) : e1;
                g8_unpack2(e2, a, b); g8_unpack2(rB.d[k], ra, rb); d2.d[k] = g.resid ? g8_pack2(a + ra, b + rb) : e2;
            }
            const int r1 = r1b + mb * 16;
            if (colok && r1 < g.M) *(piece_t*)(g.C + (long)min(r1, mlast) * g.ldc + cLc) = d1;
            if (colok && r1 + 1 < g.M) *(piece_t*)(g.C + (long)min(r1 + 1, mlast) * g.ldc + cLc) = d2;
            rA = nA; rB = nB;
        }
        G8_WAIT_VM0();
        // d(gamma), d(beta): the lane's 4 rows, then the 16 lanes of a DPP row (same q = same columns, l15 = rows), then the two row groups
        // through the exchange scratch ([2 wr][4 wc][4 q][2][24] floats = its 6 KB, free behind a barrier)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        G8_BAR();                                                     // every wave is past its reads of the exchange scratch
        float* pg = ln_red + ((wr * 4 + wc) * 4 + q) * (2 * W);
        auto dpp_add = [](float v, auto ctrl) {
            return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), decltype(ctrl)::value, 0xF, 0xF, true));
        };
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {                             // half the lane's columns at a time: 24 running sums beside acc and x, not 48
            __builtin_amdgcn_sched_barrier(0);
            float dgam[HW], dbet[HW];
#pragma unroll
            for (int c = 0; c < HW; ++c) { dgam[c] = 0.f; dbet[c] = 0.f; }
#pragma unroll
            for (int mb = 0; mb < MBN; ++mb) {
                float mean = s1[mb], rstd = s2[mb];
                G8_OPAQUE(mean); G8_OPAQUE(rstd);
                G8_FRESH_ROW(xr[mb]);
#pragma unroll
                for (int c = 0; c < HW; c += 2) {
                    const int cc = hf * HW + c;
                    float xa, xb;
                    g8_unpack2(xr[mb].d[cc / 2], xa, xb);
                    const float da = acc[mb][cc >> 2][cc & 3], db = acc[mb][(cc + 1) >> 2][(cc + 1) & 3];
                    dgam[c] = fmaf(da, (xa - mean) * rstd, dgam[c]); dgam[c + 1] = fmaf(db, (xb - mean) * rstd, dgam[c + 1]);
                    dbet[c] += da; dbet[c + 1] += db;
                }
            }
#pragma unroll
            for (int c = 0; c < HW; ++c) {                          // inclusive scan over the 16 lanes of the DPP row (row_shr 1, 2, 4, 8): lane 15 holds the sum
                float a = dgam[c], b = dbet[c];
                a = dpp_add(a, std::integral_constant<int, 0x111>{}); b = dpp_add(b, std::integral_constant<int, 0x111>{});
                a = dpp_add(a, std::integral_constant<int, 0x112>{}); b = dpp_add(b, std::integral_constant<int, 0x112>{});
                a = dpp_add(a, std::integral_constant<int, 0x114>{}); b = dpp_add(b, std::integral_constant<int, 0x114>{});
                a = dpp_add(a, std::integral_constant<int, 0x118>{}); b = dpp_add(b, std::integral_constant<int, 0x118>{});
                if (l15 == 15) { pg[hf * HW + c] = a; pg[W + hf * HW + c] = b; }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        G8_BAR();
        if (wr == 0 && l15 < 12 && ownok) {                          // 12 lanes x 4 floats = the 24 + 24 sums of this (wave column, q): lane -> (which, 4 columns)
            const int which = l15 / 6, c4 = (l15 % 6) * 4;
            const float* p0 = ln_red + ((0 * 4 + wc) * 4 + q) * (2 * W) + which * W + c4;
            const float* p1 = ln_red + ((1 * 4 + wc) * 4 + q) * (2 * W) + which * W + c4;
            const f32x4 v = *(const f32x4*)p0 + *(const f32x4*)p1;
            *(f32x4*)(g.ln_partial + ((long)tile * 2 + which) * g.N + cown + c4) = v;
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
    unsigned rdelta = (unsigned)BUF;             // (PACK) the read addresses alternate between the two K-tile buffers by +- BUF
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
        if constexpr (PACK) { ra0 += rdelta; ra1 += rdelta; rb0 += rdelta; rb1 += rdelta; rc0 += rdelta; rc1 += rdelta; rdelta = 0u - rdelta; }
        else { ra0 ^= BUF_B; ra1 ^= BUF_B; rb0 ^= BUF_B; rb1 ^= BUF_B; rc0 ^= BUF_B; rc1 ^= BUF_B; }
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
            if constexpr (MODE == G8_RESID_LN) epilogue_ln(m0c);
            else if constexpr (MODE == G8_LNBWD) epilogue_lnbwd(m0c, T);
            else epilogue(m0c, n0c);
            if (wr == 1) G8_BAR();
            if constexpr (LN_EPI) {
                lane_m = tid & 63;
                asm volatile("" : "+v"(lane_m));
                set_frag_addrs(((s + 1) & 1) ? (unsigned)BUF_B : 0u);
                set_tile(first + ld_i * per);
            }
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

constexpr int LDS_PACK_B = 2 * (2 * 32 * 2 * 128 + HALF_B + 8192);      // NB = 3, MBQ = 2 packed: 2 x 40 KB = half a CU's LDS
template <int MODE, bool SCALED, int NB, int MBQ, bool PACK = false>
int launch8(const G8Args& a, int nwg, hipStream_t stream) {
    snprintf(g_gemm8_symbuf, sizeof g_gemm8_symbuf, "gemm8_kernel<%d, %s, %d, %d, false%s>", MODE, SCALED ? "true" : "false", NB, MBQ, PACK ? ", true" : "");
    g_gemm8_symbol = g_gemm8_symbuf;
    constexpr int lds = PACK ? LDS_PACK_B : ((MODE == G8_RESID_LN || MODE == G8_LNBWD) ? LDS_LN_B : LDS_B);
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute((const void*)gemm8_kernel<MODE, SCALED, NB, MBQ, false, PACK>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); attr = true; }
    hipLaunchKernelGGL((gemm8_kernel<MODE, SCALED, NB, MBQ, false, PACK>), dim3((unsigned)nwg), dim3(512), lds, stream, a);
    PSELD_LAUNCH_CHECK("gemm8");
    return PSELD_OK;
}
template <int NB, int MBQ, bool PACK = false>
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
    if constexpr (NB != 6) {
        if (d.gelu_dual) return launch8<G8_GELU_DUAL, false, NB, MBQ, PACK>(a, nwg, stream);
    }
    if (d.resid) return sc ? launch8<G8_RESID, true, NB, MBQ, PACK>(a, nwg, stream) : launch8<G8_RESID, false, NB, MBQ, PACK>(a, nwg, stream);
    if (d.aux) return sc ? launch8<G8_MULAUX, true, NB, MBQ, PACK>(a, nwg, stream) : launch8<G8_MULAUX, false, NB, MBQ, PACK>(a, nwg, stream);
    return sc ? launch8<G8_PLAIN, true, NB, MBQ, PACK>(a, nwg, stream) : launch8<G8_PLAIN, false, NB, MBQ, PACK>(a, nwg, stream);
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
    if (pseld_cdiv(d.N, 192) * 192 > BIAS_FLOATS || pseld_cdiv(d.N, 256) * 256 > BIAS_FLOATS || pseld_cdiv(d.N, 384) * 384 > BIAS_FLOATS) return 0;
    if ((((unsigned long)d.A | (unsigned long)d.B | (unsigned long)d.C | (unsigned long)d.C2 | (unsigned long)d.resid | (unsigned long)d.aux) & 15) != 0) return 0;
    if (d.resid && d.aux) return 0;
    if (d.gelu_dual && (d.resid || d.aux || d.rowscale || !d.C2)) return 0;
    if (d.ln_mode) {
        if (d.N > 384 || d.N % 96 != 0 || d.gelu_dual || d.aux || !d.ln_gamma) return 0;
        if (d.ln_mode == 1 && (!d.resid || !d.C2 || (((unsigned long)d.C2) & 15))) return 0;
        if (d.ln_mode == 2 && (!d.ln_x || !d.ln_partial || d.rowscale || d.bias || d.ln_ldx % 8 != 0 || (((unsigned long)d.ln_x | (unsigned long)d.ln_partial) & 15))) return 0;
    }
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
    a.ln_gamma = d.ln_gamma; a.ln_beta = d.ln_beta; a.ln_x = (const bf16_t*)d.ln_x; a.ln_partial = d.ln_partial; a.ln_ldx = d.ln_ldx; a.ln_eps = d.ln_eps;
    if (d.ln_mode) {      // the row-spanning tile with a LayerNorm epilogue: one column tile, 128-row blocks
        a.nx = 1;
        a.ntiles = pseld_cdiv(d.M, 128);
        int nwgl = (a.ntiles + 7) / 8 * 8;
        if (nwgl > 256) nwgl = 256;
        if (d.ln_mode == 1) return d.rowscale ? launch8<G8_RESID_LN, true, 6, 2>(a, nwgl, stream) : launch8<G8_RESID_LN, false, 6, 2>(a, nwgl, stream);
        return launch8<G8_LNBWD, false, 6, 2>(a, nwgl, stream);
    }
    // Tile shape: the cheapest of {256, 128} rows x {256, 192} columns by rounds x (bytes a workgroup stages per K-tile ~ rows + columns).
    // 192 columns need N % 192 == 0 (12-byte store pieces are whole only when the strips are); short K (<= 4 K-tiles) takes 256 columns
    // (the epilogue dominates and the 256 tile writes whole 128-byte lines per wave). A 128-row tile pays where the 256-row grid leaves
    // CUs idle in its last round: the 32-chunk step (stages 1-3: 1.80 -> 1.53 ms per step for forward + input gradients,
    // tools/gemm8_check.py CHUNKS=32; the step 6.66 -> 6.30 ms). All four shapes give the same bits, so the choice may depend on M (the batch size). PSELD_GEMM8_BN / PSELD_GEMM8_BM
    // (knobs, common.h) and pseld_gemm8_force_tile (tools, tests) force a shape.
    const int want_bn = g_gemm8_force_bn ? g_gemm8_force_bn : pseld_knob(KNOB_GEMM8_BN, 0);
    const int want_bm = g_gemm8_force_bm ? g_gemm8_force_bm : pseld_knob(KNOB_GEMM8_BM, 0);
    // rows = 64: the 128 x 192 tile packed for two workgroups per CU. Forced (force_tile / knob GEMM8_BM = 64), or chosen (knob GEMM8_PACK,
    // A/B) for narrow outputs (N <= 384: one or two column tiles, the A operand dominates the traffic) with at least one full round of 512
    bool pack = want_bm == 64;
    if (!pack && want_bm == 0 && want_bn == 0 && pseld_knob(KNOB_GEMM8_PACK, 0) != 0 && d.N <= 384 && a.nk > 4 &&
        (long)(d.N / 192) * pseld_cdiv(d.M, 128) >= 512) pack = true;
    if (pack && d.N % 192 == 0 && (((unsigned long)d.bias) & 15) == 0) {
        a.nx = d.N / 192;
        a.ntiles = a.nx * pseld_cdiv(d.M, 128);
        int nwg2 = (a.ntiles + 7) / 8 * 8;
        if (nwg2 > 512) nwg2 = 512;
        return launch8_mode<3, 2, true>(d, a, nwg2, stream);
    }
    // columns = 384 (force / knob): the 128 x 384 row-spanning tile (no GELU-pair epilogue; N a multiple of 96: whole 24-byte store pieces)
    if (want_bn == 384 && d.N % 96 == 0 && !d.gelu_dual) {
        a.nx = pseld_cdiv(d.N, 384);
        a.ntiles = a.nx * pseld_cdiv(d.M, 128);
        int nwg6 = (a.ntiles + 7) / 8 * 8;
        if (nwg6 > 256) nwg6 = 256;
        return launch8_mode<6, 2>(d, a, nwg6, stream);
    }
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
