// Weight gradient on the eight-phase loop of gemm8.hip (bf16, gfx950): one fp32 slab tile per workgroup.
//
// dW[N_out, K_in] (one fp32 slab per token split) = dY[Mtok, N_out]^T X[Mtok, K_in] - the weight gradients of htsat.py:118,140 (qkv,
// proj), model_utilities.py:166-170 (fc1, fc2), htsat.py:309 (PatchMerging.reduction) and accdoa.py:230 (head) for the MFMA-bound
// stages. Both operands are token-major, i.e. the contraction index is the ROW of both: a K-tile is 64 token rows of dY (256 columns)
// and of X (256 or 192 columns), staged as k-major half-tile images [64 tokens][128 columns] by global_load_lds_dwordx4 (4 token rows
// of 256 B per wave-instruction) and read back as MFMA operands with ds_read_b64_tr_b16 (two per 8-token fragment). The 32-byte
// piece a lane fetches is XOR-ed with (token & 3) | ((token >> 3) & 1) << 2 on the SOURCE address, so the 8 token rows a 32-lane
// half reads per transposed read fall in 8 different 32-byte bank groups.
// Everything else is gemm8.hip's schedule: one workgroup of 8 waves (2 x 4) per CU, 128 x 64 (x 48) per wave in 16 x 16 x 32 MFMAs,
// two K-tile buffers of four half-tile images, four phases per K-tile {fragment reads + one half-tile of LDS-DMA | barrier | 16 MFMAs
// | barrier}, the two wave groups staggered by one barrier, ONE counted vmcnt per K-tile with three half-tiles left in flight.
// The product is computed as C^T[k_in][n_out] blocks (X fragments as the A operand), so a lane holds 4 consecutive k_in of one n_out
// row: the fp32 slab leaves as 16-byte pieces straight from the accumulators.
// Launch = output tiles x token splits = one resident round (<= 256 workgroups); all tiles of a split sit on one XCD (they read the
// same token rows). DropPath (rowscale: one factor per sample, a K-tile never straddles samples): K-tiles of dropped samples are
// skipped, the common factor 1 / keep_prob of the kept ones is applied once to the accumulators (any other factor scales the dY
// fragments of its K-tile). The bias gradient (column sums of dY) rides on the matrix pipe: 4 extra MFMAs per K-tile and wave
// against a fragment of ones.
#include "gemm8.h"
#include <stdlib.h>

namespace {

typedef __attribute__((address_space(3))) void* lds_vptr8w;
typedef __attribute__((address_space(3))) short4v* lds_s4p8w;
typedef __attribute__((ext_vector_type(8))) short short8w;

constexpr int HALF_B = 16384;             // one half-tile image: 64 token rows x 256 B (128 columns)
constexpr int BUF_B = 4 * HALF_B;         // one K-tile: dY-h0 | dY-h1 | X-h0 | X-h1
constexpr int LDS_B = 2 * BUF_B;

struct G8WArgs {          // one weight matrix of the launch
    const char* A;          // dY [Mtok, lda]
    const char* B;          // X  [Mtok, ldb]
    float* C;               // slabs: split z at C + z * slab_stride, [N_out, K_in] dense (splits == 1: the gradient itself)
    float* colsum;          // per-split column sums of dY (bias gradient) or null; split z at colsum + z * colsum_stride
    const float* rowscale;  // DropPath factor per sample or null
    long slab_stride, colsum_stride;
    int M, N, Mtok;         // N_out, K_in, tokens
    int lda, ldb;
    int kchunk;             // tokens per split (multiple of 64)
    int nx, ntile, splits;
    int rows_per_scale, nscale;
    int pad_;
};
constexpr int G8W_MAXP = 32;
struct G8WOne { G8WArgs p; };      // a single matrix: 104 bytes of kernel arguments instead of the group's 3.5 KB
struct G8WGroup {         // the weight matrices of one launch (a stage's layers): workgroup L of the launch belongs to the matrix p with
    int count;            // first[p] <= L < first[p + 1] and is its (split, tile) pair L - first[p] in split-major order
    int first[G8W_MAXP + 1];
    G8WArgs p[G8W_MAXP];
};

__device__ __forceinline__ void g8w_dma(unsigned lds_dst, const void* sbase, unsigned voff) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(lds_dst), "s"(sbase) : "memory");
}

#define G8W_BAR()                                 \
    do {                                          \
        __builtin_amdgcn_sched_barrier(0);        \
        __builtin_amdgcn_s_barrier();             \
        __builtin_amdgcn_sched_barrier(0);        \
    } while (0)

// NB = 16-column accumulator blocks per wave on the X side: 4 -> 256 x 256 tile, 3 -> 256 x 192 (X-h1 is then a [64][64] image)
template <int NB, typename P = G8WOne>
__global__ __launch_bounds__(512, 2) void gemm8w_kernel(const P grp) {
    constexpr bool ONE = __is_same(P, G8WOne);
    constexpr int WN = NB * 16, BN = 4 * WN;
    constexpr int NB1 = NB - 2;
    // Just-in-time waits: a half-tile is waited for in the phase BEFORE the one that reads it, so the five youngest half-tiles stay in
    // flight at every wait (B-h1 is NB1 instructions per wave, the others two): every load has five phases to land (three with the
    // single wait per K-tile of the guide's template - too few for operands that come from HBM rather than L2)
    constexpr int VM_P4 = 6 + 2 * NB1, VM_P1 = 8 + NB1, VM_P2 = 8 + NB1;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int l15 = lane & 15, kg = lane >> 4, q4 = l15 >> 2, p4 = l15 & 3;

    // workgroup -> (matrix, split, tile): the pairs, matrix by matrix in split-major order, are dealt to the XCDs in runs of 32 (one per
    // CU), so the tiles of a split (which read the same token rows) share an L2 and no XCD gets more workgroups than it has CUs
    const int id = blockIdx.x, xcd = id & 7, slot = id >> 3;
    const int L = xcd * 32 + slot;
    int pi = 0, Lp = L;
    if constexpr (ONE) {
        if (slot >= 32 || L >= grp.p.ntile * grp.p.splits) return;
    } else {
        if (slot >= 32 || L >= grp.first[grp.count]) return;
        while (pi + 1 < grp.count && L >= grp.first[pi + 1]) ++pi;
        Lp = L - grp.first[pi];
    }
    const G8WArgs& g = [&]() -> const G8WArgs& { if constexpr (ONE) return grp.p; else return grp.p[pi]; }();
    const int z = Lp / g.ntile, t = Lp - z * g.ntile;
    const int mblk = t / g.nx, nblk = t - mblk * g.nx;
    const int m0 = mblk * 256, n0 = nblk * BN;
    const int tok0 = z * g.kchunk;
    const int nk = (min(g.Mtok, tok0 + g.kchunk) - tok0) >> 6;
    if (nk <= 0) return;

    // ---- transposed-read addresses (buffer 0). Lane 4q + p of a 16-lane group supplies token row 8 kg + q (+ 4 for the second read,
    // + 32 kk), 8 bytes at column 4 p of the block; the block (16 columns = one 32-byte piece) sits at piece index b ^ f ----
    const int f8 = q4 | ((kg & 1) << 2);                        // 256-byte rows: 8 pieces
    const unsigned rowpart = (unsigned)((8 * kg + q4) * 256 + p4 * 8);
    unsigned la[4], lb[2], lc[NB1];
#pragma unroll
    for (int i = 0; i < 4; ++i) la[i] = rowpart + (unsigned)((((wr * 4 + i) ^ f8) & 7) << 5);                         // dY half images
#pragma unroll
    for (int i = 0; i < 2; ++i) lb[i] = (unsigned)(2 * HALF_B) + rowpart + (unsigned)((((wc * 2 + i) ^ f8) & 7) << 5);  // X-h0
    if constexpr (NB == 4) {
#pragma unroll
        for (int i = 0; i < 2; ++i) lc[i] = (unsigned)(3 * HALF_B) + rowpart + (unsigned)((((wc * 2 + i) ^ f8) & 7) << 5);
    } else {
        // X-h1 of the 192-wide tile: 128-byte rows (4 pieces), piece wc ^ g, g = (q >> 1) | (kg & 1) << 1
        const int g4 = (q4 >> 1) | ((kg & 1) << 1);
        lc[0] = (unsigned)(3 * HALF_B) + (unsigned)((8 * kg + q4) * 128 + p4 * 8) + (unsigned)(((wc ^ g4) & 3) << 5);
    }

    // ---- LDS-DMA source offsets: [half][instruction]; image byte (2 wave + j) * 1024 + lane * 16 ----
    unsigned offA[2][2], offB[2][2];
    {
        const int fr = ((lane >> 4) & 3) | ((wave & 1) << 2);            // f(token row) of the row this lane fills
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int r = 8 * wave + 4 * j + (lane >> 4);                  // token row inside the K-tile
            const int c16 = lane & 15;
            const int ci = ((((c16 >> 1) ^ fr) & 7) << 4) + (c16 & 1) * 8; // image column of the 16-byte chunk this lane fetches
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int ca = min(m0 + (ci >> 6) * 128 + h * 64 + (ci & 63), g.M - 8);
                offA[h][j] = (unsigned)r * (unsigned)(g.lda * 2) + (unsigned)(ca * 2);
            }
            const int cb0 = n0 + (ci >> 5) * WN + (ci & 31);
            offB[0][j] = (unsigned)r * (unsigned)(g.ldb * 2) + (unsigned)(min(cb0, g.N - 8) * 2);
            if constexpr (NB == 4) offB[1][j] = (unsigned)r * (unsigned)(g.ldb * 2) + (unsigned)(min(cb0 + 32, g.N - 8) * 2);
        }
        if constexpr (NB == 3) {       // X-h1: [64 rows][128 B], 8 rows per instruction, one instruction per wave
            const int r = 8 * wave + (lane >> 3);
            const int c16 = lane & 7;
            const int gr = ((r >> 1) & 1) | (((r >> 3) & 1) << 1);
            const int ci = ((((c16 >> 1) ^ gr) & 3) << 4) + (c16 & 1) * 8;         // 0..63: wave column block ci >> 4
            offB[1][0] = (unsigned)r * (unsigned)(g.ldb * 2) + (unsigned)(min(n0 + (ci >> 4) * WN + 32 + (ci & 15), g.N - 8) * 2);
            offB[1][1] = 0;
        }
    }
    const unsigned lds_base = (unsigned)(unsigned long)(lds_vptr8w)smem;
    const unsigned dst_w = lds_base + (unsigned)wave * 2048u;
    const char* baseA = g.A + (long)tok0 * g.lda * 2;
    const char* baseB = g.B + (long)tok0 * g.ldb * 2;
    const long stepA = (long)64 * g.lda * 2, stepB = (long)64 * g.ldb * 2;
    int ld_kt = 0;
    unsigned ld_buf = 0;
    auto dmaA = [&](int h) {
        const char* sb = baseA + ld_kt * stepA;
#pragma unroll
        for (int j = 0; j < 2; ++j) g8w_dma(dst_w + ld_buf + (unsigned)(h * HALF_B + j * 1024), sb, offA[h][j]);
    };
    auto dmaB = [&](int h) {
        const char* sb = baseB + ld_kt * stepB;
        if (NB == 3 && h == 1) { g8w_dma(lds_base + (unsigned)wave * 1024u + ld_buf + (unsigned)(3 * HALF_B), sb, offB[1][0]); return; }
#pragma unroll
        for (int j = 0; j < 2; ++j) g8w_dma(dst_w + ld_buf + (unsigned)((2 + h) * HALF_B + j * 1024), sb, offB[h][j]);
    };
    auto advance = [&]() {       // past the end the cursor re-reads the first K-tile (nobody reads those images): constant vmcnt distance
        ld_buf ^= BUF_B;
        if (++ld_kt == nk) ld_kt = 0;
    };

    // ---- DropPath: the common factor of the kept samples ----
    float s_ref = 1.f, inv_ref = 1.f;
    if (g.rowscale) {
        float mx = 0.f;
        for (int i = lane; i < g.nscale; i += 64) mx = fmaxf(mx, g.rowscale[i]);
        s_ref = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, wave_max(mx))));
        __builtin_amdgcn_s_waitcnt(0x0F70);          // (no load of the compiler's may be pending when the counted LDS-DMA loop starts)
        if (s_ref <= 0.f) s_ref = 1.f;
        inv_ref = 1.f / s_ref;
    }

    bf16x8 fa[4][2], fb0[2][2], fb1[NB1][2];
    f32x4 acc[8][NB];
    f32x4 accb[2];
#pragma unroll
    for (int mb = 0; mb < 8; ++mb)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[mb][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
    accb[0] = accb[1] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 ones;
#pragma unroll
    for (int i = 0; i < 8; ++i) ones[i] = (bf16_t)1.0f;
    const bool do_colsum = g.colsum != nullptr && nblk == 0;

    auto trfrag = [&](unsigned addr, int stride4) -> bf16x8 {      // rows +0..3 and +4..7 of one 16-column block
        const short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4p8w)(smem + addr));
        const short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4p8w)(smem + addr + stride4));
        short8w v;
        v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3]; v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
        return __builtin_bit_cast(bf16x8, v);
    };
    auto scale_frag = [&](bf16x8 v, float s) -> bf16x8 {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = (bf16_t)((float)v[i] * s);
        return v;
    };

#define G8W_LD_A(mq)                                                                                             \
    _Pragma("unroll") for (int mbl = 0; mbl < 4; ++mbl) {                                                        \
        fa[mbl][0] = trfrag(la[mbl] + (mq) * HALF_B, 1024);                                                      \
        fa[mbl][1] = trfrag(la[mbl] + (mq) * HALF_B + 8192, 1024);                                               \
    }
    // a K-tile whose DropPath factor is neither 0 nor the common one (not a DropPath mask: the general rowscale contract): its dY
    // fragments are scaled, behind the barrier, on a wave-uniform branch the common case never takes
#define G8W_SCALE_A()                                                                                            \
    if (general) {                                                                                               \
        _Pragma("unroll") for (int mbl = 0; mbl < 4; ++mbl) {                                                    \
            fa[mbl][0] = scale_frag(fa[mbl][0], ratio); fa[mbl][1] = scale_frag(fa[mbl][1], ratio);              \
        }                                                                                                        \
    }
#define G8W_LD_B0()                                                                                              \
    _Pragma("unroll") for (int nbl = 0; nbl < 2; ++nbl) {                                                        \
        fb0[nbl][0] = trfrag(lb[nbl], 1024);                                                                     \
        fb0[nbl][1] = trfrag(lb[nbl] + 8192, 1024);                                                              \
    }
#define G8W_LD_B1()                                                                                              \
    _Pragma("unroll") for (int nbl = 0; nbl < NB1; ++nbl) {                                                      \
        fb1[nbl][0] = trfrag(lc[nbl], NB == 4 ? 1024 : 512);                                                     \
        fb1[nbl][1] = trfrag(lc[nbl] + (NB == 4 ? 8192 : 4096), NB == 4 ? 1024 : 512);                           \
    }
#define G8W_MMA(mq, nq, fb)                                                                                      \
    if (live) {                                                                                                  \
        __builtin_amdgcn_s_setprio(1);                                                                           \
        _Pragma("unroll") for (int kk = 0; kk < 2; ++kk)                                                         \
            _Pragma("unroll") for (int mbl = 0; mbl < 4; ++mbl)                                                  \
                _Pragma("unroll") for (int nbl = 0; nbl < ((nq) == 0 ? 2 : NB1); ++nbl)                          \
                    acc[(mq) * 4 + mbl][(nq) * 2 + nbl] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(               \
                        fb[nbl][kk], fa[mbl][kk], acc[(mq) * 4 + mbl][(nq) * 2 + nbl], 0, 0, 0);                 \
        __builtin_amdgcn_s_setprio(0);                                                                           \
    }
    // bias gradient: wave (wr, wc) owns the column sums of its dY blocks 2 wc, 2 wc + 1 (half mq = wc >> 1)
#define G8W_COLSUM2(b0, b1)                                                                                      \
    _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) {                                                           \
        accb[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, fa[b0][kk], accb[0], 0, 0, 0);                   \
        accb[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, fa[b1][kk], accb[1], 0, 0, 0);                   \
    }
#define G8W_COLSUM(mq)                                                                                           \
    if (live && do_colsum && (wc >> 1) == (mq)) {                                                                \
        if (wc & 1) { G8W_COLSUM2(2, 3) } else { G8W_COLSUM2(0, 1) }                                             \
    }

    // ---- prologue ----
    dmaB(0); dmaA(0); dmaB(1); dmaA(1); advance();
    dmaB(0); dmaA(0); dmaB(1);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VM_P4) : "memory");
    G8W_BAR();
    if (wr == 1) G8W_BAR();

    for (int s = 0; s < nk; ++s) {
        // this K-tile's DropPath factor through the scalar cache (a vector load would drain the LDS-DMA queue at its wait)
        bool live = true, general = false;
        float ratio = 1.f;
        if (g.rowscale) {
            float sc;
            const float* sp = g.rowscale + __builtin_amdgcn_readfirstlane((tok0 + s * 64) / g.rows_per_scale);
            asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(sc) : "s"(sp) : "memory");
            live = sc != 0.f;
            general = live && sc != s_ref;
            ratio = sc * inv_ref;
        }
        // phase 1
        G8W_LD_B0();
        __builtin_amdgcn_sched_barrier(0);
        G8W_LD_A(0);
        dmaA(1); advance();
        asm volatile("s_waitcnt lgkmcnt(15)" ::: "memory");      // (4-bit counter) the 8 X-h0 reads, issued first, have left LDS: X-h0 may be refilled next phase
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VM_P1) : "memory");
        G8W_BAR();
        G8W_SCALE_A();
        G8W_MMA(0, 0, fb0);
        G8W_COLSUM(0);
        G8W_BAR();
        // phase 2
        G8W_LD_B1();
        dmaB(0);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VM_P2) : "memory");
        G8W_BAR();
        G8W_MMA(0, 1, fb1);
        G8W_BAR();
        // phase 3
        G8W_LD_A(1);
        dmaA(0);
        G8W_BAR();
        G8W_SCALE_A();
        G8W_MMA(1, 1, fb1);
        G8W_COLSUM(1);
        G8W_BAR();
        // phase 4
        dmaB(1);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VM_P4) : "memory");
        G8W_BAR();
        G8W_MMA(1, 0, fb0);
        G8W_BAR();
#pragma unroll
        for (int i = 0; i < 4; ++i) la[i] ^= BUF_B;
#pragma unroll
        for (int i = 0; i < 2; ++i) lb[i] ^= BUF_B;
#pragma unroll
        for (int i = 0; i < NB1; ++i) lc[i] ^= BUF_B;
    }
    if (wr == 0) G8W_BAR();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // ---- slab tile: lane = n_out row m0 + wr*128 + mb*16 + l15, 4 consecutive k_in at n0 + wc*WN + nb*16 + 4 kg ----
    // Written as they lie, an instruction stores 16 rows x 64 bytes (half cache lines; the CU's vector-memory path is paced by the lines an
    // instruction touches, gemm8.hip). The lanes of adjacent rows (l15 even / odd) trade the blocks of a pair (nb, nb + 1): the even lane keeps
    // block nb of both rows, the odd lane block nb + 1 - one instruction then writes 8 rows x 128 contiguous bytes.
    float* Cz = g.C + (long)z * g.slab_stride;
    const bool odd = (l15 & 1) != 0;
#pragma unroll
    for (int mb = 0; mb < 8; ++mb) {
        const int row = m0 + wr * 128 + mb * 16 + l15;
        const int re = row - (odd ? 1 : 0);                      // the pair's even row
#pragma unroll
        for (int nb = 0; nb + 1 < NB; nb += 2) {
            const f32x4 a0 = acc[mb][nb] * s_ref, a1 = acc[mb][nb + 1] * s_ref;
            f32x4 d0, d1;                                        // rows re, re + 1
#pragma unroll
            for (int k = 0; k < 4; ++k) {                        // even lane gives block nb + 1, odd lane block nb (DPP quad_perm [1, 0, 3, 2])
                const float give = odd ? a0[k] : a1[k];
                const float got = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, give), 0xB1, 0xF, 0xF, true));
                d0[k] = odd ? got : a0[k];
                d1[k] = odd ? a1[k] : got;
            }
            const int col = n0 + wc * WN + (nb + (odd ? 1 : 0)) * 16 + 4 * kg;
            if (re < g.M && col < g.N) *(f32x4*)(Cz + (long)re * g.N + col) = d0;
            if (re + 1 < g.M && col < g.N) *(f32x4*)(Cz + (long)(re + 1) * g.N + col) = d1;
        }
        if constexpr (NB & 1) {
            const int col = n0 + wc * WN + (NB - 1) * 16 + 4 * kg;
            if (row < g.M && col < g.N) *(f32x4*)(Cz + (long)row * g.N + col) = acc[mb][NB - 1] * s_ref;
        }
    }
    if (do_colsum && kg == 0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = m0 + wr * 128 + (2 * wc + i) * 16 + l15;
            if (row < g.M) g.colsum[(long)z * g.colsum_stride + row] = accb[i][0] * s_ref;
        }
    }
}

}  // namespace

static bool g8w_eligible(int Mtok, int N, int K, int lddy, int ldx, int rows_per_scale, int has_rowscale) {
    if (Mtok % 64 != 0 || Mtok < 512 || N % 8 != 0 || K % 8 != 0 || lddy % 8 != 0 || ldx % 8 != 0 || N < 128 || K < 128) return false;
    if (has_rowscale && (rows_per_scale <= 0 || rows_per_scale % 64 != 0)) return false;
    if ((long)64 * lddy * 2 + (long)N * 2 >= (1L << 31) || (long)64 * ldx * 2 + (long)K * 2 >= (1L << 31)) return false;
    return true;
}

// Plan: tile width and split count for dW[N, K] over Mtok tokens; 0 when the eight-phase kernel does not take the shape.
int pseld_gemm8w_plan(int Mtok, int N, int K, int lddy, int ldx, int rows_per_scale, int has_rowscale, int max_splits, int* bn_out, int* kchunk_out) {
    if (!g8w_eligible(Mtok, N, K, lddy, ldx, rows_per_scale, has_rowscale)) return 0;
    int bn = pseld_knob(KNOB_GEMM8W_BN, 0);
    if (bn != 256 && bn != 192) {
        auto pad = [&](int w) { return (double)pseld_cdiv(K, w) * w; };
        bn = pad(192) < pad(256) ? 192 : 256;
    }
    const int tiles = pseld_cdiv(N, 256) * pseld_cdiv(K, bn);
    if (tiles > 256) return 0;
    int splits = 256 / tiles;
    const int nk_all = Mtok / 64;
    if (splits > nk_all / 8) splits = nk_all / 8;                 // at least 8 K-tiles (512 tokens) per workgroup
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) return 0;
    int kt = pseld_cdiv(nk_all, splits);
    splits = pseld_cdiv(nk_all, kt);
    *bn_out = bn; *kchunk_out = kt * 64;
    return splits;
}

static void g8w_fill(G8WArgs& a, const void* dY, const void* X, float* slabs, float* colsum, long slab_stride, long colsum_stride, int Mtok,
                     int N, int K, int lddy, int ldx, int bn, int kchunk, int splits, const float* rowscale, int rows_per_scale) {
    a.A = (const char*)dY; a.B = (const char*)X; a.C = slabs; a.colsum = colsum; a.rowscale = rowscale;
    a.slab_stride = slab_stride; a.colsum_stride = colsum_stride;
    a.M = N; a.N = K; a.Mtok = Mtok; a.lda = lddy; a.ldb = ldx; a.kchunk = kchunk;
    a.nx = pseld_cdiv(K, bn); a.ntile = a.nx * pseld_cdiv(N, 256); a.splits = splits;
    a.rows_per_scale = rows_per_scale > 0 ? rows_per_scale : 1;
    a.nscale = rowscale ? pseld_cdiv(Mtok, a.rows_per_scale) : 0;
    a.pad_ = 0;
}
template <typename P>
static int g8w_launch_group(const P& grp, int bn, hipStream_t stream) {
    if (bn == 192) {
        static bool attr = false;
        if (!attr) { (void)hipFuncSetAttribute((const void*)gemm8w_kernel<3, P>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_B); attr = true; }
        hipLaunchKernelGGL((gemm8w_kernel<3, P>), dim3(256), dim3(512), LDS_B, stream, grp);
    } else {
        static bool attr = false;
        if (!attr) { (void)hipFuncSetAttribute((const void*)gemm8w_kernel<4, P>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_B); attr = true; }
        hipLaunchKernelGGL((gemm8w_kernel<4, P>), dim3(256), dim3(512), LDS_B, stream, grp);
    }
    PSELD_LAUNCH_CHECK("gemm8w");
    return PSELD_OK;
}

int pseld_gemm8w_launch(const void* dY, const void* X, float* slabs, float* colsum, long slab_stride, long colsum_stride, int Mtok, int N,
                        int K, int lddy, int ldx, int bn, int kchunk, int splits, const float* rowscale, int rows_per_scale, hipStream_t stream) {
    G8WOne one;
    g8w_fill(one.p, dY, X, slabs, colsum, slab_stride, colsum_stride, Mtok, N, K, lddy, ldx, bn, kchunk, splits, rowscale, rows_per_scale);
    return g8w_launch_group(one, bn, stream);
}

void pseld_reduce_slabs(const float* slabs, float* out, long n, int splits, long slab_stride, int accumulate, hipStream_t stream);

// ---- the weight gradients of several Linear layers (a stage's qkv / proj / fc1 / fc2 of every block, its PatchMerging reduction) in ONE
// launch. A single weight matrix has 4-16 output tiles of 256 x 192, so on its own it is split ~20 ways over the tokens to fill the chip
// and pays one fp32 slab per workgroup (49 MB written and read back per launch: as much as its operands); together the ~25 matrices of
// stage 2 have 252 tiles - one per CU, no token split, no slab, no reduction: every workgroup walks ALL the tokens for its tile and
// writes the gradient itself. Matrices the eight-phase kernel does not take are reported back (return value = bit mask of skipped ones).
extern "C" long pseld_gemm_wgrad_group_workspace(int count, const int* Mtok, const int* N, const int* K) {
    long need = 0;
    for (int i = 0; i < count; ++i) {
        const int tiles = pseld_cdiv(N[i], 256) * pseld_cdiv(K[i], 256);     // (fewest tiles -> most splits)
        int s = 256 / (tiles > 0 ? tiles : 1);
        if (s > Mtok[i] / 512) s = Mtok[i] / 512;
        if (s < 1) s = 1;
        need += (long)(s + 1) * ((long)N[i] * K[i] + N[i]) * (long)sizeof(float);
    }
    return need;
}
extern "C" int pseld_gemm_wgrad_group(int count, const void* const* dY, const void* const* X, float* const* dW, float* const* dbias,
                                      const int* Mtok, const int* N, const int* K, const int* lddy, const int* ldx,
                                      const float* const* rowscale, const int* rows_per_scale, float* workspace, long workspace_bytes,
                                      unsigned* skipped_mask, void* stream) {
    PSELD_CHECK_ARG(count > 0 && count <= 32 && dY && X && dW && Mtok && N && K && lddy && ldx && skipped_mask, "gemm_wgrad_group: bad arguments (count %d)", count);
    hipStream_t s = (hipStream_t)stream;
    unsigned skipped = 0;
    int idx[G8W_MAXP], n = 0;
    for (int i = 0; i < count; ++i) {
        PSELD_CHECK_ARG(dY[i] && X[i] && dW[i], "gemm_wgrad_group: null operand in entry %d", i);
        // (a matrix that ends up with one split is written as 16-byte pieces straight into dW: the slot must be 16-byte aligned)
        const bool ok = g8w_eligible(Mtok[i], N[i], K[i], lddy[i], ldx[i], rows_per_scale ? rows_per_scale[i] : 1, rowscale && rowscale[i]) &&
                        N[i] >= 192 && K[i] >= 192 && Mtok[i] >= 4096 && ((unsigned long)dW[i] & 15) == 0;
        if (ok) idx[n++] = i; else skipped |= 1u << i;
    }
    *skipped_mask = skipped;
    static G8WGroup grp;
    int at = 0;
    while (at < n) {
        // the longest run of matrices whose tiles fit one resident round at one of the two tile widths; at equal reach the width with
        // less padded work
        int reach[2] = {at, at};
        double work[2] = {0, 0};
        for (int w = 0; w < 2; ++w) {
            const int bn = w ? 256 : 192;
            int tiles = 0, e = at;
            while (e < n) {
                const int i = idx[e];
                const int t = pseld_cdiv(N[i], 256) * pseld_cdiv(K[i], bn);
                if (tiles + t > 256) break;
                tiles += t; work[w] += (double)t * bn * Mtok[i]; ++e;
            }
            reach[w] = e;
        }
        const int pick = reach[1] > reach[0] ? 1 : (reach[1] < reach[0] ? 0 : (work[1] < work[0] ? 1 : 0));
        const int best_bn = pick ? 256 : 192, best_end = reach[pick];
        if (best_end == at) { skipped |= 1u << idx[at]; ++at; continue; }      // a single matrix with more than 256 tiles
        const int bn = best_bn;
        int tiles = 0;
        for (int q = at; q < best_end; ++q) tiles += pseld_cdiv(N[idx[q]], 256) * pseld_cdiv(K[idx[q]], bn);
        int S = 256 / tiles;
        for (int q = at; q < best_end; ++q) { const int lim = (Mtok[idx[q]] / 64) / 8; if (S > lim) S = lim; }
        if (S < 1) S = 1;
        long woff = 0;           // floats
        grp.count = 0; grp.first[0] = 0;
        bool fits = true;
        for (int q = at; q < best_end; ++q) {
            const int i = idx[q];
            const int nk_all = Mtok[i] / 64, kt = pseld_cdiv(nk_all, S), sp = pseld_cdiv(nk_all, kt);
            float* db = dbias ? dbias[i] : nullptr;
            G8WArgs& a = grp.p[grp.count];
            if (sp == 1) {
                g8w_fill(a, dY[i], X[i], dW[i], db, 0, 0, Mtok[i], N[i], K[i], lddy[i], ldx[i], bn, kt * 64, 1, rowscale ? rowscale[i] : nullptr, rows_per_scale ? rows_per_scale[i] : 1);
            } else {
                const long per = (long)N[i] * K[i] + N[i];
                if ((woff + (long)sp * per) * (long)sizeof(float) > workspace_bytes || !workspace) { fits = false; break; }
                g8w_fill(a, dY[i], X[i], workspace + woff, db ? workspace + woff + (long)N[i] * K[i] : nullptr, per, per, Mtok[i], N[i], K[i], lddy[i], ldx[i], bn, kt * 64, sp,
                         rowscale ? rowscale[i] : nullptr, rows_per_scale ? rows_per_scale[i] : 1);
                woff += (long)sp * per;
            }
            grp.first[grp.count + 1] = grp.first[grp.count] + a.ntile * a.splits;
            ++grp.count;
        }
        PSELD_CHECK_ARG(fits, "gemm_wgrad_group: workspace %ld bytes too small", workspace_bytes);
        const int rc = g8w_launch_group(grp, bn, s);
        if (rc != PSELD_OK) return rc;
        for (int q = 0; q < grp.count; ++q) {
            const G8WArgs& a = grp.p[q];
            if (a.splits > 1) {
                const int i = idx[at + q];
                const long nk = (long)N[i] * K[i];
                pseld_reduce_slabs(a.C, dW[i], nk, a.splits, a.slab_stride, 0, s);
                if (a.colsum) pseld_reduce_slabs(a.colsum, dbias[i], (long)N[i], a.splits, a.colsum_stride, 0, s);
            }
        }
        PSELD_LAUNCH_CHECK("gemm_wgrad_group(reduce)");
        at = best_end;
    }
    *skipped_mask = skipped;
    return PSELD_OK;
}
