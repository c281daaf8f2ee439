// Row-panel-stationary bf16 GEMM for the tall-skinny Linear products of HTS-AT with K = 192 / 384 (stages 1-2: qkv, attn.proj, mlp.fc1 and
// the input gradients of proj / fc2; htsat.py:118,140, model_utilities.py:166-170) on gfx950.
//
// C[M,N] = A[M,K] B[N,K]^T with the fused epilogues of pseld_gemm (+ bias, x DropPath factor, x aux, + residual, GELU pair), bit for bit
// what gemm8.hip writes (same MFMA, same operand slots, same K order, same epilogue arithmetic), so a launch may take either kernel.
//
// Why a second kernel (round 6; profiles/r06_dispatch_timeline.txt, docs/EXPERIMENTS.md): these products are HBM-bound by shape, and the
// eight-phase tile loop stages A AND B through LDS once per 256 x 192 output tile - 343 KB of LDS-DMA per 37.7 MFLOP at K = 384, the A rows
// six times per row block. What paces a CU is the rate at which bytes are delivered to it (HBM ~10-12, memory-side cache ~14, L2 ~29
// B/clk/CU): 196 KB / 12 + 147 KB / 29 = 21k cycles per tile is the measured K loop. Here the A operand never passes through LDS and is
// read from memory ONCE:
//  * a workgroup (8 waves) owns a PANEL of 64 MB rows (MB = 3: 192 rows -> M = 49 152 is exactly one panel per CU); wave (wrow, hw) keeps
//    the MFMA fragments of rows wrow*16MB .. +16MB of A for the WHOLE K in registers (K = 384, MB = 3: 144 registers), loaded straight
//    from global memory in fragment layout;
//  * the weights stream through a 3- or 4-slot LDS ring in tiles of 64 columns x K (48 / 24 KB, LDS-DMA, the 16-byte chunk XOR-swizzled on
//    the source address as in gemm8.hip: conflict-free ds_read_b128); every workgroup walks the same tiles in the same order, so B is
//    served by L2. The two waves of a row group (hw = 0 / 1, the two waves of a SIMD) take the two 32-column halves of the tile;
//  * ONE s_barrier per tile (slot hand-over), counted vmcnt; the epilogue stores 16 rows x 64 B per instruction straight from the
//    accumulators (a lane holds 8 consecutive columns of one token row: the B image -> weight row map of gemm8.hip at NB = 2), and nothing
//    waits for a store: stores are only ever OLDER than the youngest awaited LDS-DMA by a whole tile.
#include "gemm8.h"
#include <stdio.h>
#include <type_traits>

namespace {

typedef __attribute__((address_space(3))) void* lds_vptr8p;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4p;

enum { P_PLAIN = 0, P_RESID = 1, P_MULAUX = 2, P_GELU_DUAL = 3 };
constexpr int P_BIAS_FLOATS = 4096;          // 3 x 48 KB of ring + 16 KB = the CU's 160 KB

struct G8PArgs {
    const char* A; const char* B; bf16_t* C; bf16_t* C2;
    const float* bias; const bf16_t* resid; const bf16_t* aux; const float* rowscale;
    int M, N, K, lda, ldb, ldc, ldr, ldaux, rows_per_scale;
    float inv_rps;
    int nt, npanels;
    unsigned long long* dbg;     // diagnostic instantiation only: per (workgroup, wave group) cycle sums of the tile loop's phases
};

__device__ __forceinline__ int p_div(int x, int d, float rd) {       // x / d for 0 <= x < 2^24 (one fix-up step)
    int q = (int)((float)x * rd);
    const int r = x - q * d;
    q += (r >= d) - (r < 0);
    return q;
}
// one LDS-DMA wave-instruction: 64 lanes x 16 B from (uniform base + per-lane 32-bit offset + OFF) to 1 KiB of LDS at a uniform address
template <int OFF>
__device__ __forceinline__ void p_dma(unsigned lds_dst, const void* sbase, unsigned voff) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3 offset:%4\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(lds_dst), "s"(sbase), "n"(OFF) : "memory");
}
template <int KT> struct PDmaTile {          // the KT K-tile images (8 rows of this wave each) of one 64-column weight tile
    static __device__ __forceinline__ void go(unsigned dst, const void* sbase, unsigned voff) {
        PDmaTile<KT - 1>::go(dst, sbase, voff);
        // (the instruction's immediate offset is added to the global address AND to the LDS address in M0: the image base is lowered by it)
        p_dma<(KT - 1) * 128>(dst + (unsigned)((KT - 1) * (8192 - 128)), sbase, voff);
    }
};
template <> struct PDmaTile<0> { static __device__ __forceinline__ void go(unsigned, const void*, unsigned) {} };

__device__ __forceinline__ unsigned p_pack2(float a, float b) {
    bf16x2 t;
    t[0] = (bf16_t)a; t[1] = (bf16_t)b;
    return __builtin_bit_cast(unsigned, t);
}
__device__ __forceinline__ void p_unpack2(unsigned p, float& a, float& b) {
    const bf16x2 t = __builtin_bit_cast(bf16x2, p);
    a = (float)t[0]; b = (float)t[1];
}

#define P_BAR()                                   \
    do {                                          \
        __builtin_amdgcn_sched_barrier(0);        \
        __builtin_amdgcn_s_barrier();             \
        __builtin_amdgcn_sched_barrier(0);        \
    } while (0)

// KS = K / 32 MFMA steps (6: K = 192, 12: K = 384); MB = 16-row blocks per wave (panel = 64 MB rows); NSLOT = ring slots.
// STAG: waves 4-7 (the second wave of every SIMD) run their epilogue one tile late, under the first half's MFMAs (the two waves of a SIMD
// otherwise reach matrix work, epilogue and barrier together: MI355X_MICROARCH.md, two waves per SIMD, item 9).
template <int MODE, bool SCALED, int KS, int MB, int NSLOT, bool STAG, bool DBG = false>
__global__ __launch_bounds__(512, 2) void gemm8p_kernel(const G8PArgs g) {
    constexpr int KT = KS / 2;                   // 128-byte K-tile images per weight tile
    constexpr int TILE_B = KT * 8192;            // one weight tile: KT images of 64 rows x 128 B
    constexpr int D = NSLOT - 1;                 // tiles the LDS-DMA cursor runs ahead
    constexpr int NS = (MODE == P_GELU_DUAL ? 2 : 1) * MB;       // store instructions per wave and epilogue
    constexpr int VM_STRICT = (D - 1) * KT;      // everything younger than the awaited tile that is CERTAINLY in the queue
    constexpr int VM_RELAX = (D - 1) * KT + D * NS;              // ... when every epilogue in between issued all its stores
    static_assert(VM_RELAX < 64, "vmcnt is a 6-bit counter");
    constexpr bool HAS_X = MODE == P_RESID || MODE == P_MULAUX;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* bias_s = (float*)(smem + NSLOT * TILE_B);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wrow = wave & 3, hw = wave >> 2;
    const int l15 = lane & 15, q = lane >> 4;
    const int nwg = gridDim.x;
    const int my_panels = (g.npanels - (int)blockIdx.x + nwg - 1) / nwg;
    if (my_panels <= 0) return;

    for (int i = tid; i < g.N; i += 512) bias_s[i] = g.bias ? g.bias[i] : 0.f;

    // ---- LDS-DMA source offset of this lane: image row rho = wave*8 + lane/8 of every K-tile image <- weight row of the 64-column tile.
    // Image rows [32 hw + 16 nbl + i] hold the weight rows 32 hw + 8 (i >> 2) + 4 nbl + (i & 3): accumulator row i = 4 q + k of block nbl is
    // column 8 q + 4 nbl + k of the wave's 32-column half, i.e. a lane ends up with 8 consecutive columns of its token row.
    const unsigned lds_base = (unsigned)(unsigned long)(lds_vptr8p)smem;
    unsigned voffB;
    {
        const int rho = wave * 8 + (lane >> 3);
        const int ch = (lane & 7) ^ ((rho >> 1) & 7);
        const int within = rho & 31, nbl = within >> 4, i = within & 15;
        const int wr_n = (rho >> 5) * 32 + 8 * (i >> 2) + 4 * nbl + (i & 3);
        voffB = (unsigned)wr_n * (unsigned)(g.ldb * 2) + (unsigned)(ch * 16);
    }
    int ld_nt = 0;
    unsigned ld_slot = 0;
    const unsigned tile_stride = (unsigned)(64 * g.ldb * 2);
    auto dma_tile = [&]() __attribute__((always_inline)) {
        // (the cursor is advanced in both wave-group programs: made uniform again explicitly, the asm operands must be SGPRs)
        const unsigned long src = (unsigned long)g.B + (unsigned long)ld_nt * tile_stride;
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)src), hi = __builtin_amdgcn_readfirstlane((unsigned)(src >> 32));
        PDmaTile<KT>::go(__builtin_amdgcn_readfirstlane(lds_base + ld_slot + (unsigned)(wave * 1024)), (const void*)(((unsigned long)hi << 32) | lo), voffB);
        ld_slot = ld_slot + TILE_B == (unsigned)(NSLOT * TILE_B) ? 0u : ld_slot + TILE_B;
        ld_nt = ld_nt + 1 == g.nt ? 0 : ld_nt + 1;
    };
#pragma unroll
    for (int d = 0; d < D; ++d) dma_tile();

    // ---- fragment read address (slot 0, K-tile image 0, block 0, kk 0): row 32 hw + l15, chunk q ^ (l15 >> 1); kk = 1 flips bit 6 ----
    const unsigned rb0 = (unsigned)((hw * 32 + l15) * 128 + ((q ^ (l15 >> 1)) << 4));
    const unsigned rb1 = rb0 ^ 64u;

    bf16x8 fa[MB][KS];
    f32x4 acc[MB][2];
    const int mlast = g.M - 1;

    auto load_panel = [&](int m0) __attribute__((always_inline)) {
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
            const int row = min(m0 + wrow * (16 * MB) + mb * 16 + l15, mlast);
            const char* p = g.A + (long)row * (g.lda * 2) + q * 16;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) fa[mb][ks] = *(const bf16x8*)(p + ks * 64);
        }
    };
    auto mma_tile = [&](unsigned slot, int n0) __attribute__((always_inline)) {
        {
            const float* bp = bias_s + n0 + hw * 32 + 8 * q;
            const f32x4 b0 = *(const f32x4*)bp, b1 = *(const f32x4*)(bp + 4);
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) { acc[mb][0] = b0; acc[mb][1] = b1; }
        }
        const char* s0 = smem + slot + rb0;
        const char* s1 = smem + slot + rb1;
        // the fragments of step ks + 1 are read while the MFMAs of step ks issue (left to itself hipcc reads a step's two fragments right in
        // front of its six MFMAs: ~100 cycles of LDS latency per step, and the two waves of a SIMD, which leave the barrier together, stall
        // at the same time)
        bf16x8 f[2][2];
        f[0][0] = *(const bf16x8*)s0; f[0][1] = *(const bf16x8*)(s0 + 2048);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            if (ks + 1 < KS) {
                const char* sp = (((ks + 1) & 1) ? s1 : s0) + ((ks + 1) >> 1) * 8192;
                f[(ks + 1) & 1][0] = *(const bf16x8*)sp; f[(ks + 1) & 1][1] = *(const bf16x8*)(sp + 2048);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                acc[mb][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f[ks & 1][0], fa[mb][ks], acc[mb][0], 0, 0, 0);
                acc[mb][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f[ks & 1][1], fa[mb][ks], acc[mb][1], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // lane (q, l15): token row m0 + wrow*16MB + 16 mb + l15, columns n0 + 32 hw + 8 q .. + 8 = a[mb][0][0..3], a[mb][1][0..3]: one 16-byte
    // store (and one 16-byte residual / aux load) per row block. The epilogue's operands (residual / aux pieces, DropPath factor) are
    // fetched a whole tile ahead (xload) - a load issued inside the epilogue exposes one memory latency per 64-column tile (stamped: 5k
    // cycles per tile for the scaled-aux epilogue). FULL: every row of the wave is inside M - the stores are unconditional, their number
    // is what the relaxed vmcnt of the tile loop counts on.
    u32x4p xr[HAS_X ? MB : 1];
    float scr[SCALED ? MB : 1];
    auto xload = [&](int m0, int n0) __attribute__((always_inline)) {
        if constexpr (HAS_X || SCALED) {
            const int col = n0 + hw * 32 + 8 * q;
            const int rbase = m0 + wrow * (16 * MB) + l15;
            const bf16_t* X = MODE == P_RESID ? g.resid : g.aux;
            const int ldx = MODE == P_RESID ? g.ldr : g.ldaux;
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                const int row = min(rbase + mb * 16, mlast);
                if constexpr (SCALED) scr[SCALED ? mb : 0] = g.rowscale[p_div(row, g.rows_per_scale, g.inv_rps)];
                if constexpr (HAS_X) xr[HAS_X ? mb : 0] = *(const u32x4p*)(X + (long)row * ldx + col);
            }
        }
    };
    auto epilogue = [&](int m0, int n0, auto full_tag) __attribute__((always_inline)) {
        constexpr bool FULL = decltype(full_tag)::value;
        const int col = n0 + hw * 32 + 8 * q;
        const int rbase = m0 + wrow * (16 * MB) + l15;
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
            const int row = rbase + mb * 16;
            const float scm = SCALED ? scr[SCALED ? mb : 0] : 1.f;
            u32x4p o, o2;
#pragma unroll
            for (int c = 0; c < 8; c += 2) {
                float v0 = acc[mb][c >> 2][c & 3], v1 = acc[mb][c >> 2][(c & 3) + 1];
                if constexpr (MODE == P_GELU_DUAL) {
                    f32x2 xx = {v0, v1}, yy, dd;
                    gelu_both2(xx, yy, dd);
                    o[c / 2] = p_pack2(yy[0], yy[1]); o2[c / 2] = p_pack2(dd[0], dd[1]);
                } else {
                    float xa = 0.f, xb = 0.f;
                    if constexpr (HAS_X) p_unpack2(xr[HAS_X ? mb : 0][c / 2], xa, xb);
                    if constexpr (MODE == P_PLAIN) { if constexpr (SCALED) { v0 *= scm; v1 *= scm; } }
                    else if constexpr (MODE == P_RESID) { v0 = SCALED ? fmaf(v0, scm, xa) : v0 + xa; v1 = SCALED ? fmaf(v1, scm, xb) : v1 + xb; }
                    else { v0 *= SCALED ? xa * scm : xa; v1 *= SCALED ? xb * scm : xb; }
                    o[c / 2] = p_pack2(v0, v1);
                }
            }
            const long off = (long)(FULL ? row : min(row, mlast)) * g.ldc + col;
            if (FULL || row < g.M) {
                *(u32x4p*)(g.C + off) = o;
                // gelu'(u) is read again only in the backward pass: non-temporal, as in gemm8.hip
                if constexpr (MODE == P_GELU_DUAL) __builtin_nontemporal_store(o2, (u32x4p*)(g.C2 + off));
            }
        }
    };
    auto epi = [&](int m0, int n0, bool full) __attribute__((always_inline)) {
        if (full) epilogue(m0, n0, std::true_type{});
        else epilogue(m0, n0, std::false_type{});
    };

    // ---- the tile stream. One tile = two phases between workgroup barriers X_s | Y_s | X_s+1: in phase A_s (X_s .. Y_s) the first wave of
    // every SIMD (hw = 0) runs the matrix part of tile s while the second (hw = 1) runs the epilogue of tile s - 1, prefetches the operands
    // of its next epilogue and issues its share of the LDS-DMA of tile s + D; in phase B_s (Y_s .. X_s+1) they trade places. So the matrix
    // pipe of a SIMD always has one wave feeding it and the other wave's vector / memory instructions beside it (stamped, one barrier per
    // tile and both waves in step: 3.5k cycles per tile for 2.3k of matrix work; MI355X_MICROARCH.md, two waves per SIMD, items 5 and 9).
    // Slot of tile s + D = slot of tile s - 1 (D = NSLOT - 1): free behind X_s, when hw = 1 has finished its matrix part of s - 1.
    // vmcnt: in both programs an iteration issues {stores NS, operand loads NXL, LDS-DMA KT} in that order and DMA(s) is D iterations old
    // when it is awaited, so (D - 1) whole iterations are younger: all of them may stay in flight. A wave whose rows are not all inside M
    // may have skipped stores: it counts without them; the first D tiles wait for everything.
    constexpr int NXL = ((HAS_X ? 1 : 0) + (SCALED ? 1 : 0)) * MB;
    constexpr int VM2_STRICT = (D - 1) * (NXL + KT), VM2_RELAX = (D - 1) * (NS + NXL + KT);
    static_assert(VM2_RELAX < 64, "vmcnt is a 6-bit counter");
    __syncthreads();                                   // the bias copy
    [[maybe_unused]] unsigned long long c_wait = 0, c_bar = 0, c_dma = 0, c_mma = 0, c_epi = 0, c_panel = 0, c_bar2 = 0, t_entry = 0, t0 = 0, t1 = 0;
    [[maybe_unused]] unsigned long long r_entry = 0;
    if constexpr (DBG) { t_entry = __builtin_amdgcn_s_memtime(); r_entry = __builtin_amdgcn_s_memrealtime(); }
#define P_STAMP(acc_) do { if constexpr (DBG) { t1 = __builtin_amdgcn_s_memtime(); acc_ += t1 - t0; t0 = t1; } } while (0)
    unsigned rd_slot = 0;
    const int S = my_panels * g.nt;
    auto panel_m0 = [&](int pi) { return ((int)blockIdx.x + pi * nwg) * (64 * MB); };
    auto wave_full = [&](int m0) { return m0 + wrow * (16 * MB) + 16 * MB <= g.M; };
    auto next_slot = [&]() { rd_slot = rd_slot + TILE_B == (unsigned)(NSLOT * TILE_B) ? 0u : rd_slot + TILE_B; };
    if constexpr (!STAG) {
        // (kept for A/B: one barrier per tile, both waves of a SIMD in step)
        int s = 0;
        bool prev_full = true;
        for (int pi = 0; pi < my_panels; ++pi) {
            const int m0 = panel_m0(pi);
            const bool full = wave_full(m0);
            if constexpr (DBG) t0 = __builtin_amdgcn_s_memtime();
            load_panel(m0);
            if constexpr (DBG) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); P_STAMP(c_panel); }
            for (int t = 0; t < g.nt; ++t, ++s) {
                const int n0 = t * 64;
                if constexpr (DBG) t0 = __builtin_amdgcn_s_memtime();
                if (full && prev_full && s > D) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VM_RELAX) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VM_STRICT) : "memory");
                P_STAMP(c_wait);
                P_BAR();
                P_STAMP(c_bar);
                dma_tile();
                P_STAMP(c_dma);
                mma_tile(rd_slot, n0);
                if constexpr (DBG) asm volatile("s_nop 0" ::: "memory");
                P_STAMP(c_mma);
                xload(m0, n0);
                epi(m0, n0, full);
                P_STAMP(c_epi);
                next_slot();
            }
            prev_full = full;
        }
    } else if (hw == 0) {
        int pi = 0, t = 0, m0 = panel_m0(0);
        bool full = wave_full(m0), prev_full = true;
        if constexpr (DBG) t0 = __builtin_amdgcn_s_memtime();
        load_panel(m0);
        xload(m0, 0);
        if constexpr (DBG) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); P_STAMP(c_panel); }
        for (int s = 0; s < S; ++s) {
            const int n0 = t * 64;
            if constexpr (DBG) t0 = __builtin_amdgcn_s_memtime();
            if (s <= D) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if (full && prev_full) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VM2_RELAX) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VM2_STRICT) : "memory");
            P_STAMP(c_wait);
            P_BAR();                                                        // X_s
            P_STAMP(c_bar);
            mma_tile(rd_slot, n0);
            if constexpr (DBG) asm volatile("s_nop 0" ::: "memory");
            P_STAMP(c_mma);
            P_BAR();                                                        // Y_s
            P_STAMP(c_bar2);
            epi(m0, n0, full);
            next_slot();
            if (++t == g.nt) {                                              // the next tile opens a panel: its A rows, behind the last matrix part that read the old ones
                t = 0; prev_full = full;
                if (++pi < my_panels) { m0 = panel_m0(pi); full = wave_full(m0); load_panel(m0); }
            }
            if (s + 1 < S) xload(m0, t * 64);
            P_STAMP(c_epi);
            dma_tile();
            P_STAMP(c_dma);
        }
        P_BAR();                                                            // X_S: the other half's last matrix part
    } else {
        int pi = 0, t = 0, m0 = panel_m0(0);
        bool full = wave_full(m0), prev_full = true;
        int e_m0 = 0, e_n0 = 0;
        bool e_full = true;
        if constexpr (DBG) t0 = __builtin_amdgcn_s_memtime();
        load_panel(m0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                   // tile 0 (and the A rows)
        P_STAMP(c_panel);
        P_BAR();                                                            // X_0
        for (int s = 0; s < S; ++s) {
            const int n0 = t * 64;
            if constexpr (DBG) t0 = __builtin_amdgcn_s_memtime();
            if (s > 0) epi(e_m0, e_n0, e_full);                             // tile s - 1: its operands were fetched a tile ago
            xload(m0, n0);
            P_STAMP(c_epi);
            dma_tile();
            P_STAMP(c_dma);
            P_BAR();                                                        // Y_s
            P_STAMP(c_bar2);
            mma_tile(rd_slot, n0);
            if constexpr (DBG) asm volatile("s_nop 0" ::: "memory");
            P_STAMP(c_mma);
            e_m0 = m0; e_n0 = n0; e_full = full;
            next_slot();
            const bool was_full = full && prev_full;
            if (++t == g.nt) {
                t = 0; prev_full = full;
                if (++pi < my_panels) { m0 = panel_m0(pi); full = wave_full(m0); load_panel(m0); }
            }
            if (s + 1 <= D) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if (was_full && full) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VM2_RELAX) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VM2_STRICT) : "memory");
            P_STAMP(c_wait);
            P_BAR();                                                        // X_s+1
            P_STAMP(c_bar);
        }
        epi(e_m0, e_n0, e_full);
    }
#undef P_STAMP
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the cursor's surplus DMAs must not outlive the workgroup's LDS
    if constexpr (DBG) {
        if (lane == 0 && (wave == 0 || wave == 4)) {
            unsigned long long* d = g.dbg + ((long)blockIdx.x * 2 + hw) * 10;
            d[0] = c_panel; d[1] = c_wait; d[2] = c_bar; d[3] = c_dma; d[4] = c_mma; d[5] = c_epi; d[6] = __builtin_amdgcn_s_memtime() - t_entry; d[7] = c_bar2;
            d[8] = __builtin_amdgcn_s_memrealtime() - r_entry; d[9] = r_entry;          // 100 MHz ticks: lifetime, entry time (launch skew between workgroups)
        }
    }
}

char g_gemm8p_symbuf[80];
const char* g_gemm8p_symbol = "";
int g_gemm8p_force_mb = 0, g_gemm8p_force_stag = -1;
unsigned long long* g_gemm8p_dbg = nullptr;

template <int MODE, bool SCALED, int KS, int MB, bool STAG>
int launch8p(const G8PArgs& a, hipStream_t stream) {
    constexpr int NSLOT = KS == 12 ? 3 : 4;
    constexpr int lds = NSLOT * (KS / 2) * 8192 + P_BIAS_FLOATS * 4;
    snprintf(g_gemm8p_symbuf, sizeof g_gemm8p_symbuf, "gemm8p_kernel<%d, %s, %d, %d, %d, %s>", MODE, SCALED ? "true" : "false", KS, MB, NSLOT, STAG ? "true" : "false");
    g_gemm8p_symbol = g_gemm8p_symbuf;
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute((const void*)gemm8p_kernel<MODE, SCALED, KS, MB, NSLOT, STAG>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); attr = true; }
    int nwg = a.npanels < 256 ? a.npanels : 256;
    if constexpr (MB == 3 && (MODE == P_PLAIN || MODE == P_GELU_DUAL || (MODE == P_MULAUX && SCALED))) {
        if (a.dbg) {          // diagnostic build (three epilogue kinds, 192-row panels): stamps to [workgroup][wave group][8]
            (void)hipFuncSetAttribute((const void*)gemm8p_kernel<MODE, SCALED, KS, MB, NSLOT, STAG, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            hipLaunchKernelGGL((gemm8p_kernel<MODE, SCALED, KS, MB, NSLOT, STAG, true>), dim3((unsigned)nwg), dim3(512), lds, stream, a);
            PSELD_LAUNCH_CHECK("gemm8p(dbg)");
            return PSELD_OK;
        }
    }
    hipLaunchKernelGGL((gemm8p_kernel<MODE, SCALED, KS, MB, NSLOT, STAG>), dim3((unsigned)nwg), dim3(512), lds, stream, a);
    PSELD_LAUNCH_CHECK("gemm8p");
    return PSELD_OK;
}
template <int KS, int MB, bool STAG>
int launch8p_mode(const Gemm8Desc& d, const G8PArgs& a, hipStream_t stream) {
    const bool sc = d.rowscale != nullptr;
    if (d.gelu_dual) return launch8p<P_GELU_DUAL, false, KS, MB, STAG>(a, stream);
    if (d.resid) return sc ? launch8p<P_RESID, true, KS, MB, STAG>(a, stream) : launch8p<P_RESID, false, KS, MB, STAG>(a, stream);
    if (d.aux) return sc ? launch8p<P_MULAUX, true, KS, MB, STAG>(a, stream) : launch8p<P_MULAUX, false, KS, MB, STAG>(a, stream);
    return sc ? launch8p<P_PLAIN, true, KS, MB, STAG>(a, stream) : launch8p<P_PLAIN, false, KS, MB, STAG>(a, stream);
}
}  // namespace

// measurement aid (tools, tests): rows per panel = 64 mb (0 = the launch's own choice), stag -1 = own choice / 0 / 1
extern "C" void pseld_gemm8p_force(int mb, int stag) { g_gemm8p_force_mb = mb; g_gemm8p_force_stag = stag; }
extern "C" void pseld_gemm8p_set_debug_buffer(void* p) { g_gemm8p_dbg = (unsigned long long*)p; }
const char* pseld_gemm8p_last_symbol() { return g_gemm8p_symbol; }

int pseld_gemm8p_supported(const Gemm8Desc& d) {
    if (d.K != 192 && d.K != 384) return 0;
    if (d.N % 64 != 0 || d.N < 128 || d.N > P_BIAS_FLOATS || d.M < 1) return 0;
    if (d.lda % 8 != 0 || d.ldb % 8 != 0 || d.ldc % 8 != 0 || (d.resid && d.ldr % 8 != 0) || (d.aux && d.ldaux % 8 != 0)) return 0;
    if ((long)d.N * d.ldb * 2 >= (1L << 32) || d.M >= (1 << 24)) return 0;
    if ((((unsigned long)d.A | (unsigned long)d.B | (unsigned long)d.C | (unsigned long)d.C2 | (unsigned long)d.resid | (unsigned long)d.aux) & 15) != 0) return 0;
    if (d.resid && d.aux) return 0;
    if (d.gelu_dual && (d.resid || d.aux || d.rowscale || !d.C2)) return 0;
    return 1;
}
// which of the supported products the launch routes here by itself (pseld_gemm): knobs GEMM8P_MODES (bit per epilogue kind: 1 plain,
// 2 residual, 4 aux, 8 GELU pair) and GEMM8P_K (0 = both, 192, 384)
int pseld_gemm8p_wanted(const Gemm8Desc& d) {
    const int kind = d.gelu_dual ? 8 : (d.resid ? 2 : (d.aux ? 4 : 1));
    const int kk = pseld_knob(KNOB_GEMM8P_K, 0);
    return (pseld_knob(KNOB_GEMM8P_MODES, 15) & kind) != 0 && (kk == 0 || kk == d.K);
}

int pseld_gemm8p_launch(const Gemm8Desc& d, hipStream_t stream) {
    G8PArgs a;
    a.A = (const char*)d.A; a.B = (const char*)d.B; a.C = (bf16_t*)d.C; a.C2 = (bf16_t*)d.C2;
    a.bias = d.bias; a.resid = (const bf16_t*)d.resid; a.aux = (const bf16_t*)d.aux; a.rowscale = d.rowscale;
    a.M = d.M; a.N = d.N; a.K = d.K; a.lda = d.lda; a.ldb = d.ldb; a.ldc = d.ldc; a.ldr = d.ldr; a.ldaux = d.ldaux;
    a.rows_per_scale = d.rows_per_scale > 0 ? d.rows_per_scale : 1;
    a.inv_rps = 1.0f / (float)a.rows_per_scale;
    a.nt = d.N / 64;
    a.dbg = g_gemm8p_dbg;
    // Panel height: the cheapest of the built ones by rounds x rows (K = 384: 128 / 192 rows - the A fragments of 256 rows do not fit the
    // register file beside the accumulators; K = 192: 192 / 256 rows)
    int mb = g_gemm8p_force_mb;
    if (mb == 0) {
        long best = 0;
        for (int c = (d.K == 384 ? 2 : 3); c <= (d.K == 384 ? 3 : 4); ++c) {
            const long panels = pseld_cdiv(d.M, 64 * c);
            const long cost = ((panels + 255) / 256) * (64 * c);
            if (mb == 0 || cost <= best) { best = cost; mb = c; }
        }
    }
    a.npanels = pseld_cdiv(d.M, 64 * mb);
    // Two-phase (ping-pong) tile loop or one barrier per tile: the epilogues that are long against the matrix part (GELU pair, residual)
    // measure faster with both waves of a SIMD in step (r06_gemm8p_shapes: fc1 + GELU pair 93.6 against 106.2 us cold), the short ones with
    // the two phases. Knob GEMM8P_STAG (bit per epilogue kind: 1 plain, 2 residual, 4 aux, 8 GELU pair) for A/B runs.
    const int kind = d.gelu_dual ? 8 : (d.resid ? 2 : (d.aux ? 4 : 1));
    const bool stag = g_gemm8p_force_stag >= 0 ? g_gemm8p_force_stag != 0 : (pseld_knob(KNOB_GEMM8P_STAG, 1 | 4) & kind) != 0;
    if (d.K == 384) {
        if (mb == 2) return stag ? launch8p_mode<12, 2, true>(d, a, stream) : launch8p_mode<12, 2, false>(d, a, stream);
        if (mb == 3) return stag ? launch8p_mode<12, 3, true>(d, a, stream) : launch8p_mode<12, 3, false>(d, a, stream);
    } else {
        if (mb == 3) return stag ? launch8p_mode<6, 3, true>(d, a, stream) : launch8p_mode<6, 3, false>(d, a, stream);
        if (mb == 4) return stag ? launch8p_mode<6, 4, true>(d, a, stream) : launch8p_mode<6, 4, false>(d, a, stream);
    }
    pseld_set_error("gemm8p: no instantiation for K = %d, panel blocks %d", d.K, mb);
    return PSELD_ERR_UNSUPPORTED;
}
