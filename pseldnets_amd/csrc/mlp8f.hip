// Fused MLP forward for the C = 192 / 384 blocks of HTS-AT (stages 1-2), bf16, on gfx950:
//   h = gelu(xn W1^T + b1), g = gelu'(xn W1^T + b1)            [M, 4C]  (both saved for the backward pass, as fc1's GELU-pair epilogue does)
//   y = resid + s (h W2^T + b2)                                 [M, C]   (s = the DropPath factor of the row's sample)
// i.e. model_utilities.py:159-171 (Mlp.forward: fc1 -> GELU -> fc2) with the block's DropPath + shortcut (htsat.py:262-264) in ONE launch,
// bit for bit what the two pseld_gemm launches (fc1 + GELU pair, fc2 + residual) write: same MFMA, same operand slots, same K order, same
// epilogue arithmetic. What the fusion removes is fc2's read of h (one [M, 4C] row tensor per block) and a kernel boundary; h and g are
// still written once (the backward reads them).
//
// Geometry ("256-token / 4-wave / 512-register"; built on the row-panel-stationary GEMM of gemm8p.hip):
//  * a workgroup = 4 waves, ONE per SIMD, 512 registers each; it owns a panel of 64 MB token rows, wave w the rows 16 MB w .. + 16 MB.
//    A wave keeps in registers, for the whole panel: the MFMA fragments of its xn rows (K = C: 48 / 96 registers at MB = 2), the fp32 y
//    accumulators of its rows x ALL C output columns (96 / 192) - so the contraction over the 4C hidden units never leaves the registers;
//  * the hidden dimension is walked in tiles of 64 units. Per tile two half-stages, each one slot of an LDS ring filled by LDS-DMA and
//    read by every wave: (a) W1 rows of the 64 units (64 x C) -> u = xn W1^T + b1 for the wave's rows (MFMA 16x16x32, accumulator rows =
//    hidden units in the order that leaves a lane with 8 CONSECUTIVE units of its token row per k-step) -> GELU pair -> h, g stored 16 B per
//    lane, and h - already in the A-operand layout of the next product - stays in registers; (b) W2 columns of the 64 units (C x 64) ->
//    y += h W2^T: two k-steps per tile, in increasing hidden order (the K order of the stand-alone fc2 launch);
//  * every workgroup walks the same weight tiles in the same order: the 2 x 4C x C weights (2.4 MB at C = 384) come from L2, the
//    activations are read once (xn, resid) and written once (h, g, y).
#include "gemm8.h"
#include <stdio.h>
#include <type_traits>

namespace {

typedef __attribute__((address_space(3))) void* lds_vptrf;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4f;

struct M8Args {
    const char* X; const char* W1; const char* W2;
    const float* b1; const float* b2; const bf16_t* resid; const float* rowscale;
    bf16_t* Y; bf16_t* Hh; bf16_t* Hg;
    int M, C, H, ldx, ldw1, ldw2, ldr, ldy, ldh, rows_per_scale;
    float inv_rps;
    int nt, npanels;
    unsigned long long* dbg;
};

__device__ __forceinline__ int f_div(int x, int d, float rd) {
    int q = (int)((float)x * rd);
    const int r = x - q * d;
    q += (r >= d) - (r < 0);
    return q;
}
template <int OFF>
__device__ __forceinline__ void f_dma(unsigned lds_dst, const void* sbase, unsigned voff) {
    unsigned keep;
    // (the immediate offset is added to the global address AND to the LDS address in M0)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3 offset:%4\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(lds_dst), "s"(sbase), "n"(OFF) : "memory");
}
template <int KT> struct FDmaW1 {            // the KT K-tile images of one row group (8 rows) of a W1 tile
    static __device__ __forceinline__ void go(unsigned dst, const void* sbase, unsigned voff) {
        FDmaW1<KT - 1>::go(dst, sbase, voff);
        f_dma<(KT - 1) * 128>(dst + (unsigned)((KT - 1) * (8192 - 128)), sbase, voff);
    }
};
template <> struct FDmaW1<0> { static __device__ __forceinline__ void go(unsigned, const void*, unsigned) {} };

__device__ __forceinline__ unsigned f_pack2(float a, float b) {
    bf16x2 t;
    t[0] = (bf16_t)a; t[1] = (bf16_t)b;
    return __builtin_bit_cast(unsigned, t);
}
__device__ __forceinline__ void f_unpack2(unsigned p, float& a, float& b) {
    const bf16x2 t = __builtin_bit_cast(bf16x2, p);
    a = (float)t[0]; b = (float)t[1];
}
#define F_BAR()                                   \
    do {                                          \
        __builtin_amdgcn_sched_barrier(0);        \
        __builtin_amdgcn_s_barrier();             \
        __builtin_amdgcn_sched_barrier(0);        \
    } while (0)

// KS1 = C / 32 (6: C = 192, 12: C = 384); MB = 16-row blocks per wave (panel = 64 MB rows); NSLOT = ring slots of C x 128 bytes
// NW = waves per workgroup: 4 (one per SIMD, 512 registers: what C = 384 needs for its 192 y accumulators per lane at MB = 2) or 8 (two per
// SIMD, 256 registers: fits C = 192 - x 48 + y 96 + u 32 - and lets one wave's GELU / LDS-DMA issue run beside its SIMD partner's MFMAs)
template <int KS1, int MB, int NSLOT, bool SCALED, bool DBG, int NW = 4>
__global__ __launch_bounds__(64 * NW, NW / 4) void mlp8f_kernel(const M8Args g) {
    constexpr int NT = 64 * NW, RG = 8 / NW;        // threads; 8-row groups of a W1 image this wave fills
    constexpr int C = 32 * KS1, KT1 = KS1 / 2, NBY = C / 16;
    constexpr int SLOT_B = C * 128;                 // a W1 tile (KT1 images of 64 rows x 128 B) and a W2 tile (C rows x 128 B) are the same size
    constexpr int PW = SLOT_B / 1024 / NW;          // LDS-DMA pieces per wave and half-stage
    constexpr int D = NSLOT - 1;                    // half-stages the LDS-DMA cursor runs ahead
    constexpr int NS = 4 * MB;                      // store instructions of a GELU half-stage (h and g, two k-steps, MB row blocks)
    constexpr int VM_STRICT = (D - 1) * PW, VM_RELAX = (D - 1) * PW + NS;      // (among D consecutive half-stages at least one is a GELU half-stage)
    static_assert(VM_RELAX < 64, "vmcnt is a 6-bit counter");
    static_assert(D == 2 || D == 3, "the relaxed count assumes one GELU half-stage among the last D");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* b1s = (float*)(smem + NSLOT * SLOT_B);
    float* b2s = b1s + 4 * C;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, q = lane >> 4;
    const int nwg = gridDim.x;
    const int my_panels = (g.npanels - (int)blockIdx.x + nwg - 1) / nwg;
    if (my_panels <= 0) return;
    for (int i = tid; i < 4 * C; i += NT) b1s[i] = g.b1 ? g.b1[i] : 0.f;
    for (int i = tid; i < C; i += NT) b2s[i] = g.b2 ? g.b2[i] : 0.f;

    // ---- LDS-DMA source offsets. Image row 16 b + i (i = 4 q' + k: accumulator row of block b) holds the weight row 32 (b >> 1) + 8 (i >> 2)
    // + 4 (b & 1) + (i & 3) of the tile: a lane ends up with 8 consecutive columns per block pair - for the u tile that is one k-step's
    // A-operand fragment of the second product, for y one 16-byte store. Chunk XOR-swizzle on the source address as in gemm8.hip.
    const unsigned lds_base = (unsigned)(unsigned long)(lds_vptrf)smem;
    auto tile_row = [](int r) { const int b = r >> 4, i = r & 15; return 32 * (b >> 1) + 8 * (i >> 2) + 4 * (b & 1) + (i & 3); };
    unsigned voff1[RG], voff2[PW];
#pragma unroll
    for (int j = 0; j < RG; ++j) {                  // W1 tile: this wave fills the row groups RG wave + j (8 image rows each) of every K-tile image
        const int rho = (RG * wave + j) * 8 + (lane >> 3);
        const int ch = (lane & 7) ^ ((rho >> 1) & 7);
        voff1[j] = (unsigned)tile_row(rho) * (unsigned)(g.ldw1 * 2) + (unsigned)(ch * 16);
    }
#pragma unroll
    for (int j = 0; j < PW; ++j) {                  // W2 tile: one image of C rows (output columns) x 128 B (the tile's 64 hidden units); pieces wave PW + j
        const int rho = (wave * PW + j) * 8 + (lane >> 3);
        const int ch = (lane & 7) ^ ((rho >> 1) & 7);
        voff2[j] = (unsigned)tile_row(rho) * (unsigned)(g.ldw2 * 2) + (unsigned)(ch * 16);
    }
    int ld_hs = 0;                                  // half-stage of the load cursor within the panel's 2 nt (wraps: every panel walks the same weights)
    unsigned ld_slot = 0;
    auto dma_half = [&]() __attribute__((always_inline)) {
        const int t = ld_hs >> 1;
        const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + ld_slot);
        if ((ld_hs & 1) == 0) {
            const unsigned long src = (unsigned long)g.W1 + (unsigned long)t * (unsigned long)(64 * g.ldw1 * 2);
            const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)src), hi = __builtin_amdgcn_readfirstlane((unsigned)(src >> 32));
            const void* sb = (const void*)(((unsigned long)hi << 32) | lo);
#pragma unroll
            for (int j = 0; j < RG; ++j) FDmaW1<KT1>::go(dst + (unsigned)((RG * wave + j) * 1024), sb, voff1[j]);
        } else {
            const unsigned long src = (unsigned long)g.W2 + (unsigned long)t * 128ul;
            const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)src), hi = __builtin_amdgcn_readfirstlane((unsigned)(src >> 32));
            const void* sb = (const void*)(((unsigned long)hi << 32) | lo);
#pragma unroll
            for (int j = 0; j < PW; ++j) f_dma<0>(dst + (unsigned)((wave * PW + j) * 1024), sb, voff2[j]);
        }
        ld_slot = ld_slot + SLOT_B == (unsigned)(NSLOT * SLOT_B) ? 0u : ld_slot + SLOT_B;
        ld_hs = ld_hs + 1 == 2 * g.nt ? 0 : ld_hs + 1;
    };
    static_assert(RG * KT1 == PW, "a wave's share of a W1 tile (RG row groups x KT1 images) equals its share of a W2 tile");
#pragma unroll
    for (int d = 0; d < D; ++d) dma_half();

    // fragment read base: row l15 of a 16-row block, chunk q ^ (l15 >> 1); kk = 1 flips bit 6
    const unsigned rb0 = (unsigned)(l15 * 128 + ((q ^ (l15 >> 1)) << 4));
    const unsigned rb1 = rb0 ^ 64u;

    bf16x8 xa[MB][KS1];
    f32x4 yacc[MB][NBY];
    f32x4 uacc[MB][4];
    bf16x8 hf[MB][2];
    const int mlast = g.M - 1;

    auto load_x = [&](int m0) __attribute__((always_inline)) {
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
            const int row = min(m0 + wave * (16 * MB) + mb * 16 + l15, mlast);
            const char* p = g.X + (long)row * (g.ldx * 2) + q * 16;
#pragma unroll
            for (int ks = 0; ks < KS1; ++ks) xa[mb][ks] = *(const bf16x8*)(p + ks * 64);
        }
    };
    auto init_y = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int by = 0; by < NBY; by += 2) {          // y starts from b2: lane's columns 32 (by >> 1) + 8 q + 4 (by & 1) + k
            const float* bp = b2s + 16 * by + 8 * q;
            const f32x4 v0 = *(const f32x4*)bp, v1 = *(const f32x4*)(bp + 4);
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) { yacc[mb][by] = v0; yacc[mb][by + 1] = v1; }
        }
    };
    // (a) u = xn W1^T + b1 for the tile's 64 hidden units
    auto fc1_tile = [&](unsigned slot, int n0) __attribute__((always_inline)) {
#pragma unroll
        for (int b = 0; b < 4; b += 2) {
            const float* bp = b1s + n0 + 16 * b + 8 * q;
            const f32x4 v0 = *(const f32x4*)bp, v1 = *(const f32x4*)(bp + 4);
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) { uacc[mb][b] = v0; uacc[mb][b + 1] = v1; }
        }
        const char* s0 = smem + slot + rb0;
        const char* s1 = smem + slot + rb1;
        bf16x8 f[2][4];
#pragma unroll
        for (int b = 0; b < 4; ++b) f[0][b] = *(const bf16x8*)(s0 + b * 2048);
#pragma unroll
        for (int ks = 0; ks < KS1; ++ks) {
            if (ks + 1 < KS1) {
                const char* sp = (((ks + 1) & 1) ? s1 : s0) + ((ks + 1) >> 1) * 8192;
#pragma unroll
                for (int b = 0; b < 4; ++b) f[(ks + 1) & 1][b] = *(const bf16x8*)(sp + b * 2048);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                for (int b = 0; b < 4; ++b)
                    uacc[mb][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f[ks & 1][b], xa[mb][ks], uacc[mb][b], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // GELU pair of the u tile: h, g stored (16 B per lane and k-step), h kept as the two k-step fragments of the second product
    auto gelu_tile = [&](int m0, int n0, auto full_tag) __attribute__((always_inline)) {
        constexpr bool FULL = decltype(full_tag)::value;
        const int rbase = m0 + wave * (16 * MB) + l15;
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
            const int row = rbase + mb * 16;
            const long roff = (long)(FULL ? row : min(row, mlast)) * g.ldh + n0 + 8 * q;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                u32x4f oh, og;
#pragma unroll
                for (int c = 0; c < 8; c += 2) {
                    const f32x2 xx = {uacc[mb][2 * kk + (c >> 2)][c & 3], uacc[mb][2 * kk + (c >> 2)][(c & 3) + 1]};
                    f32x2 yy, dd;
                    gelu_both2(xx, yy, dd);
                    oh[c / 2] = f_pack2(yy[0], yy[1]); og[c / 2] = f_pack2(dd[0], dd[1]);
                }
                hf[mb][kk] = __builtin_bit_cast(bf16x8, oh);
                if (FULL || row < g.M) {
                    *(u32x4f*)(g.Hh + roff + 32 * kk) = oh;
                    __builtin_nontemporal_store(og, (u32x4f*)(g.Hg + roff + 32 * kk));        // read again only in the backward pass (as gemm8.hip)
                }
            }
        }
    };
    // (b) y += h W2^T over the tile's 64 hidden units: two k-steps, increasing hidden order
    auto fc2_tile = [&](unsigned slot) __attribute__((always_inline)) {
        const char* s0 = smem + slot + rb0;
        const char* s1 = smem + slot + rb1;
        constexpr int PF = NW == 8 ? 2 : 4;         // fragments read ahead (registers: the eight-wave workgroup has 256 per wave)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const char* sp = kk ? s1 : s0;
            bf16x8 f[2][PF];
#pragma unroll
            for (int i = 0; i < PF; ++i) f[0][i] = *(const bf16x8*)(sp + i * 2048);
#pragma unroll
            for (int b0 = 0; b0 < NBY; b0 += PF) {
                const int cur = (b0 / PF) & 1;
                if (b0 + PF < NBY) {
#pragma unroll
                    for (int i = 0; i < PF; ++i) f[cur ^ 1][i] = *(const bf16x8*)(sp + (b0 + PF + i) * 2048);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < PF; ++i)
#pragma unroll
                    for (int mb = 0; mb < MB; ++mb)
                        yacc[mb][b0 + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f[cur][i], hf[mb][kk], yacc[mb][b0 + i], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    static_assert(NBY % 4 == 0, "the fragment look-ahead walks the output blocks in twos / fours");
    // y = resid + s acc (b2 sits in the accumulators), 16 B per lane and block pair. ALL the residual pieces and DropPath factors of the
    // wave's rows are requested first and the NEXT panel's xn fragments right behind them (the registers of u, h and the weight fragments
    // are free here), so a panel boundary costs one memory round trip - stamped before: three to four dependent round trips per row
    // block, 28k cycles per panel at C = 192, and the xn fragments another 14k in front of the next panel's first tile.
    auto y_epilogue = [&](int m0, int next_m0, auto full_tag) __attribute__((always_inline)) {
        constexpr bool FULL = decltype(full_tag)::value;
        const int rbase = m0 + wave * (16 * MB) + l15;
        u32x4f xr[MB][NBY / 2];
        float scv[MB];
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
            const int rowc = FULL ? rbase + mb * 16 : min(rbase + mb * 16, mlast);
            scv[mb] = 1.f;
            if constexpr (SCALED) scv[mb] = g.rowscale[f_div(rowc, g.rows_per_scale, g.inv_rps)];
            const bf16_t* rp = g.resid + (long)rowc * g.ldr + 8 * q;
#pragma unroll
            for (int i = 0; i < NBY / 2; ++i) xr[mb][i] = *(const u32x4f*)(rp + 32 * i);
        }
        if (next_m0 >= 0) load_x(next_m0);
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
            const int row = rbase + mb * 16;
            const int rowc = FULL ? row : min(row, mlast);
            const float scm = scv[mb];
            bf16_t* yp = g.Y + (long)rowc * g.ldy + 8 * q;
#pragma unroll
            for (int i = 0; i < NBY / 2; ++i) {
                u32x4f o;
#pragma unroll
                for (int c = 0; c < 8; c += 2) {
                    float xa_, xb_;
                    f_unpack2(xr[mb][i][c / 2], xa_, xb_);
                    float v0 = yacc[mb][2 * i + (c >> 2)][c & 3], v1 = yacc[mb][2 * i + (c >> 2)][(c & 3) + 1];
                    v0 = SCALED ? fmaf(v0, scm, xa_) : v0 + xa_; v1 = SCALED ? fmaf(v1, scm, xb_) : v1 + xb_;
                    o[c / 2] = f_pack2(v0, v1);
                }
                if (FULL || row < g.M) *(u32x4f*)(yp + 32 * i) = o;
            }
        }
    };

    __syncthreads();
    [[maybe_unused]] unsigned long long c_wait = 0, c_bar = 0, c_dma = 0, c_fc1 = 0, c_gelu = 0, c_fc2 = 0, c_panel = 0, c_yepi = 0, t_entry = 0, t0 = 0, t1 = 0, r_entry = 0;
    if constexpr (DBG) { t_entry = __builtin_amdgcn_s_memtime(); r_entry = __builtin_amdgcn_s_memrealtime(); }
#define F_STAMP(acc_) do { if constexpr (DBG) { t1 = __builtin_amdgcn_s_memtime(); acc_ += t1 - t0; t0 = t1; } } while (0)
    unsigned rd_slot = 0;
    auto next_slot = [&]() __attribute__((always_inline)) { rd_slot = rd_slot + SLOT_B == (unsigned)(NSLOT * SLOT_B) ? 0u : rd_slot + SLOT_B; };
    int hs_total = 0;
    bool prev_full = true;
    for (int pi = 0; pi < my_panels; ++pi) {
        const int m0 = ((int)blockIdx.x + pi * nwg) * (16 * NW * MB);
        const bool full = m0 + wave * (16 * MB) + 16 * MB <= g.M;
        if constexpr (DBG) t0 = __builtin_amdgcn_s_memtime();
        if (pi == 0) load_x(m0);                     // (later panels: requested at the end of the panel before, under its y epilogue)
        init_y();
        if constexpr (DBG) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); F_STAMP(c_panel); }
        for (int t = 0; t < g.nt; ++t) {
            const int n0 = t * 64;
#pragma unroll
            for (int half = 0; half < 2; ++half, ++hs_total) {
                if constexpr (DBG) t0 = __builtin_amdgcn_s_memtime();
                if (hs_total <= D) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                else if (full && prev_full) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VM_RELAX) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VM_STRICT) : "memory");
                F_STAMP(c_wait);
                F_BAR();
                F_STAMP(c_bar);
                dma_half();
                F_STAMP(c_dma);
                if (half == 0) {
                    fc1_tile(rd_slot, n0);
                    if constexpr (DBG) asm volatile("s_nop 0" ::: "memory");
                    F_STAMP(c_fc1);
                    if (full) gelu_tile(m0, n0, std::true_type{}); else gelu_tile(m0, n0, std::false_type{});
                    F_STAMP(c_gelu);
                } else {
                    fc2_tile(rd_slot);
                    if constexpr (DBG) asm volatile("s_nop 0" ::: "memory");
                    F_STAMP(c_fc2);
                }
                next_slot();
            }
        }
        if constexpr (DBG) t0 = __builtin_amdgcn_s_memtime();
        const int next_m0 = pi + 1 < my_panels ? ((int)blockIdx.x + (pi + 1) * nwg) * (16 * NW * MB) : -1;
        if (full) y_epilogue(m0, next_m0, std::true_type{}); else y_epilogue(m0, next_m0, std::false_type{});
        F_STAMP(c_yepi);
        prev_full = full;
    }
#undef F_STAMP
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if constexpr (DBG) {
        if (lane == 0) {
            unsigned long long* d = g.dbg + ((long)blockIdx.x * NW + wave) * 12;
            d[0] = c_panel; d[1] = c_wait; d[2] = c_bar; d[3] = c_dma; d[4] = c_fc1; d[5] = c_gelu; d[6] = c_fc2; d[7] = c_yepi;
            d[8] = __builtin_amdgcn_s_memtime() - t_entry; d[9] = __builtin_amdgcn_s_memrealtime() - r_entry; d[10] = r_entry; d[11] = (unsigned long long)hs_total;
        }
    }
}

unsigned long long* g_mlp8f_dbg = nullptr;
int g_mlp8f_force_mb = 0;

template <int KS1, int MB, bool SCALED, int NW = 4>
int launch_mlp8f(const M8Args& a, hipStream_t stream) {
    constexpr int NSLOT = KS1 == 12 ? 3 : 4;
    constexpr int C = 32 * KS1;
    constexpr int lds = NSLOT * C * 128 + 5 * C * 4;
    const int nwg = a.npanels < 256 ? a.npanels : 256;
    if (a.dbg) {
        (void)hipFuncSetAttribute((const void*)mlp8f_kernel<KS1, MB, NSLOT, SCALED, true, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        hipLaunchKernelGGL((mlp8f_kernel<KS1, MB, NSLOT, SCALED, true, NW>), dim3((unsigned)nwg), dim3(64 * NW), lds, stream, a);
    } else {
        static bool attr = false;
        if (!attr) { (void)hipFuncSetAttribute((const void*)mlp8f_kernel<KS1, MB, NSLOT, SCALED, false, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); attr = true; }
        hipLaunchKernelGGL((mlp8f_kernel<KS1, MB, NSLOT, SCALED, false, NW>), dim3((unsigned)nwg), dim3(64 * NW), lds, stream, a);
    }
    PSELD_LAUNCH_CHECK("mlp8f");
    return PSELD_OK;
}
}  // namespace

extern "C" void pseld_mlp_panel_set_debug_buffer(void* p) { g_mlp8f_dbg = (unsigned long long*)p; }
// mb: 16-row blocks per wave; + 8 selects the eight-wave workgroup (C = 192 only), e.g. 10 = eight waves x 32 rows
extern "C" void pseld_mlp_panel_force(int mb) { g_mlp8f_force_mb = mb; }

extern "C" int pseld_mlp_panel_fwd_supported(int dtype, int M, int C, int H) {
    return dtype == PSELD_BF16 && (C == 192 || C == 384) && H == 4 * C && M >= 1 && M < (1 << 24);
}

// y[M, C] = resid + s (gelu(xn W1^T + b1) W2^T + b2), h = gelu(.), g = gelu'(.) [M, H]; bf16 operands, fp32 biases / DropPath factors.
extern "C" int pseld_mlp_panel_fwd(int dtype, const void* xn, const void* w1, const float* b1, const void* w2, const float* b2, const void* resid,
                                   const float* rowscale, int rows_per_scale, void* y, void* h, void* g, int M, int C, int H, int ldx, int ldr,
                                   int ldy, int ldh, void* stream) {
    PSELD_CHECK_ARG(pseld_mlp_panel_fwd_supported(dtype, M, C, H), "mlp_panel_fwd: bf16, C = 192 | 384, H = 4 C (got dtype %d, C %d, H %d)", dtype, C, H);
    PSELD_CHECK_ARG(xn && w1 && w2 && resid && y && h && g, "mlp_panel_fwd: null operand");
    PSELD_CHECK_ARG(ldx % 8 == 0 && ldr % 8 == 0 && ldy % 8 == 0 && ldh % 8 == 0, "mlp_panel_fwd: leading dimensions must be multiples of 8");
    PSELD_CHECK_ARG((((unsigned long)xn | (unsigned long)w1 | (unsigned long)w2 | (unsigned long)resid | (unsigned long)y | (unsigned long)h | (unsigned long)g) & 15) == 0,
                    "mlp_panel_fwd: operands must be 16-byte aligned");
    M8Args a;
    a.X = (const char*)xn; a.W1 = (const char*)w1; a.W2 = (const char*)w2; a.b1 = b1; a.b2 = b2; a.resid = (const bf16_t*)resid; a.rowscale = rowscale;
    a.Y = (bf16_t*)y; a.Hh = (bf16_t*)h; a.Hg = (bf16_t*)g;
    a.M = M; a.C = C; a.H = H; a.ldx = ldx; a.ldw1 = C; a.ldw2 = H; a.ldr = ldr; a.ldy = ldy; a.ldh = ldh;
    a.rows_per_scale = rows_per_scale > 0 ? rows_per_scale : 1;
    a.inv_rps = 1.0f / (float)a.rows_per_scale;
    a.nt = H / 64;
    a.dbg = g_mlp8f_dbg;
    hipStream_t s = (hipStream_t)stream;
    const bool sc = rowscale != nullptr;
    int mb = g_mlp8f_force_mb;
    if (mb == 0) mb = C == 384 ? 2 : 10;         // (C = 192 at 4 waves x 4 blocks spills: 512 registers + scratch; eight waves x 32 rows measured fastest)
    const int nw = mb >= 8 ? 8 : 4;
    mb &= 7;
    a.npanels = pseld_cdiv(M, 16 * nw * mb);
    if (nw == 8) {
        PSELD_CHECK_ARG(C == 192 && mb == 2, "mlp_panel_fwd: the eight-wave workgroup is built for C = 192, 2 blocks per wave");
        return sc ? launch_mlp8f<6, 2, true, 8>(a, s) : launch_mlp8f<6, 2, false, 8>(a, s);
    }
    if (C == 384) {
        if (mb == 2) return sc ? launch_mlp8f<12, 2, true>(a, s) : launch_mlp8f<12, 2, false>(a, s);
        if (mb == 1) return sc ? launch_mlp8f<12, 1, true>(a, s) : launch_mlp8f<12, 1, false>(a, s);
    } else {
        if (mb == 3) return sc ? launch_mlp8f<6, 3, true>(a, s) : launch_mlp8f<6, 3, false>(a, s);
        if (mb == 2) return sc ? launch_mlp8f<6, 2, true>(a, s) : launch_mlp8f<6, 2, false>(a, s);
    }
    pseld_set_error("mlp_panel_fwd: no instantiation for C = %d, panel blocks %d", C, mb);
    return PSELD_ERR_UNSUPPORTED;
}
