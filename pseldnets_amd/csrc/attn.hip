// Swin (shifted-)window attention, forward and backward, for 8x8 windows with head_dim <= 32 on gfx950.
//
// Replaces (reference, /root/reference/src/models/components/htsat.py): window_partition/window_reverse :23-50,
// the cyclic torch.roll(+-shift) :239-242,:257-260, WindowAttention.forward :118-138 (q*scale, q@k^T,
// + relative_position_bias_table[relative_position_index], + attn_mask {0,-100}, softmax, @v, head merge) and
// the autograd of all of it. qkv / out stay in NATURAL token order [B, res*res, 3C] / [B, res*res, C]: roll,
// partition and reverse are folded into the window->token address map, the mask and the relative-position index
// are recomputed from coordinates, and the [B*nW, h, 64, 64] score tensor never exists in HBM.
//
// One workgroup (4 waves) = one window x one group of 4 heads, one head per wave. The window's q|k|v (|dO) rows
// are staged once into LDS with coalesced 16-byte loads; each wave then runs on MFMA 32x32 tiles:
//   S^T = K Q^T (lane = query, so the softmax reduction is in-register + one cross-half shuffle),
//   O   = P V with the S^T accumulators re-used directly as the A operand (no LDS round trip) and V read
//         through ds_read_b64_tr_b16.
// The forward also writes the per-(token, head) log-sum-exp of the scores. The backward uses it, flash-attention style:
// P = exp(S - lse) and dS = P (dP - delta) with delta = rowsum(dO o O) taken from the saved forward output, so every
// 32 x 32 score tile is independent (no row maximum / sum / delta passes, no 64-key register footprint): the kernel needs
// ~200 registers instead of > 256 and runs two waves per SIMD instead of one. Each tile's P and dS are dropped into LDS as
// [query][key] images and read back k-major (ds_read_b64_tr_b16) for the products that contract over queries (dV, dK);
// d(bias table) is summed in registers across all windows a workgroup walks and flushed with one fp32 atomic tile per head.
// T = bf16 (v_mfma_f32_32x32x16_bf16) or f32 (v_mfma_f32_32x32x2_f32, parity mode) share every loader.
//
// Roofline: HBM-bound (reads 3C, writes C per token; ~100 flop/B at head_dim 24).
#include "common.h"
#include "mma_frag.h"
#include <stdlib.h>
#include <type_traits>

static unsigned long long* g_attn_dbg = nullptr;

namespace {

// heads per workgroup (= waves per workgroup): 4 or 2 for bf16 (PSELD_ATTN_HG, default 4), 2 for f32 (LDS budget)
static int attn_hg(int dtype) {
    if (dtype != PSELD_BF16) return 2;
    return pseld_knob(KNOB_ATTN_HG, 4) == 2 ? 2 : 4;
}


struct AttnArgs {
    const void* qkv;   // [B, L, 3C]
    void* out;         // fwd: [B, L, C]
    const void* dout;  // bwd: [B, L, C]
    void* dqkv;        // bwd: [B, L, 3C]
    const float* bias_table;  // [225, heads]
    float* dbias_acc;         // bwd: [heads, 64(key), 64(query)] fp32, atomically accumulated
    float* lse;               // fwd (out, optional) / bwd (in): [B*L, heads] log-sum-exp of the masked, biased scores
    const void* osaved;       // bwd: the forward output [B, L, C] (delta = rowsum(dO o O))
    unsigned long long* dbg;  // diagnostic: s_memtime stamps of workgroup 0's first windows
    int B, res, C, heads, hd, shift;
    int n_win_total;          // B * (res/8)^2
    const void* wproj_t;      // attn_bwd24_kernel<true>: [C, C] = attn.proj.weight^T; dout is then d(x_mid), the gradient of the projection's OUTPUT
    const float* rowscale;    // ... and the DropPath factor per sample (or null)
    float scale;
};

// token index (natural order) of window slot t of window `w` of a res x res grid, plus its mask region label
__device__ __forceinline__ void window_token(const AttnArgs& a, int wi, int t, long& tok, int& label) {
    const int nwr = a.res >> 3;
    const int nW = nwr * nwr;
    const int b = wi / nW, w = wi - b * nW;
    const int wy = w / nwr, wx = w - wy * nwr;
    const int hs = wy * 8 + (t >> 3), ws = wx * 8 + (t & 7);
    int oh = hs + a.shift, ow = ws + a.shift;
    if (oh >= a.res) oh -= a.res;
    if (ow >= a.res) ow -= a.res;
    tok = (long)b * a.res * a.res + (long)oh * a.res + ow;
    if (a.shift > 0) {
        const int rh = hs < a.res - 8 ? 0 : (hs < a.res - a.shift ? 1 : 2);
        const int rw = ws < a.res - 8 ? 0 : (ws < a.res - a.shift ? 1 : 2);
        label = 3 * rh + rw;
    } else {
        label = 0;
    }
}

__device__ __forceinline__ int rel_index(int qi, int ki) {
    return ((qi >> 3) - (ki >> 3) + 7) * 15 + ((qi & 7) - (ki & 7) + 7);
}

// Cooperative copy of `nseg` column segments (each seg_elems wide, at global column gcol[s], LDS column lcol[s])
// for the 64 tokens of a window between global rows and the LDS tile.
template <typename T, bool TO_LDS>
__device__ __forceinline__ void window_copy(char* tile, int strideB, T* gbase, int gld, int gcol, int lcol,
                                            int seg_elems, const long* toks) {
    const int cps = seg_elems >> 3;  // 8-element chunks per token
    for (int c = threadIdx.x; c < 64 * cps; c += (int)blockDim.x) {
        const int t = c / cps, k = c - t * cps;
        T* g = gbase + toks[t] * gld + gcol + k * 8;
        char* l = tile + t * strideB + (lcol + k * 8) * (int)sizeof(T);
        if constexpr (TO_LDS) {
            if (sizeof(T) == 2) *(f32x4*)l = *(const f32x4*)g;
            else { ((f32x4*)l)[0] = ((const f32x4*)g)[0]; ((f32x4*)l)[1] = ((const f32x4*)g)[1]; }
        } else {
            if (sizeof(T) == 2) *(f32x4*)g = *(const f32x4*)l;
            else { ((f32x4*)g)[0] = ((const f32x4*)l)[0]; ((f32x4*)g)[1] = ((const f32x4*)l)[1]; }
        }
    }
}

typedef __attribute__((address_space(3))) void* lds_void_ptr_a;
typedef const __attribute__((address_space(1))) void* gbl_void_ptr_a;

// LDS row geometry of a window tile with NSEG column segments of seg_elems elements per token: rows are rowbytes + 16 (or
// + 32) bytes apart so that (stride / 16) is odd: 16 consecutive rows then start in 16 different 16-byte bank slots.
__device__ __forceinline__ int window_stride_bytes(int nseg, int seg_elems, int esize) {
    const int rb = nseg * seg_elems * esize + 16;
    return ((rb >> 4) & 1) ? rb : rb + 16;
}

// The 64 token rows of a window, NSEG segments each, straight from HBM into the LDS tile by LDS-DMA (no staging registers,
// every load of the window in flight at once: the chunk-by-chunk copy loop exposed one memory latency per chunk - 12.7k of a
// 31k-cycle window, tools/attn_stamps.py). One wave-instruction fills floor(64 / lanes_per_row) whole rows: lane -> (row slot,
// 16-byte chunk); the destination is lane-linear and a row slot is exactly one stride wide, so the row padding falls on idle
// lanes. seg_src(seg) returns the global base pointer and leading dimension of a segment, gcol its first column.
template <typename T, int NSEG, typename SrcFn>
__device__ __forceinline__ void window_dma_load(char* tile, int strideB, int seg_elems, int live_elems, const long* toks, int nwaves, SrcFn seg_src) {
    constexpr int EPC = 16 / (int)sizeof(T);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int nl = strideB >> 4;                          // lanes per row slot
    const int cps = seg_elems / EPC;                      // chunks per segment in the tile layout
    const int cpl = live_elems / EPC;                     // ... of which the first cpl exist in memory (last head group of an odd head count)
    if (nl <= 64) {
        const int rpi = 64 / nl;                          // rows per instruction
        const int slot = lane / nl, c = lane - slot * nl;
        const int seg = c / cps, k = c - seg * cps;
        const bool lane_ok = slot < rpi && seg < NSEG && k < cpl;
        for (int i = wave; i * rpi < 64; i += nwaves) {
            const int row = i * rpi + slot;
            if (lane_ok && row < 64) {
                const T* base; int ld, gcol;
                seg_src(seg, base, ld, gcol);
                const T* src = base + toks[row] * ld + gcol + k * EPC;
                __builtin_amdgcn_global_load_lds((gbl_void_ptr_a)src, (lds_void_ptr_a)(tile + i * rpi * strideB), 16, 0, 0);
            }
        }
    } else {                                              // rows longer than one instruction (1 KB): several instructions per row
        const int parts = (NSEG * cps + 63) >> 6;
        for (int j = wave; j < 64 * parts; j += nwaves) {
            const int row = j / parts, part = j - row * parts;
            const int c = part * 64 + lane;
            const int seg = c / cps, k = c - seg * cps;
            if (seg < NSEG && k < cpl) {
                const T* base; int ld, gcol;
                seg_src(seg, base, ld, gcol);
                const T* src = base + toks[row] * ld + gcol + k * EPC;
                __builtin_amdgcn_global_load_lds((gbl_void_ptr_a)src, (lds_void_ptr_a)(tile + row * strideB + part * 1024), 16, 0, 0);
            }
        }
    }
}

// S-type product: acc[ta][tb] = sum_dims A[row][dim] * B[col][dim] for two 64-row LDS operands at column
// offsets ca / cb (both "row-major, dims contiguous"). KS = number of 16-wide dim steps.
template <typename T>
__device__ __forceinline__ void qk_product(f32x16 (&acc)[2][2], const char* tile, int strideB, int ca, int cb, int hd,
                                           int lane) {
    using M = AMma<T>;
    const int r = lane & 31, h2 = lane >> 5;
#pragma unroll
    for (int ta = 0; ta < 2; ++ta)
#pragma unroll
        for (int tb = 0; tb < 2; ++tb)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[ta][tb][e] = 0.f;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        const int d0 = 16 * kk + 8 * h2;
        typename M::Frag fa[2], fb[2];
        // dims >= hd belong to the neighbouring head (or the row pad): read them anyway (in-bounds) and select
        // zero, which keeps this loop branch-free
        const bool live = d0 < hd;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            fa[t] = M::ld_row(tile + (t * 32 + r) * strideB + (ca + d0) * (int)sizeof(T));
            fb[t] = M::ld_row(tile + (t * 32 + r) * strideB + (cb + d0) * (int)sizeof(T));
            fa[t] = M::keep_if(fa[t], live);
            fb[t] = M::keep_if(fb[t], live);
        }
        if (16 * kk < hd) {
#pragma unroll
            for (int ta = 0; ta < 2; ++ta)
#pragma unroll
                for (int tb = 0; tb < 2; ++tb) M::mma(fa[ta], fb[tb], acc[ta][tb]);
        }
    }
}

// Z = X^T * Bm : X given as accumulator tiles x[tr][tc] (rows tr*32.., cols tc*32..), Bm an LDS matrix
// [64 rows][cols at col0..]; result z[tc] (rows = X's columns) [32 x 32(d)].
template <typename T>
__device__ __forceinline__ void xt_product(f32x16 (&z)[2], const f32x16 (&x)[2][2], const char* tile, int strideB,
                                           int col0, int lane) {
    using M = AMma<T>;
#pragma unroll
    for (int tc = 0; tc < 2; ++tc)
#pragma unroll
        for (int e = 0; e < 16; ++e) z[tc][e] = 0.f;
#pragma unroll
    for (int tr = 0; tr < 2; ++tr)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const typename M::Frag fb = M::ld_cols(tile, strideB, tr * 32 + 16 * s, col0, lane);
#pragma unroll
            for (int tc = 0; tc < 2; ++tc) M::mma(M::from_acc(x[tr][tc], s), fb, z[tc]);
        }
}

// store a [64 tokens][hd] result held as z[t] (row = token t*32 + acc_row, col = lane&31 = d) into LDS columns
template <typename T>
__device__ __forceinline__ void store_rows(char* tile, int strideB, int col0, const f32x16 (&z)[2], float mul, int hd,
                                           int lane) {
    const int d = lane & 31, h2 = lane >> 5;
    if (d < hd) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e)
                *(T*)(tile + (t * 32 + acc_row(e, h2)) * strideB + (col0 + d) * (int)sizeof(T)) = from_f32<T>(z[t][e] * mul);
    }
}

// acc[ta] (ta = 0,1) = sum_dims A[ta*32 + row][dim] * B[tb*32 + col][dim]: both 64-row LDS operands, one B tile.
// Result tiles: rows = A rows (registers), cols = B rows of tile tb (lanes).
template <typename T>
__device__ __forceinline__ void qk_half(f32x16 (&acc)[2], const char* tile, int strideB, int ca, int cb, int tb, int hd,
                                        int lane) {
    using M = AMma<T>;
    const int r = lane & 31, h2 = lane >> 5;
#pragma unroll
    for (int ta = 0; ta < 2; ++ta)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[ta][e] = 0.f;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        const int d0 = 16 * kk + 8 * h2;
        const bool live = d0 < hd;
        typename M::Frag fb = M::keep_if(M::ld_row(tile + (tb * 32 + r) * strideB + (cb + d0) * (int)sizeof(T)), live);
        if (16 * kk < hd) {
#pragma unroll
            for (int ta = 0; ta < 2; ++ta) {
                typename M::Frag fa = M::keep_if(M::ld_row(tile + (ta * 32 + r) * strideB + (ca + d0) * (int)sizeof(T)), live);
                M::mma(fa, fb, acc[ta]);
            }
        }
    }
}
// same with the roles named for the natural orientation: acc[qt] = Q[qt tile] . K[kt tile]^T
template <typename T>
__device__ __forceinline__ void qk_half_b(f32x16 (&acc)[2], const char* tile, int strideB, int ca, int cb, int tb, int hd,
                                          int lane) {
    qk_half<T>(acc, tile, strideB, ca, cb, tb, hd, lane);
}
// store one 32-token tile t of a result (row = token t*32 + acc_row, col = lane&31 = d) into LDS columns
template <typename T>
__device__ __forceinline__ void store_tile(char* tile, int strideB, int col0, int t, const f32x16& z, float mul, int hd,
                                           int lane) {
    const int d = lane & 31, h2 = lane >> 5;
    if (d < hd) {
#pragma unroll
        for (int e = 0; e < 16; ++e)
            *(T*)(tile + (t * 32 + acc_row(e, h2)) * strideB + (col0 + d) * (int)sizeof(T)) = from_f32<T>(z[e] * mul);
    }
}

// One head of one window (one wave): S^T = K Q^T, + bias (+ mask), softmax over the keys, O = P V into the head's (dead) q columns of the tile;
// the log-sum-exp per query goes to a.lse (optional). Shared by the one-window kernel and the persistent one: same arithmetic, same bits.
template <typename T>
__device__ __forceinline__ void attn_fwd_head(const AttnArgs& a, char* tile, int strideB, const float* btab, const int* labels, const long* toks,
                                              int GW, int wave, int head, int hd, int lane) {
    constexpr float LOG2E = 1.4426950408889634f;
    const int r = lane & 31, h2 = lane >> 5;
    const int cq = wave * hd, ck = GW + wave * hd, cv = 2 * GW + wave * hd;
    f32x16 st[2][2];  // S^T tiles: [key tile][query tile]
    qk_product<T>(st, tile, strideB, ck, cq, hd, lane);
    const float* bt = btab + wave * 225;
    float inv_l[2];
    const float scale2 = a.scale * LOG2E;
    const bool mixed = labels[64] != 0;
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        const int qi = qt * 32 + r;
        const float* btq = bt + (qi >> 3) * 15 + (qi & 7) + 112 - 4 * h2;
        float m = -1e30f;
        // key = kt*32 + (e&3) + 8*(e>>2) + 4*h2: its (y, x) = (kt*4 + (e>>2), (e&3) + 4*h2), so the table index is a
        // per-lane base minus a compile-time constant (one address register). Scores in log2 units: s = S scale log2e + b log2e.
        if (mixed) {
            const int ql = labels[qi];
            const int* labh = labels + 4 * h2;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    float s = fmaf(st[kt][qt][e], scale2, btq[-((kt * 4 + (e >> 2)) * 15 + (e & 3))]);
                    s -= (labh[kt * 32 + (e & 3) + 8 * (e >> 2)] != ql) ? 100.f * LOG2E : 0.f;
                    st[kt][qt][e] = s;
                    m = fmaxf(m, s);
                }
        } else {
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const float s = fmaf(st[kt][qt][e], scale2, btq[-((kt * 4 + (e >> 2)) * 15 + (e & 3))]);
                    st[kt][qt][e] = s;
                    m = fmaxf(m, s);
                }
        }
        m = fmaxf(m, __shfl_xor(m, 32, 64));
        float l = 0.f;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float p = __builtin_amdgcn_exp2f(st[kt][qt][e] - m);
                st[kt][qt][e] = p;
                l += p;
            }
        l += __shfl_xor(l, 32, 64);
        inv_l[qt] = 1.f / l;
        if (a.lse && h2 == 0) a.lse[toks[qi] * a.heads + head] = (m + __log2f(l)) * 0.6931471805599453f;   // natural-log units
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int e = 0; e < 16; ++e) st[kt][qt][e] *= inv_l[qt];
    }
    f32x16 o[2];
    xt_product<T>(o, st, tile, strideB, cv, lane);     // O[query][d] = sum_key P^T[key][query] V[key][d]
    store_rows<T>(tile, strideB, cq, o, 1.f, hd, lane);  // into this head's (dead) q columns
}

template <typename T, int HG>
__global__ __launch_bounds__(HG * 64, sizeof(T) == 2 ? 3 : 2) void attn_fwd_kernel(AttnArgs a) {
    constexpr int NTHR = HG * 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int hd = a.hd;
    const int GW = HG * hd;                              // columns per q/k/v segment in the tile
    const int strideB = window_stride_bytes(3, GW, (int)sizeof(T));
    char* tile = smem;
    float* btab = (float*)(smem + 64 * strideB);          // [HG][225]
    long* toks = (long*)(btab + HG * 225 + (HG & 1));     // [64] (8-byte aligned)
    int* labels = (int*)(toks + 64);                      // [64] + [1]: "this window mixes mask regions"

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // workgroup -> (window, head group): the head groups of a window read neighbouring 4 x head_dim x 2 byte pieces of the same q|k|v rows
    // (192 B at head_dim 24: they share cache lines), so they must meet in ONE L2 at the same time. A (window, group) grid ran the groups a
    // whole launch apart: FETCH_SIZE 146 MB for the 113 MB of a stage-2 qkv (8 line fetches per 6 lines of a row). 1-D grid: workgroup id ->
    // XCD id & 7, slot id >> 3; each XCD owns a contiguous range of (window, group) pairs, group fastest.
    const int ngrp = (a.heads + HG - 1) / HG;
    int hg, wi;
    {
        const int n = a.n_win_total * ngrp, id = blockIdx.x;
        const int Q = n >> 3, R = n & 7, xcd = id & 7;
        const int L = xcd * Q + min(xcd, R) + (id >> 3);
        wi = L / ngrp; hg = L - wi * ngrp;
    }
    const int head = hg * HG + wave;
    const int heads_here = min(HG, a.heads - hg * HG);
    const int r = lane & 31, h2 = lane >> 5;

    constexpr float LOG2E = 1.4426950408889634f;
    if (threadIdx.x < 64) {
        long tk; int lb;
        window_token(a, wi, threadIdx.x, tk, lb);
        toks[threadIdx.x] = tk; labels[threadIdx.x] = lb;
        // wave 0 holds all 64 labels: does the window straddle mask regions at all? (only the last row / column of windows of a
        // shifted block does; everywhere else the mask is identically zero and its 64 compare-selects per lane are skipped)
        const int l0 = __builtin_amdgcn_readfirstlane(lb);
        const unsigned long long diff = __ballot(lb != l0);
        if (threadIdx.x == 0) labels[64] = diff != 0ull;
    }
    // (measured: issuing the window's LDS-DMA first and gathering the table behind it is SLOWER - 158 -> 182 / 103 -> 110 / 50.7 -> 52.1 us at
    //  stages 0 / 1 / 2: the gather's waits then drain the DMA queue one load at a time)
    for (int i = threadIdx.x; i < heads_here * 225; i += NTHR) {
        const int hh = i / 225, idx = i - hh * 225;
        btab[hh * 225 + idx] = a.bias_table[idx * a.heads + hg * HG + hh] * LOG2E;     // scores are kept in log2 units (exp2 below)
    }
    __syncthreads();
    const T* qkv = (const T*)a.qkv;
    {
        const int C = a.C, col0 = hg * GW;
        window_dma_load<T, 3>(tile, strideB, GW, heads_here * hd, toks, HG, [&](int seg, const T*& base, int& ld, int& gcol) {
            base = qkv; ld = 3 * C; gcol = seg * C + col0;
        });
    }
    __syncthreads();   // (hipcc drains the DMA with vmcnt(0) in front of this barrier)

    if (head < a.heads) attn_fwd_head<T>(a, tile, strideB, btab, labels, toks, GW, wave, head, hd, lane);
    __syncthreads();
    window_copy<T, false>(tile, strideB, (T*)a.out, a.C, hg * GW, 0, heads_here * hd, toks);
}

// ---- the same forward, persistent and double-buffered (bf16, head_dim 24, heads % 4 == 0: every stage of HTS-AT) ----
// The one-window kernel above pays two global-memory latencies per workgroup in sequence (bias-table gather, then the window's rows) with
// nothing to compute meanwhile: 50.7 us for the 151 MB of a stage-2 launch (28 us at copy speed). Here a workgroup (4 waves = 4 heads, two
// workgroups per CU) keeps its head group's bias table for the whole launch and walks its windows with TWO tile buffers: the rows of window
// k + 2 are requested (LDS-DMA, inline asm: the compiler must not see them, it would drain them with vmcnt(0) at the next barrier) as soon
// as window k has been copied out, a whole window ahead of their use; waits are counted by hand (per wave and window: 16 row DMAs, 3 output
// stores, 2 log-sum-exp stores when lse is wanted). Raw s_barrier (__syncthreads would wait for vmcnt(0)). XCD x owns a contiguous range of
// windows; the ngrp workgroups that hold the head groups of one window run side by side on that XCD (they share the rows' cache lines).
constexpr int F24_STR = 3 * 96 * 2 + 16;                 // token row: q | k | v of 4 heads + pad (592 B: an odd number of 16-byte slots)
constexpr int F24_TILE = 64 * F24_STR;
constexpr size_t FWD24P_LDS = 2 * F24_TILE + 4 * 225 * 4 + 3 * 64 * 8 + 3 * 65 * 4;      // 81 692 B: two workgroups per CU

__device__ __forceinline__ void attn_dma16(unsigned lds_dst, const void* sbase, unsigned voff) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(lds_dst), "s"(sbase) : "memory");
}
#define AT_BAR()                                                  \
    do {                                                          \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        \
        __builtin_amdgcn_sched_barrier(0);                        \
        __builtin_amdgcn_s_barrier();                             \
        __builtin_amdgcn_sched_barrier(0);                        \
        asm volatile("" ::: "memory");                            \
    } while (0)

__global__ __launch_bounds__(256, 2) void attn_fwd24p_kernel(AttnArgs a) {
    constexpr int HG = 4, HD = 24, GW = HG * HD, STR = F24_STR;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* btab = (float*)(smem + 2 * F24_TILE);          // [4][225] bias x log2(e)
    long* toks = (long*)(btab + HG * 225);                // [3][64]  (window j lives in slot j % 3)
    int* labels = (int*)(toks + 3 * 64);                  // [3][65]
    constexpr float LOG2E = 1.4426950408889634f;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int ngrp = a.heads >> 2, wpx = 64 / ngrp;       // workgroups per head group on one XCD (grid = 512 = 8 XCDs x 64)
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int hg = slot % ngrp, lg = slot / ngrp;
    const int Q = a.n_win_total >> 3, R = a.n_win_total & 7;
    const int w0 = xcd * Q + min(xcd, R) + lg, nwx = Q + (xcd < R ? 1 : 0);
    const int nmine = lg < nwx ? (nwx - lg + wpx - 1) / wpx : 0;
    if (nmine == 0) return;
    for (int i = threadIdx.x; i < HG * 225; i += 256) {
        const int hh = i / 225, idx = i - hh * 225;
        btab[i] = a.bias_table[idx * a.heads + hg * HG + hh] * LOG2E;
    }
    const int head = hg * HG + wave;
    const unsigned lds_base = (unsigned)(unsigned long)(lds_void_ptr_a)smem;
    // one DMA instruction per token row: lane -> (segment q / k / v, 16-byte chunk) of the 36 live slots of a row
    const int seg = lane / 12, kch = lane - seg * 12;
    const unsigned voff = (unsigned)((seg * a.C + hg * GW + kch * 8) * 2);
    const long rowB = (long)3 * a.C * 2;
    auto set_tokens = [&](int j) {                        // wave 0: window j's token rows and mask labels into slot j % 3
        if (threadIdx.x < 64) {
            long tk; int lb;
            window_token(a, w0 + j * wpx, threadIdx.x, tk, lb);
            const int sl = j % 3;
            toks[sl * 64 + threadIdx.x] = tk; labels[sl * 65 + threadIdx.x] = lb;
            const int l0 = __builtin_amdgcn_readfirstlane(lb);
            const unsigned long long diff = __ballot(lb != l0);
            if (threadIdx.x == 0) labels[sl * 65 + 64] = diff != 0ull;
        }
    };
    auto issue = [&](int j) {                             // window j's rows -> tile buffer j & 1 (toks of slot j % 3 published by a barrier)
        // this wave's 16 rows are wave + 4 r: lane r reads its row's byte offset ONCE, the loop hands them out through v_readlane (a row
        // at a time through LDS + readfirstlane cost 200 cycles per DMA instruction: 3.2k of a 10k-cycle window)
        const long off = toks[(j % 3) * 64 + wave + 4 * (lane & 15)] * rowB;
        const unsigned lo = (unsigned)off, hi = (unsigned)((unsigned long)off >> 32);
        const unsigned dst = lds_base + (unsigned)((j & 1) * F24_TILE + wave * STR);
        if (lane < 36) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const unsigned long sb = ((unsigned long)(unsigned)__builtin_amdgcn_readlane((int)hi, r) << 32) | (unsigned)__builtin_amdgcn_readlane((int)lo, r);
                attn_dma16(dst + (unsigned)(4 * r * STR), (const char*)a.qkv + sb, voff);
            }
        }
    };
    set_tokens(0);
    if (nmine > 1) set_tokens(1);
    AT_BAR();
    issue(0);
    if (nmine > 1) issue(1);
    const bool want_lse = a.lse != nullptr;
    const bool stamp = a.dbg && threadIdx.x == 0 && blockIdx.x < 64;      // (diagnostic) [workgroup < 64][window < 8][8] s_memtime
    for (int kx = 0; kx < nmine; ++kx) {
        if (stamp && kx < 8) a.dbg[(blockIdx.x * 8 + kx) * 8 + 0] = __builtin_amdgcn_s_memtime();
        // window kx's rows have landed. Younger in this wave's queue: the stores of window kx - 1 (3 + 2 with lse), then the 16 DMAs of kx + 1
        if (kx + 1 >= nmine) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (kx == 0) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else if (want_lse) asm volatile("s_waitcnt vmcnt(21)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(19)" ::: "memory");
        AT_BAR();
        if (stamp && kx < 8) a.dbg[(blockIdx.x * 8 + kx) * 8 + 1] = __builtin_amdgcn_s_memtime();
        if (kx + 2 < nmine) set_tokens(kx + 2);           // slot (kx + 2) % 3 = (kx - 1) % 3: window kx - 1 is finished
        char* tile = smem + (kx & 1) * F24_TILE;
        const long* tk = toks + (kx % 3) * 64;
        attn_fwd_head<bf16_t>(a, tile, STR, btab, labels + (kx % 3) * 65, tk, GW, wave, head, HD, lane);
        if (stamp && kx < 8) a.dbg[(blockIdx.x * 8 + kx) * 8 + 2] = __builtin_amdgcn_s_memtime();
        AT_BAR();                                         // the four heads' outputs are in the tile (and the tokens of window kx + 2 in their slot)
        if (stamp && kx < 8) a.dbg[(blockIdx.x * 8 + kx) * 8 + 3] = __builtin_amdgcn_s_memtime();
        window_copy<bf16_t, false>(tile, STR, (bf16_t*)a.out, a.C, hg * GW, 0, GW, tk);
        if (stamp && kx < 8) a.dbg[(blockIdx.x * 8 + kx) * 8 + 4] = __builtin_amdgcn_s_memtime();
        AT_BAR();                                         // every wave has read its chunks of the tile: the buffer is free
        if (kx + 2 < nmine) issue(kx + 2);
        if (stamp && kx < 8) a.dbg[(blockIdx.x * 8 + kx) * 8 + 5] = __builtin_amdgcn_s_memtime();
    }
}

// one 32 x 32 tile: acc = A[ta*32 + row][dims] . B[tb*32 + col][dims]^T (rows in registers, cols = lanes)
template <typename T>
__device__ __forceinline__ void qk_tile(f32x16& acc, const char* tile, int strideB, int ca, int ta, int cb, int tb, int hd, int lane) {
    using M = AMma<T>;
    const int r = lane & 31, h2 = lane >> 5;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        const int d0 = 16 * kk + 8 * h2;
        const bool live = d0 < hd;          // dims >= hd belong to the neighbouring head (or the row pad): read, then zero
        if (16 * kk < hd) {
            const typename M::Frag fa = M::keep_if(M::ld_row(tile + (ta * 32 + r) * strideB + (ca + d0) * (int)sizeof(T)), live);
            const typename M::Frag fb = M::keep_if(M::ld_row(tile + (tb * 32 + r) * strideB + (cb + d0) * (int)sizeof(T)), live);
            M::mma(fa, fb, acc);
        }
    }
}

// row stride of the per-wave [32 queries][32 keys] P / dS images: 64 (mod 128) bytes for the transposed bf16 reads; the 8-byte
// pieces of a bf16 row are XOR-swizzled by (row >> 1) & 7 (AMma::img_piece / ld_img): written straight, 16 consecutive rows put their
// pieces in TWO bank pairs (8-way conflicts on every image write: 60 % of the LDS-active cycles of this kernel were conflict cycles)
template <typename T> struct ImgStride { static constexpr int value = sizeof(T) == 2 ? 64 : 144; };

// Backward: one workgroup = one window x HG heads, TWO waves per head: wave (head, kt) owns key tile kt (32 keys) of its head —
// dK / dV of those keys are complete in the wave (they sum over queries, which the wave walks), dQ is a partial sum over
// its keys and meets the partner's partial through LDS. Per wave: 2 score tiles [32 keys x 32 queries] per window.
template <typename T, int HG>
__global__ __launch_bounds__(HG * 128, sizeof(T) == 2 ? 2 : 1) void attn_bwd_kernel(AttnArgs a) {
    constexpr int NTHR = HG * 128;
    constexpr int IMG = ImgStride<T>::value;
    using M = AMma<T>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int hd = a.hd;
    const int GW = HG * hd;
    const int strideB = window_stride_bytes(5, GW, (int)sizeof(T));    // q | k | v | dO | O
    char* tile = smem;
    float* btab = (float*)(smem + 64 * strideB);          // [HG][225]
    float* lse_s = btab + HG * 225;                       // [64][HG]
    long* toks = (long*)(lse_s + 64 * HG + (HG & 1));     // (8-byte aligned)
    int* labels = (int*)(toks + 64);                      // [64] + [1]: "this window mixes mask regions"
    char* imgs = smem + (((char*)(labels + 65) - smem + 15) & ~15);   // [2*HG waves][4 KB]: P and dS images of the current tile; then the dQ partial

    const int lane0 = threadIdx.x & 63, lane = lane0, wave = threadIdx.x >> 6;
    const int hl = wave >> 1, kt = wave & 1;              // head inside the group, key tile of this wave
    const int hg = blockIdx.y;
    const int head = hg * HG + hl;
    const int heads_here = min(HG, a.heads - hg * HG);
    const int r = lane & 31, h2 = lane >> 5;
    const bool active = head < a.heads;

    for (int i = threadIdx.x; i < heads_here * 225; i += NTHR) {
        const int hh = i / 225, idx = i - hh * 225;
        btab[hh * 225 + idx] = a.bias_table[idx * a.heads + hg * HG + hh];
    }
    f32x16 dsum[2];  // sum over windows of dS^T [this key tile][query tile]
#pragma unroll
    for (int y = 0; y < 2; ++y)
#pragma unroll
        for (int e = 0; e < 16; ++e) dsum[y][e] = 0.f;

    const T* qkv = (const T*)a.qkv;
    const T* dout = (const T*)a.dout;
    const T* osv = (const T*)a.osaved;
    const int cq = hl * hd, ck = GW + hl * hd, cv = 2 * GW + hl * hd, cdo = 3 * GW + hl * hd, co = 4 * GW + hl * hd;
    const float* bt = btab + hl * 225;
    constexpr int IMGB = sizeof(T) == 2 ? 4096 : 2 * 32 * IMG;        // bytes per wave: two images, or 64 lanes x 16 floats of a dQ partial
    char* pimg = imgs + wave * IMGB;
    char* simg = pimg + 32 * IMG;
    float* dq_mine = (float*)pimg;                                    // published after the tile loop
    const float* dq_partner = (const float*)(imgs + (wave ^ 1) * IMGB);
    constexpr float LOG2E = 1.4426950408889634f;
    const float scale2 = a.scale * LOG2E;

    int it = 0;
    const bool stamp = a.dbg && threadIdx.x == 0 && blockIdx.y == 0 && blockIdx.x < 64;
    for (int wi = blockIdx.x; wi < a.n_win_total; wi += gridDim.x, ++it) {
        __syncthreads();  // previous iteration's stores out of the tile are done
        if (stamp && it < 8) a.dbg[(blockIdx.x * 8 + it) * 8 + 0] = __builtin_amdgcn_s_memtime();
        if (threadIdx.x < 64) {
            long tk; int lb;
            window_token(a, wi, threadIdx.x, tk, lb);
            toks[threadIdx.x] = tk; labels[threadIdx.x] = lb;
            // wave 0 holds all 64 labels: does the window straddle mask regions at all? (only the last row / column of
            // windows of a shifted block does; everywhere else the mask is identically zero)
            const int l0 = __builtin_amdgcn_readfirstlane(lb);
            const unsigned long long diff = __ballot(lb != l0);
            if (threadIdx.x == 0) labels[64] = diff != 0ull;
        }
        __syncthreads();
        {
            const int C = a.C, col0 = hg * GW;
            window_dma_load<T, 5>(tile, strideB, GW, heads_here * hd, toks, 2 * HG, [&](int seg, const T*& base, int& ld, int& gcol) {
                base = seg < 3 ? qkv : (seg == 3 ? dout : osv);
                ld = seg < 3 ? 3 * C : C;
                gcol = (seg < 3 ? seg * C : 0) + col0;
            });
            for (int i = threadIdx.x; i < 64 * heads_here; i += NTHR) {
                const int t = i / heads_here, h = i - t * heads_here;
                lse_s[t * HG + h] = a.lse[toks[t] * a.heads + hg * HG + h];
            }
        }
        __syncthreads();   // (hipcc drains the DMA with vmcnt(0) in front of this barrier)
        if (stamp && it < 8) a.dbg[(blockIdx.x * 8 + it) * 8 + 1] = __builtin_amdgcn_s_memtime();

        // (kt is made a compile-time constant by running one of two instances of the body: indexing the register arrays
        //  dq[] / dsum[] with the runtime wave parity costs a 31-deep select chain per element)
        f32x16 dq_keep;          // this wave's partial of the query tile it finishes after the exchange (qt == kt)
        auto compute = [&](auto KTc) {
            constexpr int kt = decltype(KTc)::value;
            f32x16 dq[2];
            const bool mixed = labels[64] != 0;
            // per-lane row constants of its two queries (qt = 0, 1): -lse * log2(e) and delta = sum_d dO[q][d] O[q][d]
            float nlse[2], delta[2];
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) {
                const int qi = qt * 32 + r;
                nlse[qt] = -lse_s[qi * HG + hl] * LOG2E;
                float dsum_q = 0.f;
                for (int c = h2; c * 8 < hd; c += 2) {          // the two lane halves split the 8-element chunks of the row
                    float od[8], dd[8];
                    load8<T>((const T*)(tile + qi * strideB + (co + c * 8) * (int)sizeof(T)), od);
                    load8<T>((const T*)(tile + qi * strideB + (cdo + c * 8) * (int)sizeof(T)), dd);
#pragma unroll
                    for (int j = 0; j < 8; ++j) dsum_q = fmaf(od[j], dd[j], dsum_q);
                }
                delta[qt] = dsum_q + __shfl_xor(dsum_q, 32, 64);
            }
            f32x16 dv, dk;
#pragma unroll
            for (int e = 0; e < 16; ++e) { dv[e] = 0.f; dk[e] = 0.f; }
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) {
#pragma unroll
                for (int e = 0; e < 16; ++e) dq[qt][e] = 0.f;
                // the lane id is laundered per tile: otherwise hipcc hoists every fragment address of both tiles out of the
                // persistent window loop and keeps ~60 address registers alive across it (spills at two waves per SIMD)
                int lane = lane0;
                asm volatile("" : "+v"(lane));
                const int r = lane & 31, h2 = lane >> 5;
                f32x16 pt, dpt;
                qk_tile<T>(pt, tile, strideB, ck, kt, cq, qt, hd, lane);      // S^T[kt, qt] = K_kt Q_qt^T   (rows = keys, lane = query)
                qk_tile<T>(dpt, tile, strideB, cv, kt, cdo, qt, hd, lane);    // dP^T[kt, qt] = V_kt dO_qt^T
                const int qi = qt * 32 + r;
                // key = kt*32 + (e&3) + 8*(e>>2) + 4*h2: (y, x) = (kt*4 + (e>>2), (e&3) + 4*h2) -> table index = per-lane base - constant
                const float* btq = bt + (qi >> 3) * 15 + (qi & 7) + 112 - 4 * h2 - kt * 60;
                if (mixed) {
                    const int ql = labels[qi];
                    const int* labh = labels + 4 * h2 + kt * 32;
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        float b = btq[-((e >> 2) * 15 + (e & 3))];
                        b -= (labh[(e & 3) + 8 * (e >> 2)] != ql) ? 100.f : 0.f;
                        pt[e] = __builtin_amdgcn_exp2f(fmaf(pt[e], scale2, fmaf(b, LOG2E, nlse[qt])));
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const float b = btq[-((e >> 2) * 15 + (e & 3))];
                        pt[e] = __builtin_amdgcn_exp2f(fmaf(pt[e], scale2, fmaf(b, LOG2E, nlse[qt])));
                    }
                }
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const float ds = pt[e] * (dpt[e] - delta[qt]);
                    dpt[e] = ds;                   // dS^T
                    dsum[qt][e] += ds;
                }
                // images: row = query r of this tile, 4 consecutive keys per register group
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const int off = M::img_piece(r, 2 * g4 + h2);     // keys 8 g4 + 4 h2 .. + 3 (bf16: swizzled 8-byte piece)
                    store4<T>(pimg + off, pt[4 * g4], pt[4 * g4 + 1], pt[4 * g4 + 2], pt[4 * g4 + 3]);
                    store4<T>(simg + off, dpt[4 * g4], dpt[4 * g4 + 1], dpt[4 * g4 + 2], dpt[4 * g4 + 3]);
                }
                // dQ[qt][d] (partial over this wave's keys) = sum_key dS^T[key][query] K[key][d]  (accumulators re-used as A operands)
#pragma unroll
                for (int sx = 0; sx < 2; ++sx)
                    M::mma(M::from_acc(dpt, sx), M::ld_cols(tile, strideB, kt * 32 + 16 * sx, ck, lane), dq[qt]);
                // dV[kt][d] += sum_{query in qt} P[query][key] dO[query][d];  dK[kt][d] += dS[query][key] Q[query][d]
#pragma unroll
                for (int sx = 0; sx < 2; ++sx) {
                    const typename M::Frag fdo = M::ld_cols_std(tile, strideB, qt * 32 + 16 * sx, cdo, lane);
                    const typename M::Frag fq = M::ld_cols_std(tile, strideB, qt * 32 + 16 * sx, cq, lane);
                    M::mma(M::ld_img(pimg, 16 * sx, lane), fdo, dv);
                    M::mma(M::ld_img(simg, 16 * sx, lane), fq, dk);
                }
                __builtin_amdgcn_sched_barrier(0);     // the next tile's image writes stay behind this tile's image reads
            }
            // nobody reads this head's k / v rows of key tile kt again (the partner wave owns the other 32 keys)
            store_tile<T>(tile, strideB, cv, kt, dv, 1.f, hd, lane);
            store_tile<T>(tile, strideB, ck, kt, dk, a.scale, hd, lane);
            // publish the dQ partial of the query tile the PARTNER finishes (qt = 1 - kt): [16 registers][64 lanes] floats
#pragma unroll
            for (int e = 0; e < 16; ++e) dq_mine[e * 64 + lane] = dq[1 - kt][e];
            dq_keep = dq[kt];
        };
        if (active) {
            if (kt == 0) compute(std::integral_constant<int, 0>{});
            else compute(std::integral_constant<int, 1>{});
        }
        if (stamp && it < 8) a.dbg[(blockIdx.x * 8 + it) * 8 + 2] = __builtin_amdgcn_s_memtime();
        __syncthreads();
        if (stamp && it < 8) a.dbg[(blockIdx.x * 8 + it) * 8 + 3] = __builtin_amdgcn_s_memtime();
        if (active) {
            f32x16 z = dq_keep;
#pragma unroll
            for (int e = 0; e < 16; ++e) z[e] += dq_partner[e * 64 + lane];
            store_tile<T>(tile, strideB, cq, kt, z, a.scale, hd, lane);     // every read of the q rows is behind the barrier above
        }
        __syncthreads();
        if (stamp && it < 8) a.dbg[(blockIdx.x * 8 + it) * 8 + 4] = __builtin_amdgcn_s_memtime();
        for (int sel = 0; sel < 3; ++sel)
            window_copy<T, false>(tile, strideB, (T*)a.dqkv, 3 * a.C, sel * a.C + hg * GW, sel * GW, heads_here * hd, toks);
        if (stamp && it < 8) a.dbg[(blockIdx.x * 8 + it) * 8 + 5] = __builtin_amdgcn_s_memtime();
    }
    // flush d(bias) partial sums: dbias_acc[head][key][query] += dsum (lane = query: 128-byte contiguous atomics)
    if (active) {
        float* dst = a.dbias_acc + (long)head * 4096;
#pragma unroll
        for (int qt = 0; qt < 2; ++qt)
#pragma unroll
            for (int e = 0; e < 16; ++e)
                atomicAdd(dst + (kt * 32 + acc_row(e, h2)) * 64 + qt * 32 + r, dsum[qt][e]);
    }
}

__device__ __forceinline__ void unpack8f(const f32x4& p, float (&v)[8]) {
    const bf16x8 x = __builtin_bit_cast(bf16x8, p);
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = (float)x[k];
}

// ---- Backward, bf16, head_dim 24 (every stage of HTS-AT), 4 heads per workgroup, ONE wave per head --------------------------------
// The generic kernel above splits a head over two waves (key tiles), drops every P / dS tile into LDS images to read it back
// transposed, exchanges dQ partials through LDS and stores its results with 2-byte LDS writes: 2.5x its HBM floor at every stage
// (tools/attn_bench.py: 506 / 240 / 138 / 79 us). Here a wave owns a whole head of a window (all 64 x 64 scores, tile by tile), so
//   * dQ^T[d][q] = K^T dS^T contracts over the ROWS of the dS^T accumulator: accumulator-as-operand, as before;
//   * dV^T[d][k] = dO^T P and dK^T[d][k] = Q^T dS contract over queries, the LANES of P^T / dS^T: the tiles are transposed BY THE
//     MATRIX PIPE (bf16(P^T) as the A operand times an identity B fragment: exact), which idles anyway - no images, no image traffic;
//   * row fragments are read through lane r -> LDS row swap23(r) and transposed column reads fetch column swap23(lane) (the 4-column
//     piece index of ds_read_b64_tr_b16 is bit-swapped): every accumulator operand then sees the natural k-order and a lane's
//     accumulator registers are 8 CONSECUTIVE head dims - dq / dk / dv go back into the tile as 16-byte pieces;
//   * d(bias table): dS^T is summed in registers over the workgroup's windows (by tile class: 48 values per lane) and flushed once per
//     workgroup with fp32 atomics onto the [heads][64][64] accumulator the callers reduce by relative position.
// PROJ (C = 96: the four heads are the whole row): the projection's input gradient is computed in the kernel - `dout` holds d(x_mid), the
// rows land in the dO segment by DMA like any operand, each wave turns its 24 columns into dO = (s * dY) Wproj in place (12 MFMAs against an
// LDS image of Wproj^T, between two barriers), and the saved output O (only needed for delta) comes straight from global memory into
// registers instead of through the tile: 4 segments, 75 KB per workgroup. = SURVEY 8b's pseld_swin_attn_bwd up to the qkv gradient.
template <bool PROJ>
__global__ __launch_bounds__(256, 2) void attn_bwd24_kernel(AttnArgs a) {
    using M = AMma<bf16_t>;
    constexpr int HG = 4, HD = 24, GW = HG * HD, NSEG = PROJ ? 4 : 5, STR = NSEG * GW * 2 + 16;   // token rows: q | k | v | dO (| O) of 4 heads + pad
    constexpr int WSTR = GW * 2 + 16;                                                               // Wproj^T image rows (PROJ)
    static_assert(((STR >> 4) & 1) == 1 && ((WSTR >> 4) & 1) == 1, "row strides must be an odd number of 16-byte slots");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* tile = smem;
    float* btab = (float*)(smem + 64 * STR);              // [4][225] bias x log2(e)
    float* lse_s = btab + HG * 225;                       // [64][4]
    long* toks = (long*)(lse_s + 64 * HG);
    int* labels = (int*)(toks + 64);                      // [64] + [1]
    char* wimg = smem + (((char*)(labels + 65) - smem + 15) & ~15);      // PROJ: [96 rows c'][96 n] of Wproj^T
    constexpr float LOG2E = 1.4426950408889634f;
    if (PROJ) {
        for (int c = threadIdx.x; c < GW * 12; c += 256) {
            const int row = c / 12, k = c - row * 12;
            *(f32x4*)(wimg + row * WSTR + k * 16) = *(const f32x4*)((const char*)a.wproj_t + row * (GW * 2) + k * 16);
        }
    }

    const int lane0 = threadIdx.x & 63, hl = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int hg = blockIdx.y;
    // diagnostic: entry / exit of EVERY workgroup on the 100 MHz clock all XCDs share (s_memtime counters differ between XCDs)
    const int wg_lin = blockIdx.y * gridDim.x + blockIdx.x;
    const bool life = a.dbg && threadIdx.x == 0 && wg_lin < 1024;
    if (life) a.dbg[4096 + 2 * wg_lin] = __builtin_amdgcn_s_memrealtime();
    for (int i = threadIdx.x; i < HG * 225; i += 256) {
        const int hh = i / 225, idx = i - hh * 225;
        btab[i] = a.bias_table[idx * a.heads + hg * HG + hh] * LOG2E;
    }
    // sum over this workgroup's windows of dS^T, by (query tile - key tile) + 1: the tiles (0, 0) and (1, 1) hold the same relative
    // positions lane by lane and register by register, and the callers only ever sum the accumulator by relative position
    f32x16 dsum[3];
#pragma unroll
    for (int y = 0; y < 3; ++y)
#pragma unroll
        for (int e = 0; e < 16; ++e) dsum[y][e] = 0.f;
    const bf16_t* qkv = (const bf16_t*)a.qkv;
    const bf16_t* dout = (const bf16_t*)a.dout;
    const bf16_t* osv = (const bf16_t*)a.osaved;
    const int cq = hl * HD, ck = GW + hl * HD, cv = 2 * GW + hl * HD, cdo = 3 * GW + hl * HD, co = 4 * GW + hl * HD;
    const float* bt = btab + hl * 225;
    const float scale2 = a.scale * LOG2E;

    int it = 0;
    const bool stamp = a.dbg && threadIdx.x == 0 && blockIdx.y == 0 && blockIdx.x < 64;
    for (int wi = blockIdx.x; wi < a.n_win_total; wi += gridDim.x, ++it) {
        __syncthreads();                                  // the previous window's rows have left the tile
        if (stamp && it < 7) a.dbg[(blockIdx.x * 8 + it) * 8 + 0] = __builtin_amdgcn_s_memtime();
        if (threadIdx.x < 64) {
            long tk; int lb;
            window_token(a, wi, threadIdx.x, tk, lb);
            toks[threadIdx.x] = tk; labels[threadIdx.x] = lb;
            const int l0 = __builtin_amdgcn_readfirstlane(lb);
            const unsigned long long diff = __ballot(lb != l0);
            if (threadIdx.x == 0) labels[64] = diff != 0ull;
        }
        __syncthreads();
        {
            // one DMA instruction per token row: lane -> (segment, 16-byte chunk) of the 61-slot row (60 live), as window_dma_load; the lane
            // constants are recomputed per window (laundered lane id) instead of living in registers across the loop
            int ln = lane0;
            asm volatile("" : "+v"(ln));
            const int seg = ln / 12, k = ln - seg * 12;
            if (ln < NSEG * 12) {
                const int C = a.C;
                const bf16_t* base = seg < 3 ? qkv : (seg == 3 ? dout : osv);
                const int ld = seg < 3 ? 3 * C : C;
                const int gcol = (seg < 3 ? seg * C : 0) + hg * GW + k * 8;
                for (int i = hl; i < 64; i += HG)
                    __builtin_amdgcn_global_load_lds((gbl_void_ptr_a)(base + toks[i] * ld + gcol), (lds_void_ptr_a)(tile + i * STR), 16, 0, 0);
            }
            int tx = threadIdx.x;
            asm volatile("" : "+v"(tx));
            lse_s[tx] = a.lse[toks[tx >> 2] * a.heads + hg * HG + (tx & 3)];
        }
        // PROJ: this lane's chunks of the saved output rows of its two queries (for delta) and the sample's DropPath factor, requested with
        // the DMA so that the same wait covers them
        f32x4 oreg[2][2];
        float sfac = 1.f;
        if (PROJ) {
            int ln = lane0;
            asm volatile("" : "+v"(ln));
            const int r = ln & 31, h2 = ln >> 5, pr = (r & ~12) | ((r & 4) << 1) | ((r & 8) >> 1);
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) {
                const bf16_t* orow = osv + toks[qt * 32 + pr] * a.C + hg * GW + hl * HD;
                oreg[qt][0] = *(const f32x4*)(orow + 8 * h2);
                oreg[qt][1] = *(const f32x4*)(orow + 16);
            }
            if (a.rowscale) { const int nw = a.res >> 3; sfac = a.rowscale[wi / (nw * nw)]; }
        }
        __syncthreads();                                  // (hipcc drains the DMA with vmcnt(0) in front of this barrier)
        if (stamp && it < 7) a.dbg[(blockIdx.x * 8 + it) * 8 + 1] = __builtin_amdgcn_s_memtime();
        if (PROJ) {
            // dO^T[c'][m] = Wproj^T[c'][:] . dY[m][:] for this head's 24 columns c' (rows through swap23) and both token tiles, then scaled by
            // the DropPath factor and written over the dY rows' own columns once every wave has read the rows
            int ln = lane0;
            asm volatile("" : "+v"(ln));
            const int r = ln & 31, h2 = ln >> 5, pr = (r & ~12) | ((r & 4) << 1) | ((r & 8) >> 1);
            const bool rowlive = pr < HD;
            const char* wr = wimg + (hl * HD + (rowlive ? pr : 0)) * WSTR;
            f32x16 y[2];
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
#pragma unroll
                for (int e = 0; e < 16; ++e) y[tt][e] = 0.f;
#pragma unroll
                for (int kk = 0; kk < 6; ++kk) {
                    const bf16x8 wf = M::keep_if(M::ld_row(wr + (16 * kk + 8 * h2) * 2), rowlive);
                    const bf16x8 df = M::ld_row(tile + (tt * 32 + r) * STR + (3 * GW + 16 * kk + 8 * h2) * 2);
                    M::mma(wf, df, y[tt]);
                }
            }
            __syncthreads();
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                bf16x8 f0, f1;          // rounded where the GEMM epilogue rounds: the product, then the scaled value
#pragma unroll
                for (int j = 0; j < 8; ++j) { f0[j] = (bf16_t)((float)(bf16_t)y[tt][j] * sfac); f1[j] = (bf16_t)((float)(bf16_t)y[tt][8 + j] * sfac); }
                char* p = tile + (tt * 32 + r) * STR + (3 * GW + hl * HD + 8 * h2) * 2;
                *(bf16x8*)p = f0;
                if (h2 == 0) *(bf16x8*)(p + 32) = f1;
            }
        }

        const bool mixed = labels[64] != 0;
        // identity fragments of the transposing products: element j of step s is 1 where 16 s + 8 h2 + j == lane & 31
        bf16x8 idf[2];
        float nlse[2], delta[2];
        {
            const int r = lane0 & 31, h2 = lane0 >> 5, pr = (r & ~12) | ((r & 4) << 1) | ((r & 8) >> 1);
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int j = 0; j < 8; ++j) idf[s][j] = (bf16_t)((16 * s + 8 * h2 + j == r) ? 1.f : 0.f);
            // per-lane constants of its two queries (lane r <-> query swap23(r) of the tile): -lse log2(e) and delta = sum_d dO O
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) {
                const int qi = qt * 32 + pr;
                nlse[qt] = -lse_s[qi * HG + hl] * LOG2E;
                float dsum_q = 0.f;
                for (int c = h2; c < 3; c += 2) {             // the lane halves split the three 8-dim chunks of the row
                    float od[8], dd[8];
                    if (PROJ) unpack8f(oreg[qt][c >> 1], od);
                    else load8<bf16_t>((const bf16_t*)(tile + qi * STR + (co + c * 8) * 2), od);
                    load8<bf16_t>((const bf16_t*)(tile + qi * STR + (cdo + c * 8) * 2), dd);
#pragma unroll
                    for (int j = 0; j < 8; ++j) dsum_q = fmaf(od[j], dd[j], dsum_q);
                }
                delta[qt] = dsum_q + __shfl_xor(dsum_q, 32, 64);
            }
        }
        // transposed column fragment: element j = tile[row0 + 8 h2 + j][col0 + swap23(lane & 31)] (lanes 4p..4p+3 of a 16-lane group fetch
        // the 4-column piece bitswap(p))
        auto ld_colp = [&](int row0, int col0, int lane) {
            const int i = lane & 15, q = i >> 2, p = i & 3, gsel = (lane >> 4) & 1, h2 = lane >> 5;
            const int pp = ((p & 1) << 1) | (p >> 1);
            const char* addr = tile + (row0 + 8 * h2 + q) * STR + (col0 + 16 * gsel + 4 * pp) * 2;
            const short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_ptr)(addr));
            const short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_ptr)(addr + 4 * STR));
            short8v sv;
            sv[0] = lo[0]; sv[1] = lo[1]; sv[2] = lo[2]; sv[3] = lo[3];
            sv[4] = hi[0]; sv[5] = hi[1]; sv[6] = hi[2]; sv[7] = hi[3];
            return __builtin_bit_cast(bf16x8, sv);
        };
        // a [32 tokens][24 dims] result (rows = dims through swap23: register e of lane half h2 is dim (e & 7) + 8 h2 + 16 (e >> 3); lane =
        // token row `row`) back into the tile: 16-byte pieces
        auto put = [&](const f32x16& z, float mul, int row, int col0, int h2) {
            bf16x8 f0, f1;
#pragma unroll
            for (int j = 0; j < 8; ++j) { f0[j] = (bf16_t)(z[j] * mul); f1[j] = (bf16_t)(z[8 + j] * mul); }
            char* p = tile + row * STR + (col0 + 8 * h2) * 2;
            *(bf16x8*)p = f0;
            if (h2 == 0) *(bf16x8*)(p + 32) = f1;
        };

        f32x16 dqT[2];
#pragma unroll
        for (int qt = 0; qt < 2; ++qt)
#pragma unroll
            for (int e = 0; e < 16; ++e) dqT[qt][e] = 0.f;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            f32x16 dvT, dkT;
#pragma unroll
            for (int e = 0; e < 16; ++e) { dvT[e] = 0.f; dkT[e] = 0.f; }
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) {
                // (the lane id is laundered per tile: otherwise hipcc hoists every fragment address out of the window loop)
                int lane = lane0;
                asm volatile("" : "+v"(lane));
                const int r = lane & 31, h2 = lane >> 5, pr = (r & ~12) | ((r & 4) << 1) | ((r & 8) >> 1);
                f32x16 pt, dpt;
#pragma unroll
                for (int e = 0; e < 16; ++e) { pt[e] = 0.f; dpt[e] = 0.f; }
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    // (the key-side fragments are re-read per tile rather than kept across the query tiles: 24 registers)
                    const bool live = 16 * kk + 8 * h2 < HD;
                    const bf16x8 kr = M::keep_if(M::ld_row(tile + (kt * 32 + pr) * STR + (ck + 16 * kk + 8 * h2) * 2), live);
                    // (one masked operand per product is enough: the other side's dead dims 24-31 are the NEXT head's or segment's finite bf16
                    //  values, times zero. q and v run into live segments; dO may run into the row's uninitialised pad, so it keeps the mask)
                    const bf16x8 vr = M::ld_row(tile + (kt * 32 + pr) * STR + (cv + 16 * kk + 8 * h2) * 2);
                    const bf16x8 qr = M::ld_row(tile + (qt * 32 + pr) * STR + (cq + 16 * kk + 8 * h2) * 2);
                    const bf16x8 dor = M::keep_if(M::ld_row(tile + (qt * 32 + pr) * STR + (cdo + 16 * kk + 8 * h2) * 2), live);
                    M::mma(kr, qr, pt);                   // S^T[key][query]: row i <-> key swap23(i), lane <-> query swap23(lane)
                    M::mma(vr, dor, dpt);                 // dP^T
                }
                // accumulator register e of lane half h2 is key kt*32 + (e & 7) + 8 h2 + 16 (e >> 3): (y, x) = (kt*4 + 2 (e>>3) + h2, e & 7)
                const int qi = qt * 32 + pr;
                const int lidx = (qi >> 3) * 15 + (qi & 7) + 112 - 15 * h2 - 97;        // table index = lidx + 97 - (kt*60 + 30 (e>>3) + (e&7))
                const float* btq = bt + lidx;
                const float nl = nlse[qt], dl = delta[qt];
                if (mixed) {
                    const int ql = labels[qi];
                    const int* labh = labels + kt * 32 + 8 * h2;
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        float b = btq[97 - (kt * 60 + 30 * (e >> 3) + (e & 7))] + nl;
                        b -= (labh[16 * (e >> 3) + (e & 7)] != ql) ? 100.f * LOG2E : 0.f;
                        pt[e] = __builtin_amdgcn_exp2f(fmaf(pt[e], scale2, b));
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 16; ++e)
                        pt[e] = __builtin_amdgcn_exp2f(fmaf(pt[e], scale2, btq[97 - (kt * 60 + 30 * (e >> 3) + (e & 7))] + nl));
                }
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const float ds = pt[e] * (dpt[e] - dl);
                    dpt[e] = ds;                           // dS^T
                    dsum[qt - kt + 1][e] += ds;
                }
                const bf16x8 pf0 = M::from_acc(pt, 0), pf1 = M::from_acc(pt, 1), sf0 = M::from_acc(dpt, 0), sf1 = M::from_acc(dpt, 1);
                // dQ^T[d][query] += K^T dS^T (contraction over the key rows of the accumulator)
                M::mma(ld_colp(kt * 32, ck, lane), sf0, dqT[qt]);
                M::mma(ld_colp(kt * 32 + 16, ck, lane), sf1, dqT[qt]);
                // transposes through the matrix pipe: rows = queries (lane order), lanes = keys (natural)
                f32x16 pq, sq;
#pragma unroll
                for (int e = 0; e < 16; ++e) { pq[e] = 0.f; sq[e] = 0.f; }
                M::mma(pf0, idf[0], pq); M::mma(pf1, idf[1], pq);
                M::mma(sf0, idf[0], sq); M::mma(sf1, idf[1], sq);
                // dV^T[d][key] += dO^T P, dK^T[d][key] += Q^T dS (contraction over the query rows)
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    M::mma(ld_colp(qt * 32 + 16 * s, cdo, lane), M::from_acc(pq, s), dvT);
                    M::mma(ld_colp(qt * 32 + 16 * s, cq, lane), M::from_acc(sq, s), dkT);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            // nobody reads this head's k / v rows of key tile kt again
            {
                int lane = lane0;
                asm volatile("" : "+v"(lane));
                put(dvT, 1.f, kt * 32 + (lane & 31), cv, lane >> 5);
                put(dkT, a.scale, kt * 32 + (lane & 31), ck, lane >> 5);
            }
        }
        {
            int lane = lane0;
            asm volatile("" : "+v"(lane));
            const int r = lane & 31, pr = (r & ~12) | ((r & 4) << 1) | ((r & 8) >> 1);
            put(dqT[0], a.scale, pr, cq, lane >> 5);
            put(dqT[1], a.scale, 32 + pr, cq, lane >> 5);
        }
        if (stamp && it < 7) a.dbg[(blockIdx.x * 8 + it) * 8 + 2] = __builtin_amdgcn_s_memtime();
        __syncthreads();
        if (stamp && it < 7) { a.dbg[(blockIdx.x * 8 + it) * 8 + 3] = __builtin_amdgcn_s_memtime(); a.dbg[(blockIdx.x * 8 + it) * 8 + 4] = a.dbg[(blockIdx.x * 8 + it) * 8 + 3]; }
        for (int sel = 0; sel < 3; ++sel)
            window_copy<bf16_t, false>(tile, STR, (bf16_t*)a.dqkv, 3 * a.C, sel * a.C + hg * GW, sel * GW, GW, toks);
        if (stamp && it < 7) a.dbg[(blockIdx.x * 8 + it) * 8 + 5] = __builtin_amdgcn_s_memtime();
    }
    // flush d(bias): dbias_acc[head][key][query] += dsum (row i <-> key swap23(i), lane <-> query swap23(lane)): 48 fire-and-forget atomics per
    // lane. Their drain (every workgroup of a head group hits the same 4 096 words: 10 us of the stage-2 launch, 22 us of 60 at stage 3 when
    // the launch runs alone, tools/attn_life.py) is memory-system time, not CU time: in the step the next kernels' workgroups take the CUs
    // meanwhile. Round 6 tried folding the 3 072 values of a head to its 225 table sums in the workgroup first (LDS layout + two passes,
    // 2.4 us of issue time per workgroup, 13x fewer atomics, spread over the pairs of each index): alone 60.1 -> 46.9 / 89.3 -> 85.0 /
    // 162.7 -> 165.7 us at stages 3 / 2 / 1, in the step 17.92 -> 17.97 ms and 5.58 -> 5.69 ms at 32 chunks: not kept (docs/EXPERIMENTS.md).
    {
        const int r = lane0 & 31, h2 = lane0 >> 5, pr = (r & ~12) | ((r & 4) << 1) | ((r & 8) >> 1);
        float* dst = a.dbias_acc + (long)(hg * HG + hl) * 4096;
#pragma unroll
        for (int y = 0; y < 3; ++y) {
            const int kt = y == 0 ? 1 : 0, qt = y == 2 ? 1 : 0;       // (1, 0), (0, 0) [+ (1, 1)], (0, 1)
#pragma unroll
            for (int e = 0; e < 16; ++e)
                atomicAdd(dst + (kt * 32 + 16 * (e >> 3) + 8 * h2 + (e & 7)) * 64 + qt * 32 + pr, dsum[y][e]);
        }
    }
    if (life) { __builtin_amdgcn_s_waitcnt(0); a.dbg[4096 + 2 * wg_lin + 1] = __builtin_amdgcn_s_memrealtime(); }
}
constexpr size_t BWD24_LDS = 64 * 976 + 4 * 225 * 4 + 64 * 4 * 4 + 64 * 8 + 65 * 4 + 12;
constexpr size_t BWD24P_LDS = 64 * 784 + 4 * 225 * 4 + 64 * 4 * 4 + 64 * 8 + 65 * 4 + 12 + 16 + 96 * 208;

__global__ void zero_f4_kernel(float4* __restrict__ p, long n4) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n4) p[i] = make_float4(0.f, 0.f, 0.f, 0.f);
}

// dtable[idx][h] (+)= sum over (q,k) with rel_index(q,k) == idx of acc[h][q][k]
__global__ void bias_table_grad_kernel(const float* __restrict__ acc, float* __restrict__ dtable, int heads, int accumulate) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 225 * heads) return;
    const int idx = i / heads, h = i - idx * heads;
    const int dy = idx / 15 - 7, dx = idx % 15 - 7;
    float s = 0.f;
    for (int qy = 0; qy < 8; ++qy) {
        const int ky = qy - dy;
        if (ky < 0 || ky > 7) continue;
        for (int qx = 0; qx < 8; ++qx) {
            const int kx = qx - dx;
            if (kx < 0 || kx > 7) continue;
            s += acc[(long)h * 4096 + (ky * 8 + kx) * 64 + qy * 8 + qx];
        }
    }
    dtable[i] = accumulate ? dtable[i] + s : s;
}

__global__ void bias_table_grad_batched_kernel(const float* __restrict__ acc_base, float* __restrict__ grad_base, const long* __restrict__ desc,
                                               int accumulate) {
    const long* d = desc + 3 * blockIdx.y;
    const float* acc = acc_base + d[0];
    float* dtable = grad_base + d[1];
    const int heads = (int)d[2];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 225 * heads) return;
    const int idx = i / heads, h = i - idx * heads;
    const int dy = idx / 15 - 7, dx = idx % 15 - 7;
    float s = 0.f;
    for (int qy = 0; qy < 8; ++qy) {
        const int ky = qy - dy;
        if (ky < 0 || ky > 7) continue;
        for (int qx = 0; qx < 8; ++qx) {
            const int kx = qx - dx;
            if (kx < 0 || kx > 7) continue;
            s += acc[(long)h * 4096 + (ky * 8 + kx) * 64 + qy * 8 + qx];
        }
    }
    dtable[i] = accumulate ? dtable[i] + s : s;
}

template <typename T> size_t fwd_lds(int hd, int HG) { return 64 * (3 * HG * hd * sizeof(T) + 32) + HG * 225 * 4 + 8 + 64 * 8 + 65 * 4; }
template <typename T> size_t bwd_lds(int hd, int HG) { return 64 * (5 * HG * hd * sizeof(T) + 32) + HG * 225 * 4 + 64 * HG * 4 + 8 + 64 * 8 + 65 * 4 + 16 + (size_t)2 * HG * (sizeof(T) == 2 ? 4096 : 2 * 32 * ImgStride<T>::value); }

int check_args(const char* who, int B, int res, int C, int heads, int shift) {
    PSELD_CHECK_ARG(B > 0 && res >= 8 && res % 8 == 0, "%s: grid side must be a multiple of 8 (got %d)", who, res);
    PSELD_CHECK_ARG(heads > 0 && C % heads == 0, "%s: C %% heads != 0", who);
    const int hd = C / heads;
    PSELD_CHECK_ARG(hd % 8 == 0 && hd >= 8 && hd <= 32, "%s: head_dim must be 8..32 and a multiple of 8 (got %d)", who, hd);
    PSELD_CHECK_ARG(shift == 0 || (shift == 4 && res > 8), "%s: shift must be 0 or 4 (and res > 8)", who);
    return PSELD_OK;
}

}  // namespace

extern "C" void pseld_attn_set_debug_buffer(void* p) { g_attn_dbg = (unsigned long long*)p; }

extern "C" int pseld_window_attn_fwd(int dtype, const void* qkv, const float* bias_table, void* out, float* lse, int B, int res,
                                     int C, int heads, int shift, void* stream) {
    PSELD_CHECK_ARG(qkv && bias_table && out, "window_attn_fwd: null pointer");
    int rc = check_args("window_attn_fwd", B, res, C, heads, shift);
    if (rc) return rc;
    AttnArgs a; memset(&a, 0, sizeof(a));
    a.qkv = qkv; a.out = out; a.lse = lse; a.bias_table = bias_table; a.B = B; a.res = res; a.C = C; a.heads = heads;
    a.hd = C / heads; a.shift = shift; a.n_win_total = B * (res / 8) * (res / 8);
    a.scale = 1.0f / sqrtf((float)a.hd);
    hipStream_t s = (hipStream_t)stream;
    const int hgv = attn_hg(dtype);
    dim3 grid((unsigned)(a.n_win_total * pseld_cdiv(heads, hgv)));
    // PSELD_ATTN_FWD_P=0: the one-window kernel for every shape (in-process A/B; both give the same bits)
    if (dtype == PSELD_BF16 && a.hd == 24 && heads % 4 == 0 && 64 % (heads / 4) == 0 && pseld_knob(KNOB_ATTN_FWD_P, 1) != 0) {
        static bool attr_p = false;
        if (!attr_p) { (void)hipFuncSetAttribute((const void*)attn_fwd24p_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)FWD24P_LDS); attr_p = true; }
        a.dbg = g_attn_dbg;
        hipLaunchKernelGGL(attn_fwd24p_kernel, dim3(512), dim3(256), FWD24P_LDS, s, a);
    } else if (dtype == PSELD_BF16) {
        if (hgv == 4) hipLaunchKernelGGL((attn_fwd_kernel<bf16_t, 4>), grid, dim3(256), fwd_lds<bf16_t>(a.hd, 4), s, a);
        else hipLaunchKernelGGL((attn_fwd_kernel<bf16_t, 2>), grid, dim3(128), fwd_lds<bf16_t>(a.hd, 2), s, a);
    } else if (dtype == PSELD_F32) {
        static bool attr_set = false;
        if (!attr_set) { (void)hipFuncSetAttribute((const void*)attn_fwd_kernel<float, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr_set = true; }
        hipLaunchKernelGGL((attn_fwd_kernel<float, 2>), grid, dim3(128), fwd_lds<float>(a.hd, 2), s, a);
    } else { pseld_set_error("window_attn_fwd: unknown dtype"); return PSELD_ERR_BAD_ARG; }
    PSELD_LAUNCH_CHECK("window_attn_fwd");
    return PSELD_OK;
}

extern "C" long pseld_window_attn_bwd_workspace(int heads) { return (long)heads * 4096 * (long)sizeof(float); }

// dqkv [B,L,3C] from dout [B,L,C], the saved forward output out [B,L,C] and lse [B*L, heads]; dbias_table f32[225, heads] (+)=.
// workspace: pseld_window_attn_bwd_workspace bytes.
extern "C" int pseld_window_attn_bwd(int dtype, const void* qkv, const float* bias_table, const void* out, const float* lse,
                                     const void* dout, void* dqkv, float* dbias_table, int B, int res, int C, int heads, int shift,
                                     int accumulate, float* workspace, long workspace_bytes, void* stream) {
    PSELD_CHECK_ARG(qkv && bias_table && out && lse && dout && dqkv && workspace, "window_attn_bwd: null pointer");
    int rc = check_args("window_attn_bwd", B, res, C, heads, shift);
    if (rc) return rc;
    PSELD_CHECK_ARG(workspace_bytes >= pseld_window_attn_bwd_workspace(heads), "window_attn_bwd: workspace too small");
    AttnArgs a; memset(&a, 0, sizeof(a));
    a.qkv = qkv; a.dout = dout; a.dqkv = dqkv; a.bias_table = bias_table; a.dbias_acc = workspace;
    a.osaved = out; a.lse = const_cast<float*>(lse); a.dbg = g_attn_dbg;
    a.B = B; a.res = res; a.C = C; a.heads = heads; a.hd = C / heads; a.shift = shift;
    a.n_win_total = B * (res / 8) * (res / 8);
    a.scale = 1.0f / sqrtf((float)a.hd);
    hipStream_t s = (hipStream_t)stream;
    // zeroed by a kernel, not hipMemsetAsync: inside a captured hipGraph (trainer.py use_graph) the memset NODE of this image's
    // runtime left stale sums in the accumulator after a few replays (tools/graph_debug.py: inf in d(bias_table) at the 4th replay)
    // dbias_table == NULL: deferred mode - `workspace` is this block's own accumulator, zeroed by the caller, and the caller turns
    // the accumulators of several blocks into table gradients with ONE pseld_bias_table_grad_batched launch
    if (dbias_table) {
        hipLaunchKernelGGL(zero_f4_kernel, dim3(heads * 4), dim3(256), 0, s, (float4*)workspace, (long)heads * 1024);
        PSELD_LAUNCH_CHECK("window_attn_bwd(zero)");
    }
    // Persistent workgroups, one resident round: two waves per head and two waves per SIMD (<= 256 registers), i.e. one
    // 4-head (8-wave) or two 2-head (4-wave) workgroups per CU; each walks several windows, so the d(bias) flush (one atomic
    // tile per wave per workgroup) stays a small fraction of the traffic. PSELD_ATTN_HG / PSELD_ATTN_BWD_WGS: experiment knobs (tools/attn_bench.py).
    // (bf16: always two heads per workgroup since the image swizzle shortened the compute phase - with four heads, one workgroup per
    //  CU, nothing covers its window loads: stage 0 ~1 % faster, same-box A/B 493 / 488 against 486 / 483 us)
    const int hgv = (dtype == PSELD_BF16 && !pseld_knob_is_set(KNOB_ATTN_HG)) ? 2 : attn_hg(dtype);
    const int nhg = pseld_cdiv(heads, hgv);
    const int es = pseld_knob(KNOB_ATTN_BWD_WGS, 0);               // experiment knob: total workgroups of the persistent loop
    int slots = (es ? es : (hgv == 4 ? 256 : 512)) / nhg;
    if (slots > a.n_win_total) slots = a.n_win_total;
    if (slots < 1) slots = 1;
    dim3 grid(slots, nhg);
    const int v2 = pseld_knob(KNOB_ATTN_BWD_V2, 1) != 0;
    if (dtype == PSELD_BF16 && a.hd == 24 && heads % 4 == 0 && v2) {
        // one wave per head, four heads per workgroup, two workgroups per CU, persistent
        int sl = (es ? es : 512) / (heads / 4);
        if (sl > a.n_win_total) sl = a.n_win_total;
        if (sl < 1) sl = 1;
        static bool attr24 = false;
        if (!attr24) { (void)hipFuncSetAttribute((const void*)attn_bwd24_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)BWD24_LDS); attr24 = true; }
        hipLaunchKernelGGL(attn_bwd24_kernel<false>, dim3(sl, heads / 4), dim3(256), BWD24_LDS, s, a);
    } else if (dtype == PSELD_BF16) {
        static bool attr_set = false;
        if (!attr_set) { (void)hipFuncSetAttribute((const void*)attn_bwd_kernel<bf16_t, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr_set = true; }
        if (hgv == 4) hipLaunchKernelGGL((attn_bwd_kernel<bf16_t, 4>), grid, dim3(512), bwd_lds<bf16_t>(a.hd, 4), s, a);
        else hipLaunchKernelGGL((attn_bwd_kernel<bf16_t, 2>), grid, dim3(256), bwd_lds<bf16_t>(a.hd, 2), s, a);
    } else if (dtype == PSELD_F32) {
        static bool attr_set = false;
        if (!attr_set) { (void)hipFuncSetAttribute((const void*)attn_bwd_kernel<float, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr_set = true; }
        hipLaunchKernelGGL((attn_bwd_kernel<float, 2>), grid, dim3(256), bwd_lds<float>(a.hd, 2), s, a);
    } else { pseld_set_error("window_attn_bwd: unknown dtype"); return PSELD_ERR_BAD_ARG; }
    PSELD_LAUNCH_CHECK("window_attn_bwd");
    if (dbias_table) {
        hipLaunchKernelGGL(bias_table_grad_kernel, dim3(pseld_cdiv(225 * heads, 256)), dim3(256), 0, s, workspace, dbias_table, heads, accumulate);
        PSELD_LAUNCH_CHECK("bias_table_grad");
    }
    return PSELD_OK;
}

// The attention half of a Swin block's backward up to the qkv gradient, for stage 0 of HTS-AT (bf16, C = 96, 4 heads): dy = d(x_mid) [B,L,C];
// d(attention output) = (s * dy) Wproj (reference: attn.proj, htsat.py:139, under the block's DropPath :258-260) is formed inside the
// kernel from wproj_t = Wproj^T [C, C] and rowscale f32[B] (or NULL), then the backward of window attention as pseld_window_attn_bwd.
// The projection's weight gradient (dy, out) and everything behind dqkv stay separate calls.
extern "C" int pseld_swin_block_attn_bwd_supported(int dtype, int res, int C, int heads) {
    return dtype == PSELD_BF16 && C == 96 && heads == 4 && res >= 8 && res % 8 == 0 ? 1 : 0;
}
extern "C" int pseld_swin_block_attn_bwd(int dtype, const void* qkv, const float* bias_table, const void* out, const float* lse, const void* dy,
                                         const void* wproj_t, const float* rowscale, void* dqkv, float* dbias_table, int B, int res, int C,
                                         int heads, int shift, int accumulate, float* workspace, long workspace_bytes, void* stream) {
    PSELD_CHECK_ARG(qkv && bias_table && out && lse && dy && wproj_t && dqkv && workspace, "swin_block_attn_bwd: null pointer");
    PSELD_CHECK_ARG(pseld_swin_block_attn_bwd_supported(dtype, res, C, heads), "swin_block_attn_bwd: built for bf16, C = 96, 4 heads (got dtype %d C %d heads %d)",
                    dtype, C, heads);
    int rc = check_args("swin_block_attn_bwd", B, res, C, heads, shift);
    if (rc) return rc;
    PSELD_CHECK_ARG(workspace_bytes >= pseld_window_attn_bwd_workspace(heads), "swin_block_attn_bwd: workspace too small");
    AttnArgs a; memset(&a, 0, sizeof(a));
    a.qkv = qkv; a.dout = dy; a.dqkv = dqkv; a.bias_table = bias_table; a.dbias_acc = workspace;
    a.osaved = out; a.lse = const_cast<float*>(lse); a.wproj_t = wproj_t; a.rowscale = rowscale;
    a.B = B; a.res = res; a.C = C; a.heads = heads; a.hd = C / heads; a.shift = shift;
    a.n_win_total = B * (res / 8) * (res / 8);
    a.scale = 1.0f / sqrtf((float)a.hd);
    hipStream_t s = (hipStream_t)stream;
    if (dbias_table) {
        hipLaunchKernelGGL(zero_f4_kernel, dim3(heads * 4), dim3(256), 0, s, (float4*)workspace, (long)heads * 1024);
        PSELD_LAUNCH_CHECK("swin_block_attn_bwd(zero)");
    }
    int sl = 512;
    if (sl > a.n_win_total) sl = a.n_win_total;
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute((const void*)attn_bwd24_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)BWD24P_LDS); attr = true; }
    hipLaunchKernelGGL(attn_bwd24_kernel<true>, dim3(sl, 1), dim3(256), BWD24P_LDS, s, a);
    PSELD_LAUNCH_CHECK("swin_block_attn_bwd");
    if (dbias_table) {
        hipLaunchKernelGGL(bias_table_grad_kernel, dim3(pseld_cdiv(225 * heads, 256)), dim3(256), 0, s, workspace, dbias_table, heads, accumulate);
        PSELD_LAUNCH_CHECK("bias_table_grad");
    }
    return PSELD_OK;
}

// The table gradients of n blocks in one launch: desc = n x {accumulator offset, table-gradient offset, heads} (device longs; offsets
// in floats from acc_base / grad_base). For callers of pseld_window_attn_bwd's deferred mode (dbias_table == NULL).
extern "C" int pseld_bias_table_grad_batched(const float* acc_base, float* grad_base, const long* desc, int n, int max_heads,
                                             int accumulate, void* stream) {
    PSELD_CHECK_ARG(acc_base && grad_base && desc && n > 0 && max_heads > 0, "bias_table_grad_batched: bad arguments");
    hipLaunchKernelGGL(bias_table_grad_batched_kernel, dim3(pseld_cdiv(225 * max_heads, 256), n), dim3(256), 0, (hipStream_t)stream,
                       acc_base, grad_base, desc, accumulate);
    PSELD_LAUNCH_CHECK("bias_table_grad_batched");
    return PSELD_OK;
}
