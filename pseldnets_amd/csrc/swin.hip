// Fused front half of the Swin attention branch for the HBM-bound stage of HTS-AT (C = 96, 4 heads of 24): LayerNorm ->
// QKV projection -> (shifted-)window attention in ONE persistent kernel, on gfx950.
//
// Replaces (reference, /root/reference/src/models/components/htsat.py): norm1 :234, WindowAttention.forward :118-138 up to the
// head merge (qkv Linear :123, q * scale, q @ k^T + relative_position_bias + mask, softmax, @ v), window_partition / reverse /
// roll :23-50,239-242,257-260 as an address map. The layer-wise path runs this as pseld_layernorm_fwd + pseld_gemm +
// pseld_window_attn_fwd: 9 row-tensors of M*C elements through HBM (LN r1 w1, GEMM r1 w3, attention r3) before the attention
// output is written; here x is read once and q|k|v, LN(x) (operands of the backward) and the attention output are written once.
//
// One workgroup = 4 waves = the 4 heads of one 8 x 8 window (64 tokens), persistent over windows; the next window's x rows arrive by
// LDS-DMA while the current one is computed. Per wave (head), everything between the normalised rows and the output stays in
// registers (MI355X guide, "an accumulator tile as the next MFMA's operand"):
//   Q^T[d][m] = Wq xh^T, K^T[d][m] = Wk xh^T   (rows = head dims, lanes = tokens; the weights' 24 (+8 zero) rows live in registers)
//   V[m][d]   = xh Wv^T                         (rows = tokens, lanes = head dims)
//   S^T[key][query] = (K^T)^T Q^T               A and B operands are both accumulators converted to bf16: no LDS round trip
//   softmax over keys with lane = query (in-register + one cross-half shuffle), scores in log2 units
//   O^T[d][query] = V^T P                        again accumulator (V) x accumulator (P^T)
// The W rows are fetched permuted (lane r holds row swap23(r)) so that a lane's accumulator registers are 8 CONSECUTIVE head dims:
// q, k and the output leave as 16-byte pieces. bf16 only (parity mode keeps the layer-wise kernels).
//
// Roofline: HBM (x in; q|k|v, LN(x), out written: 6 row-tensors = 906 MB per 786 432 tokens).
#include "mma_frag.h"
#include <stdlib.h>

namespace {

typedef __attribute__((address_space(3))) void* lds_void_ptr_s;

struct SwinArgs {
    const void* x;            // [B*res*res, C] block input
    const float* gamma;       // norm1
    const float* beta;
    const void* wqkv;         // [3C, C]
    const float* bqkv;        // [3C]
    const float* bias_table;  // [225, heads]
    void* qkv;                // out [M, 3C]
    void* ao;                 // out [M, C] attention output (heads merged)
    void* xh;                 // out [M, C] LN(x) (operand of the QKV weight gradient); may be null
    float* lse;               // out [M, heads] (may be null)
    // the block's attention tail (proj -> DropPath -> + shortcut), fused when wproj != null
    const void* wproj;        // [C, C] attn.proj.weight
    const float* bproj;       // [C]
    const float* rowscale;    // [B] DropPath factor per sample (mask / keep_prob), or null
    void* xmid;               // out [M, C] = x + s * (ao Wproj^T + bproj)
    int B, res, heads, shift, n_win_total;
    float scale, eps;
};

constexpr int SC = 96, SHD = 24, SNCH = 12, SRB = 192, XBUF = 64 * SRB;

__device__ __forceinline__ int sswz(int row) { return (row >> 2) & 3; }
__device__ __forceinline__ int schunk(int row, int ch) { return row * SRB + ((ch ^ sswz(row)) << 4); }
__device__ __forceinline__ int sswap23(int r) { return (r & ~12) | ((r & 4) << 1) | ((r & 8) >> 1); }

__device__ __forceinline__ void sdma16(char* lds_dst, const void* sbase, unsigned voff) {
    const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long)(lds_void_ptr_s)lds_dst);
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(dst), "s"(sbase) : "memory");
}

// token index (natural order) of slot t of window wi, and its mask region label (same map as attn.hip:window_token)
__device__ __forceinline__ void swin_token(const SwinArgs& a, int wi, int t, int& tok, int& label) {
    const int nwr = a.res >> 3, nW = nwr * nwr;
    const int b = wi / nW, w = wi - b * nW;
    const int wy = w / nwr, wx = w - wy * nwr;
    const int hs = wy * 8 + (t >> 3), ws = wx * 8 + (t & 7);
    int oh = hs + a.shift, ow = ws + a.shift;
    if (oh >= a.res) oh -= a.res;
    if (ow >= a.res) ow -= a.res;
    tok = (b * a.res + oh) * a.res + ow;
    if (a.shift > 0) {
        const int rh = hs < a.res - 8 ? 0 : (hs < a.res - a.shift ? 1 : 2);
        const int rw = ws < a.res - 8 ? 0 : (ws < a.res - a.shift ? 1 : 2);
        label = 3 * rh + rw;
    } else label = 0;
}

__device__ __forceinline__ bf16x8 pack8f(const f32x16& a, int e0) {
    bf16x8 f;
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = (bf16_t)a[e0 + j];
    return f;
}

// LDS: [W image: 288 qkv weight rows + 1 zero row + 96 proj weight rows, 192 B each, chunk-swizzled] [2 buffers x 2 windows x 64 raw token rows]
// [2 windows x 64 rows: LN(x), later the heads' attention output] [tables]
constexpr int WROWS = 3 * SC + 1 + SC, WIMG = WROWS * SRB, PROW0 = 3 * SC + 1;
constexpr int SW_LDS = WIMG + 6 * XBUF + (904 + 256 + 3 * SC) * 4 + (4 * 64 + 4 * 68) * 4;

__global__ __launch_bounds__(512, 2) void swin_attn_fwd_kernel(SwinArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* wimg = smem;                                   // rows: q dims of head 0..3, k dims, v dims (row = which * 96 + head * 24 + d); row 288 = zeros
    char* xbuf = smem + WIMG;                            // [2 buffers][2 windows][64 rows x 192 B], chunk-swizzled
    char* hbuf = xbuf + 4 * XBUF;                        // [2 windows][64 rows x 192 B]: LN(x) of the current pair; then the merged attention output
    float* btab = (float*)(hbuf + 2 * XBUF);             // [4][225] relative-position bias x log2(e), padded to 904
    float* bqk = btab + 904;                             // [2 (q, k)][4 heads][32] biases by head dim (0 for dims >= 24)
    float* gam = bqk + 256;                              // [96]
    float* bet = gam + SC;                               // [96]
    float* bpr = bet + SC;                               // [96] proj bias
    int* toks_all = (int*)(bpr + SC);                    // [2 buffers][2 windows][64] natural token index of the window's slots
    int* labels_all = toks_all + 256;                    // [2][2][68] mask-region labels + "this window mixes regions"

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int head = wave & 3, ws = wave >> 2;           // the workgroup's 8 waves: 2 windows x 4 heads
    const int wt = tid & 255;                            // thread index inside the window's four waves
    const int r = lane & 31, h = lane >> 5;
    constexpr float LOG2E = 1.4426950408889634f;
    const float scale2 = a.scale * LOG2E;

    for (int i = tid; i < 4 * 225; i += 512) {
        const int hh = i / 225, idx = i - hh * 225;
        btab[hh * 225 + idx] = a.bias_table[idx * a.heads + hh] * LOG2E;
    }
    if (tid < 256) {
        const int which = tid >> 7, hh = (tid >> 5) & 3, d = tid & 31;
        bqk[tid] = d < SHD ? a.bqkv[which * SC + hh * SHD + d] : 0.f;
    }
    const bool tail = a.wproj != nullptr;
    if (tid < SC) { gam[tid] = a.gamma[tid]; bet[tid] = a.beta[tid]; bpr[tid] = tail ? a.bproj[tid] : 0.f; }
    if (tid < SNCH) *(f32x4*)(wimg + schunk(3 * SC, tid)) = f32x4{0.f, 0.f, 0.f, 0.f};
    const int dl = sswap23(r);                           // this lane's head dim in every accumulator row / column map below
    const float bv = dl < SHD ? a.bqkv[2 * SC + head * SHD + dl] : 0.f;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the compiler-visible loads are retired before the first DMA
    // the 288 weight rows (55 KB) once per workgroup: every head's operand fragments are then plain 16-byte LDS reads
    for (int i = wave; i < 3 * SC * SNCH / 64; i += 8) {
        const int q = i * 64 + lane, row = q / SNCH, ch = q - row * SNCH;
        sdma16(wimg + i * 1024, a.wqkv, (unsigned)(row * SRB + ((ch ^ sswz(row)) << 4)));
    }
    if (tail) {                                            // the 96 proj weight rows behind the zero row (image rows PROW0 ..)
        for (int i = wave; i < SC * SNCH / 64; i += 8) {
            const int q = i * 64 + lane, row = q / SNCH, ch = q - row * SNCH;
            sdma16(wimg + PROW0 * SRB + i * 1024, a.wproj, (unsigned)(row * SRB + ((ch ^ sswz(PROW0 + row)) << 4)));
        }
    }
    const int wrow = dl < SHD ? head * SHD + dl : -1;     // weight row of this lane inside a (q | k | v) block; padding lanes read the zero row
    const int wq_row = wrow < 0 ? 3 * SC : wrow, wk_row = wrow < 0 ? 3 * SC : SC + wrow, wv_row = wrow < 0 ? 3 * SC : 2 * SC + wrow;

    const char* xs = (const char*)a.x;
    const int n_it = (a.n_win_total + 2 * (int)gridDim.x - 1) / (2 * (int)gridDim.x);
    auto window_of = [&](int it) { return (it * (int)gridDim.x + (int)blockIdx.x) * 2 + ws; };
    auto issue = [&](int it) {                              // the 64 token rows of this wave group's window of iteration `it` (12 DMA instructions, 3 per wave)
        const int wi = window_of(it);
        if (wi >= a.n_win_total) return;
        int ln = lane;
        asm volatile("" : "+v"(ln));
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int i = head + 4 * k, q = i * 64 + ln, t = q / SNCH, ch = q - t * SNCH;
            int tok, lb;
            swin_token(a, wi, t, tok, lb);
            sdma16(xbuf + ((it & 1) * 2 + ws) * XBUF + i * 1024, xs, (unsigned)(tok * SRB + ((ch ^ sswz(t)) << 4)));
        }
    };

    issue(0);
    for (int it = 0; it < n_it; ++it) {
        const int wi = window_of(it);
        const bool live = wi < a.n_win_total;               // (wave-group uniform: an odd window count leaves one group idle in the last iteration)
        char* xb = xbuf + ((it & 1) * 2 + ws) * XBUF;
        int* toks = toks_all + ((it & 1) * 2 + ws) * 64;
        int* labels = labels_all + ((it & 1) * 2 + ws) * 68;      // [64] labels + [64]: "this window mixes mask regions"
        if (wt < 64 && live) {
            int tk, lb;
            swin_token(a, wi, wt, tk, lb);
            toks[wt] = tk; labels[wt] = lb;
            const int l0 = __builtin_amdgcn_readfirstlane(lb);
            const unsigned long long diff = __ballot(lb != l0);
            if (wt == 0) labels[64] = diff != 0ull;
        }
        // DropPath factor of this window's sample by a SCALAR load (a vector-memory load here would make hipcc wait on vmcnt, i.e. on the
        // next pair's DMA, wherever it schedules the use); complete at the s_waitcnt below
        unsigned sc_bits = 0x3f800000u;
        if (tail && a.rowscale && live) {
            const float* sp = a.rowscale + __builtin_amdgcn_readfirstlane(wi / ((a.res >> 3) * (a.res >> 3)));
            asm volatile("s_load_dword %0, %1, 0x0" : "=s"(sc_bits) : "s"(sp) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // this wave's share of the window has landed (and its last stores)
        __builtin_amdgcn_s_barrier();                       // everybody's has; everybody has left the other buffer
        if (it + 1 < n_it) issue(it + 1);
        char* hb = hbuf + ws * XBUF;
        if (!live) {                                        // idle wave group: keep the barrier count of the iteration
            __builtin_amdgcn_s_barrier();
            if (tail) { __builtin_amdgcn_s_barrier(); __builtin_amdgcn_s_barrier(); }
            continue;
        }

        // ---- LayerNorm, raw rows -> hb (the raw rows stay: they are the shortcut of the fused tail): 4 threads per token row, 3 chunks (24 channels) each; same arithmetic as norm.hip:ln_fwd_kernel
        {
            const int row = wt >> 2, part = wt & 3;
            float v[24];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const bf16x8 f = *(const bf16x8*)(xb + schunk(row, part * 3 + c));
#pragma unroll
                for (int j = 0; j < 8; ++j) v[c * 8 + j] = (float)f[j];
            }
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < 24; ++j) s += v[j];
            s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64);
            const float mean = s / SC;
            float q = 0.f;
#pragma unroll
            for (int j = 0; j < 24; ++j) { const float d = v[j] - mean; q += d * d; }
            q += __shfl_xor(q, 1, 64); q += __shfl_xor(q, 2, 64);
            const float rstd = rsqrtf(q / SC + a.eps);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const int c0 = (part * 3 + c) * 8;
                bf16x8 f;
#pragma unroll
                for (int j = 0; j < 8; ++j) f[j] = (bf16_t)((v[c * 8 + j] - mean) * rstd * gam[c0 + j] + bet[c0 + j]);
                *(bf16x8*)(hb + schunk(row, part * 3 + c)) = f;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();

        // ---- LN(x) out (operand of the QKV weight gradient): 16-byte chunks straight from the image
        if (a.xh) {
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int q = wt + 256 * i, t = q / SNCH, chp = q - t * SNCH;      // LDS position (t, chp) holds logical chunk chp ^ swz(t)
                *(f32x4*)((char*)a.xh + (long)toks[t] * SRB + ((chp ^ sswz(t)) << 4)) = *(const f32x4*)(hb + q * 16);
            }
        }

        // ---- per 32-token tile: Q^T, K^T (rows = dims, lanes = tokens) and V (rows = tokens, lanes = dims), converted to operand
        // fragments at once (the fp32 accumulators of a tile die before the next tile starts: register pressure), and q | k | v out
        bf16x8 Qf[2][2], Kf[2][2], Vf[2][2];
        const float* bq = bqk + head * 32, * bk = bqk + 128 + head * 32;
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            f32x16 qT, kT, vA;
            {
                // accumulator register e of lane half h is head dim (e & 7) + 8 h + 16 (e >> 3) (rows fetched through swap23)
                const f32x4 q0 = *(const f32x4*)(bq + 8 * h), q1 = *(const f32x4*)(bq + 8 * h + 4), q2 = *(const f32x4*)(bq + 16 + 8 * h),
                            q3 = *(const f32x4*)(bq + 20 + 8 * h);
                const f32x4 k0 = *(const f32x4*)(bk + 8 * h), k1 = *(const f32x4*)(bk + 8 * h + 4), k2 = *(const f32x4*)(bk + 16 + 8 * h),
                            k3 = *(const f32x4*)(bk + 20 + 8 * h);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    qT[j] = q0[j]; qT[4 + j] = q1[j]; qT[8 + j] = q2[j]; qT[12 + j] = q3[j];
                    kT[j] = k0[j]; kT[4 + j] = k1[j]; kT[8 + j] = k2[j]; kT[12 + j] = k3[j];
                }
#pragma unroll
                for (int e = 0; e < 16; ++e) vA[e] = bv;
            }
#pragma unroll
            for (int kk = 0; kk < 6; ++kk) {
                const bf16x8 xf = *(const bf16x8*)(hb + schunk(32 * tt + r, 2 * kk + h));       // token 32 tt + r, channels 16 kk + 8 h ..
                const bf16x8 wqf = *(const bf16x8*)(wimg + schunk(wq_row, 2 * kk + h));         // weight row of head dim swap23(r)
                const bf16x8 wkf = *(const bf16x8*)(wimg + schunk(wk_row, 2 * kk + h));
                const bf16x8 wvf = *(const bf16x8*)(wimg + schunk(wv_row, 2 * kk + h));
                qT = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wqf, xf, qT, 0, 0, 0);
                kT = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wkf, xf, kT, 0, 0, 0);
                vA = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xf, wvf, vA, 0, 0, 0);
            }
            Qf[tt][0] = pack8f(qT, 0); Qf[tt][1] = pack8f(qT, 8);
            Kf[tt][0] = pack8f(kT, 0); Kf[tt][1] = pack8f(kT, 8);
            Vf[tt][0] = pack8f(vA, 0); Vf[tt][1] = pack8f(vA, 8);
            if (a.qkv) {
            // q, k: lane = token, 8 consecutive dims per register group
            char* orow = (char*)a.qkv + (long)toks[32 * tt + r] * (3 * SC * 2) + (head * SHD + 8 * h) * 2;
            *(bf16x8*)(orow) = Qf[tt][0];                                             // dims 8 h .. 8 h + 7
            *(bf16x8*)(orow + SC * 2) = Kf[tt][0];
            if (h == 0) {                                                              // dims 16 .. 23 (the upper half's second run is padding)
                *(bf16x8*)(orow + 32) = Qf[tt][1];
                *(bf16x8*)(orow + SC * 2 + 32) = Kf[tt][1];
            }
            // v: register e = token row acc_row(e, h), lane = head dim swap23(r)
            {
                char* vcol = (char*)a.qkv + (2 * SC + head * SHD + (dl < SHD ? dl : 0)) * 2;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const long off = (long)toks[32 * tt + acc_row(e, h)] * (3 * SC * 2);     // (64-bit: a saved qkv row is 3x an x row)
                    if (dl < SHD) *(bf16_t*)(vcol + off) = (bf16_t)vA[e];
                    if ((e & 3) == 3) __builtin_amdgcn_sched_barrier(0);      // four rows at a time: otherwise all sixteen addresses are kept live
                }
            }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (tail) {                                         // every head of the pair has read LN(x): hb becomes the attention-output tile
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }

        // ---- per 32-query tile: S^T[key tile] = K Q^T (both operands are accumulators: contraction over the head-dim rows), softmax over
        // the keys with lane = query (scores in log2 units; same arithmetic as attn.hip:attn_fwd_kernel), O^T = V^T P, out
        const float* bt = btab + head * 225;
        const bool mixed = labels[64] != 0;
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            f32x16 st[2];
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
                for (int e = 0; e < 16; ++e) st[kt][e] = 0.f;
#pragma unroll
                for (int s = 0; s < 2; ++s) st[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Kf[kt][s], Qf[qt][s], st[kt], 0, 0, 0);
            }
            const int qi = qt * 32 + r;
            const float* btq = bt + (qi >> 3) * 15 + (qi & 7) + 112 - 4 * h;
            float m = -1e30f;
            if (mixed) {
                const int ql = labels[qi];
                const int* labh = labels + 4 * h;
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        float sv = fmaf(st[kt][e], scale2, btq[-((kt * 4 + (e >> 2)) * 15 + (e & 3))]);
                        sv -= (labh[kt * 32 + (e & 3) + 8 * (e >> 2)] != ql) ? 100.f * LOG2E : 0.f;
                        st[kt][e] = sv;
                        m = fmaxf(m, sv);
                    }
            } else {
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const float sv = fmaf(st[kt][e], scale2, btq[-((kt * 4 + (e >> 2)) * 15 + (e & 3))]);
                        st[kt][e] = sv;
                        m = fmaxf(m, sv);
                    }
            }
            m = fmaxf(m, __shfl_xor(m, 32, 64));
            float l = 0.f;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const float p = __builtin_amdgcn_exp2f(st[kt][e] - m);
                    st[kt][e] = p;
                    l += p;
                }
            l += __shfl_xor(l, 32, 64);
            const float inv_l = 1.f / l;
            if (a.lse && h == 0) a.lse[(long)toks[qi] * a.heads + head] = (m + __log2f(l)) * 0.6931471805599453f;   // natural-log units
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int e = 0; e < 16; ++e) st[kt][e] *= inv_l;
            // O^T[dim][query] = V^T P: A = V accumulator fragments (rows = keys), B = P^T accumulator (rows = keys)
            f32x16 oT;
#pragma unroll
            for (int e = 0; e < 16; ++e) oT[e] = 0.f;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int s = 0; s < 2; ++s) oT = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Vf[kt][s], pack8f(st[kt], 8 * s), oT, 0, 0, 0);
            // row i of O^T is V's lane i = head dim swap23(i): register e of lane half h is dim (e & 7) + 8 h + 16 (e >> 3)
            const bf16x8 o0 = pack8f(oT, 0), o1 = pack8f(oT, 8);
            if (a.ao) {
                char* orow = (char*)a.ao + (long)toks[qi] * (SC * 2) + (head * SHD + 8 * h) * 2;
                *(bf16x8*)(orow) = o0;
                if (h == 0) *(bf16x8*)(orow + 32) = o1;
            }
            if (tail) {                                     // the same pieces into the tile: chunk = 3 head + h (and + 2 for dims 16..23)
                *(bf16x8*)(hb + schunk(qi, 3 * head + h)) = o0;
                if (h == 0) *(bf16x8*)(hb + schunk(qi, 3 * head + 2)) = o1;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (tail) {
            // ---- x_mid^T[c][m] = Wproj ao^T: this wave's 24 output channels (rows through swap23, as everywhere), both token tiles; then
            // + bias, x DropPath factor of the sample, + the raw x rows still sitting in the DMA buffer: 16-byte pieces out
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            const int prow = wrow < 0 ? 3 * SC : PROW0 + wrow;
            const float sc = __builtin_bit_cast(float, sc_bits);
            const f32x4 b0 = *(const f32x4*)(bpr + head * SHD + 8 * h), b1 = *(const f32x4*)(bpr + head * SHD + 8 * h + 4);
            const f32x4 b2 = *(const f32x4*)(bpr + head * SHD + (h ? 0 : 16)), b3 = *(const f32x4*)(bpr + head * SHD + (h ? 0 : 20));
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                f32x16 y;
#pragma unroll
                for (int j = 0; j < 4; ++j) { y[j] = b0[j]; y[4 + j] = b1[j]; y[8 + j] = b2[j]; y[12 + j] = b3[j]; }
#pragma unroll
                for (int kk = 0; kk < 6; ++kk) {
                    const bf16x8 wf = *(const bf16x8*)(wimg + schunk(prow, 2 * kk + h));
                    const bf16x8 af = *(const bf16x8*)(hb + schunk(32 * tt + r, 2 * kk + h));
                    y = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, af, y, 0, 0, 0);
                }
                const int m = 32 * tt + r;
                const bf16x8 x0 = *(const bf16x8*)(xb + schunk(m, 3 * head + h));
                const bf16x8 x1 = *(const bf16x8*)(xb + schunk(m, 3 * head + 2));
                bf16x8 r0, r1;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    r0[j] = (bf16_t)fmaf((float)(bf16_t)y[j], sc, (float)x0[j]);
                    r1[j] = (bf16_t)fmaf((float)(bf16_t)y[8 + j], sc, (float)x1[j]);
                }
                char* orow = (char*)a.xmid + (long)toks[m] * (SC * 2) + (head * SHD + 8 * h) * 2;
                *(bf16x8*)(orow) = r0;
                if (h == 0) *(bf16x8*)(orow + 32) = r1;
            }
        }
    }
}

}  // namespace

extern "C" int pseld_swin_attn_supported(int dtype, int res, int C, int heads) {
    return dtype == PSELD_BF16 && C == SC && heads == 4 && res >= 8 && res % 8 == 0 ? 1 : 0;
}

namespace {
int swin_launch(SwinArgs& a, int dtype, int B, int res, int C, int heads, int shift, float eps, void* stream, const char* who) {
    PSELD_CHECK_ARG(pseld_swin_attn_supported(dtype, res, C, heads), "%s: built for bf16, C = 96, 4 heads (got dtype %d C %d heads %d res %d)",
                    who, dtype, C, heads, res);
    PSELD_CHECK_ARG(B > 0 && shift >= 0 && shift < 8 && (long)B * res * res * (3 * SC * 2) < (1L << 32), "%s: bad B / shift, or more than 4 GB of saved q|k|v rows", who);
    a.B = B; a.res = res; a.heads = heads; a.shift = shift; a.n_win_total = B * (res / 8) * (res / 8);
    a.scale = 1.0f / sqrtf((float)SHD); a.eps = eps;
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute((const void*)swin_attn_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, SW_LDS); attr = true; }
    const int pairs = (a.n_win_total + 1) / 2;
    const int grid = pairs < 256 ? pairs : 256;                  // persistent: one 8-wave workgroup per CU, two windows per iteration
    hipLaunchKernelGGL(swin_attn_fwd_kernel, dim3(grid), dim3(512), SW_LDS, (hipStream_t)stream, a);
    PSELD_LAUNCH_CHECK(who);
    return PSELD_OK;
}
}  // namespace

// LayerNorm(norm1) -> qkv Linear -> window attention. Outputs: qkv [M, 3C], out [M, C] (heads merged, before proj), xh = LN(x)
// [M, C] (optional), lse f32 [M, heads] (optional): exactly what pseld_layernorm_fwd + pseld_gemm + pseld_window_attn_fwd leave.
extern "C" int pseld_swin_attn_fwd(int dtype, const void* x, const float* gamma, const float* beta, const void* wqkv, const float* bqkv,
                                   const float* bias_table, void* qkv, void* out, void* xh, float* lse, int B, int res, int C, int heads,
                                   int shift, float eps, void* stream) {
    PSELD_CHECK_ARG(x && gamma && beta && wqkv && bqkv && bias_table && qkv && out, "swin_attn_fwd: null pointer");
    SwinArgs a;
    memset(&a, 0, sizeof(a));
    a.x = x; a.gamma = gamma; a.beta = beta; a.wqkv = wqkv; a.bqkv = bqkv; a.bias_table = bias_table;
    a.qkv = qkv; a.ao = out; a.xh = xh; a.lse = lse;
    return swin_launch(a, dtype, B, res, C, heads, shift, eps, stream, "swin_attn_fwd");
}

// The whole attention half of a Swin block (htsat.py:234-260): x_mid = x + s * (proj(window_attention(qkv(norm1(x)))) + b), s = the DropPath
// factor of the token's sample (rowscale f32[B], or NULL). qkv / out / xh / lse are the operands of the backward: all four may be NULL in
// a no-grad forward, which then reads x and writes x_mid and nothing else.
extern "C" int pseld_swin_block_attn_fwd(int dtype, const void* x, const float* gamma, const float* beta, const void* wqkv, const float* bqkv,
                                         const float* bias_table, const void* wproj, const float* bproj, const float* rowscale, void* qkv,
                                         void* out, void* xh, float* lse, void* xmid, int B, int res, int C, int heads, int shift, float eps,
                                         void* stream) {
    PSELD_CHECK_ARG(x && gamma && beta && wqkv && bqkv && bias_table && wproj && bproj && xmid, "swin_block_attn_fwd: null pointer");
    SwinArgs a;
    memset(&a, 0, sizeof(a));
    a.x = x; a.gamma = gamma; a.beta = beta; a.wqkv = wqkv; a.bqkv = bqkv; a.bias_table = bias_table;
    a.wproj = wproj; a.bproj = bproj; a.rowscale = rowscale; a.xmid = xmid;
    a.qkv = qkv; a.ao = out; a.xh = xh; a.lse = lse;
    return swin_launch(a, dtype, B, res, C, heads, shift, eps, stream, "swin_block_attn_fwd");
}
