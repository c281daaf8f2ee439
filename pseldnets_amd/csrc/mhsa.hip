// Global multi-head self-attention of the PaSST blocks, forward and backward, head_dim 64, any sequence length, gfx950.
//
// Replaces (reference, /root/reference/src/models/components/passt.py): Attention.forward :62-82 — the qkv
// reshape/permute, q @ k^T * scale, softmax, attn @ v, head merge — and its autograd. qkv [B, N, 3E] and out
// [B, N, E] stay in the GEMMs' natural row-major layout; the [B, heads, N, N] score tensor never exists in HBM
// (flash-style streaming over 64-key tiles with a running max / sum), backward recomputes the probabilities from
// the saved log-sum-exp.
//
// Orientation trick shared with attn.hip: scores are computed TRANSPOSED, S^T[key][query] = K Q^T, so a lane owns one
// query: running max / sum / rescale are lane-local (+ one cross-half shuffle), and the P^T accumulators are re-used
// directly as the B operand of O^T[d][query] += V^T P^T with V^T read through ds_read_b64_tr_b16. dK/dV use the
// natural orientation (lane = key) in a second kernel so that no tile is ever transposed and no atomics are needed.
//
// Roofline: MFMA-leaning (4*N*64 flop per query-key pair set vs 4*64 elements of traffic per token); at N = 602 the
// K/V tiles of a (batch, head) are re-read ceil(N/128) = 5 times from L2.
#include "common.h"
#include "mma_frag.h"

namespace {

constexpr int HD = 64;

struct MhsaArgs {
    const void* qkv;   // [B, N, 3E]
    void* out;         // fwd: [B, N, E]
    const void* dout;  // bwd: [B, N, E]
    void* dqkv;        // bwd: [B, N, 3E]
    float* lse;        // [B, heads, N] log-sum-exp of the scaled scores (written by fwd, read by bwd)
    float* delta;      // [B, heads, N] sum_d dO*O
    int B, N, E, heads;
    float scale;
};

template <typename T> struct RowS { static constexpr int value = HD * (int)sizeof(T) + 16; };  // 144 B / 272 B rows

template <typename T>
__device__ __forceinline__ void copy16(char* l, const T* g) {
    if constexpr (sizeof(T) == 2) *(f32x4*)l = *(const f32x4*)g;
    else { ((f32x4*)l)[0] = ((const f32x4*)g)[0]; ((f32x4*)l)[1] = ((const f32x4*)g)[1]; }
}
template <typename T>
__device__ __forceinline__ void zero16(char* l) {
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    ((f32x4*)l)[0] = z;
    if constexpr (sizeof(T) == 4) ((f32x4*)l)[1] = z;
}

// rows [r0, r0+nrows) x 64 columns starting at column gcol of a [N, ld] matrix -> LDS tile (zero rows past N)
template <typename T>
__device__ __forceinline__ void load_rows(char* tile, const T* g, long ld, int gcol, int r0, int nrows, int N) {
    constexpr int SB = RowS<T>::value;
    for (int c = threadIdx.x; c < nrows * 8; c += (int)blockDim.x) {
        const int t = c >> 3, k = c & 7;
        char* l = tile + t * SB + k * 8 * (int)sizeof(T);
        if (r0 + t < N) copy16<T>(l, g + (long)(r0 + t) * ld + gcol + k * 8);
        else zero16<T>(l);
    }
}
// The streamed 64-row tiles are requested one iteration ahead into registers (2 chunks per thread and tile with 256
// threads) and dropped into LDS after the barrier: their latency hides behind the MFMA work of the current tile instead
// of sitting between two barriers (PaSST, 192 chunks: attention backward 15.0 -> 10.2 ms, forward 4.0 -> 3.4 ms per step).
template <typename T> struct Chunk;
template <> struct Chunk<bf16_t> { f32x4 a; };
template <> struct Chunk<float> { f32x4 a, b; };
template <typename T>
__device__ __forceinline__ void fetch_tile64(Chunk<T> (&reg)[2], const T* g, long ld, int gcol, int r0, int N) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int c = threadIdx.x + 256 * i, t = c >> 3, k = c & 7;
        const T* src = g + (long)min(r0 + t, N - 1) * ld + gcol + k * 8;       // clamped: rows past N are zeroed in put_tile64
        reg[i].a = ((const f32x4*)src)[0];
        if constexpr (sizeof(T) == 4) reg[i].b = ((const f32x4*)src)[1];
    }
}
template <typename T>
__device__ __forceinline__ void put_tile64(char* tile, const Chunk<T> (&reg)[2], int r0, int N) {
    constexpr int SB = RowS<T>::value;
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int c = threadIdx.x + 256 * i, t = c >> 3, k = c & 7;
        char* l = tile + t * SB + k * 8 * (int)sizeof(T);
        const bool in = r0 + t < N;
        ((f32x4*)l)[0] = in ? reg[i].a : z;
        if constexpr (sizeof(T) == 4) ((f32x4*)l)[1] = in ? reg[i].b : z;
    }
}
// the calling wave's 32 LDS rows -> global rows [r0, r0+32) (those below N), 64 columns at gcol
template <typename T>
__device__ __forceinline__ void store_wave_rows(const char* rows, T* g, long ld, int gcol, int r0, int N, int lane) {
    constexpr int SB = RowS<T>::value;
    for (int c = lane; c < 32 * 8; c += 64) {
        const int t = c >> 3, k = c & 7;
        if (r0 + t < N) {
            const char* l = rows + t * SB + k * 8 * (int)sizeof(T);
            T* d = g + (long)(r0 + t) * ld + gcol + k * 8;
            if constexpr (sizeof(T) == 2) *(f32x4*)d = *(const f32x4*)l;
            else { ((f32x4*)d)[0] = ((const f32x4*)l)[0]; ((f32x4*)d)[1] = ((const f32x4*)l)[1]; }
        }
    }
}
// a transposed result z[dt] (rows = d in registers, lane&31 = token) -> the wave's LDS rows as [token][d]
template <typename T>
__device__ __forceinline__ void put_transposed(char* rows, const f32x16 (&z)[2], float mul, int lane) {
    constexpr int SB = RowS<T>::value;
    const int r = lane & 31, h2 = lane >> 5;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g)
            store4<T>(rows + r * SB + (dt * 32 + 8 * g + 4 * h2) * (int)sizeof(T), z[dt][4 * g] * mul, z[dt][4 * g + 1] * mul,
                      z[dt][4 * g + 2] * mul, z[dt][4 * g + 3] * mul);
}

__device__ __forceinline__ void zero_tiles(f32x16 (&z)[2]) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) z[t][e] = 0.f;
}

// z[dt] (rows = d, lanes = the accumulator's columns) += M[rows of x][d]^T * x, where x[t] are accumulator tiles whose
// ROWS index the 64 LDS rows of `tile` (t*32 + acc_row) — the accumulator is the B operand, M^T is tr-read from LDS.
template <typename T>
__device__ __forceinline__ void acc_t_product(f32x16 (&z)[2], const f32x16 (&x)[2], const char* tile, int lane) {
    using M = AMma<T>;
    constexpr int SB = RowS<T>::value;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const typename M::Frag fx = M::from_acc(x[t], s);
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) M::mma(M::ld_cols(tile, SB, t * 32 + 16 * s, dt * 32, lane), fx, z[dt]);
        }
}

// ---- forward: one workgroup = 128 queries of one (batch, head); wave = 32 queries ----------------------------------
template <typename T>
__global__ __launch_bounds__(256) void mhsa_fwd_kernel(MhsaArgs a) {
    using M = AMma<T>;
    constexpr int SB = RowS<T>::value;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* qt = smem;
    char* kt = qt + 128 * SB;
    char* vt = kt + 64 * SB;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, h2 = lane >> 5;
    const int head = blockIdx.y, b = blockIdx.z, q0 = blockIdx.x * 128, N = a.N;
    const T* base = (const T*)a.qkv + (long)b * N * 3 * a.E;
    const long ld = 3 * a.E;
    load_rows<T>(qt, base, ld, head * HD, q0, 128, N);
    char* qw = qt + wave * 32 * SB;
    f32x16 ot[2];
    zero_tiles(ot);
    float m = -1e30f, l = 0.f;
    const int nkt = (N + 63) >> 6;
    Chunk<T> rk[2], rv[2];
    fetch_tile64<T>(rk, base, ld, a.E + head * HD, 0, N);
    fetch_tile64<T>(rv, base, ld, 2 * a.E + head * HD, 0, N);
    for (int ki = 0; ki < nkt; ++ki) {
        __syncthreads();
        put_tile64<T>(kt, rk, ki * 64, N);
        put_tile64<T>(vt, rv, ki * 64, N);
        __syncthreads();
        if (ki + 1 < nkt) {
            fetch_tile64<T>(rk, base, ld, a.E + head * HD, (ki + 1) * 64, N);
            fetch_tile64<T>(rv, base, ld, 2 * a.E + head * HD, (ki + 1) * 64, N);
        }
        f32x16 st[2];
        zero_tiles(st);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int off = (16 * kk + 8 * h2) * (int)sizeof(T);
            const typename M::Frag fq = M::ld_row(qw + r * SB + off);
#pragma unroll
            for (int t = 0; t < 2; ++t) M::mma(M::ld_row(kt + (t * 32 + r) * SB + off), fq, st[t]);
        }
        float mx = -1e30f;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int key = ki * 64 + t * 32 + acc_row(e, h2);
                const float s = key < N ? st[t][e] * a.scale : -1e30f;
                st[t][e] = s;
                mx = fmaxf(mx, s);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float mn = fmaxf(m, mx);
        const float alpha = __expf(m - mn);
        float ps = 0.f;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float p = __expf(st[t][e] - mn);
                st[t][e] = p;
                ps += p;
            }
        ps += __shfl_xor(ps, 32, 64);
        l = l * alpha + ps;
        m = mn;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int e = 0; e < 16; ++e) ot[dt][e] *= alpha;
        acc_t_product<T>(ot, st, vt, lane);
    }
    __syncthreads();
    put_transposed<T>(qw, ot, 1.f / l, lane);
    const int q = q0 + wave * 32 + r;
    if (h2 == 0 && q < N) a.lse[((long)b * a.heads + head) * N + q] = m + __logf(l);
    __syncthreads();
    store_wave_rows<T>(qw, (T*)a.out + (long)b * N * a.E, a.E, head * HD, q0 + wave * 32, N, lane);
}

// delta[b, head, q] = sum_d dO[b,q,head*64+d] * O[b,q,head*64+d]; 8 lanes per (row, head)
template <typename T>
__global__ __launch_bounds__(256) void mhsa_delta_kernel(const T* __restrict__ o, const T* __restrict__ dout,
                                                         float* __restrict__ delta, long chunks, int N, int heads) {
    const long c = (long)blockIdx.x * 256 + threadIdx.x;
    float s = 0.f;
    if (c < chunks) {
        float x[8], y[8];
        load8<T>(o + c * 8, x);
        load8<T>(dout + c * 8, y);
#pragma unroll
        for (int i = 0; i < 8; ++i) s += x[i] * y[i];
    }
    s = group_sum<8>(s);
    if (c < chunks && (c & 7) == 0) {
        const long rh = c >> 3;                 // (b*N + q)*heads + head
        const int head = (int)(rh % heads);
        const long row = rh / heads;
        const long bb = row / N;
        const int q = (int)(row - bb * N);
        delta[(bb * heads + head) * N + q] = s;
    }
}

// ---- backward, dQ: same tiling as forward ------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void mhsa_bwd_dq_kernel(MhsaArgs a) {
    using M = AMma<T>;
    constexpr int SB = RowS<T>::value;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* qt = smem;
    char* dot = qt + 128 * SB;
    char* kt = dot + 128 * SB;
    char* vt = kt + 64 * SB;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, h2 = lane >> 5;
    const int head = blockIdx.y, b = blockIdx.z, q0 = blockIdx.x * 128, N = a.N;
    const T* base = (const T*)a.qkv + (long)b * N * 3 * a.E;
    const long ld = 3 * a.E;
    load_rows<T>(qt, base, ld, head * HD, q0, 128, N);
    load_rows<T>(dot, (const T*)a.dout + (long)b * N * a.E, a.E, head * HD, q0, 128, N);
    char* qw = qt + wave * 32 * SB;
    const char* dw = dot + wave * 32 * SB;
    const int q = q0 + wave * 32 + r;
    const long sidx = ((long)b * a.heads + head) * N + q;
    const float lse = q < N ? a.lse[sidx] : 0.f;
    const float dl = q < N ? a.delta[sidx] : 0.f;
    f32x16 dq[2];
    zero_tiles(dq);
    const int nkt = (N + 63) >> 6;
    Chunk<T> rk[2], rv[2];
    fetch_tile64<T>(rk, base, ld, a.E + head * HD, 0, N);
    fetch_tile64<T>(rv, base, ld, 2 * a.E + head * HD, 0, N);
    for (int ki = 0; ki < nkt; ++ki) {
        __syncthreads();
        put_tile64<T>(kt, rk, ki * 64, N);
        put_tile64<T>(vt, rv, ki * 64, N);
        __syncthreads();
        if (ki + 1 < nkt) {
            fetch_tile64<T>(rk, base, ld, a.E + head * HD, (ki + 1) * 64, N);
            fetch_tile64<T>(rv, base, ld, 2 * a.E + head * HD, (ki + 1) * 64, N);
        }
        f32x16 st[2], dp[2];
        zero_tiles(st);
        zero_tiles(dp);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int off = (16 * kk + 8 * h2) * (int)sizeof(T);
            const typename M::Frag fq = M::ld_row(qw + r * SB + off);
            const typename M::Frag fo = M::ld_row(dw + r * SB + off);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                M::mma(M::ld_row(kt + (t * 32 + r) * SB + off), fq, st[t]);
                M::mma(M::ld_row(vt + (t * 32 + r) * SB + off), fo, dp[t]);
            }
        }
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int key = ki * 64 + t * 32 + acc_row(e, h2);
                const float p = key < N ? __expf(st[t][e] * a.scale - lse) : 0.f;
                st[t][e] = p * (dp[t][e] - dl) * a.scale;
            }
        acc_t_product<T>(dq, st, kt, lane);
    }
    __syncthreads();
    put_transposed<T>(qw, dq, 1.f, lane);
    __syncthreads();
    store_wave_rows<T>(qw, (T*)a.dqkv + (long)b * N * 3 * a.E, ld, head * HD, q0 + wave * 32, N, lane);
}

// ---- backward, dK and dV: one workgroup = 128 keys of one (batch, head), streaming 64-query tiles -------------------
template <typename T>
__global__ __launch_bounds__(256) void mhsa_bwd_dkv_kernel(MhsaArgs a) {
    using M = AMma<T>;
    constexpr int SB = RowS<T>::value;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* kt = smem;
    char* vt = kt + 128 * SB;
    char* qt = vt + 128 * SB;
    char* dot = qt + 64 * SB;
    float* lse_s = (float*)(dot + 64 * SB);
    float* dl_s = lse_s + 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, h2 = lane >> 5;
    const int head = blockIdx.y, b = blockIdx.z, k0 = blockIdx.x * 128, N = a.N;
    const T* base = (const T*)a.qkv + (long)b * N * 3 * a.E;
    const T* dob = (const T*)a.dout + (long)b * N * a.E;
    const long ld = 3 * a.E;
    load_rows<T>(kt, base, ld, a.E + head * HD, k0, 128, N);
    load_rows<T>(vt, base, ld, 2 * a.E + head * HD, k0, 128, N);
    char* kw = kt + wave * 32 * SB;
    char* vw = vt + wave * 32 * SB;
    const long sbase = ((long)b * a.heads + head) * N;
    f32x16 dk[2], dv[2];
    zero_tiles(dk);
    zero_tiles(dv);
    const int nqt = (N + 63) >> 6;
    Chunk<T> rq[2], rd[2];
    fetch_tile64<T>(rq, base, ld, head * HD, 0, N);
    fetch_tile64<T>(rd, dob, a.E, head * HD, 0, N);
    for (int qi = 0; qi < nqt; ++qi) {
        __syncthreads();
        put_tile64<T>(qt, rq, qi * 64, N);
        put_tile64<T>(dot, rd, qi * 64, N);
        if (qi + 1 < nqt) {
            fetch_tile64<T>(rq, base, ld, head * HD, (qi + 1) * 64, N);
            fetch_tile64<T>(rd, dob, a.E, head * HD, (qi + 1) * 64, N);
        }
        if (threadIdx.x < 64) {
            const int q = qi * 64 + threadIdx.x;
            lse_s[threadIdx.x] = q < N ? a.lse[sbase + q] : 1e30f;   // exp(s - 1e30) = 0 switches padded queries off
            dl_s[threadIdx.x] = q < N ? a.delta[sbase + q] : 0.f;
        }
        __syncthreads();
        f32x16 s[2], dp[2];
        zero_tiles(s);
        zero_tiles(dp);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int off = (16 * kk + 8 * h2) * (int)sizeof(T);
            const typename M::Frag fk = M::ld_row(kw + r * SB + off);
            const typename M::Frag fv = M::ld_row(vw + r * SB + off);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                M::mma(M::ld_row(qt + (t * 32 + r) * SB + off), fk, s[t]);      // rows = queries, lanes = keys
                M::mma(M::ld_row(dot + (t * 32 + r) * SB + off), fv, dp[t]);
            }
        }
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 lv = *(const f32x4*)(lse_s + t * 32 + 8 * g + 4 * h2);
                const f32x4 dv4 = *(const f32x4*)(dl_s + t * 32 + 8 * g + 4 * h2);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int e = 4 * g + i;
                    const float p = __expf(s[t][e] * a.scale - lv[i]);
                    s[t][e] = p;
                    dp[t][e] = p * (dp[t][e] - dv4[i]) * a.scale;
                }
            }
        acc_t_product<T>(dv, s, dot, lane);
        acc_t_product<T>(dk, dp, qt, lane);
    }
    __syncthreads();
    put_transposed<T>(kw, dk, 1.f, lane);
    put_transposed<T>(vw, dv, 1.f, lane);
    __syncthreads();
    T* dbase = (T*)a.dqkv + (long)b * N * 3 * a.E;
    store_wave_rows<T>(kw, dbase, ld, a.E + head * HD, k0 + wave * 32, N, lane);
    store_wave_rows<T>(vw, dbase, ld, 2 * a.E + head * HD, k0 + wave * 32, N, lane);
}

template <typename K>
int set_lds(K kernel, int bytes) {
    static thread_local const void* done[8];
    static thread_local int ndone = 0;
    for (int i = 0; i < ndone; ++i) if (done[i] == (const void*)kernel) return 0;
    if (hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return -1;
    if (ndone < 8) done[ndone++] = (const void*)kernel;
    return 0;
}

template <typename T>
int launch_fwd(const MhsaArgs& a, hipStream_t s) {
    const int lds = (128 + 64 + 64) * RowS<T>::value;
    if (set_lds(mhsa_fwd_kernel<T>, lds)) return -1;
    hipLaunchKernelGGL(mhsa_fwd_kernel<T>, dim3(pseld_cdiv(a.N, 128), a.heads, a.B), dim3(256), lds, s, a);
    return 0;
}
template <typename T>
int launch_bwd(const MhsaArgs& a, hipStream_t s) {
    const long chunks = (long)a.B * a.N * a.E / 8;
    hipLaunchKernelGGL(mhsa_delta_kernel<T>, dim3(pseld_cdiv(chunks, 256)), dim3(256), 0, s, (const T*)a.out, (const T*)a.dout,
                       a.delta, chunks, a.N, a.heads);
    const int lds_q = (128 + 128 + 64 + 64) * RowS<T>::value;
    const int lds_k = (128 + 128 + 64 + 64) * RowS<T>::value + 128 * (int)sizeof(float);
    if (set_lds(mhsa_bwd_dq_kernel<T>, lds_q) || set_lds(mhsa_bwd_dkv_kernel<T>, lds_k)) return -1;
    const dim3 grid(pseld_cdiv(a.N, 128), a.heads, a.B);
    hipLaunchKernelGGL(mhsa_bwd_dq_kernel<T>, grid, dim3(256), lds_q, s, a);
    hipLaunchKernelGGL(mhsa_bwd_dkv_kernel<T>, grid, dim3(256), lds_k, s, a);
    return 0;
}

}  // namespace

extern "C" int pseld_mhsa_fwd(int dtype, const void* qkv, void* out, float* lse, int B, int N, int E, int heads,
                              void* stream) {
    PSELD_CHECK_ARG(qkv && out && lse, "mhsa_fwd: null pointer");
    PSELD_CHECK_ARG(B > 0 && N > 0 && heads > 0 && E == heads * HD, "mhsa_fwd: built for head_dim 64 (E = 64 * heads)");
    MhsaArgs a{};
    a.qkv = qkv; a.out = out; a.lse = lse; a.B = B; a.N = N; a.E = E; a.heads = heads; a.scale = 0.125f;
    int rc;
    if (dtype == PSELD_BF16) rc = launch_fwd<bf16_t>(a, (hipStream_t)stream);
    else if (dtype == PSELD_F32) rc = launch_fwd<float>(a, (hipStream_t)stream);
    else { pseld_set_error("mhsa_fwd: unknown dtype"); return PSELD_ERR_BAD_ARG; }
    if (rc) { pseld_set_error("mhsa_fwd: cannot reserve LDS"); return PSELD_ERR_HIP; }
    PSELD_LAUNCH_CHECK("mhsa_fwd");
    return PSELD_OK;
}

extern "C" long pseld_mhsa_bwd_workspace(int B, int N, int heads) { return (long)B * N * heads * (long)sizeof(float); }

extern "C" int pseld_mhsa_bwd(int dtype, const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv,
                              int B, int N, int E, int heads, float* workspace, long workspace_bytes, void* stream) {
    PSELD_CHECK_ARG(qkv && out && dout && lse && dqkv && workspace, "mhsa_bwd: null pointer");
    PSELD_CHECK_ARG(B > 0 && N > 0 && heads > 0 && E == heads * HD, "mhsa_bwd: built for head_dim 64 (E = 64 * heads)");
    PSELD_CHECK_ARG(workspace_bytes >= pseld_mhsa_bwd_workspace(B, N, heads), "mhsa_bwd: workspace too small");
    MhsaArgs a{};
    a.qkv = qkv; a.out = const_cast<void*>(out); a.dout = dout; a.dqkv = dqkv; a.lse = const_cast<float*>(lse);
    a.delta = workspace; a.B = B; a.N = N; a.E = E; a.heads = heads; a.scale = 0.125f;
    int rc;
    if (dtype == PSELD_BF16) rc = launch_bwd<bf16_t>(a, (hipStream_t)stream);
    else if (dtype == PSELD_F32) rc = launch_bwd<float>(a, (hipStream_t)stream);
    else { pseld_set_error("mhsa_bwd: unknown dtype"); return PSELD_ERR_BAD_ARG; }
    if (rc) { pseld_set_error("mhsa_bwd: cannot reserve LDS"); return PSELD_ERR_HIP; }
    PSELD_LAUNCH_CHECK("mhsa_bwd");
    return PSELD_OK;
}
