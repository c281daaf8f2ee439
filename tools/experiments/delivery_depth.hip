// How fast can ONE persistent workgroup per CU be fed the operands of a weight-gradient K-loop, and is the limit the bytes in flight?
// (Round 6, DESIGN 4.3: gemm8w delivers 57 KB per 64-token K-tile in ~3.7k cycles = 15.5 B/clk/CU, 1 part from HBM and 1.8 parts re-read from
// L2; the matrix work of a K-tile is 1.5k cycles. The ring of two K-tiles in LDS caps the bytes in flight at ~70 KB per CU.)
// The production geometry is reproduced: 252 workgroups = 12 output tiles (6 row tiles of 256 dY columns x 2 column tiles of 192 X columns)
// x 21 token splits of stage-2 fc1 (49 152 tokens, dY 3 072 B and X 768 B per token), the 12 tiles of a split on one XCD; per K-tile a
// workgroup fetches 64 tokens x (512 B of dY + 384 B of X) and runs 48 MFMA per wave on what it fetched (operands read back from LDS).
//   A: LDS-DMA (global_load_lds_dwordx4) into a ring of two K-tiles, one counted wait per K-tile        (what gemm8w does)
//   B: global_load_dwordx4 into REGISTERS, DEPTH K-tiles ahead, ds_write_b128 into the ring when a slot frees (register-staged prefetch:
//      the register file - 136 free registers per lane in gemm8w - holds the bytes in flight instead of LDS)
//   C: A with the matrix work removed (pure delivery), D: B with the matrix work removed
//   hipcc --offload-arch=gfx950 -O3 -o tools/experiments/delivery_depth tools/experiments/delivery_depth.hip && tools/experiments/delivery_depth
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((address_space(3))) void* lds_ptr;

constexpr int TOK = 49152, DY_B = 3072, X_B = 768;           // bytes per token row
constexpr int KT = 64;                                        // tokens per K-tile
constexpr int TILE_B = KT * (512 + 384);                      // 57 344 B per K-tile
constexpr int PIECES = TILE_B / 16 / 512;                     // 16-byte pieces per thread and K-tile: 7

__device__ __forceinline__ void dma16(unsigned lds_dst, const void* gsrc) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
// source address of 16-byte piece p (0 .. 7*512-1) of K-tile kt for (row tile rt, column tile ct), token base t0
__device__ __forceinline__ const char* src_of(const char* dy, const char* x, int t0, int kt, int rt, int ct, int p) {
    // pieces 0 .. 2047: dY slice (64 tokens x 512 B = 32 pieces per token); 2048 .. 3583: X slice (64 tokens x 384 B = 24 pieces per token)
    if (p < 2048) { const int tok = p >> 5, c = p & 31; return dy + (long)(t0 + kt * KT + tok) * DY_B + rt * 512 + c * 16; }
    const int q = p - 2048, tok = q / 24, c = q - tok * 24;
    return x + (long)(t0 + kt * KT + tok) * X_B + ct * 384 + c * 16;
}

template <int MODE, int DEPTH>     // MODE 0: LDS-DMA ring; 1: register-staged; +2: no matrix work
__global__ __launch_bounds__(512, 2) void feed(const char* dy, const char* x, float* out, int splits, int ktiles, int pattern) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr bool REG = (MODE & 1) != 0, WORK = (MODE & 2) == 0;
    const int id = blockIdx.x, xcd = id & 7, slot = id >> 3, L = xcd * 32 + slot;
    if (slot >= 32 || L >= 12 * splits) return;
    int z = L / 12, t = L - z * 12, rt = t >> 1, ct = t & 1;
    int t0 = z * ktiles * KT;
    // pattern 1: every workgroup its own token rows (no sharing: pure HBM); 2: every workgroup the same rows (pure L2 / memory-side cache)
    if (pattern == 1) { t0 = (L % 21) * ktiles * KT; rt = (L / 21) % 6; ct = (L / 21) & 1; }
    if (pattern == 2) { t0 = 0; rt = 0; ct = 0; }
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds0 = (unsigned)(unsigned long)(lds_ptr)smem;
    f32x4 acc[8] = {};
    auto work = [&](int buf) {                          // 48 MFMA per wave on fragments read from the K-tile's LDS image
        const char* b = smem + buf * TILE_B;
#pragma unroll
        for (int s = 0; s < 6; ++s) {
            const bf16x8 fa = *(const bf16x8*)(b + ((wave * 6 + s) * 64 + lane) * 16 % (TILE_B - 16));
            const bf16x8 fb = *(const bf16x8*)(b + 32768 + ((s * 8 + wave) * 64 + lane) * 16 % (TILE_B - 32768 - 16));
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, acc[i], 0, 0, 0);
        }
    };
    if constexpr (!REG) {
        auto issue = [&](int kt, int buf) {
#pragma unroll
            for (int j = 0; j < PIECES; ++j) {
                const int p = (j * 8 + wave) * 64 + lane;                      // a wave-instruction covers 64 consecutive pieces = 1 KiB of the image
                dma16(lds0 + buf * TILE_B + (j * 8 + wave) * 1024, src_of(dy, x, t0, kt, rt, ct, p));
            }
        };
        issue(0, 0);
        if (ktiles > 1) issue(1, 1);
        for (int kt = 0; kt < ktiles; ++kt) {
            if (kt + 1 < ktiles) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES) : "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if constexpr (WORK) work(kt & 1);
            __builtin_amdgcn_s_barrier();                                       // everyone is done with the slot
            if (kt + 2 < ktiles) issue(kt + 2, kt & 1);
        }
    } else {
        u32x4 r[DEPTH][PIECES];
        auto fetch = [&](int kt, int d) {
#pragma unroll
            for (int j = 0; j < PIECES; ++j) r[d][j] = *(const u32x4*)src_of(dy, x, t0, kt, rt, ct, (j * 8 + wave) * 64 + lane);
        };
        auto put = [&](int d, int buf) {
#pragma unroll
            for (int j = 0; j < PIECES; ++j) *(u32x4*)(smem + buf * TILE_B + ((j * 8 + wave) * 64 + lane) * 16) = r[d][j];
        };
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) if (d < ktiles) fetch(d, d);
        // K-tile kt lives in register set kt % DEPTH until it is written to LDS slot kt & 1, one K-tile before it is consumed
        put(0, 0);
        if (DEPTH < ktiles) fetch(DEPTH, 0);
        __builtin_amdgcn_s_barrier();
        for (int kt0 = 0; kt0 < ktiles; kt0 += DEPTH) {
#pragma unroll
            for (int dd = 0; dd < DEPTH; ++dd) {
                const int kt = kt0 + dd;
                if (kt < ktiles) {
                    const int nd = (dd + 1) % DEPTH;                             // register set of K-tile kt + 1
                    if (kt + 1 < ktiles) put(nd, (kt + 1) & 1);                 // slot (kt + 1) & 1 was freed by the barrier behind K-tile kt - 1
                    if (kt + 1 + DEPTH < ktiles) fetch(kt + 1 + DEPTH, nd);
                    if constexpr (WORK) work(kt & 1);
                    __builtin_amdgcn_s_barrier();
                }
            }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
    if (s == 12345.678f) out[id] = s;
}

// G: as A / C, but the HBM misses are taken out of the consumers' request streams: the 12 workgroups of a token split share the job of
// TOUCHING the split's unique bytes of K-tile kt + TD (64 tokens x 3 840 B = 1 920 lines: 160 lines per workgroup = 2.5 wave-instructions of
// one 4-byte load per lane and line, issued by wave 7), so that the LDS-DMA of every workgroup finds its lines in L2.
template <bool WORK, int TD>
__global__ __launch_bounds__(512, 2) void feed_touch(const char* dy, const char* x, float* out, int splits, int ktiles) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int id = blockIdx.x, xcd = id & 7, slot = id >> 3, L = xcd * 32 + slot;
    if (slot >= 32 || L >= 12 * splits) return;
    const int z = L / 12, t = L - z * 12, rt = t >> 1, ct = t & 1;
    const int t0 = z * ktiles * KT;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds0 = (unsigned)(unsigned long)(lds_ptr)smem;
    f32x4 acc[8] = {};
    unsigned sink = 0, tv[3] = {0u, 0u, 0u};
    // (plain loads the compiler tracks; their values are consumed by the NEXT touch, a K-tile later)
    auto touch = [&](int kt) {                           // this workgroup's twelfth of the split's lines of K-tile kt: lines [160 t, 160 t + 160)
        if (wave != 7) return;
        sink ^= tv[0] ^ tv[1] ^ tv[2];
        if (kt >= ktiles) return;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int ln = min(160 * t + j * 64 + lane, 160 * t + 159);      // 0 .. 1535: dY lines (24 per token), 1536 .. 1919: X lines (6 per token)
            const char* p = ln < 1536 ? dy + (long)(t0 + kt * KT + ln / 24) * DY_B + (ln % 24) * 128
                                      : x + (long)(t0 + kt * KT + (ln - 1536) / 6) * X_B + ((ln - 1536) % 6) * 128;
            tv[j] = *(const volatile unsigned*)p;
        }
    };
    auto work = [&](int buf) {
        const char* b = smem + buf * TILE_B;
#pragma unroll
        for (int s = 0; s < 6; ++s) {
            const bf16x8 fa = *(const bf16x8*)(b + ((wave * 6 + s) * 64 + lane) * 16 % (TILE_B - 16));
            const bf16x8 fb = *(const bf16x8*)(b + 32768 + ((s * 8 + wave) * 64 + lane) * 16 % (TILE_B - 32768 - 16));
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, acc[i], 0, 0, 0);
        }
    };
    auto issue = [&](int kt, int buf) {
#pragma unroll
        for (int j = 0; j < PIECES; ++j) dma16(lds0 + buf * TILE_B + (j * 8 + wave) * 1024, src_of(dy, x, t0, kt, rt, ct, (j * 8 + wave) * 64 + lane));
    };
#pragma unroll
    for (int k = 0; k < TD + 2; ++k) { touch(k); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
    __builtin_amdgcn_s_barrier();
    issue(0, 0);
    if (ktiles > 1) issue(1, 1);
    for (int kt = 0; kt < ktiles; ++kt) {
        // (wave 7 has 3 touch loads per K-tile in its queue besides its 7 LDS-DMA pieces: it waits for everything but the youngest K-tile's pieces
        //  and touches - conservative, the touches are TD K-tiles old by then)
        if (kt + 1 < ktiles) { if (wave == 7) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES) : "memory"); else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES) : "memory"); }
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if constexpr (WORK) work(kt & 1);
        __builtin_amdgcn_s_barrier();
        touch(kt + 2 + TD);
        if (kt + 2 < ktiles) issue(kt + 2, kt & 1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    sink ^= tv[0] ^ tv[1] ^ tv[2];
    float s = (float)sink;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
    if (s == 12345.678f) out[id] = s;
}

int main() {
    const int splits = 21, ktiles = TOK / splits / KT;          // 36 K-tiles of 64 tokens per workgroup
    char *dy, *x; float* out;
    hipMalloc(&dy, (size_t)TOK * DY_B); hipMalloc(&x, (size_t)TOK * X_B); hipMalloc(&out, 4096);
    hipMemset(dy, 0x11, (size_t)TOK * DY_B); hipMemset(x, 0x22, (size_t)TOK * X_B);
    char* junk; hipMalloc(&junk, 768u << 20);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto time = [&](auto kern, int lds, const char* name, int pattern = 0) {
        hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        std::vector<float> us;
        for (int r = 0; r < 12; ++r) {
            hipMemsetAsync(junk, r, 768u << 20);                                  // operands out of the memory-side cache: as inside the step
            hipEventRecord(e0);
            hipLaunchKernelGGL(kern, dim3(256), dim3(512), lds, 0, (const char*)dy, (const char*)x, out, splits, ktiles, pattern);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float t; hipEventElapsedTime(&t, e0, e1); us.push_back(t * 1e3f);
        }
        std::sort(us.begin(), us.end());
        const float med = us[us.size() / 2];
        const double delivered = 252.0 * ktiles * TILE_B;
        printf("%-64s %7.1f us | delivered %.0f MB at %.2f TB/s = %.1f B/clk/CU at 2.2 GHz | %.0f cycles per K-tile\n", name, med, delivered / 1e6, delivered / med / 1e6,
               delivered / 252 / (med * 1e-6 * 2.2e9), med * 1e-6 * 2.2e9 / ktiles);
    };
    time(feed<0, 1>, 2 * TILE_B, "A  LDS-DMA ring of two K-tiles + 48 MFMA per wave and K-tile");
    time(feed<1, 2>, 2 * TILE_B, "B2 register-staged, 2 K-tiles ahead + matrix work");
    time(feed<1, 3>, 2 * TILE_B, "B3 register-staged, 3 K-tiles ahead + matrix work");
    time(feed<1, 4>, 2 * TILE_B, "B4 register-staged, 4 K-tiles ahead + matrix work");
    time(feed<2, 1>, 2 * TILE_B, "C  LDS-DMA ring alone (no matrix work)");
    time(feed<3, 3>, 2 * TILE_B, "D3 register-staged alone, 3 K-tiles ahead");
    time(feed<3, 4>, 2 * TILE_B, "D4 register-staged alone, 4 K-tiles ahead");
    // (pattern 1: 12 x 21 workgroups read 12 DISJOINT 512 / 384-byte column slices of the 21 token ranges: every byte from HBM exactly once, same
    //  strided shape; pattern 2: all workgroups read the same 36 K-tiles: everything but the first touch from L2 / the memory-side cache)
    time(feed<2, 1>, 2 * TILE_B, "C1 LDS-DMA alone, no sharing between workgroups (all from HBM)", 1);
    time(feed<3, 4>, 2 * TILE_B, "D1 register-staged alone, 4 ahead, no sharing (all from HBM)", 1);
    time(feed<2, 1>, 2 * TILE_B, "C2 LDS-DMA alone, every workgroup the same rows (all from L2)", 2);
    time(feed<3, 4>, 2 * TILE_B, "D2 register-staged alone, 4 ahead, every workgroup the same rows", 2);
    time(feed<0, 1>, 2 * TILE_B, "A2 LDS-DMA + matrix work, every workgroup the same rows", 2);
    auto time_t = [&](auto kern, const char* name) {
        hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * TILE_B);
        std::vector<float> us;
        for (int r = 0; r < 12; ++r) {
            hipMemsetAsync(junk, r, 768u << 20);
            hipEventRecord(e0);
            hipLaunchKernelGGL(kern, dim3(256), dim3(512), 2 * TILE_B, 0, (const char*)dy, (const char*)x, out, splits, ktiles);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float t; hipEventElapsedTime(&t, e0, e1); us.push_back(t * 1e3f);
        }
        std::sort(us.begin(), us.end());
        const float med = us[us.size() / 2];
        const double delivered = 252.0 * ktiles * TILE_B;
        printf("%-64s %7.1f us | delivered %.0f MB at %.2f TB/s = %.1f B/clk/CU at 2.2 GHz | %.0f cycles per K-tile\n", name, med, delivered / 1e6, delivered / med / 1e6,
               delivered / 252 / (med * 1e-6 * 2.2e9), med * 1e-6 * 2.2e9 / ktiles);
    };
    time_t(feed_touch<false, 1>, "G1 LDS-DMA alone + cooperative L2 touches 1 K-tile further ahead");
    time_t(feed_touch<false, 2>, "G2 LDS-DMA alone + cooperative L2 touches 2 K-tiles further ahead");
    time_t(feed_touch<false, 4>, "G4 LDS-DMA alone + cooperative L2 touches 4 K-tiles further ahead");
    time_t(feed_touch<true, 2>, "G2w LDS-DMA + matrix work + touches 2 K-tiles further ahead");
    return 0;
}
