// Which part of the forward-GEMM slice loop caps it at ~36 % of the MFMA rate? The loop of gemm_dma_kernel<2,2> rebuilt
// piece by piece: V0 MFMAs only; V1 + the 10 ds_read_b128 fragment loads per slice; V2 + one s_barrier per slice; V3 + the
// five LDS-DMA loads per wave per slice (from an L2-resident buffer) with s_waitcnt vmcnt(0) before the barrier.
// hipcc --offload-arch=gfx950 -O3 tools/experiments/loop_ceiling.hip -o tools/experiments/loop_ceiling
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void* lds_void_ptr;
typedef const __attribute__((address_space(1))) void* gbl_void_ptr;

template <int V, int WTM, int WTN>      // wave tile = WTM*32 x WTN*32
__global__ __launch_bounds__(256) void loop_kernel(const __bf16* src, float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int STAGE = 20480;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    for (int i = tid; i < 2 * STAGE / 4; i += 256) ((float*)smem)[i] = 0.001f * (i & 63);
    __syncthreads();
    f32x16 acc[WTM][WTN];
    for (int a = 0; a < WTM; ++a) for (int b = 0; b < WTN; ++b) for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;
    bf16x8 fa[WTM], fb[WTN];
    for (int a = 0; a < WTM; ++a) for (int i = 0; i < 8; ++i) fa[a][i] = (__bf16)(0.01f * (lane + i));
    for (int b = 0; b < WTN; ++b) for (int i = 0; i < 8; ++i) fb[b][i] = (__bf16)(0.02f * (lane - i));
    for (int s = 0; s < iters; ++s) {
        if (V >= 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (V >= 2) __builtin_amdgcn_s_barrier();
        const char* As = smem + (s & 1) * STAGE;
        const char* Bs = As + 8192;
        if (V >= 3) {
            char* st = smem + ((s + 1) & 1) * STAGE;
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                const int i = wave + 4 * j;
                const int row = i * 16 + (lane >> 2), chunk = (lane & 3) ^ ((row >> 2) & 3);
                __builtin_amdgcn_global_load_lds((gbl_void_ptr)(src + (long)(blockIdx.x % 64 * 320 + row) * 64 + ((s & 1) * 32) + chunk * 8),
                                                 (lds_void_ptr)(st + i * 1024), 16, 0, 0);
            }
        }
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            if (V >= 1) {
#pragma unroll
                for (int a = 0; a < WTM; ++a) {
                    const int row = wm * 64 + a * 32 + r;
                    fa[a] = *(const bf16x8*)(As + (row & 127) * 64 + (((2 * kk + h) ^ ((row >> 2) & 3)) << 4));
                }
#pragma unroll
                for (int b = 0; b < WTN; ++b) {
                    const int row = wn * 96 + b * 32 + r;
                    fb[b] = *(const bf16x8*)(Bs + (row % 192) * 64 + (((2 * kk + h) ^ ((row >> 2) & 3)) << 4));
                }
            }
#pragma unroll
            for (int a = 0; a < WTM; ++a)
#pragma unroll
                for (int b = 0; b < WTN; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[b], fa[a], acc[a][b], 0, 0, 0);
        }
    }
    float sum = 0.f;
    for (int a = 0; a < WTM; ++a) for (int b = 0; b < WTN; ++b) for (int e = 0; e < 16; ++e) sum += acc[a][b][e];
    out[blockIdx.x * 256 + tid] = sum;
}

template <int V, int WTM, int WTN>
void run(const char* name, const __bf16* src, float* out, size_t lds) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000, blocks = 256 * 3 * 4;
    hipFuncSetAttribute((const void*)loop_kernel<V, WTM, WTN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((loop_kernel<V, WTM, WTN>), dim3(blocks), dim3(256), lds, 0, src, out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    const double flops = (double)blocks * 4 * iters * 2 * WTM * WTN * 32768.0;
    printf("%-58s wave tile %3dx%3d, LDS %3zu KB/wg: %7.1f TFLOP/s\n", name, WTM * 32, WTN * 32, lds / 1024, flops / ms / 1e9);
}

// BK = 64 variant: 128-byte rows (one full cache line per row and slice), 4 k-steps per slice, XOR swizzle over 8 chunks
template <int V, int WTM, int WTN>
__global__ __launch_bounds__(256) void loop64_kernel(const __bf16* src, float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int STAGE = 40960;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    for (int i = tid; i < 2 * STAGE / 4; i += 256) ((float*)smem)[i] = 0.001f * (i & 63);
    __syncthreads();
    f32x16 acc[WTM][WTN];
    for (int a = 0; a < WTM; ++a) for (int b = 0; b < WTN; ++b) for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;
    bf16x8 fa[WTM], fb[WTN];
    for (int a = 0; a < WTM; ++a) for (int i = 0; i < 8; ++i) fa[a][i] = (__bf16)(0.01f * (lane + i));
    for (int b = 0; b < WTN; ++b) for (int i = 0; i < 8; ++i) fb[b][i] = (__bf16)(0.02f * (lane - i));
    for (int s = 0; s < iters; ++s) {
        if (V >= 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (V >= 2) __builtin_amdgcn_s_barrier();
        const char* As = smem + (s & 1) * STAGE;
        const char* Bs = As + 16384;
        if (V >= 3) {
            char* st = smem + ((s + 1) & 1) * STAGE;
#pragma unroll
            for (int j = 0; j < 10; ++j) {
                const int i = wave + 4 * j;
                const int row = i * 8 + (lane >> 3), chunk = (lane & 7) ^ (row & 7);
                __builtin_amdgcn_global_load_lds((gbl_void_ptr)(src + (long)(blockIdx.x % 64 * 320 + row) * 64 + chunk * 8),
                                                 (lds_void_ptr)(st + i * 1024), 16, 0, 0);
            }
        }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            if (V >= 1) {
#pragma unroll
                for (int a = 0; a < WTM; ++a) {
                    const int row = wm * 64 + a * 32 + r;
                    fa[a] = *(const bf16x8*)(As + (row & 127) * 128 + (((2 * kk + h) ^ (row & 7)) << 4));
                }
#pragma unroll
                for (int b = 0; b < WTN; ++b) {
                    const int row = wn * 96 + b * 32 + r;
                    fb[b] = *(const bf16x8*)(Bs + (row % 192) * 128 + (((2 * kk + h) ^ (row & 7)) << 4));
                }
            }
#pragma unroll
            for (int a = 0; a < WTM; ++a)
#pragma unroll
                for (int b = 0; b < WTN; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[b], fa[a], acc[a][b], 0, 0, 0);
        }
    }
    float sum = 0.f;
    for (int a = 0; a < WTM; ++a) for (int b = 0; b < WTN; ++b) for (int e = 0; e < 16; ++e) sum += acc[a][b][e];
    out[blockIdx.x * 256 + tid] = sum;
}
template <int V, int WTM, int WTN>
void run64(const char* name, const __bf16* src, float* out, size_t lds) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 1000, blocks = 256 * 3 * 4;
    hipFuncSetAttribute((const void*)loop64_kernel<V, WTM, WTN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((loop64_kernel<V, WTM, WTN>), dim3(blocks), dim3(256), lds, 0, src, out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    const double flops = (double)blocks * 4 * iters * 4 * WTM * WTN * 32768.0;
    printf("BK=64 %-52s wave tile %3dx%3d, LDS %3zu KB/wg: %7.1f TFLOP/s\n", name, WTM * 32, WTN * 32, lds / 1024, flops / ms / 1e9);
}

// ring variant of the BK = 32 loop: NST stages, NST - 1 slices in flight, vmcnt leaves the NST - 2 younger ones outstanding
template <int NST, int WTM, int WTN>
__global__ __launch_bounds__(256) void ring_kernel(const __bf16* src, float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int STAGE = 20480;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    for (int i = tid; i < NST * STAGE / 4; i += 256) ((float*)smem)[i] = 0.001f * (i & 63);
    __syncthreads();
    f32x16 acc[WTM][WTN];
    for (int a = 0; a < WTM; ++a) for (int b = 0; b < WTN; ++b) for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;
    bf16x8 fa[WTM], fb[WTN];
    auto issue = [&](int s) {
        char* st = smem + (s % NST) * STAGE;
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const int i = wave + 4 * j;
            const int row = i * 16 + (lane >> 2), chunk = (lane & 3) ^ ((row >> 2) & 3);
            __builtin_amdgcn_global_load_lds((gbl_void_ptr)(src + (long)(blockIdx.x % 64 * 320 + row) * 64 + ((s & 1) * 32) + chunk * 8),
                                             (lds_void_ptr)(st + i * 1024), 16, 0, 0);
        }
    };
    for (int p = 0; p < NST - 1; ++p) issue(p);
    for (int s = 0; s < iters; ++s) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 2) * 5) : "memory");
        __builtin_amdgcn_s_barrier();
        issue(s + NST - 1);
        const char* As = smem + (s % NST) * STAGE;
        const char* Bs = As + 8192;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
            for (int a = 0; a < WTM; ++a) {
                const int row = wm * 64 + a * 32 + r;
                fa[a] = *(const bf16x8*)(As + (row & 127) * 64 + (((2 * kk + h) ^ ((row >> 2) & 3)) << 4));
            }
#pragma unroll
            for (int b = 0; b < WTN; ++b) {
                const int row = wn * 96 + b * 32 + r;
                fb[b] = *(const bf16x8*)(Bs + (row % 192) * 64 + (((2 * kk + h) ^ ((row >> 2) & 3)) << 4));
            }
#pragma unroll
            for (int a = 0; a < WTM; ++a)
#pragma unroll
                for (int b = 0; b < WTN; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[b], fa[a], acc[a][b], 0, 0, 0);
        }
    }
    float sum = 0.f;
    for (int a = 0; a < WTM; ++a) for (int b = 0; b < WTN; ++b) for (int e = 0; e < 16; ++e) sum += acc[a][b][e];
    out[blockIdx.x * 256 + tid] = sum;
}
template <int NST, int WTM, int WTN>
void runring(const __bf16* src, float* out, size_t lds) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000, blocks = 256 * 3 * 4;
    hipFuncSetAttribute((const void*)ring_kernel<NST, WTM, WTN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((ring_kernel<NST, WTM, WTN>), dim3(blocks), dim3(256), lds, 0, src, out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    const double flops = (double)blocks * 4 * iters * 2 * WTM * WTN * 32768.0;
    printf("ring, %d stages (%d slices in flight)                          wave tile %3dx%3d, LDS %3zu KB/wg: %7.1f TFLOP/s\n", NST, NST - 1,
           WTM * 32, WTN * 32, lds / 1024, flops / ms / 1e9);
}

int main() {
    __bf16* src; float* out;
    hipMalloc(&src, 64 * 320 * 64 * 2 + 4096); hipMemset(src, 0, 64 * 320 * 64 * 2 + 4096);
    hipMalloc(&out, 256 * 12 * 256 * sizeof(float));
    for (size_t lds : {(size_t)51200, (size_t)80000}) {      // 3 and 2 workgroups per CU
        run<0, 2, 3>("V0 MFMA only", src, out, lds);
        run<1, 2, 3>("V1 + fragment ds_read_b128", src, out, lds);
        run<2, 2, 3>("V2 + s_barrier per slice", src, out, lds);
        run<3, 2, 3>("V3 + LDS-DMA loads, vmcnt(0) before the barrier", src, out, lds);
    }
    run<1, 4, 3>("V1 + fragment ds_read_b128", src, out, 80000);
    run<2, 4, 3>("V2 + s_barrier per slice", src, out, 80000);
    run<3, 4, 3>("V3 + LDS-DMA loads", src, out, 80000);
    runring<2, 2, 3>(src, out, 51200);
    runring<3, 2, 3>(src, out, 3 * 20480);
    runring<3, 2, 3>(src, out, 80000);
    runring<4, 2, 3>(src, out, 4 * 20480);
    runring<6, 2, 3>(src, out, 6 * 20480);
    run64<1, 2, 3>("V1 + fragment ds_read_b128", src, out, 81920);
    run64<2, 2, 3>("V2 + s_barrier per slice", src, out, 81920);
    run64<3, 2, 3>("V3 + LDS-DMA loads (128-byte rows)", src, out, 81920);
    return 0;
}
