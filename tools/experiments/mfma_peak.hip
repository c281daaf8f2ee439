// Measured MFMA issue ceiling on gfx950: independent v_mfma_f32_32x32x16_bf16 chains, no memory traffic.
// hipcc --offload-arch=gfx950 -O3 tools/experiments/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop(float* out, int iters) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(i * 0.5f); }
    f32x16 acc[NACC];
    for (int n = 0; n < NACC; ++n) for (int e = 0; e < 16; ++e) acc[n][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int n = 0; n < NACC; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[n], 0, 0, 0);
    }
    float s = 0.f;
    for (int n = 0; n < NACC; ++n) for (int e = 0; e < 16; ++e) s += acc[n][e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
__global__ __launch_bounds__(256) void mfma16_loop(float* out, int iters) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(i * 0.5f); }
    f32x4 acc[NACC];
    for (int n = 0; n < NACC; ++n) for (int e = 0; e < 4; ++e) acc[n][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int n = 0; n < NACC; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[n], 0, 0, 0);
    }
    float s = 0.f;
    for (int n = 0; n < NACC; ++n) for (int e = 0; e < 4; ++e) s += acc[n][e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
    float* out; hipMalloc(&out, 4096 * 256 * sizeof(float));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4000;
    for (int blocks : {256, 512, 1024, 2048}) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(mfma_loop<6>, dim3(blocks), dim3(256), 0, 0, out, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double flops = (double)blocks * 4 * iters * 6 * 32768.0;
            if (rep) printf("32x32x16 bf16, %4d blocks x 4 waves, 6 chains: %7.3f ms  %7.1f TFLOP/s\n", blocks, ms, flops / ms / 1e9);
        }
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(mfma16_loop<8>, dim3(blocks), dim3(256), 0, 0, out, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double flops = (double)blocks * 4 * iters * 8 * 16384.0;
            if (rep) printf("16x16x32 bf16, %4d blocks x 4 waves, 8 chains: %7.3f ms  %7.1f TFLOP/s\n", blocks, ms, flops / ms / 1e9);
        }
    }
    // sustained: 400 back-to-back launches (about 1.2 s of pure MFMA issue) — does the rate hold once the power limit reacts?
    for (int chunk = 0; chunk < 4; ++chunk) {
        hipEventRecord(e0);
        for (int i = 0; i < 100; ++i) hipLaunchKernelGGL(mfma_loop<6>, dim3(2048), dim3(256), 0, 0, out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("sustained chunk %d: %8.1f ms  %7.1f TFLOP/s\n", chunk, ms, 100.0 * 2048 * 4 * iters * 6 * 32768.0 / ms / 1e9);
    }
    return 0;
}
