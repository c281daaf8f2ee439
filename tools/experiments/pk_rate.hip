// Issue rate of v_fma_f32 vs v_pk_fma_f32 (and v_exp_f32) on gfx950: N independent chains per lane, 2 waves per SIMD.
// hipcc --offload-arch=gfx950 -O3 -o pk_rate tools/experiments/pk_rate.hip && ./pk_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(2))) float f32x2;
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    float a[16];
    f32x2 p[8];
    for (int i = 0; i < 16; ++i) a[i] = threadIdx.x * 1e-3f + i;
    for (int i = 0; i < 8; ++i) p[i] = f32x2{a[2 * i], a[2 * i + 1]};
    const float c = 0.999f, d = 1e-3f;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c), "v"(d));
        } else if (MODE == 1) {
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(f32x2{c, c}), "v"(f32x2{d, d}));
        } else if (MODE == 2) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(f32x2{c, c}));
        }
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += a[i];
    for (int i = 0; i < 8; ++i) s += p[i][0] + p[i][1];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE> float run(float* out, int iters) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(256 * 2), dim3(256), 0, 0, out, iters);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(256 * 2), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
    float* out; hipMalloc(&out, 256 * 2 * 256 * 4);
    const int iters = 20000;
    const char* names[4] = {"v_fma_f32 (16 per iter)", "v_pk_fma_f32 (8 per iter)", "v_exp_f32 (16 per iter)", "v_pk_mul_f32 (8 per iter)"};
    float ms[4] = {run<0>(out, iters), run<1>(out, iters), run<2>(out, iters), run<3>(out, iters)};
    for (int m = 0; m < 4; ++m) {
        const double n = (m == 1 || m == 3) ? 8.0 : 16.0;
        // 2 workgroups x 4 waves per CU = 2 waves per SIMD; instructions per SIMD = 2 * n * iters
        printf("%-28s %8.3f ms  -> %.2f ns per wave-instruction per SIMD (%.1f cycles at 2.4 GHz)\n", names[m], ms[m], ms[m] * 1e6 / (2 * n * iters), ms[m] * 1e6 / (2 * n * iters) * 2.4);
    }
    return 0;
}
