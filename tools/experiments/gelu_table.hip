// GELU evaluation cost on gfx950 at 2 waves per SIMD: A&S 7.1.26 erf (1 rcp + 1 exp + FMAs) vs an LDS lookup table with linear interpolation
// (1024 entries over [-8, 8], 16 bytes per entry: gelu, gelu' and their forward differences; |error| ~ 2e-5) on N(0, 1.5) inputs (bank
// conflicts as the activations produce them).  hipcc --offload-arch=gfx950 -O3 -o gelu_table tools/experiments/gelu_table.hip && ./gelu_table
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(4))) float f32x4;
__device__ __forceinline__ void gelu_both_as(float x, float& y, float& dy) {
    const float ax = fabsf(x) * 0.70710678118654752f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f); p = fmaf(p, t, -0.284496736f); p = fmaf(p, t, 0.254829592f);
    const float e = __expf(-ax * ax);
    const float erfv = copysignf(1.0f - p * t * e, x);
    const float cdf = 0.5f * (1.0f + erfv);
    y = x * cdf; dy = cdf + x * 0.3989422804014327f * e;
}
template <int MODE>
__global__ __launch_bounds__(256) void k(const float* __restrict__ in, float* __restrict__ out, const f32x4* __restrict__ tab, int iters) {
    __shared__ f32x4 lt[1024];
    for (int i = threadIdx.x; i < 1024; i += 256) lt[i] = tab[i];
    __syncthreads();
    float x[16], a[16], b[16];
    for (int i = 0; i < 16; ++i) { x[i] = in[(blockIdx.x * 256 + threadIdx.x) * 16 + i]; a[i] = 0.f; b[i] = 0.f; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            float y, dy;
            if (MODE == 0) gelu_both_as(x[i], y, dy);
            else {
                const float u = fminf(fmaxf(fmaf(x[i], 64.0f, 512.0f), 0.0f), 1022.99f);
                const int idx = (int)u;
                const float fr = u - (float)idx;
                const f32x4 e = lt[idx];
                y = fmaf(e[1], fr, e[0]); dy = fmaf(e[3], fr, e[2]);
            }
            a[i] += y; b[i] += dy;
            x[i] = x[i] * 0.999f + 1e-4f * dy;      // keep the chain data dependent
        }
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += a[i] + b[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
    const int nb = 512, n = nb * 256 * 16, iters = 2000;
    float* hin = (float*)malloc(n * 4);
    srand(1);
    for (int i = 0; i < n; ++i) { float u1 = (rand() + 1.0f) / (RAND_MAX + 2.0f), u2 = rand() / (float)RAND_MAX; hin[i] = 1.5f * sqrtf(-2 * logf(u1)) * cosf(6.2831853f * u2); }
    f32x4* htab = (f32x4*)malloc(1024 * 16);
    auto G = [](double x) { return 0.5 * x * (1 + erf(x / sqrt(2.0))); };
    auto D = [](double x) { return 0.5 * (1 + erf(x / sqrt(2.0))) + x * exp(-0.5 * x * x) / sqrt(2 * M_PI); };
    for (int i = 0; i < 1024; ++i) { double x0 = (i - 512) / 64.0, x1 = (i - 511) / 64.0; htab[i] = f32x4{(float)G(x0), (float)(G(x1) - G(x0)), (float)D(x0), (float)(D(x1) - D(x0))}; }
    float *din, *dout; f32x4* dtab;
    hipMalloc(&din, n * 4); hipMalloc(&dout, nb * 256 * 4); hipMalloc(&dtab, 1024 * 16);
    hipMemcpy(din, hin, n * 4, hipMemcpyHostToDevice); hipMemcpy(dtab, htab, 1024 * 16, hipMemcpyHostToDevice);
    // accuracy of the table on the sample
    double worst = 0, worstd = 0;
    for (int i = 0; i < 100000; ++i) { double x = hin[i]; double u = fmin(fmax(x * 64 + 512, 0), 1022.99); int idx = (int)u; double fr = u - idx;
        worst = fmax(worst, fabs(htab[idx][0] + htab[idx][1] * fr - G(x))); worstd = fmax(worstd, fabs(htab[idx][2] + htab[idx][3] * fr - D(x))); }
    printf("table max |error| on the sample: gelu %.2e, gelu' %.2e\n", worst, worstd);
    for (int mode = 0; mode < 2; ++mode) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(nb), dim3(256), 0, 0, din, dout, dtab, iters);
            else hipLaunchKernelGGL(k<1>, dim3(nb), dim3(256), 0, 0, din, dout, dtab, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        // 512 blocks x 4 waves on 256 CUs = 2 waves per SIMD; per SIMD: 2 waves x 16 x iters evaluations of 64 lanes
        printf("%s: %.3f ms -> %.1f cycles per wave-evaluation (gelu and gelu' of 64 lanes) per SIMD at 2.4 GHz\n", mode ? "LDS table" : "A&S erf  ", ms, ms * 1e-3 * 2.4e9 / (2.0 * 16 * iters));
    }
    return 0;
}
