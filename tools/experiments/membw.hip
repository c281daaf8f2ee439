// HBM yardstick for the byte ledger of DESIGN.md: hand-written streamed read / fill / float4 copy kernels (16 B per lane, grid-stride over
// whole-chip grids), timed with HIP events at the tensor sizes of the HTS-AT step (one stage-2 row tensor 37.7 MB ... the 1.2 GB of a stage-0
// block). MI355X_MICROARCH.md quotes 6.29 TB/s for a float4 copy and 6.0-6.1 TB/s for streamed reads; torch.Tensor.copy_ (what tools/membw.py
// timed until round 4) reaches 5.0 on the same box.
//   hipcc --offload-arch=gfx950 -O3 -o tools/experiments/membw tools/experiments/membw.hip && tools/experiments/membw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

typedef float f4 __attribute__((ext_vector_type(4)));

template <int U>
__global__ __launch_bounds__(256) void copy_kernel(const f4* __restrict__ src, f4* __restrict__ dst, long n) {
    const long stride = (long)gridDim.x * 256 * U;
    for (long i = (long)blockIdx.x * 256 * U + threadIdx.x; i < n; i += stride) {
        f4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) if (i + u * 256 < n) v[u] = src[i + u * 256];
#pragma unroll
        for (int u = 0; u < U; ++u) if (i + u * 256 < n) dst[i + u * 256] = v[u];
    }
}
template <int U>
__global__ __launch_bounds__(256) void copy_nt_kernel(const f4* __restrict__ src, f4* __restrict__ dst, long n) {
    const long stride = (long)gridDim.x * 256 * U;
    for (long i = (long)blockIdx.x * 256 * U + threadIdx.x; i < n; i += stride) {
        f4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) if (i + u * 256 < n) v[u] = __builtin_nontemporal_load(src + i + u * 256);
#pragma unroll
        for (int u = 0; u < U; ++u) if (i + u * 256 < n) __builtin_nontemporal_store(v[u], dst + i + u * 256);
    }
}
template <int U>
__global__ __launch_bounds__(256) void read_kernel(const f4* __restrict__ src, float* __restrict__ out, long n) {
    const long stride = (long)gridDim.x * 256 * U;
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    for (long i = (long)blockIdx.x * 256 * U + threadIdx.x; i < n; i += stride) {
        f4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = (i + u * 256 < n) ? src[i + u * 256] : f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u];
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) out[0] = acc[0];       // never true: keeps the loads alive
}
__global__ __launch_bounds__(256) void fill_kernel(f4* __restrict__ dst, long n, float val) {
    const long stride = (long)gridDim.x * 256;
    const f4 v = {val, val, val, val};
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) dst[i] = v;
}

template <class F>
static double time_us(F&& fn, int reps = 20) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    fn(); fn();
    hipDeviceSynchronize();
    std::vector<float> t;
    for (int r = 0; r < reps; ++r) {
        hipEventRecord(a, 0); fn(); hipEventRecord(b, 0); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); t.push_back(ms * 1e3f);
    }
    std::sort(t.begin(), t.end());
    return t[t.size() / 2];
}

int main() {
    const long sizes_mb[] = {38, 151, 453, 1208};
    float* sink; hipMalloc(&sink, 64);
    printf("%-8s %-28s %10s %10s\n", "MB", "kernel", "us", "TB/s (bytes moved)");
    for (long mb : sizes_mb) {
        const long bytes = mb * 1000 * 1000 / 4096 * 4096, n = bytes / 16;
        f4 *a, *b;
        hipMalloc(&a, bytes); hipMalloc(&b, bytes);
        hipMemset(a, 1, bytes); hipMemset(b, 2, bytes);
        for (int per : {4, 8, 16}) {                                    // workgroups of 256 threads per CU
            const int grid = 256 * per;
            double t;
            t = time_us([&] { hipLaunchKernelGGL(read_kernel<4>, dim3(grid), dim3(256), 0, 0, a, sink, n); });
            printf("%-8ld read   U=4  %2d wg/CU           %10.1f %10.2f\n", mb, per, t, bytes / t * 1e-6);
            t = time_us([&] { hipLaunchKernelGGL(fill_kernel, dim3(grid), dim3(256), 0, 0, b, n, 1.0f); });
            printf("%-8ld fill        %2d wg/CU           %10.1f %10.2f\n", mb, per, t, bytes / t * 1e-6);
            t = time_us([&] { hipLaunchKernelGGL(copy_kernel<1>, dim3(grid), dim3(256), 0, 0, a, b, n); });
            printf("%-8ld copy   U=1  %2d wg/CU           %10.1f %10.2f\n", mb, per, t, 2.0 * bytes / t * 1e-6);
            t = time_us([&] { hipLaunchKernelGGL(copy_kernel<4>, dim3(grid), dim3(256), 0, 0, a, b, n); });
            printf("%-8ld copy   U=4  %2d wg/CU           %10.1f %10.2f\n", mb, per, t, 2.0 * bytes / t * 1e-6);
            t = time_us([&] { hipLaunchKernelGGL(copy_nt_kernel<4>, dim3(grid), dim3(256), 0, 0, a, b, n); });
            printf("%-8ld copy nt U=4 %2d wg/CU           %10.1f %10.2f\n", mb, per, t, 2.0 * bytes / t * 1e-6);
        }
        // one thread per element group, no grid-stride loop (the shape of the guide's float4 copy)
        {
            const int grid = (int)((n + 255) / 256);
            double t = time_us([&] { hipLaunchKernelGGL(copy_kernel<1>, dim3(grid), dim3(256), 0, 0, a, b, n); });
            printf("%-8ld copy one f4 per thread        %10.1f %10.2f\n", mb, t, 2.0 * bytes / t * 1e-6);
            t = time_us([&] { hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0); });
            printf("%-8ld hipMemcpyAsync D2D            %10.1f %10.2f\n", mb, t, 2.0 * bytes / t * 1e-6);
        }
        hipFree(a); hipFree(b);
    }
    return 0;
}
