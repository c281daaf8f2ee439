// Weight gradients without HBM slabs? ("measure, not paper".)
// gemm8w_kernel: 256 workgroups = output tiles x token splits; each ends by writing one fp32 slab tile of 256 x 192 (196 608 B), and a second
// launch (reduce_slabs) sums the `splits` slabs of every tile. The proposal: every workgroup adds its tile into the dW itself with
// global_atomic_add_f32 (no slab, no second launch; summation order not fixed). This program times exactly that tail on the production
// geometry, with the accumulator tile in registers as the kernel has it (96 fp32 per lane, 512 threads):
//   (a) plain 16-byte stores of the tile into the workgroup's slab + the separate whole-chip reduction launch (what runs today);
//   (b) no-return global_atomic_add_f32, one dword per lane per instruction, 256 contiguous bytes per wave-instruction (the full-rate
//       shape of MI355X_MICROARCH.md, "Global float atomics"), all `splits` workgroups of a tile into the same 196 608 B;
//   (c) the same atomics preceded by a zero-fill of dW (the fill the slab path does not need).
// Shapes: stage-2 fc1 (12 tiles x 21 splits), stage-2 proj (4 x 64), stage-1 fc1 (4 x 64 of 256 x 192), stage-3 fc1 (48 x 5).
//   hipcc --offload-arch=gfx950 -O3 -o tools/experiments/wgrad_atomic tools/experiments/wgrad_atomic.hip && tools/experiments/wgrad_atomic
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int TILE_F = 256 * 192, TILE_F4 = TILE_F / 4;

__device__ __forceinline__ void make_tile(f4 (&acc)[24], int z) {        // 96 fp32 per lane: stands for the MFMA accumulators
#pragma unroll
    for (int i = 0; i < 24; ++i) acc[i] = f4{1.f + z, 2.f + i, 3.f, 4.f};
#pragma unroll
    for (int i = 0; i < 24; ++i) asm volatile("" : "+v"(acc[i]));
}

__global__ __launch_bounds__(512, 2) void tail_slab(f4* slabs, int tiles, int splits) {
    const int L = blockIdx.x;
    if (L >= tiles * splits) return;
    const int z = L / tiles, t = L - z * tiles;
    f4 acc[24];
    make_tile(acc, z);
    f4* mine = slabs + ((long)z * tiles + t) * TILE_F4;
#pragma unroll
    for (int i = 0; i < 24; ++i) mine[i * 512 + threadIdx.x] = acc[i];
}
__global__ __launch_bounds__(256) void reduce_all(const f4* slabs, f4* out, long n4, int splits) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        f4 a = {0.f, 0.f, 0.f, 0.f};
        for (int s = 0; s < splits; ++s) a += slabs[(long)s * n4 + i];
        out[i] = a;
    }
}
__global__ __launch_bounds__(512, 2) void tail_atomic(float* dw, int tiles, int splits) {
    const int L = blockIdx.x;
    if (L >= tiles * splits) return;
    const int z = L / tiles, t = L - z * tiles;
    f4 acc[24];
    make_tile(acc, z);
    float* dst = dw + (long)t * TILE_F;
    // one dword per lane per instruction, a wave covers 256 contiguous bytes: element (i, c) of the lane goes to [(i * 4 + c) * 512 + tid]
#pragma unroll
    for (int i = 0; i < 24; ++i)
#pragma unroll
        for (int c = 0; c < 4; ++c)
            __hip_atomic_fetch_add(dst + (i * 4 + c) * 512 + threadIdx.x, acc[i][c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__global__ __launch_bounds__(256) void fill0(f4* p, long n4) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) p[i] = f4{0.f, 0.f, 0.f, 0.f};
}

int main() {
    struct Shape { const char* name; int tiles, splits; } shapes[] = {{"stage-2 fc1 dW[1536,384]", 12, 21}, {"stage-2 proj dW[384,384]", 4, 64},
                                                                      {"stage-1 fc1 dW[768,192]", 3, 85}, {"stage-3 fc1 dW[3072,768]", 48, 5}};
    f4 *slabs, *out;
    hipMalloc(&slabs, (size_t)256 * TILE_F4 * sizeof(f4));
    hipMalloc(&out, (size_t)48 * TILE_F4 * sizeof(f4));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (auto& s : shapes) {
        const int nwg = s.tiles * s.splits;
        const long n4 = (long)s.tiles * TILE_F4;
        auto time = [&](auto fn) {
            std::vector<float> ms;
            for (int r = 0; r < 25; ++r) {
                hipEventRecord(e0); fn(); hipEventRecord(e1); hipEventSynchronize(e1);
                float t; hipEventElapsedTime(&t, e0, e1); ms.push_back(t * 1e3f);
            }
            std::sort(ms.begin(), ms.end());
            return ms[ms.size() / 2];
        };
        const float a1 = time([&] { hipLaunchKernelGGL(tail_slab, dim3(nwg), dim3(512), 0, 0, slabs, s.tiles, s.splits); });
        const float a2 = time([&] { hipLaunchKernelGGL(tail_slab, dim3(nwg), dim3(512), 0, 0, slabs, s.tiles, s.splits);
                                    hipLaunchKernelGGL(reduce_all, dim3(1024), dim3(256), 0, 0, (const f4*)slabs, out, n4, s.splits); });
        const float b1 = time([&] { hipLaunchKernelGGL(tail_atomic, dim3(nwg), dim3(512), 0, 0, (float*)out, s.tiles, s.splits); });
        const float c1 = time([&] { hipLaunchKernelGGL(fill0, dim3(512), dim3(256), 0, 0, out, n4);
                                    hipLaunchKernelGGL(tail_atomic, dim3(nwg), dim3(512), 0, 0, (float*)out, s.tiles, s.splits); });
        // correctness of the atomic path (sum over splits of 1 + z in element 0)
        hipLaunchKernelGGL(fill0, dim3(512), dim3(256), 0, 0, out, n4);
        hipLaunchKernelGGL(tail_atomic, dim3(nwg), dim3(512), 0, 0, (float*)out, s.tiles, s.splits);
        float v0; hipMemcpy(&v0, out, 4, hipMemcpyDeviceToHost);
        const float want = s.splits * (s.splits + 1) / 2.f;
        const double mb = (double)nwg * TILE_F * 4 / 1e6;
        printf("%-26s %3d tiles x %2d splits = %3d workgroups, %6.1f MB of tiles: slab stores %6.1f us (%.2f TB/s) | + reduction launch %6.1f us | atomic adds %6.1f us "
               "(%.2f TB/s of added bytes) | zero-fill + atomic adds %6.1f us | check %s\n", s.name, s.tiles, s.splits, nwg, mb, a1, mb / a1, a2, b1, mb / b1, c1,
               v0 == want ? "ok" : "WRONG");
    }
    return 0;
}
