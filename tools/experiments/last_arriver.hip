// What would an in-kernel "last arriver" reduction of the weight-gradient slabs cost? (VERDICT r4 task 3.)
// gemm8w_kernel: 256 workgroups = tiles x token splits, each leaves one fp32 slab tile of 256 x 192 (196 608 B); a second launch
// (reduce_slabs) sums the `splits` slabs of every tile with the whole chip. The proposal: per output tile an atomic ticket, the workgroup
// that draws the last ticket sums the tile's slabs in fixed order - no second launch, no spinning. This program times exactly that
// tail: (a) 256 workgroups write their slab tile, fence, take a ticket; the last arriver of each tile reads all `splits` slab tiles of
// the tile (16-byte loads, 512 threads, 8 loads in flight per thread) and writes the sum; against (b) the same writes followed by a
// separate whole-chip reduction launch. Shapes: stage-2 fc1 (12 tiles x 21 splits), stage-2 proj (4 x 64), stage-3 fc1 (48 x 5).
//   hipcc --offload-arch=gfx950 -O3 -o tools/experiments/last_arriver tools/experiments/last_arriver.hip && tools/experiments/last_arriver
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int TILE_F4 = 256 * 192 / 4;      // one slab tile in 16-byte pieces

// (a) write + ticket + last-arriver reduction. slabs: [split][tile][TILE_F4]
__global__ __launch_bounds__(512) void write_and_last_arriver(f4* slabs, f4* out, unsigned* tickets, int tiles, int splits, int do_reduce) {
    const int L = blockIdx.x;
    if (L >= tiles * splits) return;
    const int z = L / tiles, t = L - z * tiles;
    f4* mine = slabs + ((long)z * tiles + t) * TILE_F4;
    const f4 v = {1.f + z, 2.f, 3.f, 4.f};
    for (int i = threadIdx.x; i < TILE_F4; i += 512) mine[i] = v;
    if (!do_reduce) return;
    __threadfence();                       // release: the slab is visible device-wide before the ticket is drawn
    __syncthreads();
    __shared__ unsigned ticket;
    if (threadIdx.x == 0) ticket = atomicAdd(&tickets[t], 1u);
    __syncthreads();
    if (ticket != (unsigned)(splits - 1)) return;
    __threadfence();                       // acquire side (the other workgroups' slabs)
    for (int i = threadIdx.x; i < TILE_F4; i += 512 * 2) {
        f4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0;
        const bool two = i + 512 < TILE_F4;
        for (int s = 0; s < splits; s += 4) {        // fixed split order; 8 loads in flight per thread
            f4 x[4], y[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int ss = min(s + u, splits - 1);
                const f4* p = slabs + ((long)ss * tiles + t) * TILE_F4;
                x[u] = __builtin_nontemporal_load(p + i);
                y[u] = two ? __builtin_nontemporal_load(p + i + 512) : f4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) if (s + u < splits) { a0 += x[u]; a1 += y[u]; }
        }
        out[(long)t * TILE_F4 + i] = a0;
        if (two) out[(long)t * TILE_F4 + i + 512] = a1;
    }
    if (threadIdx.x == 0) tickets[t] = 0;
}

// (b) the separate whole-chip reduction (the shape of reduce_slabs_vec4_kernel)
__global__ __launch_bounds__(256) void reduce_all(const f4* slabs, f4* out, long n4, int splits) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        f4 a = {0.f, 0.f, 0.f, 0.f};
        for (int s = 0; s < splits; ++s) a += slabs[(long)s * n4 + i];
        out[i] = a;
    }
}

template <class F>
static double time_us(F&& fn, int reps = 30) {
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    fn(); fn();
    (void)hipDeviceSynchronize();
    std::vector<float> t;
    for (int r = 0; r < reps; ++r) {
        (void)hipEventRecord(a, 0); fn(); (void)hipEventRecord(b, 0); (void)hipEventSynchronize(b);
        float ms; (void)hipEventElapsedTime(&ms, a, b); t.push_back(ms * 1e3f);
    }
    std::sort(t.begin(), t.end());
    return t[t.size() / 2];
}

int main() {
    const int shapes[][2] = {{12, 21}, {4, 64}, {16, 16}, {48, 5}};
    f4 *slabs, *out; unsigned* tickets;
    (void)hipMalloc(&slabs, (long)256 * TILE_F4 * 16); (void)hipMalloc(&out, (long)64 * TILE_F4 * 16); (void)hipMalloc(&tickets, 256 * 4);
    (void)hipMemset(tickets, 0, 256 * 4);
    printf("%-18s %12s %12s %12s %12s\n", "tiles x splits", "write only", "last arriver", "write+reduce", "(us)");
    for (auto& sh : shapes) {
        const int tiles = sh[0], splits = sh[1];
        const long n4 = (long)tiles * TILE_F4;
        const double w = time_us([&] { hipLaunchKernelGGL(write_and_last_arriver, dim3(256), dim3(512), 0, 0, slabs, out, tickets, tiles, splits, 0); });
        const double la = time_us([&] { hipLaunchKernelGGL(write_and_last_arriver, dim3(256), dim3(512), 0, 0, slabs, out, tickets, tiles, splits, 1); });
        const double wr = time_us([&] {
            hipLaunchKernelGGL(write_and_last_arriver, dim3(256), dim3(512), 0, 0, slabs, out, tickets, tiles, splits, 0);
            hipLaunchKernelGGL(reduce_all, dim3(1024), dim3(256), 0, 0, slabs, out, n4, splits);
        });
        printf("%3d x %-12d %12.1f %12.1f %12.1f\n", tiles, splits, w, la, wr);
    }
    return 0;
}
