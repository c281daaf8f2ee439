// Launch-latency experiment for launch-bound inner loops (the GRU recurrence: T dependent tiny kernels per layer).
// Compares, for a chain of N dependent kernels of a few microseconds each: (a) plain stream launches, (b) one hipGraph
// captured from the same launches and replayed. Build: hipcc --offload-arch=gfx950 -O3 graph_launch.hip -o graph_launch
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <chrono>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void step_kernel(const float* __restrict__ in, float* __restrict__ out, int n, int work) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float v = in[i];
    for (int k = 0; k < work; ++k) v = v * 1.0001f + 0.5f;
    out[i] = v;
}

int main() {
    const int n = 128 * 256, N = 250;
    float *a, *b;
    CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&b, n * 4));
    CK(hipMemset(a, 0, n * 4)); CK(hipMemset(b, 0, n * 4));
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int work : {1, 200, 1000}) {
        auto chain = [&](hipStream_t st) {
            for (int k = 0; k < N; ++k) hipLaunchKernelGGL(step_kernel, dim3(128), dim3(256), 0, st, (k & 1) ? b : a, (k & 1) ? a : b, n, work);
        };
        chain(s); CK(hipStreamSynchronize(s));
        float best_stream = 1e9f, best_graph = 1e9f;
        double host_stream = 0;
        for (int rep = 0; rep < 5; ++rep) {
            auto t0 = std::chrono::steady_clock::now();
            CK(hipEventRecord(e0, s)); chain(s); CK(hipEventRecord(e1, s));
            auto t1 = std::chrono::steady_clock::now();
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best_stream) { best_stream = ms; host_stream = std::chrono::duration<double, std::micro>(t1 - t0).count(); }
        }
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        chain(s);
        CK(hipStreamEndCapture(s, &g));
        auto ti0 = std::chrono::steady_clock::now();
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        auto ti1 = std::chrono::steady_clock::now();
        CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
        for (int rep = 0; rep < 5; ++rep) {
            CK(hipEventRecord(e0, s)); CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(e1, s));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best_graph) best_graph = ms;
        }
        printf("work %4d: stream %.2f us/kernel (host issue %.2f us/kernel), graph %.2f us/kernel, instantiate %.0f us for %d nodes\n", work,
               best_stream * 1e3 / N, host_stream / N, best_graph * 1e3 / N, std::chrono::duration<double, std::micro>(ti1 - ti0).count(), N);
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    return 0;
}
