// Library-wide state that is not a kernel: the last-error string and build/device identification.
#include "common.h"
#include <stdarg.h>

static thread_local char g_err[512] = "";

void pseld_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* pseld_last_error(void) { return g_err; }

extern "C" int pseld_abi_version(void) { return 1; }

// Fills name (<= n bytes) with the gcnArchName of the current device; returns the CU count or <0.
extern "C" int pseld_device_info(char* name, int n) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { pseld_set_error("device_info: no HIP device"); return PSELD_ERR_HIP; }
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, dev) != hipSuccess) { pseld_set_error("device_info: query failed"); return PSELD_ERR_HIP; }
    if (name && n > 0) { strncpy(name, p.gcnArchName, n - 1); name[n - 1] = 0; }
    return p.multiProcessorCount;
}

// Stage markers for the profiling passes (bench.py, PSELD_STAGE_MARKERS=1): an empty kernel whose SYMBOL carries the tag, launched where the
// step moves from one part of the network to the next. rocprofv3's kernel trace / PMC tables list dispatches in order, so the rows between two
// markers belong to one stage (tools/pmc_stages.py sums time and HBM bytes per stage from them). Never launched in a timed region.
template <int TAG> __global__ void stage_marker_kernel() {}
extern "C" int pseld_stage_marker(int tag, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    switch (tag) {
#define PSELD_MARK(T) case T: hipLaunchKernelGGL(stage_marker_kernel<T>, dim3(1), dim3(64), 0, s); break;
        PSELD_MARK(0) PSELD_MARK(1) PSELD_MARK(2) PSELD_MARK(3) PSELD_MARK(4) PSELD_MARK(5) PSELD_MARK(6) PSELD_MARK(7)
        PSELD_MARK(8) PSELD_MARK(9) PSELD_MARK(10) PSELD_MARK(11) PSELD_MARK(12) PSELD_MARK(13) PSELD_MARK(14) PSELD_MARK(15)
#undef PSELD_MARK
        default: pseld_set_error("stage_marker: tag %d out of range", tag); return PSELD_ERR_BAD_ARG;
    }
    PSELD_LAUNCH_CHECK("stage_marker");
    return PSELD_OK;
}
