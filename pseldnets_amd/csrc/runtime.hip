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

// ---- routing knobs (common.h): one pass over the environment per process, then array reads ----
#include <stdlib.h>
namespace {
struct KnobState { bool set[KNOB_COUNT]; int val[KNOB_COUNT]; };
const char* const g_knob_names[KNOB_COUNT] = {
#define PSELD_KNOB_NAME(n) #n,
    PSELD_KNOB_LIST(PSELD_KNOB_NAME)
#undef PSELD_KNOB_NAME
};
KnobState& knobs() {
    static KnobState st = [] {
        KnobState k;
        for (int i = 0; i < KNOB_COUNT; ++i) {
            char name[64];
            snprintf(name, sizeof name, "PSELD_%s", g_knob_names[i]);
            const char* e = getenv(name);
            k.set[i] = e != nullptr && e[0] != 0;
            k.val[i] = k.set[i] ? atoi(e) : 0;
        }
        return k;
    }();
    return st;
}
int knob_index(const char* name) {
    if (!name) return -1;
    if (strncmp(name, "PSELD_", 6) == 0) name += 6;
    for (int i = 0; i < KNOB_COUNT; ++i) if (strcmp(name, g_knob_names[i]) == 0) return i;
    return -1;
}
}  // namespace
int pseld_knob(PseldKnob k, int dflt) { const KnobState& st = knobs(); return st.set[k] ? st.val[k] : dflt; }
bool pseld_knob_is_set(PseldKnob k) { return knobs().set[k]; }
extern "C" int pseld_set_knob(const char* name, int value) {
    const int i = knob_index(name);
    PSELD_CHECK_ARG(i >= 0, "set_knob: unknown knob '%s'", name ? name : "(null)");
    KnobState& st = knobs();
    st.set[i] = true; st.val[i] = value;
    return PSELD_OK;
}
extern "C" int pseld_unset_knob(const char* name) {
    const int i = knob_index(name);
    PSELD_CHECK_ARG(i >= 0, "unset_knob: unknown knob '%s'", name ? name : "(null)");
    knobs().set[i] = false;
    return PSELD_OK;
}

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
