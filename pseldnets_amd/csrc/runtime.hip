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
