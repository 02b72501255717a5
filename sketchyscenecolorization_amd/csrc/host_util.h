// host_util.h -- host-side helpers shared by the launchers: one-time kernel setup and device facts, keyed by the CURRENT
// device (a process that drives several GPUs must set a kernel's dynamic-LDS limit on each of them, and plans its launches
// for the CU count of the device it launches on).
#pragma once
#include <hip/hip_runtime.h>
#include <stdlib.h>

// Developer A/B switches (forcing a fallback path, planner constants, tile shapes): honoured only under SSC_DEV_SWITCHES=1.
// They select paths the test suite does not cover in combination -- lab tools (scripts/), not supported configurations;
// DESIGN.md section 7 lists the supported switches.
static inline const char* ssc_dev_getenv(const char* name) {
    static int on = -1;
    if (on < 0) {
        const char* e = getenv("SSC_DEV_SWITCHES");
        on = (e != nullptr && e[0] == '1') ? 1 : 0;
    }
    return on ? getenv(name) : nullptr;
}

// true the first time this call site (its own `done` mask) runs on the current device
static inline bool ssc_first_on_device(unsigned long long* done) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    const unsigned long long bit = 1ull << (dev & 63);
    if (*done & bit) return false;
    *done |= bit;
    return true;
}

// hipFuncAttributeMaxDynamicSharedMemorySize once per device; the error code of the call that made it (0 afterwards)
static inline int ssc_set_max_lds(const void* fn, int bytes, unsigned long long* done) {
    if (!ssc_first_on_device(done)) return 0;
    return (int)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}

// compute units of the current device
static inline int ssc_num_cu() {
    static int n[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    int& v = n[dev & 63];
    if (v == 0) {
        hipDeviceProp_t prop;
        v = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    }
    return v;
}
