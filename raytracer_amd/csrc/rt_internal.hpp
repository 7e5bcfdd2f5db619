// Shared between the device-management and kernel translation units.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <mutex>
#include <vector>

#include "rt_hip.h"

// Fatal-on-error convention of the host API: message, then exit(1).
#define RT_HIP_CHECK(expr)                                                                        \
    do {                                                                                          \
        hipError_t rt_err_ = (expr);                                                              \
        if (rt_err_ != hipSuccess) {                                                              \
            fprintf(stderr, "ERROR: %s failed: %s (line %d, file %s)\n", #expr,                   \
                    hipGetErrorString(rt_err_), __LINE__, __FILE__);                              \
            fflush(stderr);                                                                       \
            exit(1);                                                                              \
        }                                                                                         \
    } while (0)

struct rt_timing_slot {
    std::vector<hipEvent_t> start, stop;   // pairs still to be read back
    uint64_t launches = 0;
    double total_ms = 0.0;
};

struct rt_device {
    int ordinal = 0;
    hipStream_t stream = nullptr;              // the stream launches currently go to (one of streams[])
    hipStream_t streams[RT_MAX_STREAMS] = {};
    int num_streams = 0, current = 0;
    hipDeviceProp_t props;
    int timing = 0;
    rt_timing_slot slots[RT_K_COUNT];
    std::vector<hipEvent_t> free_events;
    uint32_t *counters = nullptr;   // device words the kernels use as work counters: one 64-byte line per stream
    // This GPU's copy of the kernels' file-scope scratch array (work counters of the entry point that allocates
    // nothing), resolved once by rt_kernels_setup_device, and that entry point's launch geometry.
    uint32_t *static_counters = nullptr;
    int noalloc_slot = -1;          // which block of RT_MAX_STREAMS counter lines of that array is this device's (-1: none left)
    int noalloc_blocks_per_cu = 0;
    std::map<const void *, int> blocks_per_cu;   // resident one-wave blocks per CU of each persistent kernel on this device
};

// Per-device set-up of the kernel translation unit (called by rt_device_open with the device current).
void rt_kernels_setup_device(rt_device *dev);
// Gives back what set-up took from process-wide tables (called by rt_device_close).
void rt_kernels_release_device(rt_device *dev);

// Brackets one launch with events when timing is on.
struct rt_launch_timer {
    rt_device *dev;
    int id;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    rt_launch_timer(rt_device *d, int kernel_id);
    ~rt_launch_timer();
};

void rt_timing_fold(rt_device *dev, int kernel_id);
