// cge_host.hpp — host-side plumbing shared by the per-env C-ABI translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstdio>
#include <string>

#include "../../include/cge_amd.h"

namespace cge {

struct HandleBase {
    int device = 0;
    int64_t n = 0;
    int64_t env0 = 0;
    std::string last_error;
    size_t device_bytes = 0;

    int fail(int status, const char *what, hipError_t e = hipSuccess) {
        char buf[512];
        if (e != hipSuccess)
            snprintf(buf, sizeof buf, "%s: %s (%s)", what, hipGetErrorName(e), hipGetErrorString(e));
        else
            snprintf(buf, sizeof buf, "%s", what);
        last_error = buf;
        return status;
    }
};

// Makes `device` current for the scope of one ABI call (torch may have another one selected).
struct DeviceGuard {
    int prev = -1;
    bool switched = false;
    explicit DeviceGuard(int device) {
        if (hipGetDevice(&prev) == hipSuccess && prev != device) {
            switched = hipSetDevice(device) == hipSuccess;
        }
    }
    ~DeviceGuard() {
        if (switched) (void)hipSetDevice(prev);
    }
};

#define CGE_TRY(h, expr)                                                   \
    do {                                                                   \
        hipError_t _e = (expr);                                            \
        if (_e != hipSuccess) return (h)->fail(CGE_ERR_HIP, #expr, _e);    \
    } while (0)

inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

// Seeds n MT19937 stream blocks (cge_device.hpp layout, `stride_words` apart starting at `mt`).
//   kind 0: CPython random.seed(s)  = init_by_array(32-bit limbs of s)
//   kind 1: NumPy legacy np.random.seed(s) = init_genrand((uint32)s)
// s = seeds[i] if seeds != nullptr (device pointer) else base_seed + env0 + i.
hipError_t launch_mt_seed(uint32_t *mt, int64_t stride_words, int64_t n, const uint64_t *seeds, uint64_t base_seed,
                          int64_t env0, int kind, hipStream_t stream);

}  // namespace cge
