// cge_host.hpp — host-side plumbing shared by the per-env C-ABI translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstdio>
#include <string>
#include <utility>
#include <vector>
#include <cstring>

#include "../../include/cge_amd.h"

namespace cge {

struct HandleBase {
    int device = 0;
    int64_t n = 0;
    int64_t env0 = 0;
    std::string last_error;
    std::string last_kernel;       // the kernel(s) the last step() / rollout() call launched, as rocprofv3 prints them (<env>_last_kernel)
    size_t device_bytes = 0;
    double *ep_ret = nullptr;      // episode-statistics outputs registered by <env>_episode_stats (caller-owned device buffers)
    int32_t *ep_len = nullptr;

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

// Whole-handle snapshots (checkpoint / resume) for the env types without a canonical per-env record: a 32-byte header
// {magic, n_envs, env tag, extra} followed by the handle's device arrays in their device layout.  Only valid for a handle
// created with the same n_envs and config.  H provides blobs() -> vector<pair<void*, size_t>>, snap_tag, snap_extra().
struct SnapHeader { uint64_t magic; int64_t n; uint32_t tag, extra; uint64_t reserved; };
// "CGESNAP3": 1 = round 1; 2 = round 2 (MT blocks grew mirror words 624..639, ep_return fields in the records) — blobs of an older
// layout are refused instead of being read with the wrong meaning
constexpr uint64_t SNAP_MAGIC = 0x3350414e53454743ull;
template <class H>
size_t snapshot_bytes(const H *h) {
    size_t t = sizeof(SnapHeader);
    for (const auto &b : h->blobs()) t += b.second;
    return t;
}
template <class H>
int snapshot_get(H *h, void *host, hipStream_t s) {
    if (!h || !host) return CGE_ERR_INVALID_ARG;
    DeviceGuard g(h->device);
    CGE_TRY(h, hipStreamSynchronize(s));
    SnapHeader hd{SNAP_MAGIC, h->n, H::snap_tag, h->snap_extra(), 0};
    memcpy(host, &hd, sizeof hd);
    char *dst = static_cast<char *>(host) + sizeof hd;
    for (const auto &b : h->blobs()) { CGE_TRY(h, hipMemcpy(dst, b.first, b.second, hipMemcpyDeviceToHost)); dst += b.second; }
    return CGE_OK;
}
template <class H>
int snapshot_set(H *h, const void *host, hipStream_t s) {
    if (!h || !host) return CGE_ERR_INVALID_ARG;
    DeviceGuard g(h->device);
    SnapHeader hd;
    memcpy(&hd, host, sizeof hd);
    if (hd.magic != SNAP_MAGIC || hd.n != h->n || hd.tag != H::snap_tag) return h->fail(CGE_ERR_INVALID_ARG, "snapshot_set: not a snapshot of this env type / batch size");
    CGE_TRY(h, hipStreamSynchronize(s));
    const char *src = static_cast<const char *>(host) + sizeof hd;
    for (const auto &b : h->blobs()) { CGE_TRY(h, hipMemcpy(b.first, src, b.second, hipMemcpyHostToDevice)); src += b.second; }
    h->set_snap_extra(hd.extra);
    return CGE_OK;
}

// Seeds n MT19937 stream blocks (cge_device.hpp layout, `stride_words` apart starting at `mt`).
//   kind 0: CPython random.seed(s)  = init_by_array(32-bit limbs of s)
//   kind 1: NumPy legacy np.random.seed(s) = init_genrand((uint32)s)
// s = seeds[i] if seeds != nullptr (device pointer) else base_seed + env0 + i.
hipError_t launch_mt_seed(uint32_t *mt, int64_t stride_words, int64_t n, const uint64_t *seeds, uint64_t base_seed,
                          int64_t env0, int kind, hipStream_t stream);

}  // namespace cge
