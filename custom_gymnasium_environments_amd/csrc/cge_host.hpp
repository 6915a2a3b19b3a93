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
    uint8_t *done_out = nullptr;   // step(): terminated | truncated per env, registered by <env>_done_mask (caller-owned device buffer, nullable)
    // <env>_rollout_final_obs: terminal observations of SAME_STEP rollouts, compacted per SEGMENT of consecutive envs (one wave's envs)
    void *fin_rows = nullptr;      // [n_segments * fin_cap, *obs_shape]
    int64_t *fin_index = nullptr;  // [n_segments * fin_cap]: step-in-call * n_envs + env
    int64_t fin_cap = 0;           // rows per segment
    int32_t *fin_count = nullptr;  // [n_segments]: rows the last rollout delivered for the segment (may exceed fin_cap: the surplus was dropped)

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

// shared body of cge_<env>_rollout_final_obs
template <class H>
int register_final_obs(H *h, void *rows, int64_t *index, int64_t seg_capacity, int32_t *count, const char *who) {
    if (!h) return CGE_ERR_INVALID_ARG;
    if ((rows || index || count) && (!rows || !index || !count || seg_capacity <= 0)) return h->fail(CGE_ERR_INVALID_ARG, who);
    h->fin_rows = rows; h->fin_index = index; h->fin_cap = rows ? seg_capacity : 0; h->fin_count = count;
    return CGE_OK;
}

// the two entry points every env type exports for it (inside extern "C"); SEG = envs per segment = envs one wave steps
#define CGE_DEFINE_FINAL_OBS(ENV, ROWTYPE, SEG)                                                                                          \
    int cge_##ENV##_rollout_final_obs(cge_##ENV *h, ROWTYPE *rows_out, int64_t *index_out, int64_t seg_capacity, int32_t *count_out) {   \
        return cge::register_final_obs(h, rows_out, index_out, seg_capacity, count_out,                                                  \
                                       "cge_" #ENV "_rollout_final_obs: rows, index and count go together (all NULL unregisters)");      \
    }                                                                                                                                    \
    int64_t cge_##ENV##_final_obs_segment(const cge_##ENV *h) { return h ? (SEG) : 0; }

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

// device stream (block, cursor, ready mark) -> CPython layout (624 words of ONE generation + index of the next unconsumed word).
// Words [pos, pretw) are twisted already; the rest of the generation is twisted here.  A ready mark beyond 624 means the first
// chunk of the NEXT generation has been twisted in place as well (cge_device.hpp: mt_make_ready): those words are taken back
// to the current generation first — the twist is invertible word by word: new[k] ^ cur[k+397] = g(y) with y = (cur[k] & 0x80000000)
// | (cur[k+1] & 0x7fffffff), and g(y) = (y >> 1) ^ (y & 1 ? 0x9908b0df : 0) gives y back (bit 31 of g(y) is y's bit 0).  What cannot
// be recovered, the low 31 bits of cur[0], no future output depends on.
// old0: the saved word 0 of the current generation (its dead low bits), or nullptr: then they read as zero.
inline void mt_export_cpython(const uint32_t *w, uint32_t pos, uint32_t pretw, uint32_t *omt, int32_t *idx, const uint32_t *old0 = nullptr) {
    memcpy(omt, w, 624 * 4);
    // a cursor that has just wrapped (pos 0) with chunks of the new generation already twisted: CPython regenerates lazily, its
    // state at this point is index 624 over the OLD generation — the same un-twist, seen from the end of that generation
    if (pos == 0u && pretw > 0u && pretw < 624u) { pos = 624u; pretw += 624u; }
    if (pretw > 624u) {
        const uint32_t ahead = pretw - 624u;         // words [0, ahead) belong to the next generation
        std::vector<uint32_t> y(ahead);
        for (uint32_t k = 0; k < ahead; ++k) {
            const uint32_t g = omt[k] ^ omt[k + 397];          // k + 397 < 624: a word of the current generation
            const uint32_t odd = g >> 31;
            y[k] = (((g ^ (odd ? 0x9908b0dfu : 0u)) << 1) | odd);
        }
        for (uint32_t k = 0; k < ahead; ++k) {
            const uint32_t upper = y[k] & 0x80000000u, lower = k ? (y[k - 1] & 0x7fffffffu) : (old0 ? *old0 & 0x7fffffffu : 0u);
            omt[k] = upper | lower;
        }
        omt[ahead] = (omt[ahead] & 0x80000000u) | (y[ahead - 1] & 0x7fffffffu);   // (its low bits were never changed: a consistency no-op)
        pretw = 624;
    }
    if (pretw >= 624u) { *idx = (int32_t)pos; return; }
    if (pos == 0 && pretw == 0) { *idx = 624; return; }
    for (uint32_t k = pretw > pos ? pretw : pos; k < 624u; ++k) {
        const uint32_t k1 = k + 1 == 624u ? 0 : k + 1, km = k + 397 >= 624u ? k + 397 - 624 : k + 397;
        const uint32_t t = (omt[k] & 0x80000000u) | (omt[k1] & 0x7fffffffu);
        omt[k] = omt[km] ^ (t >> 1) ^ ((t & 1u) ? 0x9908b0dfu : 0u);
    }
    *idx = (int32_t)pos;
}


// Seeds n MT19937 stream blocks (cge_device.hpp layout, `stride_words` apart starting at `mt`).
//   kind 0: CPython random.seed(s)  = init_by_array(32-bit limbs of s)
//   kind 1: NumPy legacy np.random.seed(s) = init_genrand((uint32)s)
// s = seeds[i] if seeds != nullptr (device pointer) else base_seed + env0 + i.
hipError_t launch_mt_seed(uint32_t *mt, int64_t stride_words, int64_t n, const uint64_t *seeds, uint64_t base_seed,
                          int64_t env0, int kind, hipStream_t stream);

}  // namespace cge
