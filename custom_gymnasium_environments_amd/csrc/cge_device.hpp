// cge_device.hpp — device-side building blocks shared by every env kernel (gfx950 only).
//
//  * counter-hash action source (same definition as tests/golden/gen/common.py)
//  * MT19937 streams that reproduce CPython `random` / NumPy-legacy draws bit for bit, stored
//    array-of-structs per env (one 2560-byte block per stream) and advanced ONE word at a time
//    (incremental twist) so a lane never stalls its wave on a 624-word regeneration
//  * obs-row staging: each lane builds its env's observation row in LDS, the workgroup then
//    streams the tile to HBM as 16-byte-per-lane coalesced stores (gymnasium's (N, *obs_shape)
//    row-major layout makes per-lane row stores 100..1044-byte strided otherwise)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cge {

// ------------------------------------------------------------------ action hash
__host__ __device__ inline uint64_t mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__host__ __device__ inline uint64_t hash_env_key(uint64_t a_seed, uint64_t env) {
    return mix64(a_seed + env * 0x9E3779B97F4A7C15ull);
}
__host__ __device__ inline uint32_t hash_action_from_key(uint64_t key, uint64_t t, uint32_t n, uint32_t j) {
    uint64_t u = mix64(key + t * 0xD1342543DE82EF95ull + j);
    return (uint32_t)(((u >> 32) * (uint64_t)n) >> 32);
}

// ------------------------------------------------------------------ MT19937 (family P / L)
constexpr int MT_N = 624;
constexpr int MT_M = 397;
constexpr int MT_STRIDE = 640;  // words per stream block: mt[624] + pad (=2560 B, 20 x 128-B lines)
// The stream cursor lives in the ENV's state record, not in the block, so a draw never starts with a
// dependent "load the cursor" round trip:
//   pos    next word index, 0..623
//   pretw  words in [pos, pretw) are already twisted (a CPython state imported by set_state); 0 or 624
// Words are twisted one at a time, in place, exactly when they are consumed (incremental form of
// the reference generator's 624-word batch regeneration; identical output sequence).

__device__ __forceinline__ uint32_t mt_temper(uint32_t y) {
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}
__device__ __forceinline__ uint32_t mt_twist(uint32_t a, uint32_t b, uint32_t c) {
    const uint32_t t = (a & 0x80000000u) | (b & 0x7fffffffu);
    return c ^ (t >> 1) ^ ((t & 1u) ? 0x9908b0dfu : 0u);
}

// Serial stream: one memory round trip per draw (3 independent loads + 1 store).  For rare/long paths.
struct MtStream {
    uint32_t *w;
    uint32_t pos, pretw;

    __device__ __forceinline__ MtStream(uint32_t *block, uint32_t pos_, uint32_t pretw_) : w(block), pos(pos_), pretw(pretw_) {}

    __device__ __forceinline__ uint32_t next() {
        uint32_t p = pos, y;
        if (p < pretw) {
            y = w[p];
        } else {
            const uint32_t p1 = p + 1 == MT_N ? 0 : p + 1;
            const uint32_t pm = p + MT_M >= MT_N ? p + MT_M - MT_N : p + MT_M;
            y = mt_twist(w[p], w[p1], w[pm]);
            w[p] = y;
        }
        ++p;
        if (p == MT_N) { p = 0; pretw = 0; }
        pos = p;
        return mt_temper(y);
    }
    // CPython Random._randbelow_with_getrandbits(n), k = n.bit_length()
    __device__ __forceinline__ uint32_t randbelow(uint32_t n, int kbits) {
        uint32_t r = next() >> (32 - kbits);
        while (r >= n) r = next() >> (32 - kbits);
        return r;
    }
    // CPython random.random() == NumPy legacy random_sample(): 53-bit double from two words
    __device__ __forceinline__ double random53() {
        const uint32_t a = next() >> 5, b = next() >> 6;
        return (a * 67108864.0 + b) / 9007199254740992.0;
    }
    __device__ __forceinline__ double uniform(double lo, double hi) { return lo + (hi - lo) * random53(); }
};

// Windowed stream: fetches everything the next W draws can need in ONE round trip (2W+1 independent
// 4-byte loads inside the env's own block: w[pos..pos+W] and w[pos+397..pos+397+W-1], indices mod 624),
// hands out the twisted words from registers, then writes back only the words actually consumed.
// No entry of the window depends on another (397 is not congruent to any |j'-j| <= W mod 624).
template <int W>
struct MtWindow {
    uint32_t a[W + 1];
    uint32_t c[W];

    __device__ __forceinline__ void load(const uint32_t *__restrict__ blk, uint32_t pos) {
#pragma unroll
        for (int j = 0; j <= W; ++j) {
            uint32_t k = pos + j;
            k -= k >= (uint32_t)MT_N ? MT_N : 0;
            a[j] = blk[k];
        }
#pragma unroll
        for (int j = 0; j < W; ++j) {
            uint32_t k = pos + MT_M + j;
            k -= k >= (uint32_t)MT_N ? MT_N : 0;
            c[j] = blk[k];
        }
    }
    // untempered word number j (static index) of the stream starting at (pos, pretw)
    __device__ __forceinline__ uint32_t twisted(int j, uint32_t pos, uint32_t pretw) const {
        const uint32_t y = mt_twist(a[j], a[j + 1], c[j]);
        return (pos + j < pretw) ? a[j] : y;
    }
    __device__ __forceinline__ uint32_t draw(int j, uint32_t pos, uint32_t pretw) const { return mt_temper(twisted(j, pos, pretw)); }
    // persist the first `used` words and advance the cursor
    __device__ __forceinline__ void commit(uint32_t *__restrict__ blk, uint32_t &pos, uint32_t &pretw, uint32_t used) const {
#pragma unroll
        for (int j = 0; j < W; ++j) {
            uint32_t k = pos + j;
            if ((uint32_t)j < used && k >= pretw) {
                const uint32_t y = mt_twist(a[j], a[j + 1], c[j]);
                k -= k >= (uint32_t)MT_N ? MT_N : 0;
                blk[k] = y;
            }
        }
        uint32_t p = pos + used;
        if (p >= (uint32_t)MT_N) { p -= MT_N; pretw = 0; }
        pos = p;
    }
};

// Draw queue for envs whose draw COUNT per step is data dependent (traffic: lights, spawn, routes): the
// window's twisted words are parked in the lane's own LDS row, where a per-lane cursor can index them
// natively (a runtime-indexed register array would go to scratch).  `row` must hold W dwords; give rows an
// odd stride so the 64 lanes' rows start in different banks.  flush() writes back only consumed words.
template <int W>
struct LdsDraws {
    uint32_t *row;
    uint32_t *blk;
    uint32_t pos, pretw, cur;
    bool filled;

    __device__ __forceinline__ LdsDraws(uint32_t *lds_row, uint32_t *block, uint32_t pos_, uint32_t pretw_)
        : row(lds_row), blk(block), pos(pos_), pretw(pretw_), cur(0), filled(false) {}
    __device__ __forceinline__ void fill() {
        MtWindow<W> w;
        w.load(blk, pos);
#pragma unroll
        for (int j = 0; j < W; ++j) row[j] = w.twisted(j, pos, pretw);
        cur = 0;
        filled = true;
    }
    __device__ __forceinline__ void flush() {
        if (!filled) return;
        for (uint32_t j = 0; j < cur; ++j) {
            uint32_t k = pos + j;
            if (k >= pretw) {
                k -= k >= (uint32_t)MT_N ? MT_N : 0;
                blk[k] = row[j];
            }
        }
        uint32_t p = pos + cur;
        if (p >= (uint32_t)MT_N) { p -= MT_N; pretw = 0; }
        pos = p;
        cur = 0;
        filled = false;
    }
    __device__ __forceinline__ uint32_t next() {
        if (!filled) fill();
        else if (cur == (uint32_t)W) { flush(); fill(); }
        return mt_temper(row[cur++]);
    }
    __device__ __forceinline__ uint32_t randbelow(uint32_t n, int kbits) {   // CPython _randbelow_with_getrandbits
        uint32_t r = next() >> (32 - kbits);
        while (r >= n) r = next() >> (32 - kbits);
        return r;
    }
    __device__ __forceinline__ double random53() {
        const uint32_t a = next() >> 5, b = next() >> 6;
        return (a * 67108864.0 + b) / 9007199254740992.0;
    }
};

// ------------------------------------------------------------------ small register arrays
// Runtime-indexed register arrays go to scratch on hipcc; these helpers keep every index static
// (fully unrolled select chains) so the env state stays in VGPRs.
template <int W>
__device__ __forceinline__ uint32_t sel(const uint32_t (&a)[W], uint32_t idx) {
    // mask form on purpose: a ternary chain gets folded by LLVM into "select the ADDRESS, then load",
    // which pins the array in scratch memory
    uint32_t r = 0;
#pragma unroll
    for (int k = 0; k < W; ++k) r |= a[k] & (0u - (uint32_t)(idx == (uint32_t)k));
    return r;
}
template <int W>
__device__ __forceinline__ void or_word(uint32_t (&a)[W], uint32_t idx, uint32_t m) {
#pragma unroll
    for (int k = 0; k < W; ++k) a[k] |= (idx == (uint32_t)k) ? m : 0u;
}
template <int W>
__device__ __forceinline__ void andnot_word(uint32_t (&a)[W], uint32_t idx, uint32_t m) {
#pragma unroll
    for (int k = 0; k < W; ++k) a[k] &= (idx == (uint32_t)k) ? ~m : 0xffffffffu;
}

// Workgroup barrier that orders LDS traffic ONLY.  __syncthreads() also emits s_waitcnt vmcnt(0),
// which would stall every wave until its global stores (the previous obs tile) are acknowledged and
// until prefetched global loads land — exactly the latency the fused rollout is built to hide.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// ------------------------------------------------------------------ obs tile -> HBM
// Streams `bytes_valid` bytes (multiple of 4) of an LDS tile to `dst` with the widest stores the
// destination alignment allows.  Must be called by every thread of the block after a barrier.
template <int BLOCK>
__device__ __forceinline__ void store_tile(const uint32_t *tile, int8_t *dst, uint32_t bytes_valid) {
    const uint32_t tid = threadIdx.x;
    if ((reinterpret_cast<uintptr_t>(dst) & 15u) == 0) {
        const uint32_t nvec = bytes_valid >> 4;
        const uint4 *t4 = reinterpret_cast<const uint4 *>(tile);
        uint4 *d4 = reinterpret_cast<uint4 *>(dst);
        for (uint32_t q = tid; q < nvec; q += BLOCK) d4[q] = t4[q];
        const uint32_t rem0 = nvec << 2, ndw = bytes_valid >> 2;
        uint32_t *d1 = reinterpret_cast<uint32_t *>(dst);
        for (uint32_t q = rem0 + tid; q < ndw; q += BLOCK) d1[q] = tile[q];
    } else {
        const uint32_t ndw = bytes_valid >> 2;
        uint32_t *d1 = reinterpret_cast<uint32_t *>(dst);
        for (uint32_t q = tid; q < ndw; q += BLOCK) d1[q] = tile[q];
    }
}

}  // namespace cge
