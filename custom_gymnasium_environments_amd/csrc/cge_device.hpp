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
constexpr int MT_STRIDE = 640;  // words per stream block: mt[624], pos, pretw, 14 pad (=2560 B, 20 x 128-B lines)
constexpr int MT_POS = 624;     // next word index, 0..623
constexpr int MT_PRETW = 625;   // words in [pos, pretw) are already twisted (imported CPython state); 0 or 624

// A stream handle kept in registers while an env draws.  `open` costs one 8-byte load, each
// draw 3 independent 4-byte loads + 1 store inside the env's own block, `close` one 8-byte store.
struct MtStream {
    uint32_t *w;
    uint32_t pos, pretw;

    __device__ __forceinline__ void open(uint32_t *block) {
        w = block;
        uint2 pp = *reinterpret_cast<const uint2 *>(block + MT_POS);
        pos = pp.x;
        pretw = pp.y;
    }
    __device__ __forceinline__ void close() { *reinterpret_cast<uint2 *>(w + MT_POS) = make_uint2(pos, pretw); }

    __device__ __forceinline__ uint32_t next() {
        uint32_t p = pos, y;
        if (p < pretw) {
            y = w[p];
        } else {
            uint32_t p1 = p + 1 == MT_N ? 0 : p + 1;
            uint32_t pm = p + MT_M >= MT_N ? p + MT_M - MT_N : p + MT_M;
            uint32_t a = w[p], b = w[p1], c = w[pm];
            uint32_t t = (a & 0x80000000u) | (b & 0x7fffffffu);
            y = c ^ (t >> 1) ^ ((t & 1u) ? 0x9908b0dfu : 0u);
            w[p] = y;
        }
        ++p;
        if (p == MT_N) { p = 0; pretw = 0; }
        pos = p;
        y ^= y >> 11;
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= y >> 18;
        return y;
    }
    // CPython Random._randbelow_with_getrandbits(n) with k = n.bit_length() known at compile time
    template <int KBITS>
    __device__ __forceinline__ uint32_t randbelow(uint32_t n) {
        uint32_t r = next() >> (32 - KBITS);
        while (r >= n) r = next() >> (32 - KBITS);
        return r;
    }
    __device__ __forceinline__ uint32_t randbelow_k(uint32_t n, int kbits) {
        uint32_t r = next() >> (32 - kbits);
        while (r >= n) r = next() >> (32 - kbits);
        return r;
    }
    // CPython random.random() == NumPy legacy random_sample(): 53-bit double from two words
    __device__ __forceinline__ double random53() {
        uint32_t a = next() >> 5, b = next() >> 6;
        return (a * 67108864.0 + b) / 9007199254740992.0;
    }
    __device__ __forceinline__ double uniform(double lo, double hi) { return lo + (hi - lo) * random53(); }
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
