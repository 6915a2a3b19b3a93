// cge_pcg.hpp — NumPy Generator(PCG64(SeedSequence(seed))) on the device (gfx950), bit-compatible with
// np.random.default_rng(seed): SeedSequence hashing, PCG64 XSL-RR 128/64 (step, then output), the buffered
// 32-bit path used by Generator.integers (Lemire), random() = (u64 >> 11) * 2^-53, the 256-layer ziggurat of
// standard_normal (tables: tools/gen_ziggurat_tables.py, pinned bit-for-bit against NumPy) and
// choice(n=4, p) = searchsorted(cumsum(p)/sum, random(), 'right').  40 bytes of state per env: it lives in the
// env's own state record, no separate stream block.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "zig_tables.hpp"

namespace cge {

typedef unsigned __int128 u128;

struct Pcg64 {
    u128 state, inc;
    uint32_t has_uint32, uinteger;

    static __host__ __device__ __forceinline__ uint32_t hashmix(uint32_t value, uint32_t &hc) {
        value ^= hc;
        hc *= 0x931e8875u;
        value *= hc;
        value ^= value >> 16;
        return value;
    }
    static __host__ __device__ __forceinline__ uint32_t mix(uint32_t x, uint32_t y) {
        const uint32_t r = 0xca01f9ddu * x - 0x4973f715u * y;
        return r ^ (r >> 16);
    }
    static __host__ __device__ __forceinline__ u128 mult() { return ((u128)0x2360ED051FC65DA4ull << 64) | 0x4385DF649FCCF645ull; }

    // np.random.default_rng(seed), seed a non-negative int < 2**64
    __host__ __device__ __forceinline__ void seed(uint64_t s) {
        const uint32_t ent[2] = {(uint32_t)s, (uint32_t)(s >> 32)};
        const int nent = ent[1] ? 2 : 1;
        uint32_t pool[4], hc = 0x43b0d7e5u;
#pragma unroll
        for (int i = 0; i < 4; ++i) pool[i] = hashmix(i < nent ? ent[i] : 0u, hc);
#pragma unroll
        for (int is = 0; is < 4; ++is)
#pragma unroll
            for (int id = 0; id < 4; ++id)
                if (is != id) pool[id] = mix(pool[id], hashmix(pool[is], hc));
        uint32_t w[8], hb = 0x8b51f9ddu;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            uint32_t v = pool[i % 4];
            v ^= hb;
            hb *= 0x58f38dedu;
            v *= hb;
            v ^= v >> 16;
            w[i] = v;
        }
        const uint64_t q0 = (uint64_t)w[0] | ((uint64_t)w[1] << 32), q1 = (uint64_t)w[2] | ((uint64_t)w[3] << 32);
        const uint64_t q2 = (uint64_t)w[4] | ((uint64_t)w[5] << 32), q3 = (uint64_t)w[6] | ((uint64_t)w[7] << 32);
        const u128 initstate = ((u128)q0 << 64) | q1, initseq = ((u128)q2 << 64) | q3;
        inc = (initseq << 1) | 1;
        state = inc;                  // state = 0; step
        state += initstate;
        state = state * mult() + inc;
        has_uint32 = 0;
        uinteger = 0;
    }
    __device__ __forceinline__ uint64_t next64() {
        state = state * mult() + inc;
        const uint64_t hi = (uint64_t)(state >> 64), lo = (uint64_t)state, x = hi ^ lo;
        const unsigned rot = (unsigned)(state >> 122);
        return (x >> rot) | (x << ((64u - rot) & 63u));
    }
    __device__ __forceinline__ uint32_t next32() {
        if (has_uint32) { has_uint32 = 0; return uinteger; }
        const uint64_t n = next64();
        has_uint32 = 1;
        uinteger = (uint32_t)(n >> 32);
        return (uint32_t)n;
    }
    __device__ __forceinline__ double random() { return (double)(next64() >> 11) * (1.0 / 9007199254740992.0); }
    __device__ __forceinline__ double uniform(double lo, double hi) { return lo + (hi - lo) * random(); }
    __device__ __forceinline__ int64_t integers(int64_t low, int64_t high) {      // Generator.integers(low, high), range < 2**32-1
        const uint32_t rng = (uint32_t)(high - low - 1);
        if (rng == 0) return low;
        const uint32_t rng_excl = rng + 1;
        uint64_t m = (uint64_t)next32() * rng_excl;
        uint32_t leftover = (uint32_t)m;
        if (leftover < rng_excl) {
            const uint32_t threshold = (0xFFFFFFFFu - rng) % rng_excl;
            while (leftover < threshold) { m = (uint64_t)next32() * rng_excl; leftover = (uint32_t)m; }
        }
        return low + (int64_t)(m >> 32);
    }
    __device__ __forceinline__ double standard_normal() {
        for (;;) {
            uint64_t r = next64();
            const int idx = (int)(r & 0xff);
            r >>= 8;
            const int sign = (int)(r & 1);
            const uint64_t rabs = (r >> 1) & 0x000fffffffffffffull;
            double x = (double)rabs * zig_wi[idx];
            if (sign) x = -x;
            if (rabs < zig_ki[idx]) return x;                    // 98.5 % of draws
            if (idx == 0) {
                for (;;) {
                    const double xx = -ZIG_NOR_INV_R * log1p(-random());
                    const double yy = -log1p(-random());
                    if (yy + yy > xx * xx) return ((rabs >> 8) & 1) ? -(ZIG_NOR_R + xx) : ZIG_NOR_R + xx;
                }
            } else if (((zig_fi[idx - 1] - zig_fi[idx]) * random() + zig_fi[idx]) < exp(-0.5 * x * x)) {
                return x;
            }
        }
    }
    __device__ __forceinline__ double normal(double loc, double scale) { return loc + scale * standard_normal(); }
    __device__ __forceinline__ int choice4(double p0, double p1, double p2, double p3) {
        double c0 = p0, c1 = c0 + p1, c2 = c1 + p2, c3 = c2 + p3;
        c0 /= c3; c1 /= c3; c2 /= c3; c3 /= c3;
        const double u = random();
        return (c0 <= u) + (c1 <= u) + (c2 <= u) + (c3 <= u);
    }
};

}  // namespace cge
