/* oracle/orc_rng.h — CPU restatement of the random generators the reference's envs draw from.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is linked, imported or executed by the
 * product path (custom_gymnasium_environments_amd/); only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg use it, as the checker.
 *
 * The algorithms live in third-party code that is not vendored in /root/reference
 * (CPython `_random` / `random.py`, NumPy legacy RandomState); they are restated here from their
 * published definitions and pinned by the known answers in SURVEY.md section 8c and by the
 * fixtures under tests/golden/ that were produced by running the reference itself.
 *
 * Reference call sites that fix WHICH generator and WHICH draw order each env uses:
 *   snake_env_classic/snake_env.py:125-126        random.randint(0, G-1) x2
 *   crypto_trading_env/crypto_trading_env.py:135,148,177,182-186,324-330,349,353-354,455,484
 *   traffic_management_env/environment.py:227,229 ; utils.py:95,181,187
 */
#ifndef ORC_RNG_H
#define ORC_RNG_H

#include <math.h>
#include <stdint.h>

#define ORC_MT_N 624
#define ORC_MT_M 397

typedef struct {
    uint32_t mt[ORC_MT_N];
    int idx;        /* next word to hand out; ORC_MT_N means "regenerate first" */
    int has_gauss;  /* NumPy legacy polar-method cache */
    double gauss;
} orc_mt;

/* Matsumoto & Nishimura init_genrand: NumPy legacy `np.random.seed(int)` uses exactly this. */
static inline void orc_mt_init_genrand(orc_mt *s, uint32_t seed) {
    s->mt[0] = seed;
    for (int i = 1; i < ORC_MT_N; ++i)
        s->mt[i] = 1812433253u * (s->mt[i - 1] ^ (s->mt[i - 1] >> 30)) + (uint32_t)i;
    s->idx = ORC_MT_N;
    s->has_gauss = 0;
    s->gauss = 0.0;
}

static inline void orc_mt_init_by_array(orc_mt *s, const uint32_t *key, int klen) {
    orc_mt_init_genrand(s, 19650218u);
    int i = 1, j = 0;
    int k = ORC_MT_N > klen ? ORC_MT_N : klen;
    for (; k; --k) {
        s->mt[i] = (s->mt[i] ^ ((s->mt[i - 1] ^ (s->mt[i - 1] >> 30)) * 1664525u)) + key[j] + (uint32_t)j;
        ++i; ++j;
        if (i >= ORC_MT_N) { s->mt[0] = s->mt[ORC_MT_N - 1]; i = 1; }
        if (j >= klen) j = 0;
    }
    for (k = ORC_MT_N - 1; k; --k) {
        s->mt[i] = (s->mt[i] ^ ((s->mt[i - 1] ^ (s->mt[i - 1] >> 30)) * 1566083941u)) - (uint32_t)i;
        ++i;
        if (i >= ORC_MT_N) { s->mt[0] = s->mt[ORC_MT_N - 1]; i = 1; }
    }
    s->mt[0] = 0x80000000u;
    s->idx = ORC_MT_N;
}

/* CPython `random.seed(n)` for a non-negative int n: init_by_array over the 32-bit
 * little-endian limbs of n (a single zero limb for n == 0). */
static inline void orc_py_seed(orc_mt *s, uint64_t seed) {
    uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    orc_mt_init_by_array(s, key, key[1] ? 2 : 1);
    s->has_gauss = 0;
}

/* NumPy legacy `np.random.seed(n)`, 0 <= n < 2**32. */
static inline void orc_np_seed(orc_mt *s, uint32_t seed) { orc_mt_init_genrand(s, seed); }

static inline uint32_t orc_mt_next(orc_mt *s) {
    if (s->idx >= ORC_MT_N) {
        uint32_t *mt = s->mt;
        int kk;
        for (kk = 0; kk < ORC_MT_N - ORC_MT_M; ++kk) {
            uint32_t y = (mt[kk] & 0x80000000u) | (mt[kk + 1] & 0x7fffffffu);
            mt[kk] = mt[kk + ORC_MT_M] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        for (; kk < ORC_MT_N - 1; ++kk) {
            uint32_t y = (mt[kk] & 0x80000000u) | (mt[kk + 1] & 0x7fffffffu);
            mt[kk] = mt[kk + (ORC_MT_M - ORC_MT_N)] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        uint32_t y = (mt[ORC_MT_N - 1] & 0x80000000u) | (mt[0] & 0x7fffffffu);
        mt[ORC_MT_N - 1] = mt[ORC_MT_M - 1] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        s->idx = 0;
    }
    uint32_t y = s->mt[s->idx++];
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}

/* 53-bit double shared by CPython random.random() and NumPy legacy random_sample(). */
static inline double orc_mt_double(orc_mt *s) {
    uint32_t a = orc_mt_next(s) >> 5, b = orc_mt_next(s) >> 6;
    return (a * 67108864.0 + b) / 9007199254740992.0;
}

static inline int orc_bit_length(uint32_t n) {
    int k = 0;
    while (n) { ++k; n >>= 1; }
    return k;
}

/* CPython Random._randbelow_with_getrandbits(n), n < 2**32: k = n.bit_length();
 * r = getrandbits(k) = next_u32 >> (32-k); redraw while r >= n. */
static inline uint32_t orc_py_randbelow(orc_mt *s, uint32_t n) {
    int k = orc_bit_length(n);
    uint32_t r = orc_mt_next(s) >> (32 - k);
    while (r >= n) r = orc_mt_next(s) >> (32 - k);
    return r;
}

static inline int orc_py_randint(orc_mt *s, int a, int b) { return a + (int)orc_py_randbelow(s, (uint32_t)(b - a + 1)); }

static inline double orc_py_uniform(orc_mt *s, double a, double b) { return a + (b - a) * orc_mt_double(s); }

/* NumPy legacy_gauss: Marsaglia polar method, returns f*x2 and caches f*x1. */
static inline double orc_np_gauss(orc_mt *s) {
    if (s->has_gauss) {
        s->has_gauss = 0;
        double g = s->gauss;
        s->gauss = 0.0;
        return g;
    }
    double f, x1, x2, r2;
    do {
        x1 = 2.0 * orc_mt_double(s) - 1.0;
        x2 = 2.0 * orc_mt_double(s) - 1.0;
        r2 = x1 * x1 + x2 * x2;
    } while (r2 >= 1.0 || r2 == 0.0);
    f = sqrt(-2.0 * log(r2) / r2);
    s->gauss = f * x1;
    s->has_gauss = 1;
    return f * x2;
}

static inline double orc_np_normal(orc_mt *s, double loc, double scale) { return loc + scale * orc_np_gauss(s); }

/* Synthetic action source shared by every backend (see tests/golden/gen/common.py). */
static inline uint64_t orc_mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

static inline uint32_t orc_hash_action(uint64_t a_seed, uint64_t env, uint64_t t, uint32_t n, uint32_t j) {
    uint64_t u = orc_mix64(orc_mix64(a_seed + env * 0x9E3779B97F4A7C15ull) + t * 0xD1342543DE82EF95ull + j);
    return (uint32_t)(((u >> 32) * (uint64_t)n) >> 32);
}

/* ------------------------------------------------------------------------------------------------
 * Family D/G: NumPy Generator(PCG64(SeedSequence(seed))) — smartclimate/env.py:30,63-65 (default_rng) and
 * smart_manufacturing_env (gymnasium's self.np_random).  Restated from NumPy's published sources
 * (bit_generator.pyx SeedSequence, _pcg64.pyx / pcg64.h, distributions.c); pinned against NumPy executed
 * here (tests/test_oracle_rng.py) and SURVEY 8c's known answers.
 * ---------------------------------------------------------------------------------------------- */
#include "orc_zig_tables.h"

typedef unsigned __int128 orc_u128;

typedef struct {
    orc_u128 state, inc;
    int has_uint32;
    uint32_t uinteger;
} orc_pcg;

static inline uint32_t orc_ss_hashmix(uint32_t value, uint32_t *hash_const) {
    value ^= *hash_const;
    *hash_const *= 0x931e8875u;
    value *= *hash_const;
    value ^= value >> 16;
    return value;
}
static inline uint32_t orc_ss_mix(uint32_t x, uint32_t y) {
    uint32_t r = 0xca01f9ddu * x - 0x4973f715u * y;
    return r ^ (r >> 16);
}
/* default_rng(seed): SeedSequence(seed).generate_state(4, uint64) -> pcg64_srandom_r */
static inline void orc_pcg_seed(orc_pcg *g, uint64_t seed) {
    uint32_t ent[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    int nent = ent[1] ? 2 : 1;
    uint32_t pool[4], hc = 0x43b0d7e5u;
    for (int i = 0; i < 4; ++i) pool[i] = orc_ss_hashmix(i < nent ? ent[i] : 0u, &hc);
    for (int is = 0; is < 4; ++is)
        for (int id = 0; id < 4; ++id)
            if (is != id) pool[id] = orc_ss_mix(pool[id], orc_ss_hashmix(pool[is], &hc));
    uint32_t w[8], hb = 0x8b51f9ddu;
    for (int i = 0; i < 8; ++i) {
        uint32_t v = pool[i % 4];
        v ^= hb;
        hb *= 0x58f38dedu;
        v *= hb;
        v ^= v >> 16;
        w[i] = v;
    }
    uint64_t q[4];
    for (int i = 0; i < 4; ++i) q[i] = (uint64_t)w[2 * i] | ((uint64_t)w[2 * i + 1] << 32);
    const orc_u128 mult = ((orc_u128)0x2360ED051FC65DA4ull << 64) | 0x4385DF649FCCF645ull;
    orc_u128 initstate = ((orc_u128)q[0] << 64) | q[1], initseq = ((orc_u128)q[2] << 64) | q[3];
    g->inc = (initseq << 1) | 1;
    g->state = g->inc;                       /* state = 0; step */
    g->state += initstate;
    g->state = g->state * mult + g->inc;
    g->has_uint32 = 0;
    g->uinteger = 0;
}
static inline uint64_t orc_pcg_next64(orc_pcg *g) {
    const orc_u128 mult = ((orc_u128)0x2360ED051FC65DA4ull << 64) | 0x4385DF649FCCF645ull;
    g->state = g->state * mult + g->inc;
    uint64_t hi = (uint64_t)(g->state >> 64), lo = (uint64_t)g->state, x = hi ^ lo;
    unsigned rot = (unsigned)(g->state >> 122);
    return (x >> rot) | (x << ((64 - rot) & 63));
}
static inline uint32_t orc_pcg_next32(orc_pcg *g) {       /* low half first, the high half is buffered */
    if (g->has_uint32) { g->has_uint32 = 0; return g->uinteger; }
    uint64_t n = orc_pcg_next64(g);
    g->has_uint32 = 1;
    g->uinteger = (uint32_t)(n >> 32);
    return (uint32_t)n;
}
static inline double orc_pcg_double(orc_pcg *g) { return (double)(orc_pcg_next64(g) >> 11) * (1.0 / 9007199254740992.0); }
static inline double orc_pcg_uniform(orc_pcg *g, double lo, double hi) { return lo + (hi - lo) * orc_pcg_double(g); }
/* Generator.integers(low, high), high-low-1 < 2**32-1: Lemire's method on buffered 32-bit draws */
static inline int64_t orc_pcg_integers(orc_pcg *g, int64_t low, int64_t high) {
    uint32_t rng = (uint32_t)(high - low - 1);
    if (rng == 0) return low;
    uint32_t rng_excl = rng + 1;
    uint64_t m = (uint64_t)orc_pcg_next32(g) * rng_excl;
    uint32_t leftover = (uint32_t)m;
    if (leftover < rng_excl) {
        uint32_t threshold = (0xFFFFFFFFu - rng) % rng_excl;
        while (leftover < threshold) { m = (uint64_t)orc_pcg_next32(g) * rng_excl; leftover = (uint32_t)m; }
    }
    return low + (int64_t)(m >> 32);
}
/* random_standard_normal: 256-layer ziggurat (tables: tools/gen_ziggurat_tables.py) */
static inline double orc_pcg_standard_normal(orc_pcg *g) {
    for (;;) {
        uint64_t r = orc_pcg_next64(g);
        int idx = (int)(r & 0xff);
        r >>= 8;
        int sign = (int)(r & 1);
        uint64_t rabs = (r >> 1) & 0x000fffffffffffffull;
        double x = (double)rabs * zig_wi[idx];
        if (sign) x = -x;
        if (rabs < zig_ki[idx]) return x;
        if (idx == 0) {
            for (;;) {
                double xx = -ZIG_NOR_INV_R * log1p(-orc_pcg_double(g));
                double yy = -log1p(-orc_pcg_double(g));
                if (yy + yy > xx * xx) return ((rabs >> 8) & 1) ? -(ZIG_NOR_R + xx) : ZIG_NOR_R + xx;
            }
        } else if (((zig_fi[idx - 1] - zig_fi[idx]) * orc_pcg_double(g) + zig_fi[idx]) < exp(-0.5 * x * x)) return x;
    }
}
static inline double orc_pcg_normal(orc_pcg *g, double loc, double scale) { return loc + scale * orc_pcg_standard_normal(g); }
/* Generator.choice(4 values, p=p): cdf = cumsum(p); cdf /= cdf[-1]; searchsorted(cdf, random(), 'right') */
static inline int orc_pcg_choice4(orc_pcg *g, const double *p) {
    double c[4];
    c[0] = p[0]; c[1] = c[0] + p[1]; c[2] = c[1] + p[2]; c[3] = c[2] + p[3];
    for (int i = 0; i < 4; ++i) c[i] /= c[3];       /* in place: the last element divides itself last */
    double u = orc_pcg_double(g);
    int idx = 0;
    while (idx < 4 && c[idx] <= u) ++idx;
    return idx;
}

#endif
