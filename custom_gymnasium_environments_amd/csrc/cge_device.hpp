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
#include <cstring>

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
    // = (uint32_t)(((mix64(z) >> 32) * n) >> 32) with z = key + t*C + j, written out because only the HIGH word of mix64's result
    // is used: the high word of y ^ (y >> 31) is yh ^ (yh >> 31), and yh, the high word of the second 64-bit product, takes three
    // 32-bit multiplies instead of four (v_mul_lo / v_mul_hi are quarter-rate on gfx950; the snake rollout spent 2.6 us of its
    // 28 us per 1M-env step in this hash).  Checked against the two-line definition on 5e7 random inputs (tools/probes/).
    uint64_t z = key + t * 0xD1342543DE82EF95ull + j;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z ^= z >> 27;
    const uint32_t zl = (uint32_t)z, zh = (uint32_t)(z >> 32);
    constexpr uint32_t cl = 0x133111EBu, ch = 0x94D049BBu;                      // 0x94D049BB133111EB
    const uint32_t yh = (uint32_t)(((uint64_t)zl * cl) >> 32) + zl * ch + zh * cl;   // high word of z * C (mod 2^64)
    const uint32_t uh = yh ^ (yh >> 31);                                          // high word of y ^ (y >> 31)
    return (uint32_t)(((uint64_t)uh * (uint64_t)n) >> 32);
}

// ------------------------------------------------------------------ MT19937 (family P / L)
constexpr int MT_N = 624;
constexpr int MT_M = 397;
constexpr int MT_STRIDE = 640;  // words per stream block: mt[624] + 16 mirror words (=2560 B, 20 x 128-B lines)
constexpr int MT_PAD = MT_STRIDE - MT_N;
// Words 624..639 of a block MIRROR words 0..15.  A draw window is then always a run of CONSECUTIVE words — w[pos .. pos+W] and
// w[pos+397 .. ] never wrap for W <= 16 — which lets a lane fetch it with a few 16-byte loads instead of 2W+1 single-dword loads
// (every one of them a 64-address instruction for the CU's address unit: round 2 profile of the crypto dynamics, 63 % of the wave
// time was issue stall).  Every writer of a block keeps the mirror current through mt_store().
__device__ __forceinline__ void mt_store(uint32_t *blk, uint32_t k, uint32_t y) {
    blk[k] = y;
    if (k < (uint32_t)MT_PAD) blk[MT_N + k] = y;
}
struct __attribute__((packed, aligned(4))) MtQuad { uint32_t a, b, c, d; };   // 4-byte aligned 16-byte access (unaligned access mode)
template <int N>
__device__ __forceinline__ void mt_load_run(const uint32_t *__restrict__ src, uint32_t (&dst)[N]) {
#pragma unroll
    for (int q = 0; q + 4 <= N; q += 4) {
        const MtQuad v = *reinterpret_cast<const MtQuad *>(src + q);
        dst[q] = v.a; dst[q + 1] = v.b; dst[q + 2] = v.c; dst[q + 3] = v.d;
    }
#pragma unroll
    for (int q = N - N % 4; q < N; ++q) dst[q] = src[q];
}
// The stream cursor lives in the ENV's state record, not in the block, so a draw never starts with a
// dependent "load the cursor" round trip:
//   pos    next word index, 0..623
//   pretw  words in [pos, pretw) are already twisted ("ready"): 0 (nothing), 624 (a CPython state imported by set_state: the whole
//          generation), or — streams advanced by mt_make_ready() below — a multiple of 32 up to 624 + 32, where the part beyond 624
//          means words [0, pretw - 624) of the NEXT generation
// Words are twisted in place (incremental form of the reference generator's 624-word batch regeneration; identical output
// sequence): one at a time exactly when they are consumed (MtStream, MtWindow, the LDS queues), or a 32-word chunk at a time
// AHEAD of the cursor (mt_make_ready) so that the draws themselves are plain loads and nothing is written back per draw.
__host__ __device__ __forceinline__ uint32_t mt_wrap_ready(uint32_t pretw) { return pretw > (uint32_t)MT_N ? pretw - (uint32_t)MT_N : 0u; }   // the cursor wrapped into the next generation

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

// ------------------------------------------------------------------ twist-ahead chunks
// A draw that twists its own words costs a window of w[pos..], a window of w[pos+397..] and a write-back of the consumed words —
// for crypto's 10-word step window that is 3-4 scattered 128-byte lines fetched and a 40-byte PARTIAL-line write per env, stream
// and step (round 2: after a block's first wrap these commits alone cost step() ~150 us per 1M envs, and the generator traffic
// was the whole gap between obliged and moved bytes).  Twisting a 32-word chunk (one 128-byte line) at a time ahead of the cursor
// turns that into one full-line write per ~3 steps, and the per-step draw into a plain load of ready words.
//   pretw codes: a multiple of 32 below 624, 624, or 624 + a multiple of 32 (chunks of the NEXT generation, twisted while the
//   cursor is still in this one); stored in 5 bits as mt_ready_encode(pretw).
constexpr int MT_CHUNK = 32;
__host__ __device__ __forceinline__ uint32_t mt_ready_decode(uint32_t q) { return q <= 19u ? 32u * q : (uint32_t)MT_N + 32u * (q - 20u); }   // q in 0..31
__host__ __device__ __forceinline__ uint32_t mt_ready_encode(uint32_t pretw) { return pretw < (uint32_t)MT_N ? pretw / 32u : 20u + (pretw - (uint32_t)MT_N) / 32u; }
// Twists, in place, the words from `lo` (unwrapped: >= 624 means word lo - 624 of the next generation) to the end of lo's 32-word
// chunk, for the lanes with `go`; words of the chunk below lo keep their (already twisted) values.  Returns the new ready mark.
// One lane = one block: 17 sixteen-byte loads (the chunk + 1 word, and the words 397 ahead — the piece that straddles word 623
// runs into mirror word 624 = word 0, exactly the word it needs), 8 stores of the whole line (+4 for the mirror of words 0..15).
// old0 (optional): receives the outgoing generation's word 0 when the next generation's first chunk overwrites it — its low 31
// bits influence no future output, but a byte-exact CPython export (mt_export_cpython) wants them back.
__device__ __forceinline__ uint32_t mt_twist_chunk(uint32_t *__restrict__ blk, uint32_t lo, bool go, uint32_t *old0 = nullptr) {
    const uint32_t gen = lo >= (uint32_t)MT_N ? (uint32_t)MT_N : 0u, base = lo - gen, c0 = base & ~31u;
    const uint32_t len = (uint32_t)MT_N - c0 < 32u ? (uint32_t)MT_N - c0 : 32u;   // the last chunk holds 16 words
    if (go) {
        uint32_t a[MT_CHUNK + 1], c[MT_CHUNK];
#pragma unroll
        for (int q = 0; q < MT_CHUNK; q += 4) {
            if ((uint32_t)q < len) {
                const MtQuad v = *reinterpret_cast<const MtQuad *>(blk + c0 + q);
                a[q] = v.a; a[q + 1] = v.b; a[q + 2] = v.c; a[q + 3] = v.d;
                uint32_t ci = c0 + (uint32_t)q + MT_M;                        // == 1 mod 4: 621 is the only piece that touches the mirror
                ci -= ci > (uint32_t)MT_N ? MT_N : 0;
                const MtQuad w = *reinterpret_cast<const MtQuad *>(blk + ci);
                c[q] = w.a; c[q + 1] = w.b; c[q + 2] = w.c; c[q + 3] = w.d;
            }
        }
        a[MT_CHUNK] = 0;
        if (old0 && c0 == 0u) *old0 = a[0];                             // the previous generation's word 0 goes away now
        const uint32_t nxt = blk[c0 + len];                                    // word c0 + len <= 624 (mirror of word 0)
#pragma unroll
        for (int j = 0; j < MT_CHUNK; ++j) {
            if ((uint32_t)j < len) {
                const uint32_t b = (uint32_t)(j + 1) < len ? a[j + 1] : nxt;
                const uint32_t y = mt_twist(a[j], b, c[j]);
                a[j] = c0 + (uint32_t)j < base ? a[j] : y;
            }
        }
#pragma unroll
        for (int q = 0; q < MT_CHUNK; q += 4) {
            if ((uint32_t)q < len) {
                *reinterpret_cast<MtQuad *>(blk + c0 + q) = MtQuad{a[q], a[q + 1], a[q + 2], a[q + 3]};
                if (c0 == 0u && q < MT_PAD) *reinterpret_cast<MtQuad *>(blk + MT_N + q) = MtQuad{a[q], a[q + 1], a[q + 2], a[q + 3]};
            }
        }
    }
    return gen + c0 + len;
}
// Makes words [pos, pos + need) ready (pos < 624, need <= 227).  Wave-convergent: as long as any lane of the wave is short, the
// short lanes twist their next chunk (a round per 32 words); call it with all lanes of the wave.
// `slack`: when some lane is short, lanes that would be short within `slack` more words twist their next chunk in the same round —
// the lanes of a wave consume at similar rates, so this turns "some lane, nearly every step" into "most lanes, every few steps".
__device__ __forceinline__ void mt_make_ready(uint32_t *__restrict__ blk, uint32_t pos, uint32_t &pretw, uint32_t need, bool active = true,
                                              uint32_t *old0 = nullptr, uint32_t slack = 0u) {
#pragma unroll 1
    while (__ballot(active && pos + need > pretw)) {
        const bool go = active && pos + need + slack > pretw;
        const uint32_t t = mt_twist_chunk(blk, pretw > pos ? pretw : pos, go, old0);
        if (go) pretw = t;
    }
}
// W ready words from the cursor on (after mt_make_ready): tempered; words past 623 come from the mirror
template <int W>
__device__ __forceinline__ void mt_load_ready(const uint32_t *__restrict__ blk, uint32_t pos, uint32_t (&w)[W]) {
    static_assert(W <= MT_PAD, "the run may extend into the 16 mirror words");
    mt_load_run<W>(blk + pos, w);
}
__device__ __forceinline__ void mt_advance(uint32_t &pos, uint32_t &pretw, uint32_t used) {
    uint32_t p = pos + used;
    if (p >= (uint32_t)MT_N) { p -= MT_N; pretw = mt_wrap_ready(pretw); }
    pos = p;
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
            mt_store(w, p, y);
        }
        ++p;
        if (p == MT_N) { p = 0; pretw = mt_wrap_ready(pretw); }
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

    static constexpr bool RUN = W <= MT_PAD;                      // short windows are runs of consecutive words (mirror), see MT_PAD
    __device__ __forceinline__ void load(const uint32_t *__restrict__ blk, uint32_t pos) {
        if constexpr (RUN) {
            mt_load_run<W + 1>(blk + pos, a);                     // w[pos .. pos+W], mirror words past 623
            uint32_t cpos = pos + MT_M;
            cpos -= cpos >= (uint32_t)MT_N ? MT_N : 0;
            mt_load_run<W>(blk + cpos, c);                        // w[pos+397 ..]
        } else {
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
    }
    // untempered word number j (static index) of the stream starting at (pos, pretw)
    __device__ __forceinline__ uint32_t twisted(int j, uint32_t pos, uint32_t pretw) const {
        const uint32_t y = mt_twist(a[j], a[j + 1], c[j]);
        return (pos + j < pretw) ? a[j] : y;
    }
    __device__ __forceinline__ uint32_t draw(int j, uint32_t pos, uint32_t pretw) const { return mt_temper(twisted(j, pos, pretw)); }
    // persist the first `used` words and advance the cursor
    __device__ __forceinline__ void commit(uint32_t *__restrict__ blk, uint32_t &pos, uint32_t &pretw, uint32_t used) const {
        if constexpr (RUN) {
            // the window goes back as one run: consumed words twisted, the others as they were (this lane owns the block)
            uint32_t v[W];
#pragma unroll
            for (int j = 0; j < W; ++j) v[j] = ((uint32_t)j < used && pos + j >= pretw) ? mt_twist(a[j], a[j + 1], c[j]) : a[j];
#pragma unroll
            for (int q = 0; q + 4 <= W; q += 4) *reinterpret_cast<MtQuad *>(blk + pos + q) = MtQuad{v[q], v[q + 1], v[q + 2], v[q + 3]};
#pragma unroll
            for (int q = W - W % 4; q < W; ++q) blk[pos + q] = v[q];
            if (pos + W > (uint32_t)MT_N || pos < (uint32_t)MT_PAD) {     // the run touched the mirror or the mirrored words: fix the twin
#pragma unroll
                for (int j = 0; j < W; ++j) {
                    const uint32_t k = pos + j;
                    if (k >= (uint32_t)MT_N) blk[k - MT_N] = v[j];
                    else if (k < (uint32_t)MT_PAD) blk[MT_N + k] = v[j];
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < W; ++j) {
                uint32_t k = pos + j;
                if ((uint32_t)j < used && k >= pretw) {
                    const uint32_t y = mt_twist(a[j], a[j + 1], c[j]);
                    k -= k >= (uint32_t)MT_N ? MT_N : 0;
                    mt_store(blk, k, y);
                }
            }
        }
        uint32_t p = pos + used;
        if (p >= (uint32_t)MT_N) { p -= MT_N; pretw = mt_wrap_ready(pretw); }
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
        if constexpr (W <= MT_PAD) {
            MtWindow<W> w;
            w.load(blk, pos);
#pragma unroll
            for (int j = 0; j < W; ++j) row[j] = w.twisted(j, pos, pretw);
        } else {
            // long windows: MT_PAD-word runs, NB of them per round trip (a bounded register footprint; every 128-byte line
            // of the block is fetched once per window instead of once per 16-word refill)
            static_assert(W % MT_PAD == 0 && W <= MT_N - MT_M, "whole runs, mutually independent words");
            constexpr int NB = W % (3 * MT_PAD) == 0 ? 3 : W % (2 * MT_PAD) == 0 ? 2 : 1;   // (all six runs of a 96-word window at once: slower, 23 vs 17 us)
#pragma unroll 1
            for (int c0 = 0; c0 < W; c0 += NB * MT_PAD) {
                MtWindow<MT_PAD> w[NB];
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                    uint32_t start = pos + (uint32_t)(c0 + b * MT_PAD);
                    start -= start >= (uint32_t)MT_N ? MT_N : 0;
                    w[b].load(blk, start);
                }
#pragma unroll
                for (int b = 0; b < NB; ++b)
#pragma unroll
                    for (int j = 0; j < MT_PAD; ++j) row[c0 + b * MT_PAD + j] = w[b].twisted(j, pos + (uint32_t)(c0 + b * MT_PAD), pretw);
            }
        }
        cur = 0;
        filled = true;
    }
    // fill() for a stream whose next W words are READY (mt_make_ready): plain loads of the runs — no words 397 ahead, no twist —
    // and the flush that follows writes nothing back (every consumed word lies below pretw)
    __device__ __forceinline__ void fill_ready() {
        static_assert(W % 4 == 0, "whole 16-byte pieces");
#pragma unroll
        for (int c0 = 0; c0 < W; c0 += MT_PAD) {
            constexpr int N = W < MT_PAD ? W : MT_PAD;
            uint32_t start = pos + (uint32_t)c0;
            start -= start >= (uint32_t)MT_N ? MT_N : 0;           // words past 623 of a run come from the mirror
            uint32_t a[N];
            mt_load_run<N>(blk + start, a);
#pragma unroll
            for (int j = 0; j < N; ++j) row[c0 + j] = a[j];
        }
        cur = 0;
        filled = true;
    }
    __device__ __forceinline__ void flush() {
        if (!filled) return;
        if (pos + cur <= pretw) {                              // every consumed word was ready (twist-ahead streams): nothing to write back
            uint32_t p = pos + cur;
            if (p >= (uint32_t)MT_N) { p -= MT_N; pretw = mt_wrap_ready(pretw); }
            pos = p; cur = 0; filled = false;
            return;
        }
        uint32_t j = 0;
        if constexpr (W > MT_PAD) {
            // whole runs of MT_PAD consumed words that lie clear of the mirror (words 0..15 <-> 624..639) and of the wrap go
            // back as four 16-byte stores, or not at all while the block is still in its seeded generation (k < pretw)
#pragma unroll 1
            for (; j + (uint32_t)MT_PAD <= cur; j += MT_PAD) {
                const uint32_t k = pos + j;
                uint32_t kp = k;
                kp -= kp >= (uint32_t)MT_N ? MT_N : 0;
                if (k + (uint32_t)MT_PAD <= pretw) continue;
                if (k >= pretw && kp >= (uint32_t)MT_PAD && kp + (uint32_t)MT_PAD <= (uint32_t)MT_N) {
                    uint32_t v[MT_PAD];
#pragma unroll
                    for (int q = 0; q < MT_PAD; ++q) v[q] = row[j + q];
#pragma unroll
                    for (int q = 0; q < MT_PAD; q += 4) *reinterpret_cast<MtQuad *>(blk + kp + q) = MtQuad{v[q], v[q + 1], v[q + 2], v[q + 3]};
                } else {
#pragma unroll 1
                    for (uint32_t q = 0; q < (uint32_t)MT_PAD; ++q) {
                        uint32_t kq = k + q;
                        if (kq >= pretw) {
                            kq -= kq >= (uint32_t)MT_N ? MT_N : 0;
                            mt_store(blk, kq, row[j + q]);
                        }
                    }
                }
            }
        }
        if constexpr (W > MT_PAD) {
            // the last partial run: all its words read at once, stored one by one under their own predicate
            if (j < cur) {
                const uint32_t jn = j + (uint32_t)MT_PAD <= (uint32_t)W ? j : (uint32_t)W - (uint32_t)MT_PAD;    // keep the reads inside the row
                uint32_t v[MT_PAD];
#pragma unroll
                for (int q = 0; q < MT_PAD; ++q) v[q] = row[jn + q];
#pragma unroll
                for (int q = 0; q < MT_PAD; ++q) {
                    const uint32_t jq = jn + (uint32_t)q;
                    uint32_t k = pos + jq;
                    if (jq >= j && jq < cur && k >= pretw) {
                        k -= k >= (uint32_t)MT_N ? MT_N : 0;
                        mt_store(blk, k, v[q]);
                    }
                }
            }
        } else {
            for (; j < cur; ++j) {
                uint32_t k = pos + j;
                if (k >= pretw) {
                    k -= k >= (uint32_t)MT_N ? MT_N : 0;
                    mt_store(blk, k, row[j]);
                }
            }
        }
        uint32_t p = pos + cur;
        if (p >= (uint32_t)MT_N) { p -= MT_N; pretw = mt_wrap_ready(pretw); }
        pos = p;
        cur = 0;
        filled = false;
    }
    __device__ __forceinline__ uint32_t next() {
        if (!filled) fill();
        else if (cur == (uint32_t)W) { flush(); fill(); }
        return mt_temper(row[cur++]);
    }
    // Twist-ahead form of ensure() (the env's record keeps a ready mark, mt_ready_encode): the window is refilled only when a lane
    // runs low, from words made ready a 32-word chunk at a time — plain loads, and nothing to write back on the way out.
    __device__ __forceinline__ void ensure_ahead(uint32_t need, bool active) {
        const bool shortfall = !filled || (uint32_t)W - cur < need;
        if (__ballot(shortfall) != 0ull) {
            flush();
            mt_make_ready(blk, pos, pretw, (uint32_t)W, active);
            fill_ready();
        }
    }
    // Wave-convergent top-up: refill ALL active lanes as soon as ANY of them has fewer than `need` words left.  Without
    // it the lanes' cursors drift apart and every draw site ends up refilling (a ~500-instruction path plus a memory round
    // trip) for some lane; one call at the top of a step bounds that to one refill per wave-step.
    __device__ __forceinline__ void ensure(uint32_t need) {
        const bool shortfall = !filled || (uint32_t)W - cur < need;
        if (__ballot(shortfall) != 0ull) { flush(); fill(); }
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

// Same queue, but the refill is a real (by-value) call instead of being inlined at every draw site.  For kernels with
// many draw sites (hospital: ~60) the inlined flush + 16-word fill made the kernel ~90k instructions of straight-line
// code and instruction fetch the bottleneck; the cursor fields still live in registers.
template <int W>
struct LdsDrawsCall {
    uint32_t *row;
    uint32_t *blk;
    uint32_t pos, pretw, cur;
    uint32_t old0 = 0;                                          // twist-ahead streams: see mt_twist_chunk (kept in the env's record)
    bool filled;

    __device__ __forceinline__ LdsDrawsCall(uint32_t *lds_row, uint32_t *block, uint32_t pos_, uint32_t pretw_)
        : row(lds_row), blk(block), pos(pos_), pretw(pretw_), cur(0), filled(false) {}
    static __device__ __attribute__((noinline)) uint2 refill(uint32_t *row, uint32_t *blk, uint32_t pos, uint32_t pretw, uint32_t cur, bool filled) {
        LdsDraws<W> d(row, blk, pos, pretw);
        d.cur = cur; d.filled = filled;
        d.flush();
        d.fill();
        return make_uint2(d.pos, d.pretw);
    }
    __device__ __forceinline__ void flush() {
        LdsDraws<W> d(row, blk, pos, pretw);
        d.cur = cur; d.filled = filled;
        d.flush();
        pos = d.pos; pretw = d.pretw; cur = 0; filled = false;
    }
    __device__ __forceinline__ void fill() {                   // unconditional (re)fill through the same call
        const uint2 r = refill(row, blk, pos, pretw, cur, filled);
        pos = r.x; pretw = r.y; cur = 0; filled = true;
    }
    // Look-ahead for draw sequences whose word OFFSETS can be computed up front: has(n) says the next n words are parked,
    // peek(j) is the tempered word j places ahead of the cursor (any order, as often as needed), skip(n) consumes n words.
    __device__ __forceinline__ bool has(uint32_t n) const { return filled && cur + n <= (uint32_t)W; }
    __device__ __forceinline__ uint32_t peek(uint32_t j) const { return mt_temper(row[cur + j]); }
    __device__ __forceinline__ void skip(uint32_t n) { cur += n; }
    __device__ __forceinline__ uint32_t next() {
        if (!filled || cur == (uint32_t)W) {
            const uint2 r = refill(row, blk, pos, pretw, cur, filled);
            pos = r.x; pretw = r.y; cur = 0; filled = true;
        }
        return mt_temper(row[cur++]);
    }
    // Wave-convergent top-up.  Lanes consume at different rates, so with refills only "when empty" every draw site ends
    // up refilling for SOME lane and the wave executes the ~500-instruction refill at nearly every site (measured on
    // hospital: 50k VALU per wave-step).  Called at a program point that the lanes reach together, this refills ALL of them
    // as soon as ANY has fewer than `need` words left, which also re-synchronises their cursors.
    __device__ __forceinline__ void ensure(uint32_t need) {
        const bool shortfall = !filled || (uint32_t)W - cur < need;
        if (__ballot(shortfall) != 0ull) {
            const uint2 r = refill(row, blk, pos, pretw, cur, filled);
            pos = r.x; pretw = r.y; cur = 0; filled = true;
        }
    }
    // The same two entry points for streams that twist ahead (the env's record keeps a ready mark, mt_ready_encode): the W words
    // after the cursor are made ready first — whole 32-word chunks, wave-convergent — so the fill is plain loads and the flush
    // writes nothing.  `active`: lanes that own a live env.  Call with all lanes of the wave.
    __device__ __forceinline__ void fill_ahead(bool active) {
        LdsDraws<W> d(row, blk, pos, pretw);
        d.cur = cur; d.filled = filled;
        d.flush();
        mt_make_ready(blk, d.pos, d.pretw, (uint32_t)W, active, &old0);
        d.fill_ready();
        pos = d.pos; pretw = d.pretw; cur = 0; filled = true;
    }
    __device__ __forceinline__ void ensure_ahead(uint32_t need, bool active) {
        const bool shortfall = !filled || (uint32_t)W - cur < need;
        if (__ballot(shortfall) != 0ull) fill_ahead(active);
    }
    __device__ __forceinline__ void fill_inline() {            // fill() expanded in place (one site per kernel)
        LdsDraws<W> d(row, blk, pos, pretw);
        d.cur = cur; d.filled = filled;
        d.flush();
        d.fill();
        pos = d.pos; pretw = d.pretw; cur = 0; filled = true;
    }
    // The same top-up with the flush and the fill expanded in place: for the ONE site per step that refills every time (the
    // call costs the callee's register saves and the caller's spills around it, all scratch traffic).
    __device__ __forceinline__ void ensure_inline(uint32_t need) {
        const bool shortfall = !filled || (uint32_t)W - cur < need;
        if (__ballot(shortfall) != 0ull) {
            LdsDraws<W> d(row, blk, pos, pretw);
            d.cur = cur; d.filled = filled;
            d.flush();
            d.fill();
            pos = d.pos; pretw = d.pretw; cur = 0; filled = true;
        }
    }
    __device__ __forceinline__ uint32_t randbelow(uint32_t n, int kbits) {
        uint32_t r = next() >> (32 - kbits);
        while (r >= n) r = next() >> (32 - kbits);
        return r;
    }
    __device__ __forceinline__ double random53() {
        const uint32_t a = next() >> 5, b = next() >> 6;
        return (a * 67108864.0 + b) / 9007199254740992.0;
    }
};

// Bulk variant for kernels that run FEW waves and are bound by the latency of one lane's serial draw chain
// (fleet's dense reset kernel): one rolled-loop fill parks W <= 227 twisted words per lane (no word of such a
// window depends on another), so a whole episode reset normally needs a single, wave-convergent fill per
// stream.  Same (pos, pretw) cursor contract as LdsDraws; flush() commits only the consumed words.
// An LDS row is 2W words: [0,W) the twisted state words (what flush() writes back), [W,2W) the same words
// tempered at fill time, by all lanes in parallel — next() on the serial chain is then one LDS read.
template <int W>
struct LdsBulkDraws {
    static_assert(W <= MT_N - MT_M, "window words must be mutually independent");
    uint32_t *row;
    uint32_t *blk;
#ifdef CGE_FLEET_TIMING
    uint32_t refills = 0;
#endif
    uint32_t pos, pretw, cur, avail;                           // avail: parked words (W, or fewer after a partial cooperative fill)
    bool filled;

    __device__ __forceinline__ LdsBulkDraws(uint32_t *lds_row, uint32_t *block, uint32_t pos_, uint32_t pretw_)
        : row(lds_row), blk(block), pos(pos_), pretw(pretw_), cur(0), avail(0), filled(false) {}
    static constexpr int CH = 32;                              // words per round trip (2*CH+1 loads in flight)
    static_assert(W % CH == 0, "window is filled in whole chunks");
    __device__ __forceinline__ void fill() {
#pragma unroll 1
        for (int c0 = 0; c0 < W; c0 += CH) {
            MtWindow<CH> w;                                    // loads first, unconditionally: one round trip per chunk
            w.load(blk, pos + (uint32_t)c0);
#pragma unroll
            for (int j = 0; j < CH; ++j) {
                const uint32_t y = w.twisted(j, pos + (uint32_t)c0, pretw);
                row[c0 + j] = y; row[W + c0 + j] = mt_temper(y);
            }
        }
        cur = 0;
        avail = (uint32_t)W;
        filled = true;
    }
    __device__ __forceinline__ void flush() {
        if (!filled) return;
        for (uint32_t j = 0; j < cur; ++j) {
            uint32_t k = pos + j;
            if (k >= pretw) {
                k -= k >= (uint32_t)MT_N ? MT_N : 0;
                mt_store(blk, k, row[j]);
            }
        }
        uint32_t p = pos + cur;
        if (p >= (uint32_t)MT_N) { p -= MT_N; pretw = mt_wrap_ready(pretw); }
        pos = p;
        cur = 0;
        filled = false;
    }
    // rare (window exhausted mid-path): a real call, by value, so the cursor fields stay in registers
    static __device__ __attribute__((noinline)) uint2 refill(uint32_t *row, uint32_t *blk, uint32_t pos, uint32_t pretw, uint32_t cur, bool filled) {
        LdsBulkDraws d(row, blk, pos, pretw);
        d.cur = cur; d.filled = filled;
        d.flush();
        d.fill();
        return make_uint2(d.pos, d.pretw);
    }
    __device__ __forceinline__ uint32_t next() {
        if (!filled || cur == avail) {
            const uint2 r = refill(row, blk, pos, pretw, cur, filled);
            pos = r.x; pretw = r.y; cur = 0; avail = (uint32_t)W; filled = true;
#ifdef CGE_FLEET_TIMING
            refills += 1;
#endif
        }
        return row[W + cur++];
    }
    __device__ __forceinline__ uint32_t randbelow(uint32_t n, int kbits) {
        uint32_t r = next() >> (32 - kbits);
        while (r >= n) r = next() >> (32 - kbits);
        return r;
    }
    __device__ __forceinline__ double random53() {
        const uint32_t a = next() >> 5, b = next() >> 6;
        return (a * 67108864.0 + b) / 9007199254740992.0;
    }
};

// Ring window for fused rollouts: W twisted words per env parked in the lane's LDS row as a circular buffer of MT_PAD-word
// runs.  A top-up replaces only the runs that were consumed completely — the run goes back to the generator block (four
// 16-byte stores) and the run W words ahead takes its slots — so every word of the block is fetched once and written once,
// where LdsDraws<W> refetches the whole window (and re-reads every 128-byte line it straddles) whenever it runs low.
// ensure() is the wave-convergent top-up for the top of a step; after it at least W - MT_PAD + 1 words are parked.
// Same (pos, pretw) cursor contract as the other queues: pos is the stream position of the ring's base.
// AHEAD: the stream twists ahead of its cursor in 32-word chunks (the env's record keeps a ready mark, mt_ready_encode): every run the
// ring fetches is made ready first (mt_make_ready, wave-convergent), so a fetch is plain loads — no words 397 ahead, no twist —
// and a consumed run is never written back.
template <int W, bool AHEAD = false>
struct RingDraws {
    static_assert(W % MT_PAD == 0 && W >= 2 * MT_PAD && W + 2 * MT_PAD <= MT_N - MT_M, "whole runs; the two runs fetched ahead are independent of the parked ones");
    static constexpr int NRUN = W / MT_PAD;
    uint32_t *row, *blk;
    uint32_t pos, pretw, head, cur;                            // head: slot of the base (a multiple of MT_PAD); cur: words consumed since the base
    bool filled;

    __device__ __forceinline__ RingDraws(uint32_t *lds_row, uint32_t *block, uint32_t pos_, uint32_t pretw_)
        : row(lds_row), blk(block), pos(pos_), pretw(pretw_), head(0), cur(0), filled(false) {}
    // twisted words [logical, logical + MT_PAD) -> slots [slot0, slot0 + MT_PAD)
    __device__ __forceinline__ void fill_run(uint32_t slot0, uint32_t logical) {
        MtWindow<MT_PAD> w;
        uint32_t start = logical;
        start -= start >= (uint32_t)MT_N ? MT_N : 0;
        w.load(blk, start);
#pragma unroll
        for (int j = 0; j < MT_PAD; ++j) row[slot0 + j] = w.twisted(j, logical, pretw);
    }
    // words [logical, logical + count) of the run parked at slot0 go back to the block (count <= MT_PAD)
    __device__ __forceinline__ void flush_run(uint32_t slot0, uint32_t logical, uint32_t count) {
        uint32_t kp = logical;
        kp -= kp >= (uint32_t)MT_N ? MT_N : 0;
        if (logical + count <= pretw) return;                  // seeded generation: nothing was twisted
        uint32_t v[MT_PAD];
#pragma unroll
        for (int q = 0; q < MT_PAD; ++q) v[q] = row[slot0 + q];
        if (count == (uint32_t)MT_PAD && logical >= pretw && kp >= (uint32_t)MT_PAD && kp + (uint32_t)MT_PAD <= (uint32_t)MT_N) {
#pragma unroll
            for (int q = 0; q < MT_PAD; q += 4) *reinterpret_cast<MtQuad *>(blk + kp + q) = MtQuad{v[q], v[q + 1], v[q + 2], v[q + 3]};
        } else {
#pragma unroll
            for (int q = 0; q < MT_PAD; ++q) {
                uint32_t k = logical + (uint32_t)q;
                if ((uint32_t)q < count && k >= pretw) {
                    k -= k >= (uint32_t)MT_N ? MT_N : 0;
                    mt_store(blk, k, v[q]);
                }
            }
        }
    }
    __device__ __forceinline__ void advance(uint32_t n) {
        pos += n;
        if (pos >= (uint32_t)MT_N) { pos -= MT_N; pretw = mt_wrap_ready(pretw); }
    }
    // NB runs per round trip: loads of all of them first, then twist and park
    template <int NB>
    __device__ __forceinline__ void fill_runs(uint32_t slot0, uint32_t logical, uint32_t nruns) {
        if constexpr (AHEAD) {                                 // the runs are ready: plain loads
            uint32_t a[NB][MT_PAD];
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                uint32_t start = logical + (uint32_t)(b * MT_PAD);
                start -= start >= (uint32_t)MT_N ? MT_N : 0;
                if ((uint32_t)b < nruns) mt_load_run<MT_PAD>(blk + start, a[b]);
            }
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                if ((uint32_t)b < nruns) {
                    uint32_t sl = slot0 + (uint32_t)(b * MT_PAD);
                    sl -= sl >= (uint32_t)W ? W : 0;
#pragma unroll
                    for (int j = 0; j < MT_PAD; ++j) row[sl + j] = a[b][j];
                }
            }
            return;
        }
        MtWindow<MT_PAD> w[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            uint32_t start = logical + (uint32_t)(b * MT_PAD);
            start -= start >= (uint32_t)MT_N ? MT_N : 0;
            if ((uint32_t)b < nruns) w[b].load(blk, start);
        }
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            if ((uint32_t)b < nruns) {
                uint32_t sl = slot0 + (uint32_t)(b * MT_PAD);
                sl -= sl >= (uint32_t)W ? W : 0;
#pragma unroll
                for (int j = 0; j < MT_PAD; ++j) row[sl + j] = w[b].twisted(j, logical + (uint32_t)(b * MT_PAD), pretw);
            }
        }
    }
    __device__ __forceinline__ void fill_all() {
        constexpr int NB = 3;
        if constexpr (AHEAD) mt_make_ready(blk, pos, pretw, (uint32_t)(W + 2 * MT_PAD), true);
#pragma unroll 1
        for (int r = 0; r < NRUN; r += NB)
            fill_runs<NB>((uint32_t)(r * MT_PAD), pos + (uint32_t)(r * MT_PAD), (uint32_t)(NRUN - r < NB ? NRUN - r : NB));
        head = 0; cur = 0; filled = true;
    }
    // every completely consumed run is written back and replaced by the run W words ahead, two runs per round trip
    __device__ __forceinline__ void top_up() {
        if (!filled) { fill_all(); return; }
        if constexpr (AHEAD) mt_make_ready(blk, pos, pretw, cur + (uint32_t)(W + 2 * MT_PAD) < 224u ? cur + (uint32_t)(W + 2 * MT_PAD) : 224u, true);
#pragma unroll 1
        while (cur >= (uint32_t)MT_PAD) {
            const uint32_t n = cur >= 2u * MT_PAD ? 2u : 1u;
            uint32_t h2 = head + (uint32_t)MT_PAD;
            h2 -= h2 >= (uint32_t)W ? W : 0;
            flush_run(head, pos, MT_PAD);
            if (n == 2u) flush_run(h2, pos + (uint32_t)MT_PAD, MT_PAD);
            fill_runs<2>(head, pos + (uint32_t)W, n);
            head = n == 2u ? h2 + (uint32_t)MT_PAD : h2;
            head -= head >= (uint32_t)W ? W : 0;
            advance(n * MT_PAD);
            cur -= n * MT_PAD;
        }
    }
    __device__ __forceinline__ void ensure() {                 // wave-convergent: lanes with nothing to do idle through it
        const bool want = !filled || cur >= (uint32_t)MT_PAD;
        if (__ballot(want) != 0ull) { if (want) top_up(); }
    }
    __device__ __forceinline__ void ensure_inline(uint32_t) { ensure(); }          // the other queues' spellings (need <= W - MT_PAD + 1)
    __device__ __forceinline__ void ensure(uint32_t) { ensure(); }
    __device__ __forceinline__ void flush() {                  // end of the rollout: everything consumed goes back, the ring is dropped
        if (!filled) return;
#pragma unroll 1
        while (cur >= (uint32_t)MT_PAD) {
            flush_run(head, pos, MT_PAD);
            head = head + (uint32_t)MT_PAD == (uint32_t)W ? 0u : head + (uint32_t)MT_PAD;
            advance(MT_PAD);
            cur -= MT_PAD;
        }
        if (cur) { flush_run(head, pos, cur); advance(cur); }
        head = 0; cur = 0; filled = false;
    }
    // rare (more than W - MT_PAD words in one step): a real call, by value
    static __device__ __attribute__((noinline)) uint2 restart(uint32_t *row, uint32_t *blk, uint32_t pos, uint32_t pretw, uint32_t head, uint32_t cur, bool filled) {
        RingDraws d(row, blk, pos, pretw);
        d.head = head; d.cur = cur; d.filled = filled;
        d.flush();
        d.fill_all();
        return make_uint2(d.pos, d.pretw);
    }
    __device__ __forceinline__ uint32_t slot_of(uint32_t j) const {
        uint32_t sl = head + cur + j;
        sl -= sl >= (uint32_t)W ? W : 0;
        return sl;
    }
    __device__ __forceinline__ bool has(uint32_t n) const { return filled && cur + n <= (uint32_t)W; }
    __device__ __forceinline__ uint32_t peek(uint32_t j) const { return mt_temper(row[slot_of(j)]); }
    __device__ __forceinline__ void skip(uint32_t n) { cur += n; }
    __device__ __forceinline__ uint32_t next() {
        if (!filled || cur == (uint32_t)W) {
            const uint2 r = restart(row, blk, pos, pretw, head, cur, filled);
            pos = r.x; pretw = r.y; head = 0; cur = 0; filled = true;
        }
        const uint32_t y = mt_temper(row[slot_of(0)]);
        cur += 1;
        return y;
    }
    __device__ __forceinline__ uint32_t randbelow(uint32_t n, int kbits) {
        uint32_t r = next() >> (32 - kbits);
        while (r >= n) r = next() >> (32 - kbits);
        return r;
    }
    __device__ __forceinline__ double random53() {
        const uint32_t a = next() >> 5, b = next() >> 6;
        return (a * 67108864.0 + b) / 9007199254740992.0;
    }
};

// Bounds-checked debug build (-DCGE_GUARD: `python -m custom_gymnasium_environments_amd.build --guard`, probe tools/probes/guard_run.py): the
// ring / work-list / window indices of the hospital and fleet kernels go through CGE_GX(site, index, limit).  An index outside its table
// is RECORDED (first violation: site, index, limit, block, lane; plus a count; one record per translation unit, read back through
// cge_<env>_debug_guard) and replaced by 0 instead of being dereferenced.  Release builds: CGE_GX is the index.  (Manufacturing's tables
// have the same under -DCGE_MFG_GUARD, manufacturing.hip.)
#ifdef CGE_GUARD
static __device__ unsigned int g_guard[8];
__device__ __forceinline__ uint32_t guard_index(int site_, uint32_t index, uint32_t limit) {
    if (index < limit) return index;
    if (atomicAdd(&g_guard[0], 1u) == 0u) { g_guard[1] = (unsigned)site_; g_guard[2] = index; g_guard[3] = limit; g_guard[4] = blockIdx.x; g_guard[5] = threadIdx.x; }
    return 0u;
}
#define CGE_GX(site_, index, limit) guard_index(site_, (uint32_t)(index), (uint32_t)(limit))
#else
#define CGE_GX(site_, index, limit) (index)
#endif

// readlane with a wave-uniform lane index; the builtin is typed int, so cast back before widening
__device__ __forceinline__ uint32_t lane_u32(uint32_t v, int r) { return (uint32_t)__builtin_amdgcn_readlane((int)v, r); }
__device__ __forceinline__ uint32_t *lane_ptr(uint32_t lo, uint32_t hi, int r) {
    return reinterpret_cast<uint32_t *>(((uint64_t)lane_u32(hi, r) << 32) | (uint64_t)lane_u32(lo, r));
}

// Wave-cooperative fill / flush of LdsBulkDraws windows.  Lane r (< DLN) owns stream r and LDS row r; the whole
// wave then serves one stream at a time: 64 lanes load 64 CONSECUTIVE state words (one 256-byte coalesced access
// instead of 64 lanes each walking their own 2560-byte block), twist, temper and park the words in row r.  Per lane
// the own-block walk costs 2W+1 narrow loads that all hit the same 2-4 cache lines while those are still in flight.
// Two halves on purpose: coop_fill_issue() only issues loads (every row and quad of the window, one round trip; the
// caller puts its other loads — the env record — right after it), coop_fill_park() twists and parks.  `quads` is the
// per-row number of 64-word quads that row needs (0: none); quads that are not needed are not parked.
template <int W, int DLN>
struct CoopFillRegs {
    static constexpr int NQ = (W + 63) / 64;
    uint32_t a[DLN][NQ], b[DLN][NQ], c[DLN][NQ];
};
template <int W, int DLN>
__device__ __forceinline__ void coop_fill_issue(const LdsBulkDraws<W> &d, CoopFillRegs<W, DLN> &f, uint32_t quads) {
    constexpr int NQ = CoopFillRegs<W, DLN>::NQ;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t lo = (uint32_t)reinterpret_cast<uintptr_t>(d.blk), hi = (uint32_t)(reinterpret_cast<uintptr_t>(d.blk) >> 32);
#pragma unroll
    for (int g = 0; g < DLN; ++g) {
        const uint32_t *blk = lane_ptr(lo, hi, g);
        const uint32_t pos = lane_u32(d.pos, g), nq = lane_u32(quads, g);
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            // a quad that is not needed re-reads quad 0 (same lines, cache hits) instead of branching around its loads:
            // the compiler closes every conditional block of loads with s_waitcnt vmcnt(0) — one round trip per quad
            uint32_t k = pos + ((uint32_t)q < nq ? 64u * q : 0u) + lane;            // < 624 + 256: one wrap
            k -= k >= (uint32_t)MT_N ? MT_N : 0;
            const uint32_t k1 = k + 1 == (uint32_t)MT_N ? 0 : k + 1;
            const uint32_t km = k + MT_M >= (uint32_t)MT_N ? k + MT_M - MT_N : k + MT_M;
            f.a[g][q] = blk[k]; f.b[g][q] = blk[k1]; f.c[g][q] = blk[km];
        }
    }
}
template <int W, int DLN>
__device__ __forceinline__ void coop_fill_park(LdsBulkDraws<W> &d, const CoopFillRegs<W, DLN> &f, int row_stride, uint32_t quads) {
    constexpr int NQ = CoopFillRegs<W, DLN>::NQ;
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t *rows = d.row - (lane < (uint32_t)DLN ? lane : 0u) * row_stride;      // row 0 of this wave
#pragma unroll
    for (int g = 0; g < DLN; ++g) {
        const uint32_t pos = lane_u32(d.pos, g), pretw = lane_u32(d.pretw, g), nq = lane_u32(quads, g);
        uint32_t *row = rows + g * row_stride;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            if (nq <= (uint32_t)q) continue;
            const uint32_t j = 64u * q + lane;
            const uint32_t y = mt_twist(f.a[g][q], f.b[g][q], f.c[g][q]), m = 0u - (uint32_t)(pos + j < pretw);
            const uint32_t v = (f.a[g][q] & m) | (y & ~m);
            if (j < (uint32_t)W) { row[j] = v; row[W + j] = mt_temper(v); }
        }
    }
    if (quads) { d.cur = 0; d.filled = true; d.avail = quads * 64u < (uint32_t)W ? quads * 64u : (uint32_t)W; }
}
template <int W, int DLN>
__device__ __forceinline__ void coop_flush(LdsBulkDraws<W> &d, int row_stride) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t lo = (uint32_t)reinterpret_cast<uintptr_t>(d.blk), hi = (uint32_t)(reinterpret_cast<uintptr_t>(d.blk) >> 32);
    uint32_t *rows = d.row - (lane < (uint32_t)DLN ? lane : 0u) * row_stride;
    const uint32_t curw = (d.filled && lane < (uint32_t)DLN) ? d.cur : 0u;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll 2
    for (int r = 0; r < DLN; ++r) {
        uint32_t *blk = lane_ptr(lo, hi, r);
        const uint32_t pos = lane_u32(d.pos, r), pretw = lane_u32(d.pretw, r);
        const uint32_t cur = lane_u32(curw, r);
        const uint32_t *row = rows + r * row_stride;
#pragma unroll
        for (int q = 0; q < (W + 63) / 64; ++q) {
            const uint32_t j = 64u * q + lane;
            uint32_t k = pos + j;
            if (j < cur && k >= pretw) {
                k -= k >= (uint32_t)MT_N ? MT_N : 0;
                mt_store(blk, k, row[j]);
            }
        }
    }
    if (d.filled) {
        uint32_t p = d.pos + d.cur;
        if (p >= (uint32_t)MT_N) { p -= MT_N; d.pretw = mt_wrap_ready(d.pretw); }
        d.pos = p; d.cur = 0; d.filled = false;
    }
}

// ------------------------------------------------------------------ terminal observations of fused SAME_STEP rollouts
// A fused rollout writes the RESET observation of an env that finishes at step t to slot t of its trajectory; the terminal one —
// what step() hands to final_obs_out — goes to a side buffer compacted PER SEGMENT: the envs one wave steps (64 consecutive envs; 16
// for traffic) own fin_cap consecutive rows, the wave keeps the segment's fill count in a register for the whole launch and writes it
// out once at the end.  No atomics and nothing a wave waits for: a returning atomic in the step loop would wait, like a load, for
// every observation store issued before it (one in-order memory counter per wave), and one counter for 16,384 waves saturates at
// ~90 updates per microsecond.
struct FinalSeg {
    void *rows;
    int64_t *index;
    int32_t *count;
    int64_t cap, n;
};
// The lanes of the wave that are active here AND have `fin` take consecutive slots after `used` (wave-uniform); returns the lane's
// global row index into rows / index (or -1: not a finishing lane, nothing registered, segment full) and records (t, env) for it.
// The caller adds wave_count(fin) to `used` where the wave is convergent again.
__device__ __forceinline__ int64_t final_slot(const FinalSeg &f, int64_t seg, uint32_t used, bool fin, int64_t t, int64_t i) {
    if (!f.rows) return -1;
    const unsigned long long m = __ballot(fin);
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t slot = used + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
    if (!fin || (int64_t)slot >= f.cap) return -1;
    const int64_t gs = seg * f.cap + slot;
    f.index[gs] = t * f.n + i;
    return gs;
}

// Where row r of a wave's staged tile goes: row r of the destination (dense), or — compacted terminal rows — the rows of `mask` in
// lane order, at most `room` of them (the caller points the destination at the segment's first free row).
struct RowMap {
    unsigned long long mask;
    int64_t nrows, room;
    bool compact;
    __device__ __forceinline__ bool row(uint32_t r, int64_t &out) const {
        if ((int64_t)r >= nrows || !((mask >> r) & 1ull)) return false;
        if (!compact) { out = (int64_t)r; return true; }
        const int64_t k = (int64_t)__popcll(mask & ((1ull << r) - 1ull));
        out = k;
        return k < room;
    }
};
// wave-convergent bookkeeping of one step's terminal rows for the tile-staged kernels: records (t, env) of the finishing lanes, returns
// the map and the destination of the segment's first free row; the caller stores the rows and adds popcount(mask) to `used`
template <class T>
__device__ __forceinline__ RowMap final_rows(const FinalSeg &f, int64_t seg, uint32_t used, bool fin, unsigned long long fin_mask, int64_t nrows, int64_t t,
                                             int64_t i, int64_t row_elems, T *&dst) {
    (void)final_slot(f, seg, used, fin, t, i);
    const int64_t room = f.cap - (int64_t)used;
    dst = static_cast<T *>(f.rows) + (seg * f.cap + (int64_t)used) * row_elems;
    return RowMap{fin_mask, nrows, room > 0 ? room : 0, true};
}

// ------------------------------------------------------------------ small register arrays
// Runtime-indexed register arrays go to scratch on hipcc; these helpers keep every index static
// (fully unrolled select chains) so the env state stays in VGPRs.
template <int W>
__host__ __device__ __forceinline__ uint32_t sel(const uint32_t (&a)[W], uint32_t idx) {
    // mask form on purpose: a ternary chain gets folded by LLVM into "select the ADDRESS, then load",
    // which pins the array in scratch memory
    uint32_t r = 0;
#pragma unroll
    for (int k = 0; k < W; ++k) r |= a[k] & (0u - (uint32_t)(idx == (uint32_t)k));
    return r;
}
template <int W>
__host__ __device__ __forceinline__ void or_word(uint32_t (&a)[W], uint32_t idx, uint32_t m) {
#pragma unroll
    for (int k = 0; k < W; ++k) a[k] |= (idx == (uint32_t)k) ? m : 0u;
}
template <int W>
__host__ __device__ __forceinline__ void andnot_word(uint32_t (&a)[W], uint32_t idx, uint32_t m) {
#pragma unroll
    for (int k = 0; k < W; ++k) a[k] &= (idx == (uint32_t)k) ? ~m : 0xffffffffu;
}

// Workgroup barrier that orders LDS traffic ONLY.  __syncthreads() also emits s_waitcnt vmcnt(0),
// which would stall every wave until its global stores (the previous obs tile) are acknowledged and
// until prefetched global loads land — exactly the latency the fused rollout is built to hide.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// ------------------------------------------------------------------ obs tile -> HBM
// Streams `bytes_valid` bytes (multiple of 4) of an LDS tile to `dst` with the widest stores the
// destination alignment allows.  Must be called after a barrier by BLOCK cooperating threads numbered tid = 0..BLOCK-1
// (the whole workgroup by default; snake's rollout uses a dedicated 64-lane writer wave).
// FULL = bytes of a complete tile: when the tile is complete and the destination 16-byte aligned (every workgroup but
// the last of a batch), the copy is a fixed number of (ds_read_b128, global_store_dwordx4) pairs at constant offsets —
// the generic loops below spend ~20 VALU per 16 bytes on 64-bit address arithmetic.
template <int BLOCK, int FULL = 0>
__device__ __forceinline__ void store_tile(const uint32_t *tile, int8_t *dst, uint32_t bytes_valid, uint32_t tid = threadIdx.x) {
    if (FULL > 0 && bytes_valid == (uint32_t)FULL && (reinterpret_cast<uintptr_t>(dst) & 15u) == 0) {
        constexpr int NVEC = FULL / 16, ITERS = NVEC / BLOCK, TAIL = NVEC % BLOCK;
        static_assert(FULL % 16 == 0, "whole tiles are a multiple of 16 bytes");
        const uint4 *t4 = reinterpret_cast<const uint4 *>(tile) + tid;
        uint4 *d4 = reinterpret_cast<uint4 *>(dst) + tid;
#pragma unroll
        for (int it = 0; it < ITERS; ++it) d4[it * BLOCK] = t4[it * BLOCK];
        if (TAIL && tid < (uint32_t)TAIL) d4[ITERS * BLOCK] = t4[ITERS * BLOCK];
        return;
    }
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


// ------------------------------------------------------------------ a lane's own obs row -> HBM in 16-byte stores
// NF consecutive float32 values of THIS lane's row -> row[col0 ..): 16-byte stores straight from registers (+ single dwords for a
// remainder).  Rows of the float envs are hundreds of bytes apart and only 4-byte aligned (gfx950 runs with unaligned access
// enabled: the compiler emits global_store_dwordx4 for these 4-byte-aligned 16-byte stores — checked on the ISA and on the
// device), so one store instruction touches 64 different lines; but consecutive instructions walk each row front to back, every
// 128-byte line is completed by 8 back-to-back stores of one lane, and the L2 hands whole lines to HBM.  No LDS staging, no
// transposition, no index arithmetic.  (Round 1 staged column chunks in LDS and wrote them dword by dword: for crypto 261 store
// instructions and ~3,000 VALU of index arithmetic per wave and step.)
struct __attribute__((packed, aligned(4))) Piece16 { uint32_t a, b, c, d; };
struct __attribute__((packed, aligned(4))) Piece8 { uint32_t a, b; };

template <int NF>
__device__ __forceinline__ void store_own_row(float *row, int col0, const float (&v)[NF], bool mine) {
    if (!mine) return;
#pragma unroll
    for (int q = 0; q + 4 <= NF; q += 4)
        *reinterpret_cast<Piece16 *>(row + col0 + q) = Piece16{__float_as_uint(v[q]), __float_as_uint(v[q + 1]), __float_as_uint(v[q + 2]), __float_as_uint(v[q + 3])};
#pragma unroll
    for (int q = NF - NF % 4; q < NF; ++q) row[col0 + q] = v[q];
}

// A wave-uniform base pointer plus a 32-bit per-lane byte offset: the global_store/load form with the base in SGPRs.  A kernel that
// keeps a 64-bit per-lane pointer per store site instead (13 sites per row, rows for obs and final_obs) spends ~50 VGPRs on addresses.
template <class T>
__device__ __forceinline__ T *at(void *ubase, uint32_t voff) { return reinterpret_cast<T *>(static_cast<char *>(ubase) + (size_t)voff); }
template <class T>
__device__ __forceinline__ const T *at(const void *ubase, uint32_t voff) { return reinterpret_cast<const T *>(static_cast<const char *>(ubase) + (size_t)voff); }

// ------------------------------------------------------------------ quad primitives: one env = the 4 lanes of a DPP quad (traffic, hospital)
template <int CTRL>
__device__ __forceinline__ uint32_t dpp(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true); }
constexpr int QL = 4;                   // lanes per env
template <int M>
__device__ __forceinline__ uint32_t gxor(uint32_t v) {                // value of lane (ql ^ M) of the same quad
    static_assert(M == 1 || M == 2, "a quad");
    if constexpr (M == 1) return dpp<0xB1>(v);                        // quad_perm [1, 0, 3, 2]
    else return dpp<0x4E>(v);                                         // quad_perm [2, 3, 0, 1]
}
template <int K>
__device__ __forceinline__ uint32_t gbcast(uint32_t v) { return dpp<K * 0x55>(v); }    // quad_perm [K, K, K, K]
__device__ __forceinline__ uint32_t gsum(uint32_t v) { v += gxor<1>(v); v += gxor<2>(v); return v; }
__device__ __forceinline__ uint32_t gor(uint32_t v) { v |= gxor<1>(v); v |= gxor<2>(v); return v; }
template <int M>
__device__ __forceinline__ double gxor_f64(double x) {
    uint64_t u;
    memcpy(&u, &x, 8);
    u = ((uint64_t)gxor<M>((uint32_t)(u >> 32)) << 32) | gxor<M>((uint32_t)u);
    memcpy(&x, &u, 8);
    return x;
}
template <int K>
__device__ __forceinline__ double gbcast_f64(double x) {
    uint64_t u;
    memcpy(&u, &x, 8);
    u = ((uint64_t)gbcast<K>((uint32_t)(u >> 32)) << 32) | gbcast<K>((uint32_t)u);
    memcpy(&x, &u, 8);
    return x;
}

// Streams `bytes` (a multiple of 8) of the wave's LDS image to `dst` (16-byte aligned, the image's place in HBM): lane l of the
// `nact` active lanes (the first nact of the wave) moves the 16-byte pieces l, l + nact, ... — one store instruction = nact x 16
// CONTIGUOUS bytes, whole 128-byte lines.  (Pieces stored from registers where they arise — 64 contiguous bytes per quad and
// instruction, every line completed by several instructions — cost 1.38x the bytes at the memory side (PMC WRITE_SIZE) and
// 40 of the 54 us of a 262,144-env rollout step; round 4, profiles/r04_traffic_store_pattern_ab.txt.)
template <int MAXW>                    // MAXW: words of the largest image (compile time): a full wave's pieces are read in one batch
__device__ __forceinline__ void stream_image(const uint32_t *__restrict__ img, void *__restrict__ dst, uint32_t bytes, uint32_t lane, uint32_t nact) {
    const uint32_t n16 = bytes >> 4;
    if (nact == 64u) {
        constexpr int J = (MAXW / 4 + 63) / 64;
        uint4 v[J];
#pragma unroll
        for (int j = 0; j < J; ++j) {                       // all LDS reads first (past the image's end: its last piece again), then the stores
            const uint32_t k = lane + 64u * j;
            v[j] = *reinterpret_cast<const uint4 *>(img + 4u * (k < n16 ? k : n16 - 1u));
        }
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const uint32_t k = lane + 64u * j;
            if (k < n16) *at<uint4>(dst, 16u * k) = v[j];
        }
    } else {
#pragma unroll 1
        for (uint32_t k = lane; k < n16; k += nact) *at<uint4>(dst, 16u * k) = *reinterpret_cast<const uint4 *>(img + 4u * k);
    }
    // the tail of an image that is not a whole number of 16-byte pieces (an odd number of 4-byte-multiple rows): 8, then 4 bytes
    if ((bytes & 8u) && lane == 0u) *at<uint2>(dst, 16u * n16) = *reinterpret_cast<const uint2 *>(img + 4u * n16);
    if ((bytes & 4u) && lane == 1u) *at<uint32_t>(dst, (bytes & ~7u) + 0u) = img[(bytes >> 2) - 1u];
}


// ------------------------------------------------------------------ the env's draw stream, group-cooperative
// Cursor of the env's MT19937 stream as the state record keeps it (cge_device.hpp: pos, pretw, old0), plus the LDS ring:
//   u   ring counter of the cursor (slot u & 63);  hi  ring counter up to which words are parked (a multiple of 16; hi - u <= 64).
//       An EMPTY ring is hi = u & ~15: hi - u is then minus the cursor's place in its 16-word unit, and the next unit parked is the one
//       the cursor stands in (ring units line up with the generator block's 16-word units).
// Everything here is GROUP-UNIFORM: the four lanes of an env hold the same values and take the same branches.
struct Cur { uint32_t pos, pretw, old0, u, hi; };

// Twists, in place, the words from `lo` (unwrapped: >= 624 means word lo - 624 of the next generation) to the end of lo's 32-word
// chunk; lane ql does words c0 + 8 ql .. + 7 (the last chunk of a generation holds 16 words: lanes 0 and 1).  Same contract as
// cge_device.hpp: mt_twist_chunk (words of the chunk below lo keep their values; old0 receives the outgoing generation's word 0).
__device__ __forceinline__ uint32_t twist_chunk_group(uint32_t *__restrict__ ublk, uint32_t boff, uint32_t lo, bool go, uint32_t &old0, uint32_t ql) {
    const uint32_t gen = lo >= (uint32_t)MT_N ? (uint32_t)MT_N : 0u, base = lo - gen, c0 = base & ~31u;
    const uint32_t len = (uint32_t)MT_N - c0 < 32u ? (uint32_t)MT_N - c0 : 32u;
    const uint32_t k0 = c0 + 8u * ql;
    uint32_t first = 0;
    if (go && 8u * ql < len) {
        uint32_t a[9], c[8];
#pragma unroll
        for (int q = 0; q < 8; q += 4) {
            const MtQuad v = *at<MtQuad>(ublk, boff + 4u * (k0 + (uint32_t)q));
            a[q] = v.a; a[q + 1] = v.b; a[q + 2] = v.c; a[q + 3] = v.d;
            uint32_t ci = k0 + (uint32_t)q + MT_M;                          // == 1 mod 4: 621 is the only piece that touches the mirror
            ci -= ci > (uint32_t)MT_N ? MT_N : 0;
            const MtQuad w = *at<MtQuad>(ublk, boff + 4u * ci);
            c[q] = w.a; c[q + 1] = w.b; c[q + 2] = w.c; c[q + 3] = w.d;
        }
        a[8] = *at<uint32_t>(ublk, boff + 4u * (k0 + 8u));                  // <= 624: the mirror of word 0
        first = a[0];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const uint32_t y = mt_twist(a[j], a[j + 1], c[j]);
            a[j] = k0 + (uint32_t)j < base ? a[j] : y;
        }
#pragma unroll
        for (int q = 0; q < 8; q += 4) {
            *at<MtQuad>(ublk, boff + 4u * (k0 + (uint32_t)q)) = MtQuad{a[q], a[q + 1], a[q + 2], a[q + 3]};
            if (k0 + (uint32_t)q < (uint32_t)MT_PAD) *at<MtQuad>(ublk, boff + 4u * ((uint32_t)MT_N + k0 + (uint32_t)q)) = MtQuad{a[q], a[q + 1], a[q + 2], a[q + 3]};
        }
    }
    const uint32_t w0 = gbcast<0>(first);
    if (go && c0 == 0u) old0 = w0;                                          // the previous generation's word 0 goes away now
    return gen + c0 + len;
}


// A quad's LDS draw ring for envs with MANY draws per step (hospital: 50-80 words): RW tempered, READY words per env, refilled 16
// words (one 16-byte load per lane) at a time from words the quad twisted ahead of the cursor.  Everything here is quad-uniform; the
// four lanes read the same LDS words.  prepare() (top of a step, and before an episode reset's ~165 draws) also twists ahead until
// FAR = 224 words from the cursor are ready, so a draw beyond the parked words (a burst: a mass-casualty event, a long rejection run) is
// ONE load from the generator block — no loop, no call, a dozen instructions per draw site.  A step that drew more than FAR words
// would read unready words: `ovf` is raised instead (sticky, the env's overflow flag) — rejection runs of that length have a
// probability below 1e-40, and the data-dependent draws (one death roll per critical patient waiting) are bounded far below it by
// the dynamics themselves (three deaths end the episode).
// (Rounds 1-3 refilled on demand through a noinline call: values live across a call must sit in callee-saved VGPRs, every other
// block of eight registers — ~100 live values took ~230 registers; inlining the refill at ~60 draw sites was 26,000 instructions.)
template <int RW>
struct QuadRing {
    static_assert(RW % 16 == 0, "whole 16-word units");
    static constexpr uint32_t FAR = 224;                        // words ahead of the cursor that prepare() makes ready (mt_make_ready's bound is 227)
    uint32_t *ring;       // this env's LDS row: RW words
    uint32_t *blk;        // this env's generator block
    Cur c;                // c.u / c.hi: ring SLOT of the cursor / of the first unparked word (0 .. RW-1)
    int32_t nvalid;       // parked words from the cursor on (<= 0: none; negative: minus the cursor's place inside its unit, ring empty)
    uint32_t ql, p;       // p: words consumed since the last prepare()
    bool ovf;

    __device__ __forceinline__ void init(uint32_t *row, uint32_t *block, uint32_t pos, uint32_t pretw, uint32_t quad_lane) {
        ring = row; blk = block; ql = quad_lane; c.pos = pos; c.pretw = pretw; c.old0 = 0; p = 0; ovf = false;
        restart();
    }
    __device__ __forceinline__ void restart() { c.u = c.pos & 15u; c.hi = 0; nvalid = -(int32_t)(c.pos & 15u); }
    __device__ __forceinline__ uint32_t slot(uint32_t j) const { uint32_t s = c.u + j; s -= s >= (uint32_t)RW ? (uint32_t)RW : 0u; return s; }
    __device__ __forceinline__ void twist_until(uint32_t end, bool want) {       // wave-convergent: words up to `end` (unwrapped) become ready
        bool twisted = false;
#pragma unroll 1
        while (__ballot(want && (c.pretw > c.pos ? c.pretw : c.pos) < end)) {
            const uint32_t lo = c.pretw > c.pos ? c.pretw : c.pos;
            const bool go = want && lo < end;
            const uint32_t t = twist_chunk_group(blk, 0u, lo, go, c.old0, ql);
            if (go) c.pretw = t;
            twisted = true;
        }
        if (twisted) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");        // the quad's stores before the quad's loads of the same words
    }
    // the consumed words leave, FAR words from the cursor are made ready, every free unit is parked again (wave-convergent)
    __device__ __forceinline__ void prepare(bool active = true) {
        if ((int32_t)p > nvalid) {                              // the step ran past the ring: its cursor moves, the ring starts over
            mt_advance(c.pos, c.pretw, p);
            restart();
        } else {
            c.u = slot(p); nvalid -= (int32_t)p;
            mt_advance(c.pos, c.pretw, p);
        }
        p = 0;
        twist_until(c.pos + FAR, active);
        // every free unit: all their loads first (one round trip for the up to RW / 16 of them), then temper and park
        {
            constexpr int NU = RW / 16;
            const int32_t nfree = active ? (RW - nvalid) / 16 : 0;     // (nvalid + (cursor's place in its unit) is a multiple of 16)
            uint4 v[NU];
#pragma unroll
            for (int k = 0; k < NU; ++k) {
                uint32_t gidx = c.pos + (uint32_t)(nvalid + 16 * (k < nfree ? k : 0));
                gidx -= gidx >= (uint32_t)MT_N ? MT_N : 0;
                gidx -= gidx >= (uint32_t)MT_N ? MT_N : 0;
                v[k] = *reinterpret_cast<const uint4 *>(blk + gidx + 4u * ql);                 // gidx is a multiple of 16 (a unit past the free ones: unit 0 again)
            }
#pragma unroll
            for (int k = 0; k < NU; ++k) {
                if (k < nfree) {
                    *reinterpret_cast<uint4 *>(ring + c.hi + 4u * ql) = make_uint4(mt_temper(v[k].x), mt_temper(v[k].y), mt_temper(v[k].z), mt_temper(v[k].w));
                    c.hi = c.hi + 16u == (uint32_t)RW ? 0u : c.hi + 16u;
                    nvalid += 16;
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    // word j places after the cursor when it is not parked: a ready word of the generator block
    __device__ __forceinline__ uint32_t far_word(uint32_t j) {
        if (j >= FAR) { ovf = true; return 0u; }                // (0 passes every rejection test: a flagged env cannot spin)
        uint32_t pos = c.pos + j;
        pos -= pos >= (uint32_t)MT_N ? MT_N : 0;
        return mt_temper(blk[pos]);
    }
    __device__ __forceinline__ uint32_t next() {
        const uint32_t j = p++;
        return (int32_t)j < nvalid ? ring[CGE_GX(20, slot(j), RW)] : far_word(j);
    }
    // look-ahead for draw sequences whose word OFFSETS can be computed up front
    __device__ __forceinline__ bool has(uint32_t n) const { return (int32_t)(p + n) <= nvalid; }
    __device__ __forceinline__ uint32_t peek(uint32_t j) const { return ring[CGE_GX(21, slot(p + j), RW)]; }
    __device__ __forceinline__ void skip(uint32_t n) { p += n; }
    __device__ __forceinline__ uint32_t randbelow(uint32_t n, int kbits) {   // CPython _randbelow_with_getrandbits
        uint32_t r = next() >> (32 - kbits);
        while (r >= n) r = next() >> (32 - kbits);
        return r;
    }
    __device__ __forceinline__ double random53() {
        const uint32_t a = next() >> 5, b = next() >> 6;
        return (a * 67108864.0 + b) / 9007199254740992.0;
    }
    // end of a launch: the cursor as the state record keeps it
    __device__ __forceinline__ void finish() { mt_advance(c.pos, c.pretw, p); p = 0; }
};

}  // namespace cge
