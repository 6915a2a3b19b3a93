// traffic.hip — batched TrafficManagementEnv for MI355X (gfx950): kernels + C ABI (include/cge_amd.h).
//
// Re-expresses /root/reference/traffic_management_env/ for N independent instances:
//   environment.py  reset :141-166, step :168-203, _apply_actions :205-220, _spawn_vehicles :222-249,
//                   _process_intersections :271-281, _remove_completed_vehicles :283-285,
//                   _calculate_reward :287-311, _get_observation :313-363
//   utils.py        TrafficLight.update/_advance_phase :79-97, set_phase :108-118, can_pass :99-106,
//                   Intersection.process_vehicles :141-163, generate_vehicle_route :174-193,
//                   get_neighboring_intersections :196-214, get_direction_between_intersections :230-248
// The reference's O(vehicles x intersections) np.sqrt loop (_update_vehicles :251-269, its CPU hot spot) is a
// semantic no-op and has no device counterpart; the per-env state collapses, for NI controlled intersections, to
//   NI lights (phase:2 timer:5), 4 NI queues (len:7 dest:7 wait:18), NI passed, NI total_wait, counters, RNG cursor, float64 total_reward.
//
// Round 4: ONE ENV = A GROUP OF L = 4 LANES (a DPP quad), 16 envs per wave (rounds 1-3: one lane per env, 58 state dwords + the
// observation in one lane's registers = 193 VGPRs, two waves per SIMD, a wave-step was one 28-us dependent chain).
//   * lane ql of a group owns intersections ql, ql + 4, ql + 8, ql + 12 (IPL = ceil(NI / 4) slots, a template parameter): one light,
//     four queue words, `passed`, `total_wait` per slot — 6 IPL + ~10 group-uniform scalars instead of 58 + 14;
//   * per-instance reductions (vehicles passed / waiting / queued, the lights that need a timer draw, np.var's pairwise tree) are
//     quad-permute DPP steps; NumPy's eight-accumulator tree IS the butterfly over two slots (pairwise_sum);
//   * the env's MT19937 words live, tempered, in a 64-word LDS ring per env next to four 64-bit ACCEPTANCE MASKS (top bit clear,
//     top two bits != 11, top bits < NI, top five bits < 26): every rejection loop of the reference (`_randbelow`) is then
//     "count trailing zeros of mask >> cursor" — the draw chain of a step (light timers, spawn test, start, route length, hops)
//     is straight-line code on group-uniform registers, no loops, no divergence inside a group; a chain that runs off its 32-word
//     view (rare) redoes the step's draws word by word (slow_draws);
//   * the ring is refilled 16 words (64 B, one 16-byte load per lane) at a time from words the group twisted ahead of the cursor,
//     32 words = one 128-byte line per cooperative twist (twist_chunk_group; layout and ready marks as cge_device.hpp: mt_twist_chunk);
//   * the observation rows leave through an LDS image of 8 consecutive rows that the wave streams out as whole 128-byte lines
//     (emit_staged / stream_image); stored as 16-byte pieces from the registers where they arise — 64 contiguous bytes per quad and
//     instruction, every line completed by several instructions — they cost 1.38x the bytes at the memory side and 40 of a step's
//     54 us (profiles/r04_traffic_store_pattern_ab.txt); only the rare terminal rows still go out piecewise (emit_row);
//   * the state is array-of-structs (one 8 + 6 NI dword record per env): a wave loads / stores 16 consecutive records;
//   * waves are dealt to the XCDs in contiguous runs (block b -> chunk (b % 8) * per_xcd + b / 8), so the 64-byte reward and
//     16-byte flag pieces that neighbouring waves write into one 128-byte line meet in ONE L2;
//   * NI is a run-time value (2..16) — the kernels are instantiated per IPL, not per layout.
// Integer dynamics are exact; the reward (np.var included, NumPy's pairwise order) is float64 in the reference's operation
// order; the average-wait quotients are float32 divisions of exactly representable integers (wait < 2^18, len < 2^7: the
// correctly rounded float32 quotient equals the float32 of the correctly rounded float64 quotient — a float32 tie point
// (2M+1)/2^k is either hit exactly or missed by more than 2^-53 relative), so obs and reward are bit-identical to the CPU.
#include <cstring>
#include <utility>
#include <vector>

#include "cge_device.hpp"
#include "cge_host.hpp"

namespace cge {
namespace traffic {

constexpr int L = 4;                    // lanes per env: a DPP quad
constexpr int EPW = 64 / L;             // envs per wave
constexpr int MAXNI = 16;
constexpr int BLOCK = 64;
constexpr int UNIT = 16;                // the LDS draw ring of an env is refilled 16 words (64 B) at a time
// ring words per env: a fused rollout keeps 64 (a step's chain reads a 32-word view, refills trail the cursor); a step() call parks
// the two units its one step reads.  Row stride = ring + masks (4 x ring bits) + pad: 16-byte aligned rows, and 8 consecutive envs
// start in 8 different banks (76 = 12 mod 32, 44 = 12 mod 32)
template <bool ROLLOUT> struct RingLay { static constexpr int RING = ROLLOUT ? 64 : 32, STRIDE = ROLLOUT ? 76 : 44; };
// queue word: len | dest << 7 | wait << 14 (len, dest <= max_vehicles <= 127; wait <= max_vehicles * max_steps < 2^18)
constexpr uint32_t QM = 127u;
constexpr int QS_DEST = 7, QS_WAIT = 14;
enum { NS_GREEN = 0, NS_YELLOW = 1, EW_GREEN = 2, EW_YELLOW = 3 };
enum { NORTH = 0, EAST = 1, SOUTH = 2, WEST = 3 };

constexpr int bitlen(int n) { int k = 0; while (n) { ++k; n >>= 1; } return k; }

// device record of one env (dwords): [0] m0 = timestep:16 | nveh:7 << 16 | needs_reset << 23, [1] m1 = mt_pos:10 | ready mark:5 << 10,
// [2] episodes, [3] mt_old0, [4..5] total_reward (float64), [6..7] 0, [8 + 4 i + d] queue word of intersection i, direction d,
// [8 + 4 NI + 2 i] light (phase:2 | timer:5 << 2) | passed << 8, [8 + 4 NI + 2 i + 1] total_wait; padded to whole 16-byte pieces
__host__ __device__ constexpr int rec_words(int ni) { return (8 + 6 * ni + 3) & ~3; }
constexpr int O_Q = 8;

struct Cfg {
    double spawn_rate;
    int32_t max_vehicles, max_steps;
    int32_t rows, cols;            // grid_size (environment.py:62): the grid the routes walk
    int32_t ni;                    // controlled intersections (2..16)
    uint32_t inv_cols;             // ceil(2^16 / cols): start / cols == (start * inv_cols) >> 16 for start < 16, cols <= 64
    int32_t start_bits;            // random.randint(0, NI-1): _randbelow(NI), k = NI.bit_length()
    int32_t hops, hop_bits;        // route_length = randint(2, min(5, NI)) -> 1 + _randbelow(hops), k = hops.bit_length()
};

struct Params {
    uint32_t *state;
    uint32_t *mt;
    int64_t n, env0;
    Cfg cfg;
    int32_t mode, recw, obsw;
    uint32_t nwaves, per_xcd;
    const int32_t *actions;
    const uint8_t *mask;
    float *obs, *final_obs, *reward;
    uint8_t *terminated, *truncated;
    int32_t k_steps;
    uint64_t a_seed;
    int64_t t0, obs_step_stride;
    double *reward_sum;
    int32_t *done_count;
    double *ep_ret;       // episode statistics (cge_traffic_episode_stats), nullable
    int32_t *ep_len;
    FinalSeg fin;         // fused rollout, SAME_STEP: terminal observations compacted per wave (cge_traffic_rollout_final_obs), rows nullable
};

// x[i % 4 slot i / 4] of the lane that owns element i (static i), for every lane of the group
template <int IPL, int I>
__device__ __forceinline__ double elem_f64(const double (&x)[IPL]) { return gbcast_f64<I % L>(x[I / L]); }

// np.add.reduce over n <= 16 float64 values held one per (slot, lane) — element i in slot i / 4 of lane i % 4 — in NumPy's
// pairwise order (numpy/core/src/umath/loops_utils.h.src: pairwise_sum, n < 128): fewer than 8 terms left to right; otherwise
// eight accumulators r[j] = a[j] (+= a[8 + j] for each whole further block of 8), res = ((r0+r1)+(r2+r3)) + ((r4+r5)+(r6+r7)),
// then the remaining n % 8 terms left to right.  r0..r3 are slot 0 of lanes 0..3, r4..r7 slot 1: each half of the tree is the
// two-step xor butterfly over the quad (a + b == b + a bit for bit, so every lane ends with the same sums).
template <int IPL>
__device__ __forceinline__ double pairwise_sum(const double (&x)[IPL], int n) {
    double acc;
    if (IPL < 2 || n < 8) {
        acc = elem_f64<IPL, 0>(x);
        if constexpr (IPL * L > 1) { if (n > 1) acc += elem_f64<IPL, 1>(x); }
        if constexpr (IPL * L > 2) { if (n > 2) acc += elem_f64<IPL, 2>(x); }
        if constexpr (IPL * L > 3) { if (n > 3) acc += elem_f64<IPL, 3>(x); }
        if constexpr (IPL * L > 4) { if (n > 4) acc += elem_f64<IPL, 4 % (IPL * L)>(x); }
        if constexpr (IPL * L > 5) { if (n > 5) acc += elem_f64<IPL, 5 % (IPL * L)>(x); }
        if constexpr (IPL * L > 6) { if (n > 6) acc += elem_f64<IPL, 6 % (IPL * L)>(x); }
        return acc;
    }
    if constexpr (IPL >= 2) {
        double r0 = x[0], r1 = x[1];
        if constexpr (IPL == 4) { if (n == 16) { r0 += x[2]; r1 += x[3]; } }
        r0 += gxor_f64<1>(r0); r0 += gxor_f64<2>(r0);
        r1 += gxor_f64<1>(r1); r1 += gxor_f64<2>(r1);
        acc = r0 + r1;
        if constexpr (IPL >= 3) {
            if (n < 16) {
                if (n > 8) acc += elem_f64<IPL, 8>(x);
                if (n > 9) acc += elem_f64<IPL, 9>(x);
                if (n > 10) acc += elem_f64<IPL, 10>(x);
                if (n > 11) acc += elem_f64<IPL, 11>(x);
                if constexpr (IPL == 4) {
                    if (n > 12) acc += elem_f64<IPL, 12 % (IPL * L)>(x);
                    if (n > 13) acc += elem_f64<IPL, 13 % (IPL * L)>(x);
                    if (n > 14) acc += elem_f64<IPL, 14 % (IPL * L)>(x);
                }
            }
        }
    }
    return acc;
}

// (Cur and twist_chunk_group, the quad's generator cursor and its cooperative chunk twist: cge_device.hpp)
template <bool ROLLOUT>
struct Draws {
    uint32_t *ring;       // this env's LDS row: [0, RING) tempered words, then the 4 acceptance masks (RING bits each, bit r = ring slot r)
    uint32_t *blk;        // generator block of the wave's first env (wave-uniform)
    uint32_t boff;        // byte offset of this env's block from there
    Cur c;
    uint32_t ql;
    uint32_t T0, N11, S, A;      // the masks of words [u, u + 32), valid words only
    uint32_t p;                  // words of the view consumed so far
    bool ovf;                    // a link ran off the view
    // a step() call parks what one step typically needs (two units: one 128-byte line); a rollout keeps the ring full
    static constexpr uint32_t MINV = ROLLOUT ? 49u : 17u;
    static constexpr uint32_t RING = RingLay<ROLLOUT>::RING, RMASK = RING - 1u;

    __device__ __forceinline__ uint32_t word(uint32_t j) const { return ring[(c.u + j) & RMASK]; }   // tempered word j places after the cursor
    __device__ __forceinline__ int32_t valid() const { return (int32_t)(c.hi - c.u); }             // parked words from the cursor on (<= 0: none)

    // parks the next unit of 16 generator words for the groups with `want` (ring slots of a fully consumed unit)
    __device__ __forceinline__ void park_unit(bool want, const Cfg &cfg) {
        // the unit's words, unwrapped: [pos + (hi - u), + 16); they must be ready (twisted).  Wave-convergent: while some group is
        // short, the groups that would be short within another unit twist their next chunk in the same round
        const uint32_t gend = c.pos + (c.hi - c.u) + UNIT;
        bool twisted = false;
#pragma unroll 1
        while (__ballot(want && (c.pretw > c.pos ? c.pretw : c.pos) < gend)) {
            const uint32_t lo = c.pretw > c.pos ? c.pretw : c.pos;
            const bool go = want && lo < gend + UNIT;
            const uint32_t t = twist_chunk_group(blk, boff, lo, go, c.old0, ql);
            if (go) c.pretw = t;
            twisted = true;
        }
        if (twisted) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");   // the quad's stores before the quad's loads of the same words
        if (want) {
            uint32_t gidx = c.pos + (c.hi - c.u);
            gidx -= gidx >= (uint32_t)MT_N ? MT_N : 0;
            const uint4 v = *at<uint4>(blk, boff + 4u * (gidx + 4u * ql));                    // gidx is a multiple of 16
            const uint32_t w[4] = {mt_temper(v.x), mt_temper(v.y), mt_temper(v.z), mt_temper(v.w)};
            uint32_t t0 = 0, n11 = 0, s = 0, a = 0;
            const uint32_t ssh = 32u - (uint32_t)cfg.start_bits;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                t0 |= (uint32_t)((w[b] >> 31) == 0u) << b;                  // top bit clear: _randbelow(2^k) halves
                n11 |= (uint32_t)((w[b] >> 30) != 3u) << b;                 // _randbelow(3), k = 2
                s |= (uint32_t)((w[b] >> ssh) < (uint32_t)cfg.ni) << b;     // _randbelow(NI)
                a |= (uint32_t)((w[b] >> 27) < 26u) << b;                   // _randbelow(26), k = 5
            }
            const uint32_t slot = c.hi & RMASK;
            *reinterpret_cast<uint4 *>(ring + slot + 4u * ql) = make_uint4(w[0], w[1], w[2], w[3]);
            const uint32_t x = gor((t0 | (n11 << 16)) << (4u * ql)), y = gor((s | (a << 16)) << (4u * ql));
            const uint32_t mine = ql == 0u ? x & 0xFFFFu : ql == 1u ? x >> 16 : ql == 2u ? y & 0xFFFFu : y >> 16;
            reinterpret_cast<uint16_t *>(ring + RING)[ql * (RING / 16u) + (slot >> 4)] = (uint16_t)mine;     // mask ql, unit slot / 16
            c.hi += UNIT;
        }
    }
    __device__ __forceinline__ void view() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const int32_t nv = valid();
        const uint32_t vm = nv >= 32 ? 0xFFFFFFFFu : (nv > 0 ? (1u << nv) - 1u : 0u);
        const uint32_t su = c.u & RMASK, sh = su & 31u;
        if constexpr (RING == 64u) {
            const uint4 m0 = *reinterpret_cast<const uint4 *>(ring + RING), m1 = *reinterpret_cast<const uint4 *>(ring + RING + 4);
            const bool up = su >= 32u;
            T0 = __builtin_amdgcn_alignbit(up ? m0.x : m0.y, up ? m0.y : m0.x, sh) & vm;
            N11 = __builtin_amdgcn_alignbit(up ? m0.z : m0.w, up ? m0.w : m0.z, sh) & vm;
            S = __builtin_amdgcn_alignbit(up ? m1.x : m1.y, up ? m1.y : m1.x, sh) & vm;
            A = __builtin_amdgcn_alignbit(up ? m1.z : m1.w, up ? m1.w : m1.z, sh) & vm;
        } else {                                                   // 32-word ring: the view is the mask word rotated to the cursor
            const uint4 m = *reinterpret_cast<const uint4 *>(ring + RING);
            T0 = __builtin_amdgcn_alignbit(m.x, m.x, sh) & vm;
            N11 = __builtin_amdgcn_alignbit(m.y, m.y, sh) & vm;
            S = __builtin_amdgcn_alignbit(m.z, m.z, sh) & vm;
            A = __builtin_amdgcn_alignbit(m.w, m.w, sh) & vm;
        }
    }
    __device__ __forceinline__ void consume() {                 // the p words the last chain took leave the stream
        c.u += p;
        mt_advance(c.pos, c.pretw, p);
        p = 0;
    }
    // kernel start, and after slow_draws: nothing parked
    __device__ __forceinline__ void init() { c.u = c.pos & 15u; c.hi = 0; p = 0; ovf = false; }
    // top of a step: the consumed words go, free units are refilled (wave-convergent), the view is rebuilt.  live: this group owns an env
    __device__ __forceinline__ void prepare(const Cfg &cfg, uint32_t minv = MINV) {
        consume();
#pragma unroll 1
        while (__ballot(valid() < (int32_t)minv)) park_unit(valid() < (int32_t)minv, cfg);
        view();
        ovf = false;
    }
    // one `_randbelow`: the first word at or after the cursor that `mask` accepts; returns its top `bits` bits.  Words it skips
    // are the reference's rejected draws.  No such word inside the view: ovf.
    __device__ __forceinline__ uint32_t link(uint32_t mask, uint32_t bits, bool act) {
        const uint32_t m = (uint32_t)((uint64_t)mask >> p);     // p <= 32
        const uint32_t ps = p + (uint32_t)__builtin_ctz(m | 0x80000000u);
        const uint32_t v = word(ps) >> (32u - bits);
        if (act) {
            if (m == 0u) ovf = true;
            else p = ps + 1u;
        }
        return v;
    }
};

// what the draws of one step decide (group-uniform): the timers of the lights that turned green, and the spawned vehicle
struct DrawOut {
    uint32_t tv[4];       // 5-bit values of randint(5, 30) - 5 for intersections 0..15 (6 per word, 4 in the last)
    uint32_t spawn;       // spawned | dir << 1 | dest << 3 | start << 4
};
__device__ __forceinline__ void set_tv(DrawOut &o, uint32_t i, uint32_t v, bool act) {
    const uint32_t wd = i / 6u, sh = (i - wd * 6u) * 5u;
#pragma unroll
    for (int k = 0; k < 4; ++k) o.tv[k] |= (act && wd == (uint32_t)k) ? v << sh : 0u;
}
__device__ __forceinline__ uint32_t get_tv(const DrawOut &o, uint32_t i) {
    const uint32_t wd = i / 6u, sh = (i - wd * 6u) * 5u;
    uint32_t w = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) w |= wd == (uint32_t)k ? o.tv[k] : 0u;
    return (w >> sh) & 31u;
}

// one hop of generate_vehicle_route's random walk (utils.py:174-214): random.choice over the valid neighbours in the order N, S, W, E
__device__ __forceinline__ void walk(uint32_t r, uint32_t vN, uint32_t vS, uint32_t vW, uint32_t &row, uint32_t &col, uint32_t &mv) {
    uint32_t idx = r;
    if (vN && idx == 0) mv = NORTH;
    else {
        idx -= vN;
        if (vS && idx == 0) mv = SOUTH;
        else {
            idx -= vS;
            mv = (vW && idx == 0) ? WEST : EAST;
        }
    }
    row += mv == SOUTH ? 1u : (mv == NORTH ? ~0u : 0u);
    col += mv == EAST ? 1u : (mv == WEST ? ~0u : 0u);
}

// The draws of one step, in the reference's order, as one straight-line chain on the view's acceptance masks:
//   TrafficLight.update (utils.py:79-97): random.randint(5, 30) for every light that enters a green phase, intersection order;
//   _spawn_vehicles (:222-249) if nveh < max_vehicles: random.random() < spawn_rate, randint(0, NI - 1), randint(2, min(5, NI)),
//   then one random.choice per hop.
template <bool ROLLOUT>
__device__ __forceinline__ DrawOut fast_draws(Draws<ROLLOUT> &d, const Cfg &cfg, uint32_t needmask, bool try_spawn) {
    DrawOut o{{0, 0, 0, 0}, 0};
    uint32_t nm = needmask;
#pragma unroll 1
    while (__ballot(nm != 0u)) {
        const bool act = nm != 0u;
        const uint32_t i = (uint32_t)__builtin_ctz(nm | 0x10000u);
        const uint32_t v = d.link(d.A, 5, act);
        set_tv(o, i, v, act);
        nm &= nm - 1u;
    }
    if (__ballot(try_spawn)) {
        // random.random(): two words
        const uint32_t avail = d.valid() < 32 ? (uint32_t)(d.valid() > 0 ? d.valid() : 0) : 32u;
        const uint32_t a = d.word(d.p) >> 5, b = d.word(d.p + 1u) >> 6;
        const double r = ((double)a * 67108864.0 + (double)b) / 9007199254740992.0;
        bool spawned = false;
        if (try_spawn) {
            if (d.p + 2u > avail) d.ovf = true;
            else { d.p += 2u; spawned = r < cfg.spawn_rate; }
        }
        if (__ballot(spawned)) {
            const uint32_t start = d.link(d.S, (uint32_t)cfg.start_bits, spawned);
            const uint32_t hmask = cfg.hops == 3 ? d.N11 : d.T0;                 // _randbelow(hops): hops 1, 2, 4 accept "top bit clear"
            const uint32_t hops = 1u + d.link(hmask, (uint32_t)cfg.hop_bits, spawned);
            const uint32_t rows = (uint32_t)cfg.rows, cols = (uint32_t)cfg.cols;
            const uint32_t srow = (start * cfg.inv_cols) >> 16, scol = start - srow * cols;
            uint32_t row = srow, col = scol, dir = EAST;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const bool act = spawned && (uint32_t)k < hops;
                if (__ballot(act)) {
                    const uint32_t vN = row > 0, vS = row + 1 < rows, vW = col > 0, vE = col + 1 < cols;   // utils.py:206 order N, S, W, E
                    const uint32_t cnt = vN + vS + vW + vE;                  // >= 1: the grid has rows * cols >= NI >= 2 cells
                    // random.choice(neighbours) = _randbelow(cnt): k = 3 for 4 (accept < 4: top bit clear), 2 for 3 (reject 11) and 2
                    // (top bit clear), 1 for 1 (top bit clear)
                    const uint32_t r2 = d.link(cnt == 3u ? d.N11 : d.T0, cnt == 4u ? 3u : (cnt == 1u ? 1u : 2u), act);
                    uint32_t mv, nrow = row, ncol = col;
                    walk(r2, vN, vS, vW, nrow, ncol, mv);
                    if (act) { row = nrow; col = ncol; if (k == 0) dir = mv; }     // utils.py:230-248 (route[0] -> route[1])
                }
            }
            const uint32_t dest = (row == srow && col == scol) ? 1u : 0u;       // destination == this intersection
            if (spawned) o.spawn = 1u | (dir << 1) | (dest << 3) | (start << 4);
        }
    }
    return o;
}

// The same draws word by word, for a group whose chain ran off its view (a light-timer burst after a reset, a long rejection
// run): rare.  Straight from the generator block through the serial stream of cge_device.hpp (MtStream: a ready word is a load, any
// other word is twisted on the spot — the four lanes of the quad store the same value), one loop around ONE "next word" site
// driven by a small state machine over the reference's draw sequence.  The ring is dropped; the next prepare() parks afresh.
struct SlowOut { uint32_t pos, pretw; DrawOut o; };
__device__ __forceinline__ SlowOut slow_draws(uint32_t *blk, uint32_t pos, uint32_t pretw, const Cfg &cfg, uint32_t needmask, bool try_spawn) {
    MtStream st(blk, pos, pretw);
    SlowOut out;
    out.o = DrawOut{{0, 0, 0, 0}, 0};
    enum { LIGHT, RAND_A, RAND_B, START, HOPS, HOP, DONE };
    uint32_t nm = needmask, stage = nm ? LIGHT : (try_spawn ? RAND_A : DONE);
    uint32_t ra = 0, start = 0, hops = 0, k = 0, srow = 0, scol = 0, row = 0, col = 0, dir = EAST;
    const uint32_t rows = (uint32_t)cfg.rows, cols = (uint32_t)cfg.cols;
#pragma unroll 1
    while (stage != DONE) {
        const uint32_t w = st.next();
        // CPython _randbelow_with_getrandbits(n): r = getrandbits(k); a word with r >= n is dropped and the stage repeats
        if (stage == LIGHT) {
            const uint32_t r = w >> 27;
            if (r < 26u) {
                set_tv(out.o, (uint32_t)__builtin_ctz(nm), r, true);
                nm &= nm - 1u;
                if (!nm) stage = try_spawn ? RAND_A : DONE;
            }
        } else if (stage == RAND_A) {
            ra = w >> 5; stage = RAND_B;
        } else if (stage == RAND_B) {
            const double r = ((double)ra * 67108864.0 + (double)(w >> 6)) / 9007199254740992.0;
            stage = r < cfg.spawn_rate ? START : DONE;
        } else if (stage == START) {
            const uint32_t r = w >> (32u - (uint32_t)cfg.start_bits);
            if (r < (uint32_t)cfg.ni) {
                start = r; srow = (start * cfg.inv_cols) >> 16; scol = start - srow * cols; row = srow; col = scol;
                stage = HOPS;
            }
        } else if (stage == HOPS) {
            const uint32_t r = w >> (32u - (uint32_t)cfg.hop_bits);
            if (r < (uint32_t)cfg.hops) { hops = 1u + r; k = 0; stage = HOP; }
        } else {
            const uint32_t vN = row > 0, vS = row + 1 < rows, vW = col > 0, vE = col + 1 < cols;
            const uint32_t cnt = vN + vS + vW + vE;
            const uint32_t r = w >> (32u - (cnt == 4u ? 3u : (cnt == 1u ? 1u : 2u)));
            if (r < cnt) {
                uint32_t mv;
                walk(r, vN, vS, vW, row, col, mv);
                if (k == 0) dir = mv;
                if (++k == hops) {
                    out.o.spawn = 1u | (dir << 1) | ((row == srow && col == scol) ? 8u : 0u) | (start << 4);
                    stage = DONE;
                }
            }
        }
    }
    out.pos = st.pos; out.pretw = st.pretw;
    return out;
}

// on-device phase clocks (-DCGE_TRAFFIC_TIMING builds only: tools/probes/traffic_timing.py); a TICK drains the wave's memory counters
#ifdef CGE_TRAFFIC_TIMING
__device__ unsigned long long g_timing[4096 * 16];
#define TICK(k) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); const unsigned long long now_ = wall_clock64(); \
    if (threadIdx.x == 0 && blockIdx.x < 4096) { g_timing[blockIdx.x * 16 + k] += now_ - t_last; } t_last = now_; } while (0)
#define TICK_DECL unsigned long long t_last = wall_clock64();
#else
#define TICK(k)
#define TICK_DECL
#endif

// ------------------------------------------------------------------ one env, spread over its quad
template <int IPL>
struct Env {
    uint32_t q[IPL][4];         // queue words of intersection 4 s + ql, directions N, E, S, W
    uint32_t lp[IPL], tw[IPL];  // light | passed << 8; total_wait
    // group-uniform
    uint32_t timestep, nveh, needs_reset, episodes;
    double total_reward;

    __device__ __forceinline__ void load(const uint32_t *__restrict__ rec, int ni, uint32_t ql, Cur &c) {
        const uint4 m = *reinterpret_cast<const uint4 *>(rec);
        const uint2 tr = *reinterpret_cast<const uint2 *>(rec + 4);
        timestep = m.x & 0xFFFFu; nveh = (m.x >> 16) & 127u; needs_reset = (m.x >> 23) & 1u;
        c.pos = m.y & 1023u; c.pretw = mt_ready_decode((m.y >> 10) & 31u);
        episodes = m.z; c.old0 = m.w;
        const uint64_t u = ((uint64_t)tr.y << 32) | tr.x;
        memcpy(&total_reward, &u, 8);
#pragma unroll
        for (int s = 0; s < IPL; ++s) {
            const int i = s * L + (int)ql;
            uint4 v = make_uint4(0, 0, 0, 0);
            uint2 w = make_uint2(0, 0);
            if (i < ni) {
                v = *reinterpret_cast<const uint4 *>(rec + O_Q + 4 * i);
                w = *reinterpret_cast<const uint2 *>(rec + O_Q + 4 * ni + 2 * i);
            }
            q[s][0] = v.x; q[s][1] = v.y; q[s][2] = v.z; q[s][3] = v.w;
            lp[s] = w.x; tw[s] = w.y;
        }
    }
    __device__ __forceinline__ void store(uint32_t *__restrict__ rec, int ni, uint32_t ql, const Cur &c) const {
        if (ql == 0u) {
            uint64_t u;
            memcpy(&u, &total_reward, 8);
            *reinterpret_cast<uint4 *>(rec) = make_uint4(timestep | (nveh << 16) | (needs_reset << 23),
                                                         c.pos | ((c.pretw > c.pos ? mt_ready_encode(c.pretw) : 0u) << 10), episodes, c.old0);
            *reinterpret_cast<uint4 *>(rec + 4) = make_uint4((uint32_t)u, (uint32_t)(u >> 32), 0u, 0u);
        }
#pragma unroll
        for (int s = 0; s < IPL; ++s) {
            const int i = s * L + (int)ql;
            if (i < ni) {
                *reinterpret_cast<uint4 *>(rec + O_Q + 4 * i) = make_uint4(q[s][0], q[s][1], q[s][2], q[s][3]);
                *reinterpret_cast<uint2 *>(rec + O_Q + 4 * ni + 2 * i) = make_uint2(lp[s], tw[s]);
            }
        }
    }
    // the same record into the wave's LDS image of its 16 records (rec = image + g * recw), streamed out by stream_image
    __device__ __forceinline__ void stage_record(uint32_t *__restrict__ rec, int ni, int recw, uint32_t ql, const Cur &c) const {
        if (ql == 0u) {
            uint64_t u;
            memcpy(&u, &total_reward, 8);
            *reinterpret_cast<uint4 *>(rec) = make_uint4(timestep | (nveh << 16) | (needs_reset << 23),
                                                         c.pos | ((c.pretw > c.pos ? mt_ready_encode(c.pretw) : 0u) << 10), episodes, c.old0);
            *reinterpret_cast<uint4 *>(rec + 4) = make_uint4((uint32_t)u, (uint32_t)(u >> 32), 0u, 0u);
        }
        if (ql == 1u) for (int k = O_Q + 6 * ni; k < recw; ++k) rec[k] = 0;       // the record's padding
#pragma unroll
        for (int s = 0; s < IPL; ++s) {
            const int i = s * L + (int)ql;
            if (i < ni) {
                *reinterpret_cast<uint4 *>(rec + O_Q + 4 * i) = make_uint4(q[s][0], q[s][1], q[s][2], q[s][3]);
                *reinterpret_cast<uint2 *>(rec + O_Q + 4 * ni + 2 * i) = make_uint2(lp[s], tw[s]);
            }
        }
    }
    __device__ __forceinline__ void reset() {                      // environment.py:141-166 (no draws)
#pragma unroll
        for (int s = 0; s < IPL; ++s) { lp[s] = 0; tw[s] = 0; q[s][0] = q[s][1] = q[s][2] = q[s][3] = 0; }   // TrafficLight(): NS_GREEN, timer 0
        timestep = 0; nveh = 0; needs_reset = 0; total_reward = 0.0;
    }
};

struct Totals { uint32_t tp, tw, tq; };

template <int IPL>
__device__ __forceinline__ Totals totals(const Env<IPL> &e) {
    uint32_t tp = 0, tws = 0, tq = 0;
#pragma unroll
    for (int s = 0; s < IPL; ++s) {
        tp += e.lp[s] >> 8; tws += e.tw[s];
#pragma unroll
        for (int dd = 0; dd < 4; ++dd) tq += e.q[s][dd] & QM;
    }
    return Totals{gsum(tp), gsum(tws), gsum(tq)};
}

// the four global features (:352-361), one per lane: [nveh, tw / max(tp, 1), tq / NI, tp / NI] in float64 (ONE division for the quad;
// features 1 and 2 are capped at 100 / 50 by cap_feature)
__device__ __forceinline__ double global_feature(uint32_t ql, uint32_t nveh, const Totals &t, int ni) {
    const uint32_t num = ql == 0u ? nveh : ql == 1u ? t.tw : ql == 2u ? t.tq : t.tp;
    const uint32_t den = ql == 0u ? 1u : ql == 1u ? (t.tp > 1u ? t.tp : 1u) : (uint32_t)ni;
    return (double)num / (double)den;
}
__device__ __forceinline__ double cap_feature(uint32_t ql, double v) {
    const double cap = ql == 1u ? 100.0 : 50.0;
    return (ql == 1u || ql == 2u) ? (v < cap ? v : cap) : v;
}

// _get_observation :313-363: this lane's 16-byte pieces of the env's row (its intersections' columns) + its global feature.
// rows: wave-uniform pointer to the row of the wave's first env; roff: byte offset of this env's row from there
template <int IPL>
__device__ __forceinline__ void emit_row(const Env<IPL> &e, float *__restrict__ rows, uint32_t roff, int ni, uint32_t ql, float glob, bool on) {
    if (!on) return;
#pragma unroll
    for (int s = 0; s < IPL; ++s) {
        const int i = s * L + (int)ql;
        if (i < ni) {
            const uint32_t ph = e.lp[s] & 3u;
            *at<Piece16>(rows, roff + 16u * (uint32_t)i) = Piece16{ph == 0u ? 0x3F800000u : 0u, ph == 1u ? 0x3F800000u : 0u,
                                                                    ph == 2u ? 0x3F800000u : 0u, ph == 3u ? 0x3F800000u : 0u};
            float ln[4], av[4];
#pragma unroll
            for (int dd = 0; dd < 4; ++dd) {
                const uint32_t qq = e.q[s][dd], len = qq & QM;
                ln[dd] = (float)(len < 20u ? len : 20u);
                const float a = len ? (float)(qq >> QS_WAIT) / (float)len : 0.0f;      // == float32(float64 quotient): see the file header
                av[dd] = a < 100.0f ? a : 100.0f;
            }
            *at<Piece16>(rows, roff + 16u * (uint32_t)(ni + i)) = Piece16{__float_as_uint(ln[0]), __float_as_uint(ln[1]), __float_as_uint(ln[2]), __float_as_uint(ln[3])};
            *at<Piece16>(rows, roff + 16u * (uint32_t)(2 * ni + i)) = Piece16{__float_as_uint(av[0]), __float_as_uint(av[1]), __float_as_uint(av[2]), __float_as_uint(av[3])};
            const uint32_t twc = e.tw[s] < 1000u ? e.tw[s] : 1000u;
            *at<Piece8>(rows, roff + 8u * (uint32_t)(6 * ni + i)) = Piece8{__float_as_uint((float)(e.lp[s] >> 8)), __float_as_uint((float)twc)};
        }
    }
    *at<float>(rows, roff + 4u * (uint32_t)(14 * ni) + 4u * ql) = glob;
}

struct __attribute__((aligned(8))) Half16 { uint32_t a, b, c, d; };      // a 16-byte piece at 8-byte alignment (LDS rows are 520 B apart)

// The observation rows of the wave's envs, through LDS: the quads write their pieces into an image of 8 consecutive rows (two
// passes: rows 0..7, rows 8..15 — a 16-row image would leave room for three waves per SIMD), the wave streams the image out.
// `block`: wave-uniform pointer to the first row of the wave; rows_live: rows of the wave that exist (16 but for the batch's tail)
template <int IPL, int MAXW>
__device__ __forceinline__ void emit_staged(const Env<IPL> &e, uint32_t *__restrict__ stage, float *__restrict__ block, int ni, uint32_t lane,
                                            float glob, uint32_t rows_live) {
    const uint32_t ql = lane & (uint32_t)(L - 1), g = lane / (uint32_t)L;
    const uint32_t obsw = (uint32_t)(14 * ni + 4);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        if (rows_live > 8u * h) {
            if ((g >> 3) == (uint32_t)h) {
                uint32_t *row = stage + (g & 7u) * obsw;
#pragma unroll
                for (int s = 0; s < IPL; ++s) {
                    const int i = s * L + (int)ql;
                    if (i < ni) {
                        const uint32_t ph = e.lp[s] & 3u;
                        *reinterpret_cast<Half16 *>(row + 4 * i) = Half16{ph == 0u ? 0x3F800000u : 0u, ph == 1u ? 0x3F800000u : 0u,
                                                                          ph == 2u ? 0x3F800000u : 0u, ph == 3u ? 0x3F800000u : 0u};
                        float ln[4], av[4];
#pragma unroll
                        for (int dd = 0; dd < 4; ++dd) {
                            const uint32_t qq = e.q[s][dd], len = qq & QM;
                            ln[dd] = (float)(len < 20u ? len : 20u);
                            const float a = len ? (float)(qq >> QS_WAIT) / (float)len : 0.0f;      // == float32(float64 quotient): see the file header
                            av[dd] = a < 100.0f ? a : 100.0f;
                        }
                        *reinterpret_cast<Half16 *>(row + 4 * (ni + i)) = Half16{__float_as_uint(ln[0]), __float_as_uint(ln[1]), __float_as_uint(ln[2]), __float_as_uint(ln[3])};
                        *reinterpret_cast<Half16 *>(row + 4 * (2 * ni + i)) = Half16{__float_as_uint(av[0]), __float_as_uint(av[1]), __float_as_uint(av[2]), __float_as_uint(av[3])};
                        const uint32_t twc = e.tw[s] < 1000u ? e.tw[s] : 1000u;
                        *reinterpret_cast<uint2 *>(row + 12 * ni + 2 * i) = make_uint2(__float_as_uint((float)(e.lp[s] >> 8)), __float_as_uint((float)twc));
                    }
                }
                row[14 * ni + (int)ql] = __float_as_uint(glob);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const uint32_t rows = rows_live - 8u * h < 8u ? rows_live - 8u * h : 8u;
            stream_image<MAXW>(stage, block + (size_t)(8 * h) * obsw, rows * obsw * 4u, lane, rows_live * (uint32_t)L < 64u ? rows_live * (uint32_t)L : 64u);
            asm volatile("" ::: "memory");
        }
    }
}

// NI_T: the number of controlled intersections when it is known at compile time (the layouts the reference's scripts build: 4, 9, 16),
// or 0: read from the config (any 2..16; IPL_T slots per lane)
// waves per SIMD asked of the register allocator = what the LDS (rings + staging image) admits (A/B knobs: tools/build_variant.sh)
#ifndef CGE_TRAFFIC_WAVES
#define CGE_TRAFFIC_WAVES 4
#endif
#ifndef CGE_TRAFFIC_WAVES_STEP
#define CGE_TRAFFIC_WAVES_STEP 5
#endif
constexpr int stage_words(int nimax) { return 8 * (14 * nimax + 4) > EPW * rec_words(nimax) ? 8 * (14 * nimax + 4) : EPW * rec_words(nimax); }
template <int NI_T, int IPL_T, bool ROLLOUT>
constexpr int waves_per_simd() {                  // min(the knob, what 160 KB of LDS admit of this instance's workgroups)
    constexpr int lds = 4 * (EPW * RingLay<ROLLOUT>::STRIDE + stage_words(NI_T ? NI_T : L * IPL_T));
    constexpr int fit = (160 * 1024 / lds) / 4, want = ROLLOUT ? CGE_TRAFFIC_WAVES : CGE_TRAFFIC_WAVES_STEP;
    return fit < want ? (fit < 1 ? 1 : fit) : want;
}
// The arguments of the rare paths (episode statistics, terminal rows, per-launch sums) are re-read from the kernel-argument
// segment where they are used instead of being held in SGPRs across the step loop (the loop's own uniform values — output
// pointers, config, hash constants — already fill the scalar file; 42 were being spilled to VGPR lanes and read back per step).
typedef const __attribute__((address_space(4))) Params *ColdArgs;
__device__ __forceinline__ ColdArgs cold_args() {
    ColdArgs kp = (ColdArgs)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(kp));
    return kp;
}
template <int NI_T, int IPL_T, bool ROLLOUT>
__global__ __launch_bounds__(BLOCK, (waves_per_simd<NI_T, IPL_T, ROLLOUT>())) void step_kernel(Params p) {
    constexpr int IPL = NI_T ? (NI_T + L - 1) / L : IPL_T;
    // LDS: the 16 envs' draw rings, and the staging image: 8 observation rows or the 16 state records, whichever is larger
    constexpr int NIMAX = NI_T ? NI_T : L * IPL_T, STAGE_W = stage_words(NIMAX);
    constexpr int RSTRIDE = RingLay<ROLLOUT>::STRIDE;
    __shared__ __attribute__((aligned(16))) uint32_t lds[EPW * RSTRIDE + STAGE_W];
    uint32_t *const stage = lds + EPW * RSTRIDE;
    const uint32_t lane = threadIdx.x, ql_ = lane & (uint32_t)(L - 1), g = lane / (uint32_t)L;
    const uint32_t chunk = (blockIdx.x & 7u) * p.per_xcd + (blockIdx.x >> 3);      // XCD x serves chunks [x per_xcd, (x + 1) per_xcd)
    const int64_t i0 = (int64_t)chunk * EPW, i = i0 + g;                          // i0: wave-uniform
    if (chunk >= p.nwaves || i >= p.n) return;                                    // whole groups leave; no barrier anywhere below
    Cfg cfg = p.cfg;
    if constexpr (NI_T != 0) {                                                    // fold everything that follows from NI
        cfg.ni = NI_T; cfg.start_bits = bitlen(NI_T); cfg.hops = (NI_T < 5 ? NI_T : 5) - 1; cfg.hop_bits = bitlen((NI_T < 5 ? NI_T : 5) - 1);
    }
    const int ni = cfg.ni, OBS = 14 * ni + 4;
    const uint32_t rows_live = p.n - i0 < EPW ? (uint32_t)(p.n - i0) : (uint32_t)EPW;
    uint32_t *rec = p.state + i * p.recw;
    Env<IPL> e;
    Draws<ROLLOUT> d;
    d.ring = lds + g * RSTRIDE; d.blk = p.mt + i0 * MT_STRIDE; d.boff = g * (uint32_t)(MT_STRIDE * 4); d.ql = ql_; d.p = 0; d.ovf = false;
    const uint32_t roff_ = g * (uint32_t)(OBS * 4);                               // this env's row in the wave's block of 16 rows
    e.load(rec, ni, ql_, d.c);
    d.init();
    const uint64_t key = ROLLOUT ? hash_env_key(p.a_seed, (uint64_t)(p.env0 + i)) : 0;
    double rsum = 0.0;
    int32_t dcount = 0;
    uint32_t fin_used = 0;                                                        // terminal rows this wave has delivered (its segment's fill)
    const int ksteps = ROLLOUT ? p.k_steps : 1;
    TICK_DECL
    TICK(0);                                                                      // record load, kernel start
#pragma unroll 1
    for (int t = 0; t < ksteps; ++t) {
        // Opaque to the optimizer: everything derived from the lane's place in its quad would otherwise be hoisted out of the step loop
        // (machine LICM does not price registers) — ~35 VGPRs of precomputed shifts and offsets held across the whole loop
        uint32_t ql = ql_, roff = roff_;
        asm volatile("" : "+v"(ql), "+v"(roff));
        d.ql = ql;
        double reward = 0.0;
        bool term = false, reset_now = false;
        const bool stepping = !(p.mode == CGE_AUTORESET_NEXT_STEP && e.needs_reset);
        if (!stepping) reset_now = true;
        Totals tot{0, 0, 0};
        double gfeat = 0.0;
        if (__ballot(stepping)) {
            d.prepare(cfg);
            TICK(1);                                                              // + the outputs of the previous step draining
            uint32_t a[IPL];
            if (p.actions) {
                const int32_t *ap = p.actions + ((int64_t)t * p.n + i) * ni;
#pragma unroll
                for (int s = 0; s < IPL; ++s) a[s] = s * L + (int)ql < ni ? (uint32_t)ap[s * L + ql] : 0u;
            } else {
#pragma unroll
                for (int s = 0; s < IPL; ++s) a[s] = hash_action_from_key(key, (uint64_t)(p.t0 + t), 3u, (uint32_t)(s * L) + ql);
            }
            if (stepping) e.timestep += 1;
            // _apply_actions :205-220, TrafficLight.update utils.py:79-97
            uint32_t need = 0;
#pragma unroll
            for (int s = 0; s < IPL; ++s) {
                uint32_t phase = e.lp[s] & 3u, timer = (e.lp[s] >> 2) & 31u;
                if (a[s] == 1u && phase != NS_GREEN) { phase = NS_GREEN; timer = 5; }
                else if (a[s] == 2u && phase != EW_GREEN) { phase = EW_GREEN; timer = 5; }
                int tm = (int)timer - 1;
                if (tm <= 0) {
                    phase = (phase + 1u) & 3u;
                    tm = 3;                                                          // yellow; a green phase draws its timer below
                    if (!(phase & 1u) && s * L + (int)ql < ni) need |= 1u << (s * L + (int)ql);
                }
                if (stepping) e.lp[s] = (e.lp[s] & ~127u) | phase | ((uint32_t)tm << 2);
            }
            TICK(2);                                                              // actions (hash) + lights
            const uint32_t needmask = stepping ? gor(need) : 0u;
            const bool try_spawn = stepping && e.nveh < (uint32_t)cfg.max_vehicles;     // returns BEFORE drawing when full
            DrawOut o = fast_draws<ROLLOUT>(d, cfg, needmask, try_spawn);
            if (__ballot(d.ovf)) {
                if (d.ovf) {
                    // nothing of the fast chain has been committed: the slow path redoes the step's draws from the step's cursor
                    const SlowOut so = slow_draws(at<uint32_t>(d.blk, d.boff), d.c.pos, d.c.pretw, cfg, needmask, try_spawn);
                    d.c.pos = so.pos; d.c.pretw = so.pretw; o = so.o;
                    d.init();
                }
            }
#pragma unroll
            for (int s = 0; s < IPL; ++s) {
                const uint32_t ii = (uint32_t)(s * L) + ql;
                if ((needmask >> ii) & 1u) e.lp[s] = (e.lp[s] & ~(31u << 2)) | ((5u + get_tv(o, ii)) << 2);      // random.randint(5, 30)
            }
            if (o.spawn & 1u) {                                                          // _spawn_vehicles :222-249
                const uint32_t start = o.spawn >> 4, dir = (o.spawn >> 1) & 3u;
                const uint32_t inc = 1u | (((o.spawn >> 3) & 1u) << QS_DEST);               // len += 1, dest += (destination == this intersection)
#pragma unroll
                for (int s = 0; s < IPL; ++s)
#pragma unroll
                    for (int dd = 0; dd < 4; ++dd) e.q[s][dd] += (start == (uint32_t)(s * L) + ql && dir == (uint32_t)dd) ? inc : 0u;
                e.nveh += 1;
            }
            TICK(3);                                                              // draws + their effects
            // process_vehicles utils.py:141-163
            uint32_t tp = 0, tws = 0, pk = 0;             // pk: queued | removed << 8
            uint32_t qt[IPL];
#pragma unroll
            for (int s = 0; s < IPL; ++s) {
                const uint32_t phase = e.lp[s] & 3u;
                qt[s] = 0;
                if (stepping) {
#pragma unroll
                    for (int dd = 0; dd < 4; ++dd) {
                        uint32_t &qq = e.q[s][dd];
                        const uint32_t len = qq & QM;
                        const bool pass = (dd == NORTH || dd == SOUTH) ? phase == NS_GREEN : phase == EW_GREEN;
                        if (len) {
                            if (pass) {
                                e.lp[s] += len << 8;
                                pk += ((qq >> QS_DEST) & QM) << 8;                  // reached destination -> removed
                                qq = 0;
                            } else {
                                qq += len << QS_WAIT;
                                e.tw[s] += len;
                            }
                        }
                        qt[s] += qq & QM;
                    }
                }
                tp += e.lp[s] >> 8; tws += e.tw[s]; pk += qt[s];
            }
            tp = gsum(tp); tws = gsum(tws); pk = gsum(pk);
            const uint32_t tq = pk & 0xFFu;
            if (stepping) e.nveh -= pk >> 8;
            tot = Totals{tp, tws, tq};
            // the division pass: lane 2's quotient tq / NI is also np.var's mean
            const double quot = global_feature(ql, e.nveh, tot, ni);
            gfeat = cap_feature(ql, quot);
            const double mean = gbcast_f64<2>(quot);
            // _calculate_reward :287-311; np.var over the NI queue totals (population variance)
            double x[IPL];
#pragma unroll
            for (int s = 0; s < IPL; ++s) { const double dv = (double)qt[s] - mean; x[s] = dv * dv; }
            double var = pairwise_sum<IPL>(x, ni);
            var = var / (double)ni;
            double r = 0.0;
            r += (double)tp * 1.0;
            r += (double)tws * -0.1;
            r += (double)tq * -0.05;
            r += 0.5 / (1.0 + var);
            if (stepping) {
                e.total_reward += r;
                reward = r;
                term = e.timestep >= (uint32_t)cfg.max_steps;
                if (term) {
                    e.episodes += 1;
                    if (ql == 0u) {
                        ColdArgs c = cold_args();
                        if (c->ep_ret) c->ep_ret[i] = e.total_reward;            // environment.py:189 accumulates it, reset() zeroes it (:150)
                        if (c->ep_len) c->ep_len[i] = (int32_t)e.timestep;
                    }
                    if (p.mode == CGE_AUTORESET_SAME_STEP) reset_now = true;
                    else if (p.mode == CGE_AUTORESET_NEXT_STEP) e.needs_reset = 1;
                }
            }
        }
        TICK(4);                                                                  // vehicles + reward
        const bool fin = term && reset_now;                     // terminal obs (SAME_STEP)
        const unsigned long long fin_mask = __ballot(fin && ql == 0u);
        if (fin_mask) {
            ColdArgs c = cold_args();
            if (!ROLLOUT) {
                if (c->final_obs) emit_row<IPL>(e, c->final_obs + i0 * OBS, roff, ni, ql, (float)gfeat, fin);
            } else if (c->fin.rows) {
                // the wave's terminal rows take the next slots of the wave's segment (fin_used: register, wave-uniform)
                const uint32_t slot = fin_used + (uint32_t)__popcll(fin_mask & ((1ull << (lane & ~3u)) - 1ull));
                const bool on = fin && (int64_t)slot < c->fin.cap;
                const int64_t gs = (int64_t)chunk * c->fin.cap + slot;
                emit_row<IPL>(e, static_cast<float *>(c->fin.rows) + (int64_t)chunk * c->fin.cap * OBS, (on ? slot : 0u) * (uint32_t)(OBS * 4), ni, ql, (float)gfeat, on);
                if (on && ql == 0u) c->fin.index[gs] = (int64_t)t * c->fin.n + i;
                fin_used += (uint32_t)__popcll(fin_mask);
            }
        }
        if (reset_now) { e.reset(); tot = Totals{0, 0, 0}; gfeat = 0.0; }
        if (!stepping) gfeat = 0.0;
        TICK(5);                                                                  // final obs / reset
#ifndef CGE_TRAFFIC_NOEMIT                                                        // (measurement knob: everything computed, no row stored)
        if (p.obs) emit_staged<IPL, 8 * (14 * NIMAX + 4)>(e, stage, p.obs + (int64_t)t * p.obs_step_stride + i0 * OBS, ni, lane, (float)gfeat, rows_live);
#else
        if (p.obs && gfeat == 12345.0) emit_row<IPL>(e, p.obs + (int64_t)t * p.obs_step_stride + i0 * OBS, roff, ni, ql, (float)gfeat, true);
#endif
        TICK(6);                                                                  // observation row
        if (ql == 0u) {
            if (ROLLOUT) {
                if (p.reward) *at<float>(p.reward + (int64_t)t * p.n + i0, g * 4u) = (float)reward;
                if (p.terminated) *at<uint8_t>(p.terminated + (int64_t)t * p.n + i0, g) = term ? 1 : 0;
            } else {
                *at<float>(p.reward + i0, g * 4u) = (float)reward;
                *at<uint8_t>(p.terminated + i0, g) = term ? 1 : 0;
                if (p.truncated) *at<uint8_t>(p.truncated + i0, g) = 0;
            }
        }
        rsum += reward;
        dcount += term ? 1 : 0;
#ifdef CGE_TRAFFIC_TIMING
        if (threadIdx.x == 0 && blockIdx.x < 4096) g_timing[blockIdx.x * 16 + 15] += 1;
#endif
    }
    d.consume();
    e.stage_record(stage + g * (uint32_t)p.recw, ni, p.recw, ql_, d.c);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    stream_image<EPW * rec_words(NIMAX)>(stage, p.state + i0 * p.recw, rows_live * (uint32_t)p.recw * 4u, lane, rows_live * (uint32_t)L);
    if (ROLLOUT && ql_ == 0u) {
        ColdArgs c = cold_args();
        if (c->reward_sum) c->reward_sum[i] = rsum;
        if (c->done_count) c->done_count[i] = dcount;
        if (c->fin.count && g == 0u) c->fin.count[chunk] = (int32_t)fin_used;
    }
}

// reset(mask) + the observation of every env: one lane per env (cold path)
__global__ __launch_bounds__(256) void reset_kernel(Params p) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= p.n) return;
    const int ni = p.cfg.ni;
    uint32_t *rec = p.state + i * p.recw;
    if (!p.mask || p.mask[i]) {
        rec[0] = 0;                                             // timestep, nveh, needs_reset; the generator cursor, episodes, old0 stay
        rec[4] = 0; rec[5] = 0;
        for (int k = O_Q; k < O_Q + 6 * ni; ++k) rec[k] = 0;
    }
    if (!p.obs) return;
    float *row = p.obs + i * p.obsw;
    uint32_t tp = 0, tws = 0, tq = 0;
    for (int k = 0; k < ni; ++k) {
        const uint32_t lp = rec[O_Q + 4 * ni + 2 * k], tw = rec[O_Q + 4 * ni + 2 * k + 1];
        tp += lp >> 8; tws += tw;
        for (int dd = 0; dd < 4; ++dd) {
            const uint32_t qq = rec[O_Q + 4 * k + dd], len = qq & QM;
            tq += len;
            row[4 * k + dd] = (lp & 3u) == (uint32_t)dd ? 1.0f : 0.0f;
            row[4 * ni + 4 * k + dd] = (float)(len < 20u ? len : 20u);
            const double avg = len ? (double)(qq >> QS_WAIT) / (double)len : 0.0;
            row[8 * ni + 4 * k + dd] = (float)(avg < 100.0 ? avg : 100.0);
        }
        row[12 * ni + 2 * k] = (float)(lp >> 8);
        row[12 * ni + 2 * k + 1] = (float)(tw < 1000u ? tw : 1000u);
    }
    const double v1 = (double)tws / (double)(tp > 1u ? tp : 1u), v2 = (double)tq / (double)ni;
    row[14 * ni] = (float)((rec[0] >> 16) & 127u);
    row[14 * ni + 1] = (float)(v1 < 100.0 ? v1 : 100.0);
    row[14 * ni + 2] = (float)(v2 < 50.0 ? v2 : 50.0);
    row[14 * ni + 3] = (float)((double)tp / (double)ni);
}

__global__ __launch_bounds__(256) void rewind_kernel(uint32_t *state, int64_t n, int recw) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    state[i * recw + 1] = 0;                                    // m1: the generator cursor and its ready mark
}

__global__ __launch_bounds__(256) void info_kernel(const uint32_t *__restrict__ state, int64_t n, int recw, int ni, int field, int idx,
                                                   int32_t *__restrict__ out, double *__restrict__ out64) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t *rec = state + i * recw;
    if (out64) {
        const uint64_t u = ((uint64_t)rec[5] << 32) | rec[4];
        double v;
        memcpy(&v, &u, 8);
        out64[i] = v;
        return;
    }
    int32_t v = 0;
    const uint32_t m0 = rec[0];
    switch (field) {
        case CGE_TRAFFIC_INFO_TIMESTEP: v = (int32_t)(m0 & 0xFFFFu); break;
        case CGE_TRAFFIC_INFO_NUM_VEHICLES: v = (int32_t)((m0 >> 16) & 127u); break;
        case CGE_TRAFFIC_INFO_LIGHT_PHASE: v = (int32_t)(rec[O_Q + 4 * ni + 2 * idx] & 3u); break;
        case CGE_TRAFFIC_INFO_LIGHT_TIMER: v = (int32_t)((rec[O_Q + 4 * ni + 2 * idx] >> 2) & 31u); break;
        case CGE_TRAFFIC_INFO_VEHICLES_PASSED: v = (int32_t)(rec[O_Q + 4 * ni + 2 * idx] >> 8); break;
        case CGE_TRAFFIC_INFO_TOTAL_WAITING_TIME: v = (int32_t)rec[O_Q + 4 * ni + 2 * idx + 1]; break;
        case CGE_TRAFFIC_INFO_QUEUE_LEN: v = (int32_t)(rec[O_Q + idx] & QM); break;
        case CGE_TRAFFIC_INFO_QUEUE_DEST: v = (int32_t)((rec[O_Q + idx] >> QS_DEST) & QM); break;
        case CGE_TRAFFIC_INFO_QUEUE_WAIT: v = (int32_t)(rec[O_Q + idx] >> QS_WAIT); break;
        case CGE_TRAFFIC_INFO_EPISODES: v = (int32_t)rec[2]; break;
        case CGE_TRAFFIC_INFO_NEEDS_RESET: v = (int32_t)((m0 >> 23) & 1u); break;
    }
    out[i] = v;
}

template <int NI_T, int IPL_T>
void launch_step(const Params &p, bool rollout, hipStream_t s) {
    const unsigned blocks = p.per_xcd * 8u;
    if (rollout) hipLaunchKernelGGL((step_kernel<NI_T, IPL_T, true>), dim3(blocks), dim3(BLOCK), 0, s, p);
    else hipLaunchKernelGGL((step_kernel<NI_T, IPL_T, false>), dim3(blocks), dim3(BLOCK), 0, s, p);
}

}  // namespace traffic
}  // namespace cge

using namespace cge;

struct cge_traffic : HandleBase {
    cge_traffic_config cfg{};
    int ni = 0, recw = 0, obsw = 0, ipl = 0;
    uint32_t *state = nullptr;
    uint32_t *mt = nullptr;
    char step_name[64], rollout_name[64];

    traffic::Params params() const {
        traffic::Params p{};
        p.state = state; p.mt = mt; p.n = n; p.env0 = env0;
        traffic::Cfg c{};
        c.spawn_rate = cfg.spawn_rate; c.max_vehicles = cfg.max_vehicles; c.max_steps = cfg.max_steps;
        c.rows = cfg.grid_rows; c.cols = cfg.grid_cols; c.ni = ni;
        c.inv_cols = (65536u + (uint32_t)cfg.grid_cols - 1u) / (uint32_t)cfg.grid_cols;
        c.start_bits = traffic::bitlen(ni);
        c.hops = (ni < 5 ? ni : 5) - 1;                        // route_length = randint(2, min(5, NI)) (utils.py:181)
        c.hop_bits = traffic::bitlen(c.hops);
        p.cfg = c;
        p.mode = cfg.autoreset_mode; p.recw = recw; p.obsw = obsw;
        p.nwaves = (uint32_t)((n + traffic::EPW - 1) / traffic::EPW);
        p.per_xcd = (p.nwaves + 7u) / 8u;
        p.ep_ret = ep_ret; p.ep_len = ep_len;
        return p;
    }
    void launch(const traffic::Params &p, bool rollout, hipStream_t s) const {
        switch (ni) {                                      // the layouts the reference's scripts build get their own instance
            case 4: traffic::launch_step<4, 1>(p, rollout, s); return;       // simple_test.py:71-76
            case 9: traffic::launch_step<9, 3>(p, rollout, s); return;       // config.py:7
            case 16: traffic::launch_step<16, 4>(p, rollout, s); return;     // USAGE_EXAMPLES.md:32-38
        }
        switch (ipl) {
            case 1: traffic::launch_step<0, 1>(p, rollout, s); break;
            case 2: traffic::launch_step<0, 2>(p, rollout, s); break;
            case 3: traffic::launch_step<0, 3>(p, rollout, s); break;
            default: traffic::launch_step<0, 4>(p, rollout, s); break;
        }
    }
};

// device record <-> the canonical record's fields (w: phase[ni], timer[ni], passed[ni], total_wait[ni], qlen, qdest, qwait [4 ni])
static void to_record(const uint32_t *rec, int ni, int32_t *hd, double *total_reward, int32_t *w, uint32_t *mt_pos, uint32_t *mt_pretw, uint32_t *mt_old0) {
    using namespace traffic;
    const uint32_t m0 = rec[0], m1 = rec[1];
    hd[0] = (int32_t)(m0 & 0xFFFFu); hd[1] = (int32_t)((m0 >> 16) & 127u); hd[2] = (int32_t)((m0 >> 23) & 1u); hd[3] = 0; hd[4] = (int32_t)rec[2]; hd[5] = 0;
    const uint64_t u = ((uint64_t)rec[5] << 32) | rec[4];
    memcpy(total_reward, &u, 8);
    *mt_pos = m1 & 1023u; *mt_pretw = mt_ready_decode((m1 >> 10) & 31u); *mt_old0 = rec[3];
    for (int k = 0; k < ni; ++k) {
        const uint32_t lp = rec[O_Q + 4 * ni + 2 * k];
        w[k] = (int32_t)(lp & 3u); w[ni + k] = (int32_t)((lp >> 2) & 31u); w[2 * ni + k] = (int32_t)(lp >> 8); w[3 * ni + k] = (int32_t)rec[O_Q + 4 * ni + 2 * k + 1];
    }
    for (int k = 0; k < 4 * ni; ++k) {
        const uint32_t qq = rec[O_Q + k];
        w[4 * ni + k] = (int32_t)(qq & QM); w[8 * ni + k] = (int32_t)((qq >> QS_DEST) & QM); w[12 * ni + k] = (int32_t)(qq >> QS_WAIT);
    }
}
static void from_record(const int32_t *hd, double total_reward, const int32_t *w, int ni, int recw, uint32_t *rec) {
    using namespace traffic;
    memset(rec, 0, (size_t)recw * 4);
    rec[0] = (uint32_t)hd[0] | ((uint32_t)hd[1] << 16) | ((uint32_t)(hd[2] & 1) << 23);
    // a CPython state: index >= 624 = nothing of this generation twisted yet; otherwise the whole generation is ready
    rec[1] = hd[3] >= MT_N ? 0u : ((uint32_t)hd[3] | (mt_ready_encode(MT_N) << 10));
    rec[2] = (uint32_t)hd[4];
    uint64_t u;
    memcpy(&u, &total_reward, 8);
    rec[4] = (uint32_t)u; rec[5] = (uint32_t)(u >> 32);
    for (int k = 0; k < ni; ++k) {
        rec[O_Q + 4 * ni + 2 * k] = ((uint32_t)w[k] & 3u) | (((uint32_t)w[ni + k] & 31u) << 2) | ((uint32_t)w[2 * ni + k] << 8);
        rec[O_Q + 4 * ni + 2 * k + 1] = (uint32_t)w[3 * ni + k];
    }
    for (int k = 0; k < 4 * ni; ++k) rec[O_Q + k] = ((uint32_t)w[4 * ni + k] & QM) | (((uint32_t)w[8 * ni + k] & QM) << QS_DEST) | ((uint32_t)w[12 * ni + k] << QS_WAIT);
}

extern "C" {

#ifdef CGE_TRAFFIC_TIMING
int cge_traffic_debug_timing(unsigned long long *out, int clear) {
    static unsigned long long all[4096 * 16];
    if (hipMemcpyFromSymbol(all, HIP_SYMBOL(traffic::g_timing), sizeof all) != hipSuccess) return 1;
    for (int k = 0; k < 16; ++k) out[k] = 0;
    for (int b = 0; b < 4096; ++b)
        for (int k = 0; k < 16; ++k) out[k] += all[b * 16 + k];
    if (clear) { memset(all, 0, sizeof all); if (hipMemcpyToSymbol(HIP_SYMBOL(traffic::g_timing), all, sizeof all) != hipSuccess) return 1; }
    return 0;
}
#endif

void cge_traffic_default_config(cge_traffic_config *c) {
    if (c) *c = cge_traffic_config{5, 5, 9, 50, 0.3, 1000, CGE_AUTORESET_NEXT_STEP};
}

int cge_traffic_create(const cge_traffic_config *cfg, int64_t n_envs, int device, int64_t env_index0, cge_traffic **out) {
    if (!cfg || !out || n_envs <= 0 || env_index0 < 0) return CGE_ERR_INVALID_ARG;
    *out = nullptr;
    if (cfg->grid_rows < 1 || cfg->grid_cols < 1 || cfg->grid_rows > 64 || cfg->grid_cols > 64 || cfg->num_intersections < 1)
        return CGE_ERR_INVALID_ARG;
    // environment.py:79: num_intersections = min(num_intersections, rows * cols).  2..16: route_length = randint(2, min(5, NI)) raises
    // for a single intersection (utils.py:181), and an env is spread over 16 (slot, lane) places
    const int cells = cfg->grid_rows * cfg->grid_cols;
    const int ni = cfg->num_intersections < cells ? cfg->num_intersections : cells;
    if (ni < 2 || ni > traffic::MAXNI) return CGE_ERR_UNSUPPORTED;
    // queue word: len:7 dest:7 wait:18; timestep 16 bits; passed 24 bits
    if (cfg->autoreset_mode < 0 || cfg->autoreset_mode > 2 || cfg->max_vehicles < 0 || cfg->max_vehicles > 127 || cfg->max_steps <= 0 ||
        cfg->max_steps > 65535 || (int64_t)cfg->max_vehicles * cfg->max_steps > 262143 || !(cfg->spawn_rate >= 0.0))
        return CGE_ERR_INVALID_ARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return CGE_ERR_NO_DEVICE;
    cge_traffic *h = new cge_traffic();
    h->cfg = *cfg; h->cfg.num_intersections = ni; h->n = n_envs; h->env0 = env_index0; h->device = device;
    h->ni = ni; h->recw = traffic::rec_words(ni); h->obsw = 14 * ni + 4; h->ipl = (ni + traffic::L - 1) / traffic::L;
    const bool own = ni == 4 || ni == 9 || ni == 16;
    snprintf(h->step_name, sizeof h->step_name, "cge::traffic::step_kernel<%d, %d, false>", own ? ni : 0, h->ipl);
    snprintf(h->rollout_name, sizeof h->rollout_name, "cge::traffic::step_kernel<%d, %d, true>", own ? ni : 0, h->ipl);
    DeviceGuard g(device);
    const size_t sb = (size_t)h->recw * 4 * n_envs, mb = (size_t)n_envs * MT_STRIDE * sizeof(uint32_t);
    hipError_t e;
    if ((e = hipMalloc(&h->state, sb)) != hipSuccess || (e = hipMalloc(&h->mt, mb)) != hipSuccess ||
        (e = hipMemset(h->state, 0, sb)) != hipSuccess) {     // all-zero state == a freshly reset env
        (void)hipFree(h->state); (void)hipFree(h->mt);
        delete h;
        return CGE_ERR_HIP;
    }
    h->device_bytes = sb + mb;
    e = launch_mt_seed(h->mt, MT_STRIDE, n_envs, nullptr, 0, env_index0, 0, nullptr);
    if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
    if (e != hipSuccess) {
        (void)hipFree(h->state); (void)hipFree(h->mt);
        delete h;
        return CGE_ERR_HIP;
    }
    *out = h;
    return CGE_OK;
}

int cge_traffic_destroy(cge_traffic *h) {
    if (!h) return CGE_ERR_INVALID_ARG;
    DeviceGuard g(h->device);
    (void)hipDeviceSynchronize();
    (void)hipFree(h->state); (void)hipFree(h->mt);
    delete h;
    return CGE_OK;
}

int cge_traffic_seed(cge_traffic *h, const uint64_t *seeds, uint64_t base_seed, void *stream) {
    if (!h) return CGE_ERR_INVALID_ARG;
    DeviceGuard g(h->device);
    CGE_TRY(h, launch_mt_seed(h->mt, MT_STRIDE, h->n, seeds, base_seed, h->env0, 0, as_stream(stream)));
    hipLaunchKernelGGL(traffic::rewind_kernel, dim3((unsigned)((h->n + 255) / 256)), dim3(256), 0, as_stream(stream), h->state, h->n, h->recw);
    CGE_TRY(h, hipGetLastError());
    return CGE_OK;
}

int cge_traffic_reset(cge_traffic *h, const uint8_t *mask, float *obs_out, void *stream) {
    if (!h) return CGE_ERR_INVALID_ARG;
    DeviceGuard g(h->device);
    traffic::Params p = h->params();
    p.mask = mask; p.obs = obs_out;
    hipLaunchKernelGGL(traffic::reset_kernel, dim3((unsigned)((h->n + 255) / 256)), dim3(256), 0, as_stream(stream), p);
    CGE_TRY(h, hipGetLastError());
    return CGE_OK;
}

int cge_traffic_step(cge_traffic *h, const int32_t *actions, float *obs_out, float *reward_out, uint8_t *terminated_out,
                     uint8_t *truncated_out, float *final_obs_out, void *stream) {
    if (!h) return CGE_ERR_INVALID_ARG;
    if (!actions || !obs_out || !reward_out || !terminated_out)
        return h->fail(CGE_ERR_INVALID_ARG, "cge_traffic_step: null actions/obs/reward/terminated pointer");
    DeviceGuard g(h->device);
    traffic::Params p = h->params();
    p.actions = actions; p.obs = obs_out; p.reward = reward_out; p.terminated = terminated_out; p.truncated = truncated_out;
    p.final_obs = final_obs_out; p.k_steps = 1;
    h->launch(p, false, as_stream(stream));
    h->last_kernel = h->step_name;
    CGE_TRY(h, hipGetLastError());
    return CGE_OK;
}

int cge_traffic_rollout(cge_traffic *h, int32_t k_steps, const int32_t *actions, uint64_t action_seed, int64_t t0, float *obs_out,
                        int64_t obs_step_stride, float *reward_traj_out, uint8_t *terminated_traj_out, double *reward_sum_out,
                        int32_t *done_count_out, void *stream) {
    if (!h) return CGE_ERR_INVALID_ARG;
    if (k_steps < 0 || obs_step_stride < 0 || (obs_step_stride != 0 && obs_step_stride < h->n * h->obsw))
        return h->fail(CGE_ERR_INVALID_ARG, "cge_traffic_rollout: bad k_steps / obs_step_stride");
    if (k_steps == 0) return CGE_OK;
    DeviceGuard g(h->device);
    traffic::Params p = h->params();
    p.k_steps = k_steps; p.actions = actions; p.a_seed = action_seed; p.t0 = t0; p.obs = obs_out; p.obs_step_stride = obs_step_stride;
    p.reward = reward_traj_out; p.terminated = terminated_traj_out; p.reward_sum = reward_sum_out; p.done_count = done_count_out;
    p.fin = FinalSeg{h->fin_rows, h->fin_index, h->fin_count, h->fin_cap, h->n};
    h->launch(p, true, as_stream(stream));
    h->last_kernel = h->rollout_name;
    CGE_TRY(h, hipGetLastError());
    return CGE_OK;
}

CGE_DEFINE_FINAL_OBS(traffic, float, traffic::EPW)

int cge_traffic_info(cge_traffic *h, int32_t field_id, int32_t index, int32_t *out, void *stream) {
    if (!h) return CGE_ERR_INVALID_ARG;
    const bool per_queue = field_id >= CGE_TRAFFIC_INFO_QUEUE_LEN && field_id <= CGE_TRAFFIC_INFO_QUEUE_WAIT;
    if (!out || field_id < 0 || field_id > CGE_TRAFFIC_INFO_NEEDS_RESET || index < 0 || index >= (per_queue ? 4 : 1) * h->ni)
        return h->fail(CGE_ERR_INVALID_ARG, "cge_traffic_info: bad field / index / null out");
    DeviceGuard g(h->device);
    hipLaunchKernelGGL(traffic::info_kernel, dim3((unsigned)((h->n + 255) / 256)), dim3(256), 0, as_stream(stream), h->state, h->n, h->recw, h->ni,
                       field_id, index, out, (double *)nullptr);
    CGE_TRY(h, hipGetLastError());
    return CGE_OK;
}

int cge_traffic_total_reward(cge_traffic *h, double *out, void *stream) {
    if (!h || !out) return CGE_ERR_INVALID_ARG;
    DeviceGuard g(h->device);
    hipLaunchKernelGGL(traffic::info_kernel, dim3((unsigned)((h->n + 255) / 256)), dim3(256), 0, as_stream(stream), h->state, h->n, h->recw, h->ni,
                       0, 0, (int32_t *)nullptr, out);
    CGE_TRY(h, hipGetLastError());
    return CGE_OK;
}

size_t cge_traffic_state_bytes(const cge_traffic *h) { return h ? 6 * 4 + 8 + (size_t)16 * h->ni * 4 + MT_N * 4 : 0; }

int cge_traffic_get_state(cge_traffic *h, void *host_buf, void *stream) {
    if (!h || !host_buf) return CGE_ERR_INVALID_ARG;
    DeviceGuard g(h->device);
    const int64_t n = h->n;
    const int recw = h->recw, ni = h->ni;
    std::vector<uint32_t> st((size_t)recw * n);
    std::vector<uint32_t> mt((size_t)n * MT_STRIDE);
    CGE_TRY(h, hipStreamSynchronize(as_stream(stream)));
    CGE_TRY(h, hipMemcpy(st.data(), h->state, st.size() * 4, hipMemcpyDeviceToHost));
    CGE_TRY(h, hipMemcpy(mt.data(), h->mt, mt.size() * 4, hipMemcpyDeviceToHost));
    const size_t rec = cge_traffic_state_bytes(h);
    for (int64_t i = 0; i < n; ++i) {
        uint8_t *p = (uint8_t *)host_buf + (size_t)i * rec;
        int32_t hd[6];
        int32_t *w = (int32_t *)(p + 32);
        double total_reward;
        uint32_t mt_pos, mt_pretw, mt_old0;
        to_record(&st[(size_t)i * recw], ni, hd, &total_reward, w, &mt_pos, &mt_pretw, &mt_old0);
        mt_export_cpython(&mt[(size_t)i * MT_STRIDE], mt_pos, mt_pretw, (uint32_t *)(w + 16 * ni), &hd[3], &mt_old0);
        memcpy(p, hd, 24);
        memcpy(p + 24, &total_reward, 8);
    }
    return CGE_OK;
}

int cge_traffic_set_state(cge_traffic *h, const void *host_buf, void *stream) {
    if (!h || !host_buf) return CGE_ERR_INVALID_ARG;
    DeviceGuard g(h->device);
    const int64_t n = h->n;
    const int recw = h->recw, ni = h->ni;
    std::vector<uint32_t> st((size_t)recw * n);
    std::vector<uint32_t> mt((size_t)n * MT_STRIDE, 0u);
    const size_t rec = cge_traffic_state_bytes(h);
    for (int64_t i = 0; i < n; ++i) {
        const uint8_t *p = (const uint8_t *)host_buf + (size_t)i * rec;
        int32_t hd[6];
        memcpy(hd, p, 24);
        const int32_t *w = (const int32_t *)(p + 32);
        bool ok = hd[0] >= 0 && hd[0] <= 65535 && hd[1] >= 0 && hd[1] <= 127 && hd[3] >= 0 && hd[3] <= MT_N;
        for (int k = 0; ok && k < ni; ++k) ok = w[k] >= 0 && w[k] <= 3 && w[ni + k] >= 0 && w[ni + k] <= 31 && w[2 * ni + k] >= 0 && w[2 * ni + k] <= 65535 && w[3 * ni + k] >= 0;
        for (int k = 0; ok && k < 4 * ni; ++k) ok = w[4 * ni + k] >= 0 && w[4 * ni + k] <= 127 && w[8 * ni + k] >= 0 && w[8 * ni + k] <= 127 && w[12 * ni + k] >= 0 && w[12 * ni + k] <= 262143;
        if (!ok) return h->fail(CGE_ERR_INVALID_ARG, "cge_traffic_set_state: malformed record (a field does not fit the device record)");
        double total_reward;
        memcpy(&total_reward, p + 24, 8);
        from_record(hd, total_reward, w, ni, recw, &st[(size_t)i * recw]);
        memcpy(&mt[(size_t)i * MT_STRIDE], w + 16 * ni, MT_N * 4);
        memcpy(&mt[(size_t)i * MT_STRIDE + MT_N], &mt[(size_t)i * MT_STRIDE], MT_PAD * 4);     // mirror words (cge_device.hpp)
    }
    CGE_TRY(h, hipStreamSynchronize(as_stream(stream)));
    CGE_TRY(h, hipMemcpy(h->state, st.data(), st.size() * 4, hipMemcpyHostToDevice));
    CGE_TRY(h, hipMemcpy(h->mt, mt.data(), mt.size() * 4, hipMemcpyHostToDevice));
    return CGE_OK;
}

size_t cge_traffic_device_bytes(const cge_traffic *h) { return h ? h->device_bytes : 0; }
int cge_traffic_episode_stats(cge_traffic *h, double *return_out, int32_t *length_out) {
    if (!h) return CGE_ERR_INVALID_ARG;
    h->ep_ret = return_out; h->ep_len = length_out;
    return CGE_OK;
}

const char *cge_traffic_last_error(const cge_traffic *h) { return h ? h->last_error.c_str() : "null handle"; }

const char *cge_traffic_last_kernel(const cge_traffic *h) { return h ? h->last_kernel.c_str() : ""; }

}  // extern "C"
