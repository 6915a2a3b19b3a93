// fleet.hip — batched FleetManagementEnv for MI355X (gfx950): kernels + C ABI (include/cge_amd.h).
//
// Re-expresses /root/reference/fleet_management_env/fleet_env.py for N independent instances, one lane per env:
//   reset :185-234, step :236-276, _execute_vehicle_action :278-329, _get_new_position :331-342,
//   _get_traffic_cost :349-361, _attempt_pickup :363-391, _attempt_dropoff :393-437, _attempt_refuel :439-448,
//   _generate_delivery_requests :450-514, _update_traffic :516-522, _update_weather :524-528,
//   _check_missed_deadlines :530-535, _is_terminated :537-553, _get_observation :555-593 (76 values).
// State per env: 40 dwords in 10 uint4 columns — 3 vehicles (cell, cargo, assignment; float64 fuel), up to 12
// deliveries in two dwords each (cells, urgency, vehicle requirement, window, deadline, pickup time), the 5x5
// traffic map at 2 bits per cell, weather index, counters and both generators' cursors.  Rewards are sums of
// small integers (exact); fuel is float64 in the reference's operation order -> obs and reward bit-identical.
// RNG: draws are rare in step() (traffic every 50 steps, weather every 100) and heavy in reset() (~220 words
// over two interleaved streams: NumPy-legacy masked randint / choice(p) / random and CPython random.choice over
// 144 cells).  step_kernel therefore draws NOTHING: it lists the 1-3 % of envs that need draws, and dense_kernel
// serves that list with full waves and one bulk LDS window per stream (LdsBulkDraws).
// The (N,76) float32 obs: every lane of step_kernel streams its own 304-byte row with 16-byte stores; dense_kernel
// stages its few rows in LDS ([DL][77]) and writes each with two 64-lane stores.
#include <cstring>
#include <vector>

#include "cge_device.hpp"
#include "cge_host.hpp"

namespace cge {
namespace fleet {

constexpr int OBS = 76;
constexpr int ROW = 77;
constexpr int MAXD = 12;
constexpr int COLS = 10;
// dense kernel: DL lanes (envs) per wave, one bulk window per stream sized so a whole reset normally fits
// (L: nd + <=12 x ~8 + 58 traffic words ~ 170; P: <=12 x ~4 ~ 48), odd LDS row strides
constexpr int DL = 8;                                          // 2 and 4 measured: same ~50 us chain per wave, the mass redraw step 1.5-2.4x slower
constexpr int WL = 224, WP = 96;
constexpr int LROW = 2 * WL + 1, PROW = 2 * WP + 1;            // raw + tempered copy (LdsBulkDraws), odd stride
using DrawsL = LdsBulkDraws<WL>;
using DrawsP = LdsBulkDraws<WP>;
constexpr int BLOCK = 64;

struct Params {
    uint4 *state;
    uint32_t *mtP, *mtL;
    int64_t n, env0;
    int32_t mode, max_steps;
    const int32_t *actions;
    const uint8_t *mask;
    float *obs, *final_obs, *reward;
    FinalSeg fin;          // rollouts (SAME_STEP): terminal rows compacted per segment of 64 envs (cge_fleet_rollout_final_obs); rows nullable
    uint8_t *terminated, *truncated;
    int32_t k_steps;
    uint64_t a_seed;
    int64_t t0, obs_step_stride;
    double *reward_sum;
    int32_t *done_count;
    double *ep_ret;       // episode statistics (cge_fleet_episode_stats), nullable
    int32_t *ep_len;
    uint8_t *done;        // step(): terminated | truncated (cge_fleet_done_mask), nullable
    int32_t t_index, accumulate;
    uint32_t *work_count;      // [3][NSUB] rotating counters of the deferred-work sub-lists, one per 64-byte line
    uint64_t *work_list;       // [3][NSUB][sub_cap] entries, see work_entry()
    int64_t sub_cap;
    int32_t parity;            // the list this launch APPENDS to; a pipelined dense launch serves list (parity + 2) % 3, any other `parity` itself
    uint32_t seq;              // Env::seq of an env that has not taken this launch's step yet
};

enum : uint32_t { W_TRAFFIC = 1u, W_WEATHER = 2u, W_RESET = 4u };

// The work list is NSUB sub-lists, block b appends to sub-list b % NSUB.  One list with one counter had every wave that
// lists anything (86 % of them at 3 % of the lanes) queue on the same L2 atomic unit: step_kernel went from 22 us to
// 40 us per 131,072-env step once episodes began to end (profiles/r02_fleet_summary.txt, the kernel trace).
// entry: env index << 25 | L cursor (pos:10, pretw flag) << 14 | P cursor << 3 | W_* flags.  The cursors ride along so that
// dense_kernel can address the generator blocks straight from the entry, in the same round trip as the env record.
__device__ __forceinline__ uint64_t work_entry(int64_t i, uint32_t work, uint32_t lpos, uint32_t lpretw, uint32_t ppos, uint32_t ppretw) {
    return ((uint64_t)i << 25) | ((uint64_t)(lpos | (lpretw ? 1024u : 0u)) << 14) | ((uint64_t)(ppos | (ppretw ? 1024u : 0u)) << 3) | work;
}
constexpr int NSUB = 64;
constexpr int CNT_STRIDE = 16;                                  // dwords between counters
__device__ __forceinline__ uint32_t *work_counter(uint32_t *base, int parity, uint32_t sub) { return base + ((uint32_t)parity * NSUB + sub) * CNT_STRIDE; }
__device__ __forceinline__ uint64_t *work_sublist(const Params &p, int parity, uint32_t sub) { return p.work_list + ((int64_t)parity * NSUB + sub) * p.sub_cap; }

__device__ __forceinline__ double vrange(int k) { return k == 0 ? 80.0 : k == 1 ? 120.0 : 60.0; }     // :128-132
__device__ __forceinline__ double vcons(int k) { return k == 0 ? 1.0 : k == 1 ? 0.5 : 2.0; }
__device__ __forceinline__ uint32_t vcap(int k) { return k == 0 ? 3u : k == 1 ? 1u : 5u; }

struct Env {
    uint32_t veh[3];        // x:5 | y:5 << 5 | cargo:3 << 10 | (assigned+1):4 << 13
    double fuel[3];
    uint32_t dA[MAXD];      // px:5 | py:5<<5 | dx:5<<10 | dy:5<<15 | urgency:2<<20 | required:2<<22 | completed<<24 | (assigned_vehicle+1):2<<25
    uint32_t dB[MAXD];      // t0:8 | commercial<<8 | deadline:9<<9 | pickup_time:10<<18
    uint32_t traffic[2];    // 25 cells x 2 bits, row-major
    uint32_t timestep, nd, weather, needs_reset, missed, completed, episodes, ppos, ppretw, lpos, lpretw;
    uint32_t pending;       // on a work list: its step is not complete until a dense launch has served it (pipelined rollouts skip it meanwhile)
    uint32_t seq;           // steps taken, mod 16 — the same for every env of a handle between launches (each launch steps each env once)
    double total_reward;

    __host__ __device__ __forceinline__ void unpack(const uint32_t *raw) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            veh[k] = raw[k];
            const uint64_t u = ((uint64_t)raw[4 + 2 * k] << 32) | raw[3 + 2 * k];
            memcpy(&fuel[k], &u, 8);
        }
#pragma unroll
        for (int i = 0; i < MAXD; ++i) { dA[i] = raw[9 + i]; dB[i] = raw[21 + i]; }
        traffic[0] = raw[33]; traffic[1] = raw[34];
        const uint32_t m0 = raw[35], m1 = raw[36], m2 = raw[37];
        timestep = m0 & 1023u; nd = (m0 >> 10) & 15u; weather = (m0 >> 14) & 3u; needs_reset = (m0 >> 16) & 1u; missed = m0 >> 17;
        completed = m1 & 15u; episodes = (m1 >> 4) & 0xFFFFu; ppos = (m1 >> 20) & 1023u; ppretw = (m1 & (1u << 30)) ? (uint32_t)MT_N : 0u;
        lpos = m2 & 1023u; lpretw = (m2 & 1024u) ? (uint32_t)MT_N : 0u; pending = (m2 >> 11) & 1u; seq = (m2 >> 12) & 15u;
        const uint64_t u = ((uint64_t)raw[39] << 32) | raw[38];
        memcpy(&total_reward, &u, 8);
    }
    __host__ __device__ __forceinline__ void pack(uint32_t *raw) const {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            raw[k] = veh[k];
            uint64_t u;
            memcpy(&u, &fuel[k], 8);
            raw[3 + 2 * k] = (uint32_t)u; raw[4 + 2 * k] = (uint32_t)(u >> 32);
        }
#pragma unroll
        for (int i = 0; i < MAXD; ++i) { raw[9 + i] = dA[i]; raw[21 + i] = dB[i]; }
        raw[33] = traffic[0]; raw[34] = traffic[1];
        raw[35] = timestep | (nd << 10) | (weather << 14) | (needs_reset << 16) | (missed << 17);
        raw[36] = completed | ((episodes & 0xFFFFu) << 4) | (ppos << 20) | (ppretw ? (1u << 30) : 0u);
        raw[37] = lpos | (lpretw ? 1024u : 0u) | (pending << 11) | (seq << 12);
        uint64_t u;
        memcpy(&u, &total_reward, 8);
        raw[38] = (uint32_t)u; raw[39] = (uint32_t)(u >> 32);
    }
    __device__ __forceinline__ void load(const uint4 *__restrict__ s, int64_t n, int64_t i) {
        uint32_t raw[COLS * 4];
#pragma unroll
        for (int c = 0; c < COLS; ++c) {
            const uint4 v = s[(int64_t)c * n + i];
            raw[4 * c] = v.x; raw[4 * c + 1] = v.y; raw[4 * c + 2] = v.z; raw[4 * c + 3] = v.w;
        }
        unpack(raw);
    }
    __device__ __forceinline__ void store(uint4 *__restrict__ s, int64_t n, int64_t i) const {
        uint32_t raw[COLS * 4];
        pack(raw);
#pragma unroll
        for (int c = 0; c < COLS; ++c) s[(int64_t)c * n + i] = make_uint4(raw[4 * c], raw[4 * c + 1], raw[4 * c + 2], raw[4 * c + 3]);
    }
    // traffic_grid[r, c], 2 bits per cell
    __device__ __forceinline__ uint32_t traffic_at(uint32_t cell) const {
        const uint32_t w = (traffic[0] & (0u - (uint32_t)(cell < 16u))) | (traffic[1] & (0u - (uint32_t)(cell >= 16u)));   // mask form: no select-of-loads
        return (w >> ((cell & 15u) * 2u)) & 3u;
    }
};

// NumPy legacy helpers on the L stream
__device__ __forceinline__ uint32_t np_randint(DrawsL &L, uint32_t lo, uint32_t hi) {   // masked rejection on 32-bit words
    const uint32_t rng = hi - lo - 1u;
    uint32_t mask = rng;
    mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
    uint32_t v = L.next() & mask;
    while (v > rng) v = L.next() & mask;
    return lo + v;
}
// choice(n, p): searchsorted(cumsum(p)/sum, random_sample(), 'right'); the normalised cdf values are passed in
__device__ __forceinline__ uint32_t np_choice_cdf(DrawsL &L, double c0, double c1, double c2, double c3) {
    const double u = L.random53();
    return (uint32_t)(c0 <= u) + (uint32_t)(c1 <= u) + (uint32_t)(c2 <= u) + (uint32_t)(c3 <= u);
}

__device__ __forceinline__ void update_traffic(Env &e, DrawsL &L) {                      // :516-522
    // cdf of p=[0.6,0.3,0.1]: cumsum then / last, evaluated in float64 exactly as NumPy does
    const double a0 = 0.6, a1 = a0 + 0.3, a2 = a1 + 0.1;
    const double c0 = a0 / a2, c1 = a1 / a2, c2 = a2 / a2;
    uint32_t t0 = 0, t1 = 0;
#pragma unroll 1
    for (int cell = 0; cell < 25; ++cell) {
        const double u = L.random53();
        const uint32_t v = (uint32_t)(c0 <= u) + (uint32_t)(c1 <= u) + (uint32_t)(c2 <= u);
        if (cell < 16) t0 |= v << (cell * 2); else t1 |= v << ((cell - 16) * 2);
    }
    const double b0 = 0.4, b1 = b0 + 0.6;
    const double d0 = b0 / b1, d1 = b1 / b1;
#pragma unroll
    for (int q = 0; q < 4; ++q) {                          // traffic_grid[0:2, 3:5], row-major: cells 3, 4, 8, 9
        const int cell = q == 0 ? 3 : q == 1 ? 4 : q == 2 ? 8 : 9;
        const double u = L.random53();
        const uint32_t v = 1u + (uint32_t)(d0 <= u) + (uint32_t)(d1 <= u);
        t0 = (t0 & ~(3u << (cell * 2))) | (v << (cell * 2));
    }
    e.traffic[0] = t0; e.traffic[1] = t1;
}

// The same redraw for up to DL envs of a dense wave at once: lane c < 29 serves draw c (two window words each) of one row
// at a time, the 2-bit levels are OR-ed together in LDS (`tr`, two words per row).  Rows whose window cannot hold the 58
// words take the serial path above.
__device__ __forceinline__ void coop_update_traffic(Env &e, DrawsL &L, bool need, uint32_t *__restrict__ rows, uint32_t *__restrict__ tr) {
    const uint32_t lane = threadIdx.x & 63u;
    const bool fits = L.filled && L.cur + 58u <= L.avail;
    if (need && !fits) update_traffic(e, L);
    const bool coop = need && fits;
    const unsigned long long m = __ballot(coop);
    if (!m) return;
    if (coop) { tr[2 * lane] = 0; tr[2 * lane + 1] = 0; }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const double a0 = 0.6, a1 = a0 + 0.3, a2 = a1 + 0.1;
    const double c0 = a0 / a2, c1 = a1 / a2, c2 = a2 / a2;
    const double b0 = 0.4, b1 = b0 + 0.6;
    const double d0 = b0 / b1, d1 = b1 / b1;
    const bool hot = lane >= 25u;                                                   // draws 25..28 -> cells 3, 4, 8, 9
    const uint32_t cell = hot ? (lane == 25u ? 3u : lane == 26u ? 4u : lane == 27u ? 8u : 9u) : lane;
    const bool mine = lane < 29u && (hot || !(cell == 3u || cell == 4u || cell == 8u || cell == 9u));
#pragma unroll 1
    for (int r = 0; r < DL; ++r) {
        if (!((m >> r) & 1ull)) continue;
        const uint32_t *t = rows + r * LROW + WL + lane_u32(L.cur, r);
        if (lane < 29u) {
            const uint32_t a = t[2 * lane] >> 5, b = t[2 * lane + 1] >> 6;
            const double u = (a * 67108864.0 + b) / 9007199254740992.0;
            const uint32_t v = hot ? 1u + (uint32_t)(d0 <= u) + (uint32_t)(d1 <= u) : (uint32_t)(c0 <= u) + (uint32_t)(c1 <= u) + (uint32_t)(c2 <= u);
            if (mine) atomicOr(&tr[2 * r + (cell >= 16u ? 1 : 0)], v << ((cell & 15u) * 2u));
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (coop) { e.traffic[0] = tr[2 * lane]; e.traffic[1] = tr[2 * lane + 1]; L.cur += 58u; }
}

__device__ __forceinline__ uint32_t zone_cell(uint32_t zone, uint32_t k) {                    // customer_zones :135-140 -> x | y << 5
    const uint32_t i = k / 12u, j = k - i * 12u;
    const uint32_t x = (zone == 1u || zone == 3u) ? 13u + i : i, y = (zone == 2u || zone == 3u) ? 13u + j : j;
    return x | (y << 5);
}

// One delivery request (:456-512), the general path: every draw is a serial LDS read behind a window check.
__device__ __forceinline__ void request_serial(DrawsP &P, DrawsL &L, uint32_t &pz, uint32_t &dz, uint32_t &pc, uint32_t &dc, uint32_t &urg, uint32_t &req,
                                               uint32_t &t0) {
    const double u0 = 0.3, u1 = u0 + 0.4, u2 = u1 + 0.2, u3 = u2 + 0.1;
    pz = np_randint(L, 0u, 4u); dz = np_randint(L, 0u, 4u);                                 // np.random.choice(list(CustomerZone))
    pc = zone_cell(pz, P.randbelow(144u, 8));                                               // random.choice(positions)
    dc = zone_cell(dz, P.randbelow(144u, 8));
    while (dc == pc) dc = zone_cell(dz, P.randbelow(144u, 8));
    urg = np_choice_cdf(L, u0 / u3, u1 / u3, u2 / u3, u3 / u3);
    req = 0;                                                                                // 0 none, 1 motorcycle, 2 truck
    if (dz == 3u) req = 1;
    else if (dz == 2u) { if (L.random53() < 0.6) req = 2; }
    t0 = 0;
    if (dz == 1u) t0 = np_randint(L, 50u, 200u);
}

// The same request when the next NL / NP tempered words of both windows are parked (all but ~0.5 % of them): every word it
// can consume is read up front — one LDS latency instead of ten serial ones — and the rejection loops become bit scans
// over acceptance masks.  Returns false, consuming nothing, if the request needs more than that (fewer than two accepted
// position words among NP, six rejected time-window words, pickup == delivery cell): the caller runs request_serial().
__device__ __forceinline__ bool request_windowed(DrawsP &P, DrawsL &L, uint32_t &pz, uint32_t &dz, uint32_t &pc, uint32_t &dc, uint32_t &urg, uint32_t &req,
                                                 uint32_t &t0) {
    constexpr uint32_t NL = 10, NP = 12;
    if (!(L.filled && P.filled && L.cur + NL <= L.avail && P.cur + NP <= P.avail)) return false;
    const uint32_t *lw = L.row + WL + L.cur, *pw = P.row + WP + P.cur;
    uint32_t l[NL], q[NP];
#pragma unroll
    for (uint32_t j = 0; j < NL; ++j) l[j] = lw[j];
#pragma unroll
    for (uint32_t j = 0; j < NP; ++j) q[j] = pw[j];
    pz = l[0] & 3u; dz = l[1] & 3u;                                                         // randint(0, 4): mask 3, never rejects
    uint32_t pm = 0, tm = 0;
#pragma unroll
    for (uint32_t j = 0; j < NP; ++j) pm |= ((q[j] >> 24) < 144u ? 1u : 0u) << j;           // _randbelow(144): top 8 bits, accept < 144
#pragma unroll
    for (uint32_t j = 4; j < NL; ++j) tm |= ((l[j] & 255u) <= 149u ? 1u : 0u) << (j - 4u);  // randint(50, 200): mask 255, accept <= 149
    const uint32_t pm2 = pm & (pm - 1u);
    if (!pm2 || (dz == 1u && !tm)) return false;
    const uint32_t ia = (uint32_t)__builtin_ctz(pm), ib = (uint32_t)__builtin_ctz(pm2), it = dz == 1u ? (uint32_t)__builtin_ctz(tm) : 0u;
    const uint32_t ka = pw[ia] >> 24, kb = pw[ib] >> 24, tw = lw[4u + it] & 255u;          // re-read by index: cheaper than 12-way selects
    pc = zone_cell(pz, ka); dc = zone_cell(dz, kb);
    if (dc == pc) return false;
    const double u0 = 0.3, u1 = u0 + 0.4, u2 = u1 + 0.2, u3 = u2 + 0.1;
    const double uu = ((l[2] >> 5) * 67108864.0 + (l[3] >> 6)) / 9007199254740992.0;
    urg = (uint32_t)(u0 / u3 <= uu) + (uint32_t)(u1 / u3 <= uu) + (uint32_t)(u2 / u3 <= uu) + (uint32_t)(u3 / u3 <= uu);
    const double ur = ((l[4] >> 5) * 67108864.0 + (l[5] >> 6)) / 9007199254740992.0;
    req = dz == 3u ? 1u : (dz == 2u && ur < 0.6) ? 2u : 0u;
    t0 = dz == 1u ? 50u + tw : 0u;
    L.cur += dz == 2u ? 6u : dz == 1u ? 5u + it : 4u;
    P.cur += ib + 1u;
    return true;
}

// `ab`: 24 words of this lane's LDS scratch — the request words are parked there by index and read back once, instead of
// 24 compare-and-select pairs per request to keep a runtime-indexed register array in VGPRs
__device__ __forceinline__ void do_reset(Env &e, int32_t max_steps, DrawsP &P, DrawsL &L, uint32_t *__restrict__ ab) {   // :185-234
    e.timestep = 0; e.total_reward = 0.0; e.completed = 0; e.missed = 0; e.weather = 1; e.needs_reset = 0;   // weather index 1 = 1.0
#pragma unroll
    for (int k = 0; k < 3; ++k) { e.veh[k] = 12u | (12u << 5); e.fuel[k] = vrange(k); }
    // _generate_delivery_requests :450-514
    e.nd = np_randint(L, 8u, 13u);
#pragma unroll 1
    for (uint32_t i = 0; i < (uint32_t)MAXD; ++i) {
        uint32_t A = 0, B = 0;
        if (i < e.nd) {
            uint32_t pz, dz, pc, dc, urg, req, t0;
            if (!request_windowed(P, L, pz, dz, pc, dc, urg, req, t0)) request_serial(P, L, pz, dz, pc, dc, urg, req, t0);
            const int px = (int)(pc & 31u), py = (int)(pc >> 5), dx = (int)(dc & 31u), dy = (int)(dc >> 5);
            const int base = abs(px - dx) + abs(py - dy);
            const double mult = urg == 0 ? 4.0 : urg == 1 ? 3.0 : urg == 2 ? 2.0 : 1.5;
            const uint32_t deadline = (uint32_t)((int)((double)base * mult) + 50);
            A = pc | (dc << 10) | (urg << 20) | (req << 22);                               // not completed, unassigned
            B = t0 | ((dz == 1u ? 1u : 0u) << 8) | (deadline << 9);                        // commercial zone: time window
        }
        ab[i] = A; ab[MAXD + i] = B;
    }
#pragma unroll
    for (int k = 0; k < MAXD; ++k) { e.dA[k] = ab[k]; e.dB[k] = ab[MAXD + k]; }
    // the traffic redraw that ends reset() is the caller's (coop_update_traffic)
}

__device__ __forceinline__ double weather_value(uint32_t w) { return w == 0 ? 0.8 : w == 1 ? 1.0 : w == 2 ? 1.2 : 1.5; }

// _execute_vehicle_action :278-329 for vehicle K (compile-time)
template <int K>
__device__ __forceinline__ double vehicle_action(Env &e, int32_t a, int32_t max_steps) {
    uint32_t x = e.veh[K] & 31u, y = (e.veh[K] >> 5) & 31u, cargo = (e.veh[K] >> 10) & 7u, asg1 = (e.veh[K] >> 13) & 15u;   // asg1 = assigned + 1
    double reward = 0.0;
    if (e.fuel[K] > 0.0) reward -= 2.0;
    if (a >= 1 && a <= 4) {
        if (e.fuel[K] >= vcons(K)) {
            if (a == 1) y = y > 0 ? y - 1 : 0; else if (a == 2) y = y < 24 ? y + 1 : 24;                // :331-342 (clamped)
            else if (a == 3) x = x > 0 ? x - 1 : 0; else x = x < 24 ? x + 1 : 24;
            const uint32_t tx = x / 5u < 4u ? x / 5u : 4u, ty = y / 5u < 4u ? y / 5u : 4u;
            const uint32_t lvl = e.traffic_at(ty * 5u + tx);
            const double tc = lvl == 0 ? 1.0 : lvl == 1 ? 1.5 : 2.0;
            const double cost = vcons(K) * weather_value(e.weather) * tc;
            const double f = e.fuel[K] - cost;
            e.fuel[K] = f > 0.0 ? f : 0.0;
            if (tc > 1.5) reward -= 5.0;
        } else reward -= 50.0;
    } else if (a == 5) {                                                                                // _attempt_pickup :363-391
        int best = -1;
        uint32_t best_urg = 0;
        if (cargo < vcap(K) && asg1 == 0) {
#pragma unroll
            for (int i = 0; i < MAXD; ++i) {
                const uint32_t A = e.dA[i], B = e.dB[i];
                const uint32_t t0 = B & 255u, t1 = ((B >> 8) & 1u) ? (t0 + 300u < 600u ? t0 + 300u : 600u) : (uint32_t)max_steps;
                const uint32_t req = (A >> 22) & 3u, urg = (A >> 20) & 3u;
                const bool ok = (uint32_t)i < e.nd && (A & 1023u) == (x | (y << 5)) && !((A >> 24) & 1u) && t0 <= e.timestep && e.timestep <= t1 &&
                                ((A >> 25) & 3u) == 0 && (req == 0 || req == (uint32_t)K);
                if (ok && (best < 0 || urg > best_urg)) { best = i; best_urg = urg; }
            }
        }
        if (best < 0) reward += -10.0;
        else {
            cargo += 1; asg1 = (uint32_t)best + 1u;
#pragma unroll
            for (int i = 0; i < MAXD; ++i)
                if (best == i) { e.dA[i] |= (uint32_t)(K + 1) << 25; e.dB[i] = (e.dB[i] & 0x3FFFFu) | (e.timestep << 18); }
            reward += 20.0;
        }
    } else if (a == 6) {                                                                                // _attempt_dropoff :393-437
        uint32_t A = 0, B = 0;
#pragma unroll
        for (int i = 0; i < MAXD; ++i) if (asg1 == (uint32_t)(i + 1)) { A = e.dA[i]; B = e.dB[i]; }
        if (asg1 == 0 || ((A >> 10) & 1023u) != (x | (y << 5))) reward += -10.0;
        else {
            const uint32_t urg = (A >> 20) & 3u;
            int r = urg == 0 ? 50 : urg == 1 ? 100 : 200;
            const int px = (int)(A & 31u), py = (int)((A >> 5) & 31u), dx = (int)((A >> 10) & 31u), dy = (int)((A >> 15) & 31u);
            const int dt = (int)e.timestep - (int)(B >> 18), opt = abs(px - dx) + abs(py - dy);
            if (dt <= opt + 2) r += 15;
            if (!(e.timestep <= ((B >> 9) & 511u))) r -= 20 * ((int)urg + 1);
#pragma unroll
            for (int i = 0; i < MAXD; ++i) if (asg1 == (uint32_t)(i + 1)) e.dA[i] |= 1u << 24;          // completed
            cargo = cargo > 0 ? cargo - 1 : 0; asg1 = 0;
            e.completed += 1;
            reward += (double)r;
        }
    } else if (a == 7) {                                                                                // _attempt_refuel :439-448
        const bool at = (x == 5 && y == 5) || (x == 20 && y == 5) || (x == 5 && y == 20);
        if (at) { if (e.fuel[K] < vrange(K)) { e.fuel[K] = vrange(K); reward += 10.0; } else reward += -5.0; }
        else reward += -10.0;
    } else if (a != 0) reward -= 10.0;
    e.veh[K] = x | (y << 5) | (cargo << 10) | (asg1 << 13);
    return reward;
}

// returns terminated | truncated << 1
// The traffic (every 50 steps) and weather (every 100) redraws (:257-262) are DEFERRED: they touch the RNG, only ~2 %
// of the lanes need them in a given step, and neither reward nor termination depends on them — the flags go to the
// work list and dense_kernel applies them (in the reference's order) before the obs row is final.
__device__ __forceinline__ uint32_t env_step(Env &e, int32_t max_steps, int32_t a0, int32_t a1, int32_t a2, uint32_t &work, double &reward) {   // :236-276
    double total = 0.0;
    total += vehicle_action<0>(e, a0, max_steps);
    total += vehicle_action<1>(e, a1, max_steps);
    total += vehicle_action<2>(e, a2, max_steps);
    e.timestep += 1;
    work = (e.timestep % 50u == 0 ? W_TRAFFIC : 0u) | (e.timestep % 100u == 0 ? W_WEATHER : 0u);
    uint32_t urgent = 0;
    bool all_done = true;
#pragma unroll
    for (int i = 0; i < MAXD; ++i) {                                                                  // _check_missed_deadlines :530-535
        const uint32_t A = e.dA[i], B = e.dB[i];
        const bool live = (uint32_t)i < e.nd, done = (A >> 24) & 1u, urg = ((A >> 20) & 3u) >= 2u;
        if (live && e.timestep > ((B >> 9) & 511u) && !done && urg) e.missed += 1;
        urgent += (live && urg) ? 1u : 0u;
        all_done = all_done && (!live || done);
    }
    const bool fuel_out = !(e.fuel[0] > 0.0) && !(e.fuel[1] > 0.0) && !(e.fuel[2] > 0.0);
    e.total_reward += total;
    reward = total;
    const bool term = all_done || fuel_out || (urgent > 0 && (double)e.missed >= (double)urgent * 0.5);   // :537-553
    return (term ? 1u : 0u) | (e.timestep >= (uint32_t)max_steps ? 2u : 0u);
}

__device__ __forceinline__ void stage_row(const Env &e, float *__restrict__ row) {                    // :555-593
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        row[2 * k] = (float)(e.veh[k] & 31u); row[2 * k + 1] = (float)((e.veh[k] >> 5) & 31u);
        row[6 + k] = (float)(e.fuel[k] / vrange(k));
        row[9 + k] = (float)((double)((e.veh[k] >> 10) & 7u) / (double)vcap(k));
        row[12 + k] = (float)((int)((e.veh[k] >> 13) & 15u) - 1);
    }
#pragma unroll
    for (int i = 0; i < MAXD; ++i) {
        const uint32_t A = e.dA[i];
        const bool live = (uint32_t)i < e.nd && !((A >> 24) & 1u);
        row[15 + 2 * i] = live ? (float)(A & 31u) : -1.0f;
        row[16 + 2 * i] = live ? (float)((A >> 5) & 31u) : -1.0f;
        row[39 + i] = live ? (float)((A >> 20) & 3u) : -1.0f;
    }
#pragma unroll
    for (int c = 0; c < 25; ++c) row[51 + c] = (float)(((c < 16 ? e.traffic[0] : e.traffic[1]) >> ((c & 15) * 2)) & 3u);
}

// One step of the lane's env (not its row, not its store), NO random draws: vehicle actions, deadlines, termination; reward /
// flags outputs, episode statistics.  Everything that needs the generators — the traffic/weather redraws and the 8-12 new
// delivery requests of an episode reset (~150 draws over two interleaved streams, ~10k instructions) — happens for ~1-3 % of
// the lanes per step; executed in place it cost every wave that whole path at 1/64 lane utilisation (234 us per 131k-env step,
// profiles/r01_fleet_step_v1_summary.txt).  Those envs get an entry on work list `p.parity` instead (appended by wave `wave_id`
// to its sub-list) and a dense launch processes them with full waves.
__device__ __forceinline__ void lane_step(const Params &p, Env &e, int64_t i, uint32_t wave_id) {
    const int t = p.t_index;
    double reward = 0.0;
    uint32_t flags = 0, work = 0;
    if (p.mode == CGE_AUTORESET_NEXT_STEP && e.needs_reset) {
        work = W_RESET;                                        // reset-only step: action ignored, reward 0
    } else {
        int32_t a0, a1, a2;
        if (p.actions) {
            const int32_t *ap = p.actions + ((int64_t)t * p.n + i) * 3;
            a0 = ap[0]; a1 = ap[1]; a2 = ap[2];
        } else {
            const uint64_t key = hash_env_key(p.a_seed, (uint64_t)(p.env0 + i));
            a0 = (int32_t)hash_action_from_key(key, (uint64_t)(p.t0 + t), 8u, 0u);
            a1 = (int32_t)hash_action_from_key(key, (uint64_t)(p.t0 + t), 8u, 1u);
            a2 = (int32_t)hash_action_from_key(key, (uint64_t)(p.t0 + t), 8u, 2u);
        }
        flags = env_step(e, p.max_steps, a0, a1, a2, work, reward);
        if (flags) {
            e.episodes += 1;
            if (p.ep_ret) p.ep_ret[i] = e.total_reward;                // fleet_env.py accumulates it in step(), reset() zeroes it
            if (p.ep_len) p.ep_len[i] = (int32_t)e.timestep;
            if (p.mode == CGE_AUTORESET_SAME_STEP) work |= W_RESET;
            else if (p.mode == CGE_AUTORESET_NEXT_STEP) e.needs_reset = 1;
        }
    }
    e.pending = work ? 1u : 0u;
    e.seq = (e.seq + 1u) & 15u;
    if (work) {
        // the sub-list follows from the ENV (the block that owns it), not from whoever steps it: a pipelined dense launch steps envs of any
        // block, and filing them under its own block index let a sub-list receive more than its sub_cap = ceil(blocks / 64) * 64 entries
        // (a NextStep mass reset re-lists every env while step blocks append to the same lists)
        (void)wave_id;
        const uint32_t sub = (uint32_t)(i / BLOCK) % (uint32_t)NSUB;
        const uint32_t slot = atomicAdd(work_counter(p.work_count, p.parity, sub), 1u);
        work_sublist(p, p.parity, sub)[CGE_GX(1, slot, p.sub_cap)] = work_entry(i, work, e.lpos, e.lpretw, e.ppos, e.ppretw);
    }
    if (p.accumulate) {
        if (p.reward_sum) p.reward_sum[i] += reward;
        if (p.done_count) p.done_count[i] += flags ? 1 : 0;
        if (p.reward) p.reward[(int64_t)t * p.n + i] = (float)reward;
        if (p.terminated) p.terminated[(int64_t)t * p.n + i] = (uint8_t)flags;
    } else {
        p.reward[i] = (float)reward;
        p.terminated[i] = (uint8_t)(flags & 1u);
        p.truncated[i] = (uint8_t)((flags >> 1) & 1u);
        if (p.done) p.done[i] = flags ? 1 : 0;
    }
}

// The step of 64 envs, one per lane, and the obs row of the state as it stands.  PIPE (rollouts): envs still on the previous
// step's list are left alone — the dense launch that runs beside this one serves them and steps them itself.
template <bool PIPE>
__global__ __launch_bounds__(BLOCK) void step_kernel(Params p) {
    const int64_t i0 = (int64_t)blockIdx.x * BLOCK;
    const int64_t i = i0 + threadIdx.x;
    const bool live = i < p.n;
    const int64_t li = live ? i : i0;
    if (blockIdx.x == 0 && threadIdx.x < (unsigned)NSUB) *work_counter(p.work_count, (p.parity + 1) % 3, threadIdx.x) = 0;   // the list after this one: nobody's during this launch
    Env e;
    e.load(p.state, p.n, li);
    // (PIPE: the dense launch beside this one clears `pending` when it has served an env — but by then it has also taken the env
    // through this step, and `seq`, in the same dword, says so: whichever version of the record this load sees, the env is skipped)
    const bool mine = live && !(PIPE && (e.pending || e.seq != p.seq));
    if (mine) {
        lane_step(p, e, i, blockIdx.x);
        e.store(p.state, p.n, i);
    }
    // rows of envs on the work list are rewritten by the dense launch (also their final_obs row).  Each lane streams its own
    // 304-byte row from registers, 19 16-byte stores (store_own_row); round 1 staged the wave's rows in LDS and wrote them
    // dword by dword in a 76-trip loop, 8 of this kernel's 20 us per wave.
    if (p.obs) {
        float row[OBS];
        stage_row(e, row);
        store_own_row<OBS>(p.obs + (int64_t)p.t_index * p.obs_step_stride + li * OBS, 0, row, mine);
    }
}

// DL lanes per wave, one work-list entry each (dense): traffic / weather redraw, terminal obs -> final_obs, episode
// reset, obs row.  list == nullptr: API reset(mask) / cursor rewind / fresh-handle init over ALL envs (`what`).
// Few waves run here (1-3 % of the envs), so the kernel is bound by one lane's serial chain, not by throughput:
// both streams' windows are filled once, wave-convergently, before any draw is consumed.
// what: 0 work list, 1 reset(mask)+obs, 2 rewind cursors after seeding, 3 initial state of a fresh handle
#ifdef CGE_FLEET_TIMING
__device__ unsigned long long g_timing[1024 * 16];          // one slot row per block: no contention between the timed waves
#define TICK(k) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); const unsigned long long now_ = wall_clock64(); \
    if (timed) { g_timing[blockIdx.x * 16 + k] += now_ - t_last; } t_last = now_; } while (0)
#else
#define TICK(k)
#endif
// Pipelined (rollouts): the launch serves list (parity + 2) % 3 — the step BEFORE the one now being taken by step_kernel<true> on the
// other stream — and then takes each served env through that step too (lane_step), appending to list `parity` like everyone else.
// MODE: 0 plain, 1 pipelined, 2 = 0 under its own name (the launch that ends a rollout: the profiles keep the two paths' bytes apart)
template <int MODE>
__global__ __launch_bounds__(BLOCK) void dense_kernel(Params p, int what) {
    constexpr bool PIPE = MODE == 1;
    const int serve = PIPE ? (p.parity + 2) % 3 : p.parity;
    const int64_t t_row = PIPE ? p.t_index - 1 : p.t_index;     // the step whose rows the served work completes
#ifdef CGE_FLEET_TIMING
    unsigned long long t_last = wall_clock64();
    const unsigned long long t_begin = t_last;
    bool timed = false;
#endif
    __shared__ uint32_t tile[DL * ROW];
    __shared__ uint32_t drawsP[DL * PROW], drawsL[DL * LROW];
    __shared__ int64_t row_env[DL], row_fin[DL];
    __shared__ uint32_t traffic_acc[2 * DL];
    const uint32_t lane = threadIdx.x & 63u;
    // lane s holds sub-list s's entry count and the running total up to and including it
    uint32_t sub_n = 0, sub_end = 0;
    if (what == 0) {
        sub_n = *work_counter(p.work_count, serve, lane);
        if (!PIPE && blockIdx.x == 0) *work_counter(p.work_count, (p.parity + 1) % 3, lane) = 0;  // next step's counters (pipelined: the step launch beside this one does it)
        sub_end = sub_n;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t up = (uint32_t)__shfl_up((int)sub_end, d, 64);
            if (lane >= (uint32_t)d) sub_end += up;
        }
    }
    const uint32_t count = what == 0 ? lane_u32(sub_end, 63) : (uint32_t)p.n;
    const uint32_t slot = lane < (uint32_t)DL ? lane : 0u;
#ifdef CGE_FLEET_TIMING
    timed = threadIdx.x == 0 && what == 0 && count < 16384u && count > 64u;      // ordinary steps only, not the mass redraws
    TICK(9);
#endif
#pragma unroll 1
    for (uint32_t first = blockIdx.x * DL; first < count; first += gridDim.x * DL) {
        const uint32_t tix = first + lane;
        const bool live = lane < (uint32_t)DL && tix < count;
        int64_t i;
        uint32_t work, lcur = 0, pcur = 0;                     // cursors: pos | pretw flag << 10
        if (what == 0) {
            const uint32_t g = live ? tix : first;             // entry g of the concatenated sub-lists -> (sub-list, offset)
            uint32_t sub = 0;
#pragma unroll
            for (int b = NSUB / 2; b; b >>= 1)
                if ((uint32_t)__shfl((int)sub_end, (int)(sub + b - 1), 64) <= g) sub += b;
            const uint32_t before = (uint32_t)__shfl((int)(sub_end - sub_n), (int)sub, 64);
            const uint64_t entry = work_sublist(p, serve, sub)[CGE_GX(2, g - before, p.sub_cap)];
            i = (int64_t)CGE_GX(3, entry >> 25, p.n);
            work = live ? (uint32_t)(entry & 7u) : 0u;
            lcur = (uint32_t)(entry >> 14) & 2047u; pcur = (uint32_t)(entry >> 3) & 2047u;
        } else {
            i = live ? (int64_t)tix : (int64_t)first;
            work = (live && what == 1 && (!p.mask || p.mask[i])) ? W_RESET : 0u;
        }
        Env e;
        if (what != 0) {                                       // whole-population passes: the cursors come from the record
            e.load(p.state, p.n, i);
            if (what == 2) { if (live) { e.ppos = e.lpos = 0; e.ppretw = e.lpretw = 0; e.store(p.state, p.n, i); } continue; }
            if (what == 3) { if (live) { e.weather = 1; e.store(p.state, p.n, i); } continue; }
            lcur = e.lpos | (e.lpretw ? 1024u : 0u); pcur = e.ppos | (e.ppretw ? 1024u : 0u);
        }
        DrawsP P(drawsP + slot * PROW, p.mtP + i * MT_STRIDE, pcur & 1023u, (pcur & 1024u) ? (uint32_t)MT_N : 0u);
        DrawsL L(drawsL + slot * LROW, p.mtL + i * MT_STRIDE, lcur & 1023u, (lcur & 1024u) ? (uint32_t)MT_N : 0u);
        // one round trip: both windows of every listed env and (work list) the env records.  A redraw-only entry needs
        // 60 words of the L stream, a reset the whole window of both.
        const uint32_t quadsL = !work ? 0u : (work & W_RESET) ? (uint32_t)((WL + 63) / 64) : 1u;
        const uint32_t quadsP = (work & W_RESET) ? (uint32_t)((WP + 63) / 64) : 0u;
        CoopFillRegs<WL, DL> fl;
        CoopFillRegs<WP, DL> fp;
        TICK(12);
        coop_fill_issue<WL, DL>(L, fl, quadsL);
        TICK(13);
        coop_fill_issue<WP, DL>(P, fp, quadsP);
        TICK(14);
        if (what == 0) e.load(p.state, p.n, i);
        TICK(0);
        coop_fill_park<WL, DL>(L, fl, LROW, quadsL);
        TICK(1);
        coop_fill_park<WP, DL>(P, fp, PROW, quadsP);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        TICK(2);
        coop_update_traffic(e, L, (work & W_TRAFFIC) != 0, drawsL, traffic_acc);                      // :257-258
        if (work & W_WEATHER) {                                                                       // _update_weather :524-528
            const double w0 = 0.3, w1 = w0 + 0.5, w2 = w1 + 0.15, w3 = w2 + 0.05;
            e.weather = np_choice_cdf(L, w0 / w3, w1 / w3, w2 / w3, w3 / w3);
        }
        TICK(3);
        if (lane < (uint32_t)DL) row_env[lane] = live ? i : -1;
        const int nlive = (int)(count - first < (uint32_t)DL ? count - first : (uint32_t)DL);
        // terminal rows: step() -> row env of final_obs_out; rollout -> the next slot of the env's 64-env segment of the compacted side
        // output.  This kernel serves work-list entries, not contiguous envs, and a rollout is one launch per step, so the segment's
        // fill count lives in memory (zeroed by cge_fleet_rollout): one returning atomic per finishing env, a handful per segment and step
        const bool fin = what == 0 && (work & W_RESET) && p.mode == CGE_AUTORESET_SAME_STEP && (p.final_obs || p.fin.rows);
        if (lane < (uint32_t)DL) {
            int64_t to = live ? i : -1;
            if (p.fin.rows && fin && live) {
                const int64_t seg = i >> 6;
                const int32_t slot = atomicAdd(p.fin.count + seg, 1);
                to = -1;
                if ((int64_t)slot < p.fin.cap) { to = seg * p.fin.cap + slot; p.fin.index[to] = t_row * p.fin.n + i; }
            }
            row_fin[lane] = to;
        }
#pragma unroll 1
        for (int pass = 0; pass < 2; ++pass) {
            // pass 0: terminal obs (after the redraws) of the envs that finished -> final_obs; pass 1: reset, then obs
            const bool want = pass == 0 ? fin : (what == 1 ? live : work != 0);
            if (pass == 1) TICK(4);
            if (pass == 1) {
                if (work & W_RESET) do_reset(e, p.max_steps, P, L, tile + slot * ROW);
                coop_update_traffic(e, L, (work & W_RESET) != 0, drawsL, traffic_acc);
            }
            if (pass == 1) TICK(5);
            float *dst = pass == 0 ? (p.fin.rows ? static_cast<float *>(p.fin.rows) : p.final_obs) : (p.obs ? p.obs + t_row * p.obs_step_stride : nullptr);
            const int64_t *rows_to = pass == 0 ? row_fin : row_env;
            const unsigned long long m = __ballot(want);
            if (!m || !dst) continue;
            if (PIPE && pass == 1 && p.obs_step_stride == 0) continue;   // rows in place: this launch's own step rewrites them below
            if (lane < (uint32_t)DL) stage_row(e, reinterpret_cast<float *>(tile + lane * ROW));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll 1
            for (int r = 0; r < nlive; ++r) {                  // each listed env's 304-byte row: two 64-lane stores
                if (!((m >> r) & 1ull) || rows_to[r] < 0) continue;
                uint32_t *drow = reinterpret_cast<uint32_t *>(dst + rows_to[r] * OBS);
                drow[lane] = tile[r * ROW + lane];
                if (lane < (uint32_t)(OBS - 64)) drow[64 + lane] = tile[r * ROW + 64 + lane];
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        TICK(6);
        coop_flush<WP, DL>(P, PROW);
        coop_flush<WL, DL>(L, LROW);
        TICK(7);
        if (work) {
            e.ppos = P.pos; e.ppretw = P.pretw; e.lpos = L.pos; e.lpretw = L.pretw;
            e.pending = 0;
            if (PIPE) lane_step(p, e, i, blockIdx.x);           // the step now being taken, for the env just served (draws nothing)
            e.store(p.state, p.n, i);
        }
        if (PIPE && p.obs) {                                    // ... and its row
            const unsigned long long m = __ballot(work != 0u);
            if (lane < (uint32_t)DL) stage_row(e, reinterpret_cast<float *>(tile + lane * ROW));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            float *dst = p.obs + (int64_t)p.t_index * p.obs_step_stride;
#pragma unroll 1
            for (int r = 0; r < nlive; ++r) {
                if (!((m >> r) & 1ull)) continue;
                uint32_t *drow = reinterpret_cast<uint32_t *>(dst + row_env[r] * OBS);
                drow[lane] = tile[r * ROW + lane];
                if (lane < (uint32_t)(OBS - 64)) drow[64 + lane] = tile[r * ROW + 64 + lane];
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        TICK(8);
#ifdef CGE_FLEET_TIMING
        if (timed) { unsigned long long *tm = g_timing + blockIdx.x * 16; tm[15] += 1; const unsigned long long el = wall_clock64() - t_begin; if (el > tm[10]) tm[10] = el; if (P.refills + L.refills) tm[11] += 1; }
#endif
    }
}

__global__ __launch_bounds__(256) void info_kernel(const uint4 *__restrict__ state, int64_t n, int field, double *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    Env e;
    e.load(state, n, i);
    double v = 0.0;
    switch (field) {
        case CGE_FLEET_INFO_TIMESTEP: v = e.timestep; break;
        case CGE_FLEET_INFO_MISSED_DEADLINES: v = e.missed; break;
        case CGE_FLEET_INFO_COMPLETED_DELIVERIES: v = e.completed; break;
        case CGE_FLEET_INFO_NUM_REQUESTS: v = e.nd; break;
        case CGE_FLEET_INFO_WEATHER_EFFECT: v = weather_value(e.weather); break;
        case CGE_FLEET_INFO_TOTAL_REWARD: v = e.total_reward; break;
        case CGE_FLEET_INFO_EPISODES: v = e.episodes; break;
        case CGE_FLEET_INFO_NEEDS_RESET: v = e.needs_reset; break;
        case CGE_FLEET_INFO_FUEL0: v = e.fuel[0]; break;
        case CGE_FLEET_INFO_FUEL1: v = e.fuel[1]; break;
        case CGE_FLEET_INFO_FUEL2: v = e.fuel[2]; break;
    }
    out[i] = v;
}

}  // namespace fleet
}  // namespace cge

using namespace cge;

struct cge_fleet : HandleBase {
    cge_fleet_config cfg{};
    uint4 *state = nullptr;
    uint32_t *mtP = nullptr, *mtL = nullptr;
    uint32_t *work_count = nullptr;
    uint64_t *work_list = nullptr;
    int parity = 0;
    uint32_t seq = 0;                    // Env::seq of every env (mod 16)
    hipStream_t side = nullptr;          // pipelined rollouts: the dense launches' stream
    hipEvent_t ev_step = nullptr, ev_dense = nullptr;
    static constexpr uint32_t snap_tag = 3u;
    std::vector<std::pair<void *, size_t>> blobs() const { return {{state, (size_t)fleet::COLS * n * sizeof(uint4)}, {mtP, (size_t)n * MT_STRIDE * 4}, {mtL, (size_t)n * MT_STRIDE * 4}, {work_count, count_bytes()}}; }
    static size_t count_bytes() { return (size_t)3 * fleet::NSUB * fleet::CNT_STRIDE * sizeof(uint32_t); }
    unsigned blocks() const { return (unsigned)((n + fleet::BLOCK - 1) / fleet::BLOCK); }
    int64_t sub_cap() const { return ((int64_t)blocks() + fleet::NSUB - 1) / fleet::NSUB * fleet::BLOCK; }
    size_t list_entries() const { return (size_t)3 * fleet::NSUB * (size_t)sub_cap(); }
    uint32_t snap_extra() const { return (uint32_t)parity | (seq << 2); }
    void set_snap_extra(uint32_t v) { parity = (int)(v & 3u) % 3; seq = (v >> 2) & 15u; }
    fleet::Params params() const {
        fleet::Params p{};
        p.state = state; p.mtP = mtP; p.mtL = mtL; p.n = n; p.env0 = env0; p.mode = cfg.autoreset_mode; p.max_steps = cfg.max_timesteps;
        p.work_count = work_count; p.work_list = work_list; p.sub_cap = sub_cap(); p.parity = parity;
        p.ep_ret = ep_ret; p.ep_len = ep_len; p.done = done_out;
        return p;
    }
    void free_all() { if (side) (void)hipStreamDestroy(side); if (ev_step) (void)hipEventDestroy(ev_step); if (ev_dense) (void)hipEventDestroy(ev_dense);
                      (void)hipFree(state); (void)hipFree(mtP); (void)hipFree(mtL); (void)hipFree(work_count); (void)hipFree(work_list); }
    // k env steps.  step() (the rows must be final when the call returns): the RNG-free step kernel, then the dense kernel over the
    // envs it listed, on the caller's stream.
    hipError_t launch_steps(fleet::Params &p, int k, hipStream_t s) {
        const unsigned db = (unsigned)((n + fleet::DL - 1) / fleet::DL), dblocks = db < 1024u ? db : 1024u;
        for (int t = 0; t < k; ++t) {
            p.t_index = t; p.parity = parity; p.seq = seq;
            hipLaunchKernelGGL(fleet::step_kernel<false>, dim3(blocks()), dim3(fleet::BLOCK), 0, s, p);
            hipLaunchKernelGGL(fleet::dense_kernel<0>, dim3(dblocks), dim3(fleet::BLOCK), 0, s, p, 0);
            parity = (parity + 1) % 3; seq = (seq + 1u) & 15u;
        }
        last_kernel = "cge::fleet::step_kernel<false> + cge::fleet::dense_kernel<0>";
        return hipGetLastError();
    }
    // Rollouts, k >= 2: both kernels are bound by one wave's latency, not by throughput, so the dense launch for step t runs
    // BESIDE step t + 1's step launch, on the handle's side stream: step_kernel<true> leaves the envs on list t alone,
    // dense_kernel<1> serves them and takes them through step t + 1 itself.
    //   caller's stream:  S0 | S1' | S2' | ...  | S(k-1)' |            (S(t+1)' waits for D(t-1)': its pending flags and list entries)
    //   side stream:           D0' | D1' | ...  | D(k-2)' | D(k-1)     (D(t)' waits for S(t)': the list is complete)
    // 58 -> 3x us per 131,072-env step (round 3).  (Measured and dropped before: 2-4 partitions of the envs on forked streams —
    // partitions overlap but each is as long as the whole; one launch with step and dense ROLES — a launch has one register and
    // LDS budget, the dense role's 254 VGPRs + 23 KB would halve the step role's residency.)
    hipError_t launch_rollout(fleet::Params &p, int k, hipStream_t s) {
        if (k < 2) return launch_steps(p, k, s);
        const unsigned db = (unsigned)((n + fleet::DL - 1) / fleet::DL), dblocks = db < 1024u ? db : 1024u;
        hipError_t e;
        for (int t = 0; t < k; ++t) {
            p.t_index = t; p.parity = parity; p.seq = seq;
            if (t == 0) hipLaunchKernelGGL(fleet::step_kernel<true>, dim3(blocks()), dim3(fleet::BLOCK), 0, s, p);   // (nothing is pending: skips nobody)
            else {
                if (t >= 2 && (e = hipStreamWaitEvent(s, ev_dense, 0)) != hipSuccess) return e;      // D(t-2)' done
                hipLaunchKernelGGL(fleet::step_kernel<true>, dim3(blocks()), dim3(fleet::BLOCK), 0, s, p);
                // D(t-1)': serves list t-1 = (parity + 2) % 3, steps its envs through step t; needs S(t-1)' (event recorded below, last trip)
                hipLaunchKernelGGL(fleet::dense_kernel<1>, dim3(dblocks), dim3(fleet::BLOCK), 0, side, p, 0);
                if ((e = hipEventRecord(ev_dense, side)) != hipSuccess) return e;
            }
            if ((e = hipEventRecord(ev_step, s)) != hipSuccess) return e;                            // S(t)' done ->
            if ((e = hipStreamWaitEvent(side, ev_step, 0)) != hipSuccess) return e;                  // ... the side stream's next launch may read list t
            parity = (parity + 1) % 3; seq = (seq + 1u) & 15u;
        }
        // the last step's list: plain dense launch (serves list k-1 = (parity + 2) % 3 now; no further step)
        p.t_index = k - 1; p.parity = (parity + 2) % 3;
        hipLaunchKernelGGL(fleet::dense_kernel<2>, dim3(dblocks), dim3(fleet::BLOCK), 0, side, p, 0);
        if ((e = hipEventRecord(ev_dense, side)) != hipSuccess) return e;
        if ((e = hipStreamWaitEvent(s, ev_dense, 0)) != hipSuccess) return e;
        last_kernel = "cge::fleet::step_kernel<true> + cge::fleet::dense_kernel<1> + cge::fleet::dense_kernel<2>";
        return hipGetLastError();
    }
    hipError_t launch_all(fleet::Params &p, int what, hipStream_t s) {
        const unsigned b = (unsigned)((n + fleet::DL - 1) / fleet::DL);
        hipLaunchKernelGGL(fleet::dense_kernel<0>, dim3(b < (1u << 20) ? b : (1u << 20)), dim3(fleet::BLOCK), 0, s, p, what);
        return hipGetLastError();
    }
};

extern "C" {

#ifdef CGE_GUARD
int cge_fleet_debug_guard(unsigned int *out) {         // [count, site, index, limit, block, lane, 0, 0] of the first violation (cge_device.hpp: CGE_GX)
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(cge::g_guard), 8 * sizeof(unsigned int)) == hipSuccess ? 0 : 1;
}
#endif

#ifdef CGE_FLEET_TIMING
int cge_fleet_debug_timing(unsigned long long *out, int clear) {
    static unsigned long long all[1024 * 16];
    if (hipMemcpyFromSymbol(all, HIP_SYMBOL(fleet::g_timing), sizeof all) != hipSuccess) return 1;
    for (int k = 0; k < 16; ++k) out[k] = 0;
    for (int b = 0; b < 1024; ++b)
        for (int k = 0; k < 16; ++k) { if (k == 10) { if (all[b * 16 + k] > out[k]) out[k] = all[b * 16 + k]; } else out[k] += all[b * 16 + k]; }
    if (clear) { memset(all, 0, sizeof all); if (hipMemcpyToSymbol(HIP_SYMBOL(fleet::g_timing), all, sizeof all) != hipSuccess) return 1; }
    return 0;
}
#endif

int cge_fleet_create(const cge_fleet_config *cfg, int64_t n_envs, int device, int64_t env_index0, cge_fleet **out) {
    if (!cfg || !out || n_envs <= 0 || env_index0 < 0) return CGE_ERR_INVALID_ARG;
    *out = nullptr;
    if (cfg->autoreset_mode < 0 || cfg->autoreset_mode > 2 || cfg->max_timesteps < 0 || cfg->max_timesteps > 1023) return CGE_ERR_INVALID_ARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return CGE_ERR_NO_DEVICE;
    cge_fleet *h = new cge_fleet();
    h->cfg = *cfg;
    if (h->cfg.max_timesteps == 0) h->cfg.max_timesteps = 800;
    h->n = n_envs; h->env0 = env_index0; h->device = device;
    DeviceGuard g(device);
    const size_t sb = (size_t)fleet::COLS * n_envs * sizeof(uint4), mb = (size_t)n_envs * MT_STRIDE * sizeof(uint32_t);
    hipError_t e;
    if ((e = hipMalloc(&h->state, sb)) != hipSuccess || (e = hipMalloc(&h->mtP, mb)) != hipSuccess || (e = hipMalloc(&h->mtL, mb)) != hipSuccess ||
        (e = hipMalloc(&h->work_count, cge_fleet::count_bytes())) != hipSuccess ||
        (e = hipMalloc(&h->work_list, h->list_entries() * sizeof(uint64_t))) != hipSuccess ||
        (e = hipMemset(h->work_count, 0, cge_fleet::count_bytes())) != hipSuccess || (e = hipMemset(h->state, 0, sb)) != hipSuccess ||
        (e = hipStreamCreateWithFlags(&h->side, hipStreamNonBlocking)) != hipSuccess ||
        (e = hipEventCreateWithFlags(&h->ev_step, hipEventDisableTiming)) != hipSuccess || (e = hipEventCreateWithFlags(&h->ev_dense, hipEventDisableTiming)) != hipSuccess) {
        h->free_all();
        delete h;
        return CGE_ERR_HIP;
    }
    h->device_bytes = sb + 2 * mb + h->list_entries() * sizeof(uint64_t) + cge_fleet::count_bytes();
    e = launch_mt_seed(h->mtP, MT_STRIDE, n_envs, nullptr, 0, env_index0, 0, nullptr);
    if (e == hipSuccess) e = launch_mt_seed(h->mtL, MT_STRIDE, n_envs, nullptr, 0, env_index0, 1, nullptr);
    if (e == hipSuccess) {
        fleet::Params p = h->params();
        e = h->launch_all(p, 3, nullptr);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
    if (e != hipSuccess) {
        h->free_all();
        delete h;
        return CGE_ERR_HIP;
    }
    *out = h;
    return CGE_OK;
}

int cge_fleet_destroy(cge_fleet *h) {
    if (!h) return CGE_ERR_INVALID_ARG;
    DeviceGuard g(h->device);
    (void)hipDeviceSynchronize();
    h->free_all();
    delete h;
    return CGE_OK;
}

int cge_fleet_seed(cge_fleet *h, const uint64_t *seeds, uint64_t base_seed, void *stream) {
    if (!h) return CGE_ERR_INVALID_ARG;
    DeviceGuard g(h->device);
    if (!seeds && base_seed + (uint64_t)(h->env0 + h->n) > 0x100000000ull)
        return h->fail(CGE_ERR_INVALID_ARG, "cge_fleet_seed: np.random.seed needs seeds < 2**32");
    CGE_TRY(h, launch_mt_seed(h->mtP, MT_STRIDE, h->n, seeds, base_seed, h->env0, 0, as_stream(stream)));
    CGE_TRY(h, launch_mt_seed(h->mtL, MT_STRIDE, h->n, seeds, base_seed, h->env0, 1, as_stream(stream)));
    fleet::Params p = h->params();
    CGE_TRY(h, h->launch_all(p, 2, as_stream(stream)));
    return CGE_OK;
}

int cge_fleet_reset(cge_fleet *h, const uint8_t *mask, float *obs_out, void *stream) {
    if (!h) return CGE_ERR_INVALID_ARG;
    DeviceGuard g(h->device);
    fleet::Params p = h->params();
    p.mask = mask; p.obs = obs_out;
    CGE_TRY(h, h->launch_all(p, 1, as_stream(stream)));
    return CGE_OK;
}

int cge_fleet_step(cge_fleet *h, const int32_t *actions, float *obs_out, float *reward_out, uint8_t *terminated_out, uint8_t *truncated_out,
                   float *final_obs_out, void *stream) {
    if (!h) return CGE_ERR_INVALID_ARG;
    if (!actions || !obs_out || !reward_out || !terminated_out || !truncated_out)
        return h->fail(CGE_ERR_INVALID_ARG, "cge_fleet_step: null actions/obs/reward/terminated/truncated pointer");
    DeviceGuard g(h->device);
    fleet::Params p = h->params();
    p.actions = actions; p.obs = obs_out; p.reward = reward_out; p.terminated = terminated_out; p.truncated = truncated_out;
    p.final_obs = final_obs_out; p.k_steps = 1;
    CGE_TRY(h, h->launch_steps(p, 1, as_stream(stream)));
    return CGE_OK;
}

int cge_fleet_rollout(cge_fleet *h, int32_t k_steps, const int32_t *actions, uint64_t action_seed, int64_t t0, float *obs_out,
                      int64_t obs_step_stride, float *reward_traj_out, uint8_t *terminated_traj_out, double *reward_sum_out,
                      int32_t *done_count_out, void *stream) {
    if (!h) return CGE_ERR_INVALID_ARG;
    if (k_steps < 0 || obs_step_stride < 0 || (obs_step_stride != 0 && obs_step_stride < h->n * fleet::OBS))
        return h->fail(CGE_ERR_INVALID_ARG, "cge_fleet_rollout: bad k_steps / obs_step_stride");
    if (k_steps == 0) return CGE_OK;
    DeviceGuard g(h->device);
    fleet::Params p = h->params();
    p.k_steps = k_steps; p.actions = actions; p.a_seed = action_seed; p.t0 = t0; p.obs = obs_out; p.obs_step_stride = obs_step_stride;
    p.reward = reward_traj_out; p.terminated = terminated_traj_out; p.reward_sum = reward_sum_out; p.done_count = done_count_out;
    p.accumulate = 1;
    p.fin = FinalSeg{h->fin_rows, h->fin_index, h->fin_count, h->fin_cap, h->n};
    if (h->fin_count) CGE_TRY(h, hipMemsetAsync(h->fin_count, 0, (size_t)((h->n + 63) / 64) * sizeof(int32_t), as_stream(stream)));
    if (reward_sum_out) CGE_TRY(h, hipMemsetAsync(reward_sum_out, 0, (size_t)h->n * sizeof(double), as_stream(stream)));
    if (done_count_out) CGE_TRY(h, hipMemsetAsync(done_count_out, 0, (size_t)h->n * sizeof(int32_t), as_stream(stream)));
    CGE_TRY(h, h->launch_rollout(p, k_steps, as_stream(stream)));
    return CGE_OK;
}

CGE_DEFINE_FINAL_OBS(fleet, float, 64)

int cge_fleet_info(cge_fleet *h, int32_t field_id, double *out, void *stream) {
    if (!h || !out || field_id < 0 || field_id > CGE_FLEET_INFO_FUEL2) return CGE_ERR_INVALID_ARG;
    DeviceGuard g(h->device);
    hipLaunchKernelGGL(fleet::info_kernel, dim3((unsigned)((h->n + 255) / 256)), dim3(256), 0, as_stream(stream), h->state, h->n, field_id, out);
    CGE_TRY(h, hipGetLastError());
    return CGE_OK;
}

size_t cge_fleet_snapshot_bytes(const cge_fleet *h) { return h ? snapshot_bytes(h) : 0; }
int cge_fleet_snapshot_get(cge_fleet *h, void *host_buf, void *stream) { return snapshot_get(h, host_buf, as_stream(stream)); }
int cge_fleet_snapshot_set(cge_fleet *h, const void *host_buf, void *stream) { return snapshot_set(h, host_buf, as_stream(stream)); }
size_t cge_fleet_device_bytes(const cge_fleet *h) { return h ? h->device_bytes : 0; }
int cge_fleet_episode_stats(cge_fleet *h, double *return_out, int32_t *length_out) {
    if (!h) return CGE_ERR_INVALID_ARG;
    h->ep_ret = return_out; h->ep_len = length_out;
    return CGE_OK;
}

int cge_fleet_done_mask(cge_fleet *h, uint8_t *done_out) {
    if (!h) return CGE_ERR_INVALID_ARG;
    h->done_out = done_out;
    return CGE_OK;
}

const char *cge_fleet_last_error(const cge_fleet *h) { return h ? h->last_error.c_str() : "null handle"; }

const char *cge_fleet_last_kernel(const cge_fleet *h) { return h ? h->last_kernel.c_str() : ""; }

}  // extern "C"
