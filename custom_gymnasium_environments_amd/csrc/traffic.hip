// traffic.hip — batched TrafficManagementEnv for MI355X (gfx950): kernels + C ABI (include/cge_amd.h).
//
// Re-expresses /root/reference/traffic_management_env/ for N independent instances, one lane per env:
//   environment.py  reset :141-166, step :168-203, _apply_actions :205-220, _spawn_vehicles :222-249,
//                   _process_intersections :271-281, _remove_completed_vehicles :283-285,
//                   _calculate_reward :287-311, _get_observation :313-363
//   utils.py        TrafficLight.update/_advance_phase :79-97, set_phase :108-118, can_pass :99-106,
//                   Intersection.process_vehicles :141-163, generate_vehicle_route :174-193,
//                   get_neighboring_intersections :196-214, get_direction_between_intersections :230-248
// The reference's O(vehicles x intersections) np.sqrt loop (_update_vehicles :251-269, its CPU hot spot) is a
// semantic no-op and has no device counterpart; the per-env state collapses to 58 dwords:
//   9 lights (phase:2 timer:5), 36 queues (len:6 dest:6 wait:16), 9 passed, 9 total_wait, counters, RNG cursor,
//   float64 total_reward — 15 uint4 columns (240 B), struct-of-arrays, all in VGPRs during a step.
// Integer dynamics are exact; the reward (np.var included, NumPy's pairwise order) and the obs quotients are
// float64 in the reference's operation order, so the float32 obs and the reward are bit-identical to the CPU.
// Draw counts per step are data dependent (a light entering green: randint(5,30); a spawn: random(), randint x2,
// 1-4 random.choice hops), so the env's MT19937 window is parked in LDS (LdsDraws) and indexed by a per-lane
// cursor.  The (N,130) float32 obs is staged through LDS in five 26-dword chunks per wave and written with
// fully used 256-byte store instructions.
#include <cstring>
#include <utility>
#include <vector>

#include "cge_device.hpp"
#include "cge_host.hpp"

namespace cge {
namespace traffic {

constexpr int NI = 9;
constexpr int NQ = 36;
constexpr int OBS = 130;
constexpr int CW = 26;             // obs dwords per staged chunk (130 = 5 x 26)
constexpr int ROW = 27;            // LDS row stride, odd
constexpr int DW = 16;             // MT words per draw-queue fill: one step() call
constexpr int DWR = 48;            // fused rollout: a window lasts several steps, refilled wave-convergently (ensure)
constexpr int BLOCK = 64;
constexpr int COLS = 15;           // uint4 columns per env (58 of 60 dwords used)
enum { NS_GREEN = 0, NS_YELLOW = 1, EW_GREEN = 2, EW_YELLOW = 3 };
enum { NORTH = 0, EAST = 1, SOUTH = 2, WEST = 3 };

struct Cfg {
    double spawn_rate;
    int32_t max_vehicles, max_steps;
};

struct Params {
    uint4 *state;
    uint32_t *mt;
    int64_t n, env0;
    Cfg cfg;
    int32_t mode;
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
};

struct Env {
    uint32_t light[NI];     // phase | timer << 2
    uint32_t q[NQ];         // len:6 | dest:6 << 6 | wait:16 << 12, queue 4*i + dir
    uint32_t passed[NI], tw[NI];
    uint32_t timestep, nveh, needs_reset, episodes, mt_pos, mt_pretw;
    double total_reward;

    __host__ __device__ __forceinline__ void unpack(const uint32_t *raw) {
#pragma unroll
        for (int i = 0; i < NI; ++i) light[i] = (raw[i >> 2] >> ((i & 3) * 8)) & 0xFFu;
#pragma unroll
        for (int k = 0; k < NQ; ++k) q[k] = raw[3 + k];
#pragma unroll
        for (int i = 0; i < NI; ++i) passed[i] = (raw[39 + (i >> 1)] >> ((i & 1) * 16)) & 0xFFFFu;
#pragma unroll
        for (int i = 0; i < NI; ++i) tw[i] = raw[44 + i];
        const uint32_t m0 = raw[53], m1 = raw[54];
        timestep = m0 & 0xFFFFu; nveh = (m0 >> 16) & 63u; needs_reset = (m0 >> 22) & 1u;
        mt_pos = m1 & 1023u; mt_pretw = (m1 & 1024u) ? (uint32_t)MT_N : 0u;
        episodes = raw[55];
        const uint64_t u = ((uint64_t)raw[57] << 32) | raw[56];
        memcpy(&total_reward, &u, 8);
    }
    __host__ __device__ __forceinline__ void pack(uint32_t *raw) const {
        raw[0] = raw[1] = raw[2] = 0;
#pragma unroll
        for (int i = 0; i < NI; ++i) raw[i >> 2] |= (light[i] & 0xFFu) << ((i & 3) * 8);
#pragma unroll
        for (int k = 0; k < NQ; ++k) raw[3 + k] = q[k];
#pragma unroll
        for (int j = 0; j < 5; ++j) raw[39 + j] = 0;
#pragma unroll
        for (int i = 0; i < NI; ++i) raw[39 + (i >> 1)] |= (passed[i] & 0xFFFFu) << ((i & 1) * 16);
#pragma unroll
        for (int i = 0; i < NI; ++i) raw[44 + i] = tw[i];
        raw[53] = timestep | (nveh << 16) | (needs_reset << 22);
        raw[54] = mt_pos | (mt_pretw ? 1024u : 0u);
        raw[55] = episodes;
        uint64_t u;
        memcpy(&u, &total_reward, 8);
        raw[56] = (uint32_t)u; raw[57] = (uint32_t)(u >> 32);
        raw[58] = raw[59] = 0;
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
    __device__ __forceinline__ void reset() {                      // environment.py:141-166 (no draws)
#pragma unroll
        for (int i = 0; i < NI; ++i) { light[i] = 0; passed[i] = 0; tw[i] = 0; }   // TrafficLight(): NS_GREEN, timer 0
#pragma unroll
        for (int k = 0; k < NQ; ++k) q[k] = 0;
        timestep = 0; nveh = 0; needs_reset = 0; total_reward = 0.0;
    }
};

// _spawn_vehicles :222-249 + generate_vehicle_route utils.py:174-193 (called with nveh < max_vehicles)
template <class DRAWS>
__device__ __forceinline__ void spawn(Env &e, const Cfg &c, DRAWS &d) {
    if (!(d.random53() < c.spawn_rate)) return;
    const uint32_t start = d.randbelow(9u, 4);                      // random.randint(0, 8)
    const uint32_t hops = 1u + d.randbelow(4u, 3);                  // randint(2, 5) - 1
    uint32_t cur = start, first = 0;
    for (uint32_t k = 0; k < hops; ++k) {
        const uint32_t row = cur / 5u, col = cur - row * 5u;
        const uint32_t vN = row > 0, vS = row < 4, vW = col > 0, vE = col < 4;   // utils.py:206 order N, S, W, E
        const uint32_t cnt = vN + vS + vW + vE;
        const uint32_t r = d.randbelow(cnt, cnt == 4u ? 3 : 2);     // random.choice(neighbours)
        // r-th valid neighbour
        uint32_t idx = r, nxt;
        if (vN && idx == 0) nxt = cur - 5u;
        else {
            idx -= vN;
            if (vS && idx == 0) nxt = cur + 5u;
            else {
                idx -= vS;
                if (vW && idx == 0) nxt = cur - 1u;
                else nxt = cur + 1u;
            }
        }
        cur = nxt;
        if (k == 0) first = nxt;
    }
    uint32_t dir;                                                   // utils.py:230-248 (route[0] -> route[1])
    if (first + 5u == start) dir = NORTH;
    else if (first == start + 5u) dir = SOUTH;
    else if (first + 1u == start) dir = WEST;
    else dir = EAST;
    const uint32_t qi = start * 4u + dir;
    const uint32_t inc = 1u | ((cur == start) ? (1u << 6) : 0u);    // len += 1, dest += (destination == this intersection)
#pragma unroll
    for (int k = 0; k < NQ; ++k) e.q[k] += (qi == (uint32_t)k) ? inc : 0u;
    e.nveh += 1;
}

#ifdef CGE_TRAFFIC_TIMING
__device__ unsigned long long g_timing[4096 * 16];
#define TICK(k) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); const unsigned long long now_ = wall_clock64(); \
    if (threadIdx.x == 0 && blockIdx.x < 4096) { g_timing[blockIdx.x * 16 + k] += now_ - t_last; } t_last = now_; } while (0)
#define TICK_DECL unsigned long long t_last = wall_clock64();
#define TICK_ARG , unsigned long long &t_last
#define TICK_PASS , t_last
#else
#define TICK(k)
#define TICK_DECL
#define TICK_ARG
#define TICK_PASS
#endif
// one reference step() (:168-203); returns terminated, reward in float64
template <class DRAWS>
__device__ __forceinline__ bool env_step(Env &e, const Cfg &c, const uint32_t (&a)[NI], DRAWS &d, double &reward TICK_ARG) {
    e.timestep += 1;
    d.ensure_inline(12);
    TICK(1);                                                                      // typical step: 1-2 light timers + a spawn with 1-4 hops
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        uint32_t phase = e.light[i] & 3u, timer = e.light[i] >> 2;
        if (a[i] == 1u && phase != NS_GREEN) { phase = NS_GREEN; timer = 5; }          // _apply_actions :205-220
        else if (a[i] == 2u && phase != EW_GREEN) { phase = EW_GREEN; timer = 5; }
        e.light[i] = phase | (timer << 2);
    }
#pragma unroll
    for (int i = 0; i < NI; ++i) {                                                     // TrafficLight.update utils.py:79-97
        uint32_t phase = e.light[i] & 3u;
        int timer = (int)(e.light[i] >> 2) - 1;
        if (timer <= 0) {
            phase = (phase + 1u) & 3u;
            timer = (phase & 1u) ? 3 : 5 + (int)d.randbelow(26u, 5);                   // random.randint(5, 30)
        }
        e.light[i] = phase | ((uint32_t)timer << 2);
    }
    TICK(2);
    if (e.nveh < (uint32_t)c.max_vehicles) spawn(e, c, d);                             // returns BEFORE drawing when full
    TICK(3);
    uint32_t tp = 0, twsum = 0, tq = 0, qt[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {                                                     // process_vehicles utils.py:141-163
        const uint32_t phase = e.light[i] & 3u;
        qt[i] = 0;
#pragma unroll
        for (int dd = 0; dd < 4; ++dd) {
            uint32_t &qq = e.q[i * 4 + dd];
            const uint32_t len = qq & 63u;
            const bool pass = (dd == NORTH || dd == SOUTH) ? phase == NS_GREEN : phase == EW_GREEN;
            if (len) {
                if (pass) {
                    e.passed[i] += len;
                    e.nveh -= (qq >> 6) & 63u;                                         // reached destination -> removed
                    qq = 0;
                } else {
                    qq += len << 12;
                    e.tw[i] += len;
                }
            }
            qt[i] += qq & 63u;
        }
        tp += e.passed[i]; twsum += e.tw[i]; tq += qt[i];
    }
    // _calculate_reward :287-311; np.var over the 9 queue totals (population variance, NumPy pairwise order)
    const double mean = (double)tq / 9.0;
    double x[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) { const double dv = (double)qt[i] - mean; x[i] = dv * dv; }
    double var = ((x[0] + x[1]) + (x[2] + x[3])) + ((x[4] + x[5]) + (x[6] + x[7]));
    var += x[8];
    var = var / 9.0;
    double r = 0.0;
    r += (double)tp * 1.0;
    r += (double)twsum * -0.1;
    r += (double)tq * -0.05;
    r += 0.5 / (1.0 + var);
    e.total_reward += r;
    reward = r;
    TICK(4);
    return e.timestep >= (uint32_t)c.max_steps;
}

struct ObsTotals { uint32_t tp, tw, tq; };

// _get_observation :313-363, element J of the 130 (compile-time J)
template <int J>
__device__ __forceinline__ float obs_val(const Env &e, const ObsTotals &t) {
    if constexpr (J < 36) return (e.light[J / 4] & 3u) == (uint32_t)(J % 4) ? 1.0f : 0.0f;
    else if constexpr (J < 72) { const uint32_t len = e.q[J - 36] & 63u; return (float)(len < 20u ? len : 20u); }
    else if constexpr (J < 108) {
        const uint32_t qq = e.q[J - 72], len = qq & 63u;
        const double avg = len ? (double)(qq >> 12) / (double)len : 0.0;
        return (float)(avg < 100.0 ? avg : 100.0);
    } else if constexpr (J < 126) {
        constexpr int k = J - 108;
        if constexpr (k % 2 == 0) return (float)e.passed[k / 2];
        else return (float)(e.tw[k / 2] < 1000u ? e.tw[k / 2] : 1000u);
    } else if constexpr (J == 126) return (float)e.nveh;
    else if constexpr (J == 127) { const double v = (double)t.tw / (double)(t.tp > 1u ? t.tp : 1u); return (float)(v < 100.0 ? v : 100.0); }
    else if constexpr (J == 128) { const double v = (double)t.tq / 9.0; return (float)(v < 50.0 ? v : 50.0); }
    else return (float)((double)t.tp / 9.0);
}

template <int BASE, int... Js>
__device__ __forceinline__ void fill_chunk(const Env &e, const ObsTotals &t, float (&out)[sizeof...(Js)], std::integer_sequence<int, Js...>) {
    ((out[Js] = obs_val<BASE + Js>(e, t)), ...);
}

template <int BASE, int... Js>
__device__ __forceinline__ void stage_chunk(const Env &e, const ObsTotals &t, float *row, std::integer_sequence<int, Js...>) {
    ((row[Js] = obs_val<BASE + Js>(e, t)), ...);
}

// writes the wave's 64 obs rows (rows with their bit in rowmask) to dst (+ row*130 floats).
// OWN (the fused rollout): every lane streams its own row in 16-byte stores straight from registers (cge_device.hpp:
// store_own_row), 32 values at a time + the last two — 96 -> 77 us per 262,144-env rollout step (A/B on one box, round 2).
// !OWN (step()): five 26-value chunks staged in LDS (row stride 27) and written with coalesced dword stores; the own-row form
// was 5 % slower there (99 vs 95 us), where all waves of the launch reach their stores together.
template <bool OWN>
__device__ __forceinline__ void observe(const Env &e, int64_t nrows, float *__restrict__ dst, unsigned long long rowmask,
                                        uint32_t *__restrict__ tile) {
    const uint32_t lane = threadIdx.x & 63u;
    ObsTotals t{0, 0, 0};
#pragma unroll
    for (int i = 0; i < NI; ++i) { t.tp += e.passed[i]; t.tw += e.tw[i]; }
#pragma unroll
    for (int k = 0; k < NQ; ++k) t.tq += e.q[k] & 63u;
    if constexpr (OWN) {
        const bool mine = (int64_t)lane < nrows && ((rowmask >> lane) & 1ull);
        float *row = dst + (int64_t)lane * OBS;
        static_assert(OBS == 4 * 32 + 2, "the chunk sequence below covers 130 values");
#define CGE_CHUNK(C)                                                                   \
        {                                                                              \
            float out[32];                                                             \
            fill_chunk<C * 32>(e, t, out, std::make_integer_sequence<int, 32>{});      \
            store_own_row<32>(row, C * 32, out, mine);                                 \
        }
        CGE_CHUNK(0) CGE_CHUNK(1) CGE_CHUNK(2) CGE_CHUNK(3)
#undef CGE_CHUNK
        float last[2];
        fill_chunk<128>(e, t, last, std::make_integer_sequence<int, 2>{});
        store_own_row<2>(row, 128, last, mine);
    } else {
        float *row = reinterpret_cast<float *>(tile) + lane * ROW;
#define CGE_CHUNK(C)                                                                                   \
        stage_chunk<C * CW>(e, t, row, std::make_integer_sequence<int, CW>{});                         \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                             \
        {                                                                                              \
            uint32_t r = lane / (uint32_t)CW, col = lane - r * (uint32_t)CW;                           \
            _Pragma("unroll 1") for (int m = 0; m < CW; ++m) {                                         \
                if ((int64_t)r < nrows && ((rowmask >> r) & 1ull))                                     \
                    reinterpret_cast<uint32_t *>(dst)[(int64_t)r * OBS + C * CW + col] = tile[r * ROW + col]; \
                col += 64u % CW; r += 64u / CW;                                                        \
                if (col >= (uint32_t)CW) { col -= CW; r += 1u; }                                       \
            }                                                                                          \
        }                                                                                              \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        CGE_CHUNK(0) CGE_CHUNK(1) CGE_CHUNK(2) CGE_CHUNK(3) CGE_CHUNK(4)
#undef CGE_CHUNK
    }
}

template <bool ROLLOUT>
__global__ __launch_bounds__(BLOCK) void step_kernel(Params p) {
    __shared__ uint32_t tile[64 * ROW];
    constexpr int W = ROLLOUT ? DWR : DW, DROW = W + 1;
    __shared__ uint32_t draws[64 * DROW];
    const int64_t i0 = (int64_t)blockIdx.x * BLOCK;
    const int64_t i = i0 + threadIdx.x;
    const bool live = i < p.n;
    const int64_t li = live ? i : i0;
    const int64_t nrows = p.n - i0 < 64 ? p.n - i0 : 64;
    Env e;
    e.load(p.state, p.n, li);
    LdsDrawsCall<W> d(draws + (threadIdx.x & 63u) * DROW, p.mt + li * MT_STRIDE, e.mt_pos, e.mt_pretw);
    const uint64_t key = ROLLOUT ? hash_env_key(p.a_seed, (uint64_t)(p.env0 + li)) : 0;
    double rsum = 0.0;
    int32_t dcount = 0;
    const int ksteps = ROLLOUT ? p.k_steps : 1;
    TICK_DECL
#pragma unroll 1
    for (int t = 0; t < ksteps; ++t) {
        double reward = 0.0;
        bool term = false, reset_now = false;
        if (live) {
            if (p.mode == CGE_AUTORESET_NEXT_STEP && e.needs_reset) {
                reset_now = true;
            } else {
                uint32_t a[NI];
                if (p.actions) {
                    const int32_t *ap = p.actions + ((int64_t)t * p.n + i) * NI;
#pragma unroll
                    for (int j = 0; j < NI; ++j) a[j] = (uint32_t)ap[j];
                } else {
#pragma unroll
                    for (int j = 0; j < NI; ++j) a[j] = hash_action_from_key(key, (uint64_t)(p.t0 + t), 3u, (uint32_t)j);
                }
                TICK(0);
                if (!ROLLOUT && e.nveh < (uint32_t)p.cfg.max_vehicles) d.fill_inline();      // a spawn attempt always draws: fetch the window now
                term = env_step(e, p.cfg, a, d, reward TICK_PASS);
                if (!ROLLOUT) d.flush();                          // a rollout keeps its window across steps
                if (term) {
                    e.episodes += 1;
                    if (p.ep_ret) p.ep_ret[i] = e.total_reward;            // environment.py:189 accumulates it, reset() zeroes it (:150)
                    if (p.ep_len) p.ep_len[i] = (int32_t)e.timestep;
                    if (p.mode == CGE_AUTORESET_SAME_STEP) reset_now = true;
                    else if (p.mode == CGE_AUTORESET_NEXT_STEP) e.needs_reset = 1;
                }
            }
        }
        const unsigned long long fin_mask = __ballot(live && term && reset_now);
        if (fin_mask && p.final_obs) observe<ROLLOUT>(e, nrows, p.final_obs + i0 * OBS, fin_mask, tile);   // terminal obs (SAME_STEP)
        if (reset_now) e.reset();
        TICK(5);
        if (p.obs) observe<ROLLOUT>(e, nrows, p.obs + (int64_t)t * p.obs_step_stride + i0 * OBS, ~0ull, tile);
        TICK(6);
#ifdef CGE_TRAFFIC_TIMING
        if (threadIdx.x == 0 && blockIdx.x < 4096) g_timing[blockIdx.x * 16 + 15] += 1;
#endif
        if (live) {
            if (ROLLOUT) {
                rsum += reward;
                dcount += term ? 1 : 0;
                if (p.reward) p.reward[(int64_t)t * p.n + i] = (float)reward;
                if (p.terminated) p.terminated[(int64_t)t * p.n + i] = term ? 1 : 0;
            } else {
                p.reward[i] = (float)reward;
                p.terminated[i] = term ? 1 : 0;
                if (p.truncated) p.truncated[i] = 0;
            }
        }
    }
    if (live) {
        if (ROLLOUT) d.flush();                                   // the rollout's window is written back once, here
        e.mt_pos = d.pos; e.mt_pretw = d.pretw;
        e.store(p.state, p.n, i);
        if (ROLLOUT) {
            if (p.reward_sum) p.reward_sum[i] = rsum;
            if (p.done_count) p.done_count[i] = dcount;
        }
    }
}

__global__ __launch_bounds__(BLOCK) void reset_kernel(Params p) {
    __shared__ uint32_t tile[64 * ROW];
    const int64_t i0 = (int64_t)blockIdx.x * BLOCK;
    const int64_t i = i0 + threadIdx.x;
    const bool live = i < p.n;
    const int64_t nrows = p.n - i0 < 64 ? p.n - i0 : 64;
    Env e;
    e.load(p.state, p.n, live ? i : i0);
    if (live && (!p.mask || p.mask[i])) {
        e.reset();
        e.store(p.state, p.n, i);
    }
    if (p.obs) observe<false>(e, nrows, p.obs + i0 * OBS, ~0ull, tile);
}

__global__ __launch_bounds__(256) void rewind_kernel(uint4 *state, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    uint4 v = state[13 * n + i];          // dwords 52..55: m1 (cursor) is dword 54
    v.z = 0;
    state[13 * n + i] = v;
}

__global__ __launch_bounds__(256) void info_kernel(const uint4 *__restrict__ state, int64_t n, int field, int idx, int32_t *__restrict__ out,
                                                   double *__restrict__ out64) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    Env e;
    e.load(state, n, i);
    if (out64) { out64[i] = e.total_reward; return; }
    int32_t v = 0;
    uint32_t li = 0, pi = 0, ti = 0, qi = 0;
#pragma unroll
    for (int k = 0; k < NI; ++k) if (idx == k) { li = e.light[k]; pi = e.passed[k]; ti = e.tw[k]; }
#pragma unroll
    for (int k = 0; k < NQ; ++k) if (idx == k) qi = e.q[k];
    switch (field) {
        case CGE_TRAFFIC_INFO_TIMESTEP: v = (int32_t)e.timestep; break;
        case CGE_TRAFFIC_INFO_NUM_VEHICLES: v = (int32_t)e.nveh; break;
        case CGE_TRAFFIC_INFO_LIGHT_PHASE: v = (int32_t)(li & 3u); break;
        case CGE_TRAFFIC_INFO_LIGHT_TIMER: v = (int32_t)(li >> 2); break;
        case CGE_TRAFFIC_INFO_VEHICLES_PASSED: v = (int32_t)pi; break;
        case CGE_TRAFFIC_INFO_TOTAL_WAITING_TIME: v = (int32_t)ti; break;
        case CGE_TRAFFIC_INFO_QUEUE_LEN: v = (int32_t)(qi & 63u); break;
        case CGE_TRAFFIC_INFO_QUEUE_DEST: v = (int32_t)((qi >> 6) & 63u); break;
        case CGE_TRAFFIC_INFO_QUEUE_WAIT: v = (int32_t)(qi >> 12); break;
        case CGE_TRAFFIC_INFO_EPISODES: v = (int32_t)e.episodes; break;
        case CGE_TRAFFIC_INFO_NEEDS_RESET: v = (int32_t)e.needs_reset; break;
    }
    out[i] = v;
}

}  // namespace traffic
}  // namespace cge

using namespace cge;

struct cge_traffic : HandleBase {
    cge_traffic_config cfg{};
    uint4 *state = nullptr;
    uint32_t *mt = nullptr;

    traffic::Params params() const {
        traffic::Params p{};
        p.state = state; p.mt = mt; p.n = n; p.env0 = env0;
        p.cfg = traffic::Cfg{cfg.spawn_rate, cfg.max_vehicles, cfg.max_steps};
        p.mode = cfg.autoreset_mode;
        p.ep_ret = ep_ret; p.ep_len = ep_len;
        return p;
    }
    unsigned blocks() const { return (unsigned)((n + traffic::BLOCK - 1) / traffic::BLOCK); }
};

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
    if (cfg->grid_rows != 5 || cfg->grid_cols != 5 || cfg->num_intersections != 9) return CGE_ERR_UNSUPPORTED;
    if (cfg->autoreset_mode < 0 || cfg->autoreset_mode > 2 || cfg->max_vehicles < 0 || cfg->max_vehicles > 63 || cfg->max_steps <= 0 ||
        (int64_t)cfg->max_vehicles * cfg->max_steps > 65535 || !(cfg->spawn_rate >= 0.0))
        return CGE_ERR_INVALID_ARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return CGE_ERR_NO_DEVICE;
    cge_traffic *h = new cge_traffic();
    h->cfg = *cfg; h->n = n_envs; h->env0 = env_index0; h->device = device;
    DeviceGuard g(device);
    const size_t sb = (size_t)traffic::COLS * n_envs * sizeof(uint4), mb = (size_t)n_envs * MT_STRIDE * sizeof(uint32_t);
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
    hipLaunchKernelGGL(traffic::rewind_kernel, dim3((unsigned)((h->n + 255) / 256)), dim3(256), 0, as_stream(stream), h->state, h->n);
    CGE_TRY(h, hipGetLastError());
    return CGE_OK;
}

int cge_traffic_reset(cge_traffic *h, const uint8_t *mask, float *obs_out, void *stream) {
    if (!h) return CGE_ERR_INVALID_ARG;
    DeviceGuard g(h->device);
    traffic::Params p = h->params();
    p.mask = mask; p.obs = obs_out;
    hipLaunchKernelGGL(traffic::reset_kernel, dim3(h->blocks()), dim3(traffic::BLOCK), 0, as_stream(stream), p);
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
    hipLaunchKernelGGL(traffic::step_kernel<false>, dim3(h->blocks()), dim3(traffic::BLOCK), 0, as_stream(stream), p);
    h->last_kernel = "cge::traffic::step_kernel<false>";
    CGE_TRY(h, hipGetLastError());
    return CGE_OK;
}

int cge_traffic_rollout(cge_traffic *h, int32_t k_steps, const int32_t *actions, uint64_t action_seed, int64_t t0, float *obs_out,
                        int64_t obs_step_stride, float *reward_traj_out, uint8_t *terminated_traj_out, double *reward_sum_out,
                        int32_t *done_count_out, void *stream) {
    if (!h) return CGE_ERR_INVALID_ARG;
    if (k_steps < 0 || obs_step_stride < 0 || (obs_step_stride != 0 && obs_step_stride < h->n * traffic::OBS))
        return h->fail(CGE_ERR_INVALID_ARG, "cge_traffic_rollout: bad k_steps / obs_step_stride");
    if (k_steps == 0) return CGE_OK;
    DeviceGuard g(h->device);
    traffic::Params p = h->params();
    p.k_steps = k_steps; p.actions = actions; p.a_seed = action_seed; p.t0 = t0; p.obs = obs_out; p.obs_step_stride = obs_step_stride;
    p.reward = reward_traj_out; p.terminated = terminated_traj_out; p.reward_sum = reward_sum_out; p.done_count = done_count_out;
    hipLaunchKernelGGL(traffic::step_kernel<true>, dim3(h->blocks()), dim3(traffic::BLOCK), 0, as_stream(stream), p);
    h->last_kernel = "cge::traffic::step_kernel<true>";
    CGE_TRY(h, hipGetLastError());
    return CGE_OK;
}

int cge_traffic_info(cge_traffic *h, int32_t field_id, int32_t index, int32_t *out, void *stream) {
    if (!h) return CGE_ERR_INVALID_ARG;
    if (!out || field_id < 0 || field_id > CGE_TRAFFIC_INFO_NEEDS_RESET || index < 0 || index >= traffic::NQ)
        return h->fail(CGE_ERR_INVALID_ARG, "cge_traffic_info: bad field / index / null out");
    DeviceGuard g(h->device);
    hipLaunchKernelGGL(traffic::info_kernel, dim3((unsigned)((h->n + 255) / 256)), dim3(256), 0, as_stream(stream), h->state, h->n,
                       field_id, index, out, (double *)nullptr);
    CGE_TRY(h, hipGetLastError());
    return CGE_OK;
}

int cge_traffic_total_reward(cge_traffic *h, double *out, void *stream) {
    if (!h || !out) return CGE_ERR_INVALID_ARG;
    DeviceGuard g(h->device);
    hipLaunchKernelGGL(traffic::info_kernel, dim3((unsigned)((h->n + 255) / 256)), dim3(256), 0, as_stream(stream), h->state, h->n, 0, 0,
                       (int32_t *)nullptr, out);
    CGE_TRY(h, hipGetLastError());
    return CGE_OK;
}

size_t cge_traffic_state_bytes(const cge_traffic *h) { return h ? 6 * 4 + 8 + (4 * 9 + 3 * 36) * 4 + MT_N * 4 : 0; }

int cge_traffic_get_state(cge_traffic *h, void *host_buf, void *stream) {
    if (!h || !host_buf) return CGE_ERR_INVALID_ARG;
    DeviceGuard g(h->device);
    const int64_t n = h->n;
    std::vector<uint4> st((size_t)traffic::COLS * n);
    std::vector<uint32_t> mt((size_t)n * MT_STRIDE);
    CGE_TRY(h, hipStreamSynchronize(as_stream(stream)));
    CGE_TRY(h, hipMemcpy(st.data(), h->state, st.size() * sizeof(uint4), hipMemcpyDeviceToHost));
    CGE_TRY(h, hipMemcpy(mt.data(), h->mt, mt.size() * 4, hipMemcpyDeviceToHost));
    const size_t rec = cge_traffic_state_bytes(h);
    for (int64_t i = 0; i < n; ++i) {
        uint32_t raw[traffic::COLS * 4];
        for (int c = 0; c < traffic::COLS; ++c) {
            const uint4 v = st[(size_t)c * n + i];
            raw[4 * c] = v.x; raw[4 * c + 1] = v.y; raw[4 * c + 2] = v.z; raw[4 * c + 3] = v.w;
        }
        traffic::Env e;
        e.unpack(raw);
        uint8_t *p = (uint8_t *)host_buf + (size_t)i * rec;
        int32_t hd[6] = {(int32_t)e.timestep, (int32_t)e.nveh, (int32_t)e.needs_reset, 0, (int32_t)e.episodes, 0};
        int32_t *w = (int32_t *)(p + 32);
        for (int k = 0; k < 9; ++k) { w[k] = (int32_t)(e.light[k] & 3u); w[9 + k] = (int32_t)(e.light[k] >> 2); w[18 + k] = (int32_t)e.passed[k]; w[27 + k] = (int32_t)e.tw[k]; }
        for (int k = 0; k < 36; ++k) { w[36 + k] = (int32_t)(e.q[k] & 63u); w[72 + k] = (int32_t)((e.q[k] >> 6) & 63u); w[108 + k] = (int32_t)(e.q[k] >> 12); }
        uint32_t *omt = (uint32_t *)(w + 144);
        const uint32_t *src = &mt[(size_t)i * MT_STRIDE];
        memcpy(omt, src, MT_N * 4);
        if (e.mt_pretw >= (uint32_t)MT_N) hd[3] = (int32_t)e.mt_pos;
        else if (e.mt_pos == 0) hd[3] = MT_N;
        else {
            for (uint32_t k = e.mt_pos; k < (uint32_t)MT_N; ++k) {
                const uint32_t k1 = k + 1 == (uint32_t)MT_N ? 0 : k + 1, km = k + MT_M >= (uint32_t)MT_N ? k + MT_M - MT_N : k + MT_M;
                const uint32_t t = (omt[k] & 0x80000000u) | (omt[k1] & 0x7fffffffu);
                omt[k] = omt[km] ^ (t >> 1) ^ ((t & 1u) ? 0x9908b0dfu : 0u);
            }
            hd[3] = (int32_t)e.mt_pos;
        }
        memcpy(p, hd, 24);
        memcpy(p + 24, &e.total_reward, 8);
    }
    return CGE_OK;
}

int cge_traffic_set_state(cge_traffic *h, const void *host_buf, void *stream) {
    if (!h || !host_buf) return CGE_ERR_INVALID_ARG;
    DeviceGuard g(h->device);
    const int64_t n = h->n;
    std::vector<uint4> st((size_t)traffic::COLS * n);
    std::vector<uint32_t> mt((size_t)n * MT_STRIDE, 0u);
    const size_t rec = cge_traffic_state_bytes(h);
    for (int64_t i = 0; i < n; ++i) {
        const uint8_t *p = (const uint8_t *)host_buf + (size_t)i * rec;
        int32_t hd[6];
        memcpy(hd, p, 24);
        const int32_t *w = (const int32_t *)(p + 32);
        if (hd[0] < 0 || hd[0] > 65535 || hd[1] < 0 || hd[1] > 63 || hd[3] < 0 || hd[3] > MT_N)
            return h->fail(CGE_ERR_INVALID_ARG, "cge_traffic_set_state: malformed record");
        traffic::Env e;
        memset(&e, 0, sizeof e);
        e.timestep = (uint32_t)hd[0]; e.nveh = (uint32_t)hd[1]; e.needs_reset = (uint32_t)(hd[2] & 1); e.episodes = (uint32_t)hd[4];
        memcpy(&e.total_reward, p + 24, 8);
        for (int k = 0; k < 9; ++k) { e.light[k] = ((uint32_t)w[k] & 3u) | ((uint32_t)w[9 + k] << 2); e.passed[k] = (uint32_t)w[18 + k]; e.tw[k] = (uint32_t)w[27 + k]; }
        for (int k = 0; k < 36; ++k) e.q[k] = ((uint32_t)w[36 + k] & 63u) | (((uint32_t)w[72 + k] & 63u) << 6) | ((uint32_t)w[108 + k] << 12);
        if (hd[3] >= MT_N) { e.mt_pos = 0; e.mt_pretw = 0; } else { e.mt_pos = (uint32_t)hd[3]; e.mt_pretw = MT_N; }
        uint32_t raw[traffic::COLS * 4];
        e.pack(raw);
        for (int c = 0; c < traffic::COLS; ++c) st[(size_t)c * n + i] = make_uint4(raw[4 * c], raw[4 * c + 1], raw[4 * c + 2], raw[4 * c + 3]);
        memcpy(&mt[(size_t)i * MT_STRIDE], w + 144, MT_N * 4);
        memcpy(&mt[(size_t)i * MT_STRIDE + MT_N], &mt[(size_t)i * MT_STRIDE], MT_PAD * 4);     // mirror words (cge_device.hpp)
    }
    CGE_TRY(h, hipStreamSynchronize(as_stream(stream)));
    CGE_TRY(h, hipMemcpy(h->state, st.data(), st.size() * sizeof(uint4), hipMemcpyHostToDevice));
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
