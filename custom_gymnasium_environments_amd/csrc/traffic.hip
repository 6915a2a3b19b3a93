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
// semantic no-op and has no device counterpart; the per-env state collapses, for NI controlled intersections, to
//   NI lights (phase:2 timer:5), 4 NI queues (len:7 dest:7 wait:18), NI passed, NI total_wait, counters, RNG cursor,
//   float64 total_reward — 58 dwords = 15 uint4 columns (240 B) for the default 9, struct-of-arrays, all in VGPRs during a step.
// Layouts: the kernels are templates on the number of controlled intersections — 4, 9 and 16 are instantiated, the three the
// reference's own scripts construct (simple_test.py:71-76 (3,3)/4, config.py:6-7 (5,5)/9, USAGE_EXAMPLES.md:32-38 (6,6)/16) —
// and take the grid (rows, cols) at run time: routes walk the WHOLE grid (utils.py:196-214), so cells without an Intersection
// are visited too; the walk tracks (row, col) and never divides.  max_vehicles and spawn_rate are run-time values as well.
// Integer dynamics are exact; the reward (np.var included, NumPy's pairwise order) and the obs quotients are
// float64 in the reference's operation order, so the float32 obs and the reward are bit-identical to the CPU.
// Draw counts per step are data dependent (a light entering green: randint(5,30); a spawn: random(), randint x2,
// 1-4 random.choice hops), so the env's MT19937 window is parked in LDS (LdsDraws) and indexed by a per-lane
// cursor.  The (N, 14 NI + 4) float32 obs is staged through LDS in a few chunks per wave and written with
// fully used 256-byte store instructions (step()), or streamed row-wise from registers (fused rollout).
#include <cstring>
#include <utility>
#include <vector>

#include "cge_device.hpp"
#include "cge_host.hpp"

namespace cge {
namespace traffic {

constexpr int bitlen(int n) { int k = 0; while (n) { ++k; n >>= 1; } return k; }

template <int NI_>
struct Lay {
    static constexpr int NI = NI_;
    static constexpr int NQ = 4 * NI;
    static constexpr int OBS = 14 * NI + 4;                    // environment.py:108-121
    // state words: lights (8 bits each) | queues | passed (16 bits each) | total_wait | m0 m1 episodes total_reward(2) mt_old0
    static constexpr int O_Q = (NI + 3) / 4, O_P = O_Q + NQ, O_TW = O_P + (NI + 1) / 2, O_M = O_TW + NI, NW = O_M + 6;
    static constexpr int COLS = (NW + 3) / 4;                  // uint4 columns per env (15 for NI = 9)
    static constexpr int CW = 32, ROW = 33;                    // step(): obs staged through LDS 32 dwords at a time, odd row stride
    static constexpr int START_BITS = bitlen(NI);              // random.randint(0, NI-1): _randbelow(NI), k = NI.bit_length()
    static constexpr int HOPS = (NI < 5 ? NI : 5) - 1;         // route_length = randint(2, min(5, NI)) -> 1..HOPS hops (utils.py:181)
    static constexpr int HOP_BITS = bitlen(HOPS);
    static_assert(NI >= 2 && NI <= 16, "randint(2, min(5, NI)) needs NI >= 2; 16 intersections fill the register file");
};
constexpr int DW = 16;             // MT words per draw-queue fill: one step() call
constexpr int DWR = 48;            // fused rollout: a window lasts several steps, refilled wave-convergently (ensure)
constexpr int BLOCK = 64;
// queue word: len | dest << 7 | wait << 14 (len, dest <= max_vehicles <= 127; wait <= max_vehicles * max_steps < 2^18)
constexpr uint32_t QM = 127u;
constexpr int QS_DEST = 7, QS_WAIT = 14;
enum { NS_GREEN = 0, NS_YELLOW = 1, EW_GREEN = 2, EW_YELLOW = 3 };
enum { NORTH = 0, EAST = 1, SOUTH = 2, WEST = 3 };

struct Cfg {
    double spawn_rate;
    int32_t max_vehicles, max_steps;
    int32_t rows, cols;            // grid_size (environment.py:62): the grid the routes walk
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

template <int NI_>
struct Env {
    using L = Lay<NI_>;
    static constexpr int NI = L::NI, NQ = L::NQ, COLS = L::COLS;
    uint32_t light[NI];     // phase | timer << 2
    uint32_t q[NQ];         // len:7 | dest:7 << 7 | wait:18 << 14, queue 4*i + dir
    uint32_t passed[NI], tw[NI];
    uint32_t timestep, nveh, needs_reset, episodes, mt_pos, mt_pretw;
    uint32_t mt_old0;       // word 0 of the generator's current generation once the next one has overwritten it (mt_twist_chunk)
    double total_reward;

    __host__ __device__ __forceinline__ void unpack(const uint32_t *raw) {
#pragma unroll
        for (int i = 0; i < NI; ++i) light[i] = (raw[i >> 2] >> ((i & 3) * 8)) & 0xFFu;
#pragma unroll
        for (int k = 0; k < NQ; ++k) q[k] = raw[L::O_Q + k];
#pragma unroll
        for (int i = 0; i < NI; ++i) passed[i] = (raw[L::O_P + (i >> 1)] >> ((i & 1) * 16)) & 0xFFFFu;
#pragma unroll
        for (int i = 0; i < NI; ++i) tw[i] = raw[L::O_TW + i];
        const uint32_t m0 = raw[L::O_M], m1 = raw[L::O_M + 1];
        timestep = m0 & 0xFFFFu; nveh = (m0 >> 16) & 127u; needs_reset = (m0 >> 23) & 1u;
        mt_pos = m1 & 1023u; mt_pretw = mt_ready_decode((m1 >> 10) & 31u);     // ready mark of the twist-ahead stream (cge_device.hpp)
        episodes = raw[L::O_M + 2];
        const uint64_t u = ((uint64_t)raw[L::O_M + 4] << 32) | raw[L::O_M + 3];
        memcpy(&total_reward, &u, 8);
        mt_old0 = raw[L::O_M + 5];
    }
    __host__ __device__ __forceinline__ void pack(uint32_t *raw) const {
#pragma unroll
        for (int j = 0; j < L::O_Q; ++j) raw[j] = 0;
#pragma unroll
        for (int i = 0; i < NI; ++i) raw[i >> 2] |= (light[i] & 0xFFu) << ((i & 3) * 8);
#pragma unroll
        for (int k = 0; k < NQ; ++k) raw[L::O_Q + k] = q[k];
#pragma unroll
        for (int j = L::O_P; j < L::O_TW; ++j) raw[j] = 0;
#pragma unroll
        for (int i = 0; i < NI; ++i) raw[L::O_P + (i >> 1)] |= (passed[i] & 0xFFFFu) << ((i & 1) * 16);
#pragma unroll
        for (int i = 0; i < NI; ++i) raw[L::O_TW + i] = tw[i];
        raw[L::O_M] = timestep | (nveh << 16) | (needs_reset << 23);
        raw[L::O_M + 1] = mt_pos | ((mt_pretw > mt_pos ? mt_ready_encode(mt_pretw) : 0u) << 10);
        raw[L::O_M + 2] = episodes;
        uint64_t u;
        memcpy(&u, &total_reward, 8);
        raw[L::O_M + 3] = (uint32_t)u; raw[L::O_M + 4] = (uint32_t)(u >> 32);
        raw[L::O_M + 5] = mt_old0;
#pragma unroll
        for (int j = L::NW; j < COLS * 4; ++j) raw[j] = 0;
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

// _spawn_vehicles :222-249 + generate_vehicle_route utils.py:174-193 (called with nveh < max_vehicles).  The route is a random
// walk over the rows x cols grid (get_neighboring_intersections utils.py:196-214, neighbour order N, S, W, E); only where it
// starts (the queue), its first hop (the direction, utils.py:230-248) and whether it ends where it started are observable.
template <int NI, class DRAWS>
__device__ __forceinline__ void spawn(Env<NI> &e, const Cfg &c, DRAWS &d) {
    using L = Lay<NI>;
    if (!(d.random53() < c.spawn_rate)) return;
    const uint32_t start = d.randbelow((uint32_t)NI, L::START_BITS);   // random.randint(0, num_intersections - 1)
    const uint32_t hops = 1u + d.randbelow((uint32_t)L::HOPS, L::HOP_BITS);   // randint(2, min(5, num_intersections)) - 1
    const uint32_t rows = (uint32_t)c.rows, cols = (uint32_t)c.cols;
    const uint32_t srow = start / cols, scol = start - srow * cols;
    uint32_t row = srow, col = scol, dir = EAST;
    for (uint32_t k = 0; k < hops; ++k) {
        const uint32_t vN = row > 0, vS = row + 1 < rows, vW = col > 0, vE = col + 1 < cols;   // utils.py:206 order N, S, W, E
        const uint32_t cnt = vN + vS + vW + vE;                      // >= 1: the grid has rows * cols >= NI >= 2 cells (cge_traffic_create)
        const uint32_t r = d.randbelow(cnt, cnt == 4u ? 3 : (cnt == 1u ? 1 : 2));   // random.choice(neighbours)
        uint32_t idx = r, mv;                                       // the r-th valid neighbour
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
        if (k == 0) dir = mv;                                       // utils.py:230-248 (route[0] -> route[1])
    }
    const uint32_t qi = start * 4u + dir;
    const uint32_t inc = 1u | ((row == srow && col == scol) ? (1u << QS_DEST) : 0u);   // len += 1, dest += (destination == this intersection)
#pragma unroll
    for (int k = 0; k < 4 * NI; ++k) e.q[k] += (qi == (uint32_t)k) ? inc : 0u;
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
template <int NI, class DRAWS>
__device__ __forceinline__ bool env_step(Env<NI> &e, const Cfg &c, const uint32_t (&a)[NI], DRAWS &d, double &reward TICK_ARG) {
    e.timestep += 1;
    d.ensure_ahead(12, true);
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
    if (e.nveh < (uint32_t)c.max_vehicles) spawn<NI>(e, c, d);                         // returns BEFORE drawing when full
    TICK(3);
    uint32_t tp = 0, twsum = 0, tq = 0, qt[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {                                                     // process_vehicles utils.py:141-163
        const uint32_t phase = e.light[i] & 3u;
        qt[i] = 0;
#pragma unroll
        for (int dd = 0; dd < 4; ++dd) {
            uint32_t &qq = e.q[i * 4 + dd];
            const uint32_t len = qq & QM;
            const bool pass = (dd == NORTH || dd == SOUTH) ? phase == NS_GREEN : phase == EW_GREEN;
            if (len) {
                if (pass) {
                    e.passed[i] += len;
                    e.nveh -= (qq >> QS_DEST) & QM;                                    // reached destination -> removed
                    qq = 0;
                } else {
                    qq += len << QS_WAIT;
                    e.tw[i] += len;
                }
            }
            qt[i] += qq & QM;
        }
        tp += e.passed[i]; twsum += e.tw[i]; tq += qt[i];
    }
    // _calculate_reward :287-311; np.var over the NI queue totals (population variance).  NumPy's pairwise summation: fewer than
    // 8 terms are added left to right, otherwise eight accumulators run over whole blocks of 8, are combined as a tree, and the
    // remainder follows left to right
    const double mean = (double)tq / (double)NI;
    double x[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) { const double dv = (double)qt[i] - mean; x[i] = dv * dv; }
    double var;
    if constexpr (NI < 8) {
        var = x[0];
#pragma unroll
        for (int i = 1; i < NI; ++i) var += x[i];
    } else {
        double r8[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) r8[j] = x[j];
#pragma unroll
        for (int i = 8; i + 8 <= NI; i += 8)
#pragma unroll
            for (int j = 0; j < 8; ++j) r8[j] += x[i + j];
        var = ((r8[0] + r8[1]) + (r8[2] + r8[3])) + ((r8[4] + r8[5]) + (r8[6] + r8[7]));
#pragma unroll
        for (int i = NI - NI % 8; i < NI; ++i) var += x[i];
    }
    var = var / (double)NI;
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

// _get_observation :313-363, element J of the 14 NI + 4 (compile-time J)
template <int NI, int J>
__device__ __forceinline__ float obs_val(const Env<NI> &e, const ObsTotals &t) {
    if constexpr (J < 4 * NI) return (e.light[J / 4] & 3u) == (uint32_t)(J % 4) ? 1.0f : 0.0f;
    else if constexpr (J < 8 * NI) { const uint32_t len = e.q[J - 4 * NI] & QM; return (float)(len < 20u ? len : 20u); }
    else if constexpr (J < 12 * NI) {
        const uint32_t qq = e.q[J - 8 * NI], len = qq & QM;
        const double avg = len ? (double)(qq >> QS_WAIT) / (double)len : 0.0;
        return (float)(avg < 100.0 ? avg : 100.0);
    } else if constexpr (J < 14 * NI) {
        constexpr int k = J - 12 * NI;
        if constexpr (k % 2 == 0) return (float)e.passed[k / 2];
        else return (float)(e.tw[k / 2] < 1000u ? e.tw[k / 2] : 1000u);
    } else if constexpr (J == 14 * NI) return (float)e.nveh;
    else if constexpr (J == 14 * NI + 1) { const double v = (double)t.tw / (double)(t.tp > 1u ? t.tp : 1u); return (float)(v < 100.0 ? v : 100.0); }
    else if constexpr (J == 14 * NI + 2) { const double v = (double)t.tq / (double)NI; return (float)(v < 50.0 ? v : 50.0); }
    else return (float)((double)t.tp / (double)NI);
}

template <int NI, int BASE, int... Js>
__device__ __forceinline__ void fill_chunk(const Env<NI> &e, const ObsTotals &t, float (&out)[sizeof...(Js)], std::integer_sequence<int, Js...>) {
    ((out[Js] = obs_val<NI, BASE + Js>(e, t)), ...);
}

// (stored as the dwords they are read back as: float stores that are only ever read through the tile's uint32_t pointer are, by the
// type-based aliasing rules, dead to the compiler — it dropped whole chunks of them)
template <int NI, int BASE, int... Js>
__device__ __forceinline__ void stage_chunk(const Env<NI> &e, const ObsTotals &t, uint32_t *row, std::integer_sequence<int, Js...>) {
    ((row[Js] = __float_as_uint(obs_val<NI, BASE + Js>(e, t))), ...);
}

// OWN rows: chunk C of OWN_CH values (or the last OBS % OWN_CH) straight from registers.  (Chunks of 16 or 8 values leave the
// rollout kernel at 195-197 VGPRs, round 3: the register peak is in env_step, not here.)
constexpr int OWN_CH = 32;
template <int NI, int C>
__device__ __forceinline__ void own_chunks(const Env<NI> &e, const ObsTotals &t, float *row, bool mine) {
    constexpr int OBS = Lay<NI>::OBS, N = OBS - OWN_CH * C < OWN_CH ? OBS - OWN_CH * C : OWN_CH;
    if constexpr (N > 0) {
        float out[N];
        fill_chunk<NI, C * OWN_CH>(e, t, out, std::make_integer_sequence<int, N>{});
        store_own_row<N>(row, C * OWN_CH, out, mine);
        own_chunks<NI, C + 1>(e, t, row, mine);
    }
}
// staged rows: 32 values at a time through the wave's LDS tile (row stride 33: the row writes and the reads below are both
// conflict-free), then 16-byte stores — 8 lanes cover one row's 128 contiguous bytes, one instruction 8 rows — and the last
// OBS % 32 values (always even) in 8-byte stores.  Rows start 8-byte aligned (OBS * 4 = 8 (7 NI + 2)).
template <int NI, int C>
__device__ __forceinline__ void staged_chunks(const Env<NI> &e, const ObsTotals &t, int64_t nrows, float *__restrict__ dst,
                                              unsigned long long rowmask, uint32_t *__restrict__ tile) {
    using L = Lay<NI>;
    constexpr int N = L::OBS - L::CW * C < L::CW ? L::OBS - L::CW * C : L::CW;
    if constexpr (N > 0) {
        const uint32_t lane = threadIdx.x & 63u;
        stage_chunk<NI, C * L::CW>(e, t, tile + lane * L::ROW, std::make_integer_sequence<int, N>{});
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        uint32_t *out = reinterpret_cast<uint32_t *>(dst) + C * L::CW;
        if constexpr (N == L::CW) {
            const uint32_t seg = lane & 7u;
#pragma unroll 2
            for (uint32_t r = lane >> 3; r < 64u; r += 8u) {
                const uint32_t *src = tile + r * L::ROW + seg * 4u;
                const Piece16 v{src[0], src[1], src[2], src[3]};
                if ((int64_t)r < nrows && ((rowmask >> r) & 1ull)) *reinterpret_cast<Piece16 *>(out + (int64_t)r * L::OBS + seg * 4u) = v;
            }
        } else {
            constexpr uint32_t PAIRS = N / 2;
            static_assert(N % 2 == 0, "14 NI + 4 is even");
#pragma unroll 2
            for (uint32_t k = lane; k < 64u * PAIRS; k += 64u) {
                const uint32_t r = k / PAIRS, c2 = (k - r * PAIRS) * 2u;
                const Piece8 v{tile[r * L::ROW + c2], tile[r * L::ROW + c2 + 1u]};
                if ((int64_t)r < nrows && ((rowmask >> r) & 1ull)) *reinterpret_cast<Piece8 *>(out + (int64_t)r * L::OBS + c2) = v;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        staged_chunks<NI, C + 1>(e, t, nrows, dst, rowmask, tile);
    }
}

// writes the wave's 64 obs rows (rows with their bit in rowmask) to dst (+ row*130 floats).
// OWN (the fused rollout): every lane streams its own row in 16-byte stores straight from registers (cge_device.hpp:
// store_own_row), 32 values at a time + the last two — 96 -> 77 us per 262,144-env rollout step (A/B on one box, round 2).
// !OWN (step()): staged in LDS 32 values at a time and written 8 rows per 16-byte store instruction (staged_chunks); the own-row
// form is slower there (90.6 vs 81.3 us, round 3), where all waves of the launch reach their stores together.
template <int NI, bool OWN>
__device__ __forceinline__ void observe(const Env<NI> &e, int64_t nrows, float *__restrict__ dst, unsigned long long rowmask,
                                        uint32_t *__restrict__ tile) {
    using L = Lay<NI>;
    const uint32_t lane = threadIdx.x & 63u;
    ObsTotals t{0, 0, 0};
#pragma unroll
    for (int i = 0; i < NI; ++i) { t.tp += e.passed[i]; t.tw += e.tw[i]; }
#pragma unroll
    for (int k = 0; k < 4 * NI; ++k) t.tq += e.q[k] & QM;
    if constexpr (OWN) {
        const bool mine = (int64_t)lane < nrows && ((rowmask >> lane) & 1ull);
        own_chunks<NI, 0>(e, t, dst + (int64_t)lane * L::OBS, mine);
    } else {
        staged_chunks<NI, 0>(e, t, nrows, dst, rowmask, tile);
    }
}

template <int NI, bool ROLLOUT>
__global__ __launch_bounds__(BLOCK) void step_kernel(Params p) {
    using L = Lay<NI>;
    constexpr int OBS = L::OBS;
    __shared__ uint32_t tile[64 * L::ROW];
    constexpr int W = ROLLOUT ? DWR : DW, DROW = W + 1;
    __shared__ uint32_t draws[64 * DROW];
    const int64_t i0 = (int64_t)blockIdx.x * BLOCK;
    const int64_t i = i0 + threadIdx.x;
    const bool live = i < p.n;
    const int64_t li = live ? i : i0;
    const int64_t nrows = p.n - i0 < 64 ? p.n - i0 : 64;
    Env<NI> e;
    e.load(p.state, p.n, li);
    LdsDrawsCall<W> d(draws + (threadIdx.x & 63u) * DROW, p.mt + li * MT_STRIDE, e.mt_pos, e.mt_pretw);
    d.old0 = e.mt_old0;
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
                (void)0;
                term = env_step<NI>(e, p.cfg, a, d, reward TICK_PASS);
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
        if (fin_mask && p.final_obs) observe<NI, ROLLOUT>(e, nrows, p.final_obs + i0 * OBS, fin_mask, tile);   // terminal obs (SAME_STEP)
        if (reset_now) e.reset();
        TICK(5);
        if (p.obs) observe<NI, ROLLOUT>(e, nrows, p.obs + (int64_t)t * p.obs_step_stride + i0 * OBS, ~0ull, tile);
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
        e.mt_pos = d.pos; e.mt_pretw = d.pretw; e.mt_old0 = d.old0;
        e.store(p.state, p.n, i);
        if (ROLLOUT) {
            if (p.reward_sum) p.reward_sum[i] = rsum;
            if (p.done_count) p.done_count[i] = dcount;
        }
    }
}

template <int NI>
__global__ __launch_bounds__(BLOCK) void reset_kernel(Params p) {
    using L = Lay<NI>;
    constexpr int OBS = L::OBS;
    __shared__ uint32_t tile[64 * L::ROW];
    const int64_t i0 = (int64_t)blockIdx.x * BLOCK;
    const int64_t i = i0 + threadIdx.x;
    const bool live = i < p.n;
    const int64_t nrows = p.n - i0 < 64 ? p.n - i0 : 64;
    Env<NI> e;
    e.load(p.state, p.n, live ? i : i0);
    if (live && (!p.mask || p.mask[i])) {
        e.reset();
        e.store(p.state, p.n, i);
    }
    if (p.obs) observe<NI, false>(e, nrows, p.obs + i0 * OBS, ~0ull, tile);
}

template <int NI>
__global__ __launch_bounds__(256) void rewind_kernel(uint4 *state, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    constexpr int W = Lay<NI>::O_M + 1;   // m1: the generator cursor
    uint4 v = state[(int64_t)(W / 4) * n + i];
    if (W % 4 == 0) v.x = 0; else if (W % 4 == 1) v.y = 0; else if (W % 4 == 2) v.z = 0; else v.w = 0;
    state[(int64_t)(W / 4) * n + i] = v;
}

template <int NI>
__global__ __launch_bounds__(256) void info_kernel(const uint4 *__restrict__ state, int64_t n, int field, int idx, int32_t *__restrict__ out,
                                                   double *__restrict__ out64) {
    constexpr int NQ = 4 * NI;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    Env<NI> e;
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
        case CGE_TRAFFIC_INFO_QUEUE_LEN: v = (int32_t)(qi & QM); break;
        case CGE_TRAFFIC_INFO_QUEUE_DEST: v = (int32_t)((qi >> QS_DEST) & QM); break;
        case CGE_TRAFFIC_INFO_QUEUE_WAIT: v = (int32_t)(qi >> QS_WAIT); break;
        case CGE_TRAFFIC_INFO_EPISODES: v = (int32_t)e.episodes; break;
        case CGE_TRAFFIC_INFO_NEEDS_RESET: v = (int32_t)e.needs_reset; break;
    }
    out[i] = v;
}


// ------------------------------------------------------------------ one set of kernels per compiled-in layout
struct Ops {
    int ni, cols, obs;
    const char *step_name, *rollout_name;
    void (*step)(const Params &, unsigned, hipStream_t);
    void (*rollout)(const Params &, unsigned, hipStream_t);
    void (*reset)(const Params &, unsigned, hipStream_t);
    void (*rewind)(uint4 *, int64_t, hipStream_t);
    void (*info)(const uint4 *, int64_t, int, int, int32_t *, double *, hipStream_t);
    void (*to_record)(const uint32_t *raw, int32_t *hd, double *total_reward, int32_t *w, uint32_t *mt_pos, uint32_t *mt_pretw, uint32_t *mt_old0);
    void (*from_record)(const int32_t *hd, double total_reward, const int32_t *w, uint32_t *raw);
};

// device record <-> the canonical record's fields (w: phase[ni], timer[ni], passed[ni], total_wait[ni], qlen, qdest, qwait [4 ni])
template <int NI>
void to_record(const uint32_t *raw, int32_t *hd, double *total_reward, int32_t *w, uint32_t *mt_pos, uint32_t *mt_pretw, uint32_t *mt_old0) {
    Env<NI> e;
    e.unpack(raw);
    hd[0] = (int32_t)e.timestep; hd[1] = (int32_t)e.nveh; hd[2] = (int32_t)e.needs_reset; hd[3] = 0; hd[4] = (int32_t)e.episodes; hd[5] = 0;
    *total_reward = e.total_reward; *mt_pos = e.mt_pos; *mt_pretw = e.mt_pretw; *mt_old0 = e.mt_old0;
    for (int k = 0; k < NI; ++k) { w[k] = (int32_t)(e.light[k] & 3u); w[NI + k] = (int32_t)(e.light[k] >> 2); w[2 * NI + k] = (int32_t)e.passed[k]; w[3 * NI + k] = (int32_t)e.tw[k]; }
    for (int k = 0; k < 4 * NI; ++k) { w[4 * NI + k] = (int32_t)(e.q[k] & QM); w[8 * NI + k] = (int32_t)((e.q[k] >> QS_DEST) & QM); w[12 * NI + k] = (int32_t)(e.q[k] >> QS_WAIT); }
}
template <int NI>
void from_record(const int32_t *hd, double total_reward, const int32_t *w, uint32_t *raw) {
    Env<NI> e;
    memset(&e, 0, sizeof e);
    e.timestep = (uint32_t)hd[0]; e.nveh = (uint32_t)hd[1]; e.needs_reset = (uint32_t)(hd[2] & 1); e.episodes = (uint32_t)hd[4];
    e.total_reward = total_reward;
    for (int k = 0; k < NI; ++k) { e.light[k] = ((uint32_t)w[k] & 3u) | ((uint32_t)w[NI + k] << 2); e.passed[k] = (uint32_t)w[2 * NI + k]; e.tw[k] = (uint32_t)w[3 * NI + k]; }
    for (int k = 0; k < 4 * NI; ++k) e.q[k] = ((uint32_t)w[4 * NI + k] & QM) | (((uint32_t)w[8 * NI + k] & QM) << QS_DEST) | ((uint32_t)w[12 * NI + k] << QS_WAIT);
    if (hd[3] >= MT_N) { e.mt_pos = 0; e.mt_pretw = 0; } else { e.mt_pos = (uint32_t)hd[3]; e.mt_pretw = MT_N; }
    e.pack(raw);
}

template <int NI>
Ops make_ops(const char *step_name, const char *rollout_name) {
    Ops o;
    o.ni = NI; o.cols = Lay<NI>::COLS; o.obs = Lay<NI>::OBS;
    o.step_name = step_name; o.rollout_name = rollout_name;
    o.step = [](const Params &p, unsigned blocks, hipStream_t s) { hipLaunchKernelGGL((step_kernel<NI, false>), dim3(blocks), dim3(BLOCK), 0, s, p); };
    o.rollout = [](const Params &p, unsigned blocks, hipStream_t s) { hipLaunchKernelGGL((step_kernel<NI, true>), dim3(blocks), dim3(BLOCK), 0, s, p); };
    o.reset = [](const Params &p, unsigned blocks, hipStream_t s) { hipLaunchKernelGGL(reset_kernel<NI>, dim3(blocks), dim3(BLOCK), 0, s, p); };
    o.rewind = [](uint4 *st, int64_t n, hipStream_t s) { hipLaunchKernelGGL(rewind_kernel<NI>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, st, n); };
    o.info = [](const uint4 *st, int64_t n, int field, int idx, int32_t *out, double *out64, hipStream_t s) {
        hipLaunchKernelGGL(info_kernel<NI>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, st, n, field, idx, out, out64);
    };
    o.to_record = to_record<NI>;
    o.from_record = from_record<NI>;
    return o;
}

static bool ops_for(int ni, Ops &o) {
    switch (ni) {
        case 4: o = make_ops<4>("cge::traffic::step_kernel<4, false>", "cge::traffic::step_kernel<4, true>"); return true;       // simple_test.py:71-76
        case 9: o = make_ops<9>("cge::traffic::step_kernel<9, false>", "cge::traffic::step_kernel<9, true>"); return true;       // config.py:7
        case 16: o = make_ops<16>("cge::traffic::step_kernel<16, false>", "cge::traffic::step_kernel<16, true>"); return true;   // USAGE_EXAMPLES.md:32-38
    }
    return false;
}

}  // namespace traffic
}  // namespace cge

using namespace cge;

struct cge_traffic : HandleBase {
    cge_traffic_config cfg{};
    traffic::Ops ops{};
    uint4 *state = nullptr;
    uint32_t *mt = nullptr;

    traffic::Params params() const {
        traffic::Params p{};
        p.state = state; p.mt = mt; p.n = n; p.env0 = env0;
        p.cfg = traffic::Cfg{cfg.spawn_rate, cfg.max_vehicles, cfg.max_steps, cfg.grid_rows, cfg.grid_cols};
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
    if (cfg->grid_rows < 1 || cfg->grid_cols < 1 || cfg->grid_rows > 64 || cfg->grid_cols > 64 || cfg->num_intersections < 1)
        return CGE_ERR_INVALID_ARG;
    // environment.py:79: num_intersections = min(num_intersections, rows * cols); kernels exist for 4, 9 and 16 of them
    const int cells = cfg->grid_rows * cfg->grid_cols;
    const int ni = cfg->num_intersections < cells ? cfg->num_intersections : cells;
    traffic::Ops ops;
    if (!traffic::ops_for(ni, ops)) return CGE_ERR_UNSUPPORTED;
    // queue word: len:7 dest:7 wait:18; timestep 16 bits
    if (cfg->autoreset_mode < 0 || cfg->autoreset_mode > 2 || cfg->max_vehicles < 0 || cfg->max_vehicles > 127 || cfg->max_steps <= 0 ||
        cfg->max_steps > 65535 || (int64_t)cfg->max_vehicles * cfg->max_steps > 262143 || !(cfg->spawn_rate >= 0.0))
        return CGE_ERR_INVALID_ARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return CGE_ERR_NO_DEVICE;
    cge_traffic *h = new cge_traffic();
    h->cfg = *cfg; h->cfg.num_intersections = ni; h->ops = ops; h->n = n_envs; h->env0 = env_index0; h->device = device;
    DeviceGuard g(device);
    const size_t sb = (size_t)ops.cols * n_envs * sizeof(uint4), mb = (size_t)n_envs * MT_STRIDE * sizeof(uint32_t);
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
    h->ops.rewind(h->state, h->n, as_stream(stream));
    CGE_TRY(h, hipGetLastError());
    return CGE_OK;
}

int cge_traffic_reset(cge_traffic *h, const uint8_t *mask, float *obs_out, void *stream) {
    if (!h) return CGE_ERR_INVALID_ARG;
    DeviceGuard g(h->device);
    traffic::Params p = h->params();
    p.mask = mask; p.obs = obs_out;
    h->ops.reset(p, h->blocks(), as_stream(stream));
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
    h->ops.step(p, h->blocks(), as_stream(stream));
    h->last_kernel = h->ops.step_name;
    CGE_TRY(h, hipGetLastError());
    return CGE_OK;
}

int cge_traffic_rollout(cge_traffic *h, int32_t k_steps, const int32_t *actions, uint64_t action_seed, int64_t t0, float *obs_out,
                        int64_t obs_step_stride, float *reward_traj_out, uint8_t *terminated_traj_out, double *reward_sum_out,
                        int32_t *done_count_out, void *stream) {
    if (!h) return CGE_ERR_INVALID_ARG;
    if (k_steps < 0 || obs_step_stride < 0 || (obs_step_stride != 0 && obs_step_stride < h->n * h->ops.obs))
        return h->fail(CGE_ERR_INVALID_ARG, "cge_traffic_rollout: bad k_steps / obs_step_stride");
    if (k_steps == 0) return CGE_OK;
    DeviceGuard g(h->device);
    traffic::Params p = h->params();
    p.k_steps = k_steps; p.actions = actions; p.a_seed = action_seed; p.t0 = t0; p.obs = obs_out; p.obs_step_stride = obs_step_stride;
    p.reward = reward_traj_out; p.terminated = terminated_traj_out; p.reward_sum = reward_sum_out; p.done_count = done_count_out;
    h->ops.rollout(p, h->blocks(), as_stream(stream));
    h->last_kernel = h->ops.rollout_name;
    CGE_TRY(h, hipGetLastError());
    return CGE_OK;
}

int cge_traffic_info(cge_traffic *h, int32_t field_id, int32_t index, int32_t *out, void *stream) {
    if (!h) return CGE_ERR_INVALID_ARG;
    const bool per_queue = field_id >= CGE_TRAFFIC_INFO_QUEUE_LEN && field_id <= CGE_TRAFFIC_INFO_QUEUE_WAIT;
    if (!out || field_id < 0 || field_id > CGE_TRAFFIC_INFO_NEEDS_RESET || index < 0 || index >= (per_queue ? 4 : 1) * h->ops.ni)
        return h->fail(CGE_ERR_INVALID_ARG, "cge_traffic_info: bad field / index / null out");
    DeviceGuard g(h->device);
    h->ops.info(h->state, h->n, field_id, index, out, nullptr, as_stream(stream));
    CGE_TRY(h, hipGetLastError());
    return CGE_OK;
}

int cge_traffic_total_reward(cge_traffic *h, double *out, void *stream) {
    if (!h || !out) return CGE_ERR_INVALID_ARG;
    DeviceGuard g(h->device);
    h->ops.info(h->state, h->n, 0, 0, nullptr, out, as_stream(stream));
    CGE_TRY(h, hipGetLastError());
    return CGE_OK;
}

size_t cge_traffic_state_bytes(const cge_traffic *h) { return h ? 6 * 4 + 8 + (size_t)16 * h->ops.ni * 4 + MT_N * 4 : 0; }

int cge_traffic_get_state(cge_traffic *h, void *host_buf, void *stream) {
    if (!h || !host_buf) return CGE_ERR_INVALID_ARG;
    DeviceGuard g(h->device);
    const int64_t n = h->n;
    const int cols = h->ops.cols, ni = h->ops.ni;
    std::vector<uint4> st((size_t)cols * n);
    std::vector<uint32_t> mt((size_t)n * MT_STRIDE);
    CGE_TRY(h, hipStreamSynchronize(as_stream(stream)));
    CGE_TRY(h, hipMemcpy(st.data(), h->state, st.size() * sizeof(uint4), hipMemcpyDeviceToHost));
    CGE_TRY(h, hipMemcpy(mt.data(), h->mt, mt.size() * 4, hipMemcpyDeviceToHost));
    const size_t rec = cge_traffic_state_bytes(h);
    std::vector<uint32_t> raw((size_t)cols * 4);
    for (int64_t i = 0; i < n; ++i) {
        for (int c = 0; c < cols; ++c) {
            const uint4 v = st[(size_t)c * n + i];
            raw[4 * c] = v.x; raw[4 * c + 1] = v.y; raw[4 * c + 2] = v.z; raw[4 * c + 3] = v.w;
        }
        uint8_t *p = (uint8_t *)host_buf + (size_t)i * rec;
        int32_t hd[6];
        int32_t *w = (int32_t *)(p + 32);
        double total_reward;
        uint32_t mt_pos, mt_pretw, mt_old0;
        h->ops.to_record(raw.data(), hd, &total_reward, w, &mt_pos, &mt_pretw, &mt_old0);
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
    const int cols = h->ops.cols, ni = h->ops.ni;
    std::vector<uint4> st((size_t)cols * n);
    std::vector<uint32_t> mt((size_t)n * MT_STRIDE, 0u);
    const size_t rec = cge_traffic_state_bytes(h);
    std::vector<uint32_t> raw((size_t)cols * 4);
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
        h->ops.from_record(hd, total_reward, w, raw.data());
        for (int c = 0; c < cols; ++c) st[(size_t)c * n + i] = make_uint4(raw[4 * c], raw[4 * c + 1], raw[4 * c + 2], raw[4 * c + 3]);
        memcpy(&mt[(size_t)i * MT_STRIDE], w + 16 * ni, MT_N * 4);
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
