// parking.hip — batched SmartParkingEnv for MI355X (gfx950): kernels + C ABI (include/cge_amd.h).
//
// Re-expresses /root/reference/smart_parking_env/core/ for N independent instances, one lane per env:
//   parking_env.py  reset :70-108, step :110-159, _process_action :161-195, _reject_customer :197-216,
//                   _assign_customer_to_zone :218-269, _toggle_zone_price :271-304, _get_observation :306-369
//   customer.py     calculate_satisfaction :84-124, get_duration_discount :126-138, generate_customer :146-184,
//                   get_adjusted_zone_preferences :187-220, get_time_based_duration_type :223-240,
//                   should_customer_arrive :243-254, CustomerManager counters :257-353
//   parking_lot.py  add_to_queue :193-208, get_next_queued_customer :210-227, update_time_minute :229-252
//   pricing.py      _update_prices :59-64, calculate_revenue :96-118, _get_duration_multiplier :120-136
// with every quirk of the reference kept (timestep passed where an hour is expected, duration discount
// applied twice, rejected customers counted as satisfied, arrivals at a full queue dropped but counted,
// failed assignments re-queued at the back).
//
// State per env: 69 dwords in 18 uint4 columns (SoA) — 50 spots (occupied:1 duration:5 arrival:11 and the
// 9-bit code from which the customer's satisfaction is recomputed exactly at departure), a 10-slot FIFO
// (pref:2 duration:5 wait:11), price levels, counters, float64 revenue / satisfaction sums, the MT19937
// cursor.  Everything lives in VGPRs; spot and queue indexing use static unrolled scans.  Prices,
// satisfaction and reward are float64 in the reference's operation order -> obs (float32) and reward are
// bit-identical to the CPU.  Draws per step are data dependent (arrival test every step; zone, duration type,
// randint with rejection, extension on an arrival): LdsDraws window parked in LDS.
#include <cstring>
#include <vector>

#include "cge_device.hpp"
#include "cge_host.hpp"

namespace cge {
namespace parking {

constexpr int NSPOT = 50;
constexpr int QMAX = 10;
constexpr int OBS = 13;
constexpr int COLS = 18;
constexpr int DW = 16;
constexpr int DROW = 17;
constexpr int BLOCK = 64;

struct Params {
    uint4 *state;
    uint32_t *mt;
    int64_t n, env0;
    int32_t mode, max_steps;
    const int32_t *actions;
    const uint8_t *mask;
    float *obs, *final_obs, *reward;
    FinalSeg fin;          // fused rollouts (SAME_STEP): terminal rows compacted per wave (cge_parking_rollout_final_obs); rows nullable
    uint8_t *terminated, *truncated;
    int32_t k_steps;
    uint64_t a_seed;
    int64_t t0, obs_step_stride;
    double *reward_sum;
    int32_t *done_count;
    double *ep_ret;       // episode statistics (cge_parking_episode_stats), nullable
    int32_t *ep_len;
};

__host__ __device__ __forceinline__ constexpr int zone_of(int k) { return k < 15 ? 0 : k < 35 ? 1 : 2; }
__device__ __forceinline__ double base_price(uint32_t z) { return z == 0 ? 8.0 : z == 1 ? 5.0 : 3.0; }        // config.py:7-9
__device__ __forceinline__ double level_mult(uint32_t l) { return l == 0 ? 0.7 : l == 1 ? 1.0 : 1.3; }        // config.py:81-85

struct Env {
    uint32_t spot[NSPOT];   // occupied:1 | duration:5 << 1 | arrival:16 << 6 | satcode:9 << 22  (pref:2 ratio:2 wait18:5)
    uint32_t q[QMAX];       // pref:2 | duration:5 << 2 | wait:16 << 7
    uint32_t t, qlen, lv0, lv1, lv2, changes, needs_reset, mt_pos, mt_pretw;
    int32_t last_change;
    uint32_t total_customers, rejected, satisfied, episodes, total_wait;
    double revenue, satisfaction_sum;
    double ep_return;       // sum of the running episode's rewards (float64, step order)

    __host__ __device__ __forceinline__ void unpack(const uint32_t *raw) {
#pragma unroll
        for (int k = 0; k < NSPOT; ++k) spot[k] = raw[k];
#pragma unroll
        for (int k = 0; k < QMAX; ++k) q[k] = raw[50 + k];
        const uint32_t m0 = raw[60], m1 = raw[61], m2 = raw[62], m3 = raw[63];
        t = m0 & 0xFFFFu; qlen = (m0 >> 16) & 15u; lv0 = (m0 >> 20) & 3u; lv1 = (m0 >> 22) & 3u; lv2 = (m0 >> 24) & 3u;
        changes = (m0 >> 26) & 3u; needs_reset = (m0 >> 28) & 1u;
        last_change = (int32_t)(m1 & 0xFFFFu) - 1000; mt_pos = (m1 >> 16) & 1023u; mt_pretw = mt_ready_decode((m1 >> 26) & 31u);   // ready mark of the twist-ahead stream
        total_customers = m2 & 0xFFFFu; rejected = m2 >> 16; satisfied = m3 & 0xFFFFu; episodes = m3 >> 16;
        total_wait = raw[64];
        uint64_t u = ((uint64_t)raw[66] << 32) | raw[65];
        memcpy(&revenue, &u, 8);
        u = ((uint64_t)raw[68] << 32) | raw[67];
        memcpy(&satisfaction_sum, &u, 8);
        u = ((uint64_t)raw[70] << 32) | raw[69];
        memcpy(&ep_return, &u, 8);
    }
    __host__ __device__ __forceinline__ void pack(uint32_t *raw) const {
#pragma unroll
        for (int k = 0; k < NSPOT; ++k) raw[k] = spot[k];
#pragma unroll
        for (int k = 0; k < QMAX; ++k) raw[50 + k] = q[k];
        raw[60] = (t & 0xFFFFu) | (qlen << 16) | (lv0 << 20) | (lv1 << 22) | (lv2 << 24) | (changes << 26) | (needs_reset << 28);
        raw[61] = ((uint32_t)(last_change + 1000) & 0xFFFFu) | (mt_pos << 16) | ((mt_pretw > mt_pos ? mt_ready_encode(mt_pretw) : 0u) << 26);
        raw[62] = (total_customers & 0xFFFFu) | (rejected << 16);
        raw[63] = (satisfied & 0xFFFFu) | (episodes << 16);
        raw[64] = total_wait;
        uint64_t u;
        memcpy(&u, &revenue, 8);
        raw[65] = (uint32_t)u; raw[66] = (uint32_t)(u >> 32);
        memcpy(&u, &satisfaction_sum, 8);
        raw[67] = (uint32_t)u; raw[68] = (uint32_t)(u >> 32);
        memcpy(&u, &ep_return, 8);
        raw[69] = (uint32_t)u; raw[70] = (uint32_t)(u >> 32);
        raw[71] = 0;
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
    __device__ __forceinline__ void reset() {                      // parking_env.py:70-108 (no draws)
#pragma unroll
        for (int k = 0; k < NSPOT; ++k) spot[k] = 0;
#pragma unroll
        for (int k = 0; k < QMAX; ++k) q[k] = 0;
        t = 0; qlen = 0; lv0 = lv1 = lv2 = 1; changes = 0; last_change = -999; needs_reset = 0;
        total_customers = rejected = satisfied = total_wait = 0;
        revenue = 0.0; satisfaction_sum = 0.0; ep_return = 0.0;
    }
    __device__ __forceinline__ double zone_price(uint32_t z) const {   // pricing.py:59-64
        // mask form: a ternary over array elements is folded into "select the address, then load" -> scratch
        const uint32_t l = (lv0 & (0u - (uint32_t)(z == 0))) | (lv1 & (0u - (uint32_t)(z == 1))) | (lv2 & (0u - (uint32_t)(z == 2)));
        return base_price(z) * level_mult(l);
    }
};

// customer.calculate_satisfaction :84-124 from its three discrete inputs (exactly the reference's float64 product)
__device__ __forceinline__ double satisfaction_of(uint32_t prefclass, uint32_t ratioclass, uint32_t wait18) {
    double s = 1.0;
    s *= prefclass == 0 ? 0.9 : prefclass == 1 ? 1.0 : 0.6;            // flexible / match / mismatch
    s *= ratioclass == 0 ? 1.1 : ratioclass == 1 ? 1.0 : ratioclass == 2 ? 0.8 : 0.5;
    if (wait18 > 0) {
        double pen = (double)wait18 / 60.0;
        pen = pen < 0.3 ? pen : 0.3;
        s *= (1.0 - pen);
    }
    s = s < 1.0 ? s : 1.0;
    return s > 0.0 ? s : 0.0;
}

// generate_customer :146-184 — `hour` is the TIMESTEP (customer.py:284)
__device__ __forceinline__ uint32_t generate_customer(uint32_t hour, LdsDraws<DW> &d) {
    double p0 = 0.3, p1 = 0.35, p2 = 0.15, p3 = 0.2;                   // config.py:71-76
    if (6 <= hour && hour <= 9) { p0 *= 1.3; p1 *= 1.1; p2 *= 0.8; p3 *= 0.9; }
    else if (17 <= hour && hour <= 19) { p0 *= 0.9; p1 *= 1.2; p2 *= 1.1; p3 *= 1.0; }
    else if (22 <= hour || hour <= 5) { p0 *= 0.7; p1 *= 0.8; p2 *= 1.0; p3 *= 1.5; }
    double total = 0.0 + p0;
    total += p1; total += p2; total += p3;
    const double rnd = d.random53();
    double cum = 0.0;
    uint32_t pref = 3;
    cum += p0 / total;
    if (rnd <= cum) pref = 0;
    else {
        cum += p1 / total;
        if (rnd <= cum) pref = 1;
        else {
            cum += p2 / total;
            if (rnd <= cum) pref = 2;   // else flexible (the 4th comparison cannot change the default)
        }
    }
    uint32_t type;                                                      // 0 short 1 medium 2 long, customer.py:223-240
    if (6 <= hour && hour <= 9) type = d.random53() < 0.6 ? 0u : 1u;
    else if (12 <= hour && hour <= 14) type = d.randbelow(3u, 2);
    else if (17 <= hour && hour <= 19) type = d.random53() < 0.6 ? 1u : 2u;
    else type = d.random53() < 0.5 ? 2u : 1u;
    uint32_t dur;                                                       // randint(min_hours, max_hours), config.py:50-54
    if (type == 0) dur = 1u + d.randbelow(2u, 2);
    else if (type == 1) dur = 3u + d.randbelow(3u, 2);
    else dur = 6u + d.randbelow(7u, 3);
    if (d.random53() < 0.3) dur += 1u + d.randbelow(3u, 2);
    dur = dur < 24u ? dur : 24u;
    return pref | (dur << 2);                                           // wait = 0
}

__device__ __forceinline__ bool env_step(Env &e, int32_t max_steps, int32_t a, LdsDraws<DW> &d, double &reward) {   // :110-159
    if (e.t > 0 && e.t % 60u == 0) e.changes = 0;
    d.ensure(10);                                          // wave-convergent top-up (see LdsDraws::ensure)
    const uint32_t hour = e.t / 60u;
    static constexpr int RATE[24] = {3, 2, 2, 2, 2, 4, 8, 12, 15, 10, 8, 8, 12, 12, 8, 8, 10, 15, 18, 15, 12, 8, 6, 4};   // config.py:19-47
    double prob = 0.05;
#pragma unroll
    for (int hh = 0; hh < 24; ++hh) prob = hour == (uint32_t)hh ? (double)RATE[hh] / 60 : prob;
    if (d.random53() < prob) {                                          // should_customer_arrive
        const uint32_t c = generate_customer(e.t, d);
        e.total_customers += 1;
        if (e.qlen < (uint32_t)QMAX) {                                  // else dropped silently
#pragma unroll
            for (int k = 0; k < QMAX; ++k) e.q[k] = e.qlen == (uint32_t)k ? c : e.q[k];
            e.qlen += 1;
        }
    }
#pragma unroll
    for (int k = 0; k < QMAX; ++k)                                      // wait times :133-134
        if ((uint32_t)k < e.qlen) { const uint32_t w = e.q[k] >> 7; e.q[k] = (e.q[k] & 127u) | ((w < 0xFFFFu ? w + 1 : w) << 7); }
    e.total_wait += e.qlen;
    double r = 0.0;
    if (a >= 1 && a <= 4) {
        if (e.qlen > 0) {
            const uint32_t c = e.q[0];                                  // pop(0)
#pragma unroll
            for (int k = 0; k + 1 < QMAX; ++k) e.q[k] = e.q[k + 1];
            e.q[QMAX - 1] = 0;
            e.qlen -= 1;
            if (a == 4) {                                               // _reject_customer :197-216
                e.rejected += 1;
                e.satisfied += 1;                                       // default satisfaction 1.0 > 0.7
                uint32_t occ = 0;
#pragma unroll
                for (int k = 0; k < NSPOT; ++k) occ += e.spot[k] & 1u;
                r += occ < (uint32_t)NSPOT ? -2.0 : 0.0;
            } else {                                                    // _assign_customer_to_zone :218-269
                const uint32_t z = (uint32_t)a - 1u;
                int free_k = -1;
#pragma unroll
                for (int k = NSPOT - 1; k >= 0; --k)
                    if ((uint32_t)zone_of(k) == z && !(e.spot[k] & 1u)) free_k = k;
                if (free_k < 0) {                                       // back of the queue
#pragma unroll
                    for (int k = 0; k < QMAX; ++k) e.q[k] = e.qlen == (uint32_t)k ? c : e.q[k];
                    e.qlen += 1;
                } else {
                    const uint32_t pref = c & 3u, dur = (c >> 2) & 31u, wait = c >> 7;
                    const double disc = dur >= 6 ? 0.75 : dur >= 3 ? 0.85 : 1.0;
                    const double base = e.zone_price(z);
                    const double hourly = base * disc * disc;           // duration multiplier, then the same customer discount
                    const double total_price = hourly * (double)dur;
                    const double ratio = total_price / base;
                    const uint32_t prefclass = pref == 3 ? 0u : pref == z ? 1u : 2u;
                    const uint32_t ratioclass = ratio <= 0.8 ? 0u : ratio <= 1.0 ? 1u : ratio <= 1.2 ? 2u : 3u;
                    const uint32_t wait18 = wait < 18u ? wait : 18u;    // wait/60 >= 0.3 from 18 minutes on
                    const double s = satisfaction_of(prefclass, ratioclass, wait18);
                    const uint32_t nv = 1u | (dur << 1) | ((e.t & 0xFFFFu) << 6) | ((prefclass | (ratioclass << 2) | (wait18 << 4)) << 22);
#pragma unroll
                    for (int k = 0; k < NSPOT; ++k) e.spot[k] = free_k == k ? nv : e.spot[k];
                    e.revenue += total_price;
                    double rr = total_price * 1.0;
                    rr += s * 0.8;
                    r += rr;
                }
            }
        }
    } else if (a >= 5 && a <= 7) {                                      // _toggle_zone_price :271-304
        if (e.changes >= 2) r += -0.5 * 2;
        else if ((int32_t)e.t - e.last_change < 15) r += -0.5 * 2;
        else {
            const int z = a - 5;
            const uint32_t cur = (e.lv0 & (0u - (uint32_t)(z == 0))) | (e.lv1 & (0u - (uint32_t)(z == 1))) | (e.lv2 & (0u - (uint32_t)(z == 2)));
            const uint32_t nl = cur == 2u ? 0u : cur + 1u;
            e.lv0 = z == 0 ? nl : e.lv0; e.lv1 = z == 1 ? nl : e.lv1; e.lv2 = z == 2 ? nl : e.lv2;
            e.changes += 1;
            e.last_change = (int32_t)e.t;
            r += -0.5;
        }
    }
#pragma unroll
    for (int k = 0; k < NSPOT; ++k) {                                   // update_time_minute, parking_lot.py:229-252
        const uint32_t sp = e.spot[k];
        const uint32_t dur = (sp >> 1) & 31u, arr = (sp >> 6) & 0xFFFFu;
        if ((sp & 1u) && (e.t - arr) >= 60u * dur) {                    // (t-arr)/60.0 >= dur, exact in integers
            const uint32_t code = sp >> 22;
            const double s = satisfaction_of(code & 3u, (code >> 2) & 3u, code >> 4);
            e.satisfaction_sum += s;
            e.satisfied += s > 0.7 ? 1u : 0u;
            e.spot[k] = 0;
        }
    }
    e.t += 1;
    reward = r;
    return e.t >= (uint32_t)max_steps;
}

// _get_observation :306-369 -> 13 float32 staged [64][13] in LDS, then coalesced
__device__ __forceinline__ void observe(const Env &e, const RowMap &rm, float *__restrict__ dst, uint32_t *__restrict__ tile) {
    const uint32_t lane = threadIdx.x & 63u;
    double v[OBS];
    uint32_t occ[3] = {0, 0, 0};
#pragma unroll
    for (int k = 0; k < NSPOT; ++k) occ[zone_of(k)] += e.spot[k] & 1u;
    v[0] = (double)occ[0] / 15.0; v[1] = (double)occ[1] / 20.0; v[2] = (double)occ[2] / 15.0;
#pragma unroll
    for (int z = 0; z < 3; ++z) { const double p = e.zone_price((uint32_t)z) / (8.0 * 1.3); v[3 + z] = p < 1.0 ? p : 1.0; }
    { const double h = (double)(e.t / 60u) / 24.0; v[6] = h < 1.0 ? h : 1.0; }
    v[7] = (double)(e.t % 60u) / 60.0;
    { const double ql = (double)e.qlen / 10.0; v[8] = ql < 1.0 ? ql : 1.0; }
    { const double fw = (double)(e.qlen ? e.q[0] >> 7 : 0u) / 60.0; v[9] = fw < 1.0 ? fw : 1.0; }
    uint32_t tw = 0;
#pragma unroll
    for (int k = 0; k < QMAX; ++k) tw += (uint32_t)k < e.qlen ? e.q[k] >> 7 : 0u;
    { const double x = (double)tw / 300.0; v[10] = x < 1.0 ? x : 1.0; }
    { const double pc = (double)e.changes / 2.0; v[11] = pc < 1.0 ? pc : 1.0; }
    { const double sc = (double)((int32_t)e.t - e.last_change) / 60.0; v[12] = sc < 1.0 ? sc : 1.0; }
    float *row = reinterpret_cast<float *>(tile) + lane * OBS;
#pragma unroll
    for (int j = 0; j < OBS; ++j) { const float f = (float)v[j]; row[j] = f < 0.0f ? 0.0f : (f > 1.0f ? 1.0f : f); }   // np.clip in float32
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    uint32_t r = lane / (uint32_t)OBS, col = lane - r * (uint32_t)OBS;
#pragma unroll 1
    for (int m = 0; m < OBS; ++m) {
        int64_t to;
        if (rm.row(r, to)) reinterpret_cast<uint32_t *>(dst)[to * OBS + col] = tile[r * OBS + col];
        col += 64u % OBS; r += 64u / OBS;
        if (col >= (uint32_t)OBS) { col -= OBS; r += 1u; }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

template <bool ROLLOUT>
__global__ __launch_bounds__(BLOCK, 2) void step_kernel(Params p) {   // two waves per SIMD: 131,072 envs are 2,048 waves = two per SIMD, all resident at once (left alone the allocator took 258 registers in round 4: one wave per SIMD, two rounds)
    __shared__ uint32_t tile[64 * OBS];
    __shared__ uint32_t draws[64 * DROW];
    const int64_t i0 = (int64_t)blockIdx.x * BLOCK;
    const int64_t i = i0 + threadIdx.x;
    const bool live = i < p.n;
    const int64_t li = live ? i : i0;
    const int64_t nrows = p.n - i0 < 64 ? p.n - i0 : 64;
    Env e;
    e.load(p.state, p.n, li);
    LdsDraws<DW> d(draws + (threadIdx.x & 63u) * DROW, p.mt + li * MT_STRIDE, e.mt_pos, e.mt_pretw);
    const uint64_t key = ROLLOUT ? hash_env_key(p.a_seed, (uint64_t)(p.env0 + li)) : 0;
    double rsum = 0.0;
    int32_t dcount = 0;
    uint32_t fin_used = 0;                                         // terminal rows this wave has delivered to its segment (fused rollouts)
    const int ksteps = ROLLOUT ? p.k_steps : 1;
#pragma unroll 1
    for (int t = 0; t < ksteps; ++t) {
        double reward = 0.0;
        bool term = false, reset_now = false;
        if (live) {
            if (p.mode == CGE_AUTORESET_NEXT_STEP && e.needs_reset) {
                reset_now = true;
            } else {
                const int32_t a = p.actions ? p.actions[(int64_t)t * p.n + i] : (int32_t)hash_action_from_key(key, (uint64_t)(p.t0 + t), 8u, 0u);
                d.ensure_ahead(10, true);                            // the arrival test draws every step; the window outlives the step
                term = env_step(e, p.max_steps, a, d, reward);
                e.ep_return += reward;
                if (term) {
                    e.episodes += 1;
                    if (p.ep_ret) p.ep_ret[i] = e.ep_return;
                    if (p.ep_len) p.ep_len[i] = (int32_t)e.t;
                    if (p.mode == CGE_AUTORESET_SAME_STEP) reset_now = true;
                    else if (p.mode == CGE_AUTORESET_NEXT_STEP) e.needs_reset = 1;
                }
            }
        }
        const bool fin = live && term && reset_now;
        const unsigned long long fin_mask = __ballot(fin);
        if (fin_mask) {                                            // terminal rows: step() -> final_obs_out; fused rollout -> the wave's segment
            if (!ROLLOUT) {
                if (p.final_obs) observe(e, RowMap{fin_mask, nrows, 0, false}, p.final_obs + i0 * OBS, tile);
            } else if (p.fin.rows) {
                float *fdst;
                const RowMap rm = final_rows<float>(p.fin, (int64_t)blockIdx.x, fin_used, fin, fin_mask, nrows, t, i, OBS, fdst);
                observe(e, rm, fdst, tile);
            }
            fin_used += (uint32_t)__popcll(fin_mask);
        }
        if (reset_now) e.reset();
        if (p.obs) observe(e, RowMap{~0ull, nrows, 0, false}, p.obs + (int64_t)t * p.obs_step_stride + i0 * OBS, tile);
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
        d.flush();                                                // advances the cursor; the consumed words were ready: nothing is stored
        e.mt_pos = d.pos; e.mt_pretw = d.pretw;
        e.store(p.state, p.n, i);
        if (ROLLOUT) {
            if (p.reward_sum) p.reward_sum[i] = rsum;
            if (p.fin.count && threadIdx.x == 0) p.fin.count[blockIdx.x] = (int32_t)fin_used;
            if (p.done_count) p.done_count[i] = dcount;
        }
    }
}

// reset (mask) / initial state (init != 0: also rewinds the RNG cursor) + obs
__global__ __launch_bounds__(BLOCK) void reset_kernel(Params p, int init, int rewind) {
    __shared__ uint32_t tile[64 * OBS];
    const int64_t i0 = (int64_t)blockIdx.x * BLOCK;
    const int64_t i = i0 + threadIdx.x;
    const bool live = i < p.n;
    const int64_t nrows = p.n - i0 < 64 ? p.n - i0 : 64;
    Env e;
    e.load(p.state, p.n, live ? i : i0);
    if (live) {
        bool dirty = false;
        if (init) { e.reset(); e.episodes = 0; e.mt_pos = 0; e.mt_pretw = 0; dirty = true; }
        else if (!rewind && (!p.mask || p.mask[i])) { e.reset(); dirty = true; }
        if (rewind) { e.mt_pos = 0; e.mt_pretw = 0; dirty = true; }
        if (dirty) e.store(p.state, p.n, i);
    }
    if (p.obs) observe(e, RowMap{~0ull, nrows, 0, false}, p.obs + i0 * OBS, tile);
}

__global__ __launch_bounds__(256) void info_kernel(const uint4 *__restrict__ state, int64_t n, int field, int idx, int32_t *__restrict__ out,
                                                   double *__restrict__ out64) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    Env e;
    e.load(state, n, i);
    if (out64) { out64[i] = field == CGE_PARKING_INFO64_EPISODE_REVENUE ? e.revenue : e.satisfaction_sum; return; }
    int32_t v = 0;
    uint32_t occ[3] = {0, 0, 0};
#pragma unroll
    for (int k = 0; k < NSPOT; ++k) occ[zone_of(k)] += e.spot[k] & 1u;
    switch (field) {
        case CGE_PARKING_INFO_TIMESTEP: v = (int32_t)e.t; break;
        case CGE_PARKING_INFO_TOTAL_CUSTOMERS: v = (int32_t)e.total_customers; break;
        case CGE_PARKING_INFO_REJECTED: v = (int32_t)e.rejected; break;
        case CGE_PARKING_INFO_SATISFIED: v = (int32_t)e.satisfied; break;
        case CGE_PARKING_INFO_TOTAL_WAIT_TIME: v = (int32_t)e.total_wait; break;
        case CGE_PARKING_INFO_QUEUE_LENGTH: v = (int32_t)e.qlen; break;
        case CGE_PARKING_INFO_PRICE_CHANGES_THIS_HOUR: v = (int32_t)e.changes; break;
        case CGE_PARKING_INFO_ZONE_OCCUPIED: v = (int32_t)(idx == 0 ? occ[0] : idx == 1 ? occ[1] : occ[2]); break;
        case CGE_PARKING_INFO_PRICE_LEVEL: v = (int32_t)((e.lv0 & (0u - (uint32_t)(idx == 0))) | (e.lv1 & (0u - (uint32_t)(idx == 1))) | (e.lv2 & (0u - (uint32_t)(idx == 2)))); break;
        case CGE_PARKING_INFO_EPISODES: v = (int32_t)e.episodes; break;
        case CGE_PARKING_INFO_NEEDS_RESET: v = (int32_t)e.needs_reset; break;
    }
    out[i] = v;
}

}  // namespace parking
}  // namespace cge

using namespace cge;

struct cge_parking : HandleBase {
    cge_parking_config cfg{};
    uint4 *state = nullptr;
    uint32_t *mt = nullptr;
    static constexpr uint32_t snap_tag = 1u;
    std::vector<std::pair<void *, size_t>> blobs() const { return {{state, (size_t)parking::COLS * n * sizeof(uint4)}, {mt, (size_t)n * MT_STRIDE * 4}}; }
    uint32_t snap_extra() const { return 0u; }
    void set_snap_extra(uint32_t v) { (void)v; }
    parking::Params params() const {
        parking::Params p{};
        p.state = state; p.mt = mt; p.n = n; p.env0 = env0; p.mode = cfg.autoreset_mode; p.max_steps = cfg.max_steps;
        p.ep_ret = ep_ret; p.ep_len = ep_len;
        return p;
    }
    unsigned blocks() const { return (unsigned)((n + parking::BLOCK - 1) / parking::BLOCK); }
};

extern "C" {

int cge_parking_create(const cge_parking_config *cfg, int64_t n_envs, int device, int64_t env_index0, cge_parking **out) {
    if (!cfg || !out || n_envs <= 0 || env_index0 < 0) return CGE_ERR_INVALID_ARG;
    *out = nullptr;
    if (cfg->autoreset_mode < 0 || cfg->autoreset_mode > 2 || cfg->max_steps < 0 || cfg->max_steps > 60000) return CGE_ERR_INVALID_ARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return CGE_ERR_NO_DEVICE;
    cge_parking *h = new cge_parking();
    h->cfg = *cfg;
    if (h->cfg.max_steps == 0) h->cfg.max_steps = 1440;
    h->n = n_envs; h->env0 = env_index0; h->device = device;
    DeviceGuard g(device);
    const size_t sb = (size_t)parking::COLS * n_envs * sizeof(uint4), mb = (size_t)n_envs * MT_STRIDE * sizeof(uint32_t);
    hipError_t e;
    if ((e = hipMalloc(&h->state, sb)) != hipSuccess || (e = hipMalloc(&h->mt, mb)) != hipSuccess || (e = hipMemset(h->state, 0, sb)) != hipSuccess) {
        (void)hipFree(h->state); (void)hipFree(h->mt);
        delete h;
        return CGE_ERR_HIP;
    }
    h->device_bytes = sb + mb;
    e = launch_mt_seed(h->mt, MT_STRIDE, n_envs, nullptr, 0, env_index0, 0, nullptr);
    if (e == hipSuccess) {
        parking::Params p = h->params();
        hipLaunchKernelGGL(parking::reset_kernel, dim3(h->blocks()), dim3(parking::BLOCK), 0, nullptr, p, 1, 0);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
    if (e != hipSuccess) {
        (void)hipFree(h->state); (void)hipFree(h->mt);
        delete h;
        return CGE_ERR_HIP;
    }
    *out = h;
    return CGE_OK;
}

int cge_parking_destroy(cge_parking *h) {
    if (!h) return CGE_ERR_INVALID_ARG;
    DeviceGuard g(h->device);
    (void)hipDeviceSynchronize();
    (void)hipFree(h->state); (void)hipFree(h->mt);
    delete h;
    return CGE_OK;
}

int cge_parking_seed(cge_parking *h, const uint64_t *seeds, uint64_t base_seed, void *stream) {
    if (!h) return CGE_ERR_INVALID_ARG;
    DeviceGuard g(h->device);
    CGE_TRY(h, launch_mt_seed(h->mt, MT_STRIDE, h->n, seeds, base_seed, h->env0, 0, as_stream(stream)));
    parking::Params p = h->params();
    hipLaunchKernelGGL(parking::reset_kernel, dim3(h->blocks()), dim3(parking::BLOCK), 0, as_stream(stream), p, 0, 1);   // rewind cursors
    CGE_TRY(h, hipGetLastError());
    return CGE_OK;
}

int cge_parking_reset(cge_parking *h, const uint8_t *mask, float *obs_out, void *stream) {
    if (!h) return CGE_ERR_INVALID_ARG;
    DeviceGuard g(h->device);
    parking::Params p = h->params();
    p.mask = mask; p.obs = obs_out;
    hipLaunchKernelGGL(parking::reset_kernel, dim3(h->blocks()), dim3(parking::BLOCK), 0, as_stream(stream), p, 0, 0);
    CGE_TRY(h, hipGetLastError());
    return CGE_OK;
}

int cge_parking_step(cge_parking *h, const int32_t *actions, float *obs_out, float *reward_out, uint8_t *terminated_out,
                     uint8_t *truncated_out, float *final_obs_out, void *stream) {
    if (!h) return CGE_ERR_INVALID_ARG;
    if (!actions || !obs_out || !reward_out || !terminated_out)
        return h->fail(CGE_ERR_INVALID_ARG, "cge_parking_step: null actions/obs/reward/terminated pointer");
    DeviceGuard g(h->device);
    parking::Params p = h->params();
    p.actions = actions; p.obs = obs_out; p.reward = reward_out; p.terminated = terminated_out; p.truncated = truncated_out;
    p.final_obs = final_obs_out; p.k_steps = 1;
    hipLaunchKernelGGL(parking::step_kernel<false>, dim3(h->blocks()), dim3(parking::BLOCK), 0, as_stream(stream), p);
    h->last_kernel = "cge::parking::step_kernel<false>";
    CGE_TRY(h, hipGetLastError());
    return CGE_OK;
}

int cge_parking_rollout(cge_parking *h, int32_t k_steps, const int32_t *actions, uint64_t action_seed, int64_t t0, float *obs_out,
                        int64_t obs_step_stride, float *reward_traj_out, uint8_t *terminated_traj_out, double *reward_sum_out,
                        int32_t *done_count_out, void *stream) {
    if (!h) return CGE_ERR_INVALID_ARG;
    if (k_steps < 0 || obs_step_stride < 0 || (obs_step_stride != 0 && obs_step_stride < h->n * parking::OBS))
        return h->fail(CGE_ERR_INVALID_ARG, "cge_parking_rollout: bad k_steps / obs_step_stride");
    if (k_steps == 0) return CGE_OK;
    DeviceGuard g(h->device);
    parking::Params p = h->params();
    p.k_steps = k_steps; p.actions = actions; p.a_seed = action_seed; p.t0 = t0; p.obs = obs_out; p.obs_step_stride = obs_step_stride;
    p.reward = reward_traj_out; p.terminated = terminated_traj_out; p.reward_sum = reward_sum_out; p.done_count = done_count_out;
    p.fin = FinalSeg{h->fin_rows, h->fin_index, h->fin_count, h->fin_cap, h->n};
    hipLaunchKernelGGL(parking::step_kernel<true>, dim3(h->blocks()), dim3(parking::BLOCK), 0, as_stream(stream), p);
    h->last_kernel = "cge::parking::step_kernel<true>";
    CGE_TRY(h, hipGetLastError());
    return CGE_OK;
}

CGE_DEFINE_FINAL_OBS(parking, float, 64)

int cge_parking_info(cge_parking *h, int32_t field_id, int32_t index, int32_t *out, void *stream) {
    if (!h) return CGE_ERR_INVALID_ARG;
    if (!out || field_id < 0 || field_id > CGE_PARKING_INFO_NEEDS_RESET || index < 0 || index > 2)
        return h->fail(CGE_ERR_INVALID_ARG, "cge_parking_info: bad field / index / null out");
    DeviceGuard g(h->device);
    hipLaunchKernelGGL(parking::info_kernel, dim3((unsigned)((h->n + 255) / 256)), dim3(256), 0, as_stream(stream), h->state, h->n, field_id,
                       index, out, (double *)nullptr);
    CGE_TRY(h, hipGetLastError());
    return CGE_OK;
}

int cge_parking_info64(cge_parking *h, int32_t field_id, double *out, void *stream) {
    if (!h || !out || field_id < 0 || field_id > 1) return CGE_ERR_INVALID_ARG;
    DeviceGuard g(h->device);
    hipLaunchKernelGGL(parking::info_kernel, dim3((unsigned)((h->n + 255) / 256)), dim3(256), 0, as_stream(stream), h->state, h->n, field_id, 0,
                       (int32_t *)nullptr, out);
    CGE_TRY(h, hipGetLastError());
    return CGE_OK;
}

size_t cge_parking_snapshot_bytes(const cge_parking *h) { return h ? snapshot_bytes(h) : 0; }
int cge_parking_snapshot_get(cge_parking *h, void *host_buf, void *stream) { return snapshot_get(h, host_buf, as_stream(stream)); }
int cge_parking_snapshot_set(cge_parking *h, const void *host_buf, void *stream) { return snapshot_set(h, host_buf, as_stream(stream)); }
size_t cge_parking_device_bytes(const cge_parking *h) { return h ? h->device_bytes : 0; }
int cge_parking_episode_stats(cge_parking *h, double *return_out, int32_t *length_out) {
    if (!h) return CGE_ERR_INVALID_ARG;
    h->ep_ret = return_out; h->ep_len = length_out;
    return CGE_OK;
}

const char *cge_parking_last_error(const cge_parking *h) { return h ? h->last_error.c_str() : "null handle"; }

const char *cge_parking_last_kernel(const cge_parking *h) { return h ? h->last_kernel.c_str() : ""; }

}  // extern "C"
