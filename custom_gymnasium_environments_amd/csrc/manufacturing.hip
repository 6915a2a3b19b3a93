// manufacturing.hip — batched SmartManufacturingEnv for MI355X (gfx950): kernels + C ABI (include/cge_amd.h).
//
// Re-expresses /root/reference/smart_manufacturing_env/manufacturing_env.py for N independent instances, one lane per env:
//   reset :113-192, _get_observation :194-250, step :252-301, _process_action :303-359, _start_production :361-379,
//   _update_production :381-425, _update_machine_status :427-462, _quality_control :464-480, _complete_product :482-500
//   (its return value is discarded at :418), _calculate_timestep_rewards :502-531, _update_metrics :533-553,
//   _check_termination :555-578, _update_supply_chain :580-595.
// Fixed state per env: 88 dwords in 22 uint4 columns (PCG64, 5 stations incl. a cached copy of the product each is
// working on, counters, thresholds, cached recent-quality means).  Variable state lives in per-env TABLES:
//   pq/pm/pnext/pts[320][N]  one row per product started in the episode (row = product id; an episode can start at most
//                        (250 + 29*99)/10 = 312), laid out [row][env]: quality f64 as of the product's last station exit,
//                        {type, station+1, alive, remaining} u16, queue link u16, the product's place in its type list u16
//   tq/tid[6][320][N]    per product type, the DENSE list of the qualities (f64) and ids (u16) of the products of that type
//                        in the system, in list order, occupying places [head, head + count) — laid out [place][env] like the
//                        other tables: lanes whose lists stand at the same places read one line (per-env contiguous lists,
//                        64 scattered 64-byte runs per load, cost 1.3-1.8 us of latency per run, round 3)
//   comp[20][N]          ring of the last 20 completed qualities (:525, :552)
//   hist[100][N]         ring of the last 100 quality_rate_history entries (:573-576)
// The per-type quality means of the observation (:220-228) are np.mean over ALL products in the system in list order,
// i.e. NumPy's pairwise summation, and the qualities of the products at the stations change every step: the means are
// recomputed every step (bit-identical to the reference), which is why each type's qualities are kept dense — the sum is
// then 8 register accumulators over runs of 8 places, the next run loaded while the last is added.
// A product leaves its list from (or near) the front: same-type products overtake nobody (FIFO queues, equal station times),
// only products that were started while station 0 was down stay behind forever; the few older entries move up one place.
// Rewards are integers -> exact.
#include <cstring>
#include <vector>

#include "cge_device.hpp"
#include "cge_host.hpp"
#include "cge_pcg.hpp"

namespace cge {
namespace mfg {

constexpr int OBS = 73;
constexpr int BLOCK = 64;
constexpr int COLS = 23;
constexpr int CAP = 320;
constexpr uint32_t NONE = 1023u;
constexpr int TROW = 6 * CAP;        // places per env in tq / tid
constexpr uint32_t M_ALIVE = 1u << 6;
enum : uint32_t { OPERATIONAL = 0, BROKEN = 1, MAINTENANCE = 2 };
enum : uint32_t { BALANCED = 0, RUSH = 1, QUALITY = 2 };

struct Params {
    uint4 *state;
    double *pq;
    uint16_t *pm, *pnext, *pts;
    double *tq;
    uint16_t *tid;
    double *comp, *hist;
    int64_t n, env0;
    int32_t mode, max_steps, k_steps;
    const int32_t *actions;
    uint64_t a_seed;
    int64_t t0;
    float *obs;
    int64_t obs_step_stride;
    float *reward;
    uint8_t *terminated, *truncated;
    float *final_obs;
    FinalSeg fin;          // fused rollouts (SAME_STEP): terminal rows compacted per wave (cge_manufacturing_rollout_final_obs); rows nullable
    const uint8_t *mask;
    const uint64_t *seeds;
    uint64_t base_seed;
    double *reward_sum;
    int32_t *done_count;
    double *ep_ret;       // episode statistics (cge_manufacturing_episode_stats), nullable
    int32_t *ep_len;
    uint8_t *done;        // step(): terminated | truncated (cge_manufacturing_done_mask), nullable
};

__device__ __forceinline__ double mk_double(uint32_t lo, uint32_t hi) { return __hiloint2double((int)hi, (int)lo); }
__device__ __forceinline__ uint32_t fld9(uint64_t w, uint32_t k) { return (uint32_t)(w >> (9u * k)) & 511u; }
__device__ __forceinline__ uint32_t timesteps2(uint32_t t) {                                                  // 2 x timesteps
    return t == 0 ? 20u : t == 1 ? 30u : t == 2 ? 40u : t == 3 ? 50u : t == 4 ? 60u : 36u;
}
// product meta word: type(3) | station+1 (3) << 3 | alive << 6 | remaining half-steps (6) << 7
__device__ __forceinline__ uint32_t mk_meta(uint32_t type, uint32_t csp1, uint32_t alive, uint32_t rem2) { return type | (csp1 << 3) | (alive << 6) | (rem2 << 7); }

struct Env {
    Pcg64 g;
    uint32_t mode, emergency, disruption, disruption_cd, timestep, needs_reset, overflow, raw, energy, targets;
    uint32_t nprod, lo, ncomp, ngood, nscrap, nhist, cnt_lt, cnt_gt, episodes;
    uint64_t completed, qlen, cnt[5], nT, curm;
    uint64_t th, curs;             // 9-bit fields: head place of each type's list; list place of the product at each station
    int32_t total_reward;
    double util[5], degr[5], curq[5], mean20, mean10, thr[3];
    double hsum;                   // running sum of the last <= 100 quality_rate_history entries (decides the sign of mean - 0.6 when it is not close)
    uint32_t status[5], ops[5], cur[5], qhead[5], qtail[5];
    int32_t mcount[5];

    __device__ __forceinline__ void unpack(const uint32_t (&r)[COLS * 4]) {
        g.state = ((u128)(((uint64_t)r[3] << 32) | r[2]) << 64) | (((uint64_t)r[1] << 32) | r[0]);
        g.inc = ((u128)(((uint64_t)r[7] << 32) | r[6]) << 64) | (((uint64_t)r[5] << 32) | r[4]);
        g.uinteger = r[8];
        const uint32_t m = r[9];
        g.has_uint32 = m & 1u; mode = (m >> 1) & 3u; emergency = (m >> 3) & 1u; disruption = (m >> 4) & 1u; disruption_cd = (m >> 5) & 63u;
        timestep = (m >> 11) & 2047u; needs_reset = (m >> 22) & 1u; overflow = (m >> 23) & 1u;
        raw = r[10] & 1023u; energy = r[10] >> 10; targets = r[11];
        completed = ((uint64_t)r[13] << 32) | r[12];
        total_reward = (int32_t)r[14];
        nprod = r[15] & 1023u; lo = (r[15] >> 10) & 1023u; ncomp = (r[15] >> 20) & 1023u;
        ngood = r[16] & 1023u; nscrap = (r[16] >> 10) & 1023u; nhist = r[16] >> 20;
        cnt_lt = r[17] & 127u; cnt_gt = (r[17] >> 7) & 127u; episodes = r[17] >> 14;
#pragma unroll
        for (int s = 0; s < 5; ++s) {
            util[s] = mk_double(r[18 + 2 * s], r[19 + 2 * s]); degr[s] = mk_double(r[28 + 2 * s], r[29 + 2 * s]); curq[s] = mk_double(r[38 + 2 * s], r[39 + 2 * s]);
            status[s] = r[58 + s] & 3u; ops[s] = (r[58 + s] >> 2) & 4095u; mcount[s] = (int32_t)(r[58 + s] >> 14) - 32768;
            cur[s] = r[63 + s] & 1023u; qhead[s] = (r[63 + s] >> 10) & 1023u; qtail[s] = (r[63 + s] >> 20) & 1023u;
            cnt[s] = ((uint64_t)r[71 + 2 * s] << 32) | r[70 + 2 * s];
        }
        mean20 = mk_double(r[48], r[49]); mean10 = mk_double(r[50], r[51]);
#pragma unroll
        for (int c = 0; c < 3; ++c) thr[c] = mk_double(r[52 + 2 * c], r[53 + 2 * c]);
        qlen = ((uint64_t)r[69] << 32) | r[68];
        nT = ((uint64_t)r[81] << 32) | r[80];
        curm = ((uint64_t)r[83] << 32) | r[82];
        th = ((uint64_t)r[85] << 32) | r[84];
        curs = ((uint64_t)r[87] << 32) | r[86];
        hsum = mk_double(r[88], r[89]);
    }
    __device__ __forceinline__ void pack(uint32_t (&r)[COLS * 4]) const {
        const uint64_t sl = (uint64_t)g.state, sh = (uint64_t)(g.state >> 64), il = (uint64_t)g.inc, ih = (uint64_t)(g.inc >> 64);
        r[0] = (uint32_t)sl; r[1] = (uint32_t)(sl >> 32); r[2] = (uint32_t)sh; r[3] = (uint32_t)(sh >> 32);
        r[4] = (uint32_t)il; r[5] = (uint32_t)(il >> 32); r[6] = (uint32_t)ih; r[7] = (uint32_t)(ih >> 32);
        r[8] = g.uinteger;
        r[9] = g.has_uint32 | (mode << 1) | (emergency << 3) | (disruption << 4) | (disruption_cd << 5) | (timestep << 11) | (needs_reset << 22) | (overflow << 23);
        r[10] = raw | (energy << 10); r[11] = targets;
        r[12] = (uint32_t)completed; r[13] = (uint32_t)(completed >> 32);
        r[14] = (uint32_t)total_reward;
        r[15] = nprod | (lo << 10) | (ncomp << 20);
        r[16] = ngood | (nscrap << 10) | (nhist << 20);
        r[17] = cnt_lt | (cnt_gt << 7) | (episodes << 14);
#pragma unroll
        for (int s = 0; s < 5; ++s) {
            r[18 + 2 * s] = (uint32_t)__double2loint(util[s]); r[19 + 2 * s] = (uint32_t)__double2hiint(util[s]);
            r[28 + 2 * s] = (uint32_t)__double2loint(degr[s]); r[29 + 2 * s] = (uint32_t)__double2hiint(degr[s]);
            r[38 + 2 * s] = (uint32_t)__double2loint(curq[s]); r[39 + 2 * s] = (uint32_t)__double2hiint(curq[s]);
            r[58 + s] = status[s] | (ops[s] << 2) | ((uint32_t)(mcount[s] + 32768) << 14);
            r[63 + s] = cur[s] | (qhead[s] << 10) | (qtail[s] << 20);
            r[70 + 2 * s] = (uint32_t)cnt[s]; r[71 + 2 * s] = (uint32_t)(cnt[s] >> 32);
        }
        r[48] = (uint32_t)__double2loint(mean20); r[49] = (uint32_t)__double2hiint(mean20);
        r[50] = (uint32_t)__double2loint(mean10); r[51] = (uint32_t)__double2hiint(mean10);
#pragma unroll
        for (int c = 0; c < 3; ++c) { r[52 + 2 * c] = (uint32_t)__double2loint(thr[c]); r[53 + 2 * c] = (uint32_t)__double2hiint(thr[c]); }
        r[68] = (uint32_t)qlen; r[69] = (uint32_t)(qlen >> 32);
        r[80] = (uint32_t)nT; r[81] = (uint32_t)(nT >> 32);
        r[82] = (uint32_t)curm; r[83] = (uint32_t)(curm >> 32);
        r[84] = (uint32_t)th; r[85] = (uint32_t)(th >> 32);
        r[86] = (uint32_t)curs; r[87] = (uint32_t)(curs >> 32);
        r[88] = (uint32_t)__double2loint(hsum); r[89] = (uint32_t)__double2hiint(hsum); r[90] = 0; r[91] = 0;
    }
    __device__ __forceinline__ void load(const uint4 *__restrict__ s, int64_t n, int64_t i) {
        uint32_t r[COLS * 4];
#pragma unroll
        for (int c = 0; c < COLS; ++c) {
            const uint4 v = s[(int64_t)c * n + i];
            r[4 * c] = v.x; r[4 * c + 1] = v.y; r[4 * c + 2] = v.z; r[4 * c + 3] = v.w;
        }
        unpack(r);
    }
    __device__ __forceinline__ void store(uint4 *__restrict__ s, int64_t n, int64_t i) const {
        uint32_t r[COLS * 4];
        pack(r);
#pragma unroll
        for (int c = 0; c < COLS; ++c) s[(int64_t)c * n + i] = make_uint4(r[4 * c], r[4 * c + 1], r[4 * c + 2], r[4 * c + 3]);
    }
};

// Bounds-checked debug build (-DCGE_MFG_GUARD: `python -m custom_gymnasium_environments_amd.build --guard`, probe tools/probes/guard_run.py):
// every [row][env] table index goes through GX(site, index, limit); an index outside its table is RECORDED (first violation: site,
// index, limit, block, lane; plus a count) and replaced by row 0 instead of being dereferenced — the out-of-range store of round 3's
// dense-list rewrite aborted the process at the next copy to the host, far from its cause.  Release builds: GX is the index.
#ifdef CGE_MFG_GUARD
__device__ unsigned int g_guard[8];
__device__ __forceinline__ uint32_t guard_index(int site_, uint32_t index, uint32_t limit) {
    if (index < limit) return index;
    if (atomicAdd(&g_guard[0], 1u) == 0u) { g_guard[1] = (unsigned)site_; g_guard[2] = index; g_guard[3] = limit; g_guard[4] = blockIdx.x; g_guard[5] = threadIdx.x; }
    return 0u;
}
#define GX(site_, index, limit) guard_index(site_, (uint32_t)(index), (uint32_t)(limit))
#else
#define GX(site_, index, limit) (index)
#endif

struct Tab {       // this env's column of every [row][env] table
    double *pq;
    uint16_t *pm, *pnext, *pts;
    double *tq;
    uint16_t *tid;
    double *comp, *hist;
    int64_t n;
    __device__ __forceinline__ Tab(const Params &p, int64_t i)
        : pq(p.pq + i), pm(p.pm + i), pnext(p.pnext + i), pts(p.pts + i), tq(p.tq + i), tid(p.tid + i), comp(p.comp + i), hist(p.hist + i), n(p.n) {}
};

// A product of `type` at place r of its list leaves the system (completed :482-500, scrapped :468-478): the older entries
// [head, r) move up one place (none, when it is the oldest), the head advances.  `curs` follows for the products at the stations.
__device__ __forceinline__ void list_remove(Env &e, const Tab &tb, uint32_t type, uint32_t r) {
    const uint32_t head = fld9(e.th, type);
    if (r != head) {
        double *q = tb.tq + (int64_t)(type * CAP) * tb.n;
        uint16_t *id = tb.tid + (int64_t)(type * CAP) * tb.n;
#pragma unroll 1
        for (uint32_t s = r; s > head;) {
            const uint32_t m = s - head < 4u ? s - head : 4u;
            double qv[4];
            uint32_t iv[4];
#pragma unroll
            for (uint32_t j = 0; j < 4u; ++j) if (j < m) { qv[j] = q[(int64_t)GX(25, s - 1u - j, CAP) * tb.n]; iv[j] = id[(int64_t)GX(25, s - 1u - j, CAP) * tb.n]; }
#pragma unroll
            for (uint32_t j = 0; j < 4u; ++j) if (j < m) { q[(int64_t)GX(26, s - j, CAP) * tb.n] = qv[j]; id[(int64_t)GX(26, s - j, CAP) * tb.n] = (uint16_t)iv[j]; tb.pts[(int64_t)GX(1, iv[j], CAP) * tb.n] = (uint16_t)(s - j); }
            s -= m;
        }
#pragma unroll
        for (int S = 0; S < 5; ++S) {
            const uint32_t cs = fld9(e.curs, S);
            if (e.cur[S] != NONE && ((uint32_t)(e.curm >> (12 * S)) & 7u) == type && cs >= head && cs < r) e.curs += 1ull << (9 * S);
        }
    }
    e.th += 1ull << (9u * type);
    e.nT -= 1ull << (9u * type);
}

__device__ __forceinline__ double combine8(const double *a) { return ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7])); }

// np.mean of the last m = min(ncomp, last) completed qualities (ring of 20), NumPy pairwise order, m <= 20
__device__ __forceinline__ double recent_mean(const Env &e, const Tab &tb, uint32_t last) {
    const uint32_t m = e.ncomp < last ? e.ncomp : last;
    double v[20];
    uint32_t idx = (e.ncomp - m) % 20u;
#pragma unroll
    for (int k = 0; k < 20; ++k) {
        v[k] = (uint32_t)k < m ? tb.comp[(int64_t)GX(2, idx, 20) * tb.n] : 0.0;
        idx = idx + 1u == 20u ? 0u : idx + 1u;
    }
    double res;
    if (m < 8u) {
        res = 0.0;
#pragma unroll
        for (int k = 0; k < 7; ++k) res = (uint32_t)k < m ? res + v[k] : res;
    } else {
        const uint32_t nfull = m & ~7u;
        double r[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] = nfull == 16u ? v[j] + v[8 + j] : v[j];
        res = combine8(r);
#pragma unroll
        for (int k = 8; k < 20; ++k) res = ((uint32_t)k >= nfull && (uint32_t)k < m) ? res + v[k] : res;
    }
    return res / (double)m;
}

// exact np.mean(quality_rate_history[-100:]) over the ring, oldest entry at nhist % 100
__device__ __forceinline__ double hist_mean(const Env &e, const Tab &tb) {
    uint32_t idx = e.nhist % 100u;
    double r[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = 0.0;
#pragma unroll 1
    for (int b = 0; b < 12; ++b) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const double x = tb.hist[(int64_t)GX(3, idx, 100) * tb.n];
            r[j] = b == 0 ? x : r[j] + x;
            idx = idx + 1u == 100u ? 0u : idx + 1u;
        }
    }
    double res = combine8(r);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        res += tb.hist[(int64_t)GX(4, idx, 100) * tb.n];
        idx = idx + 1u == 100u ? 0u : idx + 1u;
    }
    return res / 100.0;
}

template <int S>
__device__ __forceinline__ void queue_push(Env &e, const Tab &tb, uint32_t id) {
    if (fld9(e.qlen, S) == 0) e.qhead[S] = id;
    else tb.pnext[(int64_t)GX(5, e.qtail[S], CAP) * tb.n] = (uint16_t)id;
    e.qtail[S] = id;
    e.qlen += 1ull << (9 * S);
}

__device__ __forceinline__ void do_reset(Env &e) {                                                   // :113-192
#pragma unroll
    for (int s = 0; s < 5; ++s) {
        e.status[s] = OPERATIONAL; e.util[s] = 0.0; e.ops[s] = 0;
        e.mcount[s] = (int32_t)e.g.integers(100, 200);
        e.cur[s] = NONE; e.qhead[s] = 0; e.qtail[s] = 0; e.degr[s] = 0.0; e.curq[s] = 0.0; e.cnt[s] = 0;
    }
    e.qlen = 0; e.nT = 0; e.curm = 0; e.th = 0; e.curs = 0; e.nprod = 0; e.lo = 0; e.ncomp = 0; e.ngood = 0; e.nscrap = 0; e.nhist = 0; e.cnt_lt = 0; e.cnt_gt = 0;
    e.thr[0] = 0.70; e.thr[1] = 0.80; e.thr[2] = 0.85;
    e.raw = 250;
    uint32_t tg = 0;
    tg |= (uint32_t)e.g.integers(5, 10);
    tg |= (uint32_t)e.g.integers(4, 8) << 4;
    tg |= (uint32_t)e.g.integers(3, 6) << 8;
    tg |= (uint32_t)e.g.integers(2, 5) << 12;
    tg |= (uint32_t)e.g.integers(2, 4) << 16;
    tg |= (uint32_t)e.g.integers(3, 7) << 20;
    e.targets = tg; e.completed = 0;
    e.mode = BALANCED; e.emergency = 0; e.timestep = 0; e.total_reward = 0; e.disruption = 0; e.disruption_cd = 0; e.energy = 0;
    e.mean20 = 0.0; e.mean10 = 0.0; e.hsum = 0.0; e.needs_reset = 0;
}

// one station of _update_production :386-425
template <int S>
__device__ __forceinline__ void station_update(Env &e, const Tab &tb, bool &completed_any) {
    if (e.status[S] != OPERATIONAL) return;
    if (e.cur[S] != NONE) {
        const uint32_t id = e.cur[S];
        uint32_t cm = (uint32_t)(e.curm >> (12 * S)) & 4095u;                       // type(3) | station+1 (3) | remaining half-steps (6)
        const uint32_t type = cm & 7u, csp1 = (cm >> 3) & 7u;
        int32_t rem2 = (int32_t)(cm >> 6) - 2;
        double q = e.curq[S];
        if (e.mode == RUSH) { rem2 -= 1; q *= 0.98; }
        else if (e.mode == QUALITY) q *= 1.02;
        q *= (1 - e.degr[S]);
        if (rem2 <= 0) {
            e.cur[S] = NONE; e.ops[S] += 1;
            const uint32_t next = csp1;                                             // current_station + 1
            const uint32_t req = type == 5u ? 3u : type + 1u;                       // stations_required :35-40
            if (next < req) {
                tb.pq[(int64_t)GX(6, id, CAP) * tb.n] = q;
                tb.tq[(int64_t)GX(7, (type * CAP + fld9(e.curs, S)), TROW) * tb.n] = q;
                tb.pm[(int64_t)GX(8, id, CAP) * tb.n] = (uint16_t)mk_meta(type, next + 1u, 1u, 0u);
                if (S == 0 && csp1 == 0) {                                          // first visit ends: queued at station 0 again (:367, :410-415)
                    queue_push<0>(e, tb, id);
                    e.cnt[0] += 1ull << (9u * type);
                } else {
                    e.cnt[S] -= 1ull << (9u * type);
                    if (S < 4) { queue_push<(S < 4 ? S + 1 : 4)>(e, tb, id); e.cnt[S < 4 ? S + 1 : 4] += 1ull << (9u * type); }
                }
            } else {                                                                // _complete_product :482-500
                e.completed += 1ull << (9u * type);
                tb.comp[(int64_t)GX(9, (e.ncomp % 20u), 20) * tb.n] = q;
                e.ncomp += 1; e.ngood += q > 0.7 ? 1u : 0u;
                tb.pm[(int64_t)GX(10, id, CAP) * tb.n] = (uint16_t)mk_meta(type, csp1, 0u, 0u);
                list_remove(e, tb, type, fld9(e.curs, S));
                e.cnt[S] -= 1ull << (9u * type);
                completed_any = true;
            }
        } else {
            e.curq[S] = q;
            tb.tq[(int64_t)GX(11, (type * CAP + fld9(e.curs, S)), TROW) * tb.n] = q;                                // the observation's per-type mean reads the list
            cm = (cm & 63u) | ((uint32_t)rem2 << 6);
            e.curm = (e.curm & ~(4095ull << (12 * S))) | ((uint64_t)cm << (12 * S));
        }
    }
    if (e.cur[S] == NONE && fld9(e.qlen, S) > 0) {                                 // load next product from queue
        const uint32_t id = e.qhead[S];
        e.cur[S] = id;
        e.curq[S] = tb.pq[(int64_t)GX(12, id, CAP) * tb.n];
        const uint32_t m = tb.pm[(int64_t)GX(13, id, CAP) * tb.n];
        const uint32_t cm = (m & 63u) | (((m >> 7) & 63u) << 6);
        e.curm = (e.curm & ~(4095ull << (12 * S))) | ((uint64_t)cm << (12 * S));
        e.curs = (e.curs & ~(511ull << (9 * S))) | ((uint64_t)tb.pts[(int64_t)GX(14, id, CAP) * tb.n] << (9 * S));
        e.qlen -= 1ull << (9 * S);
        if (fld9(e.qlen, S) > 0) e.qhead[S] = tb.pnext[(int64_t)GX(15, id, CAP) * tb.n];
        e.util[S] = 0.8;
    } else {
        e.util[S] *= 0.95;
    }
}

template <int S>
__device__ __forceinline__ void machine_update(Env &e) {                                             // :429-462
    if (e.status[S] == MAINTENANCE) {
        e.mcount[S] -= 1;
        if (e.mcount[S] <= 0) { e.status[S] = OPERATIONAL; e.degr[S] = 0.0; e.ops[S] = 0; e.mcount[S] = (int32_t)e.g.integers(100, 200); }
    } else if (e.status[S] == OPERATIONAL) {
        const double prob = 0.001 * (1 + (double)e.ops[S] / 100);
        if (e.g.random() < prob) { e.status[S] = BROKEN; e.mcount[S] = 30; }
        if (e.ops[S] % 100u == 0) e.degr[S] += 0.005;
        e.mcount[S] -= 1;
    } else {
        e.mcount[S] -= 1;
        if (e.mcount[S] <= 0) { e.status[S] = OPERATIONAL; e.mcount[S] = (int32_t)e.g.integers(100, 200); }
    }
}

template <int C, int S>
__device__ __forceinline__ void quality_check(Env &e, const Tab &tb, int32_t &reward) {              // :468-478
    if (e.cur[S] != NONE && e.curq[S] < e.thr[C]) {
        const uint32_t cm = (uint32_t)(e.curm >> (12 * S)) & 4095u, type = cm & 7u;
        tb.pm[(int64_t)GX(16, e.cur[S], CAP) * tb.n] = (uint16_t)mk_meta(type, (cm >> 3) & 7u, 0u, 0u);
        list_remove(e, tb, type, fld9(e.curs, S));
        e.cnt[S] -= 1ull << (9u * type);
        e.cur[S] = NONE; e.nscrap += 1;
        reward -= 100;
    }
}

// returns terminated | truncated << 1
__device__ __forceinline__ uint32_t env_step(Env &e, const Tab &tb, int32_t max_steps, int32_t action, int32_t &reward_out) {   // :252-301
    int32_t reward = 0;
    e.timestep += 1;
    if (action >= 0 && action <= 5) {                                                                 // _process_action :303-359
        if (e.raw >= 10u) {
            if (e.nprod < (uint32_t)CAP) {                                                            // _start_production :361-379
                const uint32_t id = e.nprod, type = (uint32_t)action;
                const double q = 0.85 + e.g.uniform(-0.1, 0.1);
                const uint32_t place = fld9(e.th, type) + fld9(e.nT, type);             // < CAP: places are never reused within an episode
                tb.pq[(int64_t)GX(17, id, CAP) * tb.n] = q;
                tb.pm[(int64_t)GX(18, id, CAP) * tb.n] = (uint16_t)mk_meta(type, 0u, 1u, timesteps2(type));
                tb.pts[(int64_t)GX(19, id, CAP) * tb.n] = (uint16_t)place;
                tb.tq[(int64_t)GX(20, (type * CAP + place), TROW) * tb.n] = q;
                tb.tid[(int64_t)GX(21, (type * CAP + place), TROW) * tb.n] = (uint16_t)id;
                e.nT += 1ull << (9u * type);
                if (e.status[0] == OPERATIONAL) queue_push<0>(e, tb, id);
                e.nprod += 1;
            } else {
                e.overflow = 1;                                                                       // cannot happen inside an episode
            }
            e.raw -= 10u;
        } else reward -= 50;
    } else if (action >= 6 && action <= 10) {
        const uint32_t s = (uint32_t)(action - 6);
#pragma unroll
        for (int k = 0; k < 5; ++k) if (s == (uint32_t)k) { const double u = e.util[k] + 0.2; e.util[k] = u < 1.0 ? u : 1.0; }
        e.energy += 5;
    } else if (action >= 11 && action <= 15) {
        const uint32_t s = (uint32_t)(action - 11);
#pragma unroll
        for (int k = 0; k < 5; ++k)
            if (s == (uint32_t)k && e.status[k] == OPERATIONAL) { e.status[k] = MAINTENANCE; e.mcount[k] = 20; reward += 50; }
    } else if (action >= 16 && action <= 20) {
        const uint32_t c = (uint32_t)(action - 16);
#pragma unroll
        for (int k = 0; k < 3; ++k) if (c == (uint32_t)k) { const double t = e.thr[k] + 0.05; e.thr[k] = t < 0.95 ? t : 0.95; }
    } else if (action == 21) {
        e.emergency ^= 1u;
        if (e.emergency) reward -= 100;
    } else if (action == 22) { e.mode = RUSH; e.energy += 10; }
    else if (action == 23) e.mode = QUALITY;
    else if (action == 24) e.mode = BALANCED;
    bool completed_any = false;
    if (!e.emergency) {                                                                               // _update_production :381-425
        station_update<0>(e, tb, completed_any); station_update<1>(e, tb, completed_any); station_update<2>(e, tb, completed_any);
        station_update<3>(e, tb, completed_any); station_update<4>(e, tb, completed_any);
    }
    machine_update<0>(e); machine_update<1>(e); machine_update<2>(e); machine_update<3>(e); machine_update<4>(e);
    quality_check<0, 1>(e, tb, reward); quality_check<1, 3>(e, tb, reward); quality_check<2, 4>(e, tb, reward);
    // _update_metrics :533-553 — both recent-quality means only change when a product completes
    if (completed_any) { e.mean20 = recent_mean(e, tb, 20u); e.mean10 = recent_mean(e, tb, 10u); }
    if (e.ncomp > 0) {
        const uint32_t pos = e.nhist % 100u;
        if (e.nhist >= 100u) {
            const double old = tb.hist[(int64_t)GX(22, pos, 100) * tb.n];
            e.cnt_lt -= old < 0.61 ? 1u : 0u; e.cnt_gt -= old > 0.59 ? 1u : 0u;
            e.hsum -= old;
        }
        e.hsum += e.mean20;
        tb.hist[(int64_t)GX(23, pos, 100) * tb.n] = e.mean20;
        e.cnt_lt += e.mean20 < 0.61 ? 1u : 0u; e.cnt_gt += e.mean20 > 0.59 ? 1u : 0u;
        e.nhist += 1;
    }
    // _calculate_timestep_rewards :502-531
    uint32_t broken = 0;
    bool all_met = true;
#pragma unroll
    for (int s = 0; s < 5; ++s) {
        if (e.status[s] == OPERATIONAL && e.cur[s] == NONE && fld9(e.qlen, s) == 0) reward -= 10;
        broken += e.status[s] == BROKEN ? 1u : 0u;
    }
    reward -= (int32_t)broken * 50;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const uint32_t done = fld9(e.completed, k), target = (e.targets >> (4 * k)) & 15u;
        if (e.timestep > 1000u && done < target) reward -= 50;
        all_met = all_met && done >= target;
    }
    if (e.ncomp > 0) { if (e.mean10 > 0.9) reward += 20; else if (e.mean10 < 0.6) reward -= 30; }
    // _check_termination :555-578.  The 100-entry history mean is only evaluated when its sign is not already decided:
    // no entry below 0.61 -> mean > 0.6; no entry above 0.59 -> mean < 0.6 (summation error is ~1e-14); and the running sum
    // (entries in [0, 1], <= 3,000 additions and subtractions per episode: off by < 1e-11) decides it unless it lands within
    // 1e-9 of 0.6 — the exact 100-term pairwise sum (13 dependent round trips; late in an episode some lane of nearly every wave
    // wanted it at every step) is left for that case.
    bool term = all_met || broken >= 3u || e.timestep >= (uint32_t)max_steps;
    if (!term && e.nhist >= 100u && e.cnt_lt != 0u) {
        const double approx = e.hsum / 100.0, off = approx - 0.6;
        term = e.cnt_gt == 0u ? true : (off > 1e-9 || off < -1e-9) ? off < 0.0 : hist_mean(e, tb) < 0.6;
    }
    const bool trunc = e.timestep >= (uint32_t)max_steps;
    // _update_supply_chain :580-595
    if (!e.disruption && e.g.random() < 0.01) { e.disruption = 1; e.disruption_cd = (uint32_t)e.g.integers(20, 50); }
    if (e.disruption) { e.disruption_cd -= 1; if (e.disruption_cd == 0) e.disruption = 0; }
    else if (e.timestep % 50u == 0) { const uint32_t r = e.raw + (uint32_t)e.g.integers(50, 100); e.raw = r < 500u ? r : 500u; }
    e.total_reward += reward;
    reward_out = reward;
    return (term ? 1u : 0u) | (trunc ? 2u : 0u);
}

__device__ __forceinline__ uint32_t wave_min(uint32_t v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { const uint32_t o = (uint32_t)__shfl_xor((int)v, d, 64); v = o < v ? o : v; }
    return v;
}
__device__ __forceinline__ uint32_t wave_max(uint32_t v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { const uint32_t o = (uint32_t)__shfl_xor((int)v, d, 64); v = o > v ? o : v; }
    return v;
}

// NumPy's pairwise_sum of n <= 128 values (stride `st` apart): fewer than 8 -> in order from 0.0; else 8 accumulators over the
// whole runs of 8, combined, then the remainder in order (numpy/core/src/umath/loops_utils.h.src)
__device__ __forceinline__ double leaf_sum(const double *a, int64_t st, uint32_t n) {
    double res = 0.0;
    uint32_t i = 0;
    if (n >= 8u) {
        double r[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) r[k] = a[k * st];
#pragma unroll 1
        for (i = 8u; i < (n & ~7u); i += 8u) {
#pragma unroll
            for (int k = 0; k < 8; ++k) r[k] += a[(int64_t)(i + k) * st];
        }
        res = combine8(r);
    }
#pragma unroll 1
    for (; i < n; ++i) res += a[(int64_t)i * st];
    return res;
}

// ... of 128 < n <= 320 values (rare): the recursion halves the list (left half rounded down to a multiple of 8) until every
// leaf has <= 128 — at most four leaves, two levels
__device__ __forceinline__ double big_sum(const double *a, int64_t st, uint32_t n) {
    uint32_t n2 = n / 2u; n2 -= n2 % 8u;
    const uint32_t nr = n - n2;
    double side[2];
#pragma unroll 1
    for (uint32_t h = 0; h < 2u; ++h) {
        const double *x = h ? a + (int64_t)n2 * st : a;
        const uint32_t m = h ? nr : n2;
        if (m <= 128u) side[h] = leaf_sum(x, st, m);
        else {
            uint32_t m2 = m / 2u; m2 -= m2 % 8u;
            const double l = leaf_sum(x, st, m2);
            side[h] = l + leaf_sum(x + (int64_t)m2 * st, st, m - m2);
        }
    }
    return side[0] + side[1];
}

// one run of 8 places of a lane's list from place `p0` of type block `tbase` (places past the table's end re-read its last row)
__device__ __forceinline__ void load_run(const Tab &tb, uint32_t tbase, uint32_t p0, double (&v)[8]) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const uint32_t pl = p0 + k < (uint32_t)CAP ? p0 + k : (uint32_t)CAP - 1u;
        v[k] = tb.tq[(int64_t)GX(24, (tbase + pl), TROW) * tb.n];
    }
}

// np.mean of the qualities of the products of type T (<= 128 of them; longer lists take big_sum): `nrun` = the wave's longest
// such list in runs of 8.  F holds the list's first run (loaded by the previous type's call); this call loads type T + 1's
// first run into N before anything else, and its own next run while the last one is added: the loads are unconditional (past
// the last run they re-read run 0) so that the compiler's wait counters stay exact and never drain a prefetch.
template <int T>
__device__ __forceinline__ double type_mean(const Env &e, const Tab &tb, bool live, uint32_t nrun, double (&F)[8], double (&N)[8]) {
    if constexpr (T < 5) load_run(tb, (T + 1) * CAP, fld9(e.th, T + 1), N);
    const uint32_t nt = live ? fld9(e.nT, T) : 0u, n = nt > 128u ? 0u : nt, nb = n >> 3, rem = n & 7u;
    if (nrun == 0u) return 0.0;
    const uint32_t head = fld9(e.th, T);
    double Y[8], r[8], res = 0.0;
#pragma unroll
    for (int k = 0; k < 8; ++k) r[k] = 0.0;
    const auto add_run = [&](const double (&V)[8], uint32_t c) {
#pragma unroll
        for (int k = 0; k < 8; ++k) asm volatile("" ::"v"(V[k]));   // the run counts as used on every path: a load the compiler thinks may still be pending makes it drain everything at the loop head
        if (c < nb) {
#pragma unroll
            for (int k = 0; k < 8; ++k) r[k] += V[k];           // (0.0 + a == a exactly: qualities are positive)
        } else if (c == nb && rem != 0u) {
            res = nb ? combine8(r) : 0.0;
#pragma unroll
            for (int k = 0; k < 7; ++k) res = (uint32_t)k < rem ? res + V[k] : res;
        }
    };
    uint32_t c = 0;
#pragma unroll 1
    for (;;) {
        load_run(tb, T * CAP, head + (c + 1u < nrun ? 8u * (c + 1u) : 0u), Y);
        add_run(F, c);
        if (++c >= nrun) break;
        load_run(tb, T * CAP, head + (c + 1u < nrun ? 8u * (c + 1u) : 0u), F);
        add_run(Y, c);
        if (++c >= nrun) break;
    }
    if (rem == 0u) res = nb ? combine8(r) : 0.0;
    return n ? res / (double)n : 0.0;
}

// obs[50..55]: np.mean of quality_score per product type over products_in_system (:220-228), NumPy pairwise order
__device__ __forceinline__ void type_means(const Env &e, const Tab &tb, bool live, double (&mean)[6]) {
    uint32_t nrun[6];                      // wave-uniform
    unsigned big = 0;
#pragma unroll
    for (int t = 0; t < 6; ++t) {
        const uint32_t nt = live ? fld9(e.nT, t) : 0u;
        const bool isbig = nt > 128u;
        if (__ballot(isbig) != 0ull) big |= 1u << t;
        nrun[t] = ((uint32_t)__builtin_amdgcn_readfirstlane((int)wave_max(isbig ? 0u : nt)) + 7u) / 8u;
    }
    double F[8], N[8];
    __builtin_amdgcn_s_waitcnt(0x0F70);                         // vmcnt(0): with older loads still counted the loops below would wait for them at every trip
    load_run(tb, 0, fld9(e.th, 0), F);
    mean[0] = type_mean<0>(e, tb, live, nrun[0], F, N);
    mean[1] = type_mean<1>(e, tb, live, nrun[1], N, F);
    mean[2] = type_mean<2>(e, tb, live, nrun[2], F, N);
    mean[3] = type_mean<3>(e, tb, live, nrun[3], N, F);
    mean[4] = type_mean<4>(e, tb, live, nrun[4], F, N);
    mean[5] = type_mean<5>(e, tb, live, nrun[5], N, F);
    if (big) {
#pragma unroll 1
        for (uint32_t tt = 0; tt < 6u; ++tt) {
            if (!((big >> tt) & 1u)) continue;
            const uint32_t nt = live ? fld9(e.nT, tt) : 0u;
            if (nt > 128u) {
                const double m = big_sum(tb.tq + (int64_t)(tt * CAP + fld9(e.th, tt)) * tb.n, tb.n, nt) / (double)nt;
#pragma unroll
                for (int k = 0; k < 6; ++k) mean[k] = tt == (uint32_t)k ? m : mean[k];
            }
        }
    }
}

__device__ __forceinline__ void stage_row(const Env &e, const double (&mean)[6], float *row) {       // :194-250
#pragma unroll
    for (int s = 0; s < 5; ++s) {
#pragma unroll
        for (int k = 0; k < 6; ++k) row[6 * s + k] = (float)fld9(e.cnt[s], k);
        row[30 + 3 * s] = e.status[s] == OPERATIONAL ? 1.0f : 0.0f; row[31 + 3 * s] = e.status[s] == BROKEN ? 1.0f : 0.0f;
        row[32 + 3 * s] = e.status[s] == MAINTENANCE ? 1.0f : 0.0f;
        row[45 + s] = (float)fld9(e.qlen, s);
        row[63 + s] = (float)(e.util[s] * 100);
        row[68 + s] = (float)e.mcount[s];
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        row[50 + k] = fld9(e.nT, k) ? (float)(mean[k] * 100) : 85.0f;
        const int32_t rem = (int32_t)((e.targets >> (4 * k)) & 15u) - (int32_t)fld9(e.completed, k);
        row[57 + k] = (float)(rem > 0 ? rem : 0);
    }
    row[56] = (float)e.raw;
}

__device__ __forceinline__ void store_rows(const RowMap &rm, float *__restrict__ dst, const uint32_t *__restrict__ tile) {
    const uint32_t lane = threadIdx.x & 63u;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (rm.nrows == 64 && rm.mask == ~0ull && !rm.compact && (reinterpret_cast<uintptr_t>(dst) & 15u) == 0u) {
        // the wave's 64 rows are one contiguous 18,688-byte image, in LDS as in the destination: 1168 16-byte pieces
#pragma unroll 5
        for (uint32_t k = lane; k < (uint32_t)(64 * OBS / 4); k += 64u) reinterpret_cast<uint4 *>(dst)[k] = reinterpret_cast<const uint4 *>(tile)[k];
    } else {
        uint32_t r = 0, col = lane;                             // OBS = 73 > 64: lane walks the tile linearly
#pragma unroll 1
        for (int m = 0; m < OBS; ++m) {
            int64_t to;
            if (rm.row(r, to)) reinterpret_cast<uint32_t *>(dst)[to * OBS + col] = tile[r * OBS + col];
            col += 64u;
            if (col >= (uint32_t)OBS) { col -= OBS; r += 1u; }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

#ifdef CGE_MFG_TIMING
__device__ unsigned long long g_timing[2048 * 8];
#define TICK(k) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); const unsigned long long now_ = wall_clock64(); \
    if (threadIdx.x == 0 && blockIdx.x < 2048) { g_timing[blockIdx.x * 8 + k] += now_ - t_last; } t_last = now_; } while (0)
#else
#define TICK(k)
#endif
template <bool ROLLOUT>
__global__ __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(2, 2))) void step_kernel(Params p) {   // 131,072 envs are 2 waves per SIMD: both must be resident (a few spilled registers cost less, round 3)
    __shared__ __attribute__((aligned(16))) uint32_t tile[64 * OBS];   // the wave's obs rows, staged for coalesced stores
    const int64_t i0 = (int64_t)blockIdx.x * BLOCK;
    const int64_t i = i0 + threadIdx.x;
    const bool live = i < p.n;
    const int64_t li = live ? i : i0;
    const int64_t nrows = p.n - i0 < 64 ? p.n - i0 : 64;
    const uint32_t lane = threadIdx.x & 63u;
    Env e;
    e.load(p.state, p.n, li);
    const Tab tb(p, li);
    const uint64_t key = ROLLOUT ? hash_env_key(p.a_seed, (uint64_t)(p.env0 + li)) : 0;
    double rsum = 0.0;
    int32_t dcount = 0;
    uint32_t fin_used = 0;                                     // terminal rows this wave has delivered to its segment (fused rollouts)
    const int ksteps = ROLLOUT ? p.k_steps : 1;
#ifdef CGE_MFG_TIMING
    unsigned long long t_last = wall_clock64();
#endif
#pragma unroll 1
    for (int t = 0; t < ksteps; ++t) {
        int32_t reward = 0;
        uint32_t flags = 0;
        bool reset_now = false;
        if (live) {
            if (p.mode == CGE_AUTORESET_NEXT_STEP && e.needs_reset) {
                do_reset(e);
            } else {
                const int32_t a = p.actions ? p.actions[(int64_t)t * p.n + i] : (int32_t)hash_action_from_key(key, (uint64_t)(p.t0 + t), 25u, 0u);
                TICK(0);
                flags = env_step(e, tb, p.max_steps, a, reward);
                TICK(1);
                if (flags) {
                    e.episodes += 1;
                    if (p.ep_ret) p.ep_ret[i] = (double)e.total_reward;    // integer rewards (:291), exact in float64
                    if (p.ep_len) p.ep_len[i] = (int32_t)e.timestep;
                    if (p.mode == CGE_AUTORESET_SAME_STEP) reset_now = true;
                    else if (p.mode == CGE_AUTORESET_NEXT_STEP) e.needs_reset = 1;
                }
            }
        }
        const bool want_obs = p.obs != nullptr;                // a rollout writes the obs of every step (stride 0: in place)
        // terminal rows: step() -> final_obs_out (SAME_STEP); fused rollout -> the wave's segment of the compacted side output
        const bool fin = live && flags && reset_now;
        const unsigned long long fin_mask = __ballot(ROLLOUT ? fin : reset_now);
        const bool want_fin = fin_mask && (ROLLOUT ? p.fin.rows != nullptr : p.final_obs != nullptr);
        if (want_obs || want_fin) {
            double mean[6];
            type_means(e, tb, live, mean);
            TICK(2);
            if (want_fin) {
                stage_row(e, mean, reinterpret_cast<float *>(tile) + lane * OBS);
                if (!ROLLOUT) {
                    store_rows(RowMap{fin_mask, nrows, 0, false}, p.final_obs + i0 * OBS, tile);
                } else {
                    float *fdst;
                    const RowMap rm = final_rows<float>(p.fin, (int64_t)blockIdx.x, fin_used, fin, fin_mask, nrows, t, i, OBS, fdst);
                    store_rows(rm, fdst, tile);
                }
            }
            if (ROLLOUT) fin_used += (uint32_t)__popcll(fin_mask);
            if (reset_now) do_reset(e);                        // fresh episode: nothing in the system, means default to 85
            if (want_obs) {
                stage_row(e, mean, reinterpret_cast<float *>(tile) + lane * OBS);
                store_rows(RowMap{~0ull, nrows, 0, false}, p.obs + (int64_t)t * p.obs_step_stride + i0 * OBS, tile);
            }
        } else {
            if (ROLLOUT) fin_used += (uint32_t)__popcll(fin_mask);     // (no side output registered: the count still says what was dropped)
            if (reset_now) do_reset(e);
        }
        TICK(3);
#ifdef CGE_MFG_TIMING
        if (threadIdx.x == 0 && blockIdx.x < 2048) g_timing[blockIdx.x * 8 + 7] += 1;
#endif
        if (live) {
            if (ROLLOUT) {
                rsum += (double)reward;
                dcount += flags ? 1 : 0;
                if (p.reward) p.reward[(int64_t)t * p.n + i] = (float)reward;
                if (p.terminated) p.terminated[(int64_t)t * p.n + i] = (uint8_t)flags;
            } else {
                p.reward[i] = (float)reward;
                p.terminated[i] = (uint8_t)(flags & 1u);
                p.truncated[i] = (uint8_t)((flags >> 1) & 1u);
                if (p.done) p.done[i] = flags ? 1 : 0;
            }
        }
    }
    if (live) {
        e.store(p.state, p.n, i);
        if (ROLLOUT) {
            if (p.reward_sum) p.reward_sum[i] = rsum;
            if (p.fin.count && threadIdx.x == 0) p.fin.count[blockIdx.x] = (int32_t)fin_used;
            if (p.done_count) p.done_count[i] = dcount;
        }
    }
}

// what: 0 = reset(mask) + obs, 1 = reseed the generators, 2 = fresh-handle state (generators seeded, nothing in the system)
__global__ __launch_bounds__(BLOCK) void reset_kernel(Params p, int what) {
    __shared__ __attribute__((aligned(16))) uint32_t tile[64 * OBS];
    const int64_t i0 = (int64_t)blockIdx.x * BLOCK;
    const int64_t i = i0 + threadIdx.x;
    const bool live = i < p.n;
    const int64_t li = live ? i : i0;
    const int64_t nrows = p.n - i0 < 64 ? p.n - i0 : 64;
    const uint32_t lane = threadIdx.x & 63u;
    Env e;
    e.load(p.state, p.n, li);
    const Tab tb(p, li);
    if (live) {
        if (what == 1 || what == 2) {
            e.g.seed(p.seeds ? p.seeds[i] : p.base_seed + (uint64_t)(p.env0 + i));
            if (what == 2) {
#pragma unroll
                for (int s = 0; s < 5; ++s) { e.cur[s] = NONE; e.status[s] = OPERATIONAL; }
                e.thr[0] = 0.70; e.thr[1] = 0.80; e.thr[2] = 0.85; e.raw = 250;
            }
            e.store(p.state, p.n, i);
        } else if (!p.mask || p.mask[i]) {
            do_reset(e);
            e.store(p.state, p.n, i);
        }
    }
    if (what == 0 && p.obs) {
        double mean[6];
        type_means(e, tb, live, mean);
        stage_row(e, mean, reinterpret_cast<float *>(tile) + lane * OBS);
        store_rows(RowMap{~0ull, nrows, 0, false}, p.obs + i0 * OBS, tile);
    }
}

__global__ __launch_bounds__(256) void info_kernel(const uint4 *__restrict__ state, int64_t n, int field, double *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    Env e;
    e.load(state, n, i);
    double v = 0.0;
    uint32_t c = 0;
    switch (field) {
        case CGE_MANUFACTURING_INFO_RAW_MATERIAL: v = e.raw; break;
        case CGE_MANUFACTURING_INFO_ENERGY_CONSUMPTION: v = e.energy; break;
        case CGE_MANUFACTURING_INFO_TOTAL_REWARD: v = e.total_reward; break;
        case CGE_MANUFACTURING_INFO_IN_SYSTEM:
            for (int k = 0; k < 6; ++k) c += fld9(e.nT, k);
            v = c; break;
        case CGE_MANUFACTURING_INFO_COMPLETED: v = e.ncomp; break;
        case CGE_MANUFACTURING_INFO_SCRAPPED: v = e.nscrap; break;
        case CGE_MANUFACTURING_INFO_PRODUCT_IDS: v = e.nprod; break;
        case CGE_MANUFACTURING_INFO_HISTORY_LEN: v = e.nhist; break;
        case CGE_MANUFACTURING_INFO_OEE_AVAILABILITY:
            for (int s = 0; s < 5; ++s) c += e.status[s] == OPERATIONAL ? 1u : 0u;
            v = (double)c / 5; break;
        case CGE_MANUFACTURING_INFO_OEE_PERFORMANCE: {                                   // np.mean of 5 utilisation rates (:541-542); 1.0 after reset
            double sum = 0.0;
            for (int s = 0; s < 5; ++s) sum += e.util[s];
            v = e.timestep == 0 ? 1.0 : sum / 5.0; break;
        }
        case CGE_MANUFACTURING_INFO_OEE_QUALITY: v = e.ncomp ? (double)e.ngood / (double)e.ncomp : 1.0; break;
        case CGE_MANUFACTURING_INFO_TIMESTEP: v = e.timestep; break;
        case CGE_MANUFACTURING_INFO_EPISODES: v = e.episodes; break;
        case CGE_MANUFACTURING_INFO_NEEDS_RESET: v = e.needs_reset; break;
        case CGE_MANUFACTURING_INFO_OVERFLOW: v = e.overflow; break;
        default: if (field >= CGE_MANUFACTURING_INFO_COMPLETED_TYPE0 && field < CGE_MANUFACTURING_INFO_COMPLETED_TYPE0 + 6) v = fld9(e.completed, field - CGE_MANUFACTURING_INFO_COMPLETED_TYPE0);
    }
    out[i] = v;
}

}  // namespace mfg
}  // namespace cge

using namespace cge;

struct cge_manufacturing : HandleBase {
    cge_manufacturing_config cfg{};
    uint4 *state = nullptr;
    double *pq = nullptr, *comp = nullptr, *hist = nullptr;
    uint16_t *pm = nullptr, *pnext = nullptr, *pts = nullptr, *tid = nullptr;
    double *tq = nullptr;
    static constexpr uint32_t snap_tag = 4u;
    std::vector<std::pair<void *, size_t>> blobs() const { return {{state, (size_t)mfg::COLS * n * sizeof(uint4)}, {pq, (size_t)mfg::CAP * n * 8}, {pm, (size_t)mfg::CAP * n * 2}, {pnext, (size_t)mfg::CAP * n * 2}, {pts, (size_t)mfg::CAP * n * 2}, {tq, (size_t)mfg::TROW * n * 8}, {tid, (size_t)mfg::TROW * n * 2}, {comp, (size_t)20 * n * 8}, {hist, (size_t)100 * n * 8}}; }
    uint32_t snap_extra() const { return 0u; }
    void set_snap_extra(uint32_t v) { (void)v; }
    mfg::Params params() const {
        mfg::Params p{};
        p.state = state; p.pq = pq; p.pm = pm; p.pnext = pnext; p.pts = pts; p.tq = tq; p.tid = tid; p.comp = comp; p.hist = hist;
        p.n = n; p.env0 = env0; p.mode = cfg.autoreset_mode; p.max_steps = cfg.max_steps;
        p.ep_ret = ep_ret; p.ep_len = ep_len; p.done = done_out;
        return p;
    }
    unsigned blocks() const { return (unsigned)((n + mfg::BLOCK - 1) / mfg::BLOCK); }
    void free_all() { (void)hipFree(state); (void)hipFree(pq); (void)hipFree(pm); (void)hipFree(pnext); (void)hipFree(pts); (void)hipFree(tq); (void)hipFree(tid); (void)hipFree(comp); (void)hipFree(hist); }
};

extern "C" {

#ifdef CGE_MFG_GUARD
int cge_manufacturing_debug_guard(unsigned int *out) {      // [count, site, index, limit, block, lane, 0, 0] of the first violation
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(mfg::g_guard), 8 * sizeof(unsigned int)) == hipSuccess ? 0 : 1;
}
#endif

#ifdef CGE_MFG_TIMING
int cge_manufacturing_debug_timing(unsigned long long *out, int clear) {
    static unsigned long long all[2048 * 8];
    if (hipMemcpyFromSymbol(all, HIP_SYMBOL(mfg::g_timing), sizeof all) != hipSuccess) return 1;
    for (int k = 0; k < 8; ++k) out[k] = 0;
    for (int b = 0; b < 2048; ++b)
        for (int k = 0; k < 8; ++k) out[k] += all[b * 8 + k];
    if (clear) { memset(all, 0, sizeof all); if (hipMemcpyToSymbol(HIP_SYMBOL(mfg::g_timing), all, sizeof all) != hipSuccess) return 1; }
    return 0;
}
#endif

int cge_manufacturing_create(const cge_manufacturing_config *cfg, int64_t n_envs, int device, int64_t env_index0, cge_manufacturing **out) {
    if (!cfg || !out || n_envs <= 0 || env_index0 < 0) return CGE_ERR_INVALID_ARG;
    *out = nullptr;
    if (cfg->autoreset_mode < 0 || cfg->autoreset_mode > 2 || cfg->max_steps < 0 || cfg->max_steps > 1500) return CGE_ERR_INVALID_ARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return CGE_ERR_NO_DEVICE;
    cge_manufacturing *h = new cge_manufacturing();
    h->cfg = *cfg;
    if (h->cfg.max_steps == 0) h->cfg.max_steps = 1500;
    h->n = n_envs; h->env0 = env_index0; h->device = device;
    DeviceGuard g(device);
    const size_t N = (size_t)n_envs;
    const size_t sb = (size_t)mfg::COLS * N * sizeof(uint4), qb = (size_t)mfg::CAP * N * 8, mb = (size_t)mfg::CAP * N * 2, cb = 20 * N * 8, hb = 100 * N * 8;
    const size_t tqb = (size_t)mfg::TROW * N * 8;
    hipError_t e;
    if ((e = hipMalloc(&h->state, sb)) != hipSuccess || (e = hipMalloc(&h->pq, qb)) != hipSuccess || (e = hipMalloc(&h->pm, mb)) != hipSuccess ||
        (e = hipMalloc(&h->pnext, mb)) != hipSuccess || (e = hipMalloc(&h->pts, mb)) != hipSuccess ||
        (e = hipMalloc(&h->tq, tqb)) != hipSuccess || (e = hipMalloc(&h->tid, (size_t)mfg::TROW * N * 2)) != hipSuccess ||
        (e = hipMalloc(&h->comp, cb)) != hipSuccess || (e = hipMalloc(&h->hist, hb)) != hipSuccess ||
        (e = hipMemset(h->state, 0, sb)) != hipSuccess) {
        h->free_all();
        delete h;
        return CGE_ERR_HIP;
    }
    h->device_bytes = sb + qb + 3 * mb + tqb + (size_t)mfg::TROW * N * 2 + cb + hb;
    mfg::Params p = h->params();                               // default generators: PCG64(SeedSequence(env_index0 + i)); no reset
    hipLaunchKernelGGL(mfg::reset_kernel, dim3(h->blocks()), dim3(mfg::BLOCK), 0, nullptr, p, 2);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
    if (e != hipSuccess) {
        h->free_all();
        delete h;
        return CGE_ERR_HIP;
    }
    *out = h;
    return CGE_OK;
}

int cge_manufacturing_destroy(cge_manufacturing *h) {
    if (!h) return CGE_ERR_INVALID_ARG;
    DeviceGuard g(h->device);
    (void)hipDeviceSynchronize();
    h->free_all();
    delete h;
    return CGE_OK;
}

int cge_manufacturing_seed(cge_manufacturing *h, const uint64_t *seeds, uint64_t base_seed, void *stream) {
    if (!h) return CGE_ERR_INVALID_ARG;
    DeviceGuard g(h->device);
    mfg::Params p = h->params();
    p.seeds = seeds; p.base_seed = base_seed;
    hipLaunchKernelGGL(mfg::reset_kernel, dim3(h->blocks()), dim3(mfg::BLOCK), 0, as_stream(stream), p, 1);
    CGE_TRY(h, hipGetLastError());
    return CGE_OK;
}

int cge_manufacturing_reset(cge_manufacturing *h, const uint8_t *mask, float *obs_out, void *stream) {
    if (!h) return CGE_ERR_INVALID_ARG;
    DeviceGuard g(h->device);
    mfg::Params p = h->params();
    p.mask = mask; p.obs = obs_out;
    hipLaunchKernelGGL(mfg::reset_kernel, dim3(h->blocks()), dim3(mfg::BLOCK), 0, as_stream(stream), p, 0);
    CGE_TRY(h, hipGetLastError());
    return CGE_OK;
}

int cge_manufacturing_step(cge_manufacturing *h, const int32_t *actions, float *obs_out, float *reward_out, uint8_t *terminated_out,
                           uint8_t *truncated_out, float *final_obs_out, void *stream) {
    if (!h) return CGE_ERR_INVALID_ARG;
    if (!actions || !obs_out || !reward_out || !terminated_out || !truncated_out)
        return h->fail(CGE_ERR_INVALID_ARG, "cge_manufacturing_step: null actions/obs/reward/terminated/truncated pointer");
    DeviceGuard g(h->device);
    mfg::Params p = h->params();
    p.actions = actions; p.obs = obs_out; p.reward = reward_out; p.terminated = terminated_out; p.truncated = truncated_out;
    p.final_obs = final_obs_out; p.k_steps = 1;
    hipLaunchKernelGGL(mfg::step_kernel<false>, dim3(h->blocks()), dim3(mfg::BLOCK), 0, as_stream(stream), p);
    h->last_kernel = "cge::mfg::step_kernel<false>";
    CGE_TRY(h, hipGetLastError());
    return CGE_OK;
}

int cge_manufacturing_rollout(cge_manufacturing *h, int32_t k_steps, const int32_t *actions, uint64_t action_seed, int64_t t0, float *obs_out,
                              int64_t obs_step_stride, float *reward_traj_out, uint8_t *terminated_traj_out, double *reward_sum_out,
                              int32_t *done_count_out, void *stream) {
    if (!h) return CGE_ERR_INVALID_ARG;
    if (k_steps < 0 || obs_step_stride < 0 || (obs_step_stride != 0 && obs_step_stride < h->n * mfg::OBS))
        return h->fail(CGE_ERR_INVALID_ARG, "cge_manufacturing_rollout: bad k_steps / obs_step_stride");
    if (k_steps == 0) return CGE_OK;
    DeviceGuard g(h->device);
    mfg::Params p = h->params();
    p.k_steps = k_steps; p.actions = actions; p.a_seed = action_seed; p.t0 = t0; p.obs = obs_out; p.obs_step_stride = obs_step_stride;
    p.reward = reward_traj_out; p.terminated = terminated_traj_out; p.reward_sum = reward_sum_out; p.done_count = done_count_out;
    p.fin = FinalSeg{h->fin_rows, h->fin_index, h->fin_count, h->fin_cap, h->n};
    hipLaunchKernelGGL(mfg::step_kernel<true>, dim3(h->blocks()), dim3(mfg::BLOCK), 0, as_stream(stream), p);
    h->last_kernel = "cge::mfg::step_kernel<true>";
    CGE_TRY(h, hipGetLastError());
    return CGE_OK;
}

CGE_DEFINE_FINAL_OBS(manufacturing, float, 64)

int cge_manufacturing_info(cge_manufacturing *h, int32_t field_id, double *out, void *stream) {
    if (!h || !out || field_id < 0 || field_id > CGE_MANUFACTURING_INFO_COMPLETED_TYPE0 + 5) return CGE_ERR_INVALID_ARG;
    DeviceGuard g(h->device);
    hipLaunchKernelGGL(mfg::info_kernel, dim3((unsigned)((h->n + 255) / 256)), dim3(256), 0, as_stream(stream), h->state, h->n, field_id, out);
    CGE_TRY(h, hipGetLastError());
    return CGE_OK;
}

size_t cge_manufacturing_snapshot_bytes(const cge_manufacturing *h) { return h ? snapshot_bytes(h) : 0; }
int cge_manufacturing_snapshot_get(cge_manufacturing *h, void *host_buf, void *stream) { return snapshot_get(h, host_buf, as_stream(stream)); }
int cge_manufacturing_snapshot_set(cge_manufacturing *h, const void *host_buf, void *stream) { return snapshot_set(h, host_buf, as_stream(stream)); }
size_t cge_manufacturing_device_bytes(const cge_manufacturing *h) { return h ? h->device_bytes : 0; }
int cge_manufacturing_episode_stats(cge_manufacturing *h, double *return_out, int32_t *length_out) {
    if (!h) return CGE_ERR_INVALID_ARG;
    h->ep_ret = return_out; h->ep_len = length_out;
    return CGE_OK;
}

int cge_manufacturing_done_mask(cge_manufacturing *h, uint8_t *done_out) {
    if (!h) return CGE_ERR_INVALID_ARG;
    h->done_out = done_out;
    return CGE_OK;
}

const char *cge_manufacturing_last_error(const cge_manufacturing *h) { return h ? h->last_error.c_str() : "null handle"; }

const char *cge_manufacturing_last_kernel(const cge_manufacturing *h) { return h ? h->last_kernel.c_str() : ""; }

}  // extern "C"
