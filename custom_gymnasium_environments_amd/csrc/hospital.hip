// hospital.hip — batched HospitalManagementEnv for MI355X (gfx950): kernels + C ABI (include/cge_amd.h).
//
// Re-expresses /root/reference/hospital_management_env/hospital_env.py for N independent instances, one lane per env:
//   reset :184-254, _get_observation :256-321 (243 values; the declared space says 295), step :323-369,
//   _process_action :371-464, _generate_patients :466-510, _generate_disease_type :512-525, _process_treatments :527-605,
//   _update_queues :607-649, _update_staff_fatigue :651-667, _update_equipment :669-686, _check_special_events :688-711,
//   _update_department_metrics :713-724, _check_termination :726-742.
// State per env: 48 uint4 columns (768 B) in five GROUPS that are loaded, used and stored one after the other so that at
// most two of them are in registers at a time: MISC (6 cols: clock, counters, queue bookkeeping, nurse departments),
// DOC (12: float64 fatigue + {x, y, department, busy_until} of 15 doctors), NUR (13: float64 fatigue of 25 nurses),
// BED (10: {occupied, severity, arrival, treatment time} of 40 beds), EQ (7: float64 status of 10 machines, in-use bits,
// 15 medicine counts).  Fatigue / equipment status are genuinely float64 (uniform() starts, +-0.1..0.5 per step, clamps).
// Queues: the reference's six deques only ever hold six (department, severity) combinations, each FIFO and sorted by
// arrival: (EMERGENCY,3) (EMERGENCY,4) (EMERGENCY,5) (ICU,5) (WARD,1) (WARD,2).  Each is a power-of-two ring in the env's
// own 3008-slot region, 8 bytes per slot {u32 seq|arrival|insurance_delay, u32 treatment_time}; a department's front is the head with the
// smallest sequence number (= patient id).  That makes everything the reference does by walking the deque O(1) amortised:
// total wait = len*now - sum(arrival); "severity 3 waiting > 30" / "severity 4 waiting > 90" are prefixes tracked by a
// marker; only severity-5 patients waiting > 60 (a death roll each, in deque order) are visited one by one.
// RNG: CPython `random` (the env never seeds it; :186 seeds only the unused gymnasium generator): one MT19937 stream per
// env, ~50-80 words per step (equipment and medicine loops) through an LDS-parked window.
// Observation (N,243): every lane streams its own row from registers as the groups pass through (emit_cols); integer
// rewards -> exact; float32 obs bit-identical to the reference.
#include <cstring>
#include <vector>

#include "cge_device.hpp"
#include "cge_host.hpp"

namespace cge {
namespace hosp {

constexpr int OBS = 243;
constexpr int BLOCK = 64;
constexpr int COLS = 48;
constexpr int C_MISC = 0, C_DOC = 6, C_NUR = 18, C_BED = 31, C_EQ = 41;
constexpr int NDOC = 15, NNUR = 25, NBED = 40, NEQ = 10, NMED = 15;
constexpr int RING = 3008;
// Generator window: a ring of seven 16-word runs per env parked in LDS (RingDraws, cge_device.hpp), topped up ONCE per step:
// the runs a step consumed completely go back to the block and the runs 112 words ahead take their slots (a step draws
// ~55-80 words: <= 10 for the arrival, 2-4 per machine, ~1.3 per medicine, 4-6 for the special events; after the top-up at
// least 97 are parked).  Round 1 used 16-word windows behind ~20 wave-convergent ensure() points per step: 5-8 refills (a
// flush loop and a memory round trip each) per wave-step.
constexpr int DW = 112, DROW = DW + 1;
constexpr uint32_t STEP_WORDS = 88;         // what a step is guaranteed to find parked after the top-up (<= DW - 15)
using Draws = RingDraws<DW>;                // (RingDraws<DW, true>, the twist-ahead form, measured in round 3: rollout 177-181 -> 184-185 us per
                                            // 131,072-env step, step() unchanged — this kernel is not bound by its generator traffic; not used)

// sub-queues: 0 (E,3) 1 (E,4) 2 (E,5) 3 (ICU,5) 4 (WARD,1) 5 (WARD,2)
__host__ __device__ constexpr int q_cap(int k) { return k == 0 ? 512 : k == 1 ? 256 : k == 2 ? 64 : k == 3 ? 128 : 1024; }
__host__ __device__ constexpr int q_off(int k) { return k == 0 ? 0 : k == 1 ? 512 : k == 2 ? 768 : k == 3 ? 832 : k == 4 ? 960 : 1984; }
__host__ __device__ constexpr int q_dept3(int k) { return k <= 2 ? 0 : k == 3 ? 1 : 2; }   // index into the 3 live departments (E, ICU, WARD)
__host__ __device__ constexpr int q_sev(int k) { return k == 0 ? 3 : k == 1 ? 4 : k == 2 ? 5 : k == 3 ? 5 : k == 4 ? 1 : 2; }
static_assert(q_off(5) + q_cap(5) == RING, "ring layout");

struct Params {
    uint4 *state;
    uint32_t *mt;
    uint32_t *ring;
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
    FinalSeg fin;          // fused rollouts (SAME_STEP): terminal rows compacted per wave (cge_hospital_rollout_final_obs); rows nullable
    const uint8_t *mask;
    double *reward_sum;
    int32_t *done_count;
    double *ep_ret;       // episode statistics (cge_hospital_episode_stats), nullable
    int32_t *ep_len;
    uint8_t *done;        // step(): terminated | truncated (cge_hospital_done_mask), nullable
};

__device__ __forceinline__ double mk_double(uint32_t lo, uint32_t hi) { return __hiloint2double((int)hi, (int)lo); }
__device__ __forceinline__ uint32_t d_lo(double x) { return (uint32_t)__double2loint(x); }
__device__ __forceinline__ uint32_t d_hi(double x) { return (uint32_t)__double2hiint(x); }
__device__ __forceinline__ double dmin(double a, double b) { return a < b ? a : b; }
__device__ __forceinline__ double dmax(double a, double b) { return a > b ? a : b; }
__device__ __forceinline__ int bit_length(uint32_t n) { return 32 - __clz((int)n); }

template <int NC>
__device__ __forceinline__ void load_cols(const uint4 *__restrict__ s, int64_t n, int64_t i, int c0, uint32_t (&r)[NC * 4]) {
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const uint4 v = s[(int64_t)(c0 + c) * n + i];
        r[4 * c] = v.x; r[4 * c + 1] = v.y; r[4 * c + 2] = v.z; r[4 * c + 3] = v.w;
    }
}
template <int NC>
__device__ __forceinline__ void store_cols(uint4 *__restrict__ s, int64_t n, int64_t i, int c0, const uint32_t (&r)[NC * 4]) {
#pragma unroll
    for (int c = 0; c < NC; ++c) s[(int64_t)(c0 + c) * n + i] = make_uint4(r[4 * c], r[4 * c + 1], r[4 * c + 2], r[4 * c + 3]);
}

// ------------------------------------------------------------------ MISC group
struct Misc {
    uint32_t time, deaths, outbreak, mass, needs_reset, overflow, treated, episodes, total_wait, next_id, pos, pretw, navail;
    int32_t ep_return;                     // sum of the running episode's (integer) rewards
    double wait[3];
    uint32_t sumarr[3];
    uint32_t qh[6], qc[6], ql[6];          // ring head, count, "late" prefix length
    uint32_t ndept[3];                     // 25 x 3-bit nurse departments (10 per dword)

    __device__ __forceinline__ void load(const uint4 *s, int64_t n, int64_t i) {
        uint32_t r[24];
        load_cols<6>(s, n, i, C_MISC, r);
        time = r[0] & 4095u; deaths = (r[0] >> 12) & 4095u; outbreak = (r[0] >> 24) & 1u; mass = (r[0] >> 25) & 1u;
        needs_reset = (r[0] >> 26) & 1u; overflow = (r[0] >> 27) & 1u;
        treated = r[1] & 0xFFFFu; episodes = r[1] >> 16; total_wait = r[2];
        next_id = r[3] & 4095u; pos = (r[3] >> 12) & 1023u; pretw = ((r[3] >> 22) & 1u) ? (uint32_t)MT_N : 0u; navail = (r[3] >> 23) & 31u;
#pragma unroll
        for (int d = 0; d < 3; ++d) { wait[d] = mk_double(r[4 + 2 * d], r[5 + 2 * d]); sumarr[d] = r[10 + d]; ndept[d] = r[19 + d]; }
#pragma unroll
        for (int k = 0; k < 6; ++k) { qh[k] = r[13 + k] & 1023u; qc[k] = (r[13 + k] >> 10) & 2047u; ql[k] = r[13 + k] >> 21; }
        ep_return = (int32_t)r[22];
    }
    __device__ __forceinline__ void store(uint4 *s, int64_t n, int64_t i) const {
        uint32_t r[24];
        r[0] = time | (deaths << 12) | (outbreak << 24) | (mass << 25) | (needs_reset << 26) | (overflow << 27);
        r[1] = (treated & 0xFFFFu) | (episodes << 16); r[2] = total_wait;
        r[3] = next_id | (pos << 12) | ((pretw ? 1u : 0u) << 22) | (navail << 23);
#pragma unroll
        for (int d = 0; d < 3; ++d) { r[4 + 2 * d] = d_lo(wait[d]); r[5 + 2 * d] = d_hi(wait[d]); r[10 + d] = sumarr[d]; r[19 + d] = ndept[d]; }
#pragma unroll
        for (int k = 0; k < 6; ++k) r[13 + k] = qh[k] | (qc[k] << 10) | (ql[k] << 21);
        r[22] = (uint32_t)ep_return; r[23] = 0;
        store_cols<6>(s, n, i, C_MISC, r);
    }
    __device__ __forceinline__ uint32_t qlen(int d3) const { return d3 == 0 ? qc[0] + qc[1] + qc[2] : d3 == 1 ? qc[3] : qc[4] + qc[5]; }
    __device__ __forceinline__ uint32_t nurse_dept(int i) const { return (ndept[i / 10] >> (3 * (i % 10))) & 7u; }
};

struct Ring {
    uint2 *rec;               // .x the record below, .y the patient's treatment time: one 8-byte entry, so that the pop that needs both waits once
};
// record: seq (12) | arrival (11) << 12 | insurance_delay (5) << 23
__device__ __forceinline__ uint32_t rec_seq(uint32_t r) { return r & 4095u; }
__device__ __forceinline__ uint32_t rec_arr(uint32_t r) { return (r >> 12) & 2047u; }
__device__ __forceinline__ uint32_t rec_ins(uint32_t r) { return (r >> 23) & 31u; }

template <int K>
__device__ __forceinline__ void q_push(Misc &m, const Ring &rg, uint32_t arrival, uint32_t ins, uint32_t ttime) {
    if (m.qc[K] >= (uint32_t)q_cap(K) || m.next_id >= 4095u) { m.overflow = 1; return; }     // beyond any episode the dynamics can produce
    const uint32_t p = q_off(K) + ((m.qh[K] + m.qc[K]) & (uint32_t)(q_cap(K) - 1));
    rg.rec[p] = make_uint2(m.next_id | (arrival << 12) | (ins << 23), ttime);
    m.qc[K] += 1; m.sumarr[q_dept3(K)] += arrival; m.next_id += 1;
}
template <int K>
__device__ __forceinline__ void q_pop(Misc &m, uint32_t arrival) {
    m.qh[K] = (m.qh[K] + 1u) & (uint32_t)(q_cap(K) - 1);
    m.qc[K] -= 1; m.ql[K] -= m.ql[K] ? 1u : 0u; m.sumarr[q_dept3(K)] -= arrival;
}
// front of a department's deque: the head with the smallest sequence number.  Returns the sub-queue (or -1).
template <int K0, int K1>
__device__ __forceinline__ int q_front(const Misc &m, const Ring &rg, uint32_t &rec, uint32_t &slot, uint32_t &tt) {
    int best = -1;
    uint32_t bseq = 0xFFFFFFFFu;
#pragma unroll
    for (int k = K0; k <= K1; ++k) {
        if (m.qc[k] > 0) {
            const uint32_t p = q_off(k) + m.qh[k];
            const uint2 r2 = rg.rec[p];
            const uint32_t r = r2.x;
            if (rec_seq(r) < bseq) { bseq = rec_seq(r); best = k; rec = r; slot = p; tt = r2.y; }
        }
    }
    return best;
}
// pop with a run-time sub-queue index: arithmetic on every entry (an if-chain over q_pop<K> gets merged by the compiler
// into m.qh[k] with a run-time k, which would move the whole bookkeeping struct to scratch memory)
__device__ __forceinline__ void q_pop_dyn(Misc &m, int k, uint32_t arrival) {
#pragma unroll
    for (int K = 0; K < 6; ++K) {
        const uint32_t hit = k == K ? 1u : 0u;
        m.qh[K] = (m.qh[K] + hit) & (uint32_t)(q_cap(K) - 1);
        m.qc[K] -= hit;
        m.ql[K] -= hit & (m.ql[K] ? 1u : 0u);
    }
    const int d3 = k <= 2 ? 0 : k == 3 ? 1 : 2;
#pragma unroll
    for (int d = 0; d < 3; ++d) m.sumarr[d] -= arrival & (0u - (uint32_t)(d3 == d));
}

// department box (:97-104): position and size; randint(0, size-1) = _randbelow(size) with k = size.bit_length()
__device__ __forceinline__ uint32_t dept_px(uint32_t d) { return d == 0 ? 0u : d == 1 ? 5u : d == 2 ? 9u : d == 3 ? 0u : d == 4 ? 7u : 10u; }
__device__ __forceinline__ uint32_t dept_py(uint32_t d) { return d <= 2 ? 0u : 5u; }
__device__ __forceinline__ uint32_t dept_sx(uint32_t d) { return d == 0 ? 4u : d == 1 ? 3u : d == 2 ? 4u : d == 3 ? 6u : d == 4 ? 2u : 3u; }
__device__ __forceinline__ uint32_t dept_sy(uint32_t d) { return d == 0 ? 4u : d == 1 ? 3u : d == 2 ? 2u : d == 3 ? 4u : d == 4 ? 2u : 3u; }
__device__ __forceinline__ uint32_t treatment_time(uint32_t sev) { return sev == 5 ? 120u : sev == 4 ? 60u : sev == 3 ? 45u : sev == 2 ? 30u : 15u; }   // :148-154

// ------------------------------------------------------------------ groups in registers
struct Doctors {
    double fat[NDOC];
    uint32_t meta[NDOC];                   // x (4) | y (4) << 4 | dept (3) << 8 | busy_until (12) << 11
    __device__ __forceinline__ void load(const uint4 *s, int64_t n, int64_t i) {
        uint32_t r[48];
        load_cols<12>(s, n, i, C_DOC, r);
#pragma unroll
        for (int k = 0; k < NDOC; ++k) { fat[k] = mk_double(r[2 * k], r[2 * k + 1]); meta[k] = r[30 + k]; }
    }
    __device__ __forceinline__ void store(uint4 *s, int64_t n, int64_t i) const {
        uint32_t r[48];
#pragma unroll
        for (int k = 0; k < NDOC; ++k) { r[2 * k] = d_lo(fat[k]); r[2 * k + 1] = d_hi(fat[k]); r[30 + k] = meta[k]; }
        r[45] = 0; r[46] = 0; r[47] = 0;
        store_cols<12>(s, n, i, C_DOC, r);
    }
};
__device__ __forceinline__ uint32_t doc_dept(uint32_t m) { return (m >> 8) & 7u; }
__device__ __forceinline__ uint32_t doc_busy(uint32_t m) { return m >> 11; }

struct Beds {
    uint32_t b[NBED];                      // occupied | severity (3) << 1 | arrival (11) << 4 | treatment_time (7) << 15
    __device__ __forceinline__ void load(const uint4 *s, int64_t n, int64_t i) { load_cols<10>(s, n, i, C_BED, b); }
    __device__ __forceinline__ void store(uint4 *s, int64_t n, int64_t i) const { store_cols<10>(s, n, i, C_BED, b); }
};

struct Equip {
    double status[NEQ];
    uint32_t in_use;
    uint32_t med[NMED];
    __device__ __forceinline__ void load(const uint4 *s, int64_t n, int64_t i) {
        uint32_t r[28];
        load_cols<7>(s, n, i, C_EQ, r);
#pragma unroll
        for (int k = 0; k < NEQ; ++k) status[k] = mk_double(r[2 * k], r[2 * k + 1]);
        in_use = r[20];
#pragma unroll
        for (int k = 0; k < NMED; ++k) med[k] = (r[21 + k / 4] >> (8 * (k % 4))) & 255u;
    }
    __device__ __forceinline__ void store(uint4 *s, int64_t n, int64_t i) const {
        uint32_t r[28];
#pragma unroll
        for (int k = 0; k < NEQ; ++k) { r[2 * k] = d_lo(status[k]); r[2 * k + 1] = d_hi(status[k]); }
        r[20] = in_use; r[21] = 0; r[22] = 0; r[23] = 0; r[24] = 0; r[25] = 0; r[26] = 0; r[27] = 0;
#pragma unroll
        for (int k = 0; k < NMED; ++k) r[21 + k / 4] |= med[k] << (8 * (k % 4));
        store_cols<7>(s, n, i, C_EQ, r);
    }
};


template <int G>
__device__ __forceinline__ int dept_front(const Misc &m, const Ring &rg, uint32_t &rec, uint32_t &slot, uint32_t &tt) {
    if (G == 0) return q_front<0, 2>(m, rg, rec, slot, tt);
    if (G == 1) return q_front<3, 3>(m, rg, rec, slot, tt);
    return q_front<4, 5>(m, rg, rec, slot, tt);
}

template <int G>
__device__ __forceinline__ void transfer_dept(Misc &m, const Ring &rg, int32_t &reward) {             // :443-449
    if (m.qlen(G) > 10u) {
#pragma unroll 1
        for (int r = 0; r < 3; ++r) {
            uint32_t rec = 0, slot = 0, tt = 0;
            const int k = dept_front<G>(m, rg, rec, slot, tt);
            q_pop_dyn(m, k, rec_arr(rec));
            reward -= 200;
        }
    }
}

// G: 0 EMERGENCY (beds 0-7), 1 ICU (8-13), 2 WARD (18-39).  The while loop of :576-603, including the insurance-delay
// quirk: a delayed patient goes back to the front and the popped bed / doctor pair is lost for this step.
template <int G>
__device__ __forceinline__ void assign_dept(Misc &m, const Ring &rg, Doctors &dc, Beds &bd, uint32_t now) {
    constexpr int b0 = G == 0 ? 0 : G == 1 ? 8 : 18, nb = G == 0 ? 8 : G == 1 ? 6 : 22;
    constexpr uint32_t dept = G == 0 ? 0u : G == 1 ? 1u : 3u;
    uint32_t fb = 0, fd = 0;
#pragma unroll
    for (int b = 0; b < nb; ++b) fb |= ((bd.b[b0 + b] & 1u) ^ 1u) << b;
#pragma unroll
    for (int k = 0; k < NDOC; ++k) fd |= (doc_dept(dc.meta[k]) == dept && doc_busy(dc.meta[k]) <= now ? 1u : 0u) << k;
#pragma unroll 1
    while (fb && fd && m.qlen(G) > 0u) {
        uint32_t rec = 0, slot = 0, tt = 0;
        const int k = dept_front<G>(m, rg, rec, slot, tt);
        const uint32_t b = (uint32_t)__ffs((int)fb) - 1u, di = (uint32_t)__ffs((int)fd) - 1u;
        fb &= fb - 1u; fd &= fd - 1u;
        if (rec_ins(rec) > 0u) { rg.rec[slot].x = rec - (1u << 23); continue; }
        const uint32_t sev = (uint32_t)(k == 0 ? 3 : k == 1 ? 4 : k == 2 ? 5 : k == 3 ? 5 : k == 4 ? 1 : 2);
        const uint32_t word = 1u | (sev << 1) | (rec_arr(rec) << 4) | (tt << 15);
#pragma unroll
        for (int j = 0; j < nb; ++j) if (b == (uint32_t)j) bd.b[b0 + j] = word;
#pragma unroll
        for (int j = 0; j < NDOC; ++j)
            if (di == (uint32_t)j) { dc.meta[j] = (dc.meta[j] & 2047u) | ((now + tt / 2u) << 11); dc.fat[j] = dmin(100.0, dc.fat[j] + (double)(sev * 2u)); }
        m.total_wait += now - rec_arr(rec);
        q_pop_dyn(m, k, rec_arr(rec));
    }
}

// K: a severity-5 sub-queue.  Patients waiting > 60 get a death roll each, in deque order (:620-633).
template <int K>
__device__ __forceinline__ void death_rolls(Misc &m, const Ring &rg, Draws &D, uint32_t now, int32_t &reward) {
    constexpr uint32_t off = q_off(K), msk = q_cap(K) - 1;
    uint32_t j = 0;
#pragma unroll 1
    while (j < m.qc[K]) {
        const uint32_t r = rg.rec[off + ((m.qh[K] + j) & msk)].x;
        if (!(rec_arr(r) + 60u < now)) break;                                   // sorted by arrival: nobody behind has waited longer
        if (D.random53() < 0.1) {
            m.deaths += 1; reward -= 2000;
#pragma unroll 1
            for (uint32_t q = j; q + 1u < m.qc[K]; ++q) {                       // close the gap (a handful of entries)
                const uint32_t src = off + ((m.qh[K] + q + 1u) & msk), dst = off + ((m.qh[K] + q) & msk);
                rg.rec[dst] = rg.rec[src];
            }
            m.qc[K] -= 1; m.sumarr[q_dept3(K)] -= rec_arr(r);
        } else { reward -= 500; ++j; }
    }
}
// K: (EMERGENCY,3) with thr 30 or (EMERGENCY,4) with thr 90: everybody who has waited longer is a prefix (:634-637)
template <int K>
__device__ __forceinline__ void late_penalty(Misc &m, const Ring &rg, uint32_t now, int32_t &reward) {
    constexpr uint32_t thr = K == 0 ? 30u : 90u;
#pragma unroll 1
    while (m.ql[K] < m.qc[K]) {
        const uint32_t r = rg.rec[q_off(K) + ((m.qh[K] + m.ql[K]) & (uint32_t)(q_cap(K) - 1))].x;
        if (rec_arr(r) + thr < now) m.ql[K] += 1; else break;
    }
    reward -= (int32_t)m.ql[K] * (K == 0 ? 50 : 100);
}
template <int G>
__device__ __forceinline__ void update_queue(Misc &m, const Ring &rg, Draws &D, uint32_t now, int32_t &reward) {   // :611-647 for one department
    const uint32_t len0 = m.qlen(G);
    const uint32_t total_wait = len0 * now - m.sumarr[G];                       // sum over the deque of (now - arrival), the dead included
    if (G == 0) { late_penalty<0>(m, rg, now, reward); late_penalty<1>(m, rg, now, reward); death_rolls<2>(m, rg, D, now, reward); }
    if (G == 1) death_rolls<3>(m, rg, D, now, reward);
    const uint32_t len1 = m.qlen(G);
    m.wait[G] = len1 > 0u ? (double)total_wait / (double)len1 : 0.0;
}

// ------------------------------------------------------------------ obs row -> HBM
// Every lane streams its OWN 972-byte row straight from registers, 16 columns (four 16-byte stores, cge_device.hpp:
// store_own_row) at a time as the groups pass through: no LDS staging.  (Round 1 staged two column chunks of the wave's 64
// rows in a 33.5 KB LDS tile; that tile capped the kernel at 4 waves per CU — 2,048 waves ran as two rounds of one wave per
// SIMD.)  value(j) must be callable with a constant j: the loops below unroll completely.
template <int COL0, int N, class F>
__device__ __forceinline__ void emit_cols(float *drow, bool mine, F value) {
#pragma unroll
    for (int c = 0; c + 16 <= N; c += 16) {
        float v[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = value(c + j);
        store_own_row<16>(drow, COL0 + c, v, mine);
    }
    if constexpr (N % 16 != 0) {
        constexpr int C = N - N % 16;
        float v[N % 16];
#pragma unroll
        for (int j = 0; j < N % 16; ++j) v[j] = value(C + j);
        store_own_row<N % 16>(drow, COL0 + C, v, mine);
    }
}

__device__ __forceinline__ void emit_doctors_beds(const Doctors &dc, const Beds &bd, uint32_t now, float *drow, bool mine) {   // obs[0:45], [51:131]
    emit_cols<0, 45>(drow, mine, [&](int c) {
        const int k = c / 3, r = c % 3;
        return r == 0 ? (float)((double)(dc.meta[k] & 15u) / 20.0) : r == 1 ? (float)((double)((dc.meta[k] >> 4) & 15u) / 20.0)
                                                                           : (doc_busy(dc.meta[k]) > now ? 1.0f : 0.0f);
    });
    emit_cols<51, 80>(drow, mine, [&](int c) {
        const int b = c / 2;
        return c % 2 == 0 ? ((bd.b[b] & 1u) ? 1.0f : 0.0f) : (float)((double)((bd.b[b] >> 1) & 7u) / 5.0);
    });
}
// everything the row takes from the DOC and BED groups: the columns above, doctor fatigue [198:213], utilisation [186:192] (:713-724)
__device__ __forceinline__ void emit_doctor_side(const Doctors &dc, const Beds &bd, uint32_t now, float *drow, bool mine) {
    emit_doctors_beds(dc, bd, now, drow, mine);
    emit_cols<198, NDOC>(drow, mine, [&](int c) { return (float)(dc.fat[c] / 100.0); });
    uint32_t occ[4] = {0, 0, 0, 0};
#pragma unroll
    for (int b = 0; b < NBED; ++b) occ[b < 8 ? 0 : b < 14 ? 1 : b < 18 ? 2 : 3] += bd.b[b] & 1u;
    emit_cols<186, 6>(drow, mine, [&](int c) {
        return c == 0 ? (float)((double)occ[0] / 8.0) : c == 1 ? (float)((double)occ[1] / 6.0) : c == 2 ? (float)((double)occ[2] / 4.0)
               : c == 3 ? (float)((double)occ[3] / 22.0) : 0.0f;
    });
}
__device__ __forceinline__ void emit_nurse_counts(const uint32_t (&counts)[6], float *drow, bool mine) {                 // obs[45:51]
    if (!mine) return;
#pragma unroll
    for (int d = 0; d < 6; ++d) drow[45 + d] = (float)((double)counts[d] / 10.0);
}
__device__ __forceinline__ void emit_nurses(const float (&nurfat)[NNUR], float *drow, bool mine) {                      // obs[213:238]
    emit_cols<213, NNUR>(drow, mine, [&](int c) { return nurfat[c]; });
}
__device__ __forceinline__ void emit_equipment(const Equip &eq, float *drow, bool mine) {                               // obs[161:186]
    emit_cols<161, NEQ + NMED>(drow, mine, [&](int c) {
        return c < NEQ ? (float)eq.status[c < NEQ ? c : 0] : (float)((double)eq.med[c >= NEQ ? c - NEQ : 0] / 100.0);
    });
}
// queue histogram [131:161], waits [192:198], extras [238:243]
__device__ __forceinline__ void emit_misc(const Misc &m, int32_t max_steps, float *drow, bool mine) {
    emit_cols<131, 30>(drow, mine, [&](int c) {
        const int d = c / 5, s = c % 5 + 1;
        uint32_t q = 0;
        if (d == 0) q = s == 3 ? m.qc[0] : s == 4 ? m.qc[1] : s == 5 ? m.qc[2] : 0u;
        else if (d == 1) q = s == 5 ? m.qc[3] : 0u;
        else if (d == 3) q = s == 1 ? m.qc[4] : s == 2 ? m.qc[5] : 0u;
        const double v = (double)q / 10.0;
        return (float)(v < 1.0 ? v : 1.0);
    });
    emit_cols<192, 6>(drow, mine, [&](int d) {
        const double w = d == 0 ? m.wait[0] : d == 1 ? m.wait[1] : d == 3 ? m.wait[2] : 0.0, v = w / 60.0;
        return (float)(v < 1.0 ? v : 1.0);
    });
    emit_cols<238, 5>(drow, mine, [&](int c) {
        return c == 0 ? (float)((double)m.deaths / 10.0) : c == 1 ? (float)((double)m.treated / 100.0) : c == 2 ? (float)((double)m.time / (double)max_steps)
               : c == 3 ? (m.outbreak ? 1.0f : 0.0f) : (m.mass ? 1.0f : 0.0f);
    });
}

// ------------------------------------------------------------------ reset :184-254
// Draws, in order: 15 doctors x {department, x, y, fatigue}, 25 nurse fatigues, 10 equipment, 15 medicine counts.
// Writes the four big groups straight to memory and, when `drow` is not null, the env's observation row.
__device__ __forceinline__ void do_reset(const Params &p, int64_t i, bool mine, Misc &m, Draws &D, float *drow) {
    if (!mine) return;
    const bool want = drow != nullptr;
    {
        Doctors dc;
#pragma unroll 1
        for (int k = 0; k < NDOC; ++k) {
            const uint32_t dept = D.randbelow(6u, 3);
            const uint32_t x = dept_px(dept) + D.randbelow(dept_sx(dept), bit_length(dept_sx(dept)));
            const uint32_t y = dept_py(dept) + D.randbelow(dept_sy(dept), bit_length(dept_sy(dept)));
            const double f = 0.0 + (30.0 - 0.0) * D.random53();
#pragma unroll
            for (int j = 0; j < NDOC; ++j) if (j == k) { dc.meta[j] = x | (y << 4) | (dept << 8); dc.fat[j] = f; }
        }
        dc.store(p.state, p.n, i);
        Beds bd;
#pragma unroll
        for (int b = 0; b < NBED; ++b) bd.b[b] = 0;
        bd.store(p.state, p.n, i);
        emit_doctor_side(dc, bd, 0u, drow, want);
    }
    {
        uint32_t r[52];
        float nurfat[NNUR];
#pragma unroll 1
        for (int k = 0; k < NNUR; ++k) {
            const double f = 0.0 + (30.0 - 0.0) * D.random53();
#pragma unroll
            for (int j = 0; j < NNUR; ++j) if (j == k) { r[2 * j] = d_lo(f); r[2 * j + 1] = d_hi(f); nurfat[j] = (float)(f / 100.0); }
        }
        r[50] = 0; r[51] = 0;
        store_cols<13>(p.state, p.n, i, C_NUR, r);
        emit_nurses(nurfat, drow, want);
    }
    {
        Equip eq;
#pragma unroll 1
        for (int k = 0; k < NEQ; ++k) {
            const double s = 0.7 + (1.0 - 0.7) * D.random53();
#pragma unroll
            for (int j = 0; j < NEQ; ++j) if (j == k) eq.status[j] = s;
        }
#pragma unroll 1
        for (int k = 0; k < NMED; ++k) {
            const uint32_t v = 50u + D.randbelow(51u, 6);
#pragma unroll
            for (int j = 0; j < NMED; ++j) if (j == k) eq.med[j] = v;
        }
        eq.in_use = 0;
        eq.store(p.state, p.n, i);
        emit_equipment(eq, drow, want);
    }
    // MISC
    m.time = 0; m.deaths = 0; m.treated = 0; m.total_wait = 0; m.next_id = 0; m.outbreak = 0; m.mass = 0; m.needs_reset = 0; m.navail = NNUR; m.ep_return = 0;
#pragma unroll
    for (int d = 0; d < 3; ++d) { m.wait[d] = 0.0; m.sumarr[d] = 0; }
#pragma unroll
    for (int k = 0; k < 6; ++k) { m.qh[k] = 0; m.qc[k] = 0; m.ql[k] = 0; }
    m.ndept[0] = 0; m.ndept[1] = 0; m.ndept[2] = 0;
#pragma unroll
    for (int k = 0; k < NNUR; ++k) m.ndept[k / 10] |= (uint32_t)(k / 4 < 6 ? k / 4 : 5) << (3 * (k % 10));
    const uint32_t counts[6] = {4, 4, 4, 4, 4, 5};                                  // nurse counts :213-220
    emit_nurse_counts(counts, drow, want);
    emit_misc(m, p.max_steps, drow, want);                                          // empty queues, zero waits and counters
}

#ifdef CGE_HOSP_TIMING
__device__ unsigned long long g_timing[2048 * 16];
#define TICK(k) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); const unsigned long long now_ = wall_clock64(); \
    if (threadIdx.x == 0 && blockIdx.x < 2048) { g_timing[blockIdx.x * 16 + k] += now_ - t_last; } t_last = now_; } while (0)
#else
#define TICK(k)
#endif
// ------------------------------------------------------------------ one step for the whole wave
struct StepOut {
    int32_t reward;
    uint32_t flags;
};

// obs_row: this env's row of the obs output (null: no observation wanted)
__device__ __forceinline__ void wave_step(const Params &p, int64_t i, bool live, int32_t action, Misc &m, Draws &D, float *obs_row, StepOut &out,
                                          int64_t t = 0, uint32_t fin_used = 0) {
    const Ring rg{reinterpret_cast<uint2 *>(p.ring) + i * RING};
    const bool reset_only = live && p.mode == CGE_AUTORESET_NEXT_STEP && m.needs_reset;
    const bool run = live && !reset_only;
    int32_t reward = 0;
    uint32_t flags = 0;
    // action-dependent values carried to the phase that owns the data
    uint32_t nurse_pick = 0xFFFFFFFFu, doc_pick = 0xFFFFFFFFu, doc_new = 0;
    const int32_t a = (action >= 0 && action <= 34) ? action : -1;
    uint32_t tired = 0;
    bool done = false;
    float *my_dst = nullptr;                                   // where this step's row goes: obs, or final_obs for a SAME_STEP terminal row
#ifdef CGE_HOSP_TIMING
    unsigned long long t_last = wall_clock64();
#endif
    if (run) {
        m.time += 1;
        const uint32_t now = m.time;
        D.ensure_inline(STEP_WORDS);
        TICK(13);
        // ---- _process_action :371-464 (draws first; effects on DOC / NUR / BED / EQ are applied when those groups are loaded)
        if (a >= 0 && a <= 5) {
            if (m.navail > 0) { nurse_pick = D.randbelow(m.navail, bit_length(m.navail)); reward += 10; }
        } else if (a >= 6 && a <= 11) {
            doc_pick = D.randbelow(15u, 4);
            const uint32_t nd = (uint32_t)(a - 6);
            const uint32_t x = dept_px(nd) + D.randbelow(dept_sx(nd), bit_length(dept_sx(nd)));
            const uint32_t y = dept_py(nd) + D.randbelow(dept_sy(nd), bit_length(dept_sy(nd)));
            doc_new = x | (y << 4) | (nd << 8);
            reward += 5;
        } else if (a >= 12 && a <= 17) {
            const uint32_t crit = a == 12 ? m.qc[1] + m.qc[2] : a == 13 ? m.qc[3] : 0u;       // severity >= 4 in that department's queue
            reward += 20 * (int32_t)crit;
        } else if (a >= 24 && a <= 29) reward -= 5;
        else if (a == 30) reward -= 50;
        else if (a == 32) {                                                                    // transfer: 3 x popleft where len > 10
            transfer_dept<0>(m, rg, reward); transfer_dept<1>(m, rg, reward); transfer_dept<2>(m, rg, reward);
        } else if (a == 33) { m.mass = 1; reward -= 100; }
        else if (a == 34) { m.mass = 0; reward += 5; }
        // ---- _generate_patients :466-525
        {
            double base = (double)(20u + D.randbelow(16u, 5)) / 60.0;
            if (m.outbreak) base *= 1.5;
            if (m.mass) base *= 2.0;
            if (D.random53() < base) {
                const double roll = D.random53();
                double cum = 0.0;
                uint32_t sev = 1;
                cum += 0.05; const bool c5 = roll < cum;
                cum += 0.10; const bool c4 = roll < cum;
                cum += 0.20; const bool c3 = roll < cum;
                cum += 0.35; const bool c2 = roll < cum;
                sev = c5 ? 5u : c4 ? 4u : c3 ? 3u : c2 ? 2u : 1u;                                // the last bucket (MINOR) is also the default
                (void)D.randbelow(90u, 7);                                                      // age
                if (m.outbreak) { if (!(D.random53() < 0.6)) (void)D.randbelow(15u, 4); }       // disease type
                else (void)D.randbelow(15u, 4);
                uint32_t ins = 0;
                if (D.random53() < 0.2) ins = 10u + D.randbelow(21u, 5);
                const uint32_t tt = treatment_time(sev);
                if (sev == 5u) q_push<3>(m, rg, now, ins, tt);
                else if (sev == 4u) q_push<1>(m, rg, now, ins, tt);
                else if (sev == 3u) q_push<0>(m, rg, now, ins, tt);
                else if (sev == 2u) q_push<5>(m, rg, now, ins, tt);
                else q_push<4>(m, rg, now, ins, tt);
            }
        }
        TICK(0);
        // ---- DOC + BED in registers: deferred action effects, _process_treatments :527-605, doctor fatigue :654-658
        Doctors dc;
        Beds bd;
        dc.load(p.state, p.n, i);
        bd.load(p.state, p.n, i);
        TICK(1);
        if (a == 30) {
#pragma unroll
            for (int k = 0; k < NDOC; ++k) dc.fat[k] = dmax(0.0, dc.fat[k] - 10.0);
        } else if (a == 33) {
#pragma unroll
            for (int k = 0; k < NDOC; ++k) { const uint32_t b = doc_busy(dc.meta[k]); dc.meta[k] = (dc.meta[k] & 2047u) | ((b > 10u ? b - 10u : 0u) << 11); }
        } else if (doc_pick != 0xFFFFFFFFu) {
#pragma unroll
            for (int k = 0; k < NDOC; ++k) if (doc_pick == (uint32_t)k) dc.meta[k] = (dc.meta[k] & ~2047u) | doc_new;
        } else if (a == 31) {                                                                  // discharge up to 3 stable patients, bed order
            uint32_t discharged = 0;
#pragma unroll
            for (int b = 0; b < NBED; ++b)
                if (discharged < 3u && (bd.b[b] & 1u) && ((bd.b[b] >> 1) & 7u) <= 2u) { bd.b[b] = 0; discharged += 1; }
            reward += 30 * (int32_t)discharged;
        }
#pragma unroll
        for (int b = 0; b < NBED; ++b) {                                                        // completed treatments
            const uint32_t w = bd.b[b], sev = (w >> 1) & 7u;
            if ((w & 1u) && now - ((w >> 4) & 2047u) >= ((w >> 15) & 127u)) {                   // counted from ARRIVAL (:537)
                bd.b[b] = 0;
                reward += sev == 5u ? 1000 : sev == 4u ? 500 : sev == 3u ? 200 : 100;
                m.treated += 1;
            }
        }
        TICK(2);
        // queue -> bed assignment per department (EMERGENCY beds 0-7, ICU 8-13, WARD 18-39; SURGERY's queue is always empty)
        assign_dept<0>(m, rg, dc, bd, now); assign_dept<1>(m, rg, dc, bd, now); assign_dept<2>(m, rg, dc, bd, now);
        TICK(3);
#pragma unroll
        for (int k = 0; k < NDOC; ++k) {                                                        // doctor fatigue, termination count
            dc.fat[k] = doc_busy(dc.meta[k]) > now ? dmin(100.0, dc.fat[k] + 0.5) : dmax(0.0, dc.fat[k] - 0.2);
            tired += dc.fat[k] > 95.0 ? 1u : 0u;
        }
        dc.store(p.state, p.n, i);
        bd.store(p.state, p.n, i);
        // the columns these two groups own go to the obs row while they are in registers; a SAME_STEP terminal row — which
        // belongs in final_obs, known only after the queues below — gets them again from the stored groups (rare), and its
        // obs row is rewritten by the episode reset anyway
        emit_doctor_side(dc, bd, now, obs_row, obs_row != nullptr);
        TICK(4);
    }
    if (run) {
        const uint32_t now = m.time;
        // ---- _update_queues :607-649
        update_queue<0>(m, rg, D, now, reward); update_queue<1>(m, rg, D, now, reward); update_queue<2>(m, rg, D, now, reward);
        const bool term = m.deaths >= 3u || tired == (uint32_t)NDOC;                            // _check_termination :726-742 (utilisation never exceeds 1)
        const bool trunc = m.time >= (uint32_t)p.max_steps;
        flags = (term ? 1u : 0u) | (trunc ? 2u : 0u);
        done = flags != 0u;
        my_dst = obs_row;
        if (done && p.mode == CGE_AUTORESET_SAME_STEP) {
            // step(): row i of final_obs_out; fused rollout: the next slots of the wave's segment of the compacted side output (the lanes
            // that are in here together rank themselves by a ballot, cge_device.hpp: final_slot)
            if (p.fin.rows) {
                const int64_t gs = final_slot(p.fin, (int64_t)blockIdx.x, fin_used, true, t, i);
                my_dst = gs >= 0 ? static_cast<float *>(p.fin.rows) + gs * OBS : nullptr;
            } else {
                my_dst = p.final_obs ? p.final_obs + i * OBS : nullptr;
            }
            if (my_dst) {
                Doctors dc;
                Beds bd;
                dc.load(p.state, p.n, i);
                bd.load(p.state, p.n, i);
                emit_doctor_side(dc, bd, now, my_dst, true);
            }
        }
    }
    const bool to_final = done && p.mode == CGE_AUTORESET_SAME_STEP;
    TICK(5);
    if (run) {
        const uint32_t now = m.time;
        // ---- NUR: action 0-5 / 30, nurse fatigue :660-667, nurse counts obs[45:51]
        {
            uint32_t r[52];
            load_cols<13>(p.state, p.n, i, C_NUR, r);
            uint32_t seen = 0, navail = 0, counts[6] = {0, 0, 0, 0, 0, 0};
            float nurfat_f[NNUR];
            const uint32_t ql[3] = {m.qlen(0), m.qlen(1), m.qlen(2)};
#pragma unroll
            for (int k = 0; k < NNUR; ++k) {
                double f = mk_double(r[2 * k], r[2 * k + 1]);
                if (a >= 0 && a <= 5) {
                    if (f < 80.0) {                                                             // random.choice(available_nurses) :378-381
                        if (seen == nurse_pick) m.ndept[k / 10] = (m.ndept[k / 10] & ~(7u << (3 * (k % 10)))) | ((uint32_t)a << (3 * (k % 10)));
                        seen += 1;
                    }
                } else if (a == 30) f = dmax(0.0, f - 10.0);
                const uint32_t d = m.nurse_dept(k);
                const uint32_t qs = d == 0u ? ql[0] : d == 1u ? ql[1] : d == 3u ? ql[2] : 0u;
                f = qs > 5u ? dmin(100.0, f + 0.3) : dmax(0.0, f - 0.1);
                navail += f < 80.0 ? 1u : 0u;
#pragma unroll
                for (int dd = 0; dd < 6; ++dd) counts[dd] += d == (uint32_t)dd ? 1u : 0u;
                r[2 * k] = d_lo(f); r[2 * k + 1] = d_hi(f);
                nurfat_f[k] = (float)(f / 100.0);
            }
            m.navail = navail;
            store_cols<13>(p.state, p.n, i, C_NUR, r);
            emit_nurse_counts(counts, my_dst, my_dst != nullptr);
            emit_nurses(nurfat_f, my_dst, my_dst != nullptr);
        }
        TICK(6);
        // ---- EQ: action 18-29, _update_equipment :669-686
        {
            Equip eq;
            eq.load(p.state, p.n, i);
            TICK(10);
            if (a >= 18 && a <= 23) {
#pragma unroll
                for (int k = 0; k < 6; ++k)
                    if (a - 18 == k && !((eq.in_use >> k) & 1u)) { eq.status[k] = dmin(1.0, eq.status[k] + 0.2); reward += 15; }
            } else if (a >= 24 && a <= 29) {
#pragma unroll
                for (int k = 0; k < 6; ++k) if (a - 24 == k) eq.med[k] = eq.med[k] + 20u < 100u ? eq.med[k] + 20u : 100u;
            }
            // :669-679.  Machine k draws 4 words if it is in use (failure roll, then the usage toggle) and 2 if not, and only
            // its own toggle changes its bit: every machine's offset in the window is known up front, so all the words are read
            // at once instead of one serial LDS round trip (and a window check) per draw.
            {
                const uint32_t inuse0 = eq.in_use & ((1u << NEQ) - 1u), need = 2u * NEQ + 2u * (uint32_t)__popc(inuse0);
                if (D.has(need + 2u)) {                        // +2: the second pair is read (not used) for idle machines too
#pragma unroll
                    for (int k = 0; k < NEQ; ++k) {
                        const uint32_t off = 2u * k + 2u * (uint32_t)__popc(inuse0 & ((1u << k) - 1u));
                        const bool used = (inuse0 >> k) & 1u;
                        const double u1 = ((D.peek(off) >> 5) * 67108864.0 + (D.peek(off + 1u) >> 6)) / 9007199254740992.0;
                        const double u2 = ((D.peek(off + 2u) >> 5) * 67108864.0 + (D.peek(off + 3u) >> 6)) / 9007199254740992.0;
                        if (used) {
                            eq.status[k] = dmax(0.0, eq.status[k] - 0.01);
                            if (u1 < 0.001) eq.status[k] = 0.0;
                        }
                        if ((used ? u2 : u1) < 0.1) eq.in_use ^= 1u << k;
                    }
                    D.skip(need);
                } else {
#pragma unroll
                    for (int k = 0; k < NEQ; ++k) {
                        if ((eq.in_use >> k) & 1u) {
                            eq.status[k] = dmax(0.0, eq.status[k] - 0.01);
                            if (D.random53() < 0.001) eq.status[k] = 0.0;
                        }
                        if (D.random53() < 0.1) eq.in_use ^= 1u << k;
                    }
                }
            }
            TICK(11);
            // :681-686.  randint(0, 2) = _randbelow(3): top two bits of a word, 3 rejected.  The 15 accepted words are found
            // by a bit scan over the acceptance mask of the next 32.
            if (m.treated > 0u) {
                bool fast = D.has(32u);
                uint32_t acc = 0;
                unsigned long long vals = 0;
                if (fast) {
#pragma unroll
                    for (uint32_t j = 0; j < 32u; ++j) {
                        const uint32_t c = D.peek(j) >> 30;
                        acc |= (c < 3u ? 1u : 0u) << j;
                        vals |= (unsigned long long)c << (2u * j);
                    }
                    fast = __popc(acc) >= NMED;
                }
                if (fast) {
                    uint32_t last = 0;
#pragma unroll
                    for (int k = 0; k < NMED; ++k) {
                        last = (uint32_t)__builtin_ctz(acc);
                        acc &= acc - 1u;
                        const uint32_t c = (uint32_t)(vals >> (2u * last)) & 3u;
                        eq.med[k] = eq.med[k] > c ? eq.med[k] - c : 0u;
                    }
                    D.skip(last + 1u);
                } else {
#pragma unroll
                    for (int k = 0; k < NMED; ++k) { const uint32_t c = D.randbelow(3u, 2); eq.med[k] = eq.med[k] > c ? eq.med[k] - c : 0u; }
                }
            }
            TICK(12);
            eq.store(p.state, p.n, i);
            emit_equipment(eq, my_dst, my_dst != nullptr);
        }
        TICK(7);
        // ---- _check_special_events :688-711
        if (!m.outbreak) { if (D.random53() < 0.001) { m.outbreak = 1; (void)D.randbelow(4u, 3); } }
        else if (D.random53() < 0.01) m.outbreak = 0;
        if (!m.mass && D.random53() < 0.0005) {
            m.mass = 1;
            const uint32_t cnt = 5u + D.randbelow(6u, 3);
#pragma unroll 1
            for (uint32_t r = 0; r < cnt; ++r) {
                const uint32_t sev = 3u + D.randbelow(3u, 2), tt = 45u + D.randbelow(76u, 7);
                if (sev == 3u) q_push<0>(m, rg, now, 0u, tt); else if (sev == 4u) q_push<1>(m, rg, now, 0u, tt); else q_push<2>(m, rg, now, 0u, tt);
            }
        }
        // ---- the rest of the row: queue histogram [131:161], waits [192:198], extras [238:243]
        emit_misc(m, p.max_steps, my_dst, my_dst != nullptr);
        m.ep_return += reward;                                 // every contribution to this step's reward is in by now
        if (done) {
            m.episodes += 1;
            if (p.ep_ret) p.ep_ret[i] = (double)m.ep_return;  // integer rewards: exact in float64
            if (p.ep_len) p.ep_len[i] = (int32_t)m.time;
            if (p.mode == CGE_AUTORESET_NEXT_STEP) m.needs_reset = 1;
        }
    }
    TICK(8);
    // ---- episode reset: SAME_STEP rows that just finished, NEXT_STEP rows that finished on the previous call
    const bool reset_now = reset_only || (run && to_final);
    if (__ballot(reset_now)) do_reset(p, i, reset_now, m, D, obs_row);
    TICK(9);
#ifdef CGE_HOSP_TIMING
    if (threadIdx.x == 0 && blockIdx.x < 2048) g_timing[blockIdx.x * 16 + 15] += 1;
#endif
    out.reward = reward;
    out.flags = flags;
}

template <bool ROLLOUT>
__global__ __launch_bounds__(BLOCK) void step_kernel(Params p) {
    __shared__ uint32_t draws[64 * DROW];
    const int64_t i0 = (int64_t)blockIdx.x * BLOCK;
    const int64_t i = i0 + threadIdx.x;
    const bool live = i < p.n;
    const int64_t li = live ? i : i0;
    const uint32_t lane = threadIdx.x & 63u;
    Misc m;
    m.load(p.state, p.n, li);
    Draws D(draws + lane * DROW, p.mt + li * MT_STRIDE, m.pos, m.pretw);
    const uint64_t key = ROLLOUT ? hash_env_key(p.a_seed, (uint64_t)(p.env0 + li)) : 0;
    double rsum = 0.0;
    int32_t dcount = 0;
    uint32_t fin_used = 0;                                     // terminal rows this wave has delivered to its segment (fused rollouts)
    const int ksteps = ROLLOUT ? p.k_steps : 1;
#pragma unroll 1
    for (int t = 0; t < ksteps; ++t) {
        const int32_t a = !live ? 0 : p.actions ? p.actions[(int64_t)t * p.n + i] : (int32_t)hash_action_from_key(key, (uint64_t)(p.t0 + t), 35u, 0u);
        StepOut o;
        float *obs_row = (p.obs && live) ? p.obs + (int64_t)t * p.obs_step_stride + i * OBS : nullptr;
        int64_t lt = li;
        if (ROLLOUT) asm volatile("" : "+v"(lt));              // opaque per iteration: keeps the 48 column addresses from being hoisted out of the t loop (+72 VGPRs)
        wave_step(p, lt, live, a, m, D, obs_row, o, t, fin_used);
        if (ROLLOUT && p.mode == CGE_AUTORESET_SAME_STEP) fin_used += (uint32_t)__popcll(__ballot(live && o.flags != 0u));
        if (live) {
            if (ROLLOUT) {
                rsum += (double)o.reward;
                dcount += o.flags ? 1 : 0;
                if (p.reward) p.reward[(int64_t)t * p.n + i] = (float)o.reward;
                if (p.terminated) p.terminated[(int64_t)t * p.n + i] = (uint8_t)o.flags;
            } else {
                p.reward[i] = (float)o.reward;
                p.terminated[i] = (uint8_t)(o.flags & 1u);
                p.truncated[i] = (uint8_t)((o.flags >> 1) & 1u);
                if (p.done) p.done[i] = o.flags ? 1 : 0;
            }
        }
    }
    if (live) {
        D.flush();
        m.pos = D.pos; m.pretw = D.pretw;
        m.store(p.state, p.n, i);
        if (ROLLOUT) {
            if (p.reward_sum) p.reward_sum[i] = rsum;
            if (p.done_count) p.done_count[i] = dcount;
            if (p.fin.count && threadIdx.x == 0) p.fin.count[blockIdx.x] = (int32_t)fin_used;
        }
    }
}

// Current observation of an env that is NOT being reset (reset(mask) must return every row): the same assembly as in
// wave_step, from the stored groups.
__device__ __forceinline__ void observe_current(const Params &p, int64_t i, bool mine, const Misc &m, float *drow) {
    if (!mine) return;
    {
        Doctors dc;
        Beds bd;
        dc.load(p.state, p.n, i);
        bd.load(p.state, p.n, i);
        emit_doctor_side(dc, bd, m.time, drow, true);
    }
    {
        uint32_t r[52];
        float nurfat[NNUR];
        load_cols<13>(p.state, p.n, i, C_NUR, r);
        uint32_t counts[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int k = 0; k < NNUR; ++k) {
            nurfat[k] = (float)(mk_double(r[2 * k], r[2 * k + 1]) / 100.0);
            const uint32_t d = m.nurse_dept(k);
#pragma unroll
            for (int dd = 0; dd < 6; ++dd) counts[dd] += d == (uint32_t)dd ? 1u : 0u;
        }
        emit_nurse_counts(counts, drow, true);
        emit_nurses(nurfat, drow, true);
    }
    Equip eq;
    eq.load(p.state, p.n, i);
    emit_equipment(eq, drow, true);
    emit_misc(m, p.max_steps, drow, true);
}

// what: 0 = reset(mask) + obs, 1 = rewind the generator cursor after seeding, 2 = fresh-handle state
__global__ __launch_bounds__(BLOCK) void reset_kernel(Params p, int what) {
    __shared__ uint32_t draws[64 * DROW];
    const int64_t i0 = (int64_t)blockIdx.x * BLOCK;
    const int64_t i = i0 + threadIdx.x;
    const bool live = i < p.n;
    const int64_t li = live ? i : i0;
    const uint32_t lane = threadIdx.x & 63u;
    Misc m;
    m.load(p.state, p.n, li);
    if (what == 1 || what == 2) {
        if (live) {
            m.pos = 0; m.pretw = 0;
            if (what == 2) {
                m.navail = NNUR;
                m.ndept[0] = 0; m.ndept[1] = 0; m.ndept[2] = 0;
#pragma unroll
                for (int k = 0; k < NNUR; ++k) m.ndept[k / 10] |= (uint32_t)(k / 4 < 6 ? k / 4 : 5) << (3 * (k % 10));
            }
            m.store(p.state, p.n, i);
        }
        return;
    }
    Draws D(draws + lane * DROW, p.mt + li * MT_STRIDE, m.pos, m.pretw);
    const bool mine = live && (!p.mask || p.mask[i]);
    float *drow = (p.obs && live) ? p.obs + i * OBS : nullptr;
    if (drow) observe_current(p, li, !mine, m, drow);
    do_reset(p, li, mine, m, D, drow);
    if (mine) {
        D.flush();
        m.pos = D.pos; m.pretw = D.pretw;
        m.store(p.state, p.n, i);
    }
}

__global__ __launch_bounds__(256) void info_kernel(const uint4 *__restrict__ state, int64_t n, int field, double *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    Misc m;
    m.load(state, n, i);
    double v = 0.0;
    uint32_t c = 0;
    if (field >= CGE_HOSPITAL_INFO_QUEUE0 && field <= CGE_HOSPITAL_INFO_QUEUE0 + 5) {
        const int d = field - CGE_HOSPITAL_INFO_QUEUE0;
        v = d == 0 ? m.qlen(0) : d == 1 ? m.qlen(1) : d == 3 ? m.qlen(2) : 0u;
    } else switch (field) {
        case CGE_HOSPITAL_INFO_DEATHS: v = m.deaths; break;
        case CGE_HOSPITAL_INFO_PATIENTS_TREATED: v = m.treated; break;
        case CGE_HOSPITAL_INFO_TOTAL_WAIT_TIME: v = m.total_wait; break;
        case CGE_HOSPITAL_INFO_TIME: v = m.time; break;
        case CGE_HOSPITAL_INFO_OUTBREAK_ACTIVE: v = m.outbreak; break;
        case CGE_HOSPITAL_INFO_MASS_CASUALTY_EVENT: v = m.mass; break;
        case CGE_HOSPITAL_INFO_NEXT_PATIENT_ID: v = m.next_id; break;
        case CGE_HOSPITAL_INFO_OCCUPIED_BEDS: {
            Beds bd;
            bd.load(state, n, i);
            for (int b = 0; b < NBED; ++b) c += bd.b[b] & 1u;
            v = c; break;
        }
        case CGE_HOSPITAL_INFO_MEDICINE_TOTAL: {
            Equip eq;
            eq.load(state, n, i);
            for (int k = 0; k < NMED; ++k) c += eq.med[k];
            v = c; break;
        }
        case CGE_HOSPITAL_INFO_EPISODES: v = m.episodes; break;
        case CGE_HOSPITAL_INFO_NEEDS_RESET: v = m.needs_reset; break;
        case CGE_HOSPITAL_INFO_OVERFLOW: v = m.overflow; break;
    }
    out[i] = v;
}

}  // namespace hosp
}  // namespace cge

using namespace cge;

struct cge_hospital : HandleBase {
    cge_hospital_config cfg{};
    uint4 *state = nullptr;
    uint32_t *mt = nullptr, *ring = nullptr;
    static constexpr uint32_t snap_tag = 5u;
    std::vector<std::pair<void *, size_t>> blobs() const { return {{state, (size_t)hosp::COLS * n * sizeof(uint4)}, {mt, (size_t)n * MT_STRIDE * 4}, {ring, (size_t)n * hosp::RING * 8}}; }
    uint32_t snap_extra() const { return 0u; }
    void set_snap_extra(uint32_t v) { (void)v; }
    hosp::Params params() const {
        hosp::Params p{};
        p.state = state; p.mt = mt; p.ring = ring; p.n = n; p.env0 = env0; p.mode = cfg.autoreset_mode; p.max_steps = cfg.max_episode_length;
        p.ep_ret = ep_ret; p.ep_len = ep_len; p.done = done_out;
        return p;
    }
    unsigned blocks() const { return (unsigned)((n + hosp::BLOCK - 1) / hosp::BLOCK); }
    void free_all() { (void)hipFree(state); (void)hipFree(mt); (void)hipFree(ring); }
};

extern "C" {

#ifdef CGE_HOSP_TIMING
int cge_hospital_debug_timing(unsigned long long *out, int clear) {
    static unsigned long long all[2048 * 16];
    if (hipMemcpyFromSymbol(all, HIP_SYMBOL(hosp::g_timing), sizeof all) != hipSuccess) return 1;
    for (int k = 0; k < 16; ++k) out[k] = 0;
    for (int b = 0; b < 2048; ++b)
        for (int k = 0; k < 16; ++k) out[k] += all[b * 16 + k];
    if (clear) { memset(all, 0, sizeof all); if (hipMemcpyToSymbol(HIP_SYMBOL(hosp::g_timing), all, sizeof all) != hipSuccess) return 1; }
    return 0;
}
#endif

int cge_hospital_create(const cge_hospital_config *cfg, int64_t n_envs, int device, int64_t env_index0, cge_hospital **out) {
    if (!cfg || !out || n_envs <= 0 || env_index0 < 0) return CGE_ERR_INVALID_ARG;
    *out = nullptr;
    if (cfg->autoreset_mode < 0 || cfg->autoreset_mode > 2 || cfg->max_episode_length < 0 || cfg->max_episode_length > 2000) return CGE_ERR_INVALID_ARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return CGE_ERR_NO_DEVICE;
    cge_hospital *h = new cge_hospital();
    h->cfg = *cfg;
    if (h->cfg.max_episode_length == 0) h->cfg.max_episode_length = 1440;
    h->n = n_envs; h->env0 = env_index0; h->device = device;
    DeviceGuard g(device);
    const size_t N = (size_t)n_envs;
    const size_t sb = (size_t)hosp::COLS * N * sizeof(uint4), mb = N * MT_STRIDE * sizeof(uint32_t), rb = N * hosp::RING * 8;
    hipError_t e;
    if ((e = hipMalloc(&h->state, sb)) != hipSuccess || (e = hipMalloc(&h->mt, mb)) != hipSuccess || (e = hipMalloc(&h->ring, rb)) != hipSuccess ||
        (e = hipMemset(h->state, 0, sb)) != hipSuccess) {
        h->free_all();
        delete h;
        return CGE_ERR_HIP;
    }
    h->device_bytes = sb + mb + rb;
    e = launch_mt_seed(h->mt, MT_STRIDE, n_envs, nullptr, 0, env_index0, 0, nullptr);
    if (e == hipSuccess) {
        hosp::Params p = h->params();
        hipLaunchKernelGGL(hosp::reset_kernel, dim3(h->blocks()), dim3(hosp::BLOCK), 0, nullptr, p, 2);
        e = hipGetLastError();
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

int cge_hospital_destroy(cge_hospital *h) {
    if (!h) return CGE_ERR_INVALID_ARG;
    DeviceGuard g(h->device);
    (void)hipDeviceSynchronize();
    h->free_all();
    delete h;
    return CGE_OK;
}

int cge_hospital_seed(cge_hospital *h, const uint64_t *seeds, uint64_t base_seed, void *stream) {
    if (!h) return CGE_ERR_INVALID_ARG;
    DeviceGuard g(h->device);
    CGE_TRY(h, launch_mt_seed(h->mt, MT_STRIDE, h->n, seeds, base_seed, h->env0, 0, as_stream(stream)));
    hosp::Params p = h->params();
    hipLaunchKernelGGL(hosp::reset_kernel, dim3(h->blocks()), dim3(hosp::BLOCK), 0, as_stream(stream), p, 1);
    CGE_TRY(h, hipGetLastError());
    return CGE_OK;
}

int cge_hospital_reset(cge_hospital *h, const uint8_t *mask, float *obs_out, void *stream) {
    if (!h) return CGE_ERR_INVALID_ARG;
    DeviceGuard g(h->device);
    hosp::Params p = h->params();
    p.mask = mask; p.obs = obs_out;
    hipLaunchKernelGGL(hosp::reset_kernel, dim3(h->blocks()), dim3(hosp::BLOCK), 0, as_stream(stream), p, 0);
    CGE_TRY(h, hipGetLastError());
    return CGE_OK;
}

int cge_hospital_step(cge_hospital *h, const int32_t *actions, float *obs_out, float *reward_out, uint8_t *terminated_out, uint8_t *truncated_out,
                      float *final_obs_out, void *stream) {
    if (!h) return CGE_ERR_INVALID_ARG;
    if (!actions || !obs_out || !reward_out || !terminated_out || !truncated_out)
        return h->fail(CGE_ERR_INVALID_ARG, "cge_hospital_step: null actions/obs/reward/terminated/truncated pointer");
    DeviceGuard g(h->device);
    hosp::Params p = h->params();
    p.actions = actions; p.obs = obs_out; p.reward = reward_out; p.terminated = terminated_out; p.truncated = truncated_out;
    p.final_obs = final_obs_out; p.k_steps = 1;
    hipLaunchKernelGGL(hosp::step_kernel<false>, dim3(h->blocks()), dim3(hosp::BLOCK), 0, as_stream(stream), p);
    h->last_kernel = "cge::hosp::step_kernel<false>";
    CGE_TRY(h, hipGetLastError());
    return CGE_OK;
}

int cge_hospital_rollout(cge_hospital *h, int32_t k_steps, const int32_t *actions, uint64_t action_seed, int64_t t0, float *obs_out,
                         int64_t obs_step_stride, float *reward_traj_out, uint8_t *terminated_traj_out, double *reward_sum_out,
                         int32_t *done_count_out, void *stream) {
    if (!h) return CGE_ERR_INVALID_ARG;
    if (k_steps < 0 || obs_step_stride < 0 || (obs_step_stride != 0 && obs_step_stride < h->n * hosp::OBS))
        return h->fail(CGE_ERR_INVALID_ARG, "cge_hospital_rollout: bad k_steps / obs_step_stride");
    if (k_steps == 0) return CGE_OK;
    DeviceGuard g(h->device);
    hosp::Params p = h->params();
    p.k_steps = k_steps; p.actions = actions; p.a_seed = action_seed; p.t0 = t0; p.obs = obs_out; p.obs_step_stride = obs_step_stride;
    p.reward = reward_traj_out; p.terminated = terminated_traj_out; p.reward_sum = reward_sum_out; p.done_count = done_count_out;
    p.fin = FinalSeg{h->fin_rows, h->fin_index, h->fin_count, h->fin_cap, h->n};
    hipLaunchKernelGGL(hosp::step_kernel<true>, dim3(h->blocks()), dim3(hosp::BLOCK), 0, as_stream(stream), p);
    h->last_kernel = "cge::hosp::step_kernel<true>";
    CGE_TRY(h, hipGetLastError());
    return CGE_OK;
}

CGE_DEFINE_FINAL_OBS(hospital, float, 64)

int cge_hospital_info(cge_hospital *h, int32_t field_id, double *out, void *stream) {
    if (!h || !out || field_id < 0 || field_id > CGE_HOSPITAL_INFO_OVERFLOW) return CGE_ERR_INVALID_ARG;
    DeviceGuard g(h->device);
    hipLaunchKernelGGL(hosp::info_kernel, dim3((unsigned)((h->n + 255) / 256)), dim3(256), 0, as_stream(stream), h->state, h->n, field_id, out);
    CGE_TRY(h, hipGetLastError());
    return CGE_OK;
}

size_t cge_hospital_snapshot_bytes(const cge_hospital *h) { return h ? snapshot_bytes(h) : 0; }
int cge_hospital_snapshot_get(cge_hospital *h, void *host_buf, void *stream) { return snapshot_get(h, host_buf, as_stream(stream)); }
int cge_hospital_snapshot_set(cge_hospital *h, const void *host_buf, void *stream) { return snapshot_set(h, host_buf, as_stream(stream)); }
size_t cge_hospital_device_bytes(const cge_hospital *h) { return h ? h->device_bytes : 0; }
int cge_hospital_episode_stats(cge_hospital *h, double *return_out, int32_t *length_out) {
    if (!h) return CGE_ERR_INVALID_ARG;
    h->ep_ret = return_out; h->ep_len = length_out;
    return CGE_OK;
}

int cge_hospital_done_mask(cge_hospital *h, uint8_t *done_out) {
    if (!h) return CGE_ERR_INVALID_ARG;
    h->done_out = done_out;
    return CGE_OK;
}

const char *cge_hospital_last_error(const cge_hospital *h) { return h ? h->last_error.c_str() : "null handle"; }

const char *cge_hospital_last_kernel(const cge_hospital *h) { return h ? h->last_kernel.c_str() : ""; }

}  // extern "C"
