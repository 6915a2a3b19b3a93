// hospital.hip — batched HospitalManagementEnv for MI355X (gfx950): kernels + C ABI (include/cge_amd.h).
//
// Re-expresses /root/reference/hospital_management_env/hospital_env.py for N independent instances:
//   reset :184-254, _get_observation :256-321 (243 values; the declared space says 295), step :323-369,
//   _process_action :371-464, _generate_patients :466-510, _generate_disease_type :512-525, _process_treatments :527-605,
//   _update_queues :607-649, _update_staff_fatigue :651-667, _update_equipment :669-686, _check_special_events :688-711,
//   _update_department_metrics :713-724, _check_termination :726-742.
//
// Round 4: ONE ENV = THE 4 LANES OF A DPP QUAD, 16 envs per wave.  Rounds 1-3 ran one env per lane: 48 state columns moved through
// 370-407 registers one group at a time, ONE wave per SIMD, every step re-loading and re-storing 672 of the record's 768 bytes,
// half of a wave's life parked on those round trips (step() 0.16 of the roofline).  Now
//   * the ENTITIES are dealt to the quad: lane ql owns doctors / beds / nurses / machines / medicines number 4 s + ql (4 / 10 / 7 /
//     3 / 4 slots) — 43 registers of entity state per lane instead of 192, held in registers for a WHOLE launch: a fused rollout
//     reads and writes the record once, not once per step;
//   * the per-env BOOKKEEPING (clock, counters, the six queue rings' heads / counts, the generator cursor: Misc) and everything
//     that is serial in the reference — the action's draws, the arrival, the queue walks with their death rolls, the special
//     events — runs identically in the four lanes (same values, same branches: the quad never diverges), exactly the code of the
//     one-lane kernel; the lanes' ring-entry and generator writes are the same bytes to the same addresses;
//   * what crosses entities is a quad reduction: free beds / available doctors of a department, the stable patients to
//     discharge and the nurse to reassign are bit masks OR-ed over the quad (bit = entity index, so "first", "next" and "the r-th"
//     are count-trailing-zeros steps every lane takes alike and the OWNER applies); treated / tired / available counts, nurses per
//     department and occupied beds per department are packed counters summed over the quad (DPP quad_perm, cge_device.hpp);
//   * the 15 medicine draws scan an acceptance mask each lane builds a quarter of; the 10 machines read their words at offsets
//     that follow from the in-use bits, each lane for its own machines;
//   * a step's observation row is written into an LDS image of the wave's rows by the lanes that own the columns and leaves as
//     whole 16-byte pieces of contiguous kilobytes (stream_image), a terminal row (rare) straight from registers.
// State per env: one array-of-structs record (REC_W dwords): MISC (24, every lane reads it), then the entity tables slot-major
// so that a quad's four entities of a slot are one contiguous piece.  Fatigue / equipment status are genuinely float64.
// Queues: the reference's six deques only ever hold six (department, severity) combinations, each FIFO and sorted by
// arrival: (EMERGENCY,3) (EMERGENCY,4) (EMERGENCY,5) (ICU,5) (WARD,1) (WARD,2).  Each is a power-of-two ring in the env's
// own 3008-slot region, 8 bytes per slot {u32 seq|arrival|insurance_delay, u32 treatment_time}; a department's front is the head with the
// smallest sequence number (= patient id).  That makes everything the reference does by walking the deque O(1) amortised:
// total wait = len*now - sum(arrival); "severity 3 waiting > 30" / "severity 4 waiting > 90" are prefixes tracked by a
// marker; only severity-5 patients waiting > 60 (a death roll each, in deque order) are visited one by one.
// RNG: CPython `random` (the env never seeds it; :186 seeds only the unused gymnasium generator): one MT19937 stream per
// env, ~50-80 words per step through an LDS-parked window (one row per env, read by its four lanes).
// Integer rewards -> exact; float32 obs bit-identical to the reference.
#include <cstring>
#include <vector>

#include "cge_device.hpp"
#include "cge_host.hpp"

namespace cge {
namespace hosp {

constexpr int OBS = 243;
constexpr int BLOCK = 64;
constexpr int EPW = BLOCK / QL;          // envs per wave
constexpr int NDOC = 15, NNUR = 25, NBED = 40, NEQ = 10, NMED = 15;
constexpr int SDOC = 4, SBED = 10, SNUR = 7, SEQ = 3, SMED = 4;    // slots per lane: entity 4 s + ql
// record (dwords): MISC | doctors [slot][lane] {fatigue lo, hi, meta, 0} | beds [slot][lane] | nurse fatigue [slot][lane] lo, hi |
// machine status [slot][lane] lo, hi | medicines [lane] (byte s = medicine 4 s + lane)
constexpr int O_DOC = 24, O_BED = O_DOC + 16 * SDOC, O_NUR = O_BED + 4 * SBED, O_EQ = O_NUR + 8 * SNUR, O_MED = O_EQ + 8 * SEQ, REC_W = O_MED + 4;
static_assert(REC_W == 212 && REC_W % 4 == 0, "848-byte records, 16-byte aligned");
constexpr int RING = 3008;
// Generator window: RW tempered ready words per env in LDS (QuadRing, cge_device.hpp), refilled at the top of every step 16 words
// at a time from words the quad twisted ahead of the cursor.  A step draws ~55-80 words (<= 10 for the arrival, 2-4 per machine, ~1.3
// per medicine, 4-6 for the special events); what a burst needs beyond the ring comes straight from the generator block.
constexpr int RW = 96, DROW = 100;          // row stride 100: 16-byte aligned rows, 8 consecutive envs start in 8 different banks
using Draws = QuadRing<RW>;

// sub-queues: 0 (E,3) 1 (E,4) 2 (E,5) 3 (ICU,5) 4 (WARD,1) 5 (WARD,2)
__host__ __device__ constexpr int q_cap(int k) { return k == 0 ? 512 : k == 1 ? 256 : k == 2 ? 64 : k == 3 ? 128 : 1024; }
__host__ __device__ constexpr int q_off(int k) { return k == 0 ? 0 : k == 1 ? 512 : k == 2 ? 768 : k == 3 ? 832 : k == 4 ? 960 : 1984; }
__host__ __device__ constexpr int q_dept3(int k) { return k <= 2 ? 0 : k == 3 ? 1 : 2; }   // index into the 3 live departments (E, ICU, WARD)
__host__ __device__ constexpr int q_sev(int k) { return k == 0 ? 3 : k == 1 ? 4 : k == 2 ? 5 : k == 3 ? 5 : k == 4 ? 1 : 2; }
static_assert(q_off(5) + q_cap(5) == RING, "ring layout");

struct Params {
    uint32_t *state;        // [n][REC_W]
    uint32_t *mt;
    uint32_t *ring;
    int64_t n, env0;
    uint32_t nwaves, per_xcd;      // waves (16 envs each) and waves per XCD: block b serves chunk (b % 8) * per_xcd + b / 8
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

// ------------------------------------------------------------------ MISC: the env-wide bookkeeping, identical in the four lanes
struct Misc {
    uint32_t time, deaths, outbreak, mass, needs_reset, overflow, treated, episodes, total_wait, next_id, pos, pretw, navail;
    int32_t ep_return;                     // sum of the running episode's (integer) rewards
    double wait[3];
    uint32_t sumarr[3];
    uint32_t q[6];                         // per sub-queue, packed as in the record: ring head (10) | count (11) << 10 | "late" prefix length << 21
    __device__ __forceinline__ uint32_t qh(int k) const { return q[k] & 1023u; }
    __device__ __forceinline__ uint32_t qc(int k) const { return (q[k] >> 10) & 2047u; }
    __device__ __forceinline__ uint32_t ql(int k) const { return q[k] >> 21; }
    // one entry leaves at the head: head + 1 (mod the ring), count - 1, late - 1 if any (hit: 0 / 1, for the run-time-indexed pop)
    __device__ __forceinline__ void q_advance(int K, uint32_t hit) {     // K: a constant after unrolling
        const uint32_t w = q[K];
        q[K] = ((w & ~1023u) | ((w + hit) & (uint32_t)(q_cap(K) - 1))) - (hit << 10) - ((w >> 21) ? hit << 21 : 0u);
    }
    uint32_t ndept[3];                     // 25 x 3-bit nurse departments (10 per dword)

    uint32_t in_use;                       // 10 machine-in-use bits (was a word of the EQ group: every lane toggles its own machines' bits)

    __device__ __forceinline__ void load(const uint32_t *__restrict__ rec) {
        uint32_t r[24];
#pragma unroll
        for (int c = 0; c < 6; ++c) { const uint4 v = reinterpret_cast<const uint4 *>(rec)[c]; r[4 * c] = v.x; r[4 * c + 1] = v.y; r[4 * c + 2] = v.z; r[4 * c + 3] = v.w; }
        in_use = r[23] & 1023u; navail = (r[23] >> 16) & 31u;
        time = r[0] & 4095u; deaths = (r[0] >> 12) & 4095u; outbreak = (r[0] >> 24) & 1u; mass = (r[0] >> 25) & 1u;
        needs_reset = (r[0] >> 26) & 1u; overflow = (r[0] >> 27) & 1u;
        treated = r[1] & 0xFFFFu; episodes = r[1] >> 16; total_wait = r[2];
        next_id = r[3] & 4095u; pos = (r[3] >> 12) & 1023u; pretw = mt_ready_decode((r[3] >> 22) & 31u);      // the twist-ahead stream's ready mark (cge_device.hpp)
#pragma unroll
        for (int d = 0; d < 3; ++d) { wait[d] = mk_double(r[4 + 2 * d], r[5 + 2 * d]); sumarr[d] = r[10 + d]; ndept[d] = r[19 + d]; }
#pragma unroll
        for (int k = 0; k < 6; ++k) q[k] = r[13 + k];
        ep_return = (int32_t)r[22];
    }
    __device__ __forceinline__ void store(uint32_t *__restrict__ rec) const {
        uint32_t r[24];
        r[0] = time | (deaths << 12) | (outbreak << 24) | (mass << 25) | (needs_reset << 26) | (overflow << 27);
        r[1] = (treated & 0xFFFFu) | (episodes << 16); r[2] = total_wait;
        r[3] = next_id | (pos << 12) | ((pretw > pos ? mt_ready_encode(pretw) : 0u) << 22);
#pragma unroll
        for (int d = 0; d < 3; ++d) { r[4 + 2 * d] = d_lo(wait[d]); r[5 + 2 * d] = d_hi(wait[d]); r[10 + d] = sumarr[d]; r[19 + d] = ndept[d]; }
#pragma unroll
        for (int k = 0; k < 6; ++k) r[13 + k] = q[k];
        r[22] = (uint32_t)ep_return; r[23] = in_use | (navail << 16);
#pragma unroll
        for (int c = 0; c < 6; ++c) reinterpret_cast<uint4 *>(rec)[c] = make_uint4(r[4 * c], r[4 * c + 1], r[4 * c + 2], r[4 * c + 3]);
    }
    __device__ __forceinline__ uint32_t qlen(int d3) const { return d3 == 0 ? qc(0) + qc(1) + qc(2) : d3 == 1 ? qc(3) : qc(4) + qc(5); }
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
    if (m.qc(K) >= (uint32_t)q_cap(K) || m.next_id >= 4095u) { m.overflow = 1; return; }     // beyond any episode the dynamics can produce
    const uint32_t p = q_off(K) + ((m.qh(K) + m.qc(K)) & (uint32_t)(q_cap(K) - 1));
    rg.rec[CGE_GX(1, p, RING)] = make_uint2(m.next_id | (arrival << 12) | (ins << 23), ttime);
    m.q[K] += 1u << 10; m.sumarr[q_dept3(K)] += arrival; m.next_id += 1;
}
template <int K>
__device__ __forceinline__ void q_pop(Misc &m, uint32_t arrival) {
    m.q_advance(K, 1u); m.sumarr[q_dept3(K)] -= arrival;
}
// front of a department's deque: the head with the smallest sequence number.  Returns the sub-queue (or -1).
template <int K0, int K1>
__device__ __forceinline__ int q_front(const Misc &m, const Ring &rg, uint32_t &rec, uint32_t &slot, uint32_t &tt) {
    int best = -1;
    uint32_t bseq = 0xFFFFFFFFu;
#pragma unroll
    for (int k = K0; k <= K1; ++k) {
        if (m.qc(k) > 0) {
            const uint32_t p = q_off(k) + m.qh(k);
            const uint2 r2 = rg.rec[CGE_GX(2, p, RING)];
            const uint32_t r = r2.x;
            if (rec_seq(r) < bseq) { bseq = rec_seq(r); best = k; rec = r; slot = p; tt = r2.y; }
        }
    }
    return best;
}
// pop with a run-time sub-queue index: arithmetic on every entry (an if-chain over q_pop<K> gets merged by the compiler
// into m.qh(k) with a run-time k, which would move the whole bookkeeping struct to scratch memory)
__device__ __forceinline__ void q_pop_dyn(Misc &m, int k, uint32_t arrival) {
#pragma unroll
    for (int K = 0; K < 6; ++K) m.q_advance(K, k == K ? 1u : 0u);
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

// The six sub-queues' FRONT entries, fetched in one batch at the top of a step (Heads::fetch) so that the bed assignment — three
// departments, each of which used to start with a dependent ring load, 7.8 of a wave-step's 35 us on the phase clocks — finds them in
// registers.  Kept current by the step itself: an arrival into an empty sub-queue IS its front, a pop or a transfer re-reads the one
// sub-queue, an insurance-delay decrement patches the copy.  An entry of an empty sub-queue is never looked at.
struct Heads {
    uint2 h[6];
    __device__ __forceinline__ void fetch(const Misc &m, const Ring &rg) {
#pragma unroll
        for (int k = 0; k < 6; ++k) h[k] = rg.rec[CGE_GX(3, q_off(k) + m.qh(k), RING)];
    }
    __device__ __forceinline__ void refetch(const Misc &m, const Ring &rg, int k) {
#pragma unroll
        for (int K = 0; K < 6; ++K) if (k == K) h[K] = rg.rec[CGE_GX(4, q_off(K) + m.qh(K), RING)];
    }
};
template <int K>
__device__ __forceinline__ void q_push_h(Misc &m, const Ring &rg, Heads &hd, uint32_t arrival, uint32_t ins, uint32_t ttime) {
    const bool was_empty = m.qc(K) == 0u;
    const uint32_t id = m.next_id;
    q_push<K>(m, rg, arrival, ins, ttime);
    if (was_empty && m.qc(K) == 1u) hd.h[K] = make_uint2(id | (arrival << 12) | (ins << 23), ttime);
}
// front of a department's deque from the cached heads: the non-empty sub-queue whose front has the smallest sequence number
template <int K0, int K1>
__device__ __forceinline__ int q_front_h(const Misc &m, const Heads &hd, uint32_t &rec, uint32_t &slot, uint32_t &tt) {
    int best = -1;
    uint32_t bseq = 0xFFFFFFFFu;
#pragma unroll
    for (int k = K0; k <= K1; ++k) {
        if (m.qc(k) > 0 && rec_seq(hd.h[k].x) < bseq) { bseq = rec_seq(hd.h[k].x); best = k; rec = hd.h[k].x; slot = q_off(k) + m.qh(k); tt = hd.h[k].y; }
    }
    return best;
}
template <int G>
__device__ __forceinline__ int dept_front_h(const Misc &m, const Heads &hd, uint32_t &rec, uint32_t &slot, uint32_t &tt) {
    if (G == 0) return q_front_h<0, 2>(m, hd, rec, slot, tt);
    if (G == 1) return q_front_h<3, 3>(m, hd, rec, slot, tt);
    return q_front_h<4, 5>(m, hd, rec, slot, tt);
}

// ------------------------------------------------------------------ the lane's share of the entities
__device__ __forceinline__ uint32_t doc_dept(uint32_t m) { return (m >> 8) & 7u; }
__device__ __forceinline__ uint32_t doc_busy(uint32_t m) { return m >> 11; }
struct Ent {
    double dfat[SDOC];
    uint32_t dmeta[SDOC];                  // doctor 4 s + ql: x (4) | y (4) << 4 | dept (3) << 8 | busy_until (12) << 11
    uint32_t bed[SBED];                    // bed 4 s + ql: occupied | severity (3) << 1 | arrival (11) << 4 | treatment_time (7) << 15
    double nfat[SNUR];                     // nurse 4 s + ql
    double eqs[SEQ];                       // machine 4 s + ql
    uint32_t med[SMED];                    // medicine 4 s + ql

    __device__ __forceinline__ void load(const uint32_t *__restrict__ rec, uint32_t ql) {
#pragma unroll
        for (int s = 0; s < SDOC; ++s) { const uint4 v = *reinterpret_cast<const uint4 *>(rec + O_DOC + 4 * (4 * s + (int)ql)); dfat[s] = mk_double(v.x, v.y); dmeta[s] = v.z; }
#pragma unroll
        for (int s = 0; s < SBED; ++s) bed[s] = rec[O_BED + 4 * s + (int)ql];
#pragma unroll
        for (int s = 0; s < SNUR; ++s) { const uint2 v = *reinterpret_cast<const uint2 *>(rec + O_NUR + 2 * (4 * s + (int)ql)); nfat[s] = mk_double(v.x, v.y); }
#pragma unroll
        for (int s = 0; s < SEQ; ++s) { const uint2 v = *reinterpret_cast<const uint2 *>(rec + O_EQ + 2 * (4 * s + (int)ql)); eqs[s] = mk_double(v.x, v.y); }
        const uint32_t mw = rec[O_MED + (int)ql];
#pragma unroll
        for (int s = 0; s < SMED; ++s) med[s] = (mw >> (8 * s)) & 255u;
    }
    __device__ __forceinline__ void store(uint32_t *__restrict__ rec, uint32_t ql) const {
#pragma unroll
        for (int s = 0; s < SDOC; ++s) *reinterpret_cast<uint4 *>(rec + O_DOC + 4 * (4 * s + (int)ql)) = make_uint4(d_lo(dfat[s]), d_hi(dfat[s]), dmeta[s], 0u);
#pragma unroll
        for (int s = 0; s < SBED; ++s) rec[O_BED + 4 * s + (int)ql] = bed[s];
#pragma unroll
        for (int s = 0; s < SNUR; ++s) *reinterpret_cast<uint2 *>(rec + O_NUR + 2 * (4 * s + (int)ql)) = make_uint2(d_lo(nfat[s]), d_hi(nfat[s]));
#pragma unroll
        for (int s = 0; s < SEQ; ++s) *reinterpret_cast<uint2 *>(rec + O_EQ + 2 * (4 * s + (int)ql)) = make_uint2(d_lo(eqs[s]), d_hi(eqs[s]));
        rec[O_MED + (int)ql] = med[0] | (med[1] << 8) | (med[2] << 16) | (med[3] << 24);
    }
};
// entity index of slot s in lane ql, and whether it exists (the last slot of doctors / nurses / machines / medicines is partly empty)
__device__ __forceinline__ uint32_t ent(int s, uint32_t ql) { return 4u * (uint32_t)s + ql; }

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
// quirk: a delayed patient goes back to the front and the popped bed / doctor pair is lost for this step.  The free beds and the
// department's available doctors are bit masks OR-ed over the quad (bit = bed / doctor index); every lane walks them alike and the
// lane that owns the bed / the doctor takes the patient.
template <int G>
__device__ __forceinline__ void assign_dept(Misc &m, const Ring &rg, Heads &hd, Ent &e, uint32_t ql, uint32_t now) {
    constexpr int b0 = G == 0 ? 0 : G == 1 ? 8 : 18, nb = G == 0 ? 8 : G == 1 ? 6 : 22;
    constexpr uint32_t dept = G == 0 ? 0u : G == 1 ? 1u : 3u;
    uint32_t fb = 0, fd = 0;
#pragma unroll
    for (int s = 0; s < SBED; ++s) {
        const uint32_t b = ent(s, ql);
        if (4 * s + 3 >= b0 && 4 * s < b0 + nb) fb |= (b >= (uint32_t)b0 && b < (uint32_t)(b0 + nb) && !(e.bed[s] & 1u)) ? 1u << (b - (uint32_t)b0) : 0u;
    }
#pragma unroll
    for (int s = 0; s < SDOC; ++s) {
        const uint32_t k = ent(s, ql);
        fd |= (k < (uint32_t)NDOC && doc_dept(e.dmeta[s]) == dept && doc_busy(e.dmeta[s]) <= now) ? 1u << k : 0u;
    }
    fb = gor(fb); fd = gor(fd);
    int stale = -1;                                              // the sub-queue whose cached front went with the last pop
#pragma unroll 1
    while (fb && fd && m.qlen(G) > 0u) {
        // A pop leaves its sub-queue's cached front stale; the new front is fetched only by a quad that comes round again in the SAME
        // step (the next step fetches all six anyway).  Fetched right after the pop, the load hung on every later use of `hd` — the next
        // department's assignment, whatever it popped — and the three departments were three dependent round trips in most wave-steps.
        if (stale >= 0) { hd.refetch(m, rg, stale); stale = -1; }
        uint32_t rec = 0, slot = 0, tt = 0;
        const int k = dept_front_h<G>(m, hd, rec, slot, tt);
        const uint32_t b = (uint32_t)b0 + (uint32_t)__ffs((int)fb) - 1u, di = (uint32_t)__ffs((int)fd) - 1u;
        fb &= fb - 1u; fd &= fd - 1u;
        if (rec_ins(rec) > 0u) {
            rg.rec[CGE_GX(5, slot, RING)].x = rec - (1u << 23);
#pragma unroll
            for (int K = 0; K < 6; ++K) if (k == K) hd.h[K].x = rec - (1u << 23);
            continue;
        }
        const uint32_t sev = (uint32_t)(k == 0 ? 3 : k == 1 ? 4 : k == 2 ? 5 : k == 3 ? 5 : k == 4 ? 1 : 2);
        const uint32_t word = 1u | (sev << 1) | (rec_arr(rec) << 4) | (tt << 15);
#pragma unroll
        for (int s = 0; s < SBED; ++s) if (b == ent(s, ql)) e.bed[s] = word;
#pragma unroll
        for (int s = 0; s < SDOC; ++s)
            if (di == ent(s, ql)) { e.dmeta[s] = (e.dmeta[s] & 2047u) | ((now + tt / 2u) << 11); e.dfat[s] = dmin(100.0, e.dfat[s] + (double)(sev * 2u)); }
        m.total_wait += now - rec_arr(rec);
        q_pop_dyn(m, k, rec_arr(rec));
        stale = k;
    }
}

// K: a severity-5 sub-queue.  Patients waiting > 60 get a death roll each, in deque order (:620-633).
template <int K>
__device__ __forceinline__ void death_rolls(Misc &m, const Ring &rg, Draws &D, uint32_t now, int32_t &reward) {
    constexpr uint32_t off = q_off(K), msk = q_cap(K) - 1;
    uint32_t j = 0;
#pragma unroll 1
    while (j < m.qc(K)) {
        const uint32_t r = rg.rec[CGE_GX(6, off + ((m.qh(K) + j) & msk), RING)].x;
        if (!(rec_arr(r) + 60u < now)) break;                                   // sorted by arrival: nobody behind has waited longer
        if (D.random53() < 0.1) {
            m.deaths += 1; reward -= 2000;
#pragma unroll 1
            for (uint32_t q = j; q + 1u < m.qc(K); ++q) {                       // close the gap (a handful of entries)
                const uint32_t src = off + ((m.qh(K) + q + 1u) & msk), dst = off + ((m.qh(K) + q) & msk);
                rg.rec[CGE_GX(7, dst, RING)] = rg.rec[CGE_GX(8, src, RING)];
            }
            m.q[K] -= 1u << 10; m.sumarr[q_dept3(K)] -= rec_arr(r);
        } else { reward -= 500; ++j; }
    }
}
// K: (EMERGENCY,3) with thr 30 or (EMERGENCY,4) with thr 90: everybody who has waited longer is a prefix (:634-637)
template <int K>
__device__ __forceinline__ void late_penalty(Misc &m, const Ring &rg, uint32_t now, int32_t &reward) {
    constexpr uint32_t thr = K == 0 ? 30u : 90u;
#pragma unroll 1
    while (m.ql(K) < m.qc(K)) {
        const uint32_t r = rg.rec[CGE_GX(9, q_off(K) + ((m.qh(K) + m.ql(K)) & (uint32_t)(q_cap(K) - 1)), RING)].x;
        if (rec_arr(r) + thr < now) m.q[K] += 1u << 21; else break;
    }
    reward -= (int32_t)m.ql(K) * (K == 0 ? 50 : 100);
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

// ------------------------------------------------------------------ the observation row :256-321
// Every lane writes the columns of the entities it owns (and its share of the env-wide ones) into `row`: the env's row of the wave's
// LDS image, or — a SAME_STEP terminal row, rare — its row of final_obs in global memory.  Doctors [0:45], nurse counts [45:51],
// beds [51:131], queue histogram [131:161], machines [161:171], medicines [171:186], utilisation [186:192] (:713-724), waits
// [192:198], doctor fatigue [198:213], nurse fatigue [213:238], deaths / treated / time / outbreak / mass [238:243].
//
// The reference divides in float64 and stores float32.  50 of a lane's ~66 quotients per row are (small integer) / constant: those
// come from a table of the SAME expression evaluated by the compiler ((float)((double)k / c): IEEE arithmetic, nothing to round
// differently) parked in LDS; a float64 division is ~30 instructions at half rate on this part and the row was a third of the
// step's vector instructions.  The float64 fatigues and waits keep a real division: by a constant, as Markstein's correction step.
constexpr int L20 = 0, L5 = 16, L10 = 24, LOCC = 56, L100 = 184, LUT_N = 448;      // k/20 (16) | k/5 (8) | min(k,10)/10 .. k/10 (32) | k/{8,6,4,22} (4 x 32) | k/100 (256)
struct RowLut { float v[LUT_N]; };
constexpr RowLut make_row_lut() {
    RowLut t{};
    for (int k = 0; k < 16; ++k) t.v[L20 + k] = (float)((double)k / 20.0);
    for (int k = 0; k < 8; ++k) t.v[L5 + k] = (float)((double)k / 5.0);
    for (int k = 0; k < 32; ++k) t.v[L10 + k] = (float)((double)k / 10.0);
    for (int k = 0; k < 32; ++k) {
        t.v[LOCC + k] = (float)((double)k / 8.0); t.v[LOCC + 32 + k] = (float)((double)k / 6.0);
        t.v[LOCC + 64 + k] = (float)((double)k / 4.0); t.v[LOCC + 96 + k] = (float)((double)k / 22.0);
    }
    for (int k = 0; k < 256; ++k) t.v[L100 + k] = (float)((double)k / 100.0);
    return t;
}
__device__ const RowLut g_row_lut = make_row_lut();
__device__ __forceinline__ void park_row_lut(float *__restrict__ lut, uint32_t lane) {      // one workgroup's copy (64 lanes)
#pragma unroll
    for (int j = 0; j < LUT_N / 64; ++j) lut[lane + 64u * j] = g_row_lut.v[lane + 64u * j];
}
// a / B for a float64 a and a constant B, correctly rounded: q = RN(a y) with y = RN(1 / B) is a faithful quotient, the residual
// r = a - B q is exact in an fma, and RN(q + r y) is the correctly rounded quotient (Markstein 1990; the exception, a significand of B
// of all ones, is not 100 or 60).  a is a fatigue or a mean wait: 0 or far above the subnormal range, far below overflow.
template <int B>
__device__ __forceinline__ double div_const(double a) {
    constexpr double b = (double)B, y = 1.0 / b;
    const double q = a * y, r = __builtin_fma(-b, q, a);
    return __builtin_fma(r, y, q);
}
__device__ __forceinline__ void write_row(const Misc &m, const Ent &e, uint32_t ql, int32_t max_steps, const float *__restrict__ lut, float *__restrict__ row) {
    const uint32_t now = m.time;
#pragma unroll
    for (int s = 0; s < SDOC; ++s) {
        const uint32_t k = ent(s, ql);
        if (k < (uint32_t)NDOC) {
            row[3 * k] = lut[L20 + (e.dmeta[s] & 15u)];
            row[3 * k + 1] = lut[L20 + ((e.dmeta[s] >> 4) & 15u)];
            row[3 * k + 2] = doc_busy(e.dmeta[s]) > now ? 1.0f : 0.0f;
            row[198 + k] = (float)div_const<100>(e.dfat[s]);
        }
    }
    uint32_t occ = 0;                                            // occupied beds per bed department, 5 bits each
#pragma unroll
    for (int s = 0; s < SBED; ++s) {
        const uint32_t b = ent(s, ql), w = e.bed[s];
        row[51 + 2 * b] = (w & 1u) ? 1.0f : 0.0f;
        row[52 + 2 * b] = lut[L5 + ((w >> 1) & 7u)];
        occ += (w & 1u) << (5u * (b < 8u ? 0u : b < 14u ? 1u : b < 18u ? 2u : 3u));
    }
    occ = gsum(occ);
    uint32_t nc = 0;                                             // nurses per department, 5 bits each
#pragma unroll
    for (int s = 0; s < SNUR; ++s) {
        const uint32_t k = ent(s, ql);
        if (k < (uint32_t)NNUR) {
            row[213 + k] = (float)div_const<100>(e.nfat[s]);
            nc += 1u << (5u * ((sel<3>(m.ndept, k / 10u) >> (3u * (k % 10u))) & 7u));   // (sel: a run-time index would move Misc to scratch)
        }
    }
    nc = gsum(nc);
#pragma unroll
    for (int s = 0; s < SEQ; ++s) { const uint32_t k = ent(s, ql); if (k < (uint32_t)NEQ) row[161 + k] = (float)e.eqs[s]; }
#pragma unroll
    for (int s = 0; s < SMED; ++s) { const uint32_t k = ent(s, ql); if (k < (uint32_t)NMED) row[171 + k] = lut[L100 + (e.med[s] & 255u)]; }
    // env-wide columns, dealt to the lanes: lane ql takes the entries j = ql (mod 4) of each block
#pragma unroll
    for (int j4 = 0; j4 < 8; ++j4) {                               // queue histogram: department d, severity s + 1 -> min(count / 10, 1)
        const uint32_t c = 4u * (uint32_t)j4 + ql;
        if (c < 30u) {
            const uint32_t d = c / 5u, s1 = c % 5u + 1u;
            uint32_t q = 0;
            if (d == 0u) q = s1 == 3u ? m.qc(0) : s1 == 4u ? m.qc(1) : s1 == 5u ? m.qc(2) : 0u;
            else if (d == 1u) q = s1 == 5u ? m.qc(3) : 0u;
            else if (d == 3u) q = s1 == 1u ? m.qc(4) : s1 == 2u ? m.qc(5) : 0u;
            row[131 + c] = lut[L10 + (q < 10u ? q : 10u)];
        }
    }
#pragma unroll
    for (int j4 = 0; j4 < 2; ++j4) {
        const uint32_t d = 4u * (uint32_t)j4 + ql;
        if (d < 6u) {
            row[45 + d] = lut[L10 + ((nc >> (5u * d)) & 31u)];
            const uint32_t o = d < 4u ? (occ >> (5u * d)) & 31u : 0u;
            row[186 + d] = d < 4u ? lut[LOCC + 32u * d + o] : 0.0f;
            const double w = d == 0u ? m.wait[0] : d == 1u ? m.wait[1] : d == 3u ? m.wait[2] : 0.0, v = div_const<60>(w);
            row[192 + d] = (float)(v < 1.0 ? v : 1.0);
        }
    }
    if (ql == 0u) { row[238] = (float)((double)m.deaths / 10.0); row[242] = m.mass ? 1.0f : 0.0f; }
    if (ql == 1u) row[239] = (float)((double)m.treated / 100.0);
    if (ql == 2u) row[240] = (float)((double)m.time / (double)max_steps);
    if (ql == 3u) row[241] = m.outbreak ? 1.0f : 0.0f;
}

// ------------------------------------------------------------------ reset :184-254
// Draws, in order: 15 doctors x {department, x, y, fatigue}, 25 nurse fatigues, 10 equipment, 15 medicine counts; every lane draws
// them all (the stream is the env's), the owner keeps each.
// Called with every lane of the wave (the two prepare() calls are wave-convergent): ~100 words for the doctors, ~90 for the rest, each
// part with its own ready margin (QuadRing::FAR).
__device__ __forceinline__ void do_reset(const Params &p, bool mine, Misc &m, Ent &e, uint32_t ql, Draws &D) {
    D.prepare(mine);
    if (mine) {
#pragma unroll 1
    for (uint32_t k = 0; k < (uint32_t)NDOC; ++k) {
        const uint32_t dept = D.randbelow(6u, 3);
        const uint32_t x = dept_px(dept) + D.randbelow(dept_sx(dept), bit_length(dept_sx(dept)));
        const uint32_t y = dept_py(dept) + D.randbelow(dept_sy(dept), bit_length(dept_sy(dept)));
        const double f = 0.0 + (30.0 - 0.0) * D.random53();
#pragma unroll
        for (int s = 0; s < SDOC; ++s) if (k == ent(s, ql)) { e.dmeta[s] = x | (y << 4) | (dept << 8); e.dfat[s] = f; }
    }
#pragma unroll
    for (int s = 0; s < SBED; ++s) e.bed[s] = 0;
    }
    D.prepare(mine);
    if (!mine) return;
#pragma unroll 1
    for (uint32_t k = 0; k < (uint32_t)NNUR; ++k) {
        const double f = 0.0 + (30.0 - 0.0) * D.random53();
#pragma unroll
        for (int s = 0; s < SNUR; ++s) if (k == ent(s, ql)) e.nfat[s] = f;
    }
#pragma unroll 1
    for (uint32_t k = 0; k < (uint32_t)NEQ; ++k) {
        const double sv = 0.7 + (1.0 - 0.7) * D.random53();
#pragma unroll
        for (int s = 0; s < SEQ; ++s) if (k == ent(s, ql)) e.eqs[s] = sv;
    }
#pragma unroll 1
    for (uint32_t k = 0; k < (uint32_t)NMED; ++k) {
        const uint32_t v = 50u + D.randbelow(51u, 6);
#pragma unroll
        for (int s = 0; s < SMED; ++s) if (k == ent(s, ql)) e.med[s] = v;
    }
    m.in_use = 0;
    m.time = 0; m.deaths = 0; m.treated = 0; m.total_wait = 0; m.next_id = 0; m.outbreak = 0; m.mass = 0; m.needs_reset = 0; m.navail = NNUR; m.ep_return = 0;
#pragma unroll
    for (int d = 0; d < 3; ++d) { m.wait[d] = 0.0; m.sumarr[d] = 0; }
#pragma unroll
    for (int k = 0; k < 6; ++k) m.q[k] = 0;
    m.ndept[0] = 0; m.ndept[1] = 0; m.ndept[2] = 0;
#pragma unroll
    for (int k = 0; k < NNUR; ++k) m.ndept[k / 10] |= (uint32_t)(k / 4 < 6 ? k / 4 : 5) << (3 * (k % 10));
}

#ifdef CGE_HOSP_TIMING
__device__ unsigned long long g_timing[2048 * 16];
#define TICK(k) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); const unsigned long long now_ = wall_clock64(); \
    if (threadIdx.x == 0 && blockIdx.x < 2048) { g_timing[blockIdx.x * 16 + k] += now_ - t_last; } t_last = now_; } while (0)
#define TICK_DECL unsigned long long t_last = wall_clock64();
#define TICK_RESTART t_last = wall_clock64();
#else
#define TICK(k)
#define TICK_DECL
#define TICK_RESTART
#endif
// ------------------------------------------------------------------ one step of one env, by its quad
struct StepOut {
    int32_t reward;
    uint32_t flags;
};

__device__ __forceinline__ void quad_step(const Params &p, const Ring &rg, int64_t i, uint32_t ql, int32_t action, Misc &m, Ent &e, Draws &D, StepOut &out) {
    int32_t reward = 0;
    // action-dependent values carried to the phase that owns the data
    uint32_t nurse_pick = 0xFFFFFFFFu, doc_pick = 0xFFFFFFFFu, doc_new = 0;
    const int32_t a = (action >= 0 && action <= 34) ? action : -1;
    TICK_DECL
    m.time += 1;
    const uint32_t now = m.time;
    Heads hd;
    hd.fetch(m, rg);                                       // needed by the bed assignment, ~5 us of draws from here
    // ---- _process_action :371-464 (draws first; the effects on the entities are applied by their owners below)
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
        const uint32_t crit = a == 12 ? m.qc(1) + m.qc(2) : a == 13 ? m.qc(3) : 0u;       // severity >= 4 in that department's queue
        reward += 20 * (int32_t)crit;
    } else if (a >= 24 && a <= 29) reward -= 5;
    else if (a == 30) reward -= 50;
    else if (a == 32) {                                                                    // transfer: 3 x popleft where len > 10
        transfer_dept<0>(m, rg, reward); transfer_dept<1>(m, rg, reward); transfer_dept<2>(m, rg, reward);
        hd.fetch(m, rg);
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
            if (sev == 5u) q_push_h<3>(m, rg, hd, now, ins, tt);
            else if (sev == 4u) q_push_h<1>(m, rg, hd, now, ins, tt);
            else if (sev == 3u) q_push_h<0>(m, rg, hd, now, ins, tt);
            else if (sev == 2u) q_push_h<5>(m, rg, hd, now, ins, tt);
            else q_push_h<4>(m, rg, hd, now, ins, tt);
        }
    }
    TICK(0);
    // ---- doctors and beds: deferred action effects, _process_treatments :527-605, doctor fatigue :654-658
    if (a == 30) {
#pragma unroll
        for (int s = 0; s < SDOC; ++s) e.dfat[s] = dmax(0.0, e.dfat[s] - 10.0);
    } else if (a == 33) {
#pragma unroll
        for (int s = 0; s < SDOC; ++s) { const uint32_t b = doc_busy(e.dmeta[s]); e.dmeta[s] = (e.dmeta[s] & 2047u) | ((b > 10u ? b - 10u : 0u) << 11); }
    } else if (doc_pick != 0xFFFFFFFFu) {
#pragma unroll
        for (int s = 0; s < SDOC; ++s) if (doc_pick == ent(s, ql)) e.dmeta[s] = (e.dmeta[s] & ~2047u) | doc_new;
    } else if (a == 31) {                                                                  // discharge up to 3 stable patients, bed order
        uint32_t lo = 0, hi = 0;                                                           // stable patients, bit = bed index
#pragma unroll
        for (int s = 0; s < SBED; ++s) {
            const uint32_t b = ent(s, ql), w = e.bed[s];
            const bool stable = (w & 1u) && ((w >> 1) & 7u) <= 2u;
            if (4 * s + 3 < 32) lo |= stable ? 1u << b : 0u; else hi |= stable ? 1u << (b - 32u) : 0u;
        }
        lo = gor(lo); hi = gor(hi);
        uint32_t discharged = 0;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            if (lo | hi) {
                const uint32_t b = lo ? (uint32_t)__builtin_ctz(lo) : 32u + (uint32_t)__builtin_ctz(hi);
                if (lo) lo &= lo - 1u; else hi &= hi - 1u;
#pragma unroll
                for (int s = 0; s < SBED; ++s) if (b == ent(s, ql)) e.bed[s] = 0;
                discharged += 1;
            }
        }
        reward += 30 * (int32_t)discharged;
    }
    {
        uint32_t got = 0, done_n = 0;                                                        // completed treatments: reward and count, summed over the quad
#pragma unroll
        for (int s = 0; s < SBED; ++s) {
            const uint32_t w = e.bed[s], sev = (w >> 1) & 7u;
            if ((w & 1u) && now - ((w >> 4) & 2047u) >= ((w >> 15) & 127u)) {                   // counted from ARRIVAL (:537)
                e.bed[s] = 0;
                got += sev == 5u ? 1000u : sev == 4u ? 500u : sev == 3u ? 200u : 100u;
                done_n += 1;
            }
        }
        reward += (int32_t)gsum(got);
        m.treated += gsum(done_n);
    }
    TICK(2);
    // queue -> bed assignment per department (EMERGENCY beds 0-7, ICU 8-13, WARD 18-39; SURGERY's queue is always empty)
    assign_dept<0>(m, rg, hd, e, ql, now); assign_dept<1>(m, rg, hd, e, ql, now); assign_dept<2>(m, rg, hd, e, ql, now);
    TICK(3);
    uint32_t tired = 0;
#pragma unroll
    for (int s = 0; s < SDOC; ++s) {                                                        // doctor fatigue, termination count
        if (ent(s, ql) < (uint32_t)NDOC) {
            e.dfat[s] = doc_busy(e.dmeta[s]) > now ? dmin(100.0, e.dfat[s] + 0.5) : dmax(0.0, e.dfat[s] - 0.2);
            tired += e.dfat[s] > 95.0 ? 1u : 0u;
        }
    }
    tired = gsum(tired);
    TICK(4);
    // ---- _update_queues :607-649
    update_queue<0>(m, rg, D, now, reward); update_queue<1>(m, rg, D, now, reward); update_queue<2>(m, rg, D, now, reward);
    const bool term = m.deaths >= 3u || tired == (uint32_t)NDOC;                            // _check_termination :726-742 (utilisation never exceeds 1)
    const bool trunc = m.time >= (uint32_t)p.max_steps;
    const uint32_t flags = (term ? 1u : 0u) | (trunc ? 2u : 0u);
    TICK(5);
    // ---- nurses: action 0-5 / 30, nurse fatigue :660-667
    {
        if (a >= 0 && a <= 5 && nurse_pick != 0xFFFFFFFFu) {                                  // random.choice(available_nurses) :378-381
            uint32_t avail = 0;                                                              // nurses with fatigue < 80, bit = nurse index
#pragma unroll
            for (int s = 0; s < SNUR; ++s) { const uint32_t k = ent(s, ql); avail |= (k < (uint32_t)NNUR && e.nfat[s] < 80.0) ? 1u << k : 0u; }
            avail = gor(avail);
#pragma unroll 1
            for (uint32_t j = 0; j < nurse_pick; ++j) avail &= avail - 1u;                   // the nurse_pick-th available nurse, in nurse order
            const uint32_t k = (uint32_t)__builtin_ctz(avail | 0x80000000u);
            const uint32_t wd = k / 10u, sh = 3u * (k - wd * 10u);
#pragma unroll
            for (int w2 = 0; w2 < 3; ++w2) if (wd == (uint32_t)w2) m.ndept[w2] = (m.ndept[w2] & ~(7u << sh)) | ((uint32_t)a << sh);
        }
        uint32_t navail = 0;
        const uint32_t qlen_e = m.qlen(0), qlen_i = m.qlen(1), qlen_w = m.qlen(2);
#pragma unroll
        for (int s = 0; s < SNUR; ++s) {
            const uint32_t k = ent(s, ql);
            if (k < (uint32_t)NNUR) {
                double f = e.nfat[s];
                if (a == 30) f = dmax(0.0, f - 10.0);
                const uint32_t wd = k / 10u, d = (sel<3>(m.ndept, wd) >> (3u * (k - wd * 10u))) & 7u;
                const uint32_t qs = d == 0u ? qlen_e : d == 1u ? qlen_i : d == 3u ? qlen_w : 0u;
                f = qs > 5u ? dmin(100.0, f + 0.3) : dmax(0.0, f - 0.1);
                navail += f < 80.0 ? 1u : 0u;
                e.nfat[s] = f;
            }
        }
        m.navail = gsum(navail);
    }
    TICK(6);
    // ---- equipment: action 18-29, _update_equipment :669-686
    {
        if (a >= 18 && a <= 23) {
            const uint32_t k = (uint32_t)(a - 18);
            if (!((m.in_use >> k) & 1u)) {
#pragma unroll
                for (int s = 0; s < SEQ; ++s) if (k == ent(s, ql)) e.eqs[s] = dmin(1.0, e.eqs[s] + 0.2);
                reward += 15;
            }
        } else if (a >= 24 && a <= 29) {
            const uint32_t k = (uint32_t)(a - 24);
#pragma unroll
            for (int s = 0; s < SMED; ++s) if (k == ent(s, ql)) e.med[s] = e.med[s] + 20u < 100u ? e.med[s] + 20u : 100u;
        }
        // :669-679.  Machine k draws 4 words if it is in use (failure roll, then the usage toggle) and 2 if not, and only
        // its own toggle changes its bit: every machine's offset in the window is known up front, so each lane reads the words of
        // its own machines at once; the toggles are OR-ed over the quad.
        {
            const uint32_t inuse0 = m.in_use & ((1u << NEQ) - 1u), need = 2u * NEQ + 2u * (uint32_t)__popc(inuse0);
            if (D.has(need + 2u)) {                        // +2: the second pair is read (not used) for idle machines too
                uint32_t tog = 0;
#pragma unroll
                for (int s = 0; s < SEQ; ++s) {
                    const uint32_t k = ent(s, ql);
                    if (k < (uint32_t)NEQ) {
                        const uint32_t off = 2u * k + 2u * (uint32_t)__popc(inuse0 & ((1u << k) - 1u));
                        const bool used = (inuse0 >> k) & 1u;
                        const double u1 = ((D.peek(off) >> 5) * 67108864.0 + (D.peek(off + 1u) >> 6)) / 9007199254740992.0;
                        const double u2 = ((D.peek(off + 2u) >> 5) * 67108864.0 + (D.peek(off + 3u) >> 6)) / 9007199254740992.0;
                        if (used) {
                            e.eqs[s] = dmax(0.0, e.eqs[s] - 0.01);
                            if (u1 < 0.001) e.eqs[s] = 0.0;
                        }
                        if ((used ? u2 : u1) < 0.1) tog |= 1u << k;
                    }
                }
                m.in_use ^= gor(tog);
                D.skip(need);
            } else {
#pragma unroll 1
                for (uint32_t k = 0; k < (uint32_t)NEQ; ++k) {
                    if ((m.in_use >> k) & 1u) {
                        const bool fail = D.random53() < 0.001;
#pragma unroll
                        for (int s = 0; s < SEQ; ++s) if (k == ent(s, ql)) { e.eqs[s] = dmax(0.0, e.eqs[s] - 0.01); if (fail) e.eqs[s] = 0.0; }
                    }
                    if (D.random53() < 0.1) m.in_use ^= 1u << k;
                }
            }
        }
        TICK(11);
        // :681-686.  randint(0, 2) = _randbelow(3): top two bits of a word, 3 rejected.  The 15 accepted words are found by a bit
        // scan over the acceptance mask of the next 32 — each lane looks at 8 of them, the masks are OR-ed over the quad.
        if (m.treated > 0u) {
            bool fast = D.has(32u);
            uint32_t acc = 0, vlo = 0, vhi = 0;
            if (fast) {
#pragma unroll
                for (uint32_t j8 = 0; j8 < 8u; ++j8) {
                    const uint32_t j = 8u * ql + j8, c = D.peek(j) >> 30;
                    acc |= (c < 3u ? 1u : 0u) << j;
                    if (ql < 2u) vlo |= c << (2u * j); else vhi |= c << (2u * (j - 16u));
                }
                acc = gor(acc); vlo = gor(vlo); vhi = gor(vhi);
                fast = __popc(acc) >= NMED;
            }
            if (fast) {
                uint32_t last = 0;
#pragma unroll
                for (int k = 0; k < NMED; ++k) {
                    last = (uint32_t)__builtin_ctz(acc);
                    acc &= acc - 1u;
                    const uint32_t c = (last < 16u ? vlo >> (2u * last) : vhi >> (2u * (last - 16u))) & 3u;
                    if ((uint32_t)(k & 3) == ql) e.med[k >> 2] = e.med[k >> 2] > c ? e.med[k >> 2] - c : 0u;
                }
                D.skip(last + 1u);
            } else {
#pragma unroll
                for (int k = 0; k < NMED; ++k) {
                    const uint32_t c = D.randbelow(3u, 2);
                    if ((uint32_t)(k & 3) == ql) e.med[k >> 2] = e.med[k >> 2] > c ? e.med[k >> 2] - c : 0u;
                }
            }
        }
        TICK(12);
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
    m.ep_return += reward;                                 // every contribution to this step's reward is in by now
    if (flags) {
        m.episodes += 1;
        if (ql == 0u) {
            if (p.ep_ret) p.ep_ret[i] = (double)m.ep_return;  // integer rewards: exact in float64
            if (p.ep_len) p.ep_len[i] = (int32_t)m.time;
        }
        if (p.mode == CGE_AUTORESET_NEXT_STEP) m.needs_reset = 1;
    }
    TICK(8);
    out.reward = reward;
    out.flags = flags;
}

// the rows of the wave's envs, IMG_ROWS at a time through the LDS image (stream_image: whole 16-byte pieces of contiguous kilobytes)
constexpr int IMG_ROWS = 8;            // (registers, not LDS, bound the waves per SIMD: two passes of 8 rows instead of four of 4)
__device__ __forceinline__ void emit_rows(const Misc &m, const Ent &e, uint32_t lane, int32_t max_steps, const float *__restrict__ lut, uint32_t *__restrict__ image, float *__restrict__ block,
                                          uint32_t rows_live) {
    const uint32_t ql = lane & (uint32_t)(QL - 1), g = lane / (uint32_t)QL;
#pragma unroll 1
    for (uint32_t pass = 0; pass * IMG_ROWS < rows_live; ++pass) {
        if (g / IMG_ROWS == pass) write_row(m, e, ql, max_steps, lut, reinterpret_cast<float *>(image) + (g % IMG_ROWS) * OBS);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const uint32_t rows = rows_live - pass * IMG_ROWS < (uint32_t)IMG_ROWS ? rows_live - pass * IMG_ROWS : (uint32_t)IMG_ROWS;
        stream_image<IMG_ROWS * OBS>(image, block + (size_t)pass * IMG_ROWS * OBS, rows * (uint32_t)(OBS * 4), lane, rows_live * QL < 64u ? rows_live * QL : 64u);
        asm volatile("" ::: "memory");
    }
}

#ifndef CGE_HOSP_WAVES
#define CGE_HOSP_WAVES 2          // waves per SIMD asked of the register allocator (A/B knob: tools/build_variant.sh)
#endif
template <bool ROLLOUT>
__global__ __launch_bounds__(BLOCK, CGE_HOSP_WAVES) void step_kernel(Params p) {
    __shared__ __attribute__((aligned(16))) uint32_t lds[EPW * DROW + IMG_ROWS * OBS + LUT_N];
    uint32_t *const image = lds + EPW * DROW;
    float *const lut = reinterpret_cast<float *>(lds + EPW * DROW + IMG_ROWS * OBS);
    const uint32_t lane = threadIdx.x, ql_ = lane & (uint32_t)(QL - 1), g = lane / (uint32_t)QL;
    const uint32_t chunk = (blockIdx.x & 7u) * p.per_xcd + (blockIdx.x >> 3);      // XCD x serves chunks [x per_xcd, (x + 1) per_xcd)
    const int64_t i0 = (int64_t)chunk * EPW, i = i0 + g;
    park_row_lut(lut, lane);                                                      // all 64 lanes, before any quad leaves
    asm volatile("" ::: "memory");
    if (chunk >= p.nwaves || i >= p.n) return;                                    // whole quads leave; no barrier anywhere below
    const uint32_t rows_live = p.n - i0 < EPW ? (uint32_t)(p.n - i0) : (uint32_t)EPW;
    uint32_t *rec = p.state + i * REC_W;
    Misc m;
    Ent e;
    m.load(rec);
    e.load(rec, ql_);
    Draws D;
    D.init(lds + g * DROW, p.mt + i * MT_STRIDE, m.pos, m.pretw, ql_);
    D.prepare();
    const Ring rg{reinterpret_cast<uint2 *>(p.ring) + i * RING};
    const uint64_t key = ROLLOUT ? hash_env_key(p.a_seed, (uint64_t)(p.env0 + i)) : 0;
    double rsum = 0.0;
    int32_t dcount = 0;
    uint32_t fin_used = 0;                                     // terminal rows this wave has delivered to its segment (fused rollouts)
    const int ksteps = ROLLOUT ? p.k_steps : 1;
    TICK_DECL
#pragma unroll 1
    for (int t = 0; t < ksteps; ++t) {
        uint32_t ql = ql_;
        if (ROLLOUT) asm volatile("" : "+v"(ql));
        D.ql = ql;              // opaque per iteration: keeps lane-derived offsets from being hoisted out of the step loop
        const int32_t a = p.actions ? p.actions[(int64_t)t * p.n + i] : (int32_t)hash_action_from_key(key, (uint64_t)(p.t0 + t), 35u, 0u);
        const bool reset_only = p.mode == CGE_AUTORESET_NEXT_STEP && m.needs_reset;
        StepOut o{0, 0u};
        if (!reset_only) quad_step(p, rg, i, ql, a, m, e, D, o);
        TICK_RESTART
        const bool to_final = o.flags != 0u && p.mode == CGE_AUTORESET_SAME_STEP;
        // the terminal row of a SAME_STEP episode end: step() -> row i of final_obs_out; fused rollout -> the next slot of the wave's
        // segment of the compacted side output
        const unsigned long long fin_mask = __ballot(to_final && ql == 0u);
        if (fin_mask) {
            float *dst = nullptr;
            if (!ROLLOUT) {
                if (to_final && p.final_obs) dst = p.final_obs + i * OBS;
            } else {
                if (p.fin.rows) {
                    const uint32_t slot = fin_used + (uint32_t)__popcll(fin_mask & ((1ull << (lane & ~3u)) - 1ull));
                    if (to_final && (int64_t)slot < p.fin.cap) {
                        const int64_t gs = (int64_t)chunk * p.fin.cap + slot;
                        dst = static_cast<float *>(p.fin.rows) + gs * OBS;
                        if (ql == 0u) p.fin.index[gs] = (int64_t)t * p.n + i;
                    }
                }
                fin_used += (uint32_t)__popcll(fin_mask);
            }
            if (dst) write_row(m, e, ql, p.max_steps, lut, dst);
        }
        TICK(13);
        // ---- episode reset: SAME_STEP envs that just finished, NEXT_STEP envs that finished on the previous call
        const bool reset_now = reset_only || to_final;
        if (__ballot(reset_now)) do_reset(p, reset_now, m, e, ql, D);
        if (D.ovf) m.overflow = 1;
        TICK(9);
        // the NEXT step's generator words now, before this step's row stores are issued: the chunk twists and the ring's loads wait
        // (one in-order memory counter per wave) for every store issued before them — here those are a whole step old
        if (ROLLOUT) D.prepare();
        TICK(1);
        if (p.obs) emit_rows(m, e, lane, p.max_steps, lut, image, p.obs + (int64_t)t * p.obs_step_stride + i0 * OBS, rows_live);
        TICK(10);
#ifdef CGE_HOSP_TIMING
        if (threadIdx.x == 0 && blockIdx.x < 2048) g_timing[blockIdx.x * 16 + 15] += 1;
#endif
        if (ql == 0u) {
            if (ROLLOUT) {
                if (p.reward) p.reward[(int64_t)t * p.n + i] = (float)o.reward;
                if (p.terminated) p.terminated[(int64_t)t * p.n + i] = (uint8_t)o.flags;
            } else {
                p.reward[i] = (float)o.reward;
                p.terminated[i] = (uint8_t)(o.flags & 1u);
                p.truncated[i] = (uint8_t)((o.flags >> 1) & 1u);
                if (p.done) p.done[i] = o.flags ? 1 : 0;
            }
        }
        rsum += (double)o.reward;
        dcount += o.flags ? 1 : 0;
    }
    D.finish();
    m.pos = D.c.pos; m.pretw = D.c.pretw;
    if (ql_ == 0u) m.store(rec);
    e.store(rec, ql_);
    if (ROLLOUT && ql_ == 0u) {
        if (p.reward_sum) p.reward_sum[i] = rsum;
        if (p.done_count) p.done_count[i] = dcount;
        if (p.fin.count && g == 0u) p.fin.count[chunk] = (int32_t)fin_used;
    }
}

// what: 0 = reset(mask) + obs, 1 = rewind the generator cursor after seeding, 2 = fresh-handle state.  Quads as in the step kernel.
__global__ __launch_bounds__(BLOCK) void reset_kernel(Params p, int what) {
    __shared__ uint32_t draws[EPW * DROW];
    __shared__ float lut[LUT_N];
    const uint32_t lane = threadIdx.x, ql = lane & (uint32_t)(QL - 1), g = lane / (uint32_t)QL;
    const int64_t i = (int64_t)blockIdx.x * EPW + g;
    park_row_lut(lut, lane);
    asm volatile("" ::: "memory");
    if (i >= p.n) return;
    uint32_t *rec = p.state + i * REC_W;
    Misc m;
    m.load(rec);
    if (what == 1 || what == 2) {
        m.pos = 0; m.pretw = 0;
        if (what == 2) {
            m.navail = NNUR;
            m.ndept[0] = 0; m.ndept[1] = 0; m.ndept[2] = 0;
#pragma unroll
            for (int k = 0; k < NNUR; ++k) m.ndept[k / 10] |= (uint32_t)(k / 4 < 6 ? k / 4 : 5) << (3 * (k % 10));
        }
        if (ql == 0u) m.store(rec);
        return;
    }
    Ent e;
    e.load(rec, ql);
    Draws D;
    D.init(draws + g * DROW, p.mt + i * MT_STRIDE, m.pos, m.pretw, ql);
    const bool mine = !p.mask || p.mask[i];
    do_reset(p, mine, m, e, ql, D);
    if (mine) {
        D.finish();
        m.pos = D.c.pos; m.pretw = D.c.pretw;
        if (ql == 0u) m.store(rec);
        e.store(rec, ql);
    }
    if (p.obs) write_row(m, e, ql, p.max_steps, lut, p.obs + i * OBS);
}

__global__ __launch_bounds__(256) void info_kernel(const uint32_t *__restrict__ state, int64_t n, int field, double *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t *rec = state + i * REC_W;
    Misc m;
    m.load(rec);
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
        case CGE_HOSPITAL_INFO_OCCUPIED_BEDS:
            for (int b = 0; b < NBED; ++b) c += rec[O_BED + b] & 1u;
            v = c; break;
        case CGE_HOSPITAL_INFO_MEDICINE_TOTAL:
            for (int k = 0; k < 4; ++k) { const uint32_t w = rec[O_MED + k]; c += (w & 255u) + ((w >> 8) & 255u) + ((w >> 16) & 255u) + (w >> 24); }
            v = c; break;
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
    uint32_t *state = nullptr;
    uint32_t *mt = nullptr, *ring = nullptr;
    static constexpr uint32_t snap_tag = 5u;
    std::vector<std::pair<void *, size_t>> blobs() const { return {{state, (size_t)hosp::REC_W * 4 * n}, {mt, (size_t)n * MT_STRIDE * 4}, {ring, (size_t)n * hosp::RING * 8}}; }
    uint32_t snap_extra() const { return 0u; }
    void set_snap_extra(uint32_t v) { (void)v; }
    hosp::Params params() const {
        hosp::Params p{};
        p.state = state; p.mt = mt; p.ring = ring; p.n = n; p.env0 = env0; p.mode = cfg.autoreset_mode; p.max_steps = cfg.max_episode_length;
        p.ep_ret = ep_ret; p.ep_len = ep_len; p.done = done_out;
        p.nwaves = (uint32_t)((n + hosp::EPW - 1) / hosp::EPW);
        p.per_xcd = (p.nwaves + 7u) / 8u;
        return p;
    }
    unsigned blocks() const { return (unsigned)((n + hosp::EPW - 1) / hosp::EPW); }          // reset / seed kernels: one wave per 16 envs, in order
    unsigned step_blocks() const { return (unsigned)(((n + hosp::EPW - 1) / hosp::EPW + 7) / 8 * 8); }
    void free_all() { (void)hipFree(state); (void)hipFree(mt); (void)hipFree(ring); }
};

extern "C" {

#ifdef CGE_GUARD
int cge_hospital_debug_guard(unsigned int *out) {      // [count, site, index, limit, block, lane, 0, 0] of the first violation (cge_device.hpp: CGE_GX)
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(cge::g_guard), 8 * sizeof(unsigned int)) == hipSuccess ? 0 : 1;
}
#endif

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
    const size_t sb = (size_t)hosp::REC_W * 4 * N, mb = N * MT_STRIDE * sizeof(uint32_t), rb = N * hosp::RING * 8;
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
    hipLaunchKernelGGL(hosp::step_kernel<false>, dim3(h->step_blocks()), dim3(hosp::BLOCK), 0, as_stream(stream), p);
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
    hipLaunchKernelGGL(hosp::step_kernel<true>, dim3(h->step_blocks()), dim3(hosp::BLOCK), 0, as_stream(stream), p);
    h->last_kernel = "cge::hosp::step_kernel<true>";
    CGE_TRY(h, hipGetLastError());
    return CGE_OK;
}

CGE_DEFINE_FINAL_OBS(hospital, float, hosp::EPW)

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
