/* oracle/orc_hospital.c — CPU restatement of HospitalManagementEnv over a batch of independent envs.
 *
 * TEST INFRASTRUCTURE ONLY (see orc_rng.h).  Follows /root/reference/hospital_management_env/hospital_env.py:
 *   reset :184-254, _get_observation :256-321 (243 values; the declared space says 295), step :323-369,
 *   _process_action :371-464, _generate_patients :466-510, _generate_disease_type :512-525, _process_treatments :527-605,
 *   _update_queues :607-649, _update_staff_fatigue :651-667, _update_equipment :669-686, _check_special_events :688-711,
 *   _update_department_metrics :713-724, _check_termination :726-742.
 * Generator: family P — the process-global CPython `random`; the env never seeds it (:186 seeds only the unused gymnasium
 * generator), so env i owns one MT19937 stream seeded by the caller's random.seed(s_i).
 * The queues are kept as plain arrays of patient records in deque order (this file favours the obvious data structure; the
 * device kernel uses per-severity sub-queues and is checked against this).
 * Parity pins: tests/golden/hospital_{hash,surge}.npz + hospital_kat.json (KAT-H1) — tests/test_oracle_hospital.py.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "orc_rng.h"
#include "orc_epstats.h"

#define HOBS 243
#define NDOC 15
#define NNUR 25
#define NBED 40
#define NEQ 10
#define NMED 15
#define QCAP 2048

typedef struct { int severity, arrival, ttime, insurance; } patient;
typedef struct { patient p[QCAP]; int n; } pqueue;

typedef struct {
    orc_mt P;
    int time, deaths, treated, outbreak, mass, next_id, needs_reset, episodes, overflow;
    int64_t total_wait_time;
    int doc_x[NDOC], doc_y[NDOC], doc_dept[NDOC], doc_busy[NDOC];
    double doc_fat[NDOC], nur_fat[NNUR];
    int nur_dept[NNUR];
    int bed_occ[NBED], bed_sev[NBED], bed_arr[NBED], bed_tt[NBED];
    double eq_status[NEQ];
    int eq_use[NEQ], med[NMED];
    double wait[6], util[6];
    pqueue q[6];
} henv;

typedef struct { int64_t n; int mode, max_steps; henv *e; orc_eps eps; } orc_hospital;

static const int DEPT_POS[6][2] = {{0, 0}, {5, 0}, {9, 0}, {0, 5}, {7, 5}, {10, 5}};      /* department_configs :97-104 */
static const int DEPT_SIZE[6][2] = {{4, 4}, {3, 3}, {4, 2}, {6, 4}, {2, 2}, {3, 3}};
static const int BED_DEPT_FIRST[4] = {0, 8, 14, 18}, BED_DEPT_COUNT[4] = {8, 6, 4, 22};   /* bed_distribution :119-124: depts 0,1,2,3 */
static const int BED_DEPTS[4] = {0, 1, 2, 3};
static const int TREATMENT[6] = {0, 15, 30, 45, 60, 120};                                 /* treatment_times :148-154 by severity */

static int bed_dept(int b) { return b < 8 ? 0 : b < 14 ? 1 : b < 18 ? 2 : 3; }

static void q_push(henv *e, int d, patient p) { if (e->q[d].n < QCAP) e->q[d].p[e->q[d].n++] = p; else e->overflow += 1; }
static void q_remove(pqueue *q, int k) { memmove(&q->p[k], &q->p[k + 1], (size_t)(q->n - k - 1) * sizeof(patient)); q->n -= 1; }

static void env_reset(henv *e) {                                                          /* :184-254 */
    e->time = 0; e->deaths = 0; e->treated = 0; e->total_wait_time = 0;
    for (int i = 0; i < NDOC; ++i) {                                                      /* 6 specialisations, 15 doctors in order */
        int dept = (int)orc_py_randbelow(&e->P, 6);                                       /* random.choice(list(Department)) */
        e->doc_x[i] = DEPT_POS[dept][0] + orc_py_randint(&e->P, 0, DEPT_SIZE[dept][0] - 1);
        e->doc_y[i] = DEPT_POS[dept][1] + orc_py_randint(&e->P, 0, DEPT_SIZE[dept][1] - 1);
        e->doc_dept[i] = dept; e->doc_busy[i] = 0;
        e->doc_fat[i] = orc_py_uniform(&e->P, 0, 30);
    }
    for (int i = 0; i < NNUR; ++i) {
        e->nur_dept[i] = i / 4 < 6 ? i / 4 : 5;
        e->nur_fat[i] = orc_py_uniform(&e->P, 0, 30);
    }
    for (int b = 0; b < NBED; ++b) { e->bed_occ[b] = 0; e->bed_sev[b] = 0; e->bed_arr[b] = 0; e->bed_tt[b] = 0; }
    for (int k = 0; k < NEQ; ++k) { e->eq_status[k] = orc_py_uniform(&e->P, 0.7, 1.0); e->eq_use[k] = 0; }
    for (int k = 0; k < NMED; ++k) e->med[k] = orc_py_randint(&e->P, 50, 100);
    for (int d = 0; d < 6; ++d) { e->q[d].n = 0; e->wait[d] = 0.0; e->util[d] = 0.0; }
    e->next_id = 0; e->outbreak = 0; e->mass = 0; e->needs_reset = 0;
}

static void write_obs(const orc_hospital *h, const henv *e, float *obs) {                 /* :256-321 */
    int k = 0;
    for (int i = 0; i < NDOC; ++i) {
        obs[k++] = (float)((double)e->doc_x[i] / 20); obs[k++] = (float)((double)e->doc_y[i] / 20);
        obs[k++] = e->doc_busy[i] > e->time ? 1.0f : 0.0f;
    }
    int nc[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < NNUR; ++i) nc[e->nur_dept[i]] += 1;
    for (int d = 0; d < 6; ++d) obs[k++] = (float)((double)nc[d] / 10.0);
    for (int b = 0; b < NBED; ++b) { obs[k++] = e->bed_occ[b] ? 1.0f : 0.0f; obs[k++] = (float)((double)e->bed_sev[b] / 5.0); }
    for (int d = 0; d < 6; ++d) {
        int sc[5] = {0, 0, 0, 0, 0};
        for (int j = 0; j < e->q[d].n; ++j) sc[e->q[d].p[j].severity - 1] += 1;
        for (int s = 0; s < 5; ++s) { double v = (double)sc[s] / 10.0; obs[k++] = (float)(v < 1.0 ? v : 1.0); }
    }
    for (int j = 0; j < NEQ; ++j) obs[k++] = (float)e->eq_status[j];
    for (int j = 0; j < NMED; ++j) obs[k++] = (float)((double)e->med[j] / 100.0);
    for (int d = 0; d < 6; ++d) obs[k++] = (float)e->util[d];
    for (int d = 0; d < 6; ++d) { double v = e->wait[d] / 60.0; obs[k++] = (float)(v < 1.0 ? v : 1.0); }
    for (int i = 0; i < NDOC; ++i) obs[k++] = (float)(e->doc_fat[i] / 100.0);
    for (int i = 0; i < NNUR; ++i) obs[k++] = (float)(e->nur_fat[i] / 100.0);
    obs[k++] = (float)((double)e->deaths / 10.0);
    obs[k++] = (float)((double)e->treated / 100.0);
    obs[k++] = (float)((double)e->time / (double)h->max_steps);
    obs[k++] = e->outbreak ? 1.0f : 0.0f;
    obs[k++] = e->mass ? 1.0f : 0.0f;
}

static double dmin(double a, double b) { return a < b ? a : b; }
static double dmax(double a, double b) { return a > b ? a : b; }

static int process_action(henv *e, int action) {                                          /* :371-464 */
    int reward = 0;
    if (action < 0 || action > 34) return 0;                 /* outside Discrete(35): the batched ABI treats it as a no-op */
    if (action <= 5) {
        int avail[NNUR], na = 0;
        for (int i = 0; i < NNUR; ++i) if (e->nur_fat[i] < 80) avail[na++] = i;
        if (na) { e->nur_dept[avail[orc_py_randbelow(&e->P, (uint32_t)na)]] = action; reward += 10; }
    } else if (action <= 11) {
        int i = (int)orc_py_randbelow(&e->P, NDOC), nd = (action - 6) % 6;
        e->doc_dept[i] = nd;
        e->doc_x[i] = DEPT_POS[nd][0] + orc_py_randint(&e->P, 0, DEPT_SIZE[nd][0] - 1);
        e->doc_y[i] = DEPT_POS[nd][1] + orc_py_randint(&e->P, 0, DEPT_SIZE[nd][1] - 1);
        reward += 5;
    } else if (action <= 17) {
        int d = action - 12, c = 0;
        for (int j = 0; j < e->q[d].n; ++j) c += e->q[d].p[j].severity >= 4;
        reward += 20 * c;
    } else if (action <= 23) {
        int k = action - 18;
        if (!e->eq_use[k]) { e->eq_status[k] = dmin(1.0, e->eq_status[k] + 0.2); reward += 15; }
    } else if (action <= 29) {
        int k = action - 24;
        e->med[k] = e->med[k] + 20 < 100 ? e->med[k] + 20 : 100;
        reward -= 5;
    } else if (action == 30) {
        for (int i = 0; i < NDOC; ++i) e->doc_fat[i] = dmax(0, e->doc_fat[i] - 10);
        for (int i = 0; i < NNUR; ++i) e->nur_fat[i] = dmax(0, e->nur_fat[i] - 10);
        reward -= 50;
    } else if (action == 31) {
        int discharged = 0;
        for (int b = 0; b < NBED; ++b)
            if (e->bed_occ[b] && e->bed_sev[b] <= 2) {
                e->bed_occ[b] = 0; e->bed_sev[b] = 0; discharged += 1;
                if (discharged >= 3) break;
            }
        reward += discharged * 30;
    } else if (action == 32) {
        for (int d = 0; d < 6; ++d)
            if (e->q[d].n > 10)
                for (int r = 0; r < 3; ++r) { q_remove(&e->q[d], 0); reward -= 200; }
    } else if (action == 33) {
        e->mass = 1;
        for (int i = 0; i < NDOC; ++i) e->doc_busy[i] = e->doc_busy[i] - 10 > 0 ? e->doc_busy[i] - 10 : 0;
        reward -= 100;
    } else if (action == 34) {
        e->mass = 0;
        reward += 5;
    }
    return reward;
}

static void generate_patients(henv *e) {                                                  /* :466-525 */
    double base = (double)orc_py_randint(&e->P, 20, 35) / 60.0;
    if (e->outbreak) base *= 1.5;
    if (e->mass) base *= 2.0;
    if (orc_mt_double(&e->P) < base) {
        double roll = orc_mt_double(&e->P), cumulative = 0;
        static const double RATE[5] = {0.05, 0.10, 0.20, 0.35, 0.30};                     /* CRITICAL, URGENT, EMERGENCY, STANDARD, MINOR */
        static const int SEV[5] = {5, 4, 3, 2, 1};
        int severity = 1;
        for (int k = 0; k < 5; ++k) { cumulative += RATE[k]; if (roll < cumulative) { severity = SEV[k]; break; } }
        (void)orc_py_randint(&e->P, 1, 90);                                               /* age */
        if (e->outbreak) { if (!(orc_mt_double(&e->P) < 0.6)) (void)orc_py_randbelow(&e->P, 15); }   /* disease type */
        else (void)orc_py_randbelow(&e->P, 15);
        patient p = {severity, e->time, TREATMENT[severity], 0};
        if (orc_mt_double(&e->P) < 0.2) p.insurance = orc_py_randint(&e->P, 10, 30);
        e->next_id += 1;
        q_push(e, severity == 5 ? 1 : (severity >= 3 ? 0 : 3), p);
    }
}

static int process_treatments(henv *e) {                                                  /* :527-605 */
    int reward = 0;
    for (int b = 0; b < NBED; ++b)
        if (e->bed_occ[b] && e->time - e->bed_arr[b] >= e->bed_tt[b]) {
            int s = e->bed_sev[b];
            e->bed_occ[b] = 0; e->bed_sev[b] = 0;
            reward += s == 5 ? 1000 : s == 4 ? 500 : s == 3 ? 200 : 100;
            e->treated += 1;
        }
    for (int g = 0; g < 4; ++g) {
        int d = BED_DEPTS[g];
        int beds[NBED], nb = 0, docs[NDOC], ndoc = 0, ib = 0, id = 0;
        for (int b = BED_DEPT_FIRST[g]; b < BED_DEPT_FIRST[g] + BED_DEPT_COUNT[g]; ++b) if (!e->bed_occ[b]) beds[nb++] = b;
        for (int i = 0; i < NDOC; ++i) if (e->doc_dept[i] == d && e->doc_busy[i] <= e->time) docs[ndoc++] = i;
        while (ib < nb && id < ndoc && e->q[d].n > 0) {
            patient *p = &e->q[d].p[0];
            int b = beds[ib++], i = docs[id++];
            if (p->insurance > 0) { p->insurance -= 1; continue; }                        /* re-queued at the front; the bed/doctor slot is spent */
            e->bed_occ[b] = 1; e->bed_sev[b] = p->severity; e->bed_arr[b] = p->arrival; e->bed_tt[b] = p->ttime;
            e->doc_busy[i] = e->time + p->ttime / 2;
            e->doc_fat[i] = dmin(100, e->doc_fat[i] + p->severity * 2);
            e->total_wait_time += e->time - p->arrival;
            q_remove(&e->q[d], 0);
        }
    }
    (void)bed_dept;
    return reward;
}

static int update_queues(henv *e) {                                                       /* :607-649 */
    int reward = 0;
    for (int d = 0; d < 6; ++d) {
        pqueue *q = &e->q[d];
        int64_t total_wait = 0;
        int dead[QCAP], nd = 0;
        for (int j = 0; j < q->n; ++j) {
            int w = e->time - q->p[j].arrival, s = q->p[j].severity;
            total_wait += w;
            if (s == 5 && w > 60) {
                if (orc_mt_double(&e->P) < 0.1) { e->deaths += 1; dead[nd++] = j; reward -= 2000; }
                else reward -= 500;
            } else if (s == 4 && w > 90) reward -= 100;
            else if (s == 3 && w > 30) reward -= 50;
        }
        for (int k = nd - 1; k >= 0; --k) q_remove(q, dead[k]);
        e->wait[d] = q->n > 0 ? (double)total_wait / (double)q->n : 0.0;
    }
    return reward;
}

static void update_rest(henv *e) {
    for (int i = 0; i < NDOC; ++i)                                                        /* _update_staff_fatigue :651-667 */
        e->doc_fat[i] = e->doc_busy[i] > e->time ? dmin(100, e->doc_fat[i] + 0.5) : dmax(0, e->doc_fat[i] - 0.2);
    for (int i = 0; i < NNUR; ++i)
        e->nur_fat[i] = e->q[e->nur_dept[i]].n > 5 ? dmin(100, e->nur_fat[i] + 0.3) : dmax(0, e->nur_fat[i] - 0.1);
    for (int k = 0; k < NEQ; ++k) {                                                       /* _update_equipment :669-686 */
        if (e->eq_use[k]) {
            e->eq_status[k] = dmax(0, e->eq_status[k] - 0.01);
            if (orc_mt_double(&e->P) < 0.001) e->eq_status[k] = 0;
        }
        if (orc_mt_double(&e->P) < 0.1) e->eq_use[k] = !e->eq_use[k];
    }
    if (e->treated > 0)
        for (int k = 0; k < NMED; ++k) { int c = orc_py_randint(&e->P, 0, 2); e->med[k] = e->med[k] - c > 0 ? e->med[k] - c : 0; }
    if (!e->outbreak) {                                                                   /* _check_special_events :688-711 */
        if (orc_mt_double(&e->P) < 0.001) { e->outbreak = 1; (void)orc_py_randbelow(&e->P, 4); }
    } else if (orc_mt_double(&e->P) < 0.01) e->outbreak = 0;
    if (!e->mass && orc_mt_double(&e->P) < 0.0005) {
        e->mass = 1;
        int n = orc_py_randint(&e->P, 5, 10);
        for (int r = 0; r < n; ++r) {
            patient p;
            p.severity = orc_py_randint(&e->P, 3, 5); p.arrival = e->time; p.ttime = orc_py_randint(&e->P, 45, 120); p.insurance = 0;
            e->next_id += 1;
            q_push(e, 0, p);
        }
    }
    for (int d = 0; d < 6; ++d) e->util[d] = 0.0;                                         /* _update_department_metrics :713-724 */
    for (int g = 0; g < 4; ++g) {
        int occ = 0;
        for (int b = BED_DEPT_FIRST[g]; b < BED_DEPT_FIRST[g] + BED_DEPT_COUNT[g]; ++b) occ += e->bed_occ[b];
        e->util[BED_DEPTS[g]] = (double)occ / (double)BED_DEPT_COUNT[g];
    }
}

/* returns terminated | truncated << 1 */
static int env_step(const orc_hospital *h, henv *e, int action, double *reward_out) {      /* :323-369 */
    e->time += 1;
    int reward = process_action(e, action);
    generate_patients(e);
    reward += process_treatments(e);
    reward += update_queues(e);
    update_rest(e);
    int term = e->deaths >= 3;                                                            /* _check_termination :726-742 */
    double cap = 0;
    for (int d = 0; d < 6; ++d) cap += e->util[d];
    if (cap / 6 > 1.5) term = 1;
    int tired = 0;
    for (int i = 0; i < NDOC; ++i) tired += e->doc_fat[i] > 95;
    if (tired == NDOC) term = 1;
    *reward_out = (double)reward;
    return term | ((e->time >= h->max_steps) << 1);
}

orc_hospital *orc_hospital_create(int64_t n, int mode) {
    if (n <= 0 || mode < 0 || mode > 2) return NULL;
    orc_hospital *h = (orc_hospital *)calloc(1, sizeof(*h));
    h->n = n; h->mode = mode; h->max_steps = 1440;
    h->e = (henv *)calloc((size_t)n, sizeof(henv));
    eps_init(&h->eps, n);
    for (int64_t i = 0; i < n; ++i) orc_py_seed(&h->e[i].P, (uint64_t)i);
    return h;
}
void orc_hospital_destroy(orc_hospital *h) { if (h) { free(h->e); eps_free(&h->eps); free(h); } }
void orc_hospital_seed(orc_hospital *h, const uint64_t *seeds) { for (int64_t i = 0; i < h->n; ++i) orc_py_seed(&h->e[i].P, seeds[i]); }

void orc_hospital_reset(orc_hospital *h, const uint8_t *mask, float *obs) {
    for (int64_t i = 0; i < h->n; ++i) {
        if (!mask || mask[i]) { env_reset(&h->e[i]); eps_clear(&h->eps, i); }
        if (obs) write_obs(h, &h->e[i], obs + i * HOBS);
    }
}

void orc_hospital_step(orc_hospital *h, const int32_t *actions, float *obs, float *reward, double *reward64, uint8_t *terminated,
                       uint8_t *truncated, float *final_obs) {
    for (int64_t i = 0; i < h->n; ++i) {
        henv *e = &h->e[i];
        float *o = obs + i * HOBS;
        if (h->mode == 0 && e->needs_reset) {
            { env_reset(e); eps_clear(&h->eps, i); } write_obs(h, e, o);
            reward[i] = 0.0f; if (reward64) reward64[i] = 0.0; terminated[i] = 0; truncated[i] = 0;
            continue;
        }
        double r;
        int f = env_step(h, e, actions[i], &r);
        eps_add(&h->eps, i, (double)r);
        reward[i] = (float)r; if (reward64) reward64[i] = r;
        terminated[i] = (uint8_t)(f & 1); truncated[i] = (uint8_t)(f >> 1);
        if (f) { e->episodes += 1; eps_done(&h->eps, i); }
        if (f && h->mode == 1) {
            if (final_obs) write_obs(h, e, final_obs + i * HOBS);
            { env_reset(e); eps_clear(&h->eps, i); } write_obs(h, e, o);
        } else {
            write_obs(h, e, o);
            if (f && h->mode == 0) e->needs_reset = 1;
        }
    }
}

void orc_hospital_rollout(orc_hospital *h, int k_steps, uint64_t a_seed, int64_t t0, int64_t env0, float *obs, double *reward_sum,
                          int32_t *done_count) {
    for (int64_t i = 0; i < h->n; ++i) {
        henv *e = &h->e[i];
        double rs = 0.0;
        int dc = 0;
        for (int t = 0; t < k_steps; ++t) {
            if (h->mode == 0 && e->needs_reset) { { env_reset(e); eps_clear(&h->eps, i); } continue; }
            double r;
            int f = env_step(h, e, (int)orc_hash_action(a_seed, (uint64_t)(env0 + i), (uint64_t)(t0 + t), 35, 0), &r);
            eps_add(&h->eps, i, (double)r);
            rs += r;
            if (f) { ++dc; e->episodes += 1; eps_done(&h->eps, i); if (h->mode == 1) { env_reset(e); eps_clear(&h->eps, i); } else if (h->mode == 0) e->needs_reset = 1; }
        }
        if (obs) write_obs(h, e, obs + i * HOBS);
        if (reward_sum) reward_sum[i] = rs;
        if (done_count) done_count[i] = dc;
    }
}

/* float64 fields: 0 deaths 1 patients_treated 2 total_wait_time 3 time 4 outbreak 5 mass_casualty 6 next_patient_id
 *                 7..12 queue length of department 0..5   13 occupied beds 14 medicine total 15 episodes 16 needs_reset 17 overflow */
void orc_hospital_info(const orc_hospital *h, int field, double *out) {
    for (int64_t i = 0; i < h->n; ++i) {
        const henv *e = &h->e[i];
        double v = 0;
        int c = 0;
        if (field >= 7 && field <= 12) v = e->q[field - 7].n;
        else switch (field) {
            case 0: v = e->deaths; break; case 1: v = e->treated; break; case 2: v = (double)e->total_wait_time; break; case 3: v = e->time; break;
            case 4: v = e->outbreak; break; case 5: v = e->mass; break; case 6: v = e->next_id; break;
            case 13: for (int b = 0; b < NBED; ++b) c += e->bed_occ[b]; v = c; break;
            case 14: for (int k = 0; k < NMED; ++k) c += e->med[k]; v = c; break;
            case 15: v = e->episodes; break; case 16: v = e->needs_reset; break; case 17: v = e->overflow; break;
        }
        out[i] = v;
    }
}

/* Time-limit override for the short-horizon parity tests (the reference's limit is a constructor constant /
 * config value; the device ABI takes it in its config struct).  Call before reset(). */
void orc_hospital_set_max_steps(orc_hospital *h, int v) { h->max_steps = v; }

/* return and length of each env's last finished episode (orc_epstats.h) */
void orc_hospital_episode_stats(const orc_hospital *h, double *ret, int32_t *len) { eps_get(&h->eps, h->n, ret, len); }
