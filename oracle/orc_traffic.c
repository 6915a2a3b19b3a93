/* oracle/orc_traffic.c — CPU restatement of TrafficManagementEnv over a batch of independent envs.
 *
 * TEST INFRASTRUCTURE ONLY (see orc_rng.h).  Follows /root/reference/traffic_management_env/:
 *   config.py:6-35 (defaults: 5x5 grid, 9 intersections, 50 vehicles, spawn 0.3; phases, 5..30 / 3, 1000 steps, rewards)
 *   environment.py:62-83 constructor: grid_size, num_intersections = min(num_intersections, rows*cols), max_vehicles, spawn_rate
 *                  (the layouts its own scripts use: simple_test.py:71-76 (3,3)/4/20/0.4, USAGE_EXAMPLES.md:32-38 (6,6)/16/80/0.5)
 *   environment.py: reset :141-166, step :168-203, _apply_actions :205-220, _spawn_vehicles :222-249,
 *                   _process_intersections :271-281, _remove_completed_vehicles :283-285,
 *                   _calculate_reward :287-311, _get_observation :313-363
 *   utils.py: TrafficLight.update/_advance_phase :79-97, set_phase :108-118, can_pass :99-106,
 *             Intersection.process_vehicles :141-163, generate_vehicle_route :174-193,
 *             get_neighboring_intersections :196-214, get_direction_between_intersections :230-248,
 *             calculate_traffic_metrics :251-267
 *
 * State is the collapsed form of SURVEY.md section 8a: `_update_vehicles` (environment.py:251-269) never moves a
 * vehicle (every vehicle is created AT its intersection and stays "at an intersection"), so per queue only
 * (length, sum of waiting_time, number of members whose destination is this intersection) is observable, and
 * vehicles that passed without reaching their destination stay in `self.vehicles` forever (only the count matters).
 * Generator: family P = CPython global `random`, one private stream per env, seeded by reset(seed=) (:145-146).
 * Parity pins: tests/golden/traffic_*.npz + traffic_kat.json — checked by tests/test_oracle_traffic.py.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "orc_rng.h"
#include "orc_epstats.h"

#define NI 16                 /* capacity: num_intersections <= 16 (h->ni is the live count) */
#define TOBS_MAX (14 * NI + 4)
enum { NS_GREEN = 0, NS_YELLOW = 1, EW_GREEN = 2, EW_YELLOW = 3 };
enum { NORTH = 0, EAST = 1, SOUTH = 2, WEST = 3 };

typedef struct {
    orc_mt P;
    int phase[NI], timer[NI];
    int qlen[NI][4], qdest[NI][4], qwait[NI][4];
    int passed[NI], total_wait[NI];
    int n_vehicles, timestep, needs_reset, episodes;
    double total_reward;
} traffic_env;

typedef struct {
    int64_t n;
    int mode, max_steps, max_vehicles;
    int rows, cols, ni, obs;                                             /* grid_size, num_intersections, 14 * ni + 4 (:108-121) */
    double spawn_rate;
    traffic_env *e; orc_eps eps;
} orc_traffic;

static void env_reset(traffic_env *e) {                                  /* environment.py:141-166, no draws */
    memset(e->phase, 0, sizeof e->phase); memset(e->timer, 0, sizeof e->timer);   /* TrafficLight(): NS_GREEN, timer 0 */
    memset(e->qlen, 0, sizeof e->qlen); memset(e->qdest, 0, sizeof e->qdest); memset(e->qwait, 0, sizeof e->qwait);
    memset(e->passed, 0, sizeof e->passed); memset(e->total_wait, 0, sizeof e->total_wait);
    e->n_vehicles = 0; e->timestep = 0; e->needs_reset = 0; e->total_reward = 0.0;
}

/* neighbours of `id` on the rows x cols grid in the reference's order N, S, W, E (utils.py:196-214).  The walk is over the
 * whole grid: ids >= num_intersections are visited too (only the first num_intersections cells carry an Intersection) */
static int neighbours(const orc_traffic *h, int id, int *out) {
    int row = id / h->cols, col = id % h->cols, c = 0;
    if (row > 0) out[c++] = (row - 1) * h->cols + col;
    if (row < h->rows - 1) out[c++] = (row + 1) * h->cols + col;
    if (col > 0) out[c++] = row * h->cols + col - 1;
    if (col < h->cols - 1) out[c++] = row * h->cols + col + 1;
    return c;
}

static void spawn(const orc_traffic *h, traffic_env *e) {                /* environment.py:222-249 */
    if (e->n_vehicles >= h->max_vehicles) return;                         /* returns BEFORE drawing */
    if (!(orc_mt_double(&e->P) < h->spawn_rate)) return;
    int start = orc_py_randint(&e->P, 0, h->ni - 1);
    int route_len = orc_py_randint(&e->P, 2, h->ni < 5 ? h->ni : 5);      /* utils.py:181: randint(2, min(5, num_intersections)) */
    int cur = start, first = -1;
    for (int k = 0; k < route_len - 1; ++k) {
        int nb[4];
        int c = neighbours(h, cur, nb);
        if (!c) break;                                                    /* utils.py:186-191 (a 1x1 grid) */
        cur = nb[orc_py_randbelow(&e->P, (uint32_t)c)];                   /* random.choice */
        if (k == 0) first = cur;
    }
    if (first < 0) return;                                                /* :233 `if len(route) > 1` */
    int fr = start / h->cols, fc = start % h->cols, tr = first / h->cols, tc = first % h->cols, dir;   /* utils.py:230-248 */
    if (tr < fr) dir = NORTH; else if (tr > fr) dir = SOUTH; else if (tc < fc) dir = WEST; else dir = EAST;
    e->qlen[start][dir] += 1;
    if (cur == start) e->qdest[start][dir] += 1;                          /* destination == this intersection */
    e->n_vehicles += 1;
}

/* NumPy's pairwise summation for n <= 128 elements (numpy/_core/src/umath/loops_utils.h.src, pairwise_sum): below 8 a plain
 * left-to-right loop starting from 0., otherwise eight running accumulators over whole blocks of 8, combined as
 * ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)), then the remainder left to right */
static double np_pairwise(const double *a, int n) {
    if (n < 8) {
        double res = 0.0;
        for (int i = 0; i < n; ++i) res += a[i];
        return res;
    }
    double r[8];
    for (int j = 0; j < 8; ++j) r[j] = a[j];
    int i;
    for (i = 8; i < n - (n % 8); i += 8)
        for (int j = 0; j < 8; ++j) r[j] += a[i + j];
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) res += a[i];
    return res;
}
static double np_var(const int *q, int n) {                               /* np.var of n ints (population variance) */
    int64_t s = 0;
    for (int i = 0; i < n; ++i) s += q[i];
    double mean = (double)s / (double)n;                                  /* the integer sum is exact in float64 */
    double x[NI];
    for (int i = 0; i < n; ++i) { double d = (double)q[i] - mean; x[i] = d * d; }
    return np_pairwise(x, n) / (double)n;
}

static void write_obs(const orc_traffic *h, const traffic_env *e, float *obs) {   /* environment.py:313-363 */
    int o = 0;
    const int NI_ = h->ni;
    for (int i = 0; i < NI_; ++i) for (int p = 0; p < 4; ++p) obs[o++] = e->phase[i] == p ? 1.0f : 0.0f;
    for (int i = 0; i < NI_; ++i) for (int d = 0; d < 4; ++d) obs[o++] = (float)(e->qlen[i][d] < 20 ? e->qlen[i][d] : 20);
    for (int i = 0; i < NI_; ++i)
        for (int d = 0; d < 4; ++d) {
            double avg = e->qlen[i][d] ? (double)e->qwait[i][d] / (double)e->qlen[i][d] : 0.0;
            obs[o++] = (float)(avg < 100 ? avg : 100);
        }
    int tp = 0, tw = 0, tq = 0;
    for (int i = 0; i < NI_; ++i) {
        obs[o++] = (float)e->passed[i];
        obs[o++] = (float)(e->total_wait[i] < 1000 ? e->total_wait[i] : 1000);
        tp += e->passed[i]; tw += e->total_wait[i];
        for (int d = 0; d < 4; ++d) tq += e->qlen[i][d];
    }
    double avg_wait = (double)tw / (double)(tp > 1 ? tp : 1);             /* utils.py:257 */
    double avg_q = (double)tq / (double)NI_, thr = (double)tp / (double)NI_;
    obs[o++] = (float)e->n_vehicles;
    obs[o++] = (float)(avg_wait < 100 ? avg_wait : 100);
    obs[o++] = (float)(avg_q < 50 ? avg_q : 50);
    obs[o++] = (float)thr;
}

/* one reference step(); returns terminated */
static int env_step(const orc_traffic *h, traffic_env *e, const int32_t *a, double *reward) {   /* :168-203 */
    const int NI_ = h->ni;
    e->timestep += 1;
    for (int i = 0; i < NI_; ++i) {                                        /* _apply_actions :205-220 */
        if (a[i] == 1 && e->phase[i] != NS_GREEN) { e->phase[i] = NS_GREEN; e->timer[i] = 5; }
        else if (a[i] == 2 && e->phase[i] != EW_GREEN) { e->phase[i] = EW_GREEN; e->timer[i] = 5; }
    }
    for (int i = 0; i < NI_; ++i) {                                        /* TrafficLight.update, utils.py:79-97 */
        e->timer[i] -= 1;
        if (e->timer[i] <= 0) {
            e->phase[i] = (e->phase[i] + 1) % 4;
            e->timer[i] = (e->phase[i] & 1) ? 3 : orc_py_randint(&e->P, 5, 30);
        }
    }
    spawn(h, e);
    for (int i = 0; i < NI_; ++i)                                          /* process_vehicles, utils.py:141-163 */
        for (int d = 0; d < 4; ++d) {
            int len = e->qlen[i][d];
            if (!len) continue;
            int pass = (e->phase[i] == NS_GREEN && (d == NORTH || d == SOUTH)) || (e->phase[i] == EW_GREEN && (d == EAST || d == WEST));
            if (pass) {
                e->passed[i] += len;
                e->n_vehicles -= e->qdest[i][d];                          /* reached destination -> removed, :279-285 */
                e->qlen[i][d] = 0; e->qdest[i][d] = 0; e->qwait[i][d] = 0;
            } else {
                e->qwait[i][d] += len;
                e->total_wait[i] += len;
            }
        }
    int tp = 0, tw = 0, tq = 0, qt[NI];                                   /* _calculate_reward :287-311 */
    for (int i = 0; i < NI_; ++i) {
        tp += e->passed[i]; tw += e->total_wait[i];
        qt[i] = e->qlen[i][0] + e->qlen[i][1] + e->qlen[i][2] + e->qlen[i][3];
        tq += qt[i];
    }
    double r = 0.0;
    r += tp * 1.0;
    r += tw * -0.1;
    r += tq * -0.05;
    r += 0.5 / (1 + np_var(qt, NI_));
    e->total_reward += r;
    *reward = r;
    return e->timestep >= h->max_steps;
}

orc_traffic *orc_traffic_create(int64_t n, int mode) {
    if (n <= 0 || mode < 0 || mode > 2) return NULL;
    orc_traffic *h = (orc_traffic *)calloc(1, sizeof(*h));
    h->n = n; h->mode = mode; h->max_steps = 1000; h->max_vehicles = 50; h->spawn_rate = 0.3;
    h->rows = 5; h->cols = 5; h->ni = 9; h->obs = 130;                   /* config.py:6-7 */
    h->e = (traffic_env *)calloc((size_t)n, sizeof(traffic_env));
    eps_init(&h->eps, n);
    for (int64_t i = 0; i < n; ++i) orc_py_seed(&h->e[i].P, (uint64_t)i);
    return h;
}
/* TrafficManagementEnv(grid_size=(rows, cols), num_intersections, max_vehicles, spawn_rate) (environment.py:62-83); call
 * right after create.  Returns 0, or -1 if the layout is outside this restatement's capacity. */
int orc_traffic_set_layout(orc_traffic *h, int rows, int cols, int num_intersections, int max_vehicles, double spawn_rate) {
    if (rows < 1 || cols < 1 || num_intersections < 1) return -1;
    int ni = num_intersections < rows * cols ? num_intersections : rows * cols;   /* :79 */
    if (ni > NI) return -1;
    h->rows = rows; h->cols = cols; h->ni = ni; h->obs = 14 * ni + 4;
    h->max_vehicles = max_vehicles; h->spawn_rate = spawn_rate;
    return 0;
}
int orc_traffic_obs_dim(const orc_traffic *h) { return h->obs; }
int orc_traffic_num_intersections(const orc_traffic *h) { return h->ni; }
void orc_traffic_destroy(orc_traffic *h) { if (h) { free(h->e); eps_free(&h->eps); free(h); } }
void orc_traffic_seed(orc_traffic *h, const uint64_t *seeds) { for (int64_t i = 0; i < h->n; ++i) orc_py_seed(&h->e[i].P, seeds[i]); }

void orc_traffic_reset(orc_traffic *h, const uint8_t *mask, float *obs) {
    for (int64_t i = 0; i < h->n; ++i) {
        if (!mask || mask[i]) { env_reset(&h->e[i]); eps_clear(&h->eps, i); }
        if (obs) write_obs(h, &h->e[i], obs + i * h->obs);
    }
}

void orc_traffic_step(orc_traffic *h, const int32_t *actions, float *obs, float *reward, double *reward64,
                      uint8_t *terminated, uint8_t *truncated, float *final_obs) {
    for (int64_t i = 0; i < h->n; ++i) {
        traffic_env *e = &h->e[i];
        float *o = obs + i * h->obs;
        if (h->mode == 0 && e->needs_reset) {
            { env_reset(e); eps_clear(&h->eps, i); }
            write_obs(h, e, o);
            reward[i] = 0.0f; if (reward64) reward64[i] = 0.0; terminated[i] = 0; truncated[i] = 0;
            continue;
        }
        double r;
        int term = env_step(h, e, actions + i * h->ni, &r);
        eps_add(&h->eps, i, (double)r);
        reward[i] = (float)r; if (reward64) reward64[i] = r;
        terminated[i] = (uint8_t)term; truncated[i] = 0;
        if (term) { e->episodes += 1; eps_done(&h->eps, i); }
        if (term && h->mode == 1) {
            if (final_obs) write_obs(h, e, final_obs + i * h->obs);
            { env_reset(e); eps_clear(&h->eps, i); }
            write_obs(h, e, o);
        } else {
            write_obs(h, e, o);
            if (term && h->mode == 0) e->needs_reset = 1;
        }
    }
}

void orc_traffic_rollout(orc_traffic *h, int k_steps, uint64_t a_seed, int64_t t0, int64_t env0, float *obs,
                         double *reward_sum, int32_t *done_count) {
    float scratch[TOBS_MAX];
    for (int64_t i = 0; i < h->n; ++i) {
        traffic_env *e = &h->e[i];
        double rs = 0.0;
        int dc = 0;
        for (int t = 0; t < k_steps; ++t) {
            if (h->mode == 0 && e->needs_reset) { { env_reset(e); eps_clear(&h->eps, i); } continue; }
            int32_t a[NI];
            for (int j = 0; j < h->ni; ++j) a[j] = (int32_t)orc_hash_action(a_seed, (uint64_t)(env0 + i), (uint64_t)(t0 + t), 3, (uint32_t)j);
            double r;
            int term = env_step(h, e, a, &r);
            eps_add(&h->eps, i, (double)r);
            rs += r;
            if (obs) write_obs(h, e, scratch);
            if (term) {
                ++dc; e->episodes += 1; eps_done(&h->eps, i);
                if (h->mode == 1) { env_reset(e); eps_clear(&h->eps, i); }
                else if (h->mode == 0) e->needs_reset = 1;
            }
        }
        if (obs) write_obs(h, e, obs + i * h->obs);
        if (reward_sum) reward_sum[i] = rs;
        if (done_count) done_count[i] = dc;
    }
}

/* field: 0 timestep 1 num_vehicles 2 light_phase[idx] 3 light_timer[idx] 4 vehicles_passed[idx]
 *        5 total_waiting_time[idx] 6 queue_len[idx<36] 7 queue_dest[idx] 8 queue_wait[idx] 9 episodes 10 needs_reset */
void orc_traffic_info(const orc_traffic *h, int field, int idx, int32_t *out) {
    for (int64_t i = 0; i < h->n; ++i) {
        const traffic_env *e = &h->e[i];
        int v = 0;
        switch (field) {
            case 0: v = e->timestep; break;
            case 1: v = e->n_vehicles; break;
            case 2: v = e->phase[idx]; break;
            case 3: v = e->timer[idx]; break;
            case 4: v = e->passed[idx]; break;
            case 5: v = e->total_wait[idx]; break;
            case 6: v = e->qlen[idx / 4][idx % 4]; break;
            case 7: v = e->qdest[idx / 4][idx % 4]; break;
            case 8: v = e->qwait[idx / 4][idx % 4]; break;
            case 9: v = e->episodes; break;
            case 10: v = e->needs_reset; break;
        }
        out[i] = v;
    }
}
void orc_traffic_total_reward(const orc_traffic *h, double *out) { for (int64_t i = 0; i < h->n; ++i) out[i] = h->e[i].total_reward; }

/* Canonical state record shared with the device library: int32[6] {timestep, n_vehicles, needs_reset, mt_idx,
 * episodes, 0}; double total_reward; int32 phase[ni], timer[ni], passed[ni], total_wait[ni], qlen[4 ni], qdest[4 ni],
 * qwait[4 ni]; uint32 mt[624]  (ni = 9 for the default layout). */
size_t orc_traffic_state_bytes(const orc_traffic *h) { return 6 * 4 + 8 + (size_t)(4 * h->ni + 12 * h->ni) * 4 + 624 * 4; }

void orc_traffic_get_state(const orc_traffic *h, void *buf) {
    size_t rec = orc_traffic_state_bytes(h);
    const int ni = h->ni;
    for (int64_t i = 0; i < h->n; ++i) {
        const traffic_env *e = &h->e[i];
        uint8_t *p = (uint8_t *)buf + i * rec;
        int32_t hd[6] = {e->timestep, e->n_vehicles, e->needs_reset, e->P.idx, e->episodes, 0};
        memcpy(p, hd, 24); memcpy(p + 24, &e->total_reward, 8);
        int32_t *w = (int32_t *)(p + 32);
        memcpy(w, e->phase, 4 * ni); memcpy(w + ni, e->timer, 4 * ni); memcpy(w + 2 * ni, e->passed, 4 * ni); memcpy(w + 3 * ni, e->total_wait, 4 * ni);
        memcpy(w + 4 * ni, e->qlen, 16 * ni); memcpy(w + 8 * ni, e->qdest, 16 * ni); memcpy(w + 12 * ni, e->qwait, 16 * ni);
        memcpy(w + 16 * ni, e->P.mt, 2496);
    }
}
void orc_traffic_set_state(orc_traffic *h, const void *buf) {
    size_t rec = orc_traffic_state_bytes(h);
    const int ni = h->ni;
    for (int64_t i = 0; i < h->n; ++i) {
        traffic_env *e = &h->e[i];
        const uint8_t *p = (const uint8_t *)buf + i * rec;
        int32_t hd[6];
        memcpy(hd, p, 24); memcpy(&e->total_reward, p + 24, 8);
        e->timestep = hd[0]; e->n_vehicles = hd[1]; e->needs_reset = hd[2]; e->P.idx = hd[3]; e->episodes = hd[4];
        const int32_t *w = (const int32_t *)(p + 32);
        memcpy(e->phase, w, 4 * ni); memcpy(e->timer, w + ni, 4 * ni); memcpy(e->passed, w + 2 * ni, 4 * ni); memcpy(e->total_wait, w + 3 * ni, 4 * ni);
        memcpy(e->qlen, w + 4 * ni, 16 * ni); memcpy(e->qdest, w + 8 * ni, 16 * ni); memcpy(e->qwait, w + 12 * ni, 16 * ni);
        memcpy(e->P.mt, w + 16 * ni, 2496);
    }
}

/* Time-limit override for the short-horizon parity tests (the reference's limit is a constructor constant /
 * config value; the device ABI takes it in its config struct).  Call before reset(). */
void orc_traffic_set_max_steps(orc_traffic *h, int v) { h->max_steps = v; }

/* return and length of each env's last finished episode (orc_epstats.h) */
void orc_traffic_episode_stats(const orc_traffic *h, double *ret, int32_t *len) { eps_get(&h->eps, h->n, ret, len); }
