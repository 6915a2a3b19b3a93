/* oracle/orc_manufacturing.c — CPU restatement of SmartManufacturingEnv over a batch of independent envs.
 *
 * TEST INFRASTRUCTURE ONLY (see orc_rng.h).  Follows /root/reference/smart_manufacturing_env/manufacturing_env.py:
 *   reset :113-192, _get_observation :194-250, step :252-301, _process_action :303-359, _start_production :361-379,
 *   _update_production :381-425, _update_machine_status :427-462, _quality_control :464-480, _complete_product :482-500
 *   (its return value is discarded at :418), _calculate_timestep_rewards :502-531, _update_metrics :533-553,
 *   _check_termination :555-578, _update_supply_chain :580-595.
 * Generator: family G — gymnasium's self.np_random = Generator(PCG64(SeedSequence(seed))) (integers = Lemire on buffered
 * 32-bit draws, uniform/random = 53-bit doubles).  np.mean = NumPy pairwise summation / n, restated below for any n.
 * Parity pins: tests/golden/manufacturing_{hash,biased,typea}.npz + manufacturing_kat.json (KAT-M1) —
 * tests/test_oracle_manufacturing.py.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "orc_rng.h"
#include "orc_epstats.h"

#define MOBS 73
#define MCAP 512      /* products started per episode: <= (250 + 29*99)/10 = 312 inside an episode */
#define MHIST 2048

enum { OPERATIONAL = 0, BROKEN = 1, MAINTENANCE = 2 };
enum { BALANCED = 0, RUSH = 1, QUALITY = 2 };

static const int REQ_STATIONS[6] = {1, 2, 3, 4, 5, 3};          /* ProductType :35-40 */
static const int TIMESTEPS[6] = {10, 15, 20, 25, 30, 18};

typedef struct { double q; int type, cs, rem2, alive; } product;  /* rem2 = timesteps_remaining in half steps */

typedef struct {
    orc_pcg g;
    int status[5], ops[5], mcount[5], cur[5];
    double util[5], degr[5];
    int queue[5][MCAP], qhead[5], qlen[5];
    product prod[MCAP];
    int nprod;
    double thr[3];
    int raw, targets[6], completed[6], mode, emergency, timestep, disruption, disruption_cd, energy;
    double total_reward;
    double comp_q[MCAP];
    int ncomp, ngood, nscrap;
    double hist[MHIST];
    int nhist;
    double oee_perf;
    int needs_reset, episodes, overflow;
} menv;

typedef struct { int64_t n; int mode, max_steps; menv *e; orc_eps eps; } orc_manufacturing;

/* NumPy pairwise summation (loops_utils.h.src pairwise_sum), any n */
static double np_sum(const double *a, int n) {
    if (n < 8) {
        double res = 0.0;
        for (int i = 0; i < n; ++i) res += a[i];
        return res;
    }
    if (n <= 128) {
        double r[8];
        int i;
        for (i = 0; i < 8; ++i) r[i] = a[i];
        for (i = 8; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; ++j) r[j] += a[i + j];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res += a[i];
        return res;
    }
    int n2 = n / 2;
    n2 -= n2 % 8;
    return np_sum(a, n2) + np_sum(a + n2, n - n2);
}
static double np_mean(const double *a, int n) { return np_sum(a, n) / (double)n; }

static void env_reset(menv *e) {                                          /* :113-192 */
    for (int s = 0; s < 5; ++s) {
        e->status[s] = OPERATIONAL; e->util[s] = 0.0; e->ops[s] = 0;
        e->mcount[s] = (int)orc_pcg_integers(&e->g, 100, 200);
        e->cur[s] = -1; e->qhead[s] = 0; e->qlen[s] = 0; e->degr[s] = 0.0;
    }
    e->nprod = 0; e->ncomp = 0; e->ngood = 0; e->nscrap = 0;
    e->thr[0] = 0.70; e->thr[1] = 0.80; e->thr[2] = 0.85;
    e->raw = 250;
    static const int TLO[6] = {5, 4, 3, 2, 2, 3}, THI[6] = {10, 8, 6, 5, 4, 7};
    for (int k = 0; k < 6; ++k) { e->targets[k] = (int)orc_pcg_integers(&e->g, TLO[k], THI[k]); e->completed[k] = 0; }
    e->mode = BALANCED; e->emergency = 0; e->timestep = 0; e->total_reward = 0.0; e->nhist = 0;
    e->disruption = 0; e->disruption_cd = 0; e->energy = 0; e->oee_perf = 1.0; e->needs_reset = 0;
}

static void queue_push(menv *e, int s, int id) { e->queue[s][(e->qhead[s] + e->qlen[s]) % MCAP] = id; e->qlen[s] += 1; }
static int queue_pop(menv *e, int s) { int id = e->queue[s][e->qhead[s]]; e->qhead[s] = (e->qhead[s] + 1) % MCAP; e->qlen[s] -= 1; return id; }

static void write_obs(const menv *e, float *obs) {                         /* :194-250 */
    int idx = 0;
    for (int s = 0; s < 5; ++s)
        for (int k = 0; k < 6; ++k) {
            int c = 0;
            for (int i = 0; i < e->nprod; ++i) c += e->prod[i].alive && e->prod[i].cs == s && e->prod[i].type == k;
            obs[idx++] = (float)c;
        }
    for (int s = 0; s < 5; ++s) {
        obs[idx++] = (float)(e->status[s] == OPERATIONAL); obs[idx++] = (float)(e->status[s] == BROKEN); obs[idx++] = (float)(e->status[s] == MAINTENANCE);
    }
    for (int s = 0; s < 5; ++s) obs[idx++] = (float)e->qlen[s];
    double buf[MCAP];
    for (int k = 0; k < 6; ++k) {                                          /* mean over products_in_system (id order) */
        int n = 0;
        for (int i = 0; i < e->nprod; ++i) if (e->prod[i].alive && e->prod[i].type == k) buf[n++] = e->prod[i].q;
        obs[idx++] = n ? (float)(np_mean(buf, n) * 100) : 85.0f;
    }
    obs[idx++] = (float)e->raw;
    for (int k = 0; k < 6; ++k) { int r = e->targets[k] - e->completed[k]; obs[idx++] = (float)(r > 0 ? r : 0); }
    for (int s = 0; s < 5; ++s) obs[idx++] = (float)(e->util[s] * 100);
    for (int s = 0; s < 5; ++s) obs[idx++] = (float)e->mcount[s];
}

static double recent_mean(const double *a, int n, int last) { int m = n < last ? n : last; return np_mean(a + n - m, m); }

/* returns terminated | truncated << 1; *reward is the (integer-valued) step reward */
static int env_step(const orc_manufacturing *h, menv *e, int action, double *reward_out) {             /* :252-301 */
    int reward = 0;
    e->timestep += 1;
    /* _process_action :303-359 */
    if (action <= 5) {
        if (e->raw >= 10) {
            if (e->nprod >= MCAP) { e->overflow += 1; }
            else {                                                         /* _start_production :361-379 */
                product *p = &e->prod[e->nprod];
                p->type = action; p->cs = -1; p->q = 0.85 + orc_pcg_uniform(&e->g, -0.1, 0.1); p->rem2 = 2 * TIMESTEPS[action]; p->alive = 1;
                if (e->status[0] == OPERATIONAL) queue_push(e, 0, e->nprod);
                e->nprod += 1;
            }
            e->raw -= 10;
        } else reward -= 50;
    } else if (action <= 10) {
        int s = action - 6;
        double u = e->util[s] + 0.2;
        e->util[s] = u < 1.0 ? u : 1.0;
        e->energy += 5;
    } else if (action <= 15) {
        int s = action - 11;
        if (e->status[s] == OPERATIONAL) { e->status[s] = MAINTENANCE; e->mcount[s] = 20; reward += 50; }
    } else if (action <= 20) {
        int c = action - 16;
        if (c < 3) { double t = e->thr[c] + 0.05; e->thr[c] = t < 0.95 ? t : 0.95; }
    } else if (action == 21) {
        e->emergency = !e->emergency;
        if (e->emergency) reward -= 100;
    } else if (action == 22) { e->mode = RUSH; e->energy += 10; }
    else if (action == 23) e->mode = QUALITY;
    else if (action == 24) e->mode = BALANCED;
    /* _update_production :381-425 */
    if (!e->emergency)
        for (int s = 0; s < 5; ++s) {
            if (e->status[s] != OPERATIONAL) continue;
            if (e->cur[s] >= 0) {
                product *p = &e->prod[e->cur[s]];
                p->rem2 -= 2;
                if (e->mode == RUSH) { p->rem2 -= 1; p->q *= 0.98; }
                else if (e->mode == QUALITY) p->q *= 1.02;
                p->q *= (1 - e->degr[s]);
                if (p->rem2 <= 0) {
                    int id = e->cur[s];
                    e->cur[s] = -1; e->ops[s] += 1;
                    int next = p->cs + 1;
                    if (next < REQ_STATIONS[p->type]) { queue_push(e, next, id); p->cs = next; }
                    else {                                                 /* _complete_product :482-500 (reward discarded) */
                        e->completed[p->type] += 1;
                        e->comp_q[e->ncomp++] = p->q;
                        if (p->q > 0.7) e->ngood += 1;
                        p->alive = 0;
                    }
                }
            }
            if (e->cur[s] < 0 && e->qlen[s] > 0) { e->cur[s] = queue_pop(e, s); e->util[s] = 0.8; }
            else e->util[s] *= 0.95;
        }
    /* _update_machine_status :427-462 */
    for (int s = 0; s < 5; ++s) {
        if (e->status[s] == MAINTENANCE) {
            e->mcount[s] -= 1;
            if (e->mcount[s] <= 0) { e->status[s] = OPERATIONAL; e->degr[s] = 0; e->ops[s] = 0; e->mcount[s] = (int)orc_pcg_integers(&e->g, 100, 200); }
        } else if (e->status[s] == OPERATIONAL) {
            double prob = 0.001 * (1 + (double)e->ops[s] / 100);
            if (orc_pcg_double(&e->g) < prob) { e->status[s] = BROKEN; e->mcount[s] = 30; }
            if (e->ops[s] % 100 == 0) e->degr[s] += 0.005;
            e->mcount[s] -= 1;
        } else {
            e->mcount[s] -= 1;
            if (e->mcount[s] <= 0) { e->status[s] = OPERATIONAL; e->mcount[s] = (int)orc_pcg_integers(&e->g, 100, 200); }
        }
    }
    /* _quality_control :464-480 */
    static const int CHECKPOINT[3] = {1, 3, 4};
    for (int c = 0; c < 3; ++c) {
        int s = CHECKPOINT[c];
        if (e->cur[s] >= 0 && e->prod[e->cur[s]].q < e->thr[c]) { e->prod[e->cur[s]].alive = 0; e->nscrap += 1; e->cur[s] = -1; reward -= 100; }
    }
    /* _update_metrics :533-553 */
    e->oee_perf = np_mean(e->util, 5);
    if (e->ncomp > 0) {
        if (e->nhist < MHIST) e->hist[e->nhist++] = recent_mean(e->comp_q, e->ncomp, 20);
        else { memmove(e->hist, e->hist + 1, (MHIST - 1) * sizeof(double)); e->hist[MHIST - 1] = recent_mean(e->comp_q, e->ncomp, 20); }
    }
    /* _calculate_timestep_rewards :502-531 */
    int broken = 0;
    for (int s = 0; s < 5; ++s) {
        if (e->status[s] == OPERATIONAL && e->cur[s] < 0 && e->qlen[s] == 0) reward -= 10;
        broken += e->status[s] == BROKEN;
    }
    reward -= broken * 50;
    int all_met = 1;
    for (int k = 0; k < 6; ++k) {
        if (e->timestep > 1000 && e->completed[k] < e->targets[k]) reward -= 50;
        all_met = all_met && e->completed[k] >= e->targets[k];
    }
    if (e->ncomp > 0) {
        double rq = recent_mean(e->comp_q, e->ncomp, 10);
        if (rq > 0.9) reward += 20; else if (rq < 0.6) reward -= 30;
    }
    /* _check_termination :555-578 */
    int term = all_met || broken >= 3 || e->timestep >= h->max_steps;   /* 1500 (:282) */
    if (!term && e->nhist >= 100 && np_mean(e->hist + e->nhist - 100, 100) < 0.6) term = 1;
    int trunc = e->timestep >= h->max_steps;                          /* :569 */
    /* _update_supply_chain :580-595 */
    if (!e->disruption && orc_pcg_double(&e->g) < 0.01) { e->disruption = 1; e->disruption_cd = (int)orc_pcg_integers(&e->g, 20, 50); }
    if (e->disruption) { e->disruption_cd -= 1; if (e->disruption_cd <= 0) e->disruption = 0; }
    else if (e->timestep % 50 == 0) { int r = e->raw + (int)orc_pcg_integers(&e->g, 50, 100); e->raw = r < 500 ? r : 500; }
    e->total_reward += reward;
    *reward_out = (double)reward;
    return term | (trunc << 1);
}

orc_manufacturing *orc_manufacturing_create(int64_t n, int mode) {
    if (n <= 0 || mode < 0 || mode > 2) return NULL;
    orc_manufacturing *h = (orc_manufacturing *)calloc(1, sizeof(*h));
    h->n = n; h->mode = mode; h->max_steps = 1500;
    h->e = (menv *)calloc((size_t)n, sizeof(menv));
    eps_init(&h->eps, n);
    for (int64_t i = 0; i < n; ++i) { orc_pcg_seed(&h->e[i].g, (uint64_t)i); h->e[i].thr[0] = 0.70; h->e[i].thr[1] = 0.80; h->e[i].thr[2] = 0.85; h->e[i].raw = 250; }
    return h;
}
void orc_manufacturing_destroy(orc_manufacturing *h) { if (h) { free(h->e); eps_free(&h->eps); free(h); } }
/* reset(seed=s): self.np_random = Generator(PCG64(SeedSequence(s))) */
void orc_manufacturing_seed(orc_manufacturing *h, const uint64_t *seeds) { for (int64_t i = 0; i < h->n; ++i) orc_pcg_seed(&h->e[i].g, seeds[i]); }

void orc_manufacturing_reset(orc_manufacturing *h, const uint8_t *mask, float *obs) {
    for (int64_t i = 0; i < h->n; ++i) {
        if (!mask || mask[i]) { env_reset(&h->e[i]); eps_clear(&h->eps, i); }
        if (obs) write_obs(&h->e[i], obs + i * MOBS);
    }
}

void orc_manufacturing_step(orc_manufacturing *h, const int32_t *actions, float *obs, float *reward, double *reward64, uint8_t *terminated,
                            uint8_t *truncated, float *final_obs) {
    for (int64_t i = 0; i < h->n; ++i) {
        menv *e = &h->e[i];
        float *o = obs + i * MOBS;
        if (h->mode == 0 && e->needs_reset) {
            { env_reset(e); eps_clear(&h->eps, i); } write_obs(e, o);
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
            if (final_obs) write_obs(e, final_obs + i * MOBS);
            { env_reset(e); eps_clear(&h->eps, i); } write_obs(e, o);
        } else {
            write_obs(e, o);
            if (f && h->mode == 0) e->needs_reset = 1;
        }
    }
}

void orc_manufacturing_rollout(orc_manufacturing *h, int k_steps, uint64_t a_seed, int64_t t0, int64_t env0, float *obs,
                               double *reward_sum, int32_t *done_count) {
    for (int64_t i = 0; i < h->n; ++i) {
        menv *e = &h->e[i];
        double rs = 0.0;
        int dc = 0;
        for (int t = 0; t < k_steps; ++t) {
            if (h->mode == 0 && e->needs_reset) { { env_reset(e); eps_clear(&h->eps, i); } continue; }
            double r;
            int f = env_step(h, e, (int)orc_hash_action(a_seed, (uint64_t)(env0 + i), (uint64_t)(t0 + t), 25, 0), &r);
            eps_add(&h->eps, i, (double)r);
            rs += r;
            if (f) { ++dc; e->episodes += 1; eps_done(&h->eps, i); if (h->mode == 1) { env_reset(e); eps_clear(&h->eps, i); } else if (h->mode == 0) e->needs_reset = 1; }
        }
        if (obs) write_obs(e, obs + i * MOBS);
        if (reward_sum) reward_sum[i] = rs;
        if (done_count) done_count[i] = dc;
    }
}

/* float64 fields: 0 raw_material 1 energy 2 total_reward 3 in_system 4 completed 5 scrapped 6 product_ids 7 history_len
 *                 8 oee availability 9 oee performance 10 oee quality 11 timestep 12 episodes 13 needs_reset 14 overflow */
void orc_manufacturing_info(const orc_manufacturing *h, int field, double *out) {
    for (int64_t i = 0; i < h->n; ++i) {
        const menv *e = &h->e[i];
        double v = 0;
        int c = 0;
        switch (field) {
            case 0: v = e->raw; break; case 1: v = e->energy; break; case 2: v = e->total_reward; break;
            case 3: for (int k = 0; k < e->nprod; ++k) c += e->prod[k].alive; v = c; break;
            case 4: v = e->ncomp; break; case 5: v = e->nscrap; break; case 6: v = e->nprod; break; case 7: v = e->nhist; break;
            case 8: for (int s = 0; s < 5; ++s) c += e->status[s] == OPERATIONAL; v = (double)c / 5; break;
            case 9: v = e->oee_perf; break;
            case 10: v = e->ncomp ? (double)e->ngood / (double)e->ncomp : 1.0; break;
            case 11: v = e->timestep; break; case 12: v = e->episodes; break; case 13: v = e->needs_reset; break; case 14: v = e->overflow; break;
        }
        out[i] = v;
    }
}

/* Time-limit override for the short-horizon parity tests (the reference's limit is a constructor constant /
 * config value; the device ABI takes it in its config struct).  Call before reset(). */
void orc_manufacturing_set_max_steps(orc_manufacturing *h, int v) { h->max_steps = v; }

/* return and length of each env's last finished episode (orc_epstats.h) */
void orc_manufacturing_episode_stats(const orc_manufacturing *h, double *ret, int32_t *len) { eps_get(&h->eps, h->n, ret, len); }
