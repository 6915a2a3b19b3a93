/* oracle/orc_parking.c — CPU restatement of SmartParkingEnv over a batch of independent envs.
 *
 * TEST INFRASTRUCTURE ONLY (see orc_rng.h).  Follows /root/reference/smart_parking_env/core/:
 *   parking_env.py  reset :70-108, step :110-159, _process_action :161-195, _reject_customer :197-216,
 *                   _assign_customer_to_zone :218-269, _toggle_zone_price :271-304, _get_observation :306-369,
 *                   _get_info :371-399
 *   customer.py     Customer :18-138 (calculate_satisfaction :84-124, get_duration_discount :126-138),
 *                   generate_customer :146-184, get_adjusted_zone_preferences :187-220,
 *                   get_time_based_duration_type :223-240, should_customer_arrive :243-254,
 *                   CustomerManager :257-353
 *   parking_lot.py  add_to_queue :193-208, get_next_queued_customer :210-227, update_time_minute :229-252
 *   pricing.py      _update_prices :59-64, calculate_revenue :96-118, _get_duration_multiplier :120-136
 *   config.py       zones 15/20/15 @ 8/5/3, ARRIVAL_RATES :19-47, DURATION_TYPES :50-54, PRICE_LEVELS :81-85,
 *                   MAX_QUEUE_SIZE 10, price-change limits :91-95, REWARD_WEIGHTS :98-103
 * Quirks preserved: generate_customer() receives the TIMESTEP where an hour is expected (customer.py:284 vs
 * :146); the duration discount is applied twice (pricing.py:110-112 with customer.py:126-138); a rejected
 * customer still counts as "satisfied" (default satisfaction 1.0 > 0.7, customer.py:328); arrivals at a full
 * queue are dropped but counted; a failed assignment re-queues the customer at the BACK (parking_env.py:231-234).
 * Generator: family P = CPython global `random`, never seeded by the env (parking_env.py:81) — per-env stream.
 * Parity pins: tests/golden/parking_*.npz + parking_kat.json — checked by tests/test_oracle_parking.py.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "orc_rng.h"
#include "orc_epstats.h"

#define NSPOT 50
#define QMAX 10
#define POBS 13
static const int ZONE_FIRST[4] = {0, 15, 35, 50};
static const double BASE_PRICE[3] = {8.0, 5.0, 3.0};
static const double LEVEL_MULT[3] = {0.7, 1.0, 1.3};
static const int ARRIVAL_RATE[24] = {3, 2, 2, 2, 2, 4, 8, 12, 15, 10, 8, 8, 12, 12, 8, 8, 10, 15, 18, 15, 12, 8, 6, 4};

typedef struct { int pref, duration, wait; } customer;   /* pref: 0 A, 1 B, 2 C, 3 flexible */
typedef struct { int occupied, duration, arrival; double satisfaction; } spot;

typedef struct {
    orc_mt P;
    spot s[NSPOT];
    customer q[QMAX];
    int qlen, level[3], t, changes, last_change, needs_reset, episodes;
    int total_customers, rejected, satisfied, total_wait;
    double revenue, satisfaction_sum;
} parking_env;

typedef struct { int64_t n; int mode, max_steps; parking_env *e; orc_eps eps; } orc_parking;

static void env_reset(parking_env *e) {                                  /* parking_env.py:70-108, no draws */
    memset(e->s, 0, sizeof e->s);
    e->qlen = 0; e->level[0] = e->level[1] = e->level[2] = 1;             /* pricing.py:23 */
    e->t = 0; e->changes = 0; e->last_change = -999; e->needs_reset = 0;
    e->total_customers = e->rejected = e->satisfied = e->total_wait = 0;
    e->revenue = 0.0; e->satisfaction_sum = 0.0;
}

static double zone_price(const parking_env *e, int z) { return BASE_PRICE[z] * LEVEL_MULT[e->level[z]]; }   /* pricing.py:59-64 */

static int free_in_zone(const parking_env *e, int z) {
    for (int k = ZONE_FIRST[z]; k < ZONE_FIRST[z + 1]; ++k) if (!e->s[k].occupied) return k;
    return -1;
}

static void generate_customer(parking_env *e, int hour /* actually the timestep */, customer *c) {   /* customer.py:146-184 */
    double p[4] = {0.3, 0.35, 0.15, 0.2};                                  /* config.py:71-76 */
    if (6 <= hour && hour <= 9) { p[0] *= 1.3; p[1] *= 1.1; p[2] *= 0.8; p[3] *= 0.9; }
    else if (17 <= hour && hour <= 19) { p[0] *= 0.9; p[1] *= 1.2; p[2] *= 1.1; p[3] *= 1.0; }
    else if (22 <= hour || hour <= 5) { p[0] *= 0.7; p[1] *= 0.8; p[2] *= 1.0; p[3] *= 1.5; }
    double total = 0 + p[0]; total += p[1]; total += p[2]; total += p[3];   /* sum() */
    double rnd = orc_mt_double(&e->P), cum = 0;
    c->pref = 3;
    for (int z = 0; z < 4; ++z) { cum += p[z] / total; if (rnd <= cum) { c->pref = z; break; } }
    int type;                                                              /* 0 short 1 medium 2 long, customer.py:223-240 */
    if (6 <= hour && hour <= 9) type = orc_mt_double(&e->P) < 0.6 ? 0 : 1;
    else if (12 <= hour && hour <= 14) type = (int)orc_py_randbelow(&e->P, 3);
    else if (17 <= hour && hour <= 19) type = orc_mt_double(&e->P) < 0.6 ? 1 : 2;
    else type = orc_mt_double(&e->P) < 0.5 ? 2 : 1;
    static const int lo[3] = {1, 3, 6}, hi[3] = {2, 5, 12};
    int dur = orc_py_randint(&e->P, lo[type], hi[type]);
    if (orc_mt_double(&e->P) < 0.3) dur += orc_py_randint(&e->P, 1, 3);
    c->duration = dur < 24 ? dur : 24;
    c->wait = 0;
}

static double process_action(parking_env *e, int a) {                     /* parking_env.py:161-195 */
    if (a >= 1 && a <= 4) {
        if (e->qlen == 0) return 0.0;
        customer c = e->q[0];                                              /* pop(0) */
        memmove(e->q, e->q + 1, (size_t)(e->qlen - 1) * sizeof(customer));
        e->qlen -= 1;
        if (a == 4) {                                                      /* _reject_customer :197-216 */
            e->rejected += 1;
            e->satisfied += 1;                                             /* remove_customer(): default satisfaction 1.0 > 0.7 */
            int avail = 0;
            for (int k = 0; k < NSPOT; ++k) avail += !e->s[k].occupied;
            return avail > 0 ? -2.0 : 0.0;
        }
        int z = a - 1, k = free_in_zone(e, z);                             /* _assign_customer_to_zone :218-269 */
        if (k < 0) { e->q[e->qlen++] = c; return 0.0; }                    /* back of the queue */
        double disc = c.duration >= 6 ? 0.75 : c.duration >= 3 ? 0.85 : 1.0;
        double base = zone_price(e, z);
        double hourly = base * disc * disc;                                /* duration multiplier, then customer discount */
        double total_price = hourly * c.duration;
        double s = 1.0;                                                    /* calculate_satisfaction, customer.py:84-124 */
        if (c.pref == 3) s *= 0.9; else if (c.pref == z) s *= 1.0; else s *= 0.6;
        double ratio = total_price / base;
        if (ratio <= 0.8) s *= 1.1; else if (ratio <= 1.0) s *= 1.0; else if (ratio <= 1.2) s *= 0.8; else s *= 0.5;
        if (c.wait > 0) { double pen = c.wait / 60.0; if (pen > 0.3) pen = 0.3; s *= (1.0 - pen); }
        s = s < 1.0 ? s : 1.0; s = s > 0.0 ? s : 0.0;
        e->s[k].occupied = 1; e->s[k].duration = c.duration; e->s[k].arrival = e->t; e->s[k].satisfaction = s;
        e->revenue += total_price;
        double reward = total_price * 1.0;
        reward += s * 0.8;
        return reward;
    }
    if (a >= 5 && a <= 7) {                                                /* _toggle_zone_price :271-304 */
        if (e->changes >= 2) return -0.5 * 2;
        if (e->t - e->last_change < 15) return -0.5 * 2;
        int z = a - 5;
        e->level[z] = (e->level[z] + 1) % 3;
        e->changes += 1; e->last_change = e->t;
        return -0.5;
    }
    return 0.0;
}

static void write_obs(const parking_env *e, float *obs) {                 /* parking_env.py:306-369 */
    double v[POBS];
    for (int z = 0; z < 3; ++z) {
        int occ = 0, tot = ZONE_FIRST[z + 1] - ZONE_FIRST[z];
        for (int k = ZONE_FIRST[z]; k < ZONE_FIRST[z + 1]; ++k) occ += e->s[k].occupied;
        v[z] = (double)occ / (double)tot;
        double p = zone_price(e, z) / (8.0 * 1.3);
        v[3 + z] = p < 1.0 ? p : 1.0;
    }
    double h = (double)(e->t / 60) / 24.0; v[6] = h < 1.0 ? h : 1.0;
    v[7] = (double)(e->t % 60) / 60.0;
    double ql = e->qlen / 10.0; v[8] = ql < 1.0 ? ql : 1.0;
    double fw = (e->qlen ? e->q[0].wait : 0) / 60.0; v[9] = fw < 1.0 ? fw : 1.0;
    int tw = 0;
    for (int k = 0; k < e->qlen; ++k) tw += e->q[k].wait;
    double twn = tw / 300.0; v[10] = twn < 1.0 ? twn : 1.0;
    double pc = (double)e->changes / 2.0; v[11] = pc < 1.0 ? pc : 1.0;
    double since = (e->t - e->last_change) / 60.0; v[12] = since < 1.0 ? since : 1.0;
    for (int j = 0; j < POBS; ++j) { float f = (float)v[j]; obs[j] = f < 0.0f ? 0.0f : (f > 1.0f ? 1.0f : f); }   /* np.clip in float32 */
}

static int env_step(const orc_parking *h, parking_env *e, int a, double *reward) {   /* parking_env.py:110-159 */
    if (e->t > 0 && e->t % 60 == 0) e->changes = 0;
    int hour = e->t / 60;
    double prob = hour < 24 ? (double)ARRIVAL_RATE[hour] / 60 : 0.05;       /* ARRIVAL_PROBABILITIES.get(hour, 0.05) */
    if (orc_mt_double(&e->P) < prob) {                                     /* should_customer_arrive */
        customer c;
        generate_customer(e, e->t, &c);                                    /* timestep passed as "hour" */
        e->total_customers += 1;
        if (e->qlen < QMAX) e->q[e->qlen++] = c;                           /* else dropped silently, parking_lot.py:203-204 */
    }
    for (int k = 0; k < e->qlen; ++k) { e->q[k].wait += 1; e->total_wait += 1; }
    double r = 0.0;
    r += process_action(e, a);
    for (int k = 0; k < NSPOT; ++k)                                        /* update_time_minute, parking_lot.py:229-252 */
        if (e->s[k].occupied && (e->t - e->s[k].arrival) / 60.0 >= e->s[k].duration) {
            e->s[k].occupied = 0;
            e->satisfaction_sum += e->s[k].satisfaction;
            if (e->s[k].satisfaction > 0.7) e->satisfied += 1;
        }
    e->t += 1;
    *reward = r;
    return e->t >= h->max_steps;
}

orc_parking *orc_parking_create(int64_t n, int mode) {
    if (n <= 0 || mode < 0 || mode > 2) return NULL;
    orc_parking *h = (orc_parking *)calloc(1, sizeof(*h));
    h->n = n; h->mode = mode; h->max_steps = 1440;
    h->e = (parking_env *)calloc((size_t)n, sizeof(parking_env));
    eps_init(&h->eps, n);
    for (int64_t i = 0; i < n; ++i) { orc_py_seed(&h->e[i].P, (uint64_t)i); { env_reset(&h->e[i]); eps_clear(&h->eps, i); } }
    return h;
}
void orc_parking_destroy(orc_parking *h) { if (h) { free(h->e); eps_free(&h->eps); free(h); } }
void orc_parking_seed(orc_parking *h, const uint64_t *seeds) { for (int64_t i = 0; i < h->n; ++i) orc_py_seed(&h->e[i].P, seeds[i]); }

void orc_parking_reset(orc_parking *h, const uint8_t *mask, float *obs) {
    for (int64_t i = 0; i < h->n; ++i) {
        if (!mask || mask[i]) { env_reset(&h->e[i]); eps_clear(&h->eps, i); }
        if (obs) write_obs(&h->e[i], obs + i * POBS);
    }
}

void orc_parking_step(orc_parking *h, const int32_t *actions, float *obs, float *reward, double *reward64,
                      uint8_t *terminated, uint8_t *truncated, float *final_obs) {
    for (int64_t i = 0; i < h->n; ++i) {
        parking_env *e = &h->e[i];
        float *o = obs + i * POBS;
        if (h->mode == 0 && e->needs_reset) {
            { env_reset(e); eps_clear(&h->eps, i); } write_obs(e, o);
            reward[i] = 0.0f; if (reward64) reward64[i] = 0.0; terminated[i] = 0; truncated[i] = 0;
            continue;
        }
        double r;
        int term = env_step(h, e, actions[i], &r);
        eps_add(&h->eps, i, (double)r);
        reward[i] = (float)r; if (reward64) reward64[i] = r;
        terminated[i] = (uint8_t)term; truncated[i] = 0;
        if (term) { e->episodes += 1; eps_done(&h->eps, i); }
        if (term && h->mode == 1) {
            if (final_obs) write_obs(e, final_obs + i * POBS);
            { env_reset(e); eps_clear(&h->eps, i); } write_obs(e, o);
        } else {
            write_obs(e, o);
            if (term && h->mode == 0) e->needs_reset = 1;
        }
    }
}

void orc_parking_rollout(orc_parking *h, int k_steps, uint64_t a_seed, int64_t t0, int64_t env0, float *obs,
                         double *reward_sum, int32_t *done_count) {
    float scratch[POBS];
    for (int64_t i = 0; i < h->n; ++i) {
        parking_env *e = &h->e[i];
        double rs = 0.0;
        int dc = 0;
        for (int t = 0; t < k_steps; ++t) {
            if (h->mode == 0 && e->needs_reset) { { env_reset(e); eps_clear(&h->eps, i); } continue; }
            double r;
            int term = env_step(h, e, (int)orc_hash_action(a_seed, (uint64_t)(env0 + i), (uint64_t)(t0 + t), 8, 0), &r);
            eps_add(&h->eps, i, (double)r);
            rs += r;
            if (obs) write_obs(e, scratch);
            if (term) { ++dc; e->episodes += 1; eps_done(&h->eps, i); if (h->mode == 1) { env_reset(e); eps_clear(&h->eps, i); } else if (h->mode == 0) e->needs_reset = 1; }
        }
        if (obs) write_obs(e, obs + i * POBS);
        if (reward_sum) reward_sum[i] = rs;
        if (done_count) done_count[i] = dc;
    }
}

/* int fields: 0 timestep 1 total_customers 2 rejected 3 satisfied 4 total_wait_time 5 queue_length
 *             6 price_changes_this_hour 7 occupied[idx zone] 8 price_level[idx zone] 9 episodes 10 needs_reset */
void orc_parking_info(const orc_parking *h, int field, int idx, int32_t *out) {
    for (int64_t i = 0; i < h->n; ++i) {
        const parking_env *e = &h->e[i];
        int v = 0;
        switch (field) {
            case 0: v = e->t; break;
            case 1: v = e->total_customers; break;
            case 2: v = e->rejected; break;
            case 3: v = e->satisfied; break;
            case 4: v = e->total_wait; break;
            case 5: v = e->qlen; break;
            case 6: v = e->changes; break;
            case 7: for (int k = ZONE_FIRST[idx]; k < ZONE_FIRST[idx + 1]; ++k) v += e->s[k].occupied; break;
            case 8: v = e->level[idx]; break;
            case 9: v = e->episodes; break;
            case 10: v = e->needs_reset; break;
        }
        out[i] = v;
    }
}
/* float64 fields: 0 episode_revenue 1 episode_satisfaction */
void orc_parking_info64(const orc_parking *h, int field, double *out) {
    for (int64_t i = 0; i < h->n; ++i) out[i] = field == 0 ? h->e[i].revenue : h->e[i].satisfaction_sum;
}

/* Time-limit override for the short-horizon parity tests (the reference's limit is a constructor constant /
 * config value; the device ABI takes it in its config struct).  Call before reset(). */
void orc_parking_set_max_steps(orc_parking *h, int v) { h->max_steps = v; }

/* return and length of each env's last finished episode (orc_epstats.h) */
void orc_parking_episode_stats(const orc_parking *h, double *ret, int32_t *len) { eps_get(&h->eps, h->n, ret, len); }
