/* oracle/orc_fleet.c — CPU restatement of FleetManagementEnv over a batch of independent envs.
 *
 * TEST INFRASTRUCTURE ONLY (see orc_rng.h).  Follows /root/reference/fleet_management_env/fleet_env.py:
 *   __init__ :118-183 (grid 25, 800 steps, depot (12,12), fuel stations, vehicle specs, zones), reset :185-234,
 *   step :236-276, _execute_vehicle_action :278-329, _get_new_position :331-342, _get_traffic_cost :349-361,
 *   _attempt_pickup :363-391, _attempt_dropoff :393-437, _attempt_refuel :439-448,
 *   _generate_delivery_requests :450-514, _update_traffic :516-522, _update_weather :524-528,
 *   _check_missed_deadlines :530-535, _is_terminated :537-553, _get_observation :555-593 (76 values; the
 *   declared space says 87, :151-154), _get_info :595-608.
 * Generators: family L = NumPy legacy global np.random (randint = masked rejection on 32-bit words, choice(n) =
 * randint, choice(p) = cumsum/searchsorted on a 53-bit double, random()), family P = CPython `random`
 * (random.choice over the 144 cells of a zone); both seeded by reset(seed=) (:187-189), interleaved as in the
 * reference.  Quirks kept: `missed_deadlines` grows every step for every overdue urgent delivery (:530-535);
 * positions are clamped so the "invalid move" branch is dead (:331-347).
 * Parity pins: tests/golden/fleet_*.npz + fleet_kat.json (KAT-F1) — tests/test_oracle_fleet.py.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "orc_rng.h"
#include "orc_epstats.h"

#define FOBS 76
#define MAXD 12

typedef struct { int x, y, cargo, assigned; double fuel; } vehicle;
typedef struct { int px, py, dx, dy, urgency, required, t0, t1, deadline, completed, assigned_vehicle, pickup_time; } delivery;

typedef struct {
    orc_mt P, L;
    vehicle v[3];
    delivery d[MAXD];
    int nd, traffic[5][5], timestep, missed, completed_deliveries, needs_reset, episodes;
    double weather, total_reward;
} fleet_env;

typedef struct { int64_t n; int mode, max_steps; fleet_env *e; orc_eps eps; } orc_fleet;

static const double VRANGE[3] = {80, 120, 60}, VCONS[3] = {1.0, 0.5, 2.0};   /* van, motorcycle, truck :128-132 */
static const int VCAP[3] = {3, 1, 5};

static uint32_t np_randint(orc_mt *L, int lo, int hi) {                     /* legacy randint(lo, hi): masked rejection */
    uint32_t rng = (uint32_t)(hi - lo - 1), mask = rng;
    if (rng == 0) return (uint32_t)lo;
    mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
    uint32_t v;
    while ((v = (orc_mt_next(L) & mask)) > rng) ;
    return (uint32_t)lo + v;
}
static int np_choice_p(orc_mt *L, const double *p, int n) {                  /* legacy choice(n, p=p) */
    double c[4];
    c[0] = p[0];
    for (int i = 1; i < n; ++i) c[i] = c[i - 1] + p[i];
    double last = c[n - 1];
    for (int i = 0; i < n; ++i) c[i] /= last;
    double u = orc_mt_double(L);
    int idx = 0;
    while (idx < n && c[idx] <= u) ++idx;
    return idx;
}
static void zone_cell(int zone, int k, int *x, int *y) {                    /* customer_zones :135-140, list order */
    int i = k / 12, j = k % 12;
    *x = (zone == 1 || zone == 3) ? 13 + i : i;
    *y = (zone == 2 || zone == 3) ? 13 + j : j;
}

static void update_traffic(fleet_env *e) {                                  /* :516-522 */
    static const double p3[3] = {0.6, 0.3, 0.1}, p2[2] = {0.4, 0.6};
    for (int r = 0; r < 5; ++r) for (int c = 0; c < 5; ++c) e->traffic[r][c] = np_choice_p(&e->L, p3, 3);
    for (int r = 0; r < 2; ++r) for (int c = 3; c < 5; ++c) e->traffic[r][c] = 1 + np_choice_p(&e->L, p2, 2);
}

static void generate_requests(const orc_fleet *h, fleet_env *e) {           /* :450-514 */
    static const double pu[4] = {0.3, 0.4, 0.2, 0.1}, mult[4] = {4.0, 3.0, 2.0, 1.5};
    e->nd = (int)np_randint(&e->L, 8, 13);
    for (int i = 0; i < e->nd; ++i) {
        delivery *d = &e->d[i];
        int pz = (int)np_randint(&e->L, 0, 4), dz = (int)np_randint(&e->L, 0, 4);   /* np.random.choice(list(CustomerZone)) */
        zone_cell(pz, (int)orc_py_randbelow(&e->P, 144), &d->px, &d->py);            /* random.choice(positions) */
        zone_cell(dz, (int)orc_py_randbelow(&e->P, 144), &d->dx, &d->dy);
        while (d->px == d->dx && d->py == d->dy) zone_cell(dz, (int)orc_py_randbelow(&e->P, 144), &d->dx, &d->dy);
        d->urgency = np_choice_p(&e->L, pu, 4);
        d->required = 0;                                                    /* 0 none, 1 motorcycle, 2 truck */
        if (dz == 3) d->required = 1;
        else if (dz == 2) { if (orc_mt_double(&e->L) < 0.6) d->required = 2; }
        if (dz == 1) { d->t0 = (int)np_randint(&e->L, 50, 200); d->t1 = d->t0 + 300 < 600 ? d->t0 + 300 : 600; }
        else { d->t0 = 0; d->t1 = h->max_steps; }
        int base = abs(d->px - d->dx) + abs(d->py - d->dy);
        d->deadline = (int)(base * mult[d->urgency]) + 50;
        d->completed = 0; d->assigned_vehicle = -1; d->pickup_time = -1;
    }
}

static void env_reset(const orc_fleet *h, fleet_env *e) {                   /* :185-234 */
    e->timestep = 0; e->total_reward = 0; e->completed_deliveries = 0; e->missed = 0; e->weather = 1.0; e->needs_reset = 0;
    for (int k = 0; k < 3; ++k) { e->v[k].x = 12; e->v[k].y = 12; e->v[k].fuel = VRANGE[k]; e->v[k].cargo = 0; e->v[k].assigned = -1; }
    generate_requests(h, e);
    update_traffic(e);
}

static double vehicle_action(fleet_env *e, int vid, int a) {                /* :278-329 */
    vehicle *v = &e->v[vid];
    double reward = 0;
    if (v->fuel > 0) reward -= 2;
    if (a == 0) {
    } else if (a >= 1 && a <= 4) {
        if (v->fuel >= VCONS[vid]) {
            int x = v->x, y = v->y;
            if (a == 1) y = y - 1 > 0 ? y - 1 : 0; else if (a == 2) y = y + 1 < 24 ? y + 1 : 24;
            else if (a == 3) x = x - 1 > 0 ? x - 1 : 0; else x = x + 1 < 24 ? x + 1 : 24;
            int tx = x / 5 < 4 ? x / 5 : 4, ty = y / 5 < 4 ? y / 5 : 4, lvl = e->traffic[ty][tx];
            double tc = lvl == 0 ? 1.0 : lvl == 1 ? 1.5 : 2.0;
            double cost = VCONS[vid] * e->weather * tc;
            v->x = x; v->y = y;
            double f = v->fuel - cost;
            v->fuel = f > 0 ? f : 0;                                        /* max(0, fuel - cost) */
            if (tc > 1.5) reward -= 5;
        } else reward -= 50;
    } else if (a == 5) {                                                    /* _attempt_pickup :363-391 */
        if (!(v->cargo < VCAP[vid] && v->assigned == -1)) return reward - 10;
        int best = -1;
        for (int i = 0; i < e->nd; ++i) {
            delivery *d = &e->d[i];
            if (d->px == v->x && d->py == v->y && !d->completed && d->t0 <= e->timestep && e->timestep <= d->t1 && d->assigned_vehicle == -1 &&
                (d->required == 0 || d->required == vid))                   /* van = 0 is never required; 1 motorcycle, 2 truck */
                if (best < 0 || d->urgency > e->d[best].urgency) best = i;  /* max(): first of the maxima */
        }
        if (best < 0) return reward - 10;
        v->cargo += 1; v->assigned = best;
        e->d[best].assigned_vehicle = vid; e->d[best].pickup_time = e->timestep;
        reward += 20;
    } else if (a == 6) {                                                    /* _attempt_dropoff :393-437 */
        if (v->assigned == -1) return reward - 10;
        delivery *d = &e->d[v->assigned];
        if (v->x != d->dx || v->y != d->dy) return reward - 10;
        v->cargo = v->cargo - 1 > 0 ? v->cargo - 1 : 0; v->assigned = -1;
        d->completed = 1;
        static const int ur[4] = {50, 100, 200, 200};
        int r = ur[d->urgency];
        int dt = e->timestep - d->pickup_time, opt = abs(d->px - d->dx) + abs(d->py - d->dy);
        if (dt <= opt + 2) r += 15;
        if (!(e->timestep <= d->deadline)) r -= 20 * (d->urgency + 1);
        e->completed_deliveries += 1;
        reward += r;
    } else if (a == 7) {                                                    /* _attempt_refuel :439-448 */
        int at = (v->x == 5 && v->y == 5) || (v->x == 20 && v->y == 5) || (v->x == 5 && v->y == 20);
        if (at) { if (v->fuel < VRANGE[vid]) { v->fuel = VRANGE[vid]; reward += 10; } else reward += -5; }
        else reward += -10;
    } else reward -= 10;
    return reward;
}

static void write_obs(const fleet_env *e, float *obs) {                     /* :555-593 */
    int o = 0;
    for (int k = 0; k < 3; ++k) { obs[o++] = (float)e->v[k].x; obs[o++] = (float)e->v[k].y; }
    for (int k = 0; k < 3; ++k) obs[o++] = (float)(e->v[k].fuel / VRANGE[k]);
    for (int k = 0; k < 3; ++k) obs[o++] = (float)((double)e->v[k].cargo / (double)VCAP[k]);
    for (int k = 0; k < 3; ++k) obs[o++] = (float)e->v[k].assigned;
    for (int i = 0; i < MAXD; ++i) {
        int live = i < e->nd && !e->d[i].completed;
        obs[o++] = live ? (float)e->d[i].px : -1.0f; obs[o++] = live ? (float)e->d[i].py : -1.0f;
    }
    for (int i = 0; i < MAXD; ++i) obs[o++] = (i < e->nd && !e->d[i].completed) ? (float)e->d[i].urgency : -1.0f;
    for (int r = 0; r < 5; ++r) for (int c = 0; c < 5; ++c) obs[o++] = (float)e->traffic[r][c];
}

/* returns 1 terminated | 2 truncated bits */
static int env_step(const orc_fleet *h, fleet_env *e, const int32_t *a, double *reward) {   /* :236-276 */
    double total = 0;
    for (int k = 0; k < 3; ++k) total += vehicle_action(e, k, a[k]);       /* sum(rewards): 0 + r0 + r1 + r2 */
    e->timestep += 1;
    if (e->timestep % 50 == 0) update_traffic(e);
    if (e->timestep % 100 == 0) {                                           /* _update_weather :524-528 */
        static const double w[4] = {0.8, 1.0, 1.2, 1.5}, pw[4] = {0.3, 0.5, 0.15, 0.05};
        e->weather = w[np_choice_p(&e->L, pw, 4)];
    }
    int urgent = 0, all_done = 1, fuel_out = 1;
    for (int i = 0; i < e->nd; ++i) {                                       /* _check_missed_deadlines :530-535 */
        delivery *d = &e->d[i];
        if (e->timestep > d->deadline && !d->completed && d->urgency >= 2) e->missed += 1;
        if (d->urgency >= 2) ++urgent;
        if (!d->completed) all_done = 0;
    }
    for (int k = 0; k < 3; ++k) if (e->v[k].fuel > 0) fuel_out = 0;
    e->total_reward += total;
    *reward = total;
    int term = all_done || fuel_out || (urgent > 0 && e->missed >= urgent * 0.5);   /* :537-553 */
    return (term ? 1 : 0) | (e->timestep >= h->max_steps ? 2 : 0);
}

orc_fleet *orc_fleet_create(int64_t n, int mode) {
    if (n <= 0 || mode < 0 || mode > 2) return NULL;
    orc_fleet *h = (orc_fleet *)calloc(1, sizeof(*h));
    h->n = n; h->mode = mode; h->max_steps = 800;
    h->e = (fleet_env *)calloc((size_t)n, sizeof(fleet_env));
    eps_init(&h->eps, n);
    for (int64_t i = 0; i < n; ++i) { orc_py_seed(&h->e[i].P, (uint64_t)i); orc_np_seed(&h->e[i].L, (uint32_t)i); h->e[i].weather = 1.0; }
    return h;
}
void orc_fleet_destroy(orc_fleet *h) { if (h) { free(h->e); eps_free(&h->eps); free(h); } }
void orc_fleet_seed(orc_fleet *h, const uint64_t *seeds) {
    for (int64_t i = 0; i < h->n; ++i) { orc_py_seed(&h->e[i].P, seeds[i]); orc_np_seed(&h->e[i].L, (uint32_t)seeds[i]); }
}
void orc_fleet_reset(orc_fleet *h, const uint8_t *mask, float *obs) {
    for (int64_t i = 0; i < h->n; ++i) {
        if (!mask || mask[i]) { env_reset(h, &h->e[i]); eps_clear(&h->eps, i); }
        if (obs) write_obs(&h->e[i], obs + i * FOBS);
    }
}

void orc_fleet_step(orc_fleet *h, const int32_t *actions, float *obs, float *reward, double *reward64, uint8_t *terminated,
                    uint8_t *truncated, float *final_obs) {
    for (int64_t i = 0; i < h->n; ++i) {
        fleet_env *e = &h->e[i];
        float *o = obs + i * FOBS;
        if (h->mode == 0 && e->needs_reset) {
            { env_reset(h, e); eps_clear(&h->eps, i); } write_obs(e, o);
            reward[i] = 0.0f; if (reward64) reward64[i] = 0.0; terminated[i] = 0; truncated[i] = 0;
            continue;
        }
        double r;
        int f = env_step(h, e, actions + 3 * i, &r);
        eps_add(&h->eps, i, (double)r);
        reward[i] = (float)r; if (reward64) reward64[i] = r;
        terminated[i] = (uint8_t)(f & 1); truncated[i] = (uint8_t)((f >> 1) & 1);
        if (f) { e->episodes += 1; eps_done(&h->eps, i); }
        if (f && h->mode == 1) {
            if (final_obs) write_obs(e, final_obs + i * FOBS);
            { env_reset(h, e); eps_clear(&h->eps, i); } write_obs(e, o);
        } else {
            write_obs(e, o);
            if (f && h->mode == 0) e->needs_reset = 1;
        }
    }
}

void orc_fleet_rollout(orc_fleet *h, int k_steps, uint64_t a_seed, int64_t t0, int64_t env0, float *obs, double *reward_sum,
                       int32_t *done_count) {
    for (int64_t i = 0; i < h->n; ++i) {
        fleet_env *e = &h->e[i];
        double rs = 0.0;
        int dc = 0;
        for (int t = 0; t < k_steps; ++t) {
            if (h->mode == 0 && e->needs_reset) { { env_reset(h, e); eps_clear(&h->eps, i); } continue; }
            int32_t a[3];
            for (int j = 0; j < 3; ++j) a[j] = (int32_t)orc_hash_action(a_seed, (uint64_t)(env0 + i), (uint64_t)(t0 + t), 8, (uint32_t)j);
            double r;
            int f = env_step(h, e, a, &r);
            eps_add(&h->eps, i, (double)r);
            rs += r;
            if (f) { ++dc; e->episodes += 1; eps_done(&h->eps, i); if (h->mode == 1) { env_reset(h, e); eps_clear(&h->eps, i); } else if (h->mode == 0) e->needs_reset = 1; }
        }
        if (obs) write_obs(e, obs + i * FOBS);
        if (reward_sum) reward_sum[i] = rs;
        if (done_count) done_count[i] = dc;
    }
}

/* float64 fields: 0 timestep 1 missed_deadlines 2 completed_deliveries 3 num_requests 4 weather_effect 5 total_reward
 *                 6 episodes 7 needs_reset 8+k fuel[k] 11+4k+{0,1,2,3} vehicle k {x, y, cargo, assigned} */
void orc_fleet_info(const orc_fleet *h, int field, double *out) {
    for (int64_t i = 0; i < h->n; ++i) {
        const fleet_env *e = &h->e[i];
        double v = 0;
        if (field == 0) v = e->timestep; else if (field == 1) v = e->missed; else if (field == 2) v = e->completed_deliveries;
        else if (field == 3) v = e->nd; else if (field == 4) v = e->weather; else if (field == 5) v = e->total_reward;
        else if (field == 6) v = e->episodes; else if (field == 7) v = e->needs_reset;
        else if (field < 11) v = e->v[field - 8].fuel;
        else { int k = (field - 11) / 4, w = (field - 11) % 4; v = w == 0 ? e->v[k].x : w == 1 ? e->v[k].y : w == 2 ? e->v[k].cargo : e->v[k].assigned; }
        out[i] = v;
    }
}

/* Time-limit override for the short-horizon parity tests (the reference's limit is a constructor constant /
 * config value; the device ABI takes it in its config struct).  Call before reset(). */
void orc_fleet_set_max_steps(orc_fleet *h, int v) { h->max_steps = v; }

/* return and length of each env's last finished episode (orc_epstats.h) */
void orc_fleet_episode_stats(const orc_fleet *h, double *ret, int32_t *len) { eps_get(&h->eps, h->n, ret, len); }
