/* oracle/orc_climate.c — CPU restatement of SmartClimateEnv over a batch of independent envs.
 *
 * TEST INFRASTRUCTURE ONLY (see orc_rng.h).  Follows /root/reference/smartclimate_rl-main/smartclimate/:
 *   env.py    __init__ :17-37, _init_state :48-61, reset :63-72, _get_obs :74-83, step :85-116
 *   utils.py  get_outside_temp :5-13, update_occupancy :15-22, room_temp_dynamics :24-28, calculate_reward :30-50
 * Generator: family D — a private np.random.default_rng(seed) per env (PCG64; uniform, Lemire integers on
 * buffered 32-bit draws, ziggurat normal, choice(p) = cdf + random() + searchsorted(right)).
 * Parity pins: tests/golden/climate_hash.npz + climate_kat.json (KAT-K1) — tests/test_oracle_climate.py.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "orc_rng.h"
#include "orc_epstats.h"

#define COBS 9

typedef struct {
    orc_pcg g;
    double room_temp, outside_temp, ac_setting, energy_usage, total_reward;
    int num_people, step, lights[4], comfort_time, needs_reset, episodes;
} climate_env;

typedef struct { int64_t n; int mode, max_occupancy, episode_minutes; climate_env *e; orc_eps eps; } orc_climate;

static double outside_temp(double tod, orc_pcg *g) {                      /* utils.py:5-13 */
    double base = (0 <= tod && tod < 8) ? 25 : (8 <= tod && tod < 16) ? 45 : 35;
    return orc_pcg_normal(g, base, 5);
}

static void init_state(const orc_climate *h, climate_env *e) {            /* env.py:48-61 */
    e->room_temp = orc_pcg_uniform(&e->g, 22.0, 26.0);
    e->num_people = (int)orc_pcg_integers(&e->g, 0, h->max_occupancy + 1);
    e->outside_temp = outside_temp(0.0, &e->g);
    e->ac_setting = 24.0;
    memset(e->lights, 0, sizeof e->lights);
    e->total_reward = 0.0; e->comfort_time = 0; e->energy_usage = 0.0;
}

static void env_reset(const orc_climate *h, climate_env *e) { e->step = 0; e->needs_reset = 0; init_state(h, e); }   /* :63-72 */

static void write_obs(const climate_env *e, float *obs) {                 /* env.py:74-83 */
    obs[0] = (float)e->room_temp;
    obs[1] = (float)e->num_people;
    obs[2] = (float)((double)(e->step % 1440) / 60.0);
    obs[3] = (float)e->outside_temp;
    obs[4] = (float)e->ac_setting;
    for (int k = 0; k < 4; ++k) obs[5 + k] = (float)e->lights[k];
}

static int env_step(const orc_climate *h, climate_env *e, float ac_in, const int8_t *lights, double *reward) {   /* env.py:85-116 */
    float acf = ac_in < 16.0f ? 16.0f : (ac_in > 32.0f ? 32.0f : ac_in);   /* np.clip on np.float32, then float() */
    e->ac_setting = (double)acf;
    for (int k = 0; k < 4; ++k) e->lights[k] = lights[k];
    e->step += 1;
    double tod = (double)(e->step % 1440) / 60.0;
    e->outside_temp = outside_temp(tod, &e->g);
    static const double p_day[4] = {0.1, 0.3, 0.4, 0.2}, p_night[4] = {0.2, 0.4, 0.3, 0.1};   /* utils.py:15-22 */
    int change;
    if (9 <= tod && tod < 18) change = orc_pcg_choice4(&e->g, p_day) - 1;      /* [-1, 0, 1, 2] */
    else change = orc_pcg_choice4(&e->g, p_night) - 2;                          /* [-2, -1, 0, 1] */
    int np_ = e->num_people + change;
    e->num_people = np_ < 0 ? 0 : (np_ > h->max_occupancy ? h->max_occupancy : np_);
    double prev = e->room_temp;                                           /* utils.py:24-28 */
    double temp = prev + 0.1 * (e->outside_temp - prev) + 0.2 * (e->ac_setting - prev) + e->num_people * 1.0;
    e->room_temp = temp < 10 ? 10 : (temp > 50 ? 50 : temp);
    double T = e->room_temp, comfort;                                     /* utils.py:30-50 */
    if (20 <= T && T <= 24) comfort = 10;
    else if (18 <= T && T <= 26) comfort = 5;
    else if (16 <= T && T <= 28) comfort = 0;
    else comfort = -15 * fabs(T - 22);
    double ac_pen = -0.5 * fabs(e->ac_setting - e->outside_temp);
    int required = (e->num_people + 1) / 2; if (required > 4) required = 4;   /* min(4, ceil(n/2)) */
    int on = e->lights[0] + e->lights[1] + e->lights[2] + e->lights[3];
    int light_pen = -1 * (on - required > 0 ? on - required : 0);
    double r = comfort + ac_pen + light_pen;
    e->total_reward += r;
    if (20 <= T && T <= 24) e->comfort_time += 1;
    e->energy_usage += fabs(e->ac_setting - e->outside_temp) + on;
    *reward = r;
    return e->step >= h->episode_minutes;
}

orc_climate *orc_climate_create(int64_t n, int mode) {
    if (n <= 0 || mode < 0 || mode > 2) return NULL;
    orc_climate *h = (orc_climate *)calloc(1, sizeof(*h));
    h->n = n; h->mode = mode; h->max_occupancy = 8; h->episode_minutes = 1440;
    h->e = (climate_env *)calloc((size_t)n, sizeof(climate_env));
    eps_init(&h->eps, n);
    for (int64_t i = 0; i < n; ++i) orc_pcg_seed(&h->e[i].g, (uint64_t)i);
    return h;
}
void orc_climate_destroy(orc_climate *h) { if (h) { free(h->e); eps_free(&h->eps); free(h); } }
/* reset(seed=s): self.rng = np.random.default_rng(s) */
void orc_climate_seed(orc_climate *h, const uint64_t *seeds) { for (int64_t i = 0; i < h->n; ++i) orc_pcg_seed(&h->e[i].g, seeds[i]); }

void orc_climate_reset(orc_climate *h, const uint8_t *mask, float *obs) {
    for (int64_t i = 0; i < h->n; ++i) {
        if (!mask || mask[i]) { env_reset(h, &h->e[i]); eps_clear(&h->eps, i); }
        if (obs) write_obs(&h->e[i], obs + i * COBS);
    }
}

void orc_climate_step(orc_climate *h, const float *ac_temp, const int8_t *lights, float *obs, float *reward, double *reward64,
                      uint8_t *terminated, uint8_t *truncated, float *final_obs) {
    for (int64_t i = 0; i < h->n; ++i) {
        climate_env *e = &h->e[i];
        float *o = obs + i * COBS;
        if (h->mode == 0 && e->needs_reset) {
            { env_reset(h, e); eps_clear(&h->eps, i); } write_obs(e, o);
            reward[i] = 0.0f; if (reward64) reward64[i] = 0.0; terminated[i] = 0; truncated[i] = 0;
            continue;
        }
        double r;
        int term = env_step(h, e, ac_temp[i], lights + 4 * i, &r);
        eps_add(&h->eps, i, (double)r);
        reward[i] = (float)r; if (reward64) reward64[i] = r;
        terminated[i] = (uint8_t)term; truncated[i] = 0;
        if (term) { e->episodes += 1; eps_done(&h->eps, i); }
        if (term && h->mode == 1) {
            if (final_obs) write_obs(e, final_obs + i * COBS);
            { env_reset(h, e); eps_clear(&h->eps, i); } write_obs(e, o);
        } else {
            write_obs(e, o);
            if (term && h->mode == 0) e->needs_reset = 1;
        }
    }
}

/* hash actions: ac = float32(16 + 16*(hash(j=0) >> 40) / 2^24), lights[k] = hash(n=2, j=1+k) */
void orc_climate_hash_action(uint64_t a_seed, uint64_t env, uint64_t t, float *ac, int8_t *lights) {
    uint64_t u = orc_mix64(orc_mix64(a_seed + env * 0x9E3779B97F4A7C15ull) + t * 0xD1342543DE82EF95ull + 0) >> 40;
    *ac = (float)(16.0 + 16.0 * ((double)u / 16777216.0));
    for (int k = 0; k < 4; ++k) lights[k] = (int8_t)orc_hash_action(a_seed, env, t, 2, (uint32_t)(1 + k));
}

void orc_climate_rollout(orc_climate *h, int k_steps, uint64_t a_seed, int64_t t0, int64_t env0, float *obs,
                         double *reward_sum, int32_t *done_count) {
    for (int64_t i = 0; i < h->n; ++i) {
        climate_env *e = &h->e[i];
        double rs = 0.0;
        int dc = 0;
        for (int t = 0; t < k_steps; ++t) {
            if (h->mode == 0 && e->needs_reset) { { env_reset(h, e); eps_clear(&h->eps, i); } continue; }
            float ac; int8_t li[4];
            orc_climate_hash_action(a_seed, (uint64_t)(env0 + i), (uint64_t)(t0 + t), &ac, li);
            double r;
            int term = env_step(h, e, ac, li, &r);
            eps_add(&h->eps, i, (double)r);
            rs += r;
            if (term) { ++dc; e->episodes += 1; eps_done(&h->eps, i); if (h->mode == 1) { env_reset(h, e); eps_clear(&h->eps, i); } else if (h->mode == 0) e->needs_reset = 1; }
        }
        if (obs) write_obs(e, obs + i * COBS);
        if (reward_sum) reward_sum[i] = rs;
        if (done_count) done_count[i] = dc;
    }
}

/* float64 fields: 0 room_temp 1 outside_temp 2 ac_setting 3 energy_usage 4 total_reward 5 num_people 6 step
 *                 7 comfort_time 8 episodes 9 needs_reset */
void orc_climate_info(const orc_climate *h, int field, double *out) {
    for (int64_t i = 0; i < h->n; ++i) {
        const climate_env *e = &h->e[i];
        double v = 0;
        switch (field) {
            case 0: v = e->room_temp; break; case 1: v = e->outside_temp; break; case 2: v = e->ac_setting; break;
            case 3: v = e->energy_usage; break; case 4: v = e->total_reward; break; case 5: v = e->num_people; break;
            case 6: v = e->step; break; case 7: v = e->comfort_time; break; case 8: v = e->episodes; break; case 9: v = e->needs_reset; break;
        }
        out[i] = v;
    }
}

/* Time-limit override for the short-horizon parity tests (the reference's limit is a constructor constant /
 * config value; the device ABI takes it in its config struct).  Call before reset(). */
void orc_climate_set_max_steps(orc_climate *h, int v) { h->episode_minutes = v; }
/* SmartClimateEnv(max_occupancy=...) (env.py:18,26): bounds the reset draw integers(0, max_occupancy + 1) (:50) and the clip in
 * update_occupancy (utils.py:21).  Call right after create. */
void orc_climate_set_max_occupancy(orc_climate *h, int v) { h->max_occupancy = v; }

/* return and length of each env's last finished episode (orc_epstats.h) */
void orc_climate_episode_stats(const orc_climate *h, double *ret, int32_t *len) { eps_get(&h->eps, h->n, ret, len); }
