/* oracle/orc_snake.c — CPU restatement of SnakeEnvClassic over a batch of independent envs.
 *
 * TEST INFRASTRUCTURE ONLY (see orc_rng.h).  Follows, line by line in behaviour:
 *   reset             /root/reference/snake_env_classic/snake_env.py:49-65
 *   step              snake_env.py:67-119
 *   _place_food       snake_env.py:121-129
 *   _get_observation  snake_env.py:131-143
 * Parity pins: tests/golden/snake_*.npz + snake_kat.json (produced by tests/golden/gen/gen_snake.py
 * running the reference itself) — checked by tests/test_oracle_snake.py.
 *
 * Batch/auto-reset semantics are the build's own (the reference has no vector API); they mirror
 * include/cge_amd.h so the parity tests drive both sides with the same calls:
 *   mode 0 NEXT_STEP : a done env returns its terminal obs; the NEXT step() ignores the action,
 *                      resets it (no reseed, stream continues) and returns (reset obs, 0, 0, 0).
 *   mode 1 SAME_STEP : a done env is reset inside the same step(); obs = reset obs, the terminal
 *                      obs goes to final_obs (if not NULL).
 *   mode 2 DISABLED  : no reset at all; stepping a finished env does what the reference does.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "orc_rng.h"
#include "orc_epstats.h"

#define SNAKE_MAX_G 32

typedef struct {
    orc_mt rng;               /* family P: CPython global `random`, one private stream per env */
    int len, head;            /* ring buffer of cells, head at ring[head], tail at ring[(head+len-1)%cap] */
    int dir, score, steps;
    int food_r, food_c;       /* -1 when the board is full (reference would spin forever) */
    int needs_reset, board_full;
    int episodes;
    uint16_t ring[SNAKE_MAX_G * SNAKE_MAX_G];
    uint8_t occ[SNAKE_MAX_G * SNAKE_MAX_G];
} snake_env;

typedef struct {
    int64_t n;
    int G, mode, max_steps;
    snake_env *e; orc_eps eps;
} orc_snake;

static void place_food(orc_snake *h, snake_env *e) {      /* snake_env.py:121-129 */
    int G = h->G;
    if (e->len >= G * G) {          /* reference: infinite loop; reported, never silently skipped */
        e->food_r = e->food_c = -1;
        e->board_full = 1;
        return;
    }
    for (;;) {
        int r = orc_py_randint(&e->rng, 0, G - 1);   /* row first ... */
        int c = orc_py_randint(&e->rng, 0, G - 1);   /* ... then column */
        if (!e->occ[r * G + c]) { e->food_r = r; e->food_c = c; return; }
    }
}

static void env_reset(orc_snake *h, snake_env *e) {       /* snake_env.py:49-65 */
    int G = h->G, center = G / 2;
    memset(e->occ, 0, (size_t)G * G);
    e->len = 1; e->head = 0;
    e->ring[0] = (uint16_t)(center * G + center);
    e->occ[center * G + center] = 1;
    e->dir = 1; e->score = 0; e->steps = 0;
    e->needs_reset = 0;
    place_food(h, e);
}

static void write_obs(const orc_snake *h, const snake_env *e, int8_t *obs) {   /* :131-143 */
    int G = h->G;
    for (int i = 0; i < G * G; ++i) obs[i] = (int8_t)e->occ[i];
    if (e->food_r >= 0) obs[e->food_r * G + e->food_c] = 2;
}

/* one reference step(); returns terminated */
static int env_step(orc_snake *h, snake_env *e, int action, float *reward) {   /* :67-119 */
    int G = h->G, cap = G * G;
    int d = action - e->dir;
    if (d != 2 && d != -2) e->dir = action;                                   /* :73-74 */
    int hr = e->ring[e->head] / G, hc = e->ring[e->head] % G;
    int nr = hr, nc = hc;
    if (e->dir == 0) nr = hr - 1; else if (e->dir == 1) nc = hc + 1;          /* :77-85 */
    else if (e->dir == 2) nr = hr + 1; else nc = hc - 1;
    if (nr < 0 || nr >= G || nc < 0 || nc >= G) { *reward = -10.0f; return 1; }   /* :88-90 */
    if (e->occ[nr * G + nc]) { *reward = -10.0f; return 1; }                  /* :93-94 (tail cell counts) */
    e->head = (e->head + cap - 1) % cap;                                      /* :97 insert(0, new_head) */
    e->ring[e->head] = (uint16_t)(nr * G + nc);
    e->occ[nr * G + nc] = 1;
    e->len += 1;
    *reward = 0.0f;
    if (nr == e->food_r && nc == e->food_c) {                                 /* :101-104 */
        e->score += 1;
        *reward = 10.0f;
        place_food(h, e);
    } else {                                                                  /* :107 pop() */
        int tail = e->ring[(e->head + e->len - 1) % cap];
        e->occ[tail] = 0;
        e->len -= 1;
    }
    e->steps += 1;                                                            /* :109 */
    return e->steps >= h->max_steps;                                          /* :113-114 */
}

orc_snake *orc_snake_create(int64_t n, int grid, int mode) {
    if (n <= 0 || grid < 2 || grid > SNAKE_MAX_G || mode < 0 || mode > 2) return NULL;
    orc_snake *h = (orc_snake *)calloc(1, sizeof(*h));
    h->n = n; h->G = grid; h->mode = mode; h->max_steps = 1000;              /* :47 */
    h->e = (snake_env *)calloc((size_t)n, sizeof(snake_env));
    eps_init(&h->eps, n);
    for (int64_t i = 0; i < n; ++i) orc_py_seed(&h->e[i].rng, (uint64_t)i);
    return h;
}

void orc_snake_destroy(orc_snake *h) { if (h) { free(h->e); eps_free(&h->eps); free(h); } }

/* env i's private stream := CPython random.seed(seeds[i]) */
void orc_snake_seed(orc_snake *h, const uint64_t *seeds) {
    for (int64_t i = 0; i < h->n; ++i) orc_py_seed(&h->e[i].rng, seeds[i]);
}

void orc_snake_reset(orc_snake *h, const uint8_t *mask, int8_t *obs) {
    int cells = h->G * h->G;
    for (int64_t i = 0; i < h->n; ++i) {
        if (!mask || mask[i]) { env_reset(h, &h->e[i]); eps_clear(&h->eps, i); }
        if (obs) write_obs(h, &h->e[i], obs + i * cells);   /* every row is written, like the device ABI */
    }
}

/* returns the number of invalid actions (reference raises ValueError, snake_env.py:69-70);
 * an env with an invalid action is left untouched. */
int orc_snake_step(orc_snake *h, const int32_t *actions, int8_t *obs, float *reward,
                   uint8_t *terminated, uint8_t *truncated, int8_t *final_obs) {
    int cells = h->G * h->G, bad = 0;
    for (int64_t i = 0; i < h->n; ++i) {
        snake_env *e = &h->e[i];
        int8_t *o = obs + i * cells;
        if (h->mode == 0 && e->needs_reset) {
            { env_reset(h, e); eps_clear(&h->eps, i); }
            write_obs(h, e, o);
            reward[i] = 0.0f; terminated[i] = 0; truncated[i] = 0;
            continue;
        }
        int a = actions[i];
        if (a < 0 || a > 3) { ++bad; write_obs(h, e, o); reward[i] = 0.0f; terminated[i] = 0; truncated[i] = 0; continue; }
        float r;
        int term = env_step(h, e, a, &r);
        eps_add(&h->eps, i, (double)r);
        reward[i] = r; terminated[i] = (uint8_t)term; truncated[i] = 0;
        if (term) { e->episodes += 1; eps_done(&h->eps, i); }
        if (term && h->mode == 1) {
            if (final_obs) write_obs(h, e, final_obs + i * cells);
            { env_reset(h, e); eps_clear(&h->eps, i); }
            write_obs(h, e, o);
        } else {
            write_obs(h, e, o);
            if (term && h->mode == 0) e->needs_reset = 1;
        }
    }
    return bad;
}

/* K fused steps with the shared counter-hash action source; obs of the LAST step is written,
 * per-env reward sums and done counts are accumulated (used as bench.py's cpu_baseline leg and
 * to check the device rollout entry point). */
void orc_snake_rollout(orc_snake *h, int k_steps, uint64_t a_seed, int64_t t0, int64_t env0,
                       int8_t *obs, float *reward_sum, int32_t *done_count) {
    int cells = h->G * h->G;
    for (int64_t i = 0; i < h->n; ++i) {
        snake_env *e = &h->e[i];
        float rs = 0.0f;
        int dc = 0;
        for (int t = 0; t < k_steps; ++t) {
            if (h->mode == 0 && e->needs_reset) { { env_reset(h, e); eps_clear(&h->eps, i); } continue; }
            int a = (int)orc_hash_action(a_seed, (uint64_t)(env0 + i), (uint64_t)(t0 + t), 4, 0);
            float r;
            int term = env_step(h, e, a, &r);
            eps_add(&h->eps, i, (double)r);
            rs += r;
            if (term) {
                ++dc; e->episodes += 1; eps_done(&h->eps, i);
                if (h->mode == 1) { env_reset(h, e); eps_clear(&h->eps, i); }
                else if (h->mode == 0) e->needs_reset = 1;
            }
        }
        if (obs) write_obs(h, e, obs + i * cells);
        if (reward_sum) reward_sum[i] = rs;
        if (done_count) done_count[i] = dc;
    }
}

/* field: 0 score, 1 snake_length, 2 steps, 3 direction, 4 food_r, 5 food_c, 6 board_full, 7 episodes,
 *        8 head_r, 9 head_c, 10 needs_reset */
void orc_snake_info(const orc_snake *h, int field, int32_t *out) {
    for (int64_t i = 0; i < h->n; ++i) {
        const snake_env *e = &h->e[i];
        int v = 0;
        switch (field) {
            case 0: v = e->score; break;
            case 1: v = e->len; break;
            case 2: v = e->steps; break;
            case 3: v = e->dir; break;
            case 4: v = e->food_r; break;
            case 5: v = e->food_c; break;
            case 6: v = e->board_full; break;
            case 7: v = e->episodes; break;
            case 8: v = e->ring[e->head] / h->G; break;
            case 9: v = e->ring[e->head] % h->G; break;
            case 10: v = e->needs_reset; break;
        }
        out[i] = v;
    }
}

/* Canonical per-env state record shared with the device library's get/set_state
 * (include/cge_amd.h: cge_snake_state): all little-endian int32 unless noted.
 *   [0] len [1] dir [2] food_r [3] food_c [4] score [5] steps [6] needs_reset [7] mt_idx
 *   then uint32 mt[624], then uint16 body[G*G] (head first, unused = 0xFFFF), padded to 4 bytes. */
size_t orc_snake_state_bytes(const orc_snake *h) {
    size_t b = 8 * 4 + 624 * 4 + (size_t)h->G * h->G * 2;
    return (b + 3) & ~(size_t)3;
}

void orc_snake_get_state(const orc_snake *h, void *buf) {
    size_t rec = orc_snake_state_bytes(h);
    int cap = h->G * h->G;
    for (int64_t i = 0; i < h->n; ++i) {
        const snake_env *e = &h->e[i];
        uint8_t *p = (uint8_t *)buf + i * rec;
        int32_t hd[8] = {e->len, e->dir, e->food_r, e->food_c, e->score, e->steps, e->needs_reset, e->rng.idx};
        memcpy(p, hd, 32);
        memcpy(p + 32, e->rng.mt, 624 * 4);
        uint16_t *body = (uint16_t *)(p + 32 + 624 * 4);
        for (int k = 0; k < cap; ++k) body[k] = k < e->len ? e->ring[(e->head + k) % cap] : 0xFFFF;
    }
}

void orc_snake_set_state(orc_snake *h, const void *buf) {
    size_t rec = orc_snake_state_bytes(h);
    int cap = h->G * h->G;
    for (int64_t i = 0; i < h->n; ++i) {
        snake_env *e = &h->e[i];
        const uint8_t *p = (const uint8_t *)buf + i * rec;
        int32_t hd[8];
        memcpy(hd, p, 32);
        e->len = hd[0]; e->dir = hd[1]; e->food_r = hd[2]; e->food_c = hd[3];
        e->score = hd[4]; e->steps = hd[5]; e->needs_reset = hd[6]; e->rng.idx = hd[7];
        memcpy(e->rng.mt, p + 32, 624 * 4);
        const uint16_t *body = (const uint16_t *)(p + 32 + 624 * 4);
        memset(e->occ, 0, (size_t)cap);
        e->head = 0;
        for (int k = 0; k < e->len; ++k) { e->ring[k] = body[k]; e->occ[body[k]] = 1; }
        e->board_full = 0;
    }
}

/* Time-limit override for the short-horizon parity tests (the reference's limit is a constructor constant /
 * config value; the device ABI takes it in its config struct).  Call before reset(). */
void orc_snake_set_max_steps(orc_snake *h, int v) { h->max_steps = v; }

/* _render_rgb_array (snake_env.py:175-188): rgb[obs == 0] = (0,0,0), rgb[obs == 1] = (0,255,0), rgb[obs == 2] = (255,0,0) */
void orc_snake_render_rgb(const orc_snake *h, uint8_t *rgb) {
    int cells = h->G * h->G;
    int8_t obs[SNAKE_MAX_G * SNAKE_MAX_G];
    for (int64_t i = 0; i < h->n; ++i) {
        write_obs(h, &h->e[i], obs);
        uint8_t *o = rgb + i * cells * 3;
        for (int c = 0; c < cells; ++c) { o[3 * c] = obs[c] == 2 ? 255 : 0; o[3 * c + 1] = obs[c] == 1 ? 255 : 0; o[3 * c + 2] = 0; }
    }
}

/* return and length of each env's last finished episode (orc_epstats.h) */
void orc_snake_episode_stats(const orc_snake *h, double *ret, int32_t *len) { eps_get(&h->eps, h->n, ret, len); }
