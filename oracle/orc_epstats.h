/* oracle/orc_epstats.h — per-env episode statistics (TEST INFRASTRUCTURE ONLY, like everything under oracle/).
 *
 * What gymnasium.wrappers.vector.RecordEpisodeStatistics computes around a vector env and what the reference's training
 * scripts consume as episode_return_mean / episode_len_mean (smart_parking_env/examples/training.py:55): the float64 sum of
 * an episode's rewards in step order and the number of env steps it took (a NEXT_STEP reset step is not a step of any
 * episode).  The statistics of the LAST finished episode of each env stay readable until the next one finishes. */
#ifndef ORC_EPSTATS_H
#define ORC_EPSTATS_H
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct { double *acc, *last_r; int32_t *len, *last_l; } orc_eps;

static inline void eps_init(orc_eps *s, int64_t n) {
    s->acc = (double *)calloc((size_t)n, sizeof(double)); s->last_r = (double *)calloc((size_t)n, sizeof(double));
    s->len = (int32_t *)calloc((size_t)n, sizeof(int32_t)); s->last_l = (int32_t *)calloc((size_t)n, sizeof(int32_t));
}
static inline void eps_free(orc_eps *s) { free(s->acc); free(s->last_r); free(s->len); free(s->last_l); }
static inline void eps_add(orc_eps *s, int64_t i, double reward) { s->acc[i] += reward; s->len[i] += 1; }
/* publish at the step that ends an episode; the accumulators restart where the env is re-initialised (eps_clear next to every
 * env_reset), so a finished env that is stepped on WITHOUT a reset (DISABLED mode) keeps accumulating, like the envs' own
 * total_reward fields do */
static inline void eps_done(orc_eps *s, int64_t i) { s->last_r[i] = s->acc[i]; s->last_l[i] = s->len[i]; }
static inline void eps_clear(orc_eps *s, int64_t i) { s->acc[i] = 0.0; s->len[i] = 0; }
static inline void eps_get(const orc_eps *s, int64_t n, double *ret, int32_t *len) {
    if (ret) memcpy(ret, s->last_r, (size_t)n * sizeof(double));
    if (len) memcpy(len, s->last_l, (size_t)n * sizeof(int32_t));
}
#endif
