/* oracle/orc_rng_export.c — exported wrappers around orc_rng.h so tests can pin the generator
 * restatement against CPython / NumPy known answers (SURVEY.md section 8c "RNG identities").
 * TEST INFRASTRUCTURE ONLY. */
#include <stdlib.h>

#include "orc_rng.h"

orc_mt *orc_mt_new(void) { return (orc_mt *)calloc(1, sizeof(orc_mt)); }
void orc_mt_free(orc_mt *s) { free(s); }
void orc_mt_py_seed(orc_mt *s, uint64_t seed) { orc_py_seed(s, seed); }
void orc_mt_np_seed(orc_mt *s, uint32_t seed) { orc_np_seed(s, seed); }
uint32_t orc_mt_next_u32(orc_mt *s) { return orc_mt_next(s); }
double orc_mt_random(orc_mt *s) { return orc_mt_double(s); }
uint32_t orc_mt_randbelow(orc_mt *s, uint32_t n) { return orc_py_randbelow(s, n); }
int orc_mt_randint(orc_mt *s, int a, int b) { return orc_py_randint(s, a, b); }
double orc_mt_uniform(orc_mt *s, double a, double b) { return orc_py_uniform(s, a, b); }
double orc_mt_normal(orc_mt *s, double loc, double scale) { return orc_np_normal(s, loc, scale); }
void orc_mt_get(const orc_mt *s, uint32_t *mt624, int *idx) {
    for (int i = 0; i < ORC_MT_N; ++i) mt624[i] = s->mt[i];
    *idx = s->idx;
}
uint32_t orc_hash_action_export(uint64_t a_seed, uint64_t env, uint64_t t, uint32_t n, uint32_t j) {
    return orc_hash_action(a_seed, env, t, n, j);
}

orc_pcg *orc_pcg_new(void) { return (orc_pcg *)calloc(1, sizeof(orc_pcg)); }
void orc_pcg_free(orc_pcg *g) { free(g); }
void orc_pcg_seed_export(orc_pcg *g, uint64_t seed) { orc_pcg_seed(g, seed); }
uint64_t orc_pcg_next64_export(orc_pcg *g) { return orc_pcg_next64(g); }
double orc_pcg_random_export(orc_pcg *g) { return orc_pcg_double(g); }
double orc_pcg_uniform_export(orc_pcg *g, double lo, double hi) { return orc_pcg_uniform(g, lo, hi); }
int64_t orc_pcg_integers_export(orc_pcg *g, int64_t lo, int64_t hi) { return orc_pcg_integers(g, lo, hi); }
double orc_pcg_normal_export(orc_pcg *g, double loc, double scale) { return orc_pcg_normal(g, loc, scale); }
int orc_pcg_choice4_export(orc_pcg *g, const double *p) { return orc_pcg_choice4(g, p); }
void orc_pcg_state_export(const orc_pcg *g, uint64_t *out4) {
    out4[0] = (uint64_t)(g->state >> 64); out4[1] = (uint64_t)g->state; out4[2] = (uint64_t)(g->inc >> 64); out4[3] = (uint64_t)g->inc;
}
