/* oracle/orc_crypto.c — CPU restatement of CryptoTradingEnv over a batch of independent envs.
 *
 * TEST INFRASTRUCTURE ONLY (see orc_rng.h).  Follows /root/reference/crypto_trading_env/crypto_trading_env.py:
 *   TradingConfig :28-38            TechnicalIndicators :41-119        MarketSimulator :122-221
 *   reset :301-340                  step :342-398                      _execute_action/_buy/_sell :400-503
 *   _get_observation :505-561 (261 values although obs_dim says 260, :286)
 * All arithmetic is IEEE double in the reference's order (compile with -ffp-contract=off); NumPy
 * reductions (np.mean / np.std, :54-55,70-71) are restated with NumPy's pairwise summation so the
 * float64 results are bit-identical; the observation is cast to float32 at the end like
 * np.array(obs, dtype=np.float32) (:561).
 * Generators: family P = CPython global `random`, family L = NumPy legacy global `np.random`
 * (np.random.normal, :148), one private stream of each per env, both seeded by reset(seed=) (:305-307).
 * Parity pins: tests/golden/crypto_*.npz + crypto_kat.json (tests/golden/gen/gen_crypto.py runs the
 * reference itself) — checked by tests/test_oracle_crypto.py.
 *
 * Continuous actions: the reference multiplies np.float32 action components by Python floats
 * (:414-415).  Under NumPy >= 2 (NEP 50; the fixtures were produced with NumPy 2.2.6) that keeps
 * float32, so `amount`, `fee` and — after a buy — `self.cash` are np.float32 until the next sell makes
 * cash np.float64 (after which amounts are float64 too).  `cash_kind` tracks that; under NumPy 1.x
 * the same code computes everything in float64.  Discrete actions (the BASELINE config) are float64 only.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "orc_rng.h"
#include "orc_epstats.h"

#define HLEN 50
enum { BULL = 0, BEAR = 1, SIDEWAYS = 2, CRASH = 3, RECOVERY = 4 };

typedef struct {
    double initial_balance, fee_rate, slippage_rate, min_price, max_price, volatility_base, psychology_factor;
    int max_steps, continuous;
} crypto_cfg;

typedef struct {
    orc_mt P, L;
    double cash, holdings, psych, trend_strength;
    int cash_kind;             /* dtype of self.cash in the reference: 0 Python float, 1 np.float32, 2 np.float64 */
    int regime, step, needs_reset, episodes;
    int head;                  /* ring: logical candle k lives at hist[(head + k) % HLEN] */
    double hist[HLEN][5];      /* open, high, low, close, volume */
    double last_reward, last_pv;
} crypto_env;

typedef struct {
    int64_t n;
    int mode;
    crypto_cfg c;
    crypto_env *e; orc_eps eps;
} orc_crypto;

static const double VOL_MULT[5] = {1.2, 1.5, 0.8, 3.0, 2.0};           /* :190-196 */
static const double TREND[5] = {0.001, -0.001, 0.0, -0.005, 0.002};    /* :202-208 */
static const int NEXT_REGIME[5][2] = {{SIDEWAYS, CRASH}, {SIDEWAYS, RECOVERY}, {BULL, BEAR}, {RECOVERY, BEAR}, {BULL, SIDEWAYS}};   /* :168-174 */

static double clipd(double x, double lo, double hi) { return x < lo ? lo : (x > hi ? hi : x); }

/* NumPy pairwise summation of a contiguous float64 vector, n <= 128 (loops_utils.h.src) */
static double np_sum(const double *a, int n) {
    if (n < 8) {
        double res = 0.0;
        for (int i = 0; i < n; ++i) res += a[i];
        return res;
    }
    double r[8];
    int i;
    for (i = 0; i < 8; ++i) r[i] = a[i];
    for (i = 8; i < n - (n % 8); i += 8)
        for (int j = 0; j < 8; ++j) r[j] += a[i + j];
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) res += a[i];
    return res;
}

static void update_regime(crypto_env *e) {                              /* :166-186 */
    e->regime = NEXT_REGIME[e->regime][orc_py_randbelow(&e->P, 2)];      /* random.choice of 2 */
    if (e->regime == BULL || e->regime == RECOVERY) e->trend_strength = orc_py_uniform(&e->P, 0.5, 1.0);
    else if (e->regime == BEAR || e->regime == CRASH) e->trend_strength = orc_py_uniform(&e->P, -1.0, -0.5);
    else e->trend_strength = orc_py_uniform(&e->P, -0.2, 0.2);
}

static double next_price(const orc_crypto *h, crypto_env *e, double current, double volume) {   /* :132-164 */
    if (orc_mt_double(&e->P) < 0.01) update_regime(e);
    double volatility = h->c.volatility_base * VOL_MULT[e->regime];
    double drift = (e->psych - 0.5) * h->c.psychology_factor;
    double trend = TREND[e->regime] * e->trend_strength;
    double rc = orc_np_normal(&e->L, 0.0, volatility);
    double vf = 1.0 / (1.0 + volume * 0.1);
    double pct = (trend + drift + rc) * vf;
    double np_ = clipd(current * (1 + pct), h->c.min_price, h->c.max_price);
    e->psych += pct * 10;                                               /* :213-221 */
    e->psych = clipd(e->psych, 0.0, 1.0);
    e->psych += (0.5 - e->psych) * 0.01;
    return np_;
}

static void append(crypto_env *e, double o, double hi, double lo, double c, double v) {
    double *row = e->hist[e->head];        /* overwrite the oldest: append + pop(0), :359-365 */
    row[0] = o; row[1] = hi; row[2] = lo; row[3] = c; row[4] = v;
    e->head = (e->head + 1) % HLEN;
}
static const double *candle(const crypto_env *e, int k) { return e->hist[(e->head + k) % HLEN]; }

static void env_reset(const orc_crypto *h, crypto_env *e) {             /* :301-340 */
    e->cash = h->c.initial_balance; e->cash_kind = 0;
    e->holdings = 0.0; e->step = 0; e->needs_reset = 0;
    double price = 50000.0;
    e->head = 0;
    for (int k = 0; k < HLEN; ++k) {
        double volume = orc_py_uniform(&e->P, 0.5, 2.0);
        price = next_price(h, e, price, volume);
        double hi = price * orc_py_uniform(&e->P, 1.0, 1.02);
        double lo = price * orc_py_uniform(&e->P, 0.98, 1.0);
        double op = price * orc_py_uniform(&e->P, 0.99, 1.01);
        append(e, op, hi, lo, price, volume);
    }
}

/* returns 1 if a trade happened */
static int do_buy(const orc_crypto *h, crypto_env *e, double amount, int amount_f32, double price) {   /* :449-476 */
    if (amount_f32) {
        /* NumPy >= 2: a Python-float cash is a weak scalar, compared as float32 against the float32 amount */
        if ((float)amount <= 0.0f || (float)e->cash < (float)amount) return 0;
    } else if (amount <= 0 || e->cash < amount) return 0;
    double slippage = price * h->c.slippage_rate * orc_py_uniform(&e->P, 0.5, 1.5);
    double eff = price + slippage;
    if (amount_f32) {
        float a = (float)amount;
        float fee = a * (float)h->c.fee_rate;
        float net = a - fee;
        double bought = (double)net / eff;          /* np.float32 / np.float64 -> float64 */
        e->cash = (double)((float)e->cash - a);      /* Python float / np.float32 minus np.float32 -> np.float32 */
        e->cash_kind = 1;
        e->holdings += bought;
    } else {
        double fee = amount * h->c.fee_rate;
        double net = amount - fee;
        double bought = net / eff;
        e->cash -= amount;
        e->holdings += bought;
    }
    return 1;
}

static int do_sell(const orc_crypto *h, crypto_env *e, double qty, double price, int continuous) {   /* :478-503 */
    if (qty <= 0 || e->holdings < qty) return 0;
    double slippage = price * h->c.slippage_rate * orc_py_uniform(&e->P, 0.5, 1.5);
    double eff = price - slippage;
    double received = qty * eff;
    double fee = received * h->c.fee_rate;
    double net = received - fee;
    e->holdings -= qty;
    e->cash += net;                                  /* + np.float64 -> np.float64 */
    if (continuous) e->cash_kind = 2;
    return 1;
}

static double execute_action(const orc_crypto *h, crypto_env *e, int a_disc, const float *a_cont) {   /* :400-447 */
    double price = e->hist[(e->head + HLEN - 1) % HLEN][3];
    double pv0 = e->cash + e->holdings * price;
    int traded = 0;
    if (h->c.continuous) {
        float b = a_cont[0] < 0.0f ? 0.0f : (a_cont[0] > 1.0f ? 1.0f : a_cont[0]);   /* np.clip on np.float32, :414-415 */
        float s = a_cont[1] < 0.0f ? 0.0f : (a_cont[1] > 1.0f ? 1.0f : a_cont[1]);
        double buy;
        int buy_f32 = 1;
        if (e->cash_kind == 0) buy = (double)(b * (float)(e->cash * 0.1));          /* weak Python float max_buy */
        else if (e->cash_kind == 1) buy = (double)(b * ((float)e->cash * 0.1f));    /* np.float32 cash */
        else { buy = (double)b * (e->cash * 0.1); buy_f32 = 0; }                    /* np.float64 cash */
        /* holdings is the Python float 0.0 until the first buy (sell amount 0 in any dtype), np.float64 after */
        double sell = (double)s * (e->holdings * 0.1);
        if (buy > sell && buy > 0) traded = do_buy(h, e, buy, buy_f32, price);       /* :417-419 */
        else if (sell > 0) traded = do_sell(h, e, sell, price, 1);                   /* :420-422 */
    } else {
        if (a_disc == 1) traded = do_buy(h, e, e->cash * 0.05, 0, price);            /* :425-436 */
        else if (a_disc == 2) traded = do_buy(h, e, e->cash * 0.2, 0, price);
        else if (a_disc == 3) traded = do_sell(h, e, e->holdings * 0.05, price, 0);
        else if (a_disc == 4) traded = do_sell(h, e, e->holdings * 0.2, price, 0);
    }
    double pv1 = e->cash + e->holdings * price;
    double reward = pv1 - pv0;                                                      /* :440-441 */
    if (!traded) reward -= 1.0;                                                     /* :444-445 */
    return reward;
}

static double ema_mult(int period) { return 2.0 / (period + 1); }

/* _get_observation, :505-561 — 261 float32 */
static void write_obs(const orc_crypto *h, const crypto_env *e, float *obs) {
    double closes[HLEN];
    const double cur = candle(e, HLEN - 1)[3];
    int o = 0;
    for (int k = 0; k < HLEN; ++k) {
        const double *c = candle(e, k);
        closes[k] = c[3];
        for (int f = 0; f < 5; ++f) obs[o++] = (float)(c[f] / cur);                 /* :513-515 (volume divided too) */
    }
    const double pv = e->cash + e->holdings * cur;
    if (e->cash_kind == 1) obs[o++] = (float)e->cash / (float)h->c.initial_balance; /* np.float32 / Python float */
    else obs[o++] = (float)(e->cash / h->c.initial_balance);                        /* :524 */
    obs[o++] = (float)(e->holdings * cur / h->c.initial_balance);
    obs[o++] = (float)(pv / h->c.initial_balance);
    /* RSI(14), :45-61 */
    {
        double gains[14], losses[14];
        for (int j = 0; j < 14; ++j) {
            double d = closes[HLEN - 14 + j] - closes[HLEN - 15 + j];
            gains[j] = d > 0 ? d : 0.0;
            losses[j] = d < 0 ? -d : 0.0;
        }
        double ag = np_sum(gains, 14) / 14, al = np_sum(losses, 14) / 14, rsi;
        if (al == 0) rsi = 100.0;
        else { double rs = ag / al; rsi = 100 - (100 / (1 + rs)); }
        obs[o++] = (float)(rsi / 100.0);
    }
    /* MACD(12,26,9), :79-119: the 25 prefix EMAs of the reference equal the running EMAs at index i-1 */
    {
        const double mf = ema_mult(12), ms = ema_mult(26), mg = ema_mult(9);
        double ef = closes[0], es = closes[0], sig = 0.0, macd = 0.0;
        for (int k = 1; k < HLEN; ++k) {
            ef = (closes[k] * mf) + (ef * (1 - mf));
            es = (closes[k] * ms) + (es * (1 - ms));
            if (k >= 25) {
                macd = ef - es;
                sig = k == 25 ? macd : (macd * mg) + (sig * (1 - mg));
            }
        }
        double hist = macd - sig;
        double mx = closes[0], mn = closes[0];
        for (int k = 1; k < HLEN; ++k) { if (closes[k] > mx) mx = closes[k]; if (closes[k] < mn) mn = closes[k]; }
        double range = mx - mn;
        if (range > 0) { obs[o++] = (float)(macd / range); obs[o++] = (float)(sig / range); obs[o++] = (float)(hist / range); }
        else { obs[o++] = 0.0f; obs[o++] = 0.0f; obs[o++] = 0.0f; }
    }
    /* Bollinger(20, 2), :64-76 and :550-554 */
    {
        const double *w = closes + HLEN - 20;
        double sma = np_sum(w, 20) / 20;
        double dev[20];
        for (int j = 0; j < 20; ++j) { double x = w[j] - sma; dev[j] = x * x; }
        double sd = sqrt(np_sum(dev, 20) / 20);
        double upper = sma + (2 * sd), lower = sma - (2 * sd);
        obs[o++] = (float)(upper > lower ? (cur - lower) / (upper - lower) : 0.5);
        obs[o++] = (float)(sma > 0 ? (upper - lower) / sma : 0.0);
        obs[o++] = (float)(sma > 0 ? (cur - sma) / sma : 0.0);
    }
    obs[o++] = (float)e->psych;                                                     /* :559 */
}

#define CRYPTO_OBS 261

/* one reference step(); returns terminated */
static int env_step(const orc_crypto *h, crypto_env *e, int a_disc, const float *a_cont, double *reward) {   /* :342-398 */
    *reward = execute_action(h, e, a_disc, a_cont);
    double cur = e->hist[(e->head + HLEN - 1) % HLEN][3];
    double volume = orc_py_uniform(&e->P, 0.5, 2.0);
    double np_ = next_price(h, e, cur, volume);
    double hi = np_ * orc_py_uniform(&e->P, 1.0, 1.02);
    double lo = np_ * orc_py_uniform(&e->P, 0.98, 1.0);
    append(e, cur, hi, lo, np_, volume);                                            /* open = previous close, :355 */
    double pv = e->cash + e->holdings * np_;
    e->step += 1;
    e->last_pv = pv;
    return e->step >= h->c.max_steps || pv <= 0 || pv >= h->c.initial_balance * 10; /* :382-386 */
}

orc_crypto *orc_crypto_create(int64_t n, int continuous, int mode) {
    if (n <= 0 || mode < 0 || mode > 2) return NULL;
    orc_crypto *h = (orc_crypto *)calloc(1, sizeof(*h));
    h->n = n; h->mode = mode;
    h->c = (crypto_cfg){10000.0, 0.001, 0.0005, 100.0, 100000.0, 0.02, 0.1, 1000, continuous};   /* :28-38, :278 */
    h->e = (crypto_env *)calloc((size_t)n, sizeof(crypto_env));
    eps_init(&h->eps, n);
    for (int64_t i = 0; i < n; ++i) {
        crypto_env *e = &h->e[i];
        orc_py_seed(&e->P, (uint64_t)i); orc_np_seed(&e->L, (uint32_t)i);
        e->regime = SIDEWAYS; e->trend_strength = 0.0; e->psych = 0.5;              /* MarketSimulator.__init__, :125-130 */
        e->cash = h->c.initial_balance;
    }
    return h;
}
void orc_crypto_destroy(orc_crypto *h) { if (h) { free(h->e); eps_free(&h->eps); free(h); } }

/* CryptoTradingEnv(config=TradingConfig(...)) (:28-38, :252): cfg = {initial_balance, trading_fee_rate, slippage_rate, min_price,
 * max_price, volatility_base, market_psychology_factor}; call right after create (history_length stays 50).  __init__ sets
 * self.cash = config.initial_balance (:260). */
void orc_crypto_set_config(orc_crypto *h, const double *cfg) {
    h->c.initial_balance = cfg[0]; h->c.fee_rate = cfg[1]; h->c.slippage_rate = cfg[2]; h->c.min_price = cfg[3];
    h->c.max_price = cfg[4]; h->c.volatility_base = cfg[5]; h->c.psychology_factor = cfg[6];
    for (int64_t i = 0; i < h->n; ++i) h->e[i].cash = h->c.initial_balance;
}

/* reset(seed=s): random.seed(s); np.random.seed(s)  (:305-307).  The MarketSimulator is NOT re-created. */
void orc_crypto_seed(orc_crypto *h, const uint64_t *seeds) {
    for (int64_t i = 0; i < h->n; ++i) { orc_py_seed(&h->e[i].P, seeds[i]); orc_np_seed(&h->e[i].L, (uint32_t)seeds[i]); }
}

void orc_crypto_reset(orc_crypto *h, const uint8_t *mask, float *obs) {
    for (int64_t i = 0; i < h->n; ++i) {
        if (!mask || mask[i]) { env_reset(h, &h->e[i]); eps_clear(&h->eps, i); }
        if (obs) write_obs(h, &h->e[i], obs + i * CRYPTO_OBS);
    }
}

/* actions: int32 [n] (discrete) or float32 [n,2] (continuous).  reward64 (nullable) gets the float64 reward. */
int orc_crypto_step(orc_crypto *h, const void *actions, float *obs, float *reward, double *reward64,
                    uint8_t *terminated, uint8_t *truncated, float *final_obs) {
    int bad = 0;
    for (int64_t i = 0; i < h->n; ++i) {
        crypto_env *e = &h->e[i];
        float *o = obs + i * CRYPTO_OBS;
        if (h->mode == 0 && e->needs_reset) {
            { env_reset(h, e); eps_clear(&h->eps, i); }
            write_obs(h, e, o);
            reward[i] = 0.0f; if (reward64) reward64[i] = 0.0; terminated[i] = 0; truncated[i] = 0;
            continue;
        }
        int a = 0;
        const float *ac = NULL;
        if (h->c.continuous) ac = (const float *)actions + 2 * i;
        else {
            a = ((const int32_t *)actions)[i];
            if (a < 0 || a > 4) { ++bad; write_obs(h, e, o); reward[i] = 0.0f; if (reward64) reward64[i] = 0.0; terminated[i] = 0; truncated[i] = 0; continue; }
        }
        double r;
        int term = env_step(h, e, a, ac, &r);
        eps_add(&h->eps, i, (double)r);
        reward[i] = (float)r; if (reward64) reward64[i] = r;
        terminated[i] = (uint8_t)term; truncated[i] = 0;
        e->last_reward = r;
        if (term) { e->episodes += 1; eps_done(&h->eps, i); }
        if (term && h->mode == 1) {
            if (final_obs) write_obs(h, e, final_obs + i * CRYPTO_OBS);
            { env_reset(h, e); eps_clear(&h->eps, i); }
            write_obs(h, e, o);
        } else {
            write_obs(h, e, o);
            if (term && h->mode == 0) e->needs_reset = 1;
        }
    }
    return bad;
}

/* K fused steps with hash actions (discrete: hash mod 5); obs of the last step; used by the
 * cpu_baseline leg and to check the device rollout.  Observations are still assembled every step
 * (that is where the reference spends its time) unless obs == NULL. */
void orc_crypto_rollout(orc_crypto *h, int k_steps, uint64_t a_seed, int64_t t0, int64_t env0, float *obs,
                        double *reward_sum, int32_t *done_count) {
    float scratch[CRYPTO_OBS];
    for (int64_t i = 0; i < h->n; ++i) {
        crypto_env *e = &h->e[i];
        double rs = 0.0;
        int dc = 0;
        for (int t = 0; t < k_steps; ++t) {
            if (h->mode == 0 && e->needs_reset) { { env_reset(h, e); eps_clear(&h->eps, i); } continue; }
            int a = (int)orc_hash_action(a_seed, (uint64_t)(env0 + i), (uint64_t)(t0 + t), 5, 0);
            float ac[2] = {0, 0};
            if (h->c.continuous) {
                ac[0] = (float)((double)(orc_mix64(orc_mix64(a_seed + (uint64_t)(env0 + i) * 0x9E3779B97F4A7C15ull) + (uint64_t)(t0 + t) * 0xD1342543DE82EF95ull + 0) >> 40) / 8388608.0 - 1.0);
                ac[1] = (float)((double)(orc_mix64(orc_mix64(a_seed + (uint64_t)(env0 + i) * 0x9E3779B97F4A7C15ull) + (uint64_t)(t0 + t) * 0xD1342543DE82EF95ull + 1) >> 40) / 8388608.0 - 1.0);
            }
            double r;
            int term = env_step(h, e, a, ac, &r);
            eps_add(&h->eps, i, (double)r);
            rs += r;
            if (obs) write_obs(h, e, scratch);
            if (term) {
                ++dc; e->episodes += 1; eps_done(&h->eps, i);
                if (h->mode == 1) { env_reset(h, e); eps_clear(&h->eps, i); }
                else if (h->mode == 0) e->needs_reset = 1;
            }
        }
        if (obs) write_obs(h, e, obs + i * CRYPTO_OBS);
        if (reward_sum) reward_sum[i] = rs;
        if (done_count) done_count[i] = dc;
    }
}

/* field: 0 portfolio_value 1 cash 2 holdings 3 current_price 4 market_psychology 5 regime 6 step 7 trend_strength
 *        8 episodes 9 needs_reset 10 cash_kind */
void orc_crypto_info(const orc_crypto *h, int field, double *out) {
    for (int64_t i = 0; i < h->n; ++i) {
        const crypto_env *e = &h->e[i];
        double cur = candle(e, HLEN - 1)[3], v = 0;
        switch (field) {
            case 0: v = e->cash + e->holdings * cur; break;
            case 1: v = e->cash; break;
            case 2: v = e->holdings; break;
            case 3: v = cur; break;
            case 4: v = e->psych; break;
            case 5: v = e->regime; break;
            case 6: v = e->step; break;
            case 7: v = e->trend_strength; break;
            case 8: v = e->episodes; break;
            case 9: v = e->needs_reset; break;
            case 10: v = e->cash_kind; break;
        }
        out[i] = v;
    }
}

/* Canonical state record shared with the device library (include/cge_amd.h):
 *   int32[12]  {regime, step, needs_reset, cash_kind, P_idx, L_idx, has_gauss, episodes, 0,0,0,0}
 *   double[6]  {cash, holdings, psych, trend_strength, gauss, 0}
 *   uint32 P[624], uint32 L[624]   (CPython layout: words >= idx generated-but-unconsumed)
 *   double hist[50][5]             logical order, oldest first */
size_t orc_crypto_state_bytes(void) { return 12 * 4 + 6 * 8 + 2 * 624 * 4 + HLEN * 5 * 8; }

void orc_crypto_get_state(const orc_crypto *h, void *buf) {
    size_t rec = orc_crypto_state_bytes();
    for (int64_t i = 0; i < h->n; ++i) {
        const crypto_env *e = &h->e[i];
        uint8_t *p = (uint8_t *)buf + i * rec;
        int32_t hd[12] = {e->regime, e->step, e->needs_reset, e->cash_kind, e->P.idx, e->L.idx, e->L.has_gauss, e->episodes, 0, 0, 0, 0};
        double sc[6] = {e->cash, e->holdings, e->psych, e->trend_strength, e->L.gauss, 0};
        memcpy(p, hd, 48); memcpy(p + 48, sc, 48);
        memcpy(p + 96, e->P.mt, 2496); memcpy(p + 96 + 2496, e->L.mt, 2496);
        double *hh = (double *)(p + 96 + 2 * 2496);
        for (int k = 0; k < HLEN; ++k) memcpy(hh + 5 * k, candle(e, k), 40);
    }
}

void orc_crypto_set_state(orc_crypto *h, const void *buf) {
    size_t rec = orc_crypto_state_bytes();
    for (int64_t i = 0; i < h->n; ++i) {
        crypto_env *e = &h->e[i];
        const uint8_t *p = (const uint8_t *)buf + i * rec;
        int32_t hd[12]; double sc[6];
        memcpy(hd, p, 48); memcpy(sc, p + 48, 48);
        e->regime = hd[0]; e->step = hd[1]; e->needs_reset = hd[2]; e->cash_kind = hd[3];
        e->P.idx = hd[4]; e->L.idx = hd[5]; e->L.has_gauss = hd[6]; e->episodes = hd[7];
        e->cash = sc[0]; e->holdings = sc[1]; e->psych = sc[2]; e->trend_strength = sc[3]; e->L.gauss = sc[4];
        memcpy(e->P.mt, p + 96, 2496); memcpy(e->L.mt, p + 96 + 2496, 2496);
        e->head = 0;
        memcpy(e->hist, p + 96 + 2 * 2496, HLEN * 40);
    }
}

/* Time-limit override for the short-horizon parity tests (the reference's limit is a constructor constant /
 * config value; the device ABI takes it in its config struct).  Call before reset(). */
void orc_crypto_set_max_steps(orc_crypto *h, int v) { h->c.max_steps = v; }

/* return and length of each env's last finished episode (orc_epstats.h) */
void orc_crypto_episode_stats(const orc_crypto *h, double *ret, int32_t *len) { eps_get(&h->eps, h->n, ret, len); }
