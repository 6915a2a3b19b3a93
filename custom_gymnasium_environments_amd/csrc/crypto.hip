// crypto.hip — batched CryptoTradingEnv for MI355X (gfx950): kernels + C ABI (include/cge_amd.h).
//
// Re-expresses /root/reference/crypto_trading_env/crypto_trading_env.py for N independent instances,
// one lane per env:  MarketSimulator.generate_next_price :132-164, _update_market_regime :166-186,
// reset :301-340, step :342-398, _execute_action/_buy/_sell :400-503, _get_observation :505-561 with
// TechnicalIndicators rsi :45-61, bollinger_bands :64-76, macd/_ema :79-119.
//
// Numerics: every state quantity the reference keeps in float64 is float64 here, evaluated in the
// reference's operation order with contraction off, so trajectories track the CPU bit for bit except
// where a libm result differs in its last place (device log in the polar-method gauss).  NumPy's
// pairwise summation order is reproduced for np.mean / np.std.  The observation is cast to float32 at
// the end, as np.array(obs, dtype=np.float32) does; the 250 history ratios are computed as
// x * (1/close) in float64 (differs from x/close by < 1e-16 relative, i.e. at most one float32 ulp
// once in ~5e8 values) and O/H/L/V history is stored in float32.
//
// HBM layout (per env): 64 B of scalars in four uint4 columns (SoA), 50 closes as float64
// `closes[slot][env]` and 50 {open,high,low,volume} float4 `ohlv[slot][env]`, two MT19937 blocks.
// The 50-candle window is a ring whose phase is the SAME for every env of a handle (each step()
// appends exactly one candle; a reset rewrites all 50 slots and adopts the batch phase), so slot
// indices are wave-uniform and every history access is a coalesced 8- or 16-byte-per-lane stream.
// Dominant traffic per env-step: 1200 B history read + 1044 B obs write (+ 24 B new candle, 128 B
// scalars, RNG words) — HBM-bound, no reuse, no MFMA-shaped work.
//
// Kernels: resident_kernel (step() and the fused rollout: four waves over 64 envs whose 50-candle window lives in LDS, see below),
// reset_kernel, init_kernel, info_kernel.  Generator words are twisted a 32-word chunk at a time AHEAD of the cursors
// (cge_device.hpp: mt_make_ready), so a step's draws are plain loads of ready words and nothing is written back per draw.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "cge_device.hpp"
#include "cge_host.hpp"

namespace cge {
namespace crypto {

constexpr int HLEN = 50;
constexpr int OBS = 261;
constexpr int CH = 12;            // candles per chunk: 60 floats = 15 sixteen-byte stores per row
constexpr int ROW = CH * 5 + 1;   // LDS row stride (dwords), odd
constexpr int WPC = 10;           // P words of a step's common path: trade slippage + volume + regime test + high + low = 5 doubles
constexpr int WP = 16;            // P words handed to a step: the common path's and, behind them, what a regime switch (1 % of the steps) draws
constexpr int WL = 16;            // L words per gaussian pair: four polar-method attempts (a fifth is needed by 0.2 % of the pairs)
constexpr int MAX_STEPS_LIMIT = 16383;   // `step` has 14 bits in the record
constexpr int BLOCK = 64;
enum { BULL = 0, BEAR = 1, SIDEWAYS = 2, CRASH = 3, RECOVERY = 4 };

struct Cfg {
    double initial_balance, fee, slip, minp, maxp, volb, psyf;
    int32_t max_steps, continuous;
};

struct Params {
    uint4 *scal;          // [4][n]
    double *closes;       // [50][n]
    float4 *ohlv;         // [50][n]
    uint32_t *mtP, *mtL;  // [n][MT_STRIDE]
    int64_t n, env0;
    Cfg cfg;
    int32_t mode, phase;  // phase: slot of the OLDEST candle (= where the next one is written)
    const void *actions;
    const uint8_t *mask;
    float *obs, *final_obs;
    FinalSeg fin;          // fused rollouts (SAME_STEP): terminal rows compacted per workgroup of 64 envs (cge_crypto_rollout_final_obs); rows nullable
    float *reward;
    uint8_t *terminated, *truncated;
    int32_t k_steps;
    uint64_t a_seed;
    int64_t t0, obs_step_stride;
    double *reward_sum;
    int32_t *done_count;
    double *ep_ret;       // episode statistics (cge_crypto_episode_stats), nullable
    int32_t *ep_len;
};

__host__ __device__ __forceinline__ double mk_double(uint32_t lo, uint32_t hi) { const uint64_t u = ((uint64_t)hi << 32) | lo; double x; memcpy(&x, &u, 8); return x; }
__host__ __device__ __forceinline__ uint32_t lo32(double x) { uint64_t u; memcpy(&u, &x, 8); return (uint32_t)u; }
__host__ __device__ __forceinline__ uint32_t hi32(double x) { uint64_t u; memcpy(&u, &x, 8); return (uint32_t)(u >> 32); }
__device__ __forceinline__ double u53(uint32_t a, uint32_t b) { return ((a >> 5) * 67108864.0 + (b >> 6)) / 9007199254740992.0; }
__device__ __forceinline__ double clipd(double x, double lo, double hi) { return x < lo ? lo : (x > hi ? hi : x); }

// record: four uint4 (column c of env i at scal[c*n + i])
//   a: cash | holdings     b: close | psych     c: trend | cached gaussian
//   d.x: step:14 | regime:3 @14 | has_gauss @17 | cash_kind:2 @18 | needs_reset @20 | episodes[10:0] @21
//   d.y: ppos:10 | P ready mark:5 @10 | lpos:10 @15 | L ready mark:5 @25 | episodes[12:11] @30      d.zw: episode return (float64)
// ready mark: words [pos, pretw) of the stream are already twisted; 5-bit code of cge_device.hpp (mt_ready_encode)
struct Env {
    double cash, holdings, close, psych, trend, gauss;
    double ep_return;                      // float64 sum of the running episode's rewards, step order
    uint32_t step, regime, has_gauss, cash_kind, needs_reset, episodes;   // episodes: 13 bits (saturates at 8,191)
    uint32_t ppos, ppretw, lpos, lpretw;
    static constexpr uint32_t MAX_EPISODES = 0x1FFFu;

    __host__ __device__ __forceinline__ void unpack(const uint4 &a, const uint4 &b, const uint4 &c, const uint4 &d) {
        cash = mk_double(a.x, a.y); holdings = mk_double(a.z, a.w);
        close = mk_double(b.x, b.y); psych = mk_double(b.z, b.w);
        trend = mk_double(c.x, c.y); gauss = mk_double(c.z, c.w);
        step = d.x & 0x3fffu; regime = (d.x >> 14) & 7u; has_gauss = (d.x >> 17) & 1u;
        cash_kind = (d.x >> 18) & 3u; needs_reset = (d.x >> 20) & 1u;
        ppos = d.y & 1023u; ppretw = mt_ready_decode((d.y >> 10) & 31u);
        lpos = (d.y >> 15) & 1023u; lpretw = mt_ready_decode((d.y >> 25) & 31u);
        episodes = (d.x >> 21) | ((d.y >> 30) << 11);
        ep_return = mk_double(d.z, d.w);
    }
    __host__ __device__ __forceinline__ void pack(uint4 &a, uint4 &b, uint4 &c, uint4 &d) const {
        a = make_uint4(lo32(cash), hi32(cash), lo32(holdings), hi32(holdings));
        b = make_uint4(lo32(close), hi32(close), lo32(psych), hi32(psych));
        c = make_uint4(lo32(trend), hi32(trend), lo32(gauss), hi32(gauss));
        const uint32_t ep = episodes < MAX_EPISODES ? episodes : MAX_EPISODES;
        // a ready mark at or below the cursor means "nothing ready": 0 says the same and always fits the code
        const uint32_t pq = ppretw > ppos ? mt_ready_encode(ppretw) : 0u, lq = lpretw > lpos ? mt_ready_encode(lpretw) : 0u;
        d = make_uint4(step | (regime << 14) | (has_gauss << 17) | (cash_kind << 18) | (needs_reset << 20) | ((ep & 2047u) << 21),
                       ppos | (pq << 10) | (lpos << 15) | (lq << 25) | ((ep >> 11) << 30), lo32(ep_return), hi32(ep_return));
    }
    __device__ __forceinline__ void load(const uint4 *__restrict__ s, int64_t n, int64_t i) {
        const uint4 a = s[i], b = s[n + i], c = s[2 * n + i], d = s[3 * n + i];
        unpack(a, b, c, d);
    }
    __device__ __forceinline__ void store(uint4 *__restrict__ s, int64_t n, int64_t i) const {
        uint4 a, b, c, d;
        pack(a, b, c, d);
        s[i] = a; s[n + i] = b; s[2 * n + i] = c; s[3 * n + i] = d;
    }
};

__device__ __forceinline__ double vol_mult(uint32_t r) { return r == BULL ? 1.2 : r == BEAR ? 1.5 : r == SIDEWAYS ? 0.8 : r == CRASH ? 3.0 : 2.0; }       // :190-196
__device__ __forceinline__ double base_trend(uint32_t r) { return r == BULL ? 0.001 : r == BEAR ? -0.001 : r == SIDEWAYS ? 0.0 : r == CRASH ? -0.005 : 0.002; }   // :202-208

// _update_market_regime :166-186, draws from the serial P stream (1 % of steps)
__device__ __forceinline__ void update_regime_with(Env &e, uint32_t pick, double u) {   // pick: random.choice of the two successors; u: the trend draw
    const uint32_t r = e.regime;
    const uint32_t nx0 = r == BULL ? SIDEWAYS : r == BEAR ? SIDEWAYS : r == SIDEWAYS ? BULL : r == CRASH ? RECOVERY : BULL;
    const uint32_t nx1 = r == BULL ? CRASH : r == BEAR ? RECOVERY : r == SIDEWAYS ? BEAR : r == CRASH ? BEAR : SIDEWAYS;
    e.regime = pick ? nx1 : nx0;
    if (e.regime == BULL || e.regime == RECOVERY) e.trend = 0.5 + (1.0 - 0.5) * u;
    else if (e.regime == BEAR || e.regime == CRASH) e.trend = -1.0 + (-0.5 - -1.0) * u;
    else e.trend = -0.2 + (0.2 - -0.2) * u;
}
template <class STREAM>
__device__ __forceinline__ void update_regime(Env &e, STREAM &sp) {
    const uint32_t pick = sp.randbelow(2u, 2);
    update_regime_with(e, pick, sp.random53());
}

// legacy_gauss (polar method) from the serial L stream
template <class STREAM>
__device__ __forceinline__ double gauss_serial(Env &e, STREAM &sl) {
    if (e.has_gauss) {
        e.has_gauss = 0;
        const double g = e.gauss;
        e.gauss = 0.0;
        return g;
    }
    double x1, x2, r2;
    do {
        x1 = 2.0 * sl.random53() - 1.0;
        x2 = 2.0 * sl.random53() - 1.0;
        r2 = x1 * x1 + x2 * x2;
    } while (r2 >= 1.0 || r2 == 0.0);
    const double f = sqrt(-2.0 * log(r2) / r2);
    e.gauss = f * x1;
    e.has_gauss = 1;
    return f * x2;
}

// generate_next_price :132-164 after the regime test, given this step's gaussian
__device__ __forceinline__ double price_update(Env &e, const Cfg &c, double current, double volume, double g) {
    const double volatility = c.volb * vol_mult(e.regime);
    const double drift = (e.psych - 0.5) * c.psyf;
    const double trend = base_trend(e.regime) * e.trend;
    const double rc = 0.0 + volatility * g;                       // np.random.normal(0, volatility)
    const double vf = 1.0 / (1.0 + volume * 0.1);
    const double pct = (trend + drift + rc) * vf;
    const double np_ = clipd(current * (1.0 + pct), c.minp, c.maxp);
    e.psych += pct * 10.0;                                        // :213-221
    e.psych = clipd(e.psych, 0.0, 1.0);
    e.psych += (0.5 - e.psych) * 0.01;
    return np_;
}

// reset :301-340 — once per episode: the 50-candle history is re-simulated (~700 draws over both streams).  The draws come
// from LDS-parked windows that borrow the lane's row of the obs tile (idle at this point: 32 words of the CPython stream +
// 16 of the NumPy-legacy one, 48 <= 51 dwords): through the serial MtStream every draw was its own memory round trip, a
// ~270-us chain that every wave with one finishing env paid in that step (and 19.8 ms for a full 1M-env reset()).
// Candle k of the fresh history goes to slot (phase + k) % 50: the env adopts the batch-wide ring phase.
constexpr int RW_P = 32, RW_L = 16;
static_assert(RW_P + RW_L <= ROW, "the reset's draw windows live in the lane's obs-tile row");
// Where the 50-candle window lives.  HistGlobal: the [50][N] arrays in HBM (reset()).  HistLds: the resident kernel's copy in LDS
// ([50][64] per workgroup, loaded once per launch), with every new candle also written through to HBM.
struct HistGlobal {
    double *closes;
    float4 *ohlv;
    int64_t n, i;
    __device__ __forceinline__ double close(int slot) const { return closes[(int64_t)slot * n + i]; }
    __device__ __forceinline__ float4 rest(int slot) const { return ohlv[(int64_t)slot * n + i]; }
    __device__ __forceinline__ void put(int slot, double c, float4 o) const { closes[(int64_t)slot * n + i] = c; ohlv[(int64_t)slot * n + i] = o; }
};
struct HistLds {
    // [50][64] with the env index XOR-swizzled by the slot: a wave reading one slot for its 64 envs (lane = env) and a wave reading
    // 50 slots of ONE env (lane = candle, the row writer) are both spread over the banks
    double *lcb;
    float4 *lob;
    uint32_t lane;
    HistGlobal g;
    static __device__ __forceinline__ int at(int slot, uint32_t env) { return slot * 64 + (int)(env ^ ((uint32_t)slot & 31u)); }
    __device__ __forceinline__ double close(int slot) const { return lcb[at(slot, lane)]; }
    __device__ __forceinline__ float4 rest(int slot) const { return lob[at(slot, lane)]; }
    __device__ __forceinline__ void put_lds(int slot, double c, float4 o) const { lcb[at(slot, lane)] = c; lob[at(slot, lane)] = o; }
    __device__ __forceinline__ void put(int slot, double c, float4 o) const { put_lds(slot, c, o); g.put(slot, c, o); }
};

template <class SP, class SL, class H>
__device__ __forceinline__ void reset_body(Env &e, const Params &p, H &hist, int phase, SP &sp, SL &sl) {
    e.cash = p.cfg.initial_balance;
    e.cash_kind = 0;
    e.holdings = 0.0;
    e.step = 0;
    e.needs_reset = 0;
    e.ep_return = 0.0;
    double price = 50000.0;
    int slot = phase;
#pragma unroll 1
    for (int k = 0; k < HLEN; ++k) {
        sp.ensure(14);                                         // 10 words per candle + a regime change (refills converge across the lanes that reset)
        sl.ensure(8);
        const double volume = 0.5 + (2.0 - 0.5) * sp.random53();
        if (sp.random53() < 0.01) update_regime(e, sp);
        const double g = gauss_serial(e, sl);
        price = price_update(e, p.cfg, price, volume, g);
        const double hi = price * (1.0 + (1.02 - 1.0) * sp.random53());
        const double lo = price * (0.98 + (1.0 - 0.98) * sp.random53());
        const double op = price * (0.99 + (1.01 - 0.99) * sp.random53());
        hist.put(slot, price, make_float4((float)op, (float)hi, (float)lo, (float)volume));
        slot = slot + 1 == HLEN ? 0 : slot + 1;
    }
    e.close = price;
}
__device__ __forceinline__ void do_reset(Env &e, const Params &p, int64_t i, int phase, uint32_t *lds_row) {
    LdsDrawsCall<RW_P> sp(lds_row, p.mtP + i * MT_STRIDE, e.ppos, e.ppretw);
    LdsDrawsCall<RW_L> sl(lds_row + RW_P, p.mtL + i * MT_STRIDE, e.lpos, e.lpretw);
    HistGlobal hist{p.closes, p.ohlv, p.n, i};
    reset_body(e, p, hist, phase, sp, sl);
    sp.flush(); sl.flush();
    e.ppos = sp.pos; e.ppretw = sp.pretw; e.lpos = sl.pos; e.lpretw = sl.pretw;
}

// _execute_buy :449-476.  amount_f32: NumPy>=2 keeps np.float32 amounts in continuous mode (NEP 50)
__device__ __forceinline__ bool buy_guard(const Env &e, double amount, bool amount_f32) {
    if (amount_f32) return !((float)amount <= 0.0f || (float)e.cash < (float)amount);
    return !(amount <= 0.0 || e.cash < amount);
}
__device__ __forceinline__ void buy_apply(Env &e, const Cfg &c, double amount, bool amount_f32, double price, double u) {
    const double slippage = price * c.slip * (0.5 + (1.5 - 0.5) * u);
    const double eff = price + slippage;
    if (amount_f32) {
        const float a = (float)amount;
        const float fee = a * (float)c.fee;
        const float net = a - fee;
        const double bought = (double)net / eff;
        e.cash = (double)((float)e.cash - a);
        e.cash_kind = 1;
        e.holdings += bought;
    } else {
        const double fee = amount * c.fee;
        const double net = amount - fee;
        e.cash -= amount;
        e.holdings += net / eff;
    }
}
__device__ __forceinline__ void sell_apply(Env &e, const Cfg &c, double qty, double price, double u) {   // :478-503
    const double slippage = price * c.slip * (0.5 + (1.5 - 0.5) * u);
    const double eff = price - slippage;
    const double received = qty * eff;
    const double fee = received * c.fee;
    e.holdings -= qty;
    e.cash += received - fee;
    if (c.continuous) e.cash_kind = 2;
}

// One reference step() (:342-398) is split between the waves of the resident kernel.  market_step() is the first half, wave A's: the
// trade (_execute_action :400-447) and the CPython-stream draws of the step — [slippage] volume regime-test high low, 4-5 doubles =
// 8-10 words, handed in as `pw`: the WP READY words at the stream's cursor (mt_make_ready + a plain load, issued at the end of the
// previous step).  The 1 % regime switch continues draw by draw (MtStream).  finish_step() completes the step once the gaussian
// (drawn by wave B) is there.
struct SplitOut { double price, volume, u_hi, u_lo; };
__device__ __forceinline__ void market_step(Env &e, const Params &p, int64_t i, int32_t a_disc, float a_buy, float a_sell,
                                            double &reward, const uint32_t (&pw)[WP], SplitOut &so) {
    const Cfg &c = p.cfg;
    uint32_t *__restrict__ blkP = p.mtP + i * MT_STRIDE;

    // ---- _execute_action :400-447
    const double price = e.close;
    const double pv0 = e.cash + e.holdings * price;
    int kind = 0;   // 0 none, 1 buy, 2 sell
    double amount = 0.0;
    bool amount_f32 = false;
    if (c.continuous) {
        const float b = a_buy < 0.0f ? 0.0f : (a_buy > 1.0f ? 1.0f : a_buy);
        const float s = a_sell < 0.0f ? 0.0f : (a_sell > 1.0f ? 1.0f : a_sell);
        double buy;
        bool bf = true;
        if (e.cash_kind == 0) buy = (double)(b * (float)(e.cash * 0.1));
        else if (e.cash_kind == 1) buy = (double)(b * ((float)e.cash * 0.1f));
        else { buy = (double)b * (e.cash * 0.1); bf = false; }
        const double sell = (double)s * (e.holdings * 0.1);
        if (buy > sell && buy > 0.0) { kind = 1; amount = buy; amount_f32 = bf; }
        else if (sell > 0.0) { kind = 2; amount = sell; }
    } else {
        if (a_disc == 1) { kind = 1; amount = e.cash * 0.05; }
        else if (a_disc == 2) { kind = 1; amount = e.cash * 0.2; }
        else if (a_disc == 3) { kind = 2; amount = e.holdings * 0.05; }
        else if (a_disc == 4) { kind = 2; amount = e.holdings * 0.2; }
    }
    bool traded = false;
    if (kind == 1) traded = buy_guard(e, amount, amount_f32);
    else if (kind == 2) traded = !(amount <= 0.0 || e.holdings < amount);

    // ---- P draws of the common path: [slippage] volume regime-test high low
    double U[5];
#pragma unroll
    for (int q = 0; q < 5; ++q) U[q] = u53(mt_temper(pw[2 * q]), mt_temper(pw[2 * q + 1]));
    if (traded) {
        if (kind == 1) buy_apply(e, c, amount, amount_f32, price, U[0]);
        else sell_apply(e, c, amount, price, U[0]);
    }
    const double pv1 = e.cash + e.holdings * price;
    reward = pv1 - pv0;                                           // :440-441
    if (!traded) reward -= 1.0;                                   // :444-445
    const double volume = 0.5 + (2.0 - 0.5) * (traded ? U[1] : U[0]);     // :349
    const double ureg = traded ? U[2] : U[1];
    const uint32_t tq = traded ? 1u : 0u;
    double u_hi = traded ? U[3] : U[2], u_lo = traded ? U[4] : U[3];
    if (ureg < 0.01) {
        // :135 — a regime switch: _randbelow(2) (two-bit words until one is < 2), the trend draw, then high and low.  1 % of the
        // env-steps but SOME lane of a 64-env wave in every second step, so it is served from the words behind the common path's
        // (pw[2 (tq + 2) ..]: up to four _randbelow attempts fit) instead of draw by draw — seven or eight dependent memory round trips
        // that made wave A the last at bar1 whenever they ran.  A fifth attempt (6 % of the switches) continues on the serial stream.
        uint32_t rem[WP - 6];
#pragma unroll
        for (int k = 0; k < WP - 6; ++k) rem[k] = mt_temper(traded ? pw[6 + k] : pw[4 + k]);
        uint32_t att = 0, pick = rem[0] >> 30;
#pragma unroll
        for (uint32_t j = 1; j < 4u; ++j) if (pick >= 2u) { pick = rem[j] >> 30; att = j; }
        if (pick < 2u) {
            uint32_t w[6];
#pragma unroll
            for (int q = 0; q < 6; ++q) w[q] = att == 0u ? rem[1 + q] : att == 1u ? rem[2 + q] : att == 2u ? rem[3 + q] : rem[4 + q];
            update_regime_with(e, pick, u53(w[0], w[1]));
            u_hi = u53(w[2], w[3]);
            u_lo = u53(w[4], w[5]);
            mt_advance(e.ppos, e.ppretw, 2u * (tq + 2u) + att + 7u);
        } else {
            mt_advance(e.ppos, e.ppretw, 2u * (tq + 2u) + 4u);
            MtStream sp(blkP, e.ppos, e.ppretw);
            update_regime(e, sp);
            u_hi = sp.random53();
            u_lo = sp.random53();
            e.ppos = sp.pos; e.ppretw = sp.pretw;
        }
    } else {
        mt_advance(e.ppos, e.ppretw, 2u * (tq + 4u));
    }
    so.price = price; so.volume = volume; so.u_hi = u_hi; so.u_lo = u_lo;
}

// second half of a step: price, candle, termination (:350-386)
template <class H>
__device__ __forceinline__ bool finish_step(Env &e, const Params &p, H &hist, int phase, const SplitOut &so, double g) {
    const Cfg &c = p.cfg;
    const double np_ = price_update(e, c, so.price, so.volume, g);
    const double hi = np_ * (1.0 + (1.02 - 1.0) * so.u_hi);       // :353
    const double lo = np_ * (0.98 + (1.0 - 0.98) * so.u_lo);      // :354
    hist.put(phase, np_, make_float4((float)so.price, (float)hi, (float)lo, (float)so.volume));
    e.close = np_;
    const double pv = e.cash + e.holdings * np_;
    e.step += 1;
    return e.step >= (uint32_t)c.max_steps || pv <= 0.0 || pv >= c.initial_balance * 10.0;   // :382-386
}

// one legacy_gauss() value (polar method) from the NumPy stream whose state (cursor, ready mark, cached half) is in `e`.  `lw`: the
// WL ready words at the cursor = four attempts; only a fifth attempt (0.2 % of the pairs) goes draw by draw.  (Round 3 handed in two
// attempts: a third is needed by 4.6 % of the pairs, i.e. by SOME lane of a 64-env wave in 3 steps of 4, and the draw-by-draw path —
// four dependent memory round trips — was 4.7 us of wave B's 13.5-us step, profiles/r04_crypto_phase_clocks.txt.)
__device__ __forceinline__ double draw_gauss(Env &e, uint32_t *__restrict__ blkL, const uint32_t (&lw)[WL]) {
    if (e.has_gauss) {
        e.has_gauss = 0;
        const double g = e.gauss;
        e.gauss = 0.0;
        return g;
    }
    double x1 = 0.0, x2 = 0.0, r2 = 0.0;
    uint32_t used = 0;
    bool ok = false;
#pragma unroll
    for (int a = 0; a < WL / 4; ++a) {
        if (!ok) {
            x1 = 2.0 * u53(mt_temper(lw[4 * a]), mt_temper(lw[4 * a + 1])) - 1.0;
            x2 = 2.0 * u53(mt_temper(lw[4 * a + 2]), mt_temper(lw[4 * a + 3])) - 1.0;
            r2 = x1 * x1 + x2 * x2;
            used = 4u * (uint32_t)(a + 1);
            ok = !(r2 >= 1.0 || r2 == 0.0);
        }
    }
    mt_advance(e.lpos, e.lpretw, used);
    if (!ok) {
        MtStream sl(blkL, e.lpos, e.lpretw);
        do {
            x1 = 2.0 * sl.random53() - 1.0;
            x2 = 2.0 * sl.random53() - 1.0;
            r2 = x1 * x1 + x2 * x2;
        } while (r2 >= 1.0 || r2 == 0.0);
        e.lpos = sl.pos; e.lpretw = sl.pretw;
    }
    const double f = sqrt(-2.0 * log(r2) / r2);
    e.gauss = f * x1;
    e.has_gauss = 1;
    return f * x2;
}
// wave-level wrapper: makes the lanes' next WL words ready (a chunk twist every ~14 steps, the lanes of a wave consume in step),
// loads them and draws.  `want`: this lane needs a gaussian.
__device__ __forceinline__ double next_gauss(Env &e, uint32_t *__restrict__ blkL, bool want) {
    const bool fresh = want && !e.has_gauss;
    mt_make_ready(blkL, e.lpos, e.lpretw, WL, fresh);
    uint32_t lw[WL] = {};
    if (fresh) mt_load_run<WL>(blkL + e.lpos, lw);
    return want ? draw_gauss(e, blkL, lw) : 0.0;
}

// reset :301-340 for every lane of a wave that resets (`active`) AT ONCE, the draws taken as in a step: candle k's ten CPython-stream
// words (volume, regime test, high, low, open) are one plain load of READY words (mt_make_ready is wave-convergent: the lanes of a wave
// that reset together consume alike), its gaussian comes through next_gauss(); the 1 % regime switch continues draw by draw.  Same
// draws in the same order as reset_body() on the serial streams.  All lanes of the wave must call it (ballots inside).
#ifndef CGE_CRYPTO_WAVE_RESET_MIN
#define CGE_CRYPTO_WAVE_RESET_MIN 2   // resets of a workgroup in one step from which they run side by side (below: one by one through the draw-window slot)
#endif
template <class H>
__device__ __forceinline__ void reset_wave(Env &e, const Params &p, H &hist, int phase, bool active, uint32_t *__restrict__ blkP, uint32_t *__restrict__ blkL) {
    if (active) { e.cash = p.cfg.initial_balance; e.cash_kind = 0; e.holdings = 0.0; e.step = 0; e.needs_reset = 0; e.ep_return = 0.0; }
    double price = 50000.0;
    int slot = phase;
#pragma unroll 1
    for (int k = 0; k < HLEN; ++k) {
        mt_make_ready(blkP, e.ppos, e.ppretw, WP, active);
        uint32_t pw[WP] = {};
        if (active) mt_load_run<WP>(blkP + e.ppos, pw);
        const double g = next_gauss(e, blkL, active);
        if (active) {
            double U[5];
#pragma unroll
            for (int q = 0; q < 5; ++q) U[q] = u53(mt_temper(pw[2 * q]), mt_temper(pw[2 * q + 1]));
            const double volume = 0.5 + (2.0 - 0.5) * U[0];
            double u_hi = U[2], u_lo = U[3], u_op = U[4];
            if (U[1] < 0.01) {                                    // :135 — rare: consume up to the regime test, continue draw by draw
                mt_advance(e.ppos, e.ppretw, 4u);
                MtStream sp(blkP, e.ppos, e.ppretw);
                update_regime(e, sp);
                u_hi = sp.random53(); u_lo = sp.random53(); u_op = sp.random53();
                e.ppos = sp.pos; e.ppretw = sp.pretw;
            } else {
                mt_advance(e.ppos, e.ppretw, (uint32_t)WPC);
            }
            price = price_update(e, p.cfg, price, volume, g);
            const double hi = price * (1.0 + (1.02 - 1.0) * u_hi);
            const double lo = price * (0.98 + (1.0 - 0.98) * u_lo);
            const double op = price * (0.99 + (1.01 - 0.99) * u_op);
            hist.put(slot, price, make_float4((float)op, (float)hi, (float)lo, (float)volume));
        }
        slot = slot + 1 == HLEN ? 0 : slot + 1;
    }
    if (active) e.close = price;
}

// NumPy pairwise sum of 14 / 20 float64 values (loops_utils.h.src): 8 running partials, a balanced
// tree, then the tail added sequentially.
struct Pairwise {
    double r[8], res;
    template <int P, int N>
    __device__ __forceinline__ void add(double v) {
        constexpr int BODY = N - (N % 8);
        if constexpr (P < 8) r[P] = v;
        else if constexpr (P < BODY) r[P % 8] += v;
        if constexpr (P == BODY - 1) res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        if constexpr (P >= BODY) res += v;
    }
};

// indicator state of one env while its 50 closes stream by (K = logical candle index, 0 oldest)
struct Indicators {
    double ef, es, sig, macd, mx, mn, prev;
    double bb[20];
    Pairwise pg, pl, pm;

    template <int K>
    __device__ __forceinline__ void close(double x) {
        constexpr double mf = 2.0 / (12 + 1), ms = 2.0 / (26 + 1), mg = 2.0 / (9 + 1);
        if constexpr (K == 0) { ef = es = x; mx = mn = x; }
        else {
            ef = (x * mf) + (ef * (1.0 - mf));                     // _ema :113-117
            es = (x * ms) + (es * (1.0 - ms));
            mx = x > mx ? x : mx;
            mn = x < mn ? x : mn;
        }
        if constexpr (K >= 25) {                                    // macd_values for prefixes 26..50, :94-100
            macd = ef - es;
            sig = K == 25 ? macd : (macd * mg) + (sig * (1.0 - mg));
        }
        if constexpr (K >= 36) {                                    // last 14 deltas, :50-55
            const double d = x - prev;
            pg.add<K - 36, 14>(d > 0.0 ? d : 0.0);
            pl.add<K - 36, 14>(d < 0.0 ? -d : 0.0);
        }
        if constexpr (K >= 30) {                                    // Bollinger window, :64-76
            bb[K - 30] = x;
            pm.add<K - 30, 20>(x);
        }
        prev = x;
    }
};

constexpr int NFULL = HLEN / CH;            // 4 chunks of CH candles ...
constexpr int CT = HLEN - NFULL * CH;       // ... and 2 candles that travel with the 11 scalar features
static_assert(CH % 4 == 0 && CT * 5 + 11 == 21 && NFULL * CH * 5 + CT * 5 + 11 == OBS, "column bookkeeping of the observation row");

template <int C, int NC, int CHN, class H>
__device__ __forceinline__ void load_candles(const H &hist, int oldest, double (&cl)[CHN], float4 (&oh)[CHN]) {
#pragma unroll
    for (int j = 0; j < NC; ++j) {
        int slot = oldest + C * CHN + j;
        slot -= slot >= HLEN ? HLEN : 0;
        cl[j] = hist.close(slot);
        oh[j] = hist.rest(slot);
    }
}

// candles K0+J.. of the chunk -> observation columns 5*(K0+J).. : the five ratios of a candle (:513-515, [O,H,L,C,V] / current close)
template <int K0, int J, int NC, int NF, int CHN>
__device__ __forceinline__ void ratios(Indicators &ind, double inv, const double (&cl)[CHN], const float4 (&oh)[CHN], float (&out)[NF]) {
    if constexpr (J < NC) {
        const double x = cl[J];
        out[5 * J + 0] = (float)((double)oh[J].x * inv);
        out[5 * J + 1] = (float)((double)oh[J].y * inv);
        out[5 * J + 2] = (float)((double)oh[J].z * inv);
        out[5 * J + 3] = (float)(x * inv);
        out[5 * J + 4] = (float)((double)oh[J].w * inv);
        ind.template close<K0 + J>(x);
        ratios<K0, J + 1, NC, NF, CHN>(ind, inv, cl, oh, out);
    }
}

// _get_observation :505-561 for the wave's 64 envs.  `oldest` = slot of logical candle 0.  Rows of
// lanes whose bit is set in `rowmask` are written to dst (+ row*261 floats).
// the 11 scalar features, columns 250..260 (:517-561), from the indicator state after all 50 closes
__device__ __forceinline__ void scalar_features(const Env &e, const Params &p, const Indicators &ind, double cur, float *ft) {
    const double pv = e.cash + e.holdings * cur;
    ft[0] = e.cash_kind == 1 ? (float)e.cash / (float)p.cfg.initial_balance : (float)(e.cash / p.cfg.initial_balance);   // :524
    ft[1] = (float)(e.holdings * cur / p.cfg.initial_balance);
    ft[2] = (float)(pv / p.cfg.initial_balance);
    {
        const double ag = ind.pg.res / 14, al = ind.pl.res / 14;
        double rsi;
        if (al == 0.0) rsi = 100.0;
        else { const double rs = ag / al; rsi = 100.0 - (100.0 / (1.0 + rs)); }
        ft[3] = (float)(rsi / 100.0);
    }
    {
        const double hist = ind.macd - ind.sig, range = ind.mx - ind.mn;
        const bool ok = range > 0.0;
        ft[4] = ok ? (float)(ind.macd / range) : 0.0f;
        ft[5] = ok ? (float)(ind.sig / range) : 0.0f;
        ft[6] = ok ? (float)(hist / range) : 0.0f;
    }
    {
        const double sma = ind.pm.res / 20;
        Pairwise pd;
        pd.res = 0.0;
#define CGE_DEV(P) { const double x = ind.bb[P] - sma; pd.add<P, 20>(x * x); }
        CGE_DEV(0) CGE_DEV(1) CGE_DEV(2) CGE_DEV(3) CGE_DEV(4) CGE_DEV(5) CGE_DEV(6) CGE_DEV(7) CGE_DEV(8) CGE_DEV(9)
        CGE_DEV(10) CGE_DEV(11) CGE_DEV(12) CGE_DEV(13) CGE_DEV(14) CGE_DEV(15) CGE_DEV(16) CGE_DEV(17) CGE_DEV(18) CGE_DEV(19)
#undef CGE_DEV
        const double sd = sqrt(pd.res / 20);
        const double upper = sma + (2.0 * sd), lower = sma - (2.0 * sd);
        ft[7] = (float)(upper > lower ? (cur - lower) / (upper - lower) : 0.5);
        ft[8] = (float)(sma > 0.0 ? (upper - lower) / sma : 0.0);
        ft[9] = (float)(sma > 0.0 ? (cur - sma) / sma : 0.0);
    }
    ft[10] = (float)e.psych;
}

__device__ __forceinline__ void observe(const Env &e, const Params &p, int64_t i0, int64_t i, bool live, int oldest,
                                        float *__restrict__ dst, unsigned long long rowmask, uint32_t *__restrict__ tile) {
    (void)tile;
    const uint32_t lane = threadIdx.x & 63u;
    const int64_t li = live ? i : i0;            // dead lanes of a partial last wave read a valid column, write nothing
    const bool mine = live && ((rowmask >> lane) & 1ull);
    float *row = dst + (int64_t)lane * OBS;
    const double cur = e.close;
    const double inv = 1.0 / cur;
    const HistGlobal hist{p.closes, p.ohlv, p.n, li};
    Indicators ind;
    ind.ef = ind.es = ind.sig = ind.macd = ind.mx = ind.mn = ind.prev = 0.0;
    ind.pg.res = ind.pl.res = ind.pm.res = 0.0;
    double cl[CH];
    float4 oh[CH];
    load_candles<0, CH, CH>(hist, oldest, cl, oh);
    static_assert(NFULL == 4, "the chunk sequence below is written out for 4 full chunks + tail");
#define CGE_CHUNK(C, NEXT_NC)                                                                           \
    {                                                                                                   \
        float out[CH * 5];                                                                              \
        ratios<C * CH, 0, CH, CH * 5, CH>(ind, inv, cl, oh, out);                                       \
        load_candles<C + 1, NEXT_NC, CH>(hist, oldest, cl, oh);       /* next chunk's loads, then */    \
        store_own_row<CH * 5>(row, C * CH * 5, out, mine);            /* this chunk's 15 stores   */    \
    }
    CGE_CHUNK(0, CH) CGE_CHUNK(1, CH) CGE_CHUNK(2, CH) CGE_CHUNK(3, CT)
#undef CGE_CHUNK
    // ---- the last CT candles and the 11 scalar features: columns 240..260
    float tail[CT * 5 + 11];
    ratios<NFULL * CH, 0, CT, CT * 5 + 11, CH>(ind, inv, cl, oh, tail);
    scalar_features(e, p, ind, cur, tail + CT * 5);
    store_own_row<CT * 5 + 11>(row, NFULL * CH * 5, tail, mine);
}

// mailbox word of an env in the resident rollout: cash_kind (2 bits) | dest << 2 | this lane resets | it steps next (draw its gaussian)
enum : uint32_t { DEST_NONE = 0u, DEST_OBS = 1u, DEST_FINAL = 2u };
enum : uint32_t { F_RESET = 1u << 4, F_DRAW_NEXT = 1u << 5 };
// Rows in the resident rollout.  Wave B (lane = env) streams the 50 closes through the indicators and writes the 11 scalar
// features, columns 250..260.  Wave C writes the 250 ratio columns ROW BY ROW: lane c < 50 takes candle c of one env, so the row's
// 1,000 bytes leave as one contiguous run (a 16-byte and a 4-byte store per lane) instead of every lane scattering 16-byte
// pieces of its own row 1,044 bytes apart — with all arithmetic left in place and only the scattered stores removed the whole
// kernel took 229 us per 1M-env step instead of 593 (tools/probes/crypto_noobs.py): the store pattern was the step.
template <int K, int KEND>
__device__ __forceinline__ void feed_closes(Indicators &ind, const HistLds &hist, int oldest) {
    if constexpr (K < KEND) {
        int slot = oldest + K;
        slot -= slot >= HLEN ? HLEN : 0;
        ind.template close<K>(hist.close(slot));
        feed_closes<K + 1, KEND>(ind, hist, oldest);
    }
}
// the 11 scalar features of the lane's env from its 50 closes in the LDS window (computed; stored by the caller)
__device__ __forceinline__ void features_compute(const Env &e, const Params &p, const HistLds &hist, int oldest, float (&ft)[11]) {
    Indicators ind;
    ind.ef = ind.es = ind.sig = ind.macd = ind.mx = ind.mn = ind.prev = 0.0;
    ind.pg.res = ind.pl.res = ind.pm.res = 0.0;
    feed_closes<0, HLEN>(ind, hist, oldest);
    scalar_features(e, p, ind, e.close, ft);
}
__device__ __forceinline__ void features_resident(const Env &e, const Params &p, const HistLds &hist, int oldest, float *row, bool mine) {
    float ft[11];
    features_compute(e, p, hist, oldest, ft);
    store_own_row<11>(row, HLEN * 5, ft, mine);
}
// The (env, candle) pairs of envs [env_lo, env_hi) are dealt to the 64 lanes as one stream (pair q -> env q / 50, candle q % 50), so
// all lanes work and consecutive lanes write consecutive 20-byte pieces (a row's 1,000 bytes, then the next row's).  Only rows
// whose mailbox word has all bits of `need` and a destination are written (to obs only if to_obs_only).  flagsv: the workgroup's
// mailbox words; closes: its current closes; inv: 64 doubles of scratch for 1 / close; base_*: row 0 of the workgroup (or null).
__device__ __forceinline__ void ratio_rows_resident(const HistLds &hist, int oldest, const uint32_t *flagsv, const double *closes, double *inv_lds,
                                                    uint32_t need, float *base_obs, float *base_final, bool to_obs_only, int env_lo, int env_hi,
                                                    bool final_compact = false) {
    const uint32_t lane = hist.lane;
    inv_lds[lane] = 1.0 / closes[lane];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const uint32_t q_end = (uint32_t)env_hi * (uint32_t)HLEN;
    // two pairs per iteration (their LDS reads and float64 multiplies interleave: this wave is alone on its SIMD), (env, candle)
    // carried incrementally: 64 pairs further on is one env and 14 candles further on
    uint32_t q = (uint32_t)env_lo * (uint32_t)HLEN + lane;
    uint32_t env = q / (uint32_t)HLEN, cand = q - env * (uint32_t)HLEN;
    auto one = [&](uint32_t qq, uint32_t ev, uint32_t cd) {
        const bool in = qq < q_end;
        const uint32_t e2 = in ? ev : (uint32_t)env_hi - 1u, c2 = in ? cd : 0u;
        const uint32_t f = flagsv[e2];
        const uint32_t dest = to_obs_only ? DEST_OBS : (f >> 2) & 3u;
        float *base = dest == DEST_FINAL ? base_final : base_obs;
        const bool want = in && (f & need) == need && dest != DEST_NONE && base != nullptr;
        // a terminal row: row e2 of the workgroup's block of final_obs_out (step()), or — fused rollouts — the slot wave A assigned it in
        // the workgroup's segment of the compacted side output (bits 8.. of its mailbox word)
        const uint32_t drow = (dest == DEST_FINAL && final_compact) ? f >> 8 : e2;
        int slot = oldest + (int)c2;
        slot -= slot >= HLEN ? HLEN : 0;
        const double x = hist.lcb[HistLds::at(slot, e2)];
        const float4 o = hist.lob[HistLds::at(slot, e2)];
        const double inv = inv_lds[e2];
        const float v[5] = {(float)((double)o.x * inv), (float)((double)o.y * inv), (float)((double)o.z * inv), (float)(x * inv), (float)((double)o.w * inv)};
        if (want) {
            float *dstp = base + (int64_t)drow * OBS + 5u * c2;
            *reinterpret_cast<Piece16 *>(dstp) = Piece16{__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])};
            dstp[4] = v[4];
        }
    };
    auto advance = [&](uint32_t &ev, uint32_t &cd) { cd += 64u - (uint32_t)HLEN; ev += 1u; if (cd >= (uint32_t)HLEN) { cd -= (uint32_t)HLEN; ev += 1u; } };
#pragma unroll 1
    for (; q - lane < q_end; q += 128u) {
        uint32_t env1 = env, cand1 = cand;
        advance(env1, cand1);
        one(q, env, cand);
        one(q + 64u, env1, cand1);
        env = env1; cand = cand1;
        advance(env, cand);
    }
}

__device__ __forceinline__ void hash_cont(uint64_t key, uint64_t t, float &b, float &s) {
    const uint64_t u0 = mix64(key + t * 0xD1342543DE82EF95ull + 0) >> 40, u1 = mix64(key + t * 0xD1342543DE82EF95ull + 1) >> 40;
    b = (float)((double)u0 / 8388608.0 - 1.0);
    s = (float)((double)u1 / 8388608.0 - 1.0);
}

// ------------------------------------------------------------------ resident rollout
// k fused steps with the 64 envs' 50-candle window resident in LDS (76.8 KB: closes [50][64] f64 + ohlv [50][64] float4): the
// window is read from HBM once per launch instead of once per step (1,200 of a step's ~2,900 bytes), and every new candle is
// also written through to the [50][N] arrays, so nothing has to be copied back.  LDS allows two such windows per CU; with one
// wave per window (measured: 580 us per 1M-env step against 667 streaming) half of the CU's SIMDs idle while each wave issues
// ~5,500 mostly float64 instructions per step.  So a workgroup is FOUR waves over the same 64 envs and the same window:
//   A steps the market: trade, volume, regime, P-stream draws (market_step), then — after bar1 — price, candle, termination
//     (finish_step), rewards, resets;
//   B (lane = env) streams the 50 closes through the indicators and writes the 11 scalar features (features_resident); it owns
//     the NumPy stream and draws the next step's gaussian (next_gauss: polar method, software float64 log);
//   C (two waves, 32 envs each) writes the 250 ratio columns row by row (ratio_rows_resident: lane = candle, 1,000 contiguous
//     bytes per env) and never loads from memory: nothing ever waits for its stores.
// B's observation of step t and C's gaussian for step t+1 overlap A's first half of step t+1.  Hand-over per step, two barriers:
//   bar1: B has finished reading window(t-1), C's gaussian for step t is in LDS, A has done the first half of step t;
//         A then finishes the step and puts the candle and the scalars B needs (cash, holdings, psychology, close, flags) into LDS;
//   bar2: window(t) is complete.
// A step in which any env of the workgroup resets takes two more barriers (B: terminal rows -> final_obs; C hands the NumPy
// stream's state to A; A: the 50-candle resets into the LDS window, state back to C; B: the reset rows), so it is not pipelined;
// it is 6 % of the workgroup-steps in steady state.  An episode reset (~700 draws) uses the workgroup's one 96-word draw-window
// slot (simultaneous resets of a workgroup take turns).
constexpr int RES_SLOTS = 1, RES_SLOT_WORDS = 96;
struct HistDefer {                                             // wave A's view during a step: the candle waits in registers for the publish
    double c;
    float4 o;
    int slot;
    HistGlobal g;
    __device__ __forceinline__ void put(int s, double cc, float4 oo) { slot = s; c = cc; o = oo; g.put(s, cc, oo); }
};
constexpr size_t RES_HIST = (size_t)HLEN * 64 * (sizeof(double) + sizeof(float4));
constexpr size_t RES_MAIL = 8 * 64 * sizeof(double) + 2 * 64 * sizeof(uint32_t) + 16;   // cash, holdings, psych, close, g, L cache, 1/close x 2 writers | flags, L cursor | wave flag
constexpr size_t RES_LDS = RES_HIST + RES_MAIL + (size_t)RES_SLOTS * RES_SLOT_WORDS * 4;
static_assert(2 * RES_LDS <= 160 * 1024, "two workgroups per CU");
// The waves talk through LDS only, so their barrier orders LDS traffic (lgkmcnt) and nothing else: __syncthreads() would also
// drain vmcnt, i.e. make the row-writing wave wait for its stores every step.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// The reset hand-over is the exception: waves A and B take turns on the env's NumPy generator block in memory, so bar3 / bar4 also
// wait for the waves' own stores and loads.
__device__ __forceinline__ void full_barrier() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
#ifdef CGE_CRYPTO_TIMING
__device__ unsigned long long g_timing[16384 * 16];     // slots 0-6: the phases between the barriers; 8-10: inside wave A's first half; 15: workgroup-steps
#define TICK(k) do { const unsigned long long now_ = wall_clock64(); if (lane == 0) { g_timing[blockIdx.x * 16 + k] += now_ - t_last; } t_last = now_; } while (0)
#else
#define TICK(k)
#endif
constexpr int RES_WAVES = 4;
// the NumPy stream's state as one mailbox word: lpos | ready-mark code << 10 | has_gauss << 15 (waves A and B take turns on the stream)
__device__ __forceinline__ uint32_t lcur_pack(const Env &e) { return e.lpos | ((e.lpretw > e.lpos ? mt_ready_encode(e.lpretw) : 0u) << 10) | (e.has_gauss << 15); }
__device__ __forceinline__ void lcur_unpack(Env &e, uint32_t u) { e.lpos = u & 1023u; e.lpretw = mt_ready_decode((u >> 10) & 31u); e.has_gauss = (u >> 15) & 1u; }
// ONE_STEP: the step() entry (k = 1); a separate instantiation mainly so that profiles tell the two paths apart
template <bool ONE_STEP>
__global__ __launch_bounds__(RES_WAVES * BLOCK) void resident_kernel(Params p) {
    if (ONE_STEP) p.k_steps = 1;
    extern __shared__ __align__(16) unsigned char res_lds[];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t role = threadIdx.x >> 6;                    // wave-uniform: 0 A (market), 1 B (features, gaussians), 2 and 3 C (rows)
    double *lcb = reinterpret_cast<double *>(res_lds);
    float4 *lob = reinterpret_cast<float4 *>(res_lds + (size_t)HLEN * 64 * sizeof(double));
    double *mail = reinterpret_cast<double *>(res_lds + RES_HIST) + lane;                        // [6][64]: cash, holdings, psych, close, gaussian, L cache
    uint32_t *mailu = reinterpret_cast<uint32_t *>(res_lds + RES_HIST + 8 * 64 * sizeof(double)) + lane;   // flags; [64 +] L cursor
    uint32_t *waveflag = reinterpret_cast<uint32_t *>(res_lds + RES_HIST + 8 * 64 * sizeof(double) + 2 * 64 * sizeof(uint32_t));
    uint32_t *slots = reinterpret_cast<uint32_t *>(res_lds + RES_HIST + RES_MAIL);
    double &m_gauss = mail[4 * 64], &m_lcache = mail[5 * 64];
    uint32_t &m_lcur = mailu[64];                              // lcur_pack()
    const int64_t i0 = (int64_t)blockIdx.x * BLOCK;
    const int64_t i = i0 + lane;
    const bool live = i < p.n;
    const int64_t li = live ? i : i0;
    HistLds hist{lcb, lob, lane, HistGlobal{p.closes, p.ohlv, p.n, li}};
    if (role >= 1u) {   // the window: HBM -> LDS by waves B and C (17 + 17 + 16 slots) while wave A already steps the market
        const int s_lo = role == 1u ? 0 : role == 2u ? 17 : 34, s_hi = role == 1u ? 17 : role == 2u ? 34 : HLEN;
        // all of a wave's slots in ONE round trip (34 loads in flight, ~100 VGPRs that nothing else needs yet): in batches of six the
        // window took three dependent trips, a quarter of a step() workgroup's life
        constexpr int NB = 17;
        double c[NB];
        float4 o[NB];
#pragma unroll
        for (int j = 0; j < NB; ++j) { const int sl = s_lo + j < s_hi ? s_lo + j : s_hi - 1; c[j] = hist.g.close(sl); o[j] = hist.g.rest(sl); }
#pragma unroll
        for (int j = 0; j < NB; ++j) { const int sl = s_lo + j < s_hi ? s_lo + j : s_hi - 1; hist.put_lds(sl, c[j], o[j]); }
    }
    uint32_t *__restrict__ blkP = p.mtP + li * MT_STRIDE;
    uint32_t *__restrict__ blkL = p.mtL + li * MT_STRIDE;
    int phase = p.phase;
#ifdef CGE_CRYPTO_TIMING
    unsigned long long t_last = wall_clock64();
#endif
    if (role >= 2u) {
        const int env_lo = role == 2u ? 0 : 32, env_hi = env_lo + 32;      // two row writers, 32 envs each
        double *inv_scratch = mail - lane + (role == 2u ? 6 : 7) * 64;
        // ---------------- waves C: the ratio columns, row by row.  It never loads from memory, so its ~128 row stores per step are
        // never waited for: they drain while the next steps run (the barriers order LDS traffic only).
#pragma unroll 1
        for (int t = 0; t < p.k_steps; ++t) {
            const int next_phase = phase + 1 == HLEN ? 0 : phase + 1;
            lds_barrier();                                     // bar1
            lds_barrier();                                     // bar2
#ifdef CGE_CRYPTO_TIMING
            if (role == 2u) { t_last = wall_clock64(); }
#endif
            float *base_obs = p.obs ? p.obs + (int64_t)t * p.obs_step_stride + i0 * OBS : nullptr;
            float *base_final = p.fin.rows ? static_cast<float *>(p.fin.rows) + (int64_t)blockIdx.x * p.fin.cap * OBS : (p.final_obs ? p.final_obs + i0 * OBS : nullptr);
            ratio_rows_resident(hist, next_phase, mailu - lane, mail - lane + 192, inv_scratch, 0u, base_obs, base_final, false, env_lo, env_hi, p.fin.rows != nullptr);
#ifdef CGE_CRYPTO_TIMING
            if (role == 2u) { TICK(6); }
#endif
            if (*waveflag) {
                full_barrier();                                 // bar3: the pre-reset rows have been read out of the window
                full_barrier();                                 // bar4: the fresh windows are in LDS
                ratio_rows_resident(hist, next_phase, mailu - lane, mail - lane + 192, inv_scratch, F_RESET, base_obs, nullptr, true, env_lo, env_hi);   // the reset rows
            }
            phase = next_phase;
        }
        lds_barrier();                                         // closing barrier
        return;
    }
    if (role == 1u) {
        // ---------------- wave B: indicators + scalar features of every row; owns the NumPy stream and draws the next step's gaussian
        Env v;                                                 // the record: the stream's state (lpos, lpretw, has_gauss, gauss) is this wave's from here on;
        v.load(p.scal, p.n, li);                               // cash, holdings, psych, close, cash_kind are refreshed from the mailbox every step
        // a lane that starts with a pending NEXT_STEP reset does not step first: no gaussian for it
        {
            const bool w0 = live && !(p.mode == CGE_AUTORESET_NEXT_STEP && v.needs_reset);
            const double g0 = next_gauss(v, blkL, w0);
            if (w0) m_gauss = g0;
        }
#pragma unroll 1
        for (int t = 0; t < p.k_steps; ++t) {
            const int next_phase = phase + 1 == HLEN ? 0 : phase + 1;
            lds_barrier();                                     // bar1: the gaussian of step t is in LDS
            lds_barrier();                                     // bar2
            TICK(4);
            const uint32_t u = *mailu;
            const bool slow = *waveflag != 0u;
            // step t+1's gaussian: its words are requested BEFORE the features are computed, so the load's latency hides behind them
            // (a step with resets hands the stream to wave A first: then the words are fetched afterwards)
            const bool more = t + 1 < p.k_steps;
            const bool wn_early = more && !slow && live && (u & F_DRAW_NEXT);
            uint32_t lw[WL] = {};
            if (more && !slow) {
                // (slack: the lanes' cursors drift apart — 4 to 16 words per pair — and without it SOME lane twists in nearly every step)
                mt_make_ready(blkL, v.lpos, v.lpretw, WL, wn_early && !v.has_gauss, nullptr, (uint32_t)MT_CHUNK);
                if (wn_early && !v.has_gauss) mt_load_run<WL>(blkL + v.lpos, lw);
            }
            v.cash = mail[0]; v.holdings = mail[64]; v.psych = mail[128]; v.close = mail[192]; v.cash_kind = u & 3u;
            const uint32_t dest = (u >> 2) & 3u;
            float *obs_row = p.obs ? p.obs + (int64_t)t * p.obs_step_stride + li * OBS : nullptr;
            float *row = dest != DEST_FINAL ? obs_row
                         : p.fin.rows ? static_cast<float *>(p.fin.rows) + ((int64_t)blockIdx.x * p.fin.cap + (u >> 8)) * OBS : p.final_obs + li * OBS;
            const bool want = dest != DEST_NONE && row != nullptr;
            // The features are computed, then step t+1's gaussian is drawn, THEN the features are stored: the gaussian's words were
            // requested before any store of this step, so waiting for them (one in-order counter for a wave's loads and stores) waits
            // for nothing else; drawn after the row stores, the same wait also sat out those stores' way to memory — wave B was the last
            // at bar1 by 2-3 us in every step (profiles/r04_crypto_phase_clocks.txt).
            const bool any_row = __ballot(want) != 0ull;
            float ft[11];
            if (any_row) features_compute(v, p, hist, next_phase, ft);
            double gn = 0.0;
            if (!slow && more && wn_early) gn = draw_gauss(v, blkL, lw);
            if (any_row) store_own_row<11>(want ? row : obs_row, HLEN * 5, ft, want);
            TICK(5);
            if (slow) {
                m_lcur = lcur_pack(v); m_lcache = v.gauss;
                full_barrier();                                 // bar3: the pre-reset rows are out, A may rewrite the window and take the stream
                full_barrier();                                 // bar4: the fresh windows are in LDS, the stream is back
                const uint32_t u2 = *mailu;
                if (u2 & F_RESET) { lcur_unpack(v, m_lcur); v.gauss = m_lcache; }
                const bool again = (u2 & F_RESET) != 0u && obs_row != nullptr;
                v.cash = mail[0]; v.holdings = mail[64]; v.psych = mail[128]; v.close = mail[192]; v.cash_kind = u2 & 3u;
                if (__ballot(again)) features_resident(v, p, hist, next_phase, obs_row, again);
                if (more) {
                    const bool wn = live && (*mailu & F_DRAW_NEXT);
                    const double gn = next_gauss(v, blkL, wn);
                    if (wn) m_gauss = gn;
                }
            } else if (more) {
                if (wn_early) m_gauss = gn;
            }
            phase = next_phase;
        }
        mt_make_ready(blkL, v.lpos, v.lpretw, WL, live && !v.has_gauss);   // the next launch's first gaussian starts with a plain load
        m_lcur = lcur_pack(v); m_lcache = v.gauss;
        lds_barrier();                                         // closing barrier: A stores the record
        return;
    }
    // ---------------- wave A: the market
    Env e;
    e.load(p.scal, p.n, li);
    uint32_t pw[WP];                                           // the step's CPython-stream words: ready, loaded one step ahead
    mt_make_ready(blkP, e.ppos, e.ppretw, WP, live);
    mt_load_run<WP>(blkP + e.ppos, pw);
    const uint64_t key = hash_env_key(p.a_seed, (uint64_t)(p.env0 + li));
    double rsum = 0.0;
    int32_t dcount = 0;
    uint32_t fin_used = 0;                                      // terminal rows this workgroup has delivered to its segment (fused rollouts)
#pragma unroll 1
    for (int t = 0; t < p.k_steps; ++t) {
        double reward = 0.0;
        bool term = false, reset_now = false, stepped = false;
        const int next_phase = phase + 1 == HLEN ? 0 : phase + 1;
        HistDefer pend{0.0, make_float4(0.f, 0.f, 0.f, 0.f), phase, HistGlobal{p.closes, p.ohlv, p.n, li}};
        SplitOut so{0.0, 0.0, 0.0, 0.0};
        if (live) {
            if (p.mode == CGE_AUTORESET_NEXT_STEP && e.needs_reset) {
                reset_now = true;
            } else {
                int32_t a = 0;
                float ab = 0.0f, as = 0.0f;
                if (p.cfg.continuous) {
                    if (p.actions) { const float2 vv = reinterpret_cast<const float2 *>(p.actions)[(int64_t)t * p.n + i]; ab = vv.x; as = vv.y; }
                    else hash_cont(key, (uint64_t)(p.t0 + t), ab, as);
                } else {
                    a = p.actions ? reinterpret_cast<const int32_t *>(p.actions)[(int64_t)t * p.n + i]
                                  : (int32_t)hash_action_from_key(key, (uint64_t)(p.t0 + t), 5u, 0u);
                }
#ifdef CGE_CRYPTO_TIMING
                asm volatile("" :: "v"(a), "v"(ab), "v"(as));
                TICK(8);                                         // action (hash / load)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                TICK(9);                                         // this step's generator words have arrived (loaded a step ago)
#endif
                market_step(e, p, li, a, ab, as, reward, pw, so);
#ifdef CGE_CRYPTO_TIMING
                asm volatile("" :: "v"(so.u_hi), "v"(so.u_lo), "v"(reward));
                TICK(10);                                        // the trade and the draws
#endif
                stepped = true;
            }
        }
        // the next step's words (the cursor is final): a chunk twist every ~3 steps, in wave A's slack before bar1, then a plain load.
        // After the launch's last step the twist still happens (the next launch then starts with nothing but the load).
        mt_make_ready(blkP, e.ppos, e.ppretw, WP, stepped, nullptr, (uint32_t)MT_CHUNK);
        if (t + 1 < p.k_steps && stepped) mt_load_run<WP>(blkP + e.ppos, pw);
        TICK(0);
        lds_barrier();                                          // bar1: B is done with window(t-1), C's gaussian for this step is in LDS
        TICK(1);
        if (stepped) {
            term = finish_step(e, p, pend, phase, so, m_gauss);
            e.ep_return += reward;
            if (term) {
                e.episodes += 1;
                if (p.ep_ret) p.ep_ret[i] = e.ep_return;
                if (p.ep_len) p.ep_len[i] = (int32_t)e.step;
                if (p.mode == CGE_AUTORESET_SAME_STEP) reset_now = true;
                else if (p.mode == CGE_AUTORESET_NEXT_STEP) e.needs_reset = 1;
            }
            hist.put_lds(pend.slot, pend.c, pend.o);
        }
        const unsigned long long rm = __ballot(reset_now);
        // rows of this step: a SAME_STEP terminal row goes to final_obs (and the reset row to obs afterwards), a NEXT_STEP
        // reset-only row only exists after the reset
        // (fused rollouts: the terminal rows of the workgroup's envs take the next slots of its segment of the compacted side output)
        const bool fin = live && term && reset_now;
        const unsigned long long fm = __ballot(fin);
        const uint32_t fslot = fin_used + (uint32_t)__popcll(fm & ((1ull << lane) - 1ull));
        const bool fin_ok = p.fin.rows ? (int64_t)fslot < p.fin.cap : p.final_obs != nullptr;
        if (fin && p.fin.rows && fin_ok) p.fin.index[(int64_t)blockIdx.x * p.fin.cap + fslot] = (int64_t)t * p.n + i;
        fin_used += (uint32_t)__popcll(fm);
        const uint32_t dest = !live ? DEST_NONE : fin ? (fin_ok ? DEST_FINAL : DEST_NONE) : reset_now ? DEST_NONE : DEST_OBS;
        // does this lane step at t+1 (then wave C draws its gaussian now)?  not when the step is a NEXT_STEP reset-only one
        const bool steps_next = live && !(p.mode == CGE_AUTORESET_NEXT_STEP && e.needs_reset && !reset_now);
        mail[0] = e.cash; mail[64] = e.holdings; mail[128] = e.psych; mail[192] = e.close;
        *mailu = e.cash_kind | (dest << 2) | (reset_now ? F_RESET : 0u) | (steps_next ? F_DRAW_NEXT : 0u) | ((fin && p.fin.rows ? fslot : 0u) << 8);
        if (lane == 0) *waveflag = rm ? 1u : 0u;
        lds_barrier();                                          // bar2: window(t) is complete
        TICK(2);
        if (rm) {
            full_barrier();                                      // bar3: B's pre-reset rows are out, C's stream state is in LDS
            const uint32_t rank = (uint32_t)__popcll(rm & ((1ull << lane) - 1ull));
            if ((uint32_t)__popcll(rm) >= (uint32_t)CGE_CRYPTO_WAVE_RESET_MIN) {
                // several envs of the workgroup at once (every env at a shared time limit): side by side, draws straight from the
                // streams' ready words — taking turns on the one draw-window slot cost 64 x ~80 us per workgroup, 262 ms of a
                // 1M-env episode (profiles/r04_crypto_wave_reset.txt)
                if (reset_now) { lcur_unpack(e, m_lcur); e.gauss = m_lcache; }
                reset_wave(e, p, hist, next_phase, reset_now, blkP, blkL);
                if (reset_now) {
                    m_lcur = lcur_pack(e); m_lcache = e.gauss;
                    mail[0] = e.cash; mail[64] = e.holdings; mail[128] = e.psych; mail[192] = e.close;
                    *mailu = e.cash_kind | F_RESET | F_DRAW_NEXT;
                }
            } else {
#pragma unroll 1
            for (uint32_t round = 0; round * RES_SLOTS < (uint32_t)__popcll(rm); ++round) {
                if (reset_now && rank / RES_SLOTS == round) {
                    lcur_unpack(e, m_lcur); e.gauss = m_lcache;
                    uint32_t *row = slots + (rank % RES_SLOTS) * RES_SLOT_WORDS;
                    LdsDrawsCall<64> sp(row, blkP, e.ppos, e.ppretw);
                    LdsDrawsCall<32> sl(row + 64, blkL, e.lpos, e.lpretw);
                    reset_body(e, p, hist, next_phase, sp, sl);
                    sp.flush(); sl.flush();
                    e.ppos = sp.pos; e.ppretw = sp.pretw; e.lpos = sl.pos; e.lpretw = sl.pretw;
                    m_lcur = lcur_pack(e); m_lcache = e.gauss;
                    mail[0] = e.cash; mail[64] = e.holdings; mail[128] = e.psych; mail[192] = e.close;
                    *mailu = e.cash_kind | F_RESET | F_DRAW_NEXT;
                }
            }
            }
            mt_make_ready(blkP, e.ppos, e.ppretw, WP, reset_now);   // the prefetched words belonged to the finished episode's cursor
            if (t + 1 < p.k_steps && reset_now) mt_load_run<WP>(blkP + e.ppos, pw);
            full_barrier();                                      // bar4
        }
        if (live) {
            rsum += reward;
            dcount += term ? 1 : 0;
            if (p.reward) p.reward[(int64_t)t * p.n + i] = (float)reward;
            if (p.terminated) p.terminated[(int64_t)t * p.n + i] = term ? 1 : 0;
            if (p.truncated) p.truncated[(int64_t)t * p.n + i] = 0;                    // step() through this kernel (k = 1)
        }
        TICK(3);
#ifdef CGE_CRYPTO_TIMING
        if (lane == 0) g_timing[blockIdx.x * 16 + 15] += 1;
#endif
        phase = next_phase;
    }
    lds_barrier();                                              // closing barrier: B's last rows are out, C's stream state is in LDS
    if (live) {
        lcur_unpack(e, m_lcur); e.gauss = m_lcache;
        e.store(p.scal, p.n, i);
        if (p.reward_sum) p.reward_sum[i] = rsum;
        if (p.done_count) p.done_count[i] = dcount;
        if (p.fin.count && lane == 0) p.fin.count[blockIdx.x] = (int32_t)fin_used;
    }
}

__global__ __launch_bounds__(BLOCK) void reset_kernel(Params p) {
    __shared__ uint32_t tile[64 * ROW];
    const int64_t i0 = (int64_t)blockIdx.x * BLOCK;
    const int64_t i = i0 + threadIdx.x;
    const bool live = i < p.n;
    Env e;
    e.load(p.scal, p.n, live ? i : i0);
    if (live && (!p.mask || p.mask[i])) {
        do_reset(e, p, i, p.phase, tile + (threadIdx.x & 63u) * ROW);
        e.store(p.scal, p.n, i);
    }
    if (p.obs) observe(e, p, i0, i, live, p.phase, p.obs + i0 * OBS, ~0ull, tile);
}

// initial MarketSimulator state (:125-130) and balances; also rewinds the stream cursors after (re)seeding
__global__ __launch_bounds__(256) void init_kernel(uint4 *scal, int64_t n, double initial_balance, int rewind_only) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    Env e;
    e.load(scal, n, i);
    if (!rewind_only) {
        e.cash = initial_balance; e.holdings = 0.0; e.close = 50000.0; e.psych = 0.5; e.trend = 0.0; e.gauss = 0.0;
        e.step = 0; e.regime = SIDEWAYS; e.cash_kind = 0; e.needs_reset = 0; e.episodes = 0; e.ep_return = 0.0;
    }
    e.has_gauss = 0;            // np.random.seed() drops the cached gaussian
    e.gauss = 0.0;
    e.ppos = e.lpos = 0;
    e.ppretw = e.lpretw = 0;
    e.store(scal, n, i);
}

__global__ __launch_bounds__(256) void info_kernel(const uint4 *__restrict__ scal, int64_t n, int field, double *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    Env e;
    e.load(scal, n, i);
    double v = 0.0;
    switch (field) {
        case CGE_CRYPTO_INFO_PORTFOLIO_VALUE: v = e.cash + e.holdings * e.close; break;
        case CGE_CRYPTO_INFO_CASH: v = e.cash; break;
        case CGE_CRYPTO_INFO_HOLDINGS: v = e.holdings; break;
        case CGE_CRYPTO_INFO_CURRENT_PRICE: v = e.close; break;
        case CGE_CRYPTO_INFO_MARKET_PSYCHOLOGY: v = e.psych; break;
        case CGE_CRYPTO_INFO_REGIME: v = e.regime; break;
        case CGE_CRYPTO_INFO_STEP: v = e.step; break;
        case CGE_CRYPTO_INFO_TREND_STRENGTH: v = e.trend; break;
        case CGE_CRYPTO_INFO_EPISODES: v = e.episodes; break;
        case CGE_CRYPTO_INFO_NEEDS_RESET: v = e.needs_reset; break;
        case CGE_CRYPTO_INFO_CASH_KIND: v = e.cash_kind; break;
    }
    out[i] = v;
}

}  // namespace crypto
}  // namespace cge

using namespace cge;

struct cge_crypto : HandleBase {
    cge_crypto_config cfg{};
    uint4 *scal = nullptr;
    double *closes = nullptr;
    float4 *ohlv = nullptr;
    uint32_t *mtP = nullptr, *mtL = nullptr;
    int phase = 0;
    bool resident_ready = false;   // resident_kernel's dynamic-LDS limit has been raised

    crypto::Params params() const {
        crypto::Params p{};
        p.scal = scal; p.closes = closes; p.ohlv = ohlv; p.mtP = mtP; p.mtL = mtL; p.n = n; p.env0 = env0;
        p.cfg = crypto::Cfg{cfg.initial_balance, cfg.trading_fee_rate, cfg.slippage_rate, cfg.min_price, cfg.max_price,
                            cfg.volatility_base, cfg.market_psychology_factor, cfg.max_steps, cfg.action_type};
        p.mode = cfg.autoreset_mode; p.phase = phase;
        p.ep_ret = ep_ret; p.ep_len = ep_len;
        return p;
    }
    unsigned blocks() const { return (unsigned)((n + crypto::BLOCK - 1) / crypto::BLOCK); }
    void free_all() {
        (void)hipFree(scal); (void)hipFree(closes); (void)hipFree(ohlv); (void)hipFree(mtP); (void)hipFree(mtL);
    }
};

static hipError_t launch_resident(cge_crypto *h, const crypto::Params &p, hipStream_t s, bool one_step) {
    if (!h->resident_ready) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(crypto::resident_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)crypto::RES_LDS);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void *>(crypto::resident_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)crypto::RES_LDS);
        if (e != hipSuccess) return e;
        h->resident_ready = true;
    }
    if (one_step) hipLaunchKernelGGL(crypto::resident_kernel<true>, dim3(h->blocks()), dim3(crypto::RES_WAVES * crypto::BLOCK), crypto::RES_LDS, s, p);
    else hipLaunchKernelGGL(crypto::resident_kernel<false>, dim3(h->blocks()), dim3(crypto::RES_WAVES * crypto::BLOCK), crypto::RES_LDS, s, p);
    h->last_kernel = one_step ? "cge::crypto::resident_kernel<true>" : "cge::crypto::resident_kernel<false>";
    return hipGetLastError();
}

extern "C" {

#ifdef CGE_CRYPTO_TIMING
int cge_crypto_debug_timing(unsigned long long *out, int clear) {
    static unsigned long long all[16384 * 16];
    if (hipMemcpyFromSymbol(all, HIP_SYMBOL(crypto::g_timing), sizeof all) != hipSuccess) return 1;
    for (int k = 0; k < 16; ++k) out[k] = 0;
    for (int b = 0; b < 16384; ++b)
        for (int k = 0; k < 16; ++k) out[k] += all[b * 16 + k];
    if (clear) { memset(all, 0, sizeof all); if (hipMemcpyToSymbol(HIP_SYMBOL(crypto::g_timing), all, sizeof all) != hipSuccess) return 1; }
    return 0;
}
#endif

void cge_crypto_default_config(cge_crypto_config *c) {
    if (!c) return;
    *c = cge_crypto_config{10000.0, 0.001, 0.0005, 100.0, 100000.0, 0.02, 0.1, 1000, 0, CGE_AUTORESET_NEXT_STEP, 0};
}

int cge_crypto_create(const cge_crypto_config *cfg, int64_t n_envs, int device, int64_t env_index0, cge_crypto **out) {
    if (!cfg || !out || n_envs <= 0 || env_index0 < 0) return CGE_ERR_INVALID_ARG;
    *out = nullptr;
    if (cfg->autoreset_mode < 0 || cfg->autoreset_mode > 2 || cfg->max_steps <= 0 || cfg->max_steps > crypto::MAX_STEPS_LIMIT ||
        cfg->action_type < 0 || cfg->action_type > 1 || !(cfg->min_price > 0) || !(cfg->max_price >= cfg->min_price))
        return CGE_ERR_INVALID_ARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return CGE_ERR_NO_DEVICE;
    cge_crypto *h = new cge_crypto();
    h->cfg = *cfg; h->n = n_envs; h->env0 = env_index0; h->device = device;
    DeviceGuard g(device);
    const size_t sb = (size_t)4 * n_envs * sizeof(uint4), cb = (size_t)crypto::HLEN * n_envs * sizeof(double),
                 ob = (size_t)crypto::HLEN * n_envs * sizeof(float4), mb = (size_t)n_envs * MT_STRIDE * sizeof(uint32_t);
    hipError_t e;
    if ((e = hipMalloc(&h->scal, sb)) != hipSuccess || (e = hipMalloc(&h->closes, cb)) != hipSuccess ||
        (e = hipMalloc(&h->ohlv, ob)) != hipSuccess || (e = hipMalloc(&h->mtP, mb)) != hipSuccess ||
        (e = hipMalloc(&h->mtL, mb)) != hipSuccess || (e = hipMemset(h->scal, 0, sb)) != hipSuccess ||
        (e = hipMemset(h->closes, 0, cb)) != hipSuccess || (e = hipMemset(h->ohlv, 0, ob)) != hipSuccess) {
        h->free_all();
        delete h;
        return CGE_ERR_HIP;
    }
    h->device_bytes = sb + cb + ob + 2 * mb;
    e = launch_mt_seed(h->mtP, MT_STRIDE, n_envs, nullptr, 0, env_index0, 0, nullptr);
    if (e == hipSuccess) e = launch_mt_seed(h->mtL, MT_STRIDE, n_envs, nullptr, 0, env_index0, 1, nullptr);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(crypto::init_kernel, dim3((unsigned)((n_envs + 255) / 256)), dim3(256), 0, nullptr, h->scal, n_envs,
                           cfg->initial_balance, 0);
        // no reset here: a fresh handle is a freshly constructed env (:244-278) — the MarketSimulator state a
        // reset leaves behind would otherwise leak into the first real reset(seed=...)
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
    if (e != hipSuccess) {
        h->free_all();
        delete h;
        return CGE_ERR_HIP;
    }
    *out = h;
    return CGE_OK;
}

int cge_crypto_destroy(cge_crypto *h) {
    if (!h) return CGE_ERR_INVALID_ARG;
    DeviceGuard g(h->device);
    (void)hipDeviceSynchronize();
    h->free_all();
    delete h;
    return CGE_OK;
}

int cge_crypto_seed(cge_crypto *h, const uint64_t *seeds, uint64_t base_seed, void *stream) {
    if (!h) return CGE_ERR_INVALID_ARG;
    DeviceGuard g(h->device);
    if (!seeds && base_seed + (uint64_t)(h->env0 + h->n) > 0x100000000ull)
        return h->fail(CGE_ERR_INVALID_ARG, "cge_crypto_seed: np.random.seed needs seeds < 2**32");
    CGE_TRY(h, launch_mt_seed(h->mtP, MT_STRIDE, h->n, seeds, base_seed, h->env0, 0, as_stream(stream)));
    CGE_TRY(h, launch_mt_seed(h->mtL, MT_STRIDE, h->n, seeds, base_seed, h->env0, 1, as_stream(stream)));
    hipLaunchKernelGGL(crypto::init_kernel, dim3((unsigned)((h->n + 255) / 256)), dim3(256), 0, as_stream(stream), h->scal, h->n,
                       h->cfg.initial_balance, 1);
    CGE_TRY(h, hipGetLastError());
    return CGE_OK;
}

int cge_crypto_reset(cge_crypto *h, const uint8_t *mask, float *obs_out, void *stream) {
    if (!h) return CGE_ERR_INVALID_ARG;
    DeviceGuard g(h->device);
    crypto::Params p = h->params();
    p.mask = mask; p.obs = obs_out;
    hipLaunchKernelGGL(crypto::reset_kernel, dim3(h->blocks()), dim3(crypto::BLOCK), 0, as_stream(stream), p);
    CGE_TRY(h, hipGetLastError());
    return CGE_OK;
}

int cge_crypto_step(cge_crypto *h, const void *actions, float *obs_out, float *reward_out, uint8_t *terminated_out,
                    uint8_t *truncated_out, float *final_obs_out, void *stream) {
    if (!h) return CGE_ERR_INVALID_ARG;
    if (!actions || !obs_out || !reward_out || !terminated_out)
        return h->fail(CGE_ERR_INVALID_ARG, "cge_crypto_step: null actions/obs/reward/terminated pointer");
    DeviceGuard g(h->device);
    crypto::Params p = h->params();
    p.actions = actions; p.obs = obs_out; p.reward = reward_out; p.terminated = terminated_out; p.truncated = truncated_out;
    p.final_obs = final_obs_out; p.k_steps = 1;
    // the four-wave resident kernel also serves a single step: its waves share the row, feature and market work of the 64 envs
    CGE_TRY(h, launch_resident(h, p, as_stream(stream), true));
    h->phase = (h->phase + 1) % crypto::HLEN;
    return CGE_OK;
}

int cge_crypto_rollout(cge_crypto *h, int32_t k_steps, const void *actions, uint64_t action_seed, int64_t t0, float *obs_out,
                       int64_t obs_step_stride, float *reward_traj_out, uint8_t *terminated_traj_out, double *reward_sum_out,
                       int32_t *done_count_out, void *stream) {
    if (!h) return CGE_ERR_INVALID_ARG;
    if (k_steps < 0 || obs_step_stride < 0 || (obs_step_stride != 0 && obs_step_stride < h->n * crypto::OBS))
        return h->fail(CGE_ERR_INVALID_ARG, "cge_crypto_rollout: bad k_steps / obs_step_stride");
    if (k_steps == 0) return CGE_OK;
    DeviceGuard g(h->device);
    crypto::Params p = h->params();
    p.k_steps = k_steps; p.actions = actions; p.a_seed = action_seed; p.t0 = t0; p.obs = obs_out; p.obs_step_stride = obs_step_stride;
    p.reward = reward_traj_out; p.terminated = terminated_traj_out; p.reward_sum = reward_sum_out; p.done_count = done_count_out;
    p.fin = FinalSeg{h->fin_rows, h->fin_index, h->fin_count, h->fin_cap, h->n};
    CGE_TRY(h, launch_resident(h, p, as_stream(stream), false));
    h->phase = (h->phase + k_steps) % crypto::HLEN;
    return CGE_OK;
}

CGE_DEFINE_FINAL_OBS(crypto, float, 64)

int cge_crypto_info(cge_crypto *h, int32_t field_id, double *out, void *stream) {
    if (!h) return CGE_ERR_INVALID_ARG;
    if (!out || field_id < 0 || field_id > CGE_CRYPTO_INFO_CASH_KIND) return h->fail(CGE_ERR_INVALID_ARG, "cge_crypto_info: bad field / null out");
    DeviceGuard g(h->device);
    hipLaunchKernelGGL(crypto::info_kernel, dim3((unsigned)((h->n + 255) / 256)), dim3(256), 0, as_stream(stream), h->scal, h->n, field_id, out);
    CGE_TRY(h, hipGetLastError());
    return CGE_OK;
}

size_t cge_crypto_state_bytes(const cge_crypto *h) { return h ? 12 * 4 + 6 * 8 + 2 * MT_N * 4 + crypto::HLEN * 5 * 8 : 0; }

int cge_crypto_get_state(cge_crypto *h, void *host_buf, void *stream) {
    if (!h || !host_buf) return CGE_ERR_INVALID_ARG;
    DeviceGuard g(h->device);
    const int64_t n = h->n;
    std::vector<uint4> sc((size_t)4 * n);
    std::vector<double> cl((size_t)crypto::HLEN * n);
    std::vector<float4> oh((size_t)crypto::HLEN * n);
    std::vector<uint32_t> mp((size_t)n * MT_STRIDE), ml((size_t)n * MT_STRIDE);
    CGE_TRY(h, hipStreamSynchronize(as_stream(stream)));
    CGE_TRY(h, hipMemcpy(sc.data(), h->scal, sc.size() * sizeof(uint4), hipMemcpyDeviceToHost));
    CGE_TRY(h, hipMemcpy(cl.data(), h->closes, cl.size() * sizeof(double), hipMemcpyDeviceToHost));
    CGE_TRY(h, hipMemcpy(oh.data(), h->ohlv, oh.size() * sizeof(float4), hipMemcpyDeviceToHost));
    CGE_TRY(h, hipMemcpy(mp.data(), h->mtP, mp.size() * 4, hipMemcpyDeviceToHost));
    CGE_TRY(h, hipMemcpy(ml.data(), h->mtL, ml.size() * 4, hipMemcpyDeviceToHost));
    const size_t rec = cge_crypto_state_bytes(h);
    for (int64_t i = 0; i < n; ++i) {
        uint8_t *p = (uint8_t *)host_buf + (size_t)i * rec;
        crypto::Env e;
        e.unpack(sc[i], sc[n + i], sc[2 * n + i], sc[3 * n + i]);
        int32_t hd[12] = {(int32_t)e.regime, (int32_t)e.step, (int32_t)e.needs_reset, (int32_t)e.cash_kind,
                          0, 0, (int32_t)e.has_gauss, (int32_t)e.episodes, 0, 0, 0, 0};
        double scv[6] = {e.cash, e.holdings, e.psych, e.trend, e.gauss, e.ep_return};   // [5]: episode return so far
        mt_export_cpython(&mp[(size_t)i * MT_STRIDE], e.ppos, e.ppretw, (uint32_t *)(p + 96), &hd[4]);
        mt_export_cpython(&ml[(size_t)i * MT_STRIDE], e.lpos, e.lpretw, (uint32_t *)(p + 96 + MT_N * 4), &hd[5]);
        memcpy(p, hd, 48);
        memcpy(p + 48, scv, 48);
        double *hh = (double *)(p + 96 + 2 * MT_N * 4);
        for (int k = 0; k < crypto::HLEN; ++k) {
            const int slot = (h->phase + k) % crypto::HLEN;
            const float4 v = oh[(size_t)slot * n + i];
            hh[5 * k] = v.x; hh[5 * k + 1] = v.y; hh[5 * k + 2] = v.z; hh[5 * k + 3] = cl[(size_t)slot * n + i]; hh[5 * k + 4] = v.w;
        }
    }
    return CGE_OK;
}

int cge_crypto_set_state(cge_crypto *h, const void *host_buf, void *stream) {
    if (!h || !host_buf) return CGE_ERR_INVALID_ARG;
    DeviceGuard g(h->device);
    const int64_t n = h->n;
    std::vector<uint4> sc((size_t)4 * n);
    std::vector<double> cl((size_t)crypto::HLEN * n);
    std::vector<float4> oh((size_t)crypto::HLEN * n);
    std::vector<uint32_t> mp((size_t)n * MT_STRIDE, 0u), ml((size_t)n * MT_STRIDE, 0u);
    const size_t rec = cge_crypto_state_bytes(h);
    for (int64_t i = 0; i < n; ++i) {
        const uint8_t *p = (const uint8_t *)host_buf + (size_t)i * rec;
        int32_t hd[12];
        double scv[6];
        memcpy(hd, p, 48);
        memcpy(scv, p + 48, 48);
        if (hd[0] < 0 || hd[0] > 4 || hd[1] < 0 || hd[1] > crypto::MAX_STEPS_LIMIT || hd[4] < 0 || hd[4] > MT_N || hd[5] < 0 || hd[5] > MT_N)
            return h->fail(CGE_ERR_INVALID_ARG, "cge_crypto_set_state: malformed record");
        const double *hh = (const double *)(p + 96 + 2 * MT_N * 4);
        crypto::Env e;
        e.cash = scv[0]; e.holdings = scv[1]; e.psych = scv[2]; e.trend = scv[3]; e.gauss = scv[4]; e.ep_return = scv[5];
        e.close = hh[5 * (crypto::HLEN - 1) + 3];
        e.regime = (uint32_t)hd[0]; e.step = (uint32_t)hd[1]; e.needs_reset = (uint32_t)(hd[2] & 1); e.cash_kind = (uint32_t)(hd[3] & 3);
        e.has_gauss = (uint32_t)(hd[6] & 1); e.episodes = hd[7] < 0 ? 0u : (uint32_t)hd[7];
        // a CPython state: every word from the index on is generated-but-unconsumed (ready); index 624 = regenerate first
        e.ppos = hd[4] >= MT_N ? 0u : (uint32_t)hd[4]; e.ppretw = hd[4] >= MT_N ? 0u : (uint32_t)MT_N;
        e.lpos = hd[5] >= MT_N ? 0u : (uint32_t)hd[5]; e.lpretw = hd[5] >= MT_N ? 0u : (uint32_t)MT_N;
        e.pack(sc[i], sc[n + i], sc[2 * n + i], sc[3 * n + i]);
        memcpy(&mp[(size_t)i * MT_STRIDE], p + 96, MT_N * 4);
        memcpy(&ml[(size_t)i * MT_STRIDE], p + 96 + MT_N * 4, MT_N * 4);
        memcpy(&mp[(size_t)i * MT_STRIDE + MT_N], &mp[(size_t)i * MT_STRIDE], MT_PAD * 4);       // mirror words (cge_device.hpp)
        memcpy(&ml[(size_t)i * MT_STRIDE + MT_N], &ml[(size_t)i * MT_STRIDE], MT_PAD * 4);
        for (int k = 0; k < crypto::HLEN; ++k) {
            const int slot = (h->phase + k) % crypto::HLEN;
            cl[(size_t)slot * n + i] = hh[5 * k + 3];
            oh[(size_t)slot * n + i] = make_float4((float)hh[5 * k], (float)hh[5 * k + 1], (float)hh[5 * k + 2], (float)hh[5 * k + 4]);
        }
    }
    CGE_TRY(h, hipStreamSynchronize(as_stream(stream)));
    CGE_TRY(h, hipMemcpy(h->scal, sc.data(), sc.size() * sizeof(uint4), hipMemcpyHostToDevice));
    CGE_TRY(h, hipMemcpy(h->closes, cl.data(), cl.size() * sizeof(double), hipMemcpyHostToDevice));
    CGE_TRY(h, hipMemcpy(h->ohlv, oh.data(), oh.size() * sizeof(float4), hipMemcpyHostToDevice));
    CGE_TRY(h, hipMemcpy(h->mtP, mp.data(), mp.size() * 4, hipMemcpyHostToDevice));
    CGE_TRY(h, hipMemcpy(h->mtL, ml.data(), ml.size() * 4, hipMemcpyHostToDevice));
    return CGE_OK;
}

size_t cge_crypto_device_bytes(const cge_crypto *h) { return h ? h->device_bytes : 0; }
int cge_crypto_episode_stats(cge_crypto *h, double *return_out, int32_t *length_out) {
    if (!h) return CGE_ERR_INVALID_ARG;
    h->ep_ret = return_out; h->ep_len = length_out;
    return CGE_OK;
}

const char *cge_crypto_last_error(const cge_crypto *h) { return h ? h->last_error.c_str() : "null handle"; }

const char *cge_crypto_last_kernel(const cge_crypto *h) { return h ? h->last_kernel.c_str() : ""; }

}  // extern "C"
