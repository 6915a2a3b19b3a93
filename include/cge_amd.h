/* cge_amd.h — C ABI of the MI355X-native batched env stepper (libcge_amd.so).
 *
 * The reference (hasnainfarid/Custom_Gymnasium_Environments) has no FFI: its boundary is the
 * Gymnasium Python API of each env class.  Every entry point below therefore cites the reference
 * method it replaces for a whole batch of N independent instances:
 *
 *   <env>_create   <->  Env.__init__            e.g. snake_env_classic/snake_env.py:19-47
 *   <env>_seed     <->  random.seed()/np.random.seed()/reset(seed=) as each env uses them
 *                       (crypto_trading_env.py:305-307, traffic environment.py:145-147; snake never
 *                       seeds `random` itself, snake_env.py:50 — the per-env stream protocol is
 *                       `random.seed(seed_i)` with env i run alone)
 *   <env>_reset    <->  Env.reset()             snake_env.py:49-65, crypto:301-340, traffic:141-166
 *   <env>_step     <->  Env.step(action)        snake_env.py:67-119, crypto:342-398, traffic:168-203
 *   <env>_rollout  <->  the `while not done: env.step(a)` loops of the reference's scripts
 *                       (snake_env_classic/example.py:21-32) fused into one launch
 *   <env>_rollout_final_obs <-> the terminal observation every reference step() RETURNS (snake_env.py:88-94,113-119 and each env's
 *                       step()): step() hands it to final_obs_out in SAME_STEP mode while obs_out gets the reset observation; a fused
 *                       SAME_STEP rollout writes the reset observation to slot t of its trajectory, and the terminal rows go to a side
 *                       output registered here, so that rollout(trajectory) + this output is exactly what k step() calls return.
 *                       The rows are compacted per SEGMENT: <env>_final_obs_segment(h) consecutive envs (the envs one wavefront or
 *                       workgroup steps: 64, traffic 16) own seg_capacity consecutive rows; segment g = envs [g*S, (g+1)*S).
 *                       Caller-owned DEVICE buffers: rows_out [n_segments * seg_capacity, *obs_shape], index_out [n_segments *
 *                       seg_capacity] (int64: step-in-call * n_envs + env) and count_out [n_segments] (int32).  Every later rollout
 *                       call writes count_out[g] = terminal rows segment g produced in THAT call (no zeroing by the caller) and the
 *                       first min(count, seg_capacity) of them, in step order, to the segment's rows; the surplus is dropped (the count
 *                       still says so).  All NULL unregisters.  NEXT_STEP / DISABLED rollouts deliver nothing: slot t of their
 *                       trajectory IS the terminal observation.
 *   <env>_info     <->  the `info` dicts        snake_env.py:63,117
 *   <env>_episode_stats <-> the per-episode return / length that the reference's training scripts read back from their
 *                       vector-env wrapper as episode_return_mean / episode_len_mean
 *                       (smart_parking_env/examples/training.py:55, smartclimate_rl-main/training/train.py): what
 *                       gymnasium's RecordEpisodeStatistics publishes as infos["episode"] = {"r", "l"}.  Registers two
 *                       caller-owned DEVICE buffers of n_envs elements (either may be NULL; both NULL unregisters): every later
 *                       step() / rollout() writes, for each env that finishes an episode in that call and from inside the
 *                       step kernel itself, the episode's return (float64 sum of its rewards in step order; the accumulator
 *                       lives in the env's state record) and its length in env steps (a NEXT_STEP reset step is not a step
 *                       of any episode).  Entries of envs that did not finish are left untouched; the "_episode" mask of a
 *                       step is terminated | truncated.  A rollout leaves the LAST episode an env finished in it.
 *
 * Conventions
 *   - every function returns CGE_OK (0) or a negative cge_status; nothing throws across the ABI;
 *     <env>_last_error(h) gives a human-readable message for the last failure on that handle.
 *   - all *_out / actions / mask / seeds pointers are DEVICE pointers owned by the caller (e.g. a
 *     torch tensor's data_ptr()) and must stay alive until `stream` reaches the call; host_buf
 *     pointers are HOST pointers.  `stream` is a hipStream_t passed as void* (NULL = default stream).
 *   - all work is enqueued on `stream`; no call synchronises except get_state/set_state/
 *     error_count/destroy.
 *   - the library owns the struct-of-arrays env state inside the handle.  A handle is bound to one
 *     device and is not thread-safe; different handles are independent.
 *   - observations are written row-major as (n_envs, *single_obs_shape), exactly the layout
 *     gymnasium.vector.VectorEnv returns.
 *   - `env_index0` is the global index of the handle's first env: per-env seeds and the synthetic
 *     action hash are functions of the GLOBAL index so results do not depend on how a batch is
 *     sharded over GPUs.
 */
#ifndef CGE_AMD_H
#define CGE_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    CGE_OK = 0,
    CGE_ERR_INVALID_ARG = -1,
    CGE_ERR_HIP = -2,          /* a HIP runtime call failed; see *_last_error */
    CGE_ERR_UNSUPPORTED = -3,  /* configuration not compiled into this build */
    CGE_ERR_NO_DEVICE = -4
} cge_status;

/* gymnasium.vector.AutoresetMode equivalents */
enum {
    CGE_AUTORESET_NEXT_STEP = 0, /* done env returns terminal obs; next step() resets it (action ignored) */
    CGE_AUTORESET_SAME_STEP = 1, /* done env is reset inside step(); terminal obs -> final_obs_out */
    CGE_AUTORESET_DISABLED = 2   /* never reset; stepping a finished env does what the reference does */
};

const char *cge_version(void);
/* Synthetic action source shared by device rollouts, the oracle and the tests:
 * u = mix64(mix64(a_seed + env*0x9E3779B97F4A7C15) + t*0xD1342543DE82EF95 + j); ((u>>32)*n)>>32 */
uint32_t cge_hash_action(uint64_t a_seed, uint64_t env, uint64_t t, uint32_t n, uint32_t j);

/* ------------------------------------------------------------------------------------------ */
/* Snake  (snake_env_classic/snake_env.py: SnakeEnvClassic)                                    */
/*   obs int8 (G,G): 0 empty, 1 snake, 2 food   action int32 in {0 up,1 right,2 down,3 left}    */
/* ------------------------------------------------------------------------------------------ */
typedef struct cge_snake cge_snake;

typedef struct {
    int32_t grid_size;      /* reference default 20 (snake_env.py:19); BASELINE configs use 10; any size from 4 to 30 is compiled in
                             * (other sizes: CGE_ERR_UNSUPPORTED) */
    int32_t max_steps;      /* reference: 1000 (snake_env.py:47); 0 -> 1000; <= 4095 for grid 10, <= 65535 otherwise */
    int32_t autoreset_mode; /* CGE_AUTORESET_* */
    int32_t reserved;
} cge_snake_config;

enum { /* cge_snake_info field ids (int32 per env) */
    CGE_SNAKE_INFO_SCORE = 0,
    CGE_SNAKE_INFO_LENGTH = 1,
    CGE_SNAKE_INFO_STEPS = 2,
    CGE_SNAKE_INFO_DIRECTION = 3,
    CGE_SNAKE_INFO_FOOD_R = 4,
    CGE_SNAKE_INFO_FOOD_C = 5,
    CGE_SNAKE_INFO_BOARD_FULL = 6, /* sticky: reference's _place_food would spin forever (snake_env.py:123) */
    CGE_SNAKE_INFO_EPISODES = 7,
    CGE_SNAKE_INFO_HEAD_R = 8,
    CGE_SNAKE_INFO_HEAD_C = 9,
    CGE_SNAKE_INFO_NEEDS_RESET = 10
};

int cge_snake_create(const cge_snake_config *cfg, int64_t n_envs, int device, int64_t env_index0,
                     cge_snake **out);
int cge_snake_destroy(cge_snake *h);
/* env i's private MT19937 stream := CPython random.seed(s_i); s_i = seeds[i] if seeds != NULL
 * (device pointer, n_envs uint64) else base_seed + env_index0 + i.  Does not reset the envs. */
int cge_snake_seed(cge_snake *h, const uint64_t *seeds, uint64_t base_seed, void *stream);
/* reset envs with mask[i] != 0 (all if mask == NULL); writes ALL n_envs obs rows if obs_out != NULL */
int cge_snake_reset(cge_snake *h, const uint8_t *mask, int8_t *obs_out, void *stream);
/* one step() for every env.  An action outside {0..3} raises ValueError in the reference
 * (snake_env.py:69-70); here the env is left untouched, its row reports (current obs, 0, 0, 0) and a
 * device-side counter is bumped: read it with cge_snake_error_count (which synchronises). */
int cge_snake_step(cge_snake *h, const int32_t *actions, int8_t *obs_out, float *reward_out,
                   uint8_t *terminated_out, uint8_t *truncated_out /*nullable: always 0, snake_env.py:119*/,
                   int8_t *final_obs_out /*nullable*/, void *stream);
/* k_steps fused step()s in ONE launch (env state stays in registers between steps).  actions:
 * [k_steps, n_envs] int32 or NULL -> cge_hash_action(action_seed, env_index0+i, t0+t, 4, 0).  obs_out
 * (nullable): one [n_envs,G,G] buffer rewritten every step (obs_step_stride = 0) or a trajectory buffer
 * [k_steps, n_envs, G, G] (obs_step_stride = n_envs*G*G).  reward_traj_out / terminated_traj_out
 * (nullable): per-step [k_steps, n_envs] outputs, i.e. exactly what k step() calls would return (SAME_STEP: together with
 * the terminal rows of cge_snake_rollout_final_obs);
 * reward_sum_out / done_count_out (nullable) accumulate per env over the k steps. */
int cge_snake_rollout(cge_snake *h, int32_t k_steps, const int32_t *actions, uint64_t action_seed, int64_t t0,
                      int8_t *obs_out, int64_t obs_step_stride, float *reward_traj_out,
                      uint8_t *terminated_traj_out, float *reward_sum_out, int32_t *done_count_out,
                      void *stream);
/* terminal observations of SAME_STEP rollouts, compacted per segment of 64 envs: see <env>_rollout_final_obs at the top of this file */
int cge_snake_rollout_final_obs(cge_snake *h, int8_t *rows_out, int64_t *index_out, int64_t seg_capacity, int32_t *count_out);
int64_t cge_snake_final_obs_segment(const cge_snake *h);
int cge_snake_info(cge_snake *h, int32_t field_id, int32_t *out, void *stream);
/* render_mode="rgb_array" (snake_env.py:175-188): uint8 [n_envs, G, G, 3] (4-byte aligned) — empty (0,0,0), snake (0,255,0),
 * food (255,0,0) — of the CURRENT state of every env.  The pygame window of render_mode="human" (:153-173) is out of scope. */
int cge_snake_render_rgb(cge_snake *h, uint8_t *rgb_out, void *stream);
/* canonical per-env state record (host memory), identical to the oracle's: 8 int32 {len, dir, food_r,
 * food_c, score, steps, needs_reset, mt_idx}, uint32 mt[624] (CPython layout: words >= mt_idx are
 * generated-but-unconsumed), uint16 body[G*G] head first (0xFFFF unused), padded to 4 bytes. */
size_t cge_snake_state_bytes(const cge_snake *h);
int cge_snake_get_state(cge_snake *h, void *host_buf, void *stream);
int cge_snake_set_state(cge_snake *h, const void *host_buf, void *stream);
/* synchronises `stream`, returns and clears the number of invalid actions seen since the last call */
int64_t cge_snake_error_count(cge_snake *h, void *stream);
/* bytes of device memory held by the handle (SoA state + RNG streams) */
size_t cge_snake_device_bytes(const cge_snake *h);
int cge_snake_episode_stats(cge_snake *h, double *return_out, int32_t *length_out);
const char *cge_snake_last_error(const cge_snake *h);
/* the kernel(s) the last step() / rollout() call on this handle launched, by the name rocprofv3 --kernel-trace prints
 * ("" before the first call): lets a measurement attribute its time and PMC bytes to what actually ran. */
const char *cge_snake_last_kernel(const cge_snake *h);

/* ------------------------------------------------------------------------------------------ */
/* Crypto  (crypto_trading_env/crypto_trading_env.py: CryptoTradingEnv)                        */
/*   obs float32 (261,) (:505-561; the declared space says 260, :286)                          */
/*   action: int32 in {0 hold,1 buy 5%,2 buy 20%,3 sell 5%,4 sell 20%} (discrete) or float32[2] */
/*   (continuous: [buy, sell], :407-422).  All state arithmetic is float64 as in the reference; */
/*   O/H/L/V history is stored as float32 (it only feeds the float32 obs), closes as float64.   */
/* ------------------------------------------------------------------------------------------ */
typedef struct cge_crypto cge_crypto;

typedef struct {                 /* TradingConfig, crypto_trading_env.py:28-38 (history_length is fixed at 50) */
    double initial_balance;          /* 10000.0 */
    double trading_fee_rate;         /* 0.001 */
    double slippage_rate;            /* 0.0005 */
    double min_price;                /* 100.0 */
    double max_price;                /* 100000.0 */
    double volatility_base;          /* 0.02 */
    double market_psychology_factor; /* 0.1 */
    int32_t max_steps;               /* 1000 (:278); <= 65535 */
    int32_t action_type;             /* 0 discrete, 1 continuous (:247) */
    int32_t autoreset_mode;          /* CGE_AUTORESET_* */
    int32_t reserved;
} cge_crypto_config;

enum { /* cge_crypto_info field ids (float64 per env), the keys of step()'s info dict (:390-398) */
    CGE_CRYPTO_INFO_PORTFOLIO_VALUE = 0,
    CGE_CRYPTO_INFO_CASH = 1,
    CGE_CRYPTO_INFO_HOLDINGS = 2,
    CGE_CRYPTO_INFO_CURRENT_PRICE = 3,
    CGE_CRYPTO_INFO_MARKET_PSYCHOLOGY = 4,
    CGE_CRYPTO_INFO_REGIME = 5,       /* 0 bull_run 1 bear_market 2 sideways 3 crash 4 recovery (:20-25) */
    CGE_CRYPTO_INFO_STEP = 6,
    CGE_CRYPTO_INFO_TREND_STRENGTH = 7,
    CGE_CRYPTO_INFO_EPISODES = 8,
    CGE_CRYPTO_INFO_NEEDS_RESET = 9,
    CGE_CRYPTO_INFO_CASH_KIND = 10    /* dtype of self.cash under NumPy>=2: 0 Python float, 1 float32, 2 float64 */
};

void cge_crypto_default_config(cge_crypto_config *cfg);
int cge_crypto_create(const cge_crypto_config *cfg, int64_t n_envs, int device, int64_t env_index0,
                      cge_crypto **out);
int cge_crypto_destroy(cge_crypto *h);
/* reset(seed=s) seeding (:305-307): random.seed(s_i) AND np.random.seed(s_i); s_i as for snake; s_i must
 * be < 2**32 (np.random.seed raises otherwise).  Does not reset the envs; the MarketSimulator state
 * (regime, trend, psychology) is never reset, as in the reference (:257). */
int cge_crypto_seed(cge_crypto *h, const uint64_t *seeds, uint64_t base_seed, void *stream);
int cge_crypto_reset(cge_crypto *h, const uint8_t *mask, float *obs_out, void *stream);
/* actions: int32[n_envs] (discrete; any value outside 1..4 is a hold, as in the reference) or
 * float32[n_envs,2] (continuous).  reward_out is float32(reward); terminated = step>=max_steps or
 * portfolio<=0 or portfolio>=10*initial (:382-386); truncated always 0 (nullable). */
int cge_crypto_step(cge_crypto *h, const void *actions, float *obs_out, float *reward_out,
                    uint8_t *terminated_out, uint8_t *truncated_out, float *final_obs_out, void *stream);
/* k fused step()s.  actions NULL -> hash actions (discrete: cge_hash_action(seed,env,t,5,0); continuous:
 * (cge_hash(..,j) >> 40) / 2^23 - 1 for j = 0,1); else [k, n_envs(,2)].  obs_out / obs_step_stride
 * (in floats) as for snake; reward_sum_out is float64. */
int cge_crypto_rollout(cge_crypto *h, int32_t k_steps, const void *actions, uint64_t action_seed, int64_t t0,
                       float *obs_out, int64_t obs_step_stride, float *reward_traj_out,
                       uint8_t *terminated_traj_out, double *reward_sum_out, int32_t *done_count_out,
                       void *stream);
/* terminal observations of SAME_STEP rollouts, compacted per segment of 64 envs: see <env>_rollout_final_obs at the top of this file */
int cge_crypto_rollout_final_obs(cge_crypto *h, float *rows_out, int64_t *index_out, int64_t seg_capacity, int32_t *count_out);
int64_t cge_crypto_final_obs_segment(const cge_crypto *h);
int cge_crypto_info(cge_crypto *h, int32_t field_id, double *out, void *stream);
/* canonical record (host), identical to the oracle's: int32[12] {regime, step, needs_reset, cash_kind,
 * P_idx, L_idx, has_gauss, episodes, 0,0,0,0}; double[6] {cash, holdings, psych, trend_strength, gauss, episode_return_so_far (the oracle writes 0)};
 * uint32 P[624]; uint32 L[624]; double hist[50][5] (O,H,L,C,V, oldest first; O/H/L/V round-trip
 * through float32 on the device). */
size_t cge_crypto_state_bytes(const cge_crypto *h);
int cge_crypto_get_state(cge_crypto *h, void *host_buf, void *stream);
int cge_crypto_set_state(cge_crypto *h, const void *host_buf, void *stream);
size_t cge_crypto_device_bytes(const cge_crypto *h);
int cge_crypto_episode_stats(cge_crypto *h, double *return_out, int32_t *length_out);
const char *cge_crypto_last_error(const cge_crypto *h);
const char *cge_crypto_last_kernel(const cge_crypto *h);

/* ------------------------------------------------------------------------------------------ */
/* Traffic  (traffic_management_env/environment.py: TrafficManagementEnv, utils.py, config.py)  */
/*   obs float32 (14 NI + 4,) (:313-363; 130 for the default NI = 9 intersections)               */
/*   action int32[NI] in {0 maintain,1 NS_GREEN,2 EW_GREEN}                                      */
/*   State is the collapsed form of SURVEY.md 8a (queue length / waiting sum / arrivals-at-      */
/*   destination per queue, counters per intersection, vehicle count): _update_vehicles          */
/*   (:251-269) never moves a vehicle, so nothing else is observable.  Bit-exact: integer state,  */
/*   float64 reward arithmetic in the reference's order, float32 obs from float64 quotients.     */
/* ------------------------------------------------------------------------------------------ */
typedef struct cge_traffic cge_traffic;

typedef struct {                 /* TrafficManagementEnv.__init__ kwargs (:61-66) + config.py */
    int32_t grid_rows, grid_cols;    /* (5, 5); any grid of 1..64 rows / columns: the routes walk the whole grid (utils.py:196-214) */
    int32_t num_intersections;       /* 9; the env uses NI = min(num_intersections, rows*cols) (:79).  Any NI from 2 to 16: the layouts the
                                      * reference's scripts build (4, 9, 16: simple_test.py:71-76, config.py:6-7, USAGE_EXAMPLES.md:32-38)
                                      * have kernels of their own, the others share run-time-NI instances.  NI = 1 raises in the reference
                                      * at the first spawn (randint(2, 1), utils.py:181); NI = 1 or > 16: CGE_ERR_UNSUPPORTED */
    int32_t max_vehicles;            /* 50; <= 127 and max_vehicles*max_steps <= 262143 (queue word: len:7 dest:7 wait:18) */
    double spawn_rate;               /* 0.3 */
    int32_t max_steps;               /* MAX_TIMESTEPS = 1000 */
    int32_t autoreset_mode;          /* CGE_AUTORESET_* */
} cge_traffic_config;

enum { /* cge_traffic_info field ids (int32 per env; `index` selects the intersection 0..NI-1 or queue 0..4NI-1 = 4*i+dir) */
    CGE_TRAFFIC_INFO_TIMESTEP = 0,
    CGE_TRAFFIC_INFO_NUM_VEHICLES = 1,
    CGE_TRAFFIC_INFO_LIGHT_PHASE = 2,       /* 0 NS_GREEN 1 NS_YELLOW 2 EW_GREEN 3 EW_YELLOW */
    CGE_TRAFFIC_INFO_LIGHT_TIMER = 3,
    CGE_TRAFFIC_INFO_VEHICLES_PASSED = 4,
    CGE_TRAFFIC_INFO_TOTAL_WAITING_TIME = 5,
    CGE_TRAFFIC_INFO_QUEUE_LEN = 6,         /* dir order NORTH, EAST, SOUTH, WEST (utils.py:17-22) */
    CGE_TRAFFIC_INFO_QUEUE_DEST = 7,
    CGE_TRAFFIC_INFO_QUEUE_WAIT = 8,
    CGE_TRAFFIC_INFO_EPISODES = 9,
    CGE_TRAFFIC_INFO_NEEDS_RESET = 10
};

void cge_traffic_default_config(cge_traffic_config *cfg);
int cge_traffic_create(const cge_traffic_config *cfg, int64_t n_envs, int device, int64_t env_index0,
                       cge_traffic **out);
int cge_traffic_destroy(cge_traffic *h);
/* reset(seed=s) seeding (:145-147): random.seed(s_i); the NumPy generator it also seeds is never drawn from */
int cge_traffic_seed(cge_traffic *h, const uint64_t *seeds, uint64_t base_seed, void *stream);
int cge_traffic_reset(cge_traffic *h, const uint8_t *mask, float *obs_out, void *stream);
/* actions: int32 [n_envs, NI]; values other than 1/2 maintain the phase, as in the reference (:214-220) */
int cge_traffic_step(cge_traffic *h, const int32_t *actions, float *obs_out, float *reward_out,
                     uint8_t *terminated_out, uint8_t *truncated_out /*nullable*/, float *final_obs_out,
                     void *stream);
/* k fused steps; actions [k, n_envs, NI] or NULL -> cge_hash_action(seed, env, t, 3, j) for intersection j */
int cge_traffic_rollout(cge_traffic *h, int32_t k_steps, const int32_t *actions, uint64_t action_seed, int64_t t0,
                        float *obs_out, int64_t obs_step_stride, float *reward_traj_out,
                        uint8_t *terminated_traj_out, double *reward_sum_out, int32_t *done_count_out,
                        void *stream);
/* terminal observations of SAME_STEP rollouts, compacted per segment of 16 envs: see <env>_rollout_final_obs at the top of this file */
int cge_traffic_rollout_final_obs(cge_traffic *h, float *rows_out, int64_t *index_out, int64_t seg_capacity, int32_t *count_out);
int64_t cge_traffic_final_obs_segment(const cge_traffic *h);
int cge_traffic_info(cge_traffic *h, int32_t field_id, int32_t index, int32_t *out, void *stream);
/* info['total_reward'] (:372): float64 running sum of the episode's rewards */
int cge_traffic_total_reward(cge_traffic *h, double *out, void *stream);
/* canonical record (host), identical to the oracle's: int32[6] {timestep, n_vehicles, needs_reset, mt_idx,
 * episodes, 0}; double total_reward; int32 phase[NI], timer[NI], passed[NI], total_wait[NI], qlen[4 NI], qdest[4 NI],
 * qwait[4 NI]; uint32 mt[624]. */
size_t cge_traffic_state_bytes(const cge_traffic *h);
int cge_traffic_get_state(cge_traffic *h, void *host_buf, void *stream);
int cge_traffic_set_state(cge_traffic *h, const void *host_buf, void *stream);
size_t cge_traffic_device_bytes(const cge_traffic *h);
int cge_traffic_episode_stats(cge_traffic *h, double *return_out, int32_t *length_out);
const char *cge_traffic_last_error(const cge_traffic *h);
const char *cge_traffic_last_kernel(const cge_traffic *h);

/* ------------------------------------------------------------------------------------------ */
/* Smart parking  (smart_parking_env/core/parking_env.py: SmartParkingEnv + customer/parking_lot/pricing) */
/*   obs float32 (13,) in [0,1] (:306-369)   action int32 in 0..7 (:161-195)   1440 one-minute steps     */
/*   Bit-exact: integer state, float64 price/satisfaction/reward arithmetic in the reference's order.   */
/* ------------------------------------------------------------------------------------------ */
typedef struct cge_parking cge_parking;

typedef struct {
    int32_t max_steps;        /* TIMESTEPS_PER_EPISODE = 1440 (config.py:15); <= 60000 */
    int32_t autoreset_mode;   /* CGE_AUTORESET_* */
} cge_parking_config;

enum { /* cge_parking_info int32 fields (`index` = zone 0..2 where noted) — the counters behind _get_info (:371-399) */
    CGE_PARKING_INFO_TIMESTEP = 0,
    CGE_PARKING_INFO_TOTAL_CUSTOMERS = 1,
    CGE_PARKING_INFO_REJECTED = 2,
    CGE_PARKING_INFO_SATISFIED = 3,
    CGE_PARKING_INFO_TOTAL_WAIT_TIME = 4,
    CGE_PARKING_INFO_QUEUE_LENGTH = 5,
    CGE_PARKING_INFO_PRICE_CHANGES_THIS_HOUR = 6,
    CGE_PARKING_INFO_ZONE_OCCUPIED = 7,   /* index = zone */
    CGE_PARKING_INFO_PRICE_LEVEL = 8,     /* index = zone */
    CGE_PARKING_INFO_EPISODES = 9,
    CGE_PARKING_INFO_NEEDS_RESET = 10
};
enum { CGE_PARKING_INFO64_EPISODE_REVENUE = 0, CGE_PARKING_INFO64_EPISODE_SATISFACTION = 1 };

int cge_parking_create(const cge_parking_config *cfg, int64_t n_envs, int device, int64_t env_index0,
                       cge_parking **out);
int cge_parking_destroy(cge_parking *h);
/* the env never seeds `random` (parking_env.py:81): env i's private stream := random.seed(s_i), as for snake */
int cge_parking_seed(cge_parking *h, const uint64_t *seeds, uint64_t base_seed, void *stream);
int cge_parking_reset(cge_parking *h, const uint8_t *mask, float *obs_out, void *stream);
/* actions int32[n_envs]; values outside 0..7 are an idle step, as in the reference (no branch matches) */
int cge_parking_step(cge_parking *h, const int32_t *actions, float *obs_out, float *reward_out,
                     uint8_t *terminated_out, uint8_t *truncated_out /*nullable*/, float *final_obs_out,
                     void *stream);
int cge_parking_rollout(cge_parking *h, int32_t k_steps, const int32_t *actions, uint64_t action_seed, int64_t t0,
                        float *obs_out, int64_t obs_step_stride, float *reward_traj_out,
                        uint8_t *terminated_traj_out, double *reward_sum_out, int32_t *done_count_out,
                        void *stream);
/* terminal observations of SAME_STEP rollouts, compacted per segment of 64 envs: see <env>_rollout_final_obs at the top of this file */
int cge_parking_rollout_final_obs(cge_parking *h, float *rows_out, int64_t *index_out, int64_t seg_capacity, int32_t *count_out);
int64_t cge_parking_final_obs_segment(const cge_parking *h);
int cge_parking_info(cge_parking *h, int32_t field_id, int32_t index, int32_t *out, void *stream);
int cge_parking_info64(cge_parking *h, int32_t field_id, double *out, void *stream);
/* whole-handle checkpoint (host memory, opaque: 32-byte header + the device arrays in device layout); restores only
 * into a handle created with the same n_envs and config; both calls synchronise `stream` */
size_t cge_parking_snapshot_bytes(const cge_parking *h);
int cge_parking_snapshot_get(cge_parking *h, void *host_buf, void *stream);
int cge_parking_snapshot_set(cge_parking *h, const void *host_buf, void *stream);
size_t cge_parking_device_bytes(const cge_parking *h);
int cge_parking_episode_stats(cge_parking *h, double *return_out, int32_t *length_out);
const char *cge_parking_last_error(const cge_parking *h);
const char *cge_parking_last_kernel(const cge_parking *h);

/* ------------------------------------------------------------------------------------------ */
/* SmartClimate  (smartclimate_rl-main/smartclimate/env.py: SmartClimateEnv, utils.py)          */
/*   obs float32 (9,): room_temp, num_people, time_of_day, outside_temp, ac_setting, lights[4]  */
/*   action Dict{ac_temp float32[1] (clipped to 16..32), lights MultiBinary(4)} (:39-47)        */
/*   float64 dynamics; generator family D: a private default_rng(seed) (PCG64) per env —        */
/*   uniform, Lemire integers on buffered 32-bit draws, ziggurat normal, choice(p).             */
/* ------------------------------------------------------------------------------------------ */
typedef struct cge_climate cge_climate;

typedef struct {
    int32_t max_occupancy;    /* 8 (:19); <= 15 */
    int32_t episode_minutes;  /* 1440 (:21); <= 65535 */
    int32_t autoreset_mode;   /* CGE_AUTORESET_* */
    int32_t reserved;
} cge_climate_config;

enum { /* cge_climate_info float64 fields */
    CGE_CLIMATE_INFO_ROOM_TEMP = 0, CGE_CLIMATE_INFO_OUTSIDE_TEMP = 1, CGE_CLIMATE_INFO_AC_SETTING = 2,
    CGE_CLIMATE_INFO_ENERGY_USAGE = 3, CGE_CLIMATE_INFO_TOTAL_REWARD = 4, CGE_CLIMATE_INFO_NUM_PEOPLE = 5,
    CGE_CLIMATE_INFO_STEP = 6, CGE_CLIMATE_INFO_COMFORT_TIME = 7, CGE_CLIMATE_INFO_EPISODES = 8,
    CGE_CLIMATE_INFO_NEEDS_RESET = 9
};

int cge_climate_create(const cge_climate_config *cfg, int64_t n_envs, int device, int64_t env_index0,
                       cge_climate **out);
int cge_climate_destroy(cge_climate *h);
/* reset(seed=s) (:63-65): env i's generator := np.random.default_rng(s_i) (SeedSequence -> PCG64) */
int cge_climate_seed(cge_climate *h, const uint64_t *seeds, uint64_t base_seed, void *stream);
int cge_climate_reset(cge_climate *h, const uint8_t *mask, float *obs_out, void *stream);
/* ac_temp float32[n_envs] (action['ac_temp'][0]), lights int8[n_envs,4] */
int cge_climate_step(cge_climate *h, const float *ac_temp, const int8_t *lights, float *obs_out, float *reward_out,
                     uint8_t *terminated_out, uint8_t *truncated_out /*nullable*/, float *final_obs_out,
                     void *stream);
/* k fused steps; ac_temp [k,n] / lights [k,n,4] or both NULL -> hash actions:
 * ac = float32(16 + 16*(hash(seed,env,t,j=0) >> 40)/2^24), lights[j] = cge_hash_action(seed,env,t,2,1+j) */
int cge_climate_rollout(cge_climate *h, int32_t k_steps, const float *ac_temp, const int8_t *lights,
                        uint64_t action_seed, int64_t t0, float *obs_out, int64_t obs_step_stride,
                        float *reward_traj_out, uint8_t *terminated_traj_out, double *reward_sum_out,
                        int32_t *done_count_out, void *stream);
/* terminal observations of SAME_STEP rollouts, compacted per segment of 64 envs: see <env>_rollout_final_obs at the top of this file */
int cge_climate_rollout_final_obs(cge_climate *h, float *rows_out, int64_t *index_out, int64_t seg_capacity, int32_t *count_out);
int64_t cge_climate_final_obs_segment(const cge_climate *h);
int cge_climate_info(cge_climate *h, int32_t field_id, double *out, void *stream);
/* whole-handle checkpoint (host memory, opaque: 32-byte header + the device arrays in device layout); restores only
 * into a handle created with the same n_envs and config; both calls synchronise `stream` */
size_t cge_climate_snapshot_bytes(const cge_climate *h);
int cge_climate_snapshot_get(cge_climate *h, void *host_buf, void *stream);
int cge_climate_snapshot_set(cge_climate *h, const void *host_buf, void *stream);
size_t cge_climate_device_bytes(const cge_climate *h);
int cge_climate_episode_stats(cge_climate *h, double *return_out, int32_t *length_out);
const char *cge_climate_last_error(const cge_climate *h);
const char *cge_climate_last_kernel(const cge_climate *h);

/* ------------------------------------------------------------------------------------------ */
/* Fleet  (fleet_management_env/fleet_env.py: FleetManagementEnv)                               */
/*   obs float32 (76,) (:555-593; the declared space says 87)   action int32[3] in 0..7 (:157)   */
/*   terminated (:537-553) AND truncated (timestep >= 800, :265) are both reported.              */
/*   Generators: NumPy legacy np.random + CPython random, both seeded by reset(seed=) (:187-189). */
/* ------------------------------------------------------------------------------------------ */
typedef struct cge_fleet cge_fleet;

typedef struct {
    int32_t max_timesteps;    /* 800 (:123); <= 1023 */
    int32_t autoreset_mode;   /* CGE_AUTORESET_* */
} cge_fleet_config;

enum { /* cge_fleet_info float64 fields (_get_info :595-608 and per-vehicle state) */
    CGE_FLEET_INFO_TIMESTEP = 0, CGE_FLEET_INFO_MISSED_DEADLINES = 1, CGE_FLEET_INFO_COMPLETED_DELIVERIES = 2,
    CGE_FLEET_INFO_NUM_REQUESTS = 3, CGE_FLEET_INFO_WEATHER_EFFECT = 4, CGE_FLEET_INFO_TOTAL_REWARD = 5,
    CGE_FLEET_INFO_EPISODES = 6, CGE_FLEET_INFO_NEEDS_RESET = 7,
    CGE_FLEET_INFO_FUEL0 = 8, CGE_FLEET_INFO_FUEL1 = 9, CGE_FLEET_INFO_FUEL2 = 10
};

int cge_fleet_create(const cge_fleet_config *cfg, int64_t n_envs, int device, int64_t env_index0, cge_fleet **out);
int cge_fleet_destroy(cge_fleet *h);
/* reset(seed=s): np.random.seed(s_i) and random.seed(s_i); s_i < 2**32 */
int cge_fleet_seed(cge_fleet *h, const uint64_t *seeds, uint64_t base_seed, void *stream);
int cge_fleet_reset(cge_fleet *h, const uint8_t *mask, float *obs_out, void *stream);
/* actions int32 [n_envs, 3]; a value outside 0..7 costs -10 (:326-327).  truncated_out is REQUIRED here. */
int cge_fleet_step(cge_fleet *h, const int32_t *actions, float *obs_out, float *reward_out, uint8_t *terminated_out,
                   uint8_t *truncated_out, float *final_obs_out, void *stream);
/* done_count counts terminated-or-truncated steps; terminated_traj_out gets terminated | truncated << 1.
 * k_steps >= 2: the launches that finish step t (redraws, resets, their rows) run on a stream the handle owns, beside step t + 1's
 * step launch on `stream`; the call orders them with events and joins the side stream into `stream` before it returns, so to the
 * caller everything is ordered on `stream` as usual (buffers must stay valid until `stream` reaches that point). */
int cge_fleet_rollout(cge_fleet *h, int32_t k_steps, const int32_t *actions, uint64_t action_seed, int64_t t0,
                      float *obs_out, int64_t obs_step_stride, float *reward_traj_out, uint8_t *terminated_traj_out,
                      double *reward_sum_out, int32_t *done_count_out, void *stream);
/* terminal observations of SAME_STEP rollouts, compacted per segment of 64 envs: see <env>_rollout_final_obs at the top of this file */
int cge_fleet_rollout_final_obs(cge_fleet *h, float *rows_out, int64_t *index_out, int64_t seg_capacity, int32_t *count_out);
int64_t cge_fleet_final_obs_segment(const cge_fleet *h);
int cge_fleet_info(cge_fleet *h, int32_t field_id, double *out, void *stream);
/* whole-handle checkpoint (host memory, opaque: 32-byte header + the device arrays in device layout); restores only
 * into a handle created with the same n_envs and config; both calls synchronise `stream` */
size_t cge_fleet_snapshot_bytes(const cge_fleet *h);
int cge_fleet_snapshot_get(cge_fleet *h, void *host_buf, void *stream);
int cge_fleet_snapshot_set(cge_fleet *h, const void *host_buf, void *stream);
size_t cge_fleet_device_bytes(const cge_fleet *h);
int cge_fleet_episode_stats(cge_fleet *h, double *return_out, int32_t *length_out);
/* optional: cge_fleet_step also writes terminated | truncated per env into done_out (device, n_envs bytes) until it is set to NULL —
 * the mask gymnasium's SAME_STEP infos["_final_obs"] and the episode statistics want, without a second launch */
int cge_fleet_done_mask(cge_fleet *h, uint8_t *done_out);
const char *cge_fleet_last_error(const cge_fleet *h);
const char *cge_fleet_last_kernel(const cge_fleet *h);

/* ------------------------------------------------------------------------------------------ */
/* Manufacturing  (smart_manufacturing_env/manufacturing_env.py: SmartManufacturingEnv)         */
/*   obs float32 (73,) (:194-250)   action int32 in 0..24 (:85, :303-359)                        */
/*   terminated (:555-578) AND truncated (timestep >= 1500, :282) are both reported.             */
/*   Generator: gymnasium's self.np_random = Generator(PCG64(SeedSequence(seed))) (:115).        */
/* ------------------------------------------------------------------------------------------ */
typedef struct cge_manufacturing cge_manufacturing;

typedef struct {
    int32_t max_steps;        /* 1500 (:282, :569); <= 1500 (bounds the per-episode product table) */
    int32_t autoreset_mode;   /* CGE_AUTORESET_* */
} cge_manufacturing_config;

enum { /* cge_manufacturing_info float64 fields (info dict :293-299, oee_metrics :533-548, list lengths) */
    CGE_MANUFACTURING_INFO_RAW_MATERIAL = 0, CGE_MANUFACTURING_INFO_ENERGY_CONSUMPTION = 1, CGE_MANUFACTURING_INFO_TOTAL_REWARD = 2,
    CGE_MANUFACTURING_INFO_IN_SYSTEM = 3, CGE_MANUFACTURING_INFO_COMPLETED = 4, CGE_MANUFACTURING_INFO_SCRAPPED = 5,
    CGE_MANUFACTURING_INFO_PRODUCT_IDS = 6, CGE_MANUFACTURING_INFO_HISTORY_LEN = 7, CGE_MANUFACTURING_INFO_OEE_AVAILABILITY = 8,
    CGE_MANUFACTURING_INFO_OEE_PERFORMANCE = 9, CGE_MANUFACTURING_INFO_OEE_QUALITY = 10, CGE_MANUFACTURING_INFO_TIMESTEP = 11,
    CGE_MANUFACTURING_INFO_EPISODES = 12, CGE_MANUFACTURING_INFO_NEEDS_RESET = 13, CGE_MANUFACTURING_INFO_OVERFLOW = 14,
    CGE_MANUFACTURING_INFO_COMPLETED_TYPE0 = 15   /* .. + 5: products_completed['A'..'F'] (manufacturing_env.py:157,485), the dict info carries (:296) */
};

int cge_manufacturing_create(const cge_manufacturing_config *cfg, int64_t n_envs, int device, int64_t env_index0, cge_manufacturing **out);
int cge_manufacturing_destroy(cge_manufacturing *h);
/* reset(seed=s): env i gets Generator(PCG64(SeedSequence(s_i))); s_i = seeds[i] or base_seed + env_index0 + i */
int cge_manufacturing_seed(cge_manufacturing *h, const uint64_t *seeds, uint64_t base_seed, void *stream);
int cge_manufacturing_reset(cge_manufacturing *h, const uint8_t *mask, float *obs_out, void *stream);
/* actions int32 [n_envs]; a value outside 0..24 is a no-op (falls through every branch of :303-359). */
int cge_manufacturing_step(cge_manufacturing *h, const int32_t *actions, float *obs_out, float *reward_out, uint8_t *terminated_out,
                           uint8_t *truncated_out, float *final_obs_out, void *stream);
/* done_count counts terminated-or-truncated steps; terminated_traj_out gets terminated | truncated << 1 */
int cge_manufacturing_rollout(cge_manufacturing *h, int32_t k_steps, const int32_t *actions, uint64_t action_seed, int64_t t0,
                              float *obs_out, int64_t obs_step_stride, float *reward_traj_out, uint8_t *terminated_traj_out,
                              double *reward_sum_out, int32_t *done_count_out, void *stream);
/* terminal observations of SAME_STEP rollouts, compacted per segment of 64 envs: see <env>_rollout_final_obs at the top of this file */
int cge_manufacturing_rollout_final_obs(cge_manufacturing *h, float *rows_out, int64_t *index_out, int64_t seg_capacity, int32_t *count_out);
int64_t cge_manufacturing_final_obs_segment(const cge_manufacturing *h);
int cge_manufacturing_info(cge_manufacturing *h, int32_t field_id, double *out, void *stream);
/* whole-handle checkpoint (host memory, opaque: 32-byte header + the device arrays in device layout); restores only
 * into a handle created with the same n_envs and config; both calls synchronise `stream` */
size_t cge_manufacturing_snapshot_bytes(const cge_manufacturing *h);
int cge_manufacturing_snapshot_get(cge_manufacturing *h, void *host_buf, void *stream);
int cge_manufacturing_snapshot_set(cge_manufacturing *h, const void *host_buf, void *stream);
size_t cge_manufacturing_device_bytes(const cge_manufacturing *h);
int cge_manufacturing_episode_stats(cge_manufacturing *h, double *return_out, int32_t *length_out);
/* optional: cge_manufacturing_step also writes terminated | truncated per env into done_out (device, n_envs bytes) until it is set to NULL —
 * the mask gymnasium's SAME_STEP infos["_final_obs"] and the episode statistics want, without a second launch */
int cge_manufacturing_done_mask(cge_manufacturing *h, uint8_t *done_out);
const char *cge_manufacturing_last_error(const cge_manufacturing *h);
const char *cge_manufacturing_last_kernel(const cge_manufacturing *h);

/* ------------------------------------------------------------------------------------------ */
/* Hospital  (hospital_management_env/hospital_env.py: HospitalManagementEnv)                   */
/*   obs float32 (243,) (:256-321; the declared space says 295)   action int32 in 0..34 (:165)   */
/*   terminated (:726-742) AND truncated (current_time >= 1440, :357) are both reported.         */
/*   Generator: the process-global CPython `random`; the env never seeds it (:186), so           */
/*   cge_hospital_seed is the caller's random.seed(s_i) for env i.                               */
/* ------------------------------------------------------------------------------------------ */
typedef struct cge_hospital cge_hospital;

typedef struct {
    int32_t max_episode_length;   /* 1440 (:94); <= 2000 */
    int32_t autoreset_mode;       /* CGE_AUTORESET_* */
} cge_hospital_config;

enum { /* cge_hospital_info float64 fields (info dict :362-367 and internal counters) */
    CGE_HOSPITAL_INFO_DEATHS = 0, CGE_HOSPITAL_INFO_PATIENTS_TREATED = 1, CGE_HOSPITAL_INFO_TOTAL_WAIT_TIME = 2, CGE_HOSPITAL_INFO_TIME = 3,
    CGE_HOSPITAL_INFO_OUTBREAK_ACTIVE = 4, CGE_HOSPITAL_INFO_MASS_CASUALTY_EVENT = 5, CGE_HOSPITAL_INFO_NEXT_PATIENT_ID = 6,
    CGE_HOSPITAL_INFO_QUEUE0 = 7, /* .. QUEUE0 + 5: len(patient_queues[Department(d)]) */
    CGE_HOSPITAL_INFO_OCCUPIED_BEDS = 13, CGE_HOSPITAL_INFO_MEDICINE_TOTAL = 14, CGE_HOSPITAL_INFO_EPISODES = 15,
    CGE_HOSPITAL_INFO_NEEDS_RESET = 16, CGE_HOSPITAL_INFO_OVERFLOW = 17
};

int cge_hospital_create(const cge_hospital_config *cfg, int64_t n_envs, int device, int64_t env_index0, cge_hospital **out);
int cge_hospital_destroy(cge_hospital *h);
/* random.seed(s_i) for env i; s_i = seeds[i] or base_seed + env_index0 + i */
int cge_hospital_seed(cge_hospital *h, const uint64_t *seeds, uint64_t base_seed, void *stream);
int cge_hospital_reset(cge_hospital *h, const uint8_t *mask, float *obs_out, void *stream);
/* actions int32 [n_envs]; a value outside 0..34 is a no-op.  truncated_out is REQUIRED. */
int cge_hospital_step(cge_hospital *h, const int32_t *actions, float *obs_out, float *reward_out, uint8_t *terminated_out,
                      uint8_t *truncated_out, float *final_obs_out, void *stream);
/* done_count counts terminated-or-truncated steps; terminated_traj_out gets terminated | truncated << 1 */
int cge_hospital_rollout(cge_hospital *h, int32_t k_steps, const int32_t *actions, uint64_t action_seed, int64_t t0,
                         float *obs_out, int64_t obs_step_stride, float *reward_traj_out, uint8_t *terminated_traj_out,
                         double *reward_sum_out, int32_t *done_count_out, void *stream);
/* terminal observations of SAME_STEP rollouts, compacted per segment of 64 envs: see <env>_rollout_final_obs at the top of this file */
int cge_hospital_rollout_final_obs(cge_hospital *h, float *rows_out, int64_t *index_out, int64_t seg_capacity, int32_t *count_out);
int64_t cge_hospital_final_obs_segment(const cge_hospital *h);
int cge_hospital_info(cge_hospital *h, int32_t field_id, double *out, void *stream);
/* whole-handle checkpoint (host memory, opaque: 32-byte header + the device arrays in device layout); restores only
 * into a handle created with the same n_envs and config; both calls synchronise `stream` */
size_t cge_hospital_snapshot_bytes(const cge_hospital *h);
int cge_hospital_snapshot_get(cge_hospital *h, void *host_buf, void *stream);
int cge_hospital_snapshot_set(cge_hospital *h, const void *host_buf, void *stream);
size_t cge_hospital_device_bytes(const cge_hospital *h);
int cge_hospital_episode_stats(cge_hospital *h, double *return_out, int32_t *length_out);
/* optional: cge_hospital_step also writes terminated | truncated per env into done_out (device, n_envs bytes) until it is set to NULL —
 * the mask gymnasium's SAME_STEP infos["_final_obs"] and the episode statistics want, without a second launch */
int cge_hospital_done_mask(cge_hospital *h, uint8_t *done_out);
const char *cge_hospital_last_error(const cge_hospital *h);
const char *cge_hospital_last_kernel(const cge_hospital *h);

#ifdef __cplusplus
}
#endif
#endif /* CGE_AMD_H */
