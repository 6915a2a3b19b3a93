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
 *   <env>_info     <->  the `info` dicts        snake_env.py:63,117
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
    int32_t grid_size;      /* reference default 20 (snake_env.py:19); BASELINE configs use 10 */
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
 * (nullable): per-step [k_steps, n_envs] outputs, i.e. exactly what k step() calls would return;
 * reward_sum_out / done_count_out (nullable) accumulate per env over the k steps. */
int cge_snake_rollout(cge_snake *h, int32_t k_steps, const int32_t *actions, uint64_t action_seed, int64_t t0,
                      int8_t *obs_out, int64_t obs_step_stride, float *reward_traj_out,
                      uint8_t *terminated_traj_out, float *reward_sum_out, int32_t *done_count_out,
                      void *stream);
int cge_snake_info(cge_snake *h, int32_t field_id, int32_t *out, void *stream);
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
const char *cge_snake_last_error(const cge_snake *h);

#ifdef __cplusplus
}
#endif
#endif /* CGE_AMD_H */
