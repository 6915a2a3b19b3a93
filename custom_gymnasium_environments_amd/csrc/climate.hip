// climate.hip — batched SmartClimateEnv for MI355X (gfx950): kernels + C ABI (include/cge_amd.h).
//
// Re-expresses /root/reference/smartclimate_rl-main/smartclimate/ for N independent instances, one lane per env:
//   env.py    _init_state :48-61, reset :63-72, _get_obs :74-83, step :85-116
//   utils.py  get_outside_temp :5-13, update_occupancy :15-22, room_temp_dynamics :24-28, calculate_reward :30-50
// State per env: 80 B in five uint4 columns — the env's private PCG64 (state, inc, buffered half-word), room and
// outside temperature, energy and reward accumulators (float64), the float32 AC setting, step, people, lights.
// Dynamics in float64 in the reference's operation order; the only place the device can differ from the CPU
// is the last place of log1p/exp in the ziggurat's wedge/tail (1.5 % of normals), which reaches the float32 obs
// in < 1e-6 of values (room temperature is a contraction: no error growth).  9 float32 obs staged [64][9] in LDS.
#include <cstring>
#include <vector>

#include "cge_device.hpp"
#include "cge_host.hpp"
#include "cge_pcg.hpp"

namespace cge {
namespace climate {

constexpr int OBS = 9;
constexpr int COLS = 5;
constexpr int BLOCK = 64;

struct Params {
    uint4 *state;
    int64_t n, env0;
    int32_t mode, max_occ, max_steps;
    const float *ac;
    const int8_t *lights;
    const uint8_t *mask;
    const uint64_t *seeds;
    uint64_t base_seed;
    float *obs, *final_obs, *reward;
    FinalSeg fin;          // fused rollouts (SAME_STEP): terminal rows compacted per wave (cge_climate_rollout_final_obs); rows nullable
    uint8_t *terminated, *truncated;
    int32_t k_steps;
    uint64_t a_seed;
    int64_t t0, obs_step_stride;
    double *reward_sum;
    int32_t *done_count;
    double *ep_ret;       // episode statistics (cge_climate_episode_stats), nullable
    int32_t *ep_len;
};

__device__ __forceinline__ double mk_double(uint32_t lo, uint32_t hi) { return __hiloint2double((int)hi, (int)lo); }

struct Env {
    Pcg64 g;
    double room, outside, energy, total_reward;
    float ac;
    uint32_t step, people, lights, needs_reset, comfort_time, episodes;

    __device__ __forceinline__ void load(const uint4 *__restrict__ s, int64_t n, int64_t i) {
        const uint4 a = s[i], b = s[n + i], c = s[2 * n + i], d = s[3 * n + i], m = s[4 * n + i];
        g.state = ((u128)(((uint64_t)a.w << 32) | a.z) << 64) | (((uint64_t)a.y << 32) | a.x);
        g.inc = ((u128)(((uint64_t)b.w << 32) | b.z) << 64) | (((uint64_t)b.y << 32) | b.x);
        room = mk_double(c.x, c.y); outside = mk_double(c.z, c.w);
        energy = mk_double(d.x, d.y); total_reward = mk_double(d.z, d.w);
        step = m.x & 0xFFFFu; people = (m.x >> 16) & 15u; lights = (m.x >> 20) & 15u; g.has_uint32 = (m.x >> 24) & 1u; needs_reset = (m.x >> 25) & 1u;
        g.uinteger = m.y; comfort_time = m.z & 0xFFFFu; episodes = m.z >> 16; ac = __uint_as_float(m.w);
    }
    __device__ __forceinline__ void store(uint4 *__restrict__ s, int64_t n, int64_t i) const {
        const uint64_t sl = (uint64_t)g.state, sh = (uint64_t)(g.state >> 64), il = (uint64_t)g.inc, ih = (uint64_t)(g.inc >> 64);
        s[i] = make_uint4((uint32_t)sl, (uint32_t)(sl >> 32), (uint32_t)sh, (uint32_t)(sh >> 32));
        s[n + i] = make_uint4((uint32_t)il, (uint32_t)(il >> 32), (uint32_t)ih, (uint32_t)(ih >> 32));
        s[2 * n + i] = make_uint4((uint32_t)__double2loint(room), (uint32_t)__double2hiint(room), (uint32_t)__double2loint(outside), (uint32_t)__double2hiint(outside));
        s[3 * n + i] = make_uint4((uint32_t)__double2loint(energy), (uint32_t)__double2hiint(energy), (uint32_t)__double2loint(total_reward),
                                  (uint32_t)__double2hiint(total_reward));
        s[4 * n + i] = make_uint4(step | (people << 16) | (lights << 20) | (g.has_uint32 << 24) | (needs_reset << 25), g.uinteger,
                                  (comfort_time & 0xFFFFu) | (episodes << 16), __float_as_uint(ac));
    }
};

__device__ __forceinline__ double outside_temp(double tod, Pcg64 &g) {                  // utils.py:5-13
    const double base = (0.0 <= tod && tod < 8.0) ? 25.0 : (8.0 <= tod && tod < 16.0) ? 45.0 : 35.0;
    return g.normal(base, 5.0);
}

__device__ __forceinline__ void do_reset(Env &e, int32_t max_occ) {                     // env.py:63-72 + _init_state :48-61
    e.step = 0;
    e.needs_reset = 0;
    e.room = e.g.uniform(22.0, 26.0);
    e.people = (uint32_t)e.g.integers(0, (int64_t)max_occ + 1);
    e.outside = outside_temp(0.0, e.g);
    e.ac = 24.0f;
    e.lights = 0;
    e.total_reward = 0.0; e.comfort_time = 0; e.energy = 0.0;
}

__device__ __forceinline__ bool env_step(Env &e, int32_t max_occ, int32_t max_steps, float ac_in, uint32_t lights, double &reward) {   // :85-116
    const float acf = ac_in < 16.0f ? 16.0f : (ac_in > 32.0f ? 32.0f : ac_in);           // np.clip on np.float32, then float()
    e.ac = acf;
    e.lights = lights;
    e.step += 1;
    const double tod = (double)(e.step % 1440u) / 60.0;
    e.outside = outside_temp(tod, e.g);
    int change;                                                                          // utils.py:15-22
    if (9.0 <= tod && tod < 18.0) change = e.g.choice4(0.1, 0.3, 0.4, 0.2) - 1;
    else change = e.g.choice4(0.2, 0.4, 0.3, 0.1) - 2;
    const int np_ = (int)e.people + change;
    e.people = (uint32_t)(np_ < 0 ? 0 : (np_ > max_occ ? max_occ : np_));
    const double prev = e.room, acd = (double)acf;                                       // utils.py:24-28
    const double temp = prev + 0.1 * (e.outside - prev) + 0.2 * (acd - prev) + (double)e.people * 1.0;
    e.room = temp < 10.0 ? 10.0 : (temp > 50.0 ? 50.0 : temp);
    const double T = e.room;                                                             // utils.py:30-50
    double comfort;
    if (20.0 <= T && T <= 24.0) comfort = 10.0;
    else if (18.0 <= T && T <= 26.0) comfort = 5.0;
    else if (16.0 <= T && T <= 28.0) comfort = 0.0;
    else comfort = -15.0 * fabs(T - 22.0);
    const double ac_pen = -0.5 * fabs(acd - e.outside);
    int required = ((int)e.people + 1) / 2;
    required = required > 4 ? 4 : required;
    const int on = __popc(lights & 15u);
    const int light_pen = -1 * (on - required > 0 ? on - required : 0);
    const double r = comfort + ac_pen + (double)light_pen;
    e.total_reward += r;
    e.comfort_time += (20.0 <= T && T <= 24.0) ? 1u : 0u;
    e.energy += fabs(acd - e.outside) + (double)on;
    reward = r;
    return e.step >= (uint32_t)max_steps;
}

__device__ __forceinline__ void observe(const Env &e, const RowMap &rm, float *__restrict__ dst, uint32_t *__restrict__ tile) {
    const uint32_t lane = threadIdx.x & 63u;
    float *row = reinterpret_cast<float *>(tile) + lane * OBS;                           // env.py:74-83
    row[0] = (float)e.room; row[1] = (float)e.people; row[2] = (float)((double)(e.step % 1440u) / 60.0);
    row[3] = (float)e.outside; row[4] = e.ac;
#pragma unroll
    for (int k = 0; k < 4; ++k) row[5 + k] = (float)((e.lights >> k) & 1u);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    uint32_t r = lane / (uint32_t)OBS, col = lane - r * (uint32_t)OBS;
#pragma unroll 1
    for (int m = 0; m < OBS; ++m) {
        int64_t to;
        if (rm.row(r, to)) reinterpret_cast<uint32_t *>(dst)[to * OBS + col] = tile[r * OBS + col];
        col += 64u % OBS; r += 64u / OBS;
        if (col >= (uint32_t)OBS) { col -= OBS; r += 1u; }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

template <bool ROLLOUT>
__global__ __launch_bounds__(BLOCK, 4) void step_kernel(Params p) {   // four waves per SIMD (128 registers: round 3 took 127, round 4, left alone, 129)
    __shared__ uint32_t tile[64 * OBS];
    const int64_t i0 = (int64_t)blockIdx.x * BLOCK;
    const int64_t i = i0 + threadIdx.x;
    const bool live = i < p.n;
    const int64_t li = live ? i : i0;
    const int64_t nrows = p.n - i0 < 64 ? p.n - i0 : 64;
    Env e;
    e.load(p.state, p.n, li);
    const uint64_t key = ROLLOUT ? hash_env_key(p.a_seed, (uint64_t)(p.env0 + li)) : 0;
    double rsum = 0.0;
    int32_t dcount = 0;
    uint32_t fin_used = 0;                                         // terminal rows this wave has delivered to its segment (fused rollouts)
    const int ksteps = ROLLOUT ? p.k_steps : 1;
#pragma unroll 1
    for (int t = 0; t < ksteps; ++t) {
        double reward = 0.0;
        bool term = false, reset_now = false;
        if (live) {
            if (p.mode == CGE_AUTORESET_NEXT_STEP && e.needs_reset) {
                reset_now = true;
            } else {
                float ac;
                uint32_t lights = 0;
                if (p.ac) {
                    ac = p.ac[(int64_t)t * p.n + i];
                    const uint32_t lw = reinterpret_cast<const uint32_t *>(p.lights)[(int64_t)t * p.n + i];   // 4 int8 = one dword
#pragma unroll
                    for (int k = 0; k < 4; ++k) lights |= (((lw >> (8 * k)) & 0xFFu) ? 1u : 0u) << k;
                } else {
                    const uint64_t u = mix64(key + (uint64_t)(p.t0 + t) * 0xD1342543DE82EF95ull + 0) >> 40;
                    ac = (float)(16.0 + 16.0 * ((double)u / 16777216.0));
#pragma unroll
                    for (int k = 0; k < 4; ++k) lights |= hash_action_from_key(key, (uint64_t)(p.t0 + t), 2u, (uint32_t)(1 + k)) << k;
                }
                term = env_step(e, p.max_occ, p.max_steps, ac, lights, reward);
                if (term) {
                    e.episodes += 1;
                    if (p.ep_ret) p.ep_ret[i] = e.total_reward;            // env.py:108 accumulates it, reset() zeroes it
                    if (p.ep_len) p.ep_len[i] = (int32_t)e.step;
                    if (p.mode == CGE_AUTORESET_SAME_STEP) reset_now = true;
                    else if (p.mode == CGE_AUTORESET_NEXT_STEP) e.needs_reset = 1;
                }
            }
        }
        const bool fin = live && term && reset_now;
        const unsigned long long fin_mask = __ballot(fin);
        if (fin_mask) {                                            // terminal rows: step() -> final_obs_out; fused rollout -> the wave's segment
            if (!ROLLOUT) {
                if (p.final_obs) observe(e, RowMap{fin_mask, nrows, 0, false}, p.final_obs + i0 * OBS, tile);
            } else if (p.fin.rows) {
                float *fdst;
                const RowMap rm = final_rows<float>(p.fin, (int64_t)blockIdx.x, fin_used, fin, fin_mask, nrows, t, i, OBS, fdst);
                observe(e, rm, fdst, tile);
            }
            fin_used += (uint32_t)__popcll(fin_mask);
        }
        if (reset_now) do_reset(e, p.max_occ);
        if (p.obs) observe(e, RowMap{~0ull, nrows, 0, false}, p.obs + (int64_t)t * p.obs_step_stride + i0 * OBS, tile);
        if (live) {
            if (ROLLOUT) {
                rsum += reward;
                dcount += term ? 1 : 0;
                if (p.reward) p.reward[(int64_t)t * p.n + i] = (float)reward;
                if (p.terminated) p.terminated[(int64_t)t * p.n + i] = term ? 1 : 0;
            } else {
                p.reward[i] = (float)reward;
                p.terminated[i] = term ? 1 : 0;
                if (p.truncated) p.truncated[i] = 0;
            }
        }
    }
    if (live) {
        e.store(p.state, p.n, i);
        if (ROLLOUT) {
            if (p.reward_sum) p.reward_sum[i] = rsum;
            if (p.fin.count && threadIdx.x == 0) p.fin.count[blockIdx.x] = (int32_t)fin_used;
            if (p.done_count) p.done_count[i] = dcount;
        }
    }
}

// what: 0 = reset(mask) + obs, 1 = reseed the generators (default_rng(seed_i)); the env state is untouched
__global__ __launch_bounds__(BLOCK) void reset_kernel(Params p, int what) {
    __shared__ uint32_t tile[64 * OBS];
    const int64_t i0 = (int64_t)blockIdx.x * BLOCK;
    const int64_t i = i0 + threadIdx.x;
    const bool live = i < p.n;
    const int64_t nrows = p.n - i0 < 64 ? p.n - i0 : 64;
    Env e;
    e.load(p.state, p.n, live ? i : i0);
    if (live) {
        if (what == 1) {
            e.g.seed(p.seeds ? p.seeds[i] : p.base_seed + (uint64_t)(p.env0 + i));
            e.store(p.state, p.n, i);
        } else if (!p.mask || p.mask[i]) {
            do_reset(e, p.max_occ);
            e.store(p.state, p.n, i);
        }
    }
    if (what == 0 && p.obs) observe(e, RowMap{~0ull, nrows, 0, false}, p.obs + i0 * OBS, tile);
}

__global__ __launch_bounds__(256) void info_kernel(const uint4 *__restrict__ state, int64_t n, int field, double *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    Env e;
    e.load(state, n, i);
    double v = 0.0;
    switch (field) {
        case CGE_CLIMATE_INFO_ROOM_TEMP: v = e.room; break;
        case CGE_CLIMATE_INFO_OUTSIDE_TEMP: v = e.outside; break;
        case CGE_CLIMATE_INFO_AC_SETTING: v = (double)e.ac; break;
        case CGE_CLIMATE_INFO_ENERGY_USAGE: v = e.energy; break;
        case CGE_CLIMATE_INFO_TOTAL_REWARD: v = e.total_reward; break;
        case CGE_CLIMATE_INFO_NUM_PEOPLE: v = e.people; break;
        case CGE_CLIMATE_INFO_STEP: v = e.step; break;
        case CGE_CLIMATE_INFO_COMFORT_TIME: v = e.comfort_time; break;
        case CGE_CLIMATE_INFO_EPISODES: v = e.episodes; break;
        case CGE_CLIMATE_INFO_NEEDS_RESET: v = e.needs_reset; break;
    }
    out[i] = v;
}

}  // namespace climate
}  // namespace cge

using namespace cge;

struct cge_climate : HandleBase {
    cge_climate_config cfg{};
    uint4 *state = nullptr;
    static constexpr uint32_t snap_tag = 2u;
    std::vector<std::pair<void *, size_t>> blobs() const { return {{state, (size_t)climate::COLS * n * sizeof(uint4)}}; }
    uint32_t snap_extra() const { return 0u; }
    void set_snap_extra(uint32_t v) { (void)v; }
    climate::Params params() const {
        climate::Params p{};
        p.state = state; p.n = n; p.env0 = env0; p.mode = cfg.autoreset_mode; p.max_occ = cfg.max_occupancy; p.max_steps = cfg.episode_minutes;
        p.ep_ret = ep_ret; p.ep_len = ep_len;
        return p;
    }
    unsigned blocks() const { return (unsigned)((n + climate::BLOCK - 1) / climate::BLOCK); }
};

extern "C" {

int cge_climate_create(const cge_climate_config *cfg, int64_t n_envs, int device, int64_t env_index0, cge_climate **out) {
    if (!cfg || !out || n_envs <= 0 || env_index0 < 0) return CGE_ERR_INVALID_ARG;
    *out = nullptr;
    if (cfg->autoreset_mode < 0 || cfg->autoreset_mode > 2 || cfg->max_occupancy < 0 || cfg->max_occupancy > 15 || cfg->episode_minutes < 0 ||
        cfg->episode_minutes > 65535)
        return CGE_ERR_INVALID_ARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return CGE_ERR_NO_DEVICE;
    cge_climate *h = new cge_climate();
    h->cfg = *cfg;
    if (h->cfg.max_occupancy == 0) h->cfg.max_occupancy = 8;
    if (h->cfg.episode_minutes == 0) h->cfg.episode_minutes = 1440;
    h->n = n_envs; h->env0 = env_index0; h->device = device;
    DeviceGuard g(device);
    const size_t sb = (size_t)climate::COLS * n_envs * sizeof(uint4);
    hipError_t e;
    if ((e = hipMalloc(&h->state, sb)) != hipSuccess || (e = hipMemset(h->state, 0, sb)) != hipSuccess) {
        (void)hipFree(h->state);
        delete h;
        return CGE_ERR_HIP;
    }
    h->device_bytes = sb;
    climate::Params p = h->params();                       // default generators: default_rng(env_index0 + i); no reset (fresh env)
    hipLaunchKernelGGL(climate::reset_kernel, dim3(h->blocks()), dim3(climate::BLOCK), 0, nullptr, p, 1);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
    if (e != hipSuccess) {
        (void)hipFree(h->state);
        delete h;
        return CGE_ERR_HIP;
    }
    *out = h;
    return CGE_OK;
}

int cge_climate_destroy(cge_climate *h) {
    if (!h) return CGE_ERR_INVALID_ARG;
    DeviceGuard g(h->device);
    (void)hipDeviceSynchronize();
    (void)hipFree(h->state);
    delete h;
    return CGE_OK;
}

int cge_climate_seed(cge_climate *h, const uint64_t *seeds, uint64_t base_seed, void *stream) {
    if (!h) return CGE_ERR_INVALID_ARG;
    DeviceGuard g(h->device);
    climate::Params p = h->params();
    p.seeds = seeds; p.base_seed = base_seed;
    hipLaunchKernelGGL(climate::reset_kernel, dim3(h->blocks()), dim3(climate::BLOCK), 0, as_stream(stream), p, 1);
    CGE_TRY(h, hipGetLastError());
    return CGE_OK;
}

int cge_climate_reset(cge_climate *h, const uint8_t *mask, float *obs_out, void *stream) {
    if (!h) return CGE_ERR_INVALID_ARG;
    DeviceGuard g(h->device);
    climate::Params p = h->params();
    p.mask = mask; p.obs = obs_out;
    hipLaunchKernelGGL(climate::reset_kernel, dim3(h->blocks()), dim3(climate::BLOCK), 0, as_stream(stream), p, 0);
    CGE_TRY(h, hipGetLastError());
    return CGE_OK;
}

int cge_climate_step(cge_climate *h, const float *ac_temp, const int8_t *lights, float *obs_out, float *reward_out, uint8_t *terminated_out,
                     uint8_t *truncated_out, float *final_obs_out, void *stream) {
    if (!h) return CGE_ERR_INVALID_ARG;
    if (!ac_temp || !lights || !obs_out || !reward_out || !terminated_out)
        return h->fail(CGE_ERR_INVALID_ARG, "cge_climate_step: null ac_temp/lights/obs/reward/terminated pointer");
    DeviceGuard g(h->device);
    climate::Params p = h->params();
    p.ac = ac_temp; p.lights = lights; p.obs = obs_out; p.reward = reward_out; p.terminated = terminated_out; p.truncated = truncated_out;
    p.final_obs = final_obs_out; p.k_steps = 1;
    hipLaunchKernelGGL(climate::step_kernel<false>, dim3(h->blocks()), dim3(climate::BLOCK), 0, as_stream(stream), p);
    h->last_kernel = "cge::climate::step_kernel<false>";
    CGE_TRY(h, hipGetLastError());
    return CGE_OK;
}

int cge_climate_rollout(cge_climate *h, int32_t k_steps, const float *ac_temp, const int8_t *lights, uint64_t action_seed, int64_t t0,
                        float *obs_out, int64_t obs_step_stride, float *reward_traj_out, uint8_t *terminated_traj_out, double *reward_sum_out,
                        int32_t *done_count_out, void *stream) {
    if (!h) return CGE_ERR_INVALID_ARG;
    if (k_steps < 0 || obs_step_stride < 0 || (obs_step_stride != 0 && obs_step_stride < h->n * climate::OBS) || ((ac_temp == nullptr) != (lights == nullptr)))
        return h->fail(CGE_ERR_INVALID_ARG, "cge_climate_rollout: bad k_steps / obs_step_stride / actions");
    if (k_steps == 0) return CGE_OK;
    DeviceGuard g(h->device);
    climate::Params p = h->params();
    p.k_steps = k_steps; p.ac = ac_temp; p.lights = lights; p.a_seed = action_seed; p.t0 = t0; p.obs = obs_out; p.obs_step_stride = obs_step_stride;
    p.reward = reward_traj_out; p.terminated = terminated_traj_out; p.reward_sum = reward_sum_out; p.done_count = done_count_out;
    p.fin = FinalSeg{h->fin_rows, h->fin_index, h->fin_count, h->fin_cap, h->n};
    hipLaunchKernelGGL(climate::step_kernel<true>, dim3(h->blocks()), dim3(climate::BLOCK), 0, as_stream(stream), p);
    h->last_kernel = "cge::climate::step_kernel<true>";
    CGE_TRY(h, hipGetLastError());
    return CGE_OK;
}

CGE_DEFINE_FINAL_OBS(climate, float, 64)

int cge_climate_info(cge_climate *h, int32_t field_id, double *out, void *stream) {
    if (!h || !out || field_id < 0 || field_id > CGE_CLIMATE_INFO_NEEDS_RESET) return CGE_ERR_INVALID_ARG;
    DeviceGuard g(h->device);
    hipLaunchKernelGGL(climate::info_kernel, dim3((unsigned)((h->n + 255) / 256)), dim3(256), 0, as_stream(stream), h->state, h->n, field_id, out);
    CGE_TRY(h, hipGetLastError());
    return CGE_OK;
}

size_t cge_climate_snapshot_bytes(const cge_climate *h) { return h ? snapshot_bytes(h) : 0; }
int cge_climate_snapshot_get(cge_climate *h, void *host_buf, void *stream) { return snapshot_get(h, host_buf, as_stream(stream)); }
int cge_climate_snapshot_set(cge_climate *h, const void *host_buf, void *stream) { return snapshot_set(h, host_buf, as_stream(stream)); }
size_t cge_climate_device_bytes(const cge_climate *h) { return h ? h->device_bytes : 0; }
int cge_climate_episode_stats(cge_climate *h, double *return_out, int32_t *length_out) {
    if (!h) return CGE_ERR_INVALID_ARG;
    h->ep_ret = return_out; h->ep_len = length_out;
    return CGE_OK;
}

const char *cge_climate_last_error(const cge_climate *h) { return h ? h->last_error.c_str() : "null handle"; }

const char *cge_climate_last_kernel(const cge_climate *h) { return h ? h->last_kernel.c_str() : ""; }

}  // extern "C"
