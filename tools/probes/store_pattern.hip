// store_pattern.hip — what does the snake rollout's obs write pattern cost on its own?  (measurement tool, not product code)
// Each workgroup owns a 25,600-byte tile (256 envs x 100 B) and writes it once per step into a [K, N, 100] trajectory, i.e.
// K writes of 25.6 KB that are N*100 bytes apart.  Variants: writer lanes per workgroup, resident workgroups per CU (dummy LDS),
// non-temporal stores, ring depth (step t goes to slab t % R), and the "ideal" order (one workgroup per (t, tile)).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/store_pattern tools/probes/store_pattern.hip && /tmp/store_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

constexpr int TILE_B = 25600, TILE_V = TILE_B / 16;   // 1600 uint4 per tile
typedef unsigned int v4u __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void nt_store(uint4 v, uint4 *p) {
    v4u w = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(w, reinterpret_cast<v4u *>(p));
}

template <int THREADS, bool NT, int LDS_BYTES>
__global__ __launch_bounds__(THREADS) void tile_per_step(uint4 *dst, long long slab_v, int k_steps, int ring, int ntiles, int delay) {
    __shared__ uint4 pad[LDS_BYTES / 16 > 0 ? LDS_BYTES / 16 : 1];
    if (LDS_BYTES > 0 && threadIdx.x == 0 && k_steps < 0) pad[0] = make_uint4(1, 2, 3, 4);   // keep the allocation
    const int tile = blockIdx.x;
    if (tile >= ntiles) return;
    uint4 v = make_uint4(threadIdx.x, tile, 0, 1);
    for (int t = 0; t < k_steps; ++t) {
        uint4 *p = dst + (long long)(t % ring) * slab_v + (long long)tile * TILE_V;
        v.z = t;
#pragma unroll
        for (int q = 0; q < (TILE_V + THREADS - 1) / THREADS; ++q) {
            const int idx = q * THREADS + threadIdx.x;
            if (idx < TILE_V) {
                if (NT) nt_store(v, p + idx);
                else p[idx] = v;
            }
        }
        if (delay) __builtin_amdgcn_s_sleep(127);    // ~compute phase between two tiles
    }
}

template <bool NT>
__global__ __launch_bounds__(256) void ideal_order(uint4 *dst, long long slab_v, int ring, int ntiles) {
    const long long b = blockIdx.x;
    const int t = (int)(b / ntiles), tile = (int)(b % ntiles);
    uint4 *p = dst + (long long)(t % ring) * slab_v + (long long)tile * TILE_V;
    const uint4 v = make_uint4(threadIdx.x, tile, t, 1);
    for (int idx = threadIdx.x; idx < TILE_V; idx += 256) {
        if (NT) nt_store(v, p + idx);
        else p[idx] = v;
    }
}

template <class F>
float time_ms(F f, int reps = 3) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int r = 0; r < reps; ++r) f();
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
}

int main() {
    const int ntiles = 4096, K = 200;
    const long long slab_v = (long long)ntiles * TILE_V;
    uint4 *buf;
    CK(hipMalloc(&buf, (size_t)K * slab_v * 16 + (64u << 20) + (size_t)K * (2 << 20) + (1 << 20)));
    CK(hipMemset(buf, 0, (size_t)K * slab_v * 16));
    const double mb_step = ntiles * (double)TILE_B / 1e6;
    auto report = [&](const char *name, float ms, int steps) {
        printf("%-72s %8.2f us/step  %6.2f TB/s\n", name, ms * 1e3 / steps, mb_step * steps / ms / 1e3);
    };
#define RUN(NAME, THREADS, NT, LDS, RING, DELAY) \
    report(NAME, time_ms([&] { hipLaunchKernelGGL((tile_per_step<THREADS, NT, LDS>), dim3(ntiles), dim3(THREADS), 0, 0, buf, slab_v, K, RING, ntiles, DELAY); }), K)
    RUN("64 writers/WG, 6 WG/CU, trajectory", 64, false, 25600, K, 0);
    // does the slab stride matter?  (N*100 bytes = 100 MiB exactly: steps that are in flight together differ by multiples of 2^20)
    {
        const long long pads[] = {16, 256 / 16 * 17, 4096 / 16 + 16, 65536 / 16 + 48, (2 << 20) / 16 + 272};
        for (long long pad : pads) {
            const long long sv = slab_v + pad;
            char name[96];
            snprintf(name, sizeof name, "64 writers/WG, 6 WG/CU, trajectory, slab stride + %lld B", pad * 16);
            report(name, time_ms([&] { hipLaunchKernelGGL((tile_per_step<64, false, 25600>), dim3(ntiles), dim3(64), 0, 0, buf, sv, K, K, ntiles, 0); }), K);
        }
    }
    RUN("64 writers/WG, 6 WG/CU, trajectory, nt", 64, true, 25600, K, 0);
    RUN("64 writers/WG, 6 WG/CU, in place (ring 1)", 64, false, 25600, 1, 0);
    RUN("64 writers/WG, 6 WG/CU, ring 4", 64, false, 25600, 4, 0);
    RUN("64 writers/WG, 6 WG/CU, ring 16", 64, false, 25600, 16, 0);
    RUN("64 writers/WG, 6 WG/CU, trajectory, sleep between tiles", 64, false, 25600, K, 1);
    RUN("64 writers/WG, 12 WG/CU, trajectory", 64, false, 12800, K, 0);
    RUN("64 writers/WG, all resident, trajectory", 64, false, 0, K, 0);
    RUN("256 writers/WG, 6 WG/CU, trajectory", 256, false, 25600, K, 0);
    RUN("256 writers/WG, 6 WG/CU, trajectory, nt", 256, true, 25600, K, 0);
    RUN("256 writers/WG, all resident, trajectory", 256, false, 0, K, 0);
    report("ideal order: one WG per (t, tile), plain", time_ms([&] { hipLaunchKernelGGL((ideal_order<false>), dim3(ntiles * K), dim3(256), 0, 0, buf, slab_v, K, ntiles); }), K);
    report("ideal order: one WG per (t, tile), nt", time_ms([&] { hipLaunchKernelGGL((ideal_order<true>), dim3(ntiles * K), dim3(256), 0, 0, buf, slab_v, K, ntiles); }), K);
    report("hipMemsetAsync of the whole trajectory", time_ms([&] { CK(hipMemsetAsync(buf, 1, (size_t)K * slab_v * 16, 0)); }), K);
    CK(hipFree(buf));
    return 0;
}
