// mt_seed.hip — device-side seeding of per-env MT19937 streams (gfx950).
//
// Replaces, for a whole batch, the seeding the reference does on process-global generators:
//   CPython  random.seed(s)     (crypto_trading_env.py:307, traffic environment.py:146; snake/parking/
//                                hospital never seed it themselves — per-env protocol)
//   NumPy    np.random.seed(s)  (crypto_trading_env.py:306)
// init_by_array / init_genrand are sequential recurrences (1247 / 623 dependent steps per stream)
// but embarrassingly parallel across envs.  One wavefront seeds 64 streams with the whole state
// held in LDS ([624 words][64 lanes], row padded to 65 words so both the lane-per-env recurrence
// and the env-per-pass write-out are bank-conflict free), then streams each 2.5 KB block to HBM
// with coalesced stores.  162,240 B of the CU's 160 KiB LDS per workgroup.
#include "cge_device.hpp"
#include "cge_host.hpp"

namespace cge {

constexpr int SEED_LANES = 64;
constexpr int SEED_ROW = 65;
constexpr size_t SEED_LDS_BYTES = (size_t)MT_N * SEED_ROW * sizeof(uint32_t);

__global__ __launch_bounds__(SEED_LANES) void mt_seed_kernel(uint32_t *__restrict__ mt, int64_t stride_words, int64_t n,
                                                             const uint64_t *__restrict__ seeds, uint64_t base_seed,
                                                             int64_t env0, int kind) {
    extern __shared__ uint32_t s[];
    const int lane = threadIdx.x;
    const int64_t first = (int64_t)blockIdx.x * SEED_LANES;
    const int64_t i = first + lane;
    const uint64_t seed = i < n ? (seeds ? seeds[i] : base_seed + (uint64_t)(env0 + i)) : 0;
#define S(k) s[(k) * SEED_ROW + lane]
    uint32_t prev = kind == 0 ? 19650218u : (uint32_t)seed;
    S(0) = prev;
    for (int k = 1; k < MT_N; ++k) {
        prev = 1812433253u * (prev ^ (prev >> 30)) + (uint32_t)k;
        S(k) = prev;
    }
    if (kind == 0) {
        const uint32_t key0 = (uint32_t)seed, key1 = (uint32_t)(seed >> 32);
        const bool two = key1 != 0;
        int ii = 1;
        uint32_t j = 0;
        prev = S(0);
        for (int k = MT_N; k; --k) {
            uint32_t cur = S(ii);
            cur = (cur ^ ((prev ^ (prev >> 30)) * 1664525u)) + (j ? key1 : key0) + j;
            S(ii) = cur;
            prev = cur;
            ++ii;
            j = two ? (j ^ 1u) : 0u;
            if (ii >= MT_N) { S(0) = prev; ii = 1; }
        }
        for (int k = MT_N - 1; k; --k) {
            uint32_t cur = S(ii);
            cur = (cur ^ ((prev ^ (prev >> 30)) * 1566083941u)) - (uint32_t)ii;
            S(ii) = cur;
            prev = cur;
            ++ii;
            if (ii >= MT_N) { S(0) = prev; ii = 1; }
        }
        S(0) = 0x80000000u;
    }
#undef S
    __syncthreads();
    const int live = (int)((n - first) < SEED_LANES ? (n - first) : SEED_LANES);
    for (int e = 0; e < live; ++e) {
        uint32_t *dst = mt + (first + e) * stride_words;
        for (int w = lane; w < MT_STRIDE; w += SEED_LANES) dst[w] = s[(w < MT_N ? w : w - MT_N) * SEED_ROW + e];   // 624.. mirror 0..15
    }
}

hipError_t launch_mt_seed(uint32_t *mt, int64_t stride_words, int64_t n, const uint64_t *seeds, uint64_t base_seed,
                          int64_t env0, int kind, hipStream_t stream) {
    // per-device attribute; cheap enough to set on every (rare) seeding call
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(mt_seed_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)SEED_LDS_BYTES);
    if (e != hipSuccess) return e;
    const unsigned blocks = (unsigned)((n + SEED_LANES - 1) / SEED_LANES);
    hipLaunchKernelGGL(mt_seed_kernel, dim3(blocks), dim3(SEED_LANES), SEED_LDS_BYTES, stream, mt, stride_words, n, seeds,
                       base_seed, env0, kind);
    return hipGetLastError();
}

}  // namespace cge
