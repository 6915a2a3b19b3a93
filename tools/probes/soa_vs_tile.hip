// soa_vs_tile.hip — does the column-major SoA state layout (state[c*N + i], a wave's C columns N*16 bytes apart) cost HBM
// bandwidth against a tile-major one (state[(tile*C + c)*64 + lane], a wave's record contiguous)?  (measurement tool)
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/soa_vs_tile tools/probes/soa_vs_tile.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int C, bool TILE, bool WRITE>
__global__ __launch_bounds__(64) void k(uint4 *base, long long n, uint4 *sink) {
    const long long i = (long long)blockIdx.x * 64 + threadIdx.x;
    uint4 acc = make_uint4(0, 0, 0, 0);
    uint4 v[C];
#pragma unroll
    for (int c = 0; c < C; ++c) v[c] = TILE ? base[((long long)blockIdx.x * C + c) * 64 + threadIdx.x] : base[(long long)c * n + i];
#pragma unroll
    for (int c = 0; c < C; ++c) { acc.x += v[c].x; acc.y ^= v[c].y; acc.z += v[c].z; acc.w ^= v[c].w; }
    if (WRITE) {
#pragma unroll
        for (int c = 0; c < C; ++c) {
            uint4 w = v[c]; w.x += acc.y;
            if (TILE) base[((long long)blockIdx.x * C + c) * 64 + threadIdx.x] = w; else base[(long long)c * n + i] = w;
        }
    } else if (acc.x == 0x12345678u && acc.y == 0x9abcdef0u) sink[i] = acc;
}

template <class F> float ms(F f) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a)); for (int r = 0; r < 5; ++r) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float t; CK(hipEventElapsedTime(&t, a, b)); return t / 5;
}

template <int C> void run(uint4 *buf, uint4 *sink, long long n) {
    const double gb = (double)C * n * 16 / 1e9;
    const unsigned blocks = (unsigned)(n / 64);
    float t;
    t = ms([&] { hipLaunchKernelGGL((k<C, false, false>), dim3(blocks), dim3(64), 0, 0, buf, n, sink); }); printf("C=%2d n=%lld  read   column-major %7.1f us %5.2f TB/s\n", C, n, t * 1e3, gb / t);
    t = ms([&] { hipLaunchKernelGGL((k<C, true, false>), dim3(blocks), dim3(64), 0, 0, buf, n, sink); });  printf("C=%2d n=%lld  read   tile-major   %7.1f us %5.2f TB/s\n", C, n, t * 1e3, gb / t);
    t = ms([&] { hipLaunchKernelGGL((k<C, false, true>), dim3(blocks), dim3(64), 0, 0, buf, n, sink); });  printf("C=%2d n=%lld  r+w    column-major %7.1f us %5.2f TB/s moved\n", C, n, t * 1e3, 2 * gb / t);
    t = ms([&] { hipLaunchKernelGGL((k<C, true, true>), dim3(blocks), dim3(64), 0, 0, buf, n, sink); });   printf("C=%2d n=%lld  r+w    tile-major   %7.1f us %5.2f TB/s moved\n", C, n, t * 1e3, 2 * gb / t);
}

int main() {
    const long long nmax = 1 << 20;
    uint4 *buf, *sink;
    CK(hipMalloc(&buf, (size_t)50 * nmax * 16)); CK(hipMemset(buf, 1, (size_t)50 * nmax * 16));
    CK(hipMalloc(&sink, (size_t)nmax * 16));
    run<3>(buf, sink, 1 << 20);      // snake round 1
    run<15>(buf, sink, 1 << 18);     // traffic
    run<48>(buf, sink, 1 << 17);     // hospital
    run<50>(buf, sink, 1 << 20);     // crypto ohlv
    return 0;
}
