// Micro-benchmark: 3R+3W Adam stream over [rows = 128 episodes x 512 co][K = 4608] fp32 matrices walked in TILES (ROWS x SEG
// floats per workgroup, all of a tile's loads issued before its stores), as the fused weight-gradient + Adam kernel does.
// Which tile geometry / launch shape reaches the rate of the contiguous-chunk stream (6.3 TB/s, adam_cus.hip)?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int ROWS, int SEG, bool PERSIST>
__global__ __launch_bounds__(256) void adam_tile(float* __restrict__ w, float* __restrict__ m, float* __restrict__ v, int K, long long n_tiles,
                                                 int tiles_k) {
    constexpr int Q = SEG / 4;                   // float4 per tile row
    constexpr int U = ROWS * Q / 256;            // float4 per thread per array
    for (long long t = blockIdx.x; t < n_tiles; t += PERSIST ? gridDim.x : n_tiles) {
        const long long r0 = (t / tiles_k) * ROWS;
        const int c0 = (int)(t % tiles_k) * SEG;
        f32x4 ww[U], mm[U], vv[U];
        long long off[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int idx = threadIdx.x + u * 256;
            off[u] = ((r0 + idx / Q) * K + c0) / 4 + idx % Q;
            mm[u] = __builtin_nontemporal_load((const f32x4*)m + off[u]);
            vv[u] = __builtin_nontemporal_load((const f32x4*)v + off[u]);
            ww[u] = __builtin_nontemporal_load((const f32x4*)w + off[u]);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float ge = ww[u][e] * 1e-3f;
                mm[u][e] = 0.9f * mm[u][e] + 0.1f * ge;
                vv[u][e] = 0.999f * vv[u][e] + 0.001f * ge * ge;
                ww[u][e] -= 0.01f * (mm[u][e] / (sqrtf(vv[u][e]) + 1e-8f));
            }
            __builtin_nontemporal_store(mm[u], (f32x4*)m + off[u]);
            __builtin_nontemporal_store(vv[u], (f32x4*)v + off[u]);
            __builtin_nontemporal_store(ww[u], (f32x4*)w + off[u]);
        }
    }
}

template <typename F>
float time_ms(F f, int iters = 5) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    f();
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < iters; ++i) f();
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    return ms / iters;
}

template <int ROWS, int SEG>
void run(float* w, float* m, float* v, long long R, int K) {
    const int tiles_k = K / SEG;
    const long long n_tiles = (R / ROWS) * tiles_k;
    const double b6 = 6.0 * R * K * 4;
    float t0 = time_ms([&] { hipLaunchKernelGGL((adam_tile<ROWS, SEG, false>), dim3((unsigned)n_tiles), dim3(256), 0, 0, w, m, v, K, n_tiles, tiles_k); });
    float t1 = time_ms([&] { hipLaunchKernelGGL((adam_tile<ROWS, SEG, true>), dim3(256 * 4), dim3(256), 0, 0, w, m, v, K, n_tiles, tiles_k); });
    float t2 = time_ms([&] { hipLaunchKernelGGL((adam_tile<ROWS, SEG, true>), dim3(256 * 8), dim3(256), 0, 0, w, m, v, K, n_tiles, tiles_k); });
    float t3 = time_ms([&] { hipLaunchKernelGGL((adam_tile<ROWS, SEG, true>), dim3(256 * 3), dim3(256), 0, 0, w, m, v, K, n_tiles, tiles_k); });
    float t4 = time_ms([&] { hipLaunchKernelGGL((adam_tile<ROWS, SEG, true>), dim3(256 * 2), dim3(256), 0, 0, w, m, v, K, n_tiles, tiles_k); });
    // occupancy capped by a dummy dynamic LDS allocation (one tile per workgroup): 50 KB -> 3 workgroups per CU, 36 KB -> 4
    float t5 = time_ms([&] { hipLaunchKernelGGL((adam_tile<ROWS, SEG, false>), dim3((unsigned)n_tiles), dim3(256), 50 * 1024, 0, w, m, v, K, n_tiles, tiles_k); });
    float t6 = time_ms([&] { hipLaunchKernelGGL((adam_tile<ROWS, SEG, false>), dim3((unsigned)n_tiles), dim3(256), 36 * 1024, 0, w, m, v, K, n_tiles, tiles_k); });
    printf("tile %3d rows x %4d floats (%5d B runs, %2d KB per array per workgroup): one tile per workgroup %.2f TB/s | persistent 4/CU %.2f | 8/CU %.2f | 3/CU %.2f | 2/CU %.2f | one-tile, LDS-capped to 3/CU %.2f, 4/CU %.2f\n",
           ROWS, SEG, SEG * 4, ROWS * SEG * 4 / 1024, b6 / t0 / 1e9, b6 / t1 / 1e9, b6 / t2 / 1e9, b6 / t3 / 1e9, b6 / t4 / 1e9, b6 / t5 / 1e9, b6 / t6 / 1e9);
}

int main() {
    const long long R = 128LL * 512;
    const int K = 4608;
    float *w, *m, *v;
    hipMalloc(&w, R * K * 4); hipMalloc(&m, R * K * 4); hipMalloc(&v, R * K * 4);
    hipMemset(w, 0, R * K * 4); hipMemset(m, 0, R * K * 4); hipMemset(v, 0, R * K * 4);
    run<64, 64>(w, m, v, R, K);
    run<32, 128>(w, m, v, R, K);
    run<16, 256>(w, m, v, R, K);
    run<8, 512>(w, m, v, R, K);
    run<32, 256>(w, m, v, R, K);
    return 0;
}
