// Micro-benchmark: does the LAYOUT of Adam's state matter?  (a) three separate arrays w, m, v (the engine's slabs); (b) one array
// of [tile][w | m | v] blocks (16 KB each), so that a tile's 48 KB of reads and 48 KB of writes are one contiguous range.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

// (c) w separate (the forward / data-gradient kernels read it alone), m and v interleaved per tile: [tile][m | v]
__global__ __launch_bounds__(256) void adam_tile_mv(float* __restrict__ w, float* __restrict__ mv, long long n_tiles) {
    for (long long t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        f32x4 ww[4], mm[4], vv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long long i = threadIdx.x + u * 256;
            mm[u] = __builtin_nontemporal_load((const f32x4*)mv + t * 2048 + i);
            vv[u] = __builtin_nontemporal_load((const f32x4*)mv + t * 2048 + 1024 + i);
            ww[u] = __builtin_nontemporal_load((const f32x4*)w + t * 1024 + i);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long long i = threadIdx.x + u * 256;
            const f32x4 ge = ww[u] * 1e-3f;
            mm[u] = 0.9f * mm[u] + 0.1f * ge;
            vv[u] = 0.999f * vv[u] + 0.001f * (ge * ge);
#pragma unroll
            for (int e = 0; e < 4; ++e) ww[u][e] -= 0.01f * (mm[u][e] * __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(vv[u][e]) + 1e-8f));
            __builtin_nontemporal_store(mm[u], (f32x4*)mv + t * 2048 + i);
            __builtin_nontemporal_store(vv[u], (f32x4*)mv + t * 2048 + 1024 + i);
            __builtin_nontemporal_store(ww[u], (f32x4*)w + t * 1024 + i);
        }
    }
}

template <bool INTER>
__global__ __launch_bounds__(256) void adam_tile(float* __restrict__ w, float* __restrict__ m, float* __restrict__ v, long long n_tiles) {
    // tile = 4096 floats per array = 1024 float4; thread handles 4 float4 per array
    for (long long t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        f32x4 ww[4], mm[4], vv[4];
        long long ow[4], om[4], ov[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long long i = threadIdx.x + u * 256;
            if (INTER) { ow[u] = t * 3072 + i; om[u] = t * 3072 + 1024 + i; ov[u] = t * 3072 + 2048 + i; }
            else { ow[u] = om[u] = ov[u] = t * 1024 + i; }
            mm[u] = __builtin_nontemporal_load((const f32x4*)(INTER ? w : m) + om[u]);
            vv[u] = __builtin_nontemporal_load((const f32x4*)(INTER ? w : v) + ov[u]);
            ww[u] = __builtin_nontemporal_load((const f32x4*)w + ow[u]);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const f32x4 ge = ww[u] * 1e-3f;
            mm[u] = 0.9f * mm[u] + 0.1f * ge;
            vv[u] = 0.999f * vv[u] + 0.001f * (ge * ge);
#pragma unroll
            for (int e = 0; e < 4; ++e) ww[u][e] -= 0.01f * (mm[u][e] * __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(vv[u][e]) + 1e-8f));
            __builtin_nontemporal_store(mm[u], (f32x4*)(INTER ? w : m) + om[u]);
            __builtin_nontemporal_store(vv[u], (f32x4*)(INTER ? w : v) + ov[u]);
            __builtin_nontemporal_store(ww[u], (f32x4*)w + ow[u]);
        }
    }
}

template <typename F>
float time_ms(F f, int iters = 5) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    f(); hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < iters; ++i) f();
    hipEventRecord(b); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    return ms / iters;
}

int main() {
    const long long n_tiles = 128LL * 3673088 / 4096;          // one slab at E = 128
    const long long n = n_tiles * 4096;
    float *w, *m, *v, *a;
    hipMalloc(&w, n * 4); hipMalloc(&m, n * 4); hipMalloc(&v, n * 4); hipMalloc(&a, 3 * n * 4);
    hipMemset(w, 0, n * 4); hipMemset(m, 0, n * 4); hipMemset(v, 0, n * 4); hipMemset(a, 0, 3 * n * 4);
    const double b6 = 6.0 * n * 4;
    for (int rep = 0; rep < 3; ++rep) {
        float t0 = time_ms([&] { hipLaunchKernelGGL((adam_tile<false>), dim3(2048), dim3(256), 0, 0, w, m, v, n_tiles); });
        float t1 = time_ms([&] { hipLaunchKernelGGL((adam_tile<true>), dim3(2048), dim3(256), 0, 0, a, a, a, n_tiles); });
        float t2 = time_ms([&] { hipLaunchKernelGGL(adam_tile_mv, dim3(2048), dim3(256), 0, 0, w, a, n_tiles); });
        printf("three arrays %.2f TB/s | one array of [tile][w|m|v] blocks %.2f TB/s | w + [tile][m|v] %.2f TB/s\n", b6 / t0 / 1e9, b6 / t1 / 1e9, b6 / t2 / 1e9);
    }
    return 0;
}
