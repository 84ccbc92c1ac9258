// Micro-benchmark: how many CUs does an Adam-shaped 3R+3W stream need to reach the HBM rate?  Runs the nontemporal chunked
// kernel of adam_stream.hip on CU-masked streams (hipExtStreamCreateWithCUMask, CUs spread over all 8 XCDs) with 1..3 chunk
// depths.  If half the chip could saturate HBM, the other half could run the frozen trunk beside it without slowing it.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int U>
__global__ __launch_bounds__(256) void adam_chunk(float* __restrict__ w, float* __restrict__ m, float* __restrict__ v, long long n4) {
    const long long chunk4 = 256LL * U;
    for (long long c = blockIdx.x; c * chunk4 < n4; c += gridDim.x) {
        f32x4 ww[U], mm[U], vv[U];
        const long long base = c * chunk4 + threadIdx.x;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long long i = base + u * 256;
            ww[u] = __builtin_nontemporal_load((const f32x4*)w + i);
            mm[u] = __builtin_nontemporal_load((const f32x4*)m + i);
            vv[u] = __builtin_nontemporal_load((const f32x4*)v + i);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long long i = base + u * 256;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float ge = ww[u][e] * 1e-3f;
                mm[u][e] = 0.9f * mm[u][e] + 0.1f * ge;
                vv[u][e] = 0.999f * vv[u][e] + 0.001f * ge * ge;
                ww[u][e] -= 0.01f * (mm[u][e] / (sqrtf(vv[u][e]) + 1e-8f));
            }
            __builtin_nontemporal_store(mm[u], (f32x4*)m + i);
            __builtin_nontemporal_store(vv[u], (f32x4*)v + i);
            __builtin_nontemporal_store(ww[u], (f32x4*)w + i);
        }
    }
}

template <typename F>
float time_ms(F f, hipStream_t s, int iters = 5) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    f();
    hipStreamSynchronize(s);
    hipEventRecord(a, s);
    for (int i = 0; i < iters; ++i) f();
    hipEventRecord(b, s);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    return ms / iters;
}

int main() {
    const long long n = 128LL * 3673088;
    const long long n4 = n / 4;
    float *w, *m, *v;
    hipMalloc(&w, n * 4); hipMalloc(&m, n * 4); hipMalloc(&v, n * 4);
    hipMemset(w, 0, n * 4); hipMemset(m, 0, n * 4); hipMemset(v, 0, n * 4);
    const double b6 = 6.0 * n * 4;
    for (int per_xcd : {8, 12, 16, 20, 24, 28, 32}) {          // CUs per XCD (of 32) -> 64 .. 256 CUs
        uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        // mask bit i: bits are dealt round-robin to the XCDs (tools/cumask_probe.py): bit b -> XCD b % 8, CU b / 8
        for (int c = 0; c < per_xcd; ++c)
            for (int x = 0; x < 8; ++x) { const int b = c * 8 + x; mask[b / 32] |= 1u << (b % 32); }
        hipStream_t s;
        if (hipExtStreamCreateWithCUMask(&s, 8, mask) != hipSuccess) { printf("mask failed\n"); return 1; }
        for (int wgs_per_cu : {4, 8}) {
            const int grid = per_xcd * 8 * wgs_per_cu;
            float t4 = time_ms([&] { hipLaunchKernelGGL((adam_chunk<4>), dim3(grid), dim3(256), 0, s, w, m, v, n4); }, s);
            float t8 = time_ms([&] { hipLaunchKernelGGL((adam_chunk<8>), dim3(grid), dim3(256), 0, s, w, m, v, n4); }, s);
            float t12 = time_ms([&] { hipLaunchKernelGGL((adam_chunk<12>), dim3(grid), dim3(256), 0, s, w, m, v, n4); }, s);
            printf("%3d CUs, %d persistent workgroups per CU: 48KB/WG in flight %.2f TB/s | 96KB %.2f TB/s | 144KB %.2f TB/s\n",
                   per_xcd * 8, wgs_per_cu, b6 / t4 / 1e9, b6 / t8 / 1e9, b6 / t12 / 1e9);
        }
        hipStreamDestroy(s);
    }
    return 0;
}
