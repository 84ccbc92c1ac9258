// Micro-benchmark: what read+write streaming rate can an Adam-shaped kernel (read w,m,v [+g]; write w,m,v) reach
// on MI355X, by loads in flight per thread, temporal hint and grid shape?   hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int U, bool NT, bool WITHG>
__global__ __launch_bounds__(256) void adam_like(float* __restrict__ w, float* __restrict__ m, float* __restrict__ v,
                                                 const float* __restrict__ g, long long n4) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i0 = (long long)blockIdx.x * blockDim.x + threadIdx.x; i0 < n4; i0 += stride * U) {
        f32x4 ww[U], mm[U], vv[U], gg[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long long i = i0 + u * stride;
            if (i < n4) {
                if (NT) {
                    ww[u] = __builtin_nontemporal_load((const f32x4*)w + i);
                    mm[u] = __builtin_nontemporal_load((const f32x4*)m + i);
                    vv[u] = __builtin_nontemporal_load((const f32x4*)v + i);
                    if (WITHG) gg[u] = __builtin_nontemporal_load((const f32x4*)g + i);
                } else {
                    ww[u] = ((const f32x4*)w)[i];
                    mm[u] = ((const f32x4*)m)[i];
                    vv[u] = ((const f32x4*)v)[i];
                    if (WITHG) gg[u] = ((const f32x4*)g)[i];
                }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long long i = i0 + u * stride;
            if (i < n4) {
                f32x4 ge = WITHG ? gg[u] : ww[u] * 1e-3f;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    mm[u][e] = 0.9f * mm[u][e] + 0.1f * ge[e];
                    vv[u][e] = 0.999f * vv[u][e] + 0.001f * ge[e] * ge[e];
                    ww[u][e] -= 0.01f * (mm[u][e] / (sqrtf(vv[u][e]) * 1.0f + 1e-8f));
                }
                if (NT) {
                    __builtin_nontemporal_store(mm[u], (f32x4*)m + i);
                    __builtin_nontemporal_store(vv[u], (f32x4*)v + i);
                    __builtin_nontemporal_store(ww[u], (f32x4*)w + i);
                } else {
                    ((f32x4*)m)[i] = mm[u];
                    ((f32x4*)v)[i] = vv[u];
                    ((f32x4*)w)[i] = ww[u];
                }
            }
        }
    }
}

// contiguous-chunk variant: each workgroup owns a contiguous 16 KB x U slab of each array (tile-like locality)
template <int U, bool NT>
__global__ __launch_bounds__(256) void adam_chunk(float* __restrict__ w, float* __restrict__ m, float* __restrict__ v,
                                                  long long n4) {
    const long long chunk4 = 256LL * U;        // float4 per workgroup
    for (long long c = blockIdx.x; c * chunk4 < n4; c += gridDim.x) {
        f32x4 ww[U], mm[U], vv[U];
        const long long base = c * chunk4 + threadIdx.x;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long long i = base + u * 256;
            if (i < n4) {
                if (NT) {
                    ww[u] = __builtin_nontemporal_load((const f32x4*)w + i);
                    mm[u] = __builtin_nontemporal_load((const f32x4*)m + i);
                    vv[u] = __builtin_nontemporal_load((const f32x4*)v + i);
                } else { ww[u] = ((const f32x4*)w)[i]; mm[u] = ((const f32x4*)m)[i]; vv[u] = ((const f32x4*)v)[i]; }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long long i = base + u * 256;
            if (i < n4) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float ge = ww[u][e] * 1e-3f;
                    mm[u][e] = 0.9f * mm[u][e] + 0.1f * ge;
                    vv[u][e] = 0.999f * vv[u][e] + 0.001f * ge * ge;
                    ww[u][e] -= 0.01f * (mm[u][e] / (sqrtf(vv[u][e]) + 1e-8f));
                }
                if (NT) {
                    __builtin_nontemporal_store(mm[u], (f32x4*)m + i);
                    __builtin_nontemporal_store(vv[u], (f32x4*)v + i);
                    __builtin_nontemporal_store(ww[u], (f32x4*)w + i);
                } else { ((f32x4*)m)[i] = mm[u]; ((f32x4*)v)[i] = vv[u]; ((f32x4*)w)[i] = ww[u]; }
            }
        }
    }
}

__global__ void read_only(const float* __restrict__ w, float* out, long long n4) {
    f32x4 s = {0, 0, 0, 0};
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x)
        s += ((const f32x4*)w)[i];
    if (s[0] + s[1] + s[2] + s[3] == 12345.678f) out[0] = s[0];
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

int main() {
    const long long n = 128LL * 3673088;            // one slab at E = 128 (1.88 GB)
    const long long n4 = n / 4;
    float *w, *m, *v, *g;
    hipMalloc(&w, n * 4); hipMalloc(&m, n * 4); hipMalloc(&v, n * 4); hipMalloc(&g, n * 4);
    hipMemset(w, 0, n * 4); hipMemset(m, 0, n * 4); hipMemset(v, 0, n * 4); hipMemset(g, 0, n * 4);
    const double b6 = 6.0 * n * 4, b7 = 7.0 * n * 4;
    for (int grid : {2048, 8192, 32768, 131072}) {
        float t;
        t = time_ms([&] { hipLaunchKernelGGL((adam_like<1, false, false>), dim3(grid), dim3(256), 0, 0, w, m, v, g, n4); });
        printf("grid %6d U=1 plain   3R+3W: %.0f us %.2f TB/s\n", grid, t * 1e3, b6 / t / 1e9);
        t = time_ms([&] { hipLaunchKernelGGL((adam_like<4, false, false>), dim3(grid), dim3(256), 0, 0, w, m, v, g, n4); });
        printf("grid %6d U=4 plain   3R+3W: %.0f us %.2f TB/s\n", grid, t * 1e3, b6 / t / 1e9);
        t = time_ms([&] { hipLaunchKernelGGL((adam_like<4, true, false>), dim3(grid), dim3(256), 0, 0, w, m, v, g, n4); });
        printf("grid %6d U=4 nontemp 3R+3W: %.0f us %.2f TB/s\n", grid, t * 1e3, b6 / t / 1e9);
        t = time_ms([&] { hipLaunchKernelGGL((adam_like<8, true, false>), dim3(grid), dim3(256), 0, 0, w, m, v, g, n4); });
        printf("grid %6d U=8 nontemp 3R+3W: %.0f us %.2f TB/s\n", grid, t * 1e3, b6 / t / 1e9);
        t = time_ms([&] { hipLaunchKernelGGL((adam_like<4, true, true>), dim3(grid), dim3(256), 0, 0, w, m, v, g, n4); });
        printf("grid %6d U=4 nontemp 4R+3W: %.0f us %.2f TB/s\n", grid, t * 1e3, b7 / t / 1e9);
        t = time_ms([&] { hipLaunchKernelGGL((adam_chunk<4, false>), dim3(grid), dim3(256), 0, 0, w, m, v, n4); });
        printf("grid %6d chunk16K plain  3R+3W: %.0f us %.2f TB/s\n", grid, t * 1e3, b6 / t / 1e9);
        t = time_ms([&] { hipLaunchKernelGGL((adam_chunk<4, true>), dim3(grid), dim3(256), 0, 0, w, m, v, n4); });
        printf("grid %6d chunk16K nontemp 3R+3W: %.0f us %.2f TB/s\n", grid, t * 1e3, b6 / t / 1e9);
        t = time_ms([&] { hipLaunchKernelGGL((adam_chunk<8, true>), dim3(grid), dim3(256), 0, 0, w, m, v, n4); });
        printf("grid %6d chunk32K nontemp 3R+3W: %.0f us %.2f TB/s\n", grid, t * 1e3, b6 / t / 1e9);
    }
    float t = time_ms([&] { hipLaunchKernelGGL(read_only, dim3(8192), dim3(256), 0, 0, w, g, n4); });
    printf("read-only stream: %.0f us %.2f TB/s\n", t * 1e3, n * 4.0 / t / 1e9);
    t = time_ms([&] { hipMemcpyAsync(m, w, n * 4, hipMemcpyDeviceToDevice, 0); });
    printf("hipMemcpy D2D (1R+1W): %.0f us %.2f TB/s\n", t * 1e3, 2.0 * n * 4 / t / 1e9);
    return 0;
}
