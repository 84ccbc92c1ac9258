// Read-only stream of per-episode weights [E][512][4608] fp32 (trunk.7.C2 x 128 episodes = 1.21 GB) with the access shapes of the
// weight-streaming kernels: (a) MFMA-fragment shaped -- one wave-instruction touches 16 rows x 64 B (lane (m, kq) reads 16 B of row m),
// eight loads in flight, one 1024-thread workgroup per (episode, 256 rows), as csrc/skinny.hip does; (b) the same rows read in full
// 1 KB runs (a wave-instruction = 1024 contiguous bytes of one row); (c) as (b) with 256-thread workgroups (4 per CU).
// hipcc --offload-arch=gfx950 -O3 read_pattern.hip -o read_pattern && ./read_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int K = 4608, CO = 512;

__global__ __launch_bounds__(1024) void frag_kernel(const float* __restrict__ w, float* __restrict__ out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m = lane & 15, kq = lane >> 4;
    const int e = blockIdx.x >> 1, half = blockIdx.x & 1;
    const float* row = w + ((long long)e * CO + half * 256 + wave * 16 + m) * K + 4 * kq;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < K; k0 += 128) {
        f32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = __builtin_nontemporal_load((const f32x4*)(row + k0 + 16 * u));
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u];
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) out[threadIdx.x] = acc[0];
}

// (d) the same 16 rows x 64 B per wave-instruction as (a), but ADJACENT lanes read adjacent 16-B pieces (lane l -> row l / 4, piece l % 4):
//     is (a)'s loss the scattered lanes (each quad of lanes touches four different lines) or the short DRAM runs?
// RPI = rows per instruction: 16 (64-B runs), 8 (128-B), 4 (256-B)
template <int RPI>
__global__ __launch_bounds__(1024) void quad_kernel(const float* __restrict__ w, float* __restrict__ out) {
    constexpr int LPR = 64 / RPI;                                  // lanes per row
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int e = blockIdx.x >> 1, half = blockIdx.x & 1;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int rb = 0; rb < 16; rb += RPI) {
        const float* row = w + ((long long)e * CO + half * 256 + wave * 16 + rb + lane / LPR) * K + 4 * (lane % LPR);
        for (int k0 = 0; k0 < K; k0 += 32 * LPR) {
            f32x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = __builtin_nontemporal_load((const f32x4*)(row + k0 + 4 * LPR * u));
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += v[u];
        }
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) out[threadIdx.x] = acc[0];
}

// (e) the data-gradient kernel's shape: 8 B per lane, lanes m = 0..15 of a quarter-wave read 128 B of one row, four rows (kq) per
//     wave-instruction (full lines, but half the bytes per instruction); 512-thread workgroups, 32 loads in flight per lane pair
typedef float f32x2 __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(512) void dgrad_shape_kernel(const float* __restrict__ w, float* __restrict__ out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;      // 8 waves x 32 input channels = 256 of the 512 columns... rows streamed
    const int m = lane & 15, kq = lane >> 4;
    const int e = blockIdx.x >> 1, half = blockIdx.x & 1;
    // wave owns 32 columns (ci) x all 512 rows (co) of one tap-slice: walk the rows 16 at a time (kq, t)
    f32x2 acc = {0.f, 0.f};
    for (int tap = 0; tap < 9; ++tap) {
        const float* col = w + (long long)e * CO * K + tap * 512 + half * 256 + wave * 32 + 2 * m;
        for (int co0 = 0; co0 < CO; co0 += 32) {
            f32x2 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = __builtin_nontemporal_load((const f32x2*)(col + (long long)(co0 + 4 * u + kq) * K));
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += v[u];
        }
    }
    if (acc[0] + acc[1] == 12345.678f) out[threadIdx.x] = acc[0];
}

template <int NT>
__global__ __launch_bounds__(NT) void line_kernel(const float* __restrict__ w, float* __restrict__ out) {
    constexpr int NW = NT / 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int rows_per_wg = 256 * NW / 16;                         // 1024 threads: 256 rows; 256 threads: 64 rows
    const long long row0 = (long long)blockIdx.x * rows_per_wg + wave * 16;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int r = 0; r < 16; ++r) {
        const float* row = w + (row0 + r) * K + 4 * lane;
        for (int k0 = 0; k0 < K; k0 += 2048) {                     // 8 loads of 1 KB per wave in flight (4608 = 2 x 2048 + 512)
            f32x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int k = k0 + 256 * u;
                v[u] = k + 4 * lane < K ? __builtin_nontemporal_load((const f32x4*)(row + k)) : (f32x4){0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += v[u];
        }
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) out[threadIdx.x] = acc[0];
}

template <typename F>
float time_ms(F f, int reps = 10) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    f(); hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < reps; ++i) f();
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms / reps;
}

int main() {
    const int E = 128;
    const long long n = (long long)E * CO * K;
    float *w, *out;
    hipMalloc(&w, n * 4); hipMalloc(&out, 4096);
    hipMemset(w, 0, n * 4);
    for (int rep = 0; rep < 3; ++rep) {
        const float ta = time_ms([&] { hipLaunchKernelGGL(frag_kernel, dim3(E * 2), dim3(1024), 0, 0, w, out); });
        const float tb = time_ms([&] { hipLaunchKernelGGL(line_kernel<1024>, dim3(E * 2), dim3(1024), 0, 0, w, out); });
        const float tc = time_ms([&] { hipLaunchKernelGGL(line_kernel<256>, dim3(E * 8), dim3(256), 0, 0, w, out); });
        const float td = time_ms([&] { hipLaunchKernelGGL(quad_kernel<16>, dim3(E * 2), dim3(1024), 0, 0, w, out); });
        const float te = time_ms([&] { hipLaunchKernelGGL(quad_kernel<8>, dim3(E * 2), dim3(1024), 0, 0, w, out); });
        const float tf = time_ms([&] { hipLaunchKernelGGL(quad_kernel<4>, dim3(E * 2), dim3(1024), 0, 0, w, out); });
        const float tg = time_ms([&] { hipLaunchKernelGGL(dgrad_shape_kernel, dim3(E * 2), dim3(512), 0, 0, w, out); });
        printf("data-gradient shape (8 B per lane, 4 rows x 128 B per instruction) %.2f TB/s\n", n * 4 / tg / 1e9);
        printf("adjacent lanes, 16 rows x 64 B %.2f | 8 rows x 128 B %.2f | 4 rows x 256 B %.2f TB/s\n", n * 4 / td / 1e9, n * 4 / te / 1e9, n * 4 / tf / 1e9);
        printf("fragment-shaped (16 rows x 64 B per instruction) %.2f TB/s | 1 KB runs, 1024-thread WGs %.2f TB/s | 1 KB runs, 256-thread WGs %.2f TB/s\n",
               n * 4 / ta / 1e9, n * 4 / tb / 1e9, n * 4 / tc / 1e9);
    }
    return 0;
}
