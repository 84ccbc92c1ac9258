// Micro-benchmark: a WALKING 3R+3W Adam-shaped stream -- one workgroup per (episode, 32 output-channel rows) walks its 36 K tiles of
// 128 floats, two tiles of w/m/v in flight (two register sets, as csrc/wgrad_fwd.hip) -- over
//   layout 0: row-major [E*512][4608]  (a visit = 32 separate 512-byte runs per array, 18 KB apart)
//   layout 1: tile-major [E*16][36][32][128] (a visit = one contiguous 16 KB block per array; a workgroup's walk is one linear stream)
// with LDS padding / register padding as knobs for the occupancy (workgroups per CU).  Is the walking form's deficit against the
// one-tile-per-workgroup stream (wgrad_adam_rows_kernel) the DRAM access pattern?
//   hipcc --offload-arch=gfx950 -O3 -o walk_layout.bin walk_layout.hip && ./walk_layout.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int LAYOUT, int DEPTH>
__global__ __launch_bounds__(256) void walk(float* __restrict__ w, float* __restrict__ m, float* __restrict__ v, int K, int n_kt, int spin) {
    extern __shared__ float pad[];
    const int tid = threadIdx.x, q = tid & 31, rr = tid >> 5;
    const long long rb = (long long)blockIdx.y * 16 + blockIdx.x;          // row block (32 rows)
    const long long base = rb * 32 * K;
    auto off = [&](int kt, int u) -> long long {
        if (LAYOUT == 0) return base + (long long)(rr + 8 * u) * K + kt * 128 + 4 * q;
        return base + (long long)kt * 4096 + (rr + 8 * u) * 128 + 4 * q;
    };
    f32x4 M[DEPTH][4], V[DEPTH][4], W[DEPTH][4];
    auto ld = [&](int kt, int s) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            M[s][u] = __builtin_nontemporal_load((const f32x4*)(m + off(kt, u)));
            V[s][u] = __builtin_nontemporal_load((const f32x4*)(v + off(kt, u)));
            W[s][u] = __builtin_nontemporal_load((const f32x4*)(w + off(kt, u)));
        }
    };
#pragma unroll
    for (int s = 0; s < DEPTH; ++s) if (s < n_kt) ld(s, s);
    for (int kt0 = 0; kt0 < n_kt; kt0 += DEPTH) {
#pragma unroll
        for (int s = 0; s < DEPTH; ++s) {
            const int kt = kt0 + s;
            if (kt >= n_kt) break;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float ge = W[s][u][e] * 1e-3f;
                    M[s][u][e] = 0.9f * M[s][u][e] + 0.1f * ge;
                    V[s][u][e] = 0.999f * V[s][u][e] + 0.001f * ge * ge;
                    W[s][u][e] -= 0.01f * (M[s][u][e] * __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(V[s][u][e]) + 1e-8f));
                }
                __builtin_nontemporal_store(M[s][u], (f32x4*)(m + off(kt, u)));
                __builtin_nontemporal_store(V[s][u], (f32x4*)(v + off(kt, u)));
                __builtin_nontemporal_store(W[s][u], (f32x4*)(w + off(kt, u)));
            }
            // stand-in for the tile's matrix work: `spin` dependent FMAs per lane (~4 cycles each)
            // spin > 0: FMAs (draws power like real arithmetic); spin < 0: s_sleep (a pure delay of ~64 |spin| cycles, no power)
            float x = pad[tid & 7];
            for (int i = 0; i < spin; ++i) x = __builtin_fmaf(x, 1.0001f, 0.5f);
            for (int i = 0; i < -spin; ++i) __builtin_amdgcn_s_sleep(1);
            if (x == 123.456f) pad[0] = x;
            if (kt + DEPTH < n_kt) ld(kt + DEPTH, s);
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

int main() {
    const int E = 128, K = 4608, n_kt = 36;
    const long long n = (long long)E * 512 * K;
    float *w, *m, *v;
    hipMalloc(&w, n * 4); hipMalloc(&m, n * 4); hipMalloc(&v, n * 4);
    hipMemset(w, 0, n * 4); hipMemset(m, 0, n * 4); hipMemset(v, 0, n * 4);
    const double bytes = 24.0 * n;
    const int lds_opts[3] = {80 * 1024, 53 * 1024, 39 * 1024};          // 2 / 3 / 4 workgroups per CU
    for (int li = 0; li < 3; ++li) {
        const int lds = lds_opts[li];
        hipFuncSetAttribute((const void*)walk<0, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        hipFuncSetAttribute((const void*)walk<1, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        hipFuncSetAttribute((const void*)walk<0, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        hipFuncSetAttribute((const void*)walk<1, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        hipFuncSetAttribute((const void*)walk<0, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        hipFuncSetAttribute((const void*)walk<1, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        const int spins[5] = {0, 600, 1200, -75, -150};
        for (int si = 0; si < 5; ++si) {
            const int spin = spins[si];
            float t;
            printf("wg/CU~%d spin %4d:", 160 * 1024 / lds, spin);
            t = time_ms([&] { hipLaunchKernelGGL((walk<0, 1>), dim3(16, E), dim3(256), lds, 0, w, m, v, K, n_kt, spin); });
            printf("  row-major d1 %.2f", bytes / t / 1e9);
            t = time_ms([&] { hipLaunchKernelGGL((walk<1, 1>), dim3(16, E), dim3(256), lds, 0, w, m, v, K, n_kt, spin); });
            printf("  tile-major d1 %.2f", bytes / t / 1e9);
            t = time_ms([&] { hipLaunchKernelGGL((walk<0, 2>), dim3(16, E), dim3(256), lds, 0, w, m, v, K, n_kt, spin); });
            printf("  | row-major d2 %.2f", bytes / t / 1e9);
            t = time_ms([&] { hipLaunchKernelGGL((walk<1, 2>), dim3(16, E), dim3(256), lds, 0, w, m, v, K, n_kt, spin); });
            printf("  tile-major d2 %.2f", bytes / t / 1e9);
            t = time_ms([&] { hipLaunchKernelGGL((walk<0, 3>), dim3(16, E), dim3(256), lds, 0, w, m, v, K, n_kt, spin); });
            printf("  | row-major d3 %.2f", bytes / t / 1e9);
            t = time_ms([&] { hipLaunchKernelGGL((walk<1, 3>), dim3(16, E), dim3(256), lds, 0, w, m, v, K, n_kt, spin); });
            printf("  tile-major d3 %.2f TB/s\n", bytes / t / 1e9);
        }
    }
    return 0;
}
