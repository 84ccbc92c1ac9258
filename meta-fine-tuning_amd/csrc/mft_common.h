// Shared helpers for the gfx950 kernels of libmft_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <mutex>
#include <stdint.h>
#include "mft_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define MFT_WAVE 64

static inline int mft_launch_status() {
    hipError_t e = hipGetLastError();
    return (int)e;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}

// fp32 -> three bf16 pieces (round-to-nearest-even, exact residuals): x = p1 + p2 + p3 to 24 bits.  Same arithmetic as
// split4 in csrc/conv_x3.hip / sk_split4 in csrc/skinny.hip; used by the producers of pre-split activation planes.
typedef unsigned mft_u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned mft_pk_bf16(float a, float b) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    typedef __bf16 h2 __attribute__((ext_vector_type(2)));
    f2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, h2));
}
__device__ __forceinline__ void mft_split4_bf16(const f32x4 x, mft_u32x2& p1, mft_u32x2& p2, mft_u32x2& p3) {
    f32x4 r;
    p1[0] = mft_pk_bf16(x[0], x[1]);
    p1[1] = mft_pk_bf16(x[2], x[3]);
    r[0] = x[0] - __builtin_bit_cast(float, p1[0] << 16);
    r[1] = x[1] - __builtin_bit_cast(float, p1[0] & 0xffff0000u);
    r[2] = x[2] - __builtin_bit_cast(float, p1[1] << 16);
    r[3] = x[3] - __builtin_bit_cast(float, p1[1] & 0xffff0000u);
    p2[0] = mft_pk_bf16(r[0], r[1]);
    p2[1] = mft_pk_bf16(r[2], r[3]);
    r[0] -= __builtin_bit_cast(float, p2[0] << 16);
    r[1] -= __builtin_bit_cast(float, p2[0] & 0xffff0000u);
    r[2] -= __builtin_bit_cast(float, p2[1] << 16);
    r[3] -= __builtin_bit_cast(float, p2[1] & 0xffff0000u);
    p3[0] = mft_pk_bf16(r[0], r[1]);
    p3[1] = mft_pk_bf16(r[2], r[3]);
}

// torch.optim.Adam on four consecutive parameters (finetune.py:255,299), shared by every kernel that fuses Adam into a
// weight-gradient epilogue so that they agree BIT FOR BIT: the fused multiply-adds are written out (left to the compiler's
// contraction pass, `b1*m + c1*g` becomes fma(b1, m, c1*g) in one kernel and fma(c1, g, b1*m) in another).
//   m = fma(b1, m, (1-b1) g);  v = fma(b2, v, (1-b2) g^2);  w = fma(-step_size, m / (sqrt(v) * inv_sqrt_bc2 + eps), w)
// FAST: v_sqrt_f32 / v_rcp_f32 (1 ulp each) and packed fp32 arithmetic; exact: correctly rounded sqrtf and division.
__device__ __forceinline__ void mft_adam4_fast(f32x4& m, f32x4& v, f32x4& w, const f32x4 g, float b1, float b2, float eps,
                                               float step_size, float inv_sqrt_bc2) {
    const float c1 = 1.f - b1, c2 = 1.f - b2;
    const f32x4 b1v = {b1, b1, b1, b1}, b2v = {b2, b2, b2, b2}, ns = {-step_size, -step_size, -step_size, -step_size};
    m = __builtin_elementwise_fma(b1v, m, c1 * g);
    v = __builtin_elementwise_fma(b2v, v, c2 * (g * g));
    f32x4 den;
#pragma unroll
    for (int e = 0; e < 4; ++e) den[e] = __builtin_amdgcn_rcpf(__builtin_fmaf(__builtin_amdgcn_sqrtf(v[e]), inv_sqrt_bc2, eps));
    w = __builtin_elementwise_fma(ns, m * den, w);
}
__device__ __forceinline__ void mft_adam4_exact(f32x4& m, f32x4& v, f32x4& w, const f32x4 g, float b1, float b2, float eps,
                                                float step_size, float inv_sqrt_bc2) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        m[e] = __builtin_fmaf(b1, m[e], (1.f - b1) * g[e]);
        v[e] = __builtin_fmaf(b2, v[e], ((1.f - b2) * g[e]) * g[e]);
        w[e] = __builtin_fmaf(-step_size, m[e] / __builtin_fmaf(sqrtf(v[e]), inv_sqrt_bc2, eps), w[e]);
    }
}

// hipFuncSetAttribute applies to the CURRENT device only: a "done" flag per device (an engine may be built on cuda:1 after
// another one ran on cuda:0 in the same process, and per-thread multi-device engines are a supported pattern).
//   if (once.need()) { e = hipFuncSetAttribute(...); if (e != hipSuccess) return e; once.mark(); }
// need() is true until mark() was called on this device: a FAILED attribute call is retried (and reported again) by the next
// launch instead of leaving every later launch to die with an opaque launch error.  The flags are atomics: two threads may both
// set the (idempotent) attribute, neither can skip it before it succeeded.
struct MftPerDeviceOnce {
    std::atomic<unsigned char> done[64];
    MftPerDeviceOnce() { for (auto& d : done) d.store(0, std::memory_order_relaxed); }
    static int device() {
        int d = 0;
        if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= 64) return -1;
        return d;
    }
    bool need() const {
        const int d = device();
        return d < 0 || !done[d].load(std::memory_order_acquire);
    }
    void mark() {
        const int d = device();
        if (d >= 0) done[d].store(1, std::memory_order_release);
    }
};

static inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }
