// Shared helpers for the gfx950 kernels of libmft_hip.so.
#pragma once
#include <hip/hip_runtime.h>
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

// hipFuncSetAttribute applies to the CURRENT device only: a "done" flag per device (an engine may be built on cuda:1 after
// another one ran on cuda:0 in the same process).  need() is true the first time it is asked on a device.
struct MftPerDeviceOnce {
    bool done[64] = {};
    bool need() {
        int d = 0;
        if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= 64) return true;
        if (done[d]) return false;
        done[d] = true;
        return true;
    }
};

static inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }
