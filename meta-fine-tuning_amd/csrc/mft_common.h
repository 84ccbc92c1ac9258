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

static inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }
