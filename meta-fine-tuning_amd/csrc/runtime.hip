// Stream partitioning for the two halves of an inner step.
//
// A lockstep inner step has an MFMA-bound half (frozen trunk.0-6 of step t+1, shared weights) and an HBM-bound
// half (per-episode trunk.7 forward/backward + Adam of step t: ~112 MB of weight/moment traffic per episode).
// Run on two ordinary streams the halves merely time-slice the CUs (each kernel's occupancy halves).  A CU-masked
// stream pins a queue to a fixed subset of the 256 CUs (spread over all 8 XCDs so every L2 / fabric port stays in
// use), which lets the bandwidth-bound half saturate HBM from a minority of the CUs while the matrix-bound half
// owns the rest.
#include "mft_common.h"
#include <hip/hip_ext.h>

namespace {

__global__ void probe_placement_kernel(unsigned* out, int spin) {
    // one record per workgroup: {XCC_ID, HW_ID}; spin keeps the workgroup alive so that a grid larger than the
    // masked CU set has to spread over every allowed CU
    if (threadIdx.x == 0) {
        const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);   // HW_REG_XCC_ID[3:0]
        const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);    // HW_REG_HW_ID
        out[2 * blockIdx.x] = xcc;
        out[2 * blockIdx.x + 1] = hw;
    }
    long long t0 = clock64();
    while (clock64() - t0 < spin) {}
}

}  // namespace

extern "C" int mft_stream_create_cumask(const unsigned* mask_words, int n_words, void** stream_out) {
    hipStream_t s = nullptr;
    hipError_t e = hipExtStreamCreateWithCUMask(&s, (uint32_t)n_words, mask_words);
    if (e != hipSuccess) return (int)e;
    *stream_out = (void*)s;
    return 0;
}

// HIP exposes three queue priorities on this part (hipDeviceGetStreamPriorityRange: least .. greatest); torch.cuda.Stream
// clamps to two of them.  *range_out = {least, greatest} when non-null.
extern "C" int mft_stream_create_priority(int priority, void** stream_out, int* range_out) {
    int least = 0, greatest = 0;
    hipError_t e = hipDeviceGetStreamPriorityRange(&least, &greatest);
    if (e != hipSuccess) return (int)e;
    if (range_out) { range_out[0] = least; range_out[1] = greatest; }
    if (!stream_out) return 0;
    hipStream_t s = nullptr;
    e = hipStreamCreateWithPriority(&s, hipStreamNonBlocking, priority);
    if (e != hipSuccess) return (int)e;
    *stream_out = (void*)s;
    return 0;
}

// Adam-shaped 3-read / 3-write stream over three scratch arrays (no gradient operand, no matrix work): the rate this lease's
// memory system gives a pure w/m/v stream -- bench.py reports the fused weight-gradient + Adam kernel against it, because the same
// binary measures 5.2-6.3 TB/s on different leases (tools/microbench/adam_cus.hip).
namespace {
typedef float pf32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void stream_probe_kernel(float* __restrict__ w, float* __restrict__ m, float* __restrict__ v, long long n4) {
    constexpr int U = 4;
    const long long chunk4 = 256LL * U;
    for (long long c = blockIdx.x; (c + 1) * chunk4 <= n4; c += gridDim.x) {
        pf32x4 ww[U], mm[U], vv[U];
        const long long base = c * chunk4 + threadIdx.x;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            mm[u] = __builtin_nontemporal_load((const pf32x4*)m + base + u * 256);
            vv[u] = __builtin_nontemporal_load((const pf32x4*)v + base + u * 256);
            ww[u] = __builtin_nontemporal_load((const pf32x4*)w + base + u * 256);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const pf32x4 ge = ww[u] * 1e-3f;
            mm[u] = 0.9f * mm[u] + 0.1f * ge;
            vv[u] = 0.999f * vv[u] + 0.001f * (ge * ge);
#pragma unroll
            for (int e = 0; e < 4; ++e) ww[u][e] -= 0.01f * (mm[u][e] * __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(vv[u][e]) + 1e-8f));
            __builtin_nontemporal_store(mm[u], (pf32x4*)m + base + u * 256);
            __builtin_nontemporal_store(vv[u], (pf32x4*)v + base + u * 256);
            __builtin_nontemporal_store(ww[u], (pf32x4*)w + base + u * 256);
        }
    }
}
}  // namespace

extern "C" int mft_stream_probe(float* w, float* m, float* v, long long n, void* stream) {
    if (n < 4096 || (n & 1023) != 0) return MFT_EINVAL;
    hipLaunchKernelGGL(stream_probe_kernel, dim3(2048), dim3(256), 0, (hipStream_t)stream, w, m, v, n / 4);
    return mft_launch_status();
}

extern "C" int mft_stream_destroy(void* stream) { return (int)hipStreamDestroy((hipStream_t)stream); }

extern "C" int mft_probe_placement(unsigned* out, int n_blocks, int spin_cycles, void* stream) {
    hipLaunchKernelGGL(probe_placement_kernel, dim3(n_blocks), dim3(64), 0, (hipStream_t)stream, out, spin_cycles);
    return mft_launch_status();
}

// Measurement aid (bench.py, tools/): HIP events recorded on the stream a launcher enqueues on, so that per-launch durations can be
// measured live for ANY entry point of this library (meta_fine_tuning_amd._lib.LaunchTimer) -- torch.cuda.Event only sees torch's
// current stream, and the engine runs on raw priority / CU-masked streams of its own.
extern "C" int mft_event_create(void** event_out) {
    hipEvent_t e = nullptr;
    hipError_t rc = hipEventCreate(&e);
    if (rc != hipSuccess) return (int)rc;
    *event_out = (void*)e;
    return 0;
}
extern "C" int mft_event_record(void* event, void* stream) { return (int)hipEventRecord((hipEvent_t)event, (hipStream_t)stream); }
extern "C" int mft_event_elapsed_ms(void* start, void* stop, float* ms_out) {
    hipError_t rc = hipEventSynchronize((hipEvent_t)stop);
    if (rc != hipSuccess) return (int)rc;
    return (int)hipEventElapsedTime(ms_out, (hipEvent_t)start, (hipEvent_t)stop);
}
extern "C" int mft_event_destroy(void* event) { return (int)hipEventDestroy((hipEvent_t)event); }
