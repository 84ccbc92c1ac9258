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

extern "C" int mft_stream_destroy(void* stream) { return (int)hipStreamDestroy((hipStream_t)stream); }

extern "C" int mft_probe_placement(unsigned* out, int n_blocks, int spin_cycles, void* stream) {
    hipLaunchKernelGGL(probe_placement_kernel, dim3(n_blocks), dim3(64), 0, (hipStream_t)stream, out, spin_cycles);
    return mft_launch_status();
}
