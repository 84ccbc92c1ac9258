// Boundary layout kernels: NCHW -> NHWC ingest, OIHW <-> packed [Cout][KH][KW][Cin] weights, and the
// flipped/transposed weight image used for dgrad.  The reference keeps nn.Parameter tensors in OIHW
// fp32 (state_dict contract); the kernels consume the packed form.
#include "mft_common.h"

namespace {

__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                           int n_img, int C, int HW) {
    const long long total = (long long)n_img * C * HW;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        // i indexes dst (n, hw, c)
        const int c = (int)(i % C);
        const long long t = i / C;
        const int hw = (int)(t % HW);
        const long long n = t / HW;
        dst[i] = src[(n * C + c) * HW + hw];
    }
}

__global__ __launch_bounds__(256) void pack_oihw_kernel(const float* __restrict__ w, float* __restrict__ pk, int Cout,
                                                        int Cin, int KH, int KW, int k_pad, int unpack) {
    const long long total = (long long)Cout * k_pad;
    float* wo = const_cast<float*>(w);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int k = (int)(i % k_pad);
        const int co = (int)(i / k_pad);
        if (k < KH * KW * Cin) {
            const int ci = k % Cin;
            const int khkw = k / Cin;
            const long long src = ((long long)co * Cin + ci) * KH * KW + khkw;
            if (unpack) wo[src] = pk[i];
            else pk[i] = w[src];
        } else if (!unpack) {
            pk[i] = 0.f;
        }
    }
}

// wt[ci][(KH-1-kh)*KW + (KW-1-kw)][co] = w[co][kh*KW+kw][ci]
__global__ __launch_bounds__(256) void pack_dgrad_kernel(const float* __restrict__ w, float* __restrict__ wt, int Cout,
                                                         int Cin, int KH, int KW, long long ws, long long wts) {
    const int g = blockIdx.y;
    const float* wg = w + g * ws;
    float* wtg = wt + g * wts;
    const int taps = KH * KW;
    const long long total = (long long)Cout * taps * Cin;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        // i indexes wt: (ci, tap', co) with co fastest -> coalesced writes; reads strided (L2 absorbs)
        const int co = (int)(i % Cout);
        long long t = i / Cout;
        const int tp = (int)(t % taps);
        const int ci = (int)(t / taps);
        const int tap = taps - 1 - tp;
        wtg[i] = wg[((long long)co * taps + tap) * Cin + ci];
    }
}

inline int lgrid(long long total) {
    long long b = (total + 255) / 256;
    if (b < 1) b = 1;
    if (b > 4096) b = 4096;
    return (int)b;
}

}  // namespace

extern "C" int mft_nchw_to_nhwc(const float* src, float* dst, int n_img, int C, int H, int W, void* stream) {
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(lgrid((long long)n_img * C * H * W)), dim3(256), 0,
                       (hipStream_t)stream, src, dst, n_img, C, H * W);
    return mft_launch_status();
}

extern "C" int mft_pack_oihw(const float* w_oihw, float* w_pk, int Cout, int Cin, int KH, int KW, int k_pad,
                             void* stream) {
    if (k_pad < KH * KW * Cin) return MFT_EINVAL;
    hipLaunchKernelGGL(pack_oihw_kernel, dim3(lgrid((long long)Cout * k_pad)), dim3(256), 0, (hipStream_t)stream,
                       w_oihw, w_pk, Cout, Cin, KH, KW, k_pad, 0);
    return mft_launch_status();
}

extern "C" int mft_unpack_oihw(const float* w_pk, float* w_oihw, int Cout, int Cin, int KH, int KW, int k_pad,
                               void* stream) {
    if (k_pad < KH * KW * Cin) return MFT_EINVAL;
    hipLaunchKernelGGL(pack_oihw_kernel, dim3(lgrid((long long)Cout * k_pad)), dim3(256), 0, (hipStream_t)stream,
                       (const float*)w_oihw, const_cast<float*>(w_pk), Cout, Cin, KH, KW, k_pad, 1);
    return mft_launch_status();
}

extern "C" int mft_pack_dgrad(const float* w_pk, float* wt_pk, int Cout, int Cin, int KH, int KW, int groups,
                              long long w_stride, long long wt_stride, void* stream) {
    if ((KH * KW * Cin) % 32 != 0 || (KH * KW * Cout) % 32 != 0) return MFT_EINVAL;
    dim3 grid(lgrid((long long)Cout * KH * KW * Cin), groups, 1);
    hipLaunchKernelGGL(pack_dgrad_kernel, grid, dim3(256), 0, (hipStream_t)stream, w_pk, wt_pk, Cout, Cin, KH, KW,
                       w_stride, wt_stride);
    return mft_launch_status();
}

extern "C" int mft_version(void) { return 100; }

extern "C" int mft_device_info(int* cu_count, int* is_gfx950) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, dev);
    if (e != hipSuccess) return (int)e;
    if (cu_count) *cu_count = prop.multiProcessorCount;
    if (is_gfx950) {
        const char* a = prop.gcnArchName;
        *is_gfx950 = (a[0] == 'g' && a[1] == 'f' && a[2] == 'x' && a[3] == '9' && a[4] == '5' && a[5] == '0') ? 1 : 0;
    }
    return 0;
}
