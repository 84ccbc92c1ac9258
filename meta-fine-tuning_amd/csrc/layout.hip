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

// Episode ingest (finetune.py:208-233): the support images of every augmentation view -- view 0 twice (x_a_i doubling),
// then views 1.. -- go from the loader's [n_way, per_class, C, H, W] tensors into the NHWC support store in ONE launch, and
// view 0 of all images into the final-pass store.  One thread per pixel: C coalesced reads, one contiguous C-float write.
namespace {
constexpr int MAX_VIEWS = 32;
struct IngestArgs {
    const float* view[MAX_VIEWS];
    float* support;            // [(n_views + dbl) * n_way * n_support][HW][C]
    float* all;                // [n_way * per_class][HW][C] or nullptr
    int n_views, dbl, n_way, per_class, n_support, C, HW;
};

__global__ __launch_bounds__(256) void ingest_views_kernel(IngestArgs p) {
    const long long npv = (long long)p.n_way * p.n_support;
    const long long n_sup = (long long)(p.n_views + p.dbl) * npv * p.HW;
    const long long n_all = p.all ? (long long)p.n_way * p.per_class * p.HW : 0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n_sup + n_all;
         i += (long long)gridDim.x * blockDim.x) {
        const float* src;
        float* dst;
        int hw;
        if (i < n_sup) {
            hw = (int)(i % p.HW);
            const long long img = i / p.HW;                       // slot * npv + way * n_support + s
            const int slot = (int)(img / npv);
            const int r = (int)(img - slot * npv);
            const int way = r / p.n_support, sidx = r - way * p.n_support;
            const int v = slot - p.dbl < 0 ? 0 : slot - p.dbl;
            src = p.view[v] + ((long long)way * p.per_class + sidx) * p.C * p.HW;
            dst = p.support + i * p.C;
        } else {
            const long long j = i - n_sup;
            hw = (int)(j % p.HW);
            src = p.view[0] + (j / p.HW) * p.C * p.HW;
            dst = p.all + j * p.C;
        }
        for (int c = 0; c < p.C; ++c) dst[c] = src[(long long)c * p.HW + hw];
    }
}
}  // namespace

extern "C" int mft_ingest_episode_views(const float* const* views, int n_views, int double_first, int n_way, int per_class,
                                        int n_support, int C, int H, int W, float* support_store, float* all_store,
                                        void* stream) {
    if (n_views < 1 || n_views > MAX_VIEWS || n_support > per_class || !support_store) return MFT_EINVAL;
    IngestArgs p;
    for (int v = 0; v < n_views; ++v) p.view[v] = views[v];
    for (int v = n_views; v < MAX_VIEWS; ++v) p.view[v] = nullptr;
    p.support = support_store; p.all = all_store; p.n_views = n_views; p.dbl = double_first ? 1 : 0;
    p.n_way = n_way; p.per_class = per_class; p.n_support = n_support; p.C = C; p.HW = H * W;
    const long long total = ((long long)(n_views + p.dbl) * n_way * n_support + (all_store ? (long long)n_way * per_class : 0)) * H * W;
    hipLaunchKernelGGL(ingest_views_kernel, dim3(lgrid(total)), dim3(256), 0, (hipStream_t)stream, p);
    return mft_launch_status();
}

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

// All weight repacks of a model in ONE launch (the meta-training step re-packs every convolution / linear weight after each
// optimizer step: 43 tensors for GnnNet).  jobs: n_jobs records {src, dst, Cout, Cin, KH*KW, k_pad, first element} (7 x int64,
// device memory), ordered by first element; element e of the concatenated packed outputs belongs to the last job whose first
// element is <= e.
struct PackJob { const float* src; float* dst; long long Cout, Cin, KHKW, k_pad, start; };

__device__ __forceinline__ bool pack_job_is_3x3_quads(const PackJob& j) {
    return j.KHKW == 9 && (j.Cin & 3) == 0 && j.k_pad == 9 * j.Cin && ((unsigned long long)j.src & 15) == 0 && ((unsigned long long)j.dst & 15) == 0;
}

__device__ __forceinline__ void pack_oihw_element(const PackJob& j, const long long l) {
    const int k = (int)(l % j.k_pad);
    const long long co = l / j.k_pad;
    float v = 0.f;
    if (j.KHKW < 0) {
        // transposed job (a linear / 1x1 layer's data-gradient operand): dst [rows][k_pad] with dst[ci][co] = src[co][ci], zero
        // beyond the real Cin rows / Cout columns -- the forward GEMM kernel then computes dx = dy @ W (csrc/conv_igemm.hip)
        if (k < j.Cout && co < j.Cin) v = j.src[(long long)k * j.Cin + co];
    } else if (k < j.KHKW * j.Cin) {
        const int ci = k % (int)j.Cin, khkw = k / (int)j.Cin;
        v = j.src[(co * j.Cin + ci) * j.KHKW + khkw];
    }
    j.dst[l] = v;
}

// Work is dealt in UNITS: one element of a job's packed output -- or, for a 3x3 layer whose Cin is a multiple of 4 (no padding
// columns: 98 % of ResNet10's weights), one (output channel, four input channels) group = 36 consecutive source floats read as
// nine 16-byte loads and written as nine 16-byte stores, one per tap (the element form gathers every float on its own, 36 bytes
// from its neighbour's).  The unit prefix of the jobs is formed by every workgroup in LDS (<= 64 jobs; more: elements only).
__global__ __launch_bounds__(256) void pack_oihw_multi_kernel(const PackJob* __restrict__ jobs, int n_jobs, long long total) {
    __shared__ long long ustart[65];
    if (n_jobs > 64) {
        for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
            int lo = 0, hi = n_jobs - 1;
            while (lo < hi) {
                const int mid = (lo + hi + 1) >> 1;
                if (jobs[mid].start <= i) lo = mid; else hi = mid - 1;
            }
            const PackJob j = jobs[lo];
            pack_oihw_element(j, i - j.start);
        }
        return;
    }
    if (threadIdx.x < 64) {
        const int t = threadIdx.x;
        long long u = 0;
        if (t < n_jobs) {
            const PackJob j = jobs[t];
            const long long elems = (t + 1 < n_jobs ? jobs[t + 1].start : total) - j.start;
            const bool fast = pack_job_is_3x3_quads(j);
            u = fast ? j.Cout * (j.Cin >> 2) : elems;
        }
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {          // inclusive scan over the wave
            const long long o = __shfl_up(u, off, 64);
            if (t >= off) u += o;
        }
        ustart[t + 1] = u;
        if (t == 0) ustart[0] = 0;
    }
    __syncthreads();
    const long long units = ustart[n_jobs];
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < units; i += (long long)gridDim.x * blockDim.x) {
        int lo = 0, hi = n_jobs - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (ustart[mid] <= i) lo = mid; else hi = mid - 1;
        }
        const PackJob j = jobs[lo];
        const long long l = i - ustart[lo];
        if (pack_job_is_3x3_quads(j)) {
            const int cq = (int)(j.Cin >> 2);
            const long long co = l / cq;
            const int c4 = (int)(l - co * cq) * 4;
            const float* src = j.src + (co * j.Cin + c4) * 9;
            float v[36];
#pragma unroll
            for (int q = 0; q < 9; ++q) {
                const f32x4 x = *(const f32x4*)(src + 4 * q);
                v[4 * q] = x[0]; v[4 * q + 1] = x[1]; v[4 * q + 2] = x[2]; v[4 * q + 3] = x[3];
            }
            float* dst = j.dst + co * j.k_pad + c4;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const f32x4 o = {v[tap], v[9 + tap], v[18 + tap], v[27 + tap]};
                *(f32x4*)(dst + tap * j.Cin) = o;
            }
        } else {
            pack_oihw_element(j, l);
        }
    }
}

extern "C" int mft_pack_oihw_multi(const void* jobs, int n_jobs, long long total_elements, void* stream) {
    if (n_jobs < 1 || total_elements < 1) return MFT_EINVAL;
    hipLaunchKernelGGL(pack_oihw_multi_kernel, dim3(lgrid(total_elements)), dim3(256), 0, (hipStream_t)stream, (const PackJob*)jobs,
                       n_jobs, total_elements);
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
