// Device-side view generation for test-time fine-tuning.
//
// The reference builds every episode's 2 + gen_examples views on the host with PIL, image by image
// (datasets/EuroSAT_few_shot.py:145-170 TransformLoader2, :240-276 SubDataset2): views 0 and 1 are
// Scale(1.15*size) -> CenterCrop(size); the others RandomSizedCrop(size, scale=(0.5,0.9)) -> ImageJitter(Brightness .1,
// Contrast .1, Color .05; data/additional_transforms.py) -> RandomHorizontalFlip -> RandomVerticalFlip; all end with
// ToTensor and Normalize(ImageNet mean/std).  At 64 episodes/s per GPU that is 120,000 PIL images per second per GPU.
// Here the uint8 source images stay in HBM and ONE launch writes any number of views straight into the engine's NHWC
// stores.  The random parameters (crop box, jitter factors, flips) are drawn on the host by meta-fine-tuning_amd/augment.py
// (a few bytes per view) so that results do not depend on the launch geometry.
//
// One workgroup per (view, image).  Pass 1: bilinear resample of the crop box (align_corners=False convention, edge
// clamp inside the box), brightness blend, write un-normalised, accumulate the grey mean ImageEnhance.Contrast needs.
// Pass 2 (same threads, same pixels): contrast blend around the grey mean, colour blend around the pixel's grey value,
// normalise.  PIL's uint8 arithmetic is followed: values are truncated to integers after every enhancement
// (ImagingBlend), the grey conversion is PIL's fixed-point L = (19595 R + 38470 G + 7471 B + 32768) >> 16.
#include "mft_common.h"

namespace {

struct AugArgs {
    const unsigned char* src;      // [n_img][Hs][Ws][3] uint8 (PIL RGB order)
    const float* params;           // [n_views][n_img][10]: y0, x0, h, w (crop box in source pixels), rb, rc, rcol, flip_h, flip_v, enhance
    float* out;
    long long view_stride, img_stride;   // floats
    int n_img, Hs, Ws, size;
    float mean[3], inv_std[3];
};

__device__ __forceinline__ float trunc_u8(float v) {      // ImagingBlend: clip to [0,255], truncate
    v = fminf(fmaxf(v, 0.f), 255.f);
    return floorf(v);
}

__device__ __forceinline__ float grey_l(float r, float g, float b) {
    const unsigned v = 19595u * (unsigned)r + 38470u * (unsigned)g + 7471u * (unsigned)b + 0x8000u;
    return (float)(v >> 16);
}

__global__ __launch_bounds__(256) void augment_views_kernel(AugArgs p) {
    const int img = blockIdx.x, view = blockIdx.y;
    const float* pr = p.params + ((long long)view * p.n_img + img) * 10;
    const float y0 = pr[0], x0 = pr[1], ch = pr[2], cw = pr[3];
    const float rb = pr[4], rc = pr[5], rcol = pr[6];
    const bool fh = pr[7] != 0.f, fv = pr[8] != 0.f, enhance = pr[9] != 0.f;
    const unsigned char* s = p.src + (long long)img * p.Hs * p.Ws * 3;
    float* o = p.out + (long long)view * p.view_stride + (long long)img * p.img_stride;
    const int S = p.size;
    const float sy = ch / (float)S, sx = cw / (float)S;
    // augmented views crop first and resample the crop (edge clamp inside the box); the un-augmented views resample the
    // WHOLE image (Scale) and crop afterwards, so their interpolation may reach across the box edge
    const int iy0 = enhance ? (int)y0 : 0, ix0 = enhance ? (int)x0 : 0;
    const int iy1 = enhance ? (int)y0 + (int)ch - 1 : p.Hs - 1, ix1 = enhance ? (int)x0 + (int)cw - 1 : p.Ws - 1;
    float gsum = 0.f;
    for (int q = threadIdx.x; q < S * S; q += 256) {
        const int y = q / S, x = q - y * S;
        // output pixel (y,x) shows resampled pixel (yy,xx) of the un-flipped view
        const int yy = fv ? S - 1 - y : y, xx = fh ? S - 1 - x : x;
        float fy = y0 + ((float)yy + 0.5f) * sy - 0.5f;
        float fx = x0 + ((float)xx + 0.5f) * sx - 0.5f;
        fy = fminf(fmaxf(fy, (float)iy0), (float)iy1);
        fx = fminf(fmaxf(fx, (float)ix0), (float)ix1);
        const int ya = (int)fy, xa = (int)fx;
        const int yb = min(ya + 1, iy1), xb = min(xa + 1, ix1);
        const float wy = fy - (float)ya, wx = fx - (float)xa;
        float v[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float a = s[((long long)ya * p.Ws + xa) * 3 + c], b = s[((long long)ya * p.Ws + xb) * 3 + c];
            const float cc = s[((long long)yb * p.Ws + xa) * 3 + c], d = s[((long long)yb * p.Ws + xb) * 3 + c];
            const float top = a + (b - a) * wx, bot = cc + (d - cc) * wx;
            v[c] = floorf(top + (bot - top) * wy + 0.5f);                      // resampled image is uint8 in PIL
            if (enhance) v[c] = trunc_u8(v[c] * rb);                           // Brightness: blend with black
        }
        if (enhance) gsum += grey_l(v[0], v[1], v[2]);
#pragma unroll
        for (int c = 0; c < 3; ++c) o[(long long)q * 3 + c] = v[c];
    }
    __shared__ float red[4];
    __shared__ float s_mean;
    if (enhance) {
        gsum = wave_sum(gsum);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = gsum;
        __syncthreads();
        if (threadIdx.x == 0) s_mean = floorf((red[0] + red[1] + red[2] + red[3]) / (float)(S * S) + 0.5f);   // int(mean + 0.5)
        __syncthreads();
    }
    const float gm = enhance ? s_mean : 0.f;
    for (int q = threadIdx.x; q < S * S; q += 256) {
        float v[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) v[c] = o[(long long)q * 3 + c];
        if (enhance) {
#pragma unroll
            for (int c = 0; c < 3; ++c) v[c] = trunc_u8(gm + rc * (v[c] - gm));                  // Contrast
            const float gl = grey_l(v[0], v[1], v[2]);
#pragma unroll
            for (int c = 0; c < 3; ++c) v[c] = trunc_u8(gl + rcol * (v[c] - gl));                // Color
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) o[(long long)q * 3 + c] = (v[c] * (1.f / 255.f) - p.mean[c]) * p.inv_std[c];
    }
}

}  // namespace

extern "C" int mft_augment_views(const unsigned char* src, int n_img, int Hs, int Ws, const float* params, int n_views,
                                 float* out, long long view_stride, long long img_stride, int size, const float* mean3,
                                 const float* std3, void* stream) {
    if (n_img <= 0 || n_views <= 0 || size <= 0 || Hs <= 0 || Ws <= 0) return MFT_EINVAL;
    AugArgs p;
    p.src = src; p.params = params; p.out = out; p.view_stride = view_stride; p.img_stride = img_stride;
    p.n_img = n_img; p.Hs = Hs; p.Ws = Ws; p.size = size;
    for (int c = 0; c < 3; ++c) { p.mean[c] = mean3[c]; p.inv_std[c] = 1.f / std3[c]; }
    hipLaunchKernelGGL(augment_views_kernel, dim3(n_img, n_views), dim3(256), 0, (hipStream_t)stream, p);
    return mft_launch_status();
}
