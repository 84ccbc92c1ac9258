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
// One workgroup per (view, image).  BIT-EXACT with the PIL pipeline the reference runs (uint8 level, and the float32
// ToTensor / Normalize arithmetic):
//   * Image.resize(BILINEAR) is Pillow's two-pass convolution resampler (libImaging/Resample.c): per output coordinate a
//     window of the input (support = max(scale, 1): antialiased when shrinking), triangle weights normalised in DOUBLE and
//     rounded to 22-bit fixed point, horizontal pass -> uint8 -> vertical pass -> uint8, each sum started at 2^21 and clipped.
//     The coefficients are recomputed here per output pixel in double precision with the same operation order and without
//     fused multiply-adds (x86 Pillow wheels are SSE2 code); the two passes are evaluated on the fly (vertical taps x
//     horizontal taps), which is the same integer arithmetic.
//   * crops happen BEFORE the resize (torchvision resized_crop = img.crop(...).resize(...)): window clamps are relative to
//     the crop box; the un-augmented views resize the whole image to [1.15*size]^2 and crop the centre afterwards.
//   * ImageEnhance.{Brightness, Contrast, Color}.enhance(r) = ImagingBlend(degenerate, image, r) in float32:
//     (UINT8)(d + r * (v - d)) with truncation (r in [0,1]) or clipping (r > 1); the contrast degenerate is the grey level
//     int(mean(L) + 0.5) of the brightness-enhanced image, the colour degenerate the pixel's own grey value, L = (19595 R +
//     38470 G + 7471 B + 32768) >> 16.
//   * ToTensor / Normalize: (v / 255 - mean) / std in float32 with true divisions.
#include "mft_common.h"

namespace {

struct AugArgs {
    const unsigned char* src;      // [n_img][Hs][Ws][3] uint8 (PIL RGB order)
    const float* params;           // [n_views][n_img][10]: y0, x0, h, w (crop box in source pixels), rb, rc, rcol, flip_h, flip_v, enhance
    float* out;
    long long view_stride, img_stride;   // floats
    int n_img, Hs, Ws, size, S2, top;    // S2 = int(1.15*size), top = CenterCrop offset of the un-augmented views
    float mean[3], stdv[3];
};

constexpr int AUG_KMAX = 17;       // 2*ceil(support)+1 taps: source / target ratios up to 8

#pragma clang fp contract(off)
// Pillow precompute_coeffs + normalize_coeffs_8bpc for ONE output coordinate xx: first input index and fixed-point taps
__device__ __forceinline__ int pil_taps(int inSize, int outSize, int xx, int& xmin, int* kk) {
    const double scale = (double)((float)inSize - 0.f) / (double)outSize;
    const double filterscale = scale < 1.0 ? 1.0 : scale;
    const double support = 1.0 * filterscale;
    const double center = 0.0 + ((double)xx + 0.5) * scale;
    const double ss = 1.0 / filterscale;
    xmin = (int)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5);
    if (xmax > inSize) xmax = inSize;
    xmax -= xmin;
    double k[AUG_KMAX];
    double ww = 0.0;
    for (int x = 0; x < xmax; ++x) {
        double t = ((double)(x + xmin) - center + 0.5) * ss;
        if (t < 0.0) t = -t;
        const double w = t < 1.0 ? 1.0 - t : 0.0;
        k[x] = w;
        ww += w;
    }
    for (int x = 0; x < xmax; ++x) {
        double v = k[x];
        if (ww != 0.0) v /= ww;
        kk[x] = v < 0 ? (int)(-0.5 + v * (double)(1 << 22)) : (int)(0.5 + v * (double)(1 << 22));
    }
    return xmax;
}

__device__ __forceinline__ int pil_clip8(int v) {
    v >>= 22;
    return v < 0 ? 0 : (v > 255 ? 255 : v);
}

// ImagingBlend(degenerate d, image v, alpha) for one band value
__device__ __forceinline__ int pil_blend(int d, int v, float alpha) {
    const float t = __fadd_rn((float)d, __fmul_rn(alpha, (float)(v - d)));
    if (alpha >= 0.f && alpha <= 1.f) return (int)t & 0xff;           // (UINT8) of a value already inside [0, 255]
    if (t <= 0.f) return 0;
    if (t >= 255.f) return 255;
    return (int)t;
}

__device__ __forceinline__ int grey_l(int r, int g, int b) {
    return (int)((19595u * (unsigned)r + 38470u * (unsigned)g + 7471u * (unsigned)b + 0x8000u) >> 16);
}

__global__ __launch_bounds__(256) void augment_views_kernel(AugArgs p) {
    const int img = blockIdx.x, view = blockIdx.y;
    const float* pr = p.params + ((long long)view * p.n_img + img) * 10;
    const float rb = pr[4], rc = pr[5], rcol = pr[6];
    const bool fh = pr[7] != 0.f, fv = pr[8] != 0.f, enhance = pr[9] != 0.f;
    const unsigned char* s = p.src + (long long)img * p.Hs * p.Ws * 3;
    float* o = p.out + (long long)view * p.view_stride + (long long)img * p.img_stride;
    const int S = p.size;
    // geometry: augmented = crop box [y0, y0+ch) x [x0, x0+cw) resized to S x S; un-augmented = whole image resized to
    // S2 x S2, output pixel (y, x) = resized pixel (y + top, x + top)
    const int by0 = enhance ? (int)pr[0] : 0, bx0 = enhance ? (int)pr[1] : 0;
    const int bh = enhance ? (int)pr[2] : p.Hs, bw = enhance ? (int)pr[3] : p.Ws;
    const int outS = enhance ? S : p.S2, off = enhance ? 0 : p.top;
    const bool need_h = bw != outS, need_v = bh != outS;             // Pillow skips a pass whose size does not change
    int gsum = 0;
    for (int q = threadIdx.x; q < S * S; q += 256) {
        const int y = q / S, x = q - y * S;
        const int yy = (fv ? S - 1 - y : y) + off, xx = (fh ? S - 1 - x : x) + off;      // resized-image coordinates
        int kx[AUG_KMAX], ky[AUG_KMAX];
        int xmin = xx, ymin = yy, nx = 1, ny = 1;
        if (need_h) nx = pil_taps(bw, outS, xx, xmin, kx);
        if (need_v) ny = pil_taps(bh, outS, yy, ymin, ky);
        int v[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            int acc_v = 1 << 21;
            int single = 0;
            for (int a = 0; a < ny; ++a) {
                const unsigned char* row = s + ((long long)(by0 + ymin + a) * p.Ws + bx0 + xmin) * 3 + c;
                int hval;
                if (need_h) {
                    int acc_h = 1 << 21;
                    for (int b = 0; b < nx; ++b) acc_h += (int)row[b * 3] * kx[b];
                    hval = pil_clip8(acc_h);
                } else {
                    hval = row[0];
                }
                if (need_v) acc_v += hval * ky[a];
                else single = hval;
            }
            v[c] = need_v ? pil_clip8(acc_v) : single;
            if (enhance) v[c] = pil_blend(0, v[c], rb);                          // Brightness: degenerate = black
        }
        if (enhance) gsum += grey_l(v[0], v[1], v[2]);
#pragma unroll
        for (int c = 0; c < 3; ++c) o[(long long)q * 3 + c] = (float)v[c];
    }
    __shared__ int red[4];
    __shared__ int s_mean;
    if (enhance) {
#pragma unroll
        for (int offl = 32; offl > 0; offl >>= 1) gsum += __shfl_xor(gsum, offl, 64);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = gsum;
        __syncthreads();
        if (threadIdx.x == 0)          // int(ImageStat.Stat(L).mean[0] + 0.5): exact integer sum, double division
            s_mean = (int)((double)(red[0] + red[1] + red[2] + red[3]) / (double)(S * S) + 0.5);
        __syncthreads();
    }
    const int gm = enhance ? s_mean : 0;
    for (int q = threadIdx.x; q < S * S; q += 256) {
        int v[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) v[c] = (int)o[(long long)q * 3 + c];
        if (enhance) {
#pragma unroll
            for (int c = 0; c < 3; ++c) v[c] = pil_blend(gm, v[c], rc);          // Contrast: degenerate = solid mean grey
            const int gl = grey_l(v[0], v[1], v[2]);
#pragma unroll
            for (int c = 0; c < 3; ++c) v[c] = pil_blend(gl, v[c], rcol);        // Color: degenerate = the pixel's grey value
        }
#pragma unroll
        for (int c = 0; c < 3; ++c)     // ToTensor: float(v).div(255); Normalize: sub_(mean).div_(std) -- float32, true divisions
            o[(long long)q * 3 + c] = __fdiv_rn(__fsub_rn(__fdiv_rn((float)v[c], 255.f), p.mean[c]), p.stdv[c]);
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
    p.S2 = (int)(size * 1.15);                                   // transforms.Scale([int(1.15*size)]*2)
    p.top = (int)nearbyint((p.S2 - size) / 2.0);                 // CenterCrop: int(round(.)) -- Python 3 rounds half to even
    // tap budget: 2*ceil(max(source / target, 1)) + 1 <= AUG_KMAX (crops are never larger than the source)
    const double worst = (double)(Hs > Ws ? Hs : Ws) / (double)(size < p.S2 ? size : p.S2);
    if (2 * (int)ceil(worst < 1.0 ? 1.0 : worst) + 1 > AUG_KMAX) return MFT_EINVAL;
    for (int c = 0; c < 3; ++c) { p.mean[c] = mean3[c]; p.stdv[c] = std3[c]; }
    hipLaunchKernelGGL(augment_views_kernel, dim3(n_img, n_views), dim3(256), 0, (hipStream_t)stream, p);
    return mft_launch_status();
}
