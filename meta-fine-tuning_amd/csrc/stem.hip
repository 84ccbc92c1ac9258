// trunk.0: 7x7 stride-2 pad-3 convolution, Cin = 3 -> Cout = 64 (backbone.py:408), NHWC, on fp32 MFMA.
//
// The generic implicit-GEMM kernel gathers this layer's im2col rows element by element (Cin = 3 leaves no 16-byte
// runs).  Here a 256-thread workgroup owns 128 consecutive output pixels of one image x all 64 channels and is
// persistent over tiles:
//   * the input patch that covers the tile (<= 15 image rows for W = 84) is staged once in LDS with coalesced
//     16-byte loads (zero padded borders), double buffered across tiles;
//   * for a fixed kernel row kh the 21 values (kw, ci) of an im2col row are CONTIGUOUS in the patch, so the MFMA
//     A operand is a ds_read_b32 at (lane base + compile-time offset): K is walked as 7 x 11 pairs (j, j+1), the
//     22nd slot of each kernel row is a zero weight -> 77 MFMA steps for K = 147 (95 % useful);
//   * the weights of the wave's 32 output channels live in 77 VGPRs for the whole kernel (no B traffic at all).
#include "mft_common.h"

namespace {

template <int W_>
__global__ __launch_bounds__(256) void stem_conv_kernel(const float* __restrict__ in, const float* __restrict__ w_pk,
                                                        float* __restrict__ out, int H, int w_ld, int tiles_per_img,
                                                        int total_tiles) {
    constexpr int OW = W_ / 2;
    constexpr int PL = (W_ + 8) * 3;                 // patch row: 4 zero columns | W pixels | 4 zero columns
    constexpr int SPAN = (127 + OW - 1) / OW;        // a tile of 128 pixels touches <= SPAN+1 output rows
    constexpr int PR = 2 * SPAN + 7;
    constexpr int PQ = PR * PL / 4;                  // float4 slots per patch
    constexpr int NLD = (PQ + 255) / 256;
    constexpr int ROWQ = PL / 4;
    static_assert(PL % 4 == 0, "patch rows must be float4 multiples");
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int r = lane & 31, h = lane >> 5;
    const int OH = H / 2;
    const int OHW = OH * OW;

    // weights of output channel n = wn*32 + r: step (kh, p) pairs k = kh*21 + 2p + h (zero when 2p + h == 21)
    float breg[77];
    {
        const float* wrow = w_pk + (long long)(wn * 32 + r) * w_ld;
#pragma unroll
        for (int kh = 0; kh < 7; ++kh)
#pragma unroll
            for (int p = 0; p < 11; ++p) {
                const int j = 2 * p + h;
                breg[kh * 11 + p] = (j < 21) ? wrow[kh * 21 + j] : 0.f;
            }
    }

    f32x4 pre[NLD];
    auto load_patch = [&](int tile) {
        const int img = tile / tiles_per_img;
        const int mt = tile - img * tiles_per_img;
        const int oh0 = (mt * 128) / OW;
        const int ih0 = 2 * oh0 - 3;
        const float* src = in + (long long)img * H * W_ * 3;
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int q = tid + i * 256;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (q < PQ) {
                const int pr = q / ROWQ;
                const int cf = (q - pr * ROWQ) * 4 - 12;        // float index inside the image row
                const int ih = ih0 + pr;
                if (ih >= 0 && ih < H && cf >= 0 && cf < W_ * 3) v = *(const f32x4*)(src + (long long)ih * W_ * 3 + cf);
            }
            pre[i] = v;
        }
    };
    auto store_patch = [&](int buf) {
        float* P = smem + buf * (PR * PL);
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int q = tid + i * 256;
            if (q < PQ) *(f32x4*)(P + q * 4) = pre[i];
        }
    };

    int tile = blockIdx.x;
    if (tile >= total_tiles) return;
    load_patch(tile);
    store_patch(0);
    __syncthreads();
    int buf = 0;
    for (; tile < total_tiles; tile += gridDim.x) {
        const int next = tile + gridDim.x;
        if (next < total_tiles) load_patch(next);
        const int img = tile / tiles_per_img;
        const int mt = tile - img * tiles_per_img;
        const int oh0 = (mt * 128) / OW;
        const float* P = smem + buf * (PR * PL);
        int mb[2];
        int mrow[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int m = mt * 128 + wm * 64 + i * 32 + r;
            mrow[i] = m;
            const int mm = m < OHW ? m : OHW - 1;
            const int oh = mm / OW, ow = mm - oh * OW;
            mb[i] = ((oh - oh0) * 2) * PL + ow * 6 + 3 + h;
        }
        f32x16 acc0, acc1;
#pragma unroll
        for (int e = 0; e < 16; ++e) { acc0[e] = 0.f; acc1[e] = 0.f; }
#pragma unroll
        for (int kh = 0; kh < 7; ++kh)
#pragma unroll
            for (int p = 0; p < 11; ++p) {
                const float a0 = P[mb[0] + kh * PL + 2 * p];
                const float a1 = P[mb[1] + kh * PL + 2 * p];
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, breg[kh * 11 + p], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, breg[kh * 11 + p], acc1, 0, 0, 0);
            }
        // epilogue: C/D layout of the 32x32 MFMA: col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5)
        float* obase = out + ((long long)img * OHW) * 64 + wn * 32 + r;
        const int m00 = mt * 128 + wm * 64;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int row = (e & 3) + 8 * (e >> 2) + 4 * h;
            const int m0 = m00 + row, m1 = m00 + 32 + row;
            if (m0 < OHW) obase[(long long)m0 * 64] = acc0[e];
            if (m1 < OHW) obase[(long long)m1 * 64] = acc1[e];
        }
        (void)mrow;
        if (next < total_tiles) store_patch(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }
}

template <int W_>
int launch_stem(const float* in, const float* w, float* out, int n_img, int H, int w_ld, hipStream_t s) {
    constexpr int OW = W_ / 2;
    constexpr int PL = (W_ + 8) * 3;
    constexpr int SPAN = (127 + OW - 1) / OW;
    constexpr int PR = 2 * SPAN + 7;
    const int OHW = (H / 2) * OW;
    const int tpi = (OHW + 127) / 128;
    const long long total = (long long)n_img * tpi;
    if (total > 0x7fffffffLL) return MFT_EINVAL;
    const size_t lds = 2ull * PR * PL * sizeof(float);
    auto kern = stem_conv_kernel<W_>;
    if (lds > 64 * 1024) {
        static bool attr_done = false;
        if (!attr_done) {
            hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return (int)e;
            attr_done = true;
        }
    }
    const int wg_per_cu = 2;                          // 190 VGPR+AGPR -> 2 waves per SIMD
    long long grid = 256LL * wg_per_cu;
    if (grid > total) grid = total;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), lds, s, in, w, out, H, w_ld, tpi, (int)total);
    return mft_launch_status();
}


// ------------------------------------------------------------------------------------------------ stem cache fill in ONE pass
// functional.StemCache keeps, per resident support image, (a) the per-channel (mean, M2) of trunk.0's output and (b) per 3x3 /
// stride-2 / pad-1 pooling window its maximum and minimum.  The three-launch fill wrote the full-resolution output (3.7 GB per
// 8192 images) and read it back twice (mft_bn_image_moments, mft_pool_window_minmax).  Here the convolution's own workgroup
// finishes the job: one workgroup owns whole IMAGES and walks their output in tiles of THREE rows (126 of the 128 MFMA rows;
// 42 = 14 x 3), parks each tile's 3 x 42 x 64 outputs in an LDS ring of five rows -- a pooling window of tile t needs at most
// the last two rows of tile t-1 -- and produces from there
//   * the windows that became complete with this tile (pooled rows ((3t-2)/2, (3t+1)/2]) -> pmax / pmin, same values as
//     pool_window_minmax_kernel (max / min of the same convolution results: the K loop below is stem_conv_kernel's);
//   * the image's moments with bn_image_moments_kernel's arithmetic and summation order (shift by the image's first pixel;
//     thread (cq, rl) sums pixels p = rl (mod 16) in increasing order; the 16 partial sums are added in order at the image's end).
// The full-resolution output never reaches HBM.  LDS: 2 x 11-row input patches (24.3 KB) + the ring (53.8 KB) = 78 KB, two
// workgroups per CU as before.
template <int W_>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void stem_cache_kernel(const float* __restrict__ in, const float* __restrict__ w_pk,
                                                         float* __restrict__ pmax, float* __restrict__ pmin,
                                                         float* __restrict__ mean_img, float* __restrict__ m2_img, int w_ld,
                                                         int n_img) {
    constexpr int H_ = W_;
    constexpr int OW = W_ / 2, OH = H_ / 2;
    constexpr int PH = (OH + 2 - 3) / 2 + 1, PW = (OW + 2 - 3) / 2 + 1;
    constexpr int TR = 3, TILES = OH / TR, TPX = TR * OW;       // 3 rows = 126 pixels per tile
    static_assert(OH % TR == 0 && TPX <= 128, "tiles of three whole output rows");
    constexpr int PL = (W_ + 8) * 3;
    constexpr int PR = 2 * TR + 5;                              // input rows under three output rows
    constexpr int PQ = PR * PL / 4;
    constexpr int NLD = (PQ + 255) / 256;
    constexpr int ROWQ = PL / 4;
    constexpr int SROW = OW * 64;                               // one output row of all 64 channels
    static_assert(PL % 4 == 0 && 2 * 256 * 4 <= PR * PL, "patch rows are float4 multiples; the moment scratch fits a patch buffer");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const S = smem + 2 * PR * PL;                        // ring of five output rows, slot = row % 5

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int r = lane & 31, h = lane >> 5;
    const int cq = tid & 15, rl = tid >> 4;                     // moments / windows: channel quad, pixel lane

    float breg[77];
    {
        const float* wrow = w_pk + (long long)(wn * 32 + r) * w_ld;
#pragma unroll
        for (int kh = 0; kh < 7; ++kh)
#pragma unroll
            for (int p = 0; p < 11; ++p) {
                const int j = 2 * p + h;
                breg[kh * 11 + p] = (j < 21) ? wrow[kh * 21 + j] : 0.f;
            }
    }
    f32x4 pre[NLD];
    auto load_patch = [&](long long img, int t) {
        const int ih0 = 2 * TR * t - 3;
        const float* src = in + img * H_ * W_ * 3;
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int q = tid + i * 256;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (q < PQ) {
                const int pr = q / ROWQ;
                const int cf = (q - pr * ROWQ) * 4 - 12;
                const int ih = ih0 + pr;
                if (ih >= 0 && ih < H_ && cf >= 0 && cf < W_ * 3) v = *(const f32x4*)(src + (long long)ih * W_ * 3 + cf);
            }
            pre[i] = v;
        }
    };
    auto store_patch = [&](int buf) {
        float* P = smem + buf * (PR * PL);
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int q = tid + i * 256;
            if (q < PQ) *(f32x4*)(P + q * 4) = pre[i];
        }
    };
    // pixel geometry of this lane's two MFMA row blocks inside a tile (the same for every tile: tiles are whole rows)
    int mb[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = wm * 64 + i * 32 + r;
        const int mm = m < TPX ? m : TPX - 1;
        const int ohl = mm / OW, ow = mm - ohl * OW;
        mb[i] = (ohl * 2) * PL + ow * 6 + 3 + h;
    }
    const int mo_base = wm * 64 + 4 * h;                        // C/D layout: col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5)
    const int ch = wn * 32 + r;

    long long img = blockIdx.x;
    if (img >= n_img) return;
    load_patch(img, 0);
    store_patch(0);
    __syncthreads();
    int buf = 0;
    for (; img < n_img; img += gridDim.x) {
        f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = s1, sh = s1;
        for (int t = 0; t < TILES; ++t) {
            const bool last_t = t + 1 == TILES;
            const long long nimg = last_t ? img + gridDim.x : img;
            const bool more = nimg < n_img;
            if (more) load_patch(nimg, last_t ? 0 : t + 1);
            const float* P = smem + buf * (PR * PL);
            f32x16 acc0, acc1;
#pragma unroll
            for (int e = 0; e < 16; ++e) { acc0[e] = 0.f; acc1[e] = 0.f; }
#pragma unroll
            for (int kh = 0; kh < 7; ++kh)
#pragma unroll
                for (int p = 0; p < 11; ++p) {
                    const float a0 = P[mb[0] + kh * PL + 2 * p];
                    const float a1 = P[mb[1] + kh * PL + 2 * p];
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, breg[kh * 11 + p], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, breg[kh * 11 + p], acc1, 0, 0, 0);
                }
            __syncthreads();                                    // the previous tile's readers of the ring (and of this patch) are done
            const int row0 = TR * t;
            const int sl0 = row0 % 5, sl1 = (row0 + 1) % 5, sl2 = (row0 + 2) % 5;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int mo = mo_base + i * 32 + (e & 3) + 8 * (e >> 2);
                    if (mo < TPX) {
                        const int ol = (mo >= OW) + (mo >= 2 * OW);
                        const int slot = ol == 0 ? sl0 : (ol == 1 ? sl1 : sl2);
                        S[slot * SROW + (mo - ol * OW) * 64 + ch] = i == 0 ? acc0[e] : acc1[e];
                    }
                }
            }
            if (more) store_patch(buf ^ 1);
            __syncthreads();
            // ---- pooling windows completed by this tile: pooled rows i with 3t-1 < min(2i+1, OH-1) <= 3t+2
            const int i_first = t == 0 ? 0 : (TR * t - 2) / 2 + 1;
            int i_last = (TR * t + 1) / 2;
            if (i_last > PH - 1 || last_t) i_last = PH - 1;
            const int n_items = (i_last - i_first + 1) * PW * 16;
            for (int it = tid; it < n_items; it += 256) {
                const int c4 = (it & 15) * 4;
                const int j = (it >> 4) % PW, i = i_first + (it >> 4) / PW;
                f32x4 hi = {-3.4e38f, -3.4e38f, -3.4e38f, -3.4e38f}, lo = {3.4e38f, 3.4e38f, 3.4e38f, 3.4e38f};
#pragma unroll
                for (int dh = 0; dh < 3; ++dh) {
                    const int ih = 2 * i - 1 + dh;
                    if (ih < 0 || ih >= OH) continue;
#pragma unroll
                    for (int dw = 0; dw < 3; ++dw) {
                        const int iw = 2 * j - 1 + dw;
                        if (iw < 0 || iw >= OW) continue;
                        const f32x4 v = *(const f32x4*)(S + (ih % 5) * SROW + iw * 64 + c4);
#pragma unroll
                        for (int e = 0; e < 4; ++e) { hi[e] = fmaxf(hi[e], v[e]); lo[e] = fminf(lo[e], v[e]); }
                    }
                }
                const long long o = ((img * PH + i) * PW + j) * 64 + c4;
                *(f32x4*)(pmax + o) = hi;
                *(f32x4*)(pmin + o) = lo;
            }
            // ---- moments: pixels p = rl (mod 16) of this tile, increasing (bn_image_moments_kernel's order)
            if (t == 0) sh = *(const f32x4*)(S + cq * 4);                                   // the image's first pixel
            const int p0 = TPX * t;
            for (int pp = p0 + ((rl - p0) & 15); pp < p0 + TPX; pp += 16) {
                const int orow = pp / OW, ow = pp - orow * OW;
                f32x4 v = *(const f32x4*)(S + (orow % 5) * SROW + ow * 64 + cq * 4);
                v -= sh;
                s1 += v;
                s2 += v * v;
            }
            buf ^= 1;
        }
        // ---- the image's 16 partial sums, added in order (scratch = the patch buffer the last tile consumed: it is rewritten only
        // after the next tile's first barrier)
        f32x4* red = (f32x4*)(smem + (buf ^ 1) * (PR * PL));
        red[rl * 16 + cq] = s1;
        red[256 + rl * 16 + cq] = s2;
        __syncthreads();
        if (rl == 0) {
#pragma unroll
            for (int k = 1; k < 16; ++k) {
                s1 += red[k * 16 + cq];
                s2 += red[256 + k * 16 + cq];
            }
            const float inv = 1.f / (float)(OH * OW);
            f32x4 mu, m2;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float d = s1[e] * inv;
                mu[e] = sh[e] + d;
                m2[e] = fmaxf(s2[e] - s1[e] * d, 0.f);
            }
            *(f32x4*)(mean_img + img * 64 + cq * 4) = mu;
            *(f32x4*)(m2_img + img * 64 + cq * 4) = m2;
        }
    }
}

}  // namespace

// Returns MFT_EINVAL when the shape is not one of the specialised ones (the caller then uses the generic kernel).
int mft_stem_conv_dispatch(const float* in, const float* w, float* out, int n_img, int H, int W, int w_ld,
                           hipStream_t s) {
    if (H % 2 != 0 || H < 8) return MFT_EINVAL;
    if (W == 84) return launch_stem<84>(in, w, out, n_img, H, w_ld, s);
    if (W == 224) return launch_stem<224>(in, w, out, n_img, H, w_ld, s);
    return MFT_EINVAL;
}

/* Stem-cache fill in one launch (csrc/stem.hip, stem_cache_kernel): trunk.0 + per-image moments + per-window (max, min) without the
 * full-resolution output.  MFT_EINVAL outside the specialised shape (84 x 84): the caller runs the three separate launches. */
extern "C" int mft_stem_cache_fill(const float* in, const float* w, int w_ld, int n_img, int H, int W, float* pmax, float* pmin,
                                   float* mean_img, float* m2_img, void* stream) {
    if (H != 84 || W != 84 || n_img <= 0 || w_ld < 147) return MFT_EINVAL;
    constexpr int W_ = 84, PL = (W_ + 8) * 3, PR = 11, SROW = (W_ / 2) * 64;
    const size_t lds = (2ull * PR * PL + 5ull * SROW) * sizeof(float);
    auto kern = stem_cache_kernel<W_>;
    static MftPerDeviceOnce attr_once;
    if (attr_once.need()) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        attr_once.mark();
    }
    long long grid = 256LL * 2;
    if (grid > n_img) grid = n_img;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), lds, (hipStream_t)stream, in, w, pmax, pmin, mean_img, m2_img, w_ld,
                       n_img);
    return mft_launch_status();
}
