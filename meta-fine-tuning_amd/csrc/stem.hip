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

}  // namespace

// Returns MFT_EINVAL when the shape is not one of the specialised ones (the caller then uses the generic kernel).
int mft_stem_conv_dispatch(const float* in, const float* w, float* out, int n_img, int H, int W, int w_ld,
                           hipStream_t s) {
    if (H % 2 != 0 || H < 8) return MFT_EINVAL;
    if (W == 84) return launch_stem<84>(in, w, out, n_img, H, w_ld, s);
    if (W == 224) return launch_stem<224>(in, w, out, n_img, H, w_ld, s);
    return MFT_EINVAL;
}
