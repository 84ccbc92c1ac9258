// Weight gradient + Adam + DATA gradient of a per-episode 3x3 / stride 1 / pad 1 convolution in one pass over the weights
// (loss.backward() + optimizer.step() for trunk.7.C2, finetune.py:293-297 over backbone.py:253-256).
//
// The separate launches read trunk.7.C2's weights three times per inner step: forward, data gradient (9.4 MB per episode)
// and the fused weight-gradient + Adam kernel (read + write).  Here the Adam kernel's workgroup keeps the OLD weight tile it
// has just loaded and multiplies it with dY before moving on, so the data gradient costs no weight traffic at all:
//   workgroup = (episode g, tap (kh,kw), 64 input channels ci); it walks the Cout/64 output-channel tiles:
//     dW[co,ci]   = sum_m dY[m,co] * X[m,(tap,ci)]          (fp32 MFMA, as conv_wgrad_kernel)
//     Adam on (w, m, v)[co, tap, ci]                        (streamed once, nontemporal; next tile's loads already in flight)
//     dXp[m,ci]  += sum_co dY[m,co] * w_old[co,tap,ci]      (fp32 MFMA from the tile just updated, accumulators in registers)
//   and writes dXp[g][tap][m][ci] (0.8 MB per episode instead of 9.4 MB of weight reads); mft_col2im_bn_backward_small then
//   sums the 9 taps into dx (col2im) and applies the BatchNorm + ReLU backward of the layer in front.
// Deterministic: fixed co order inside the workgroup, fixed tap order in the reducer.
#include "mft_common.h"

namespace {

struct WdArgs {
    const float* x;          // [groups][rows][ldx]   activation entering the convolution (r1)
    const float* dy;         // [groups][rows][ldy]   gradient of the convolution output (dc2)
    float* w; float* m; float* v;      // [groups][Cout][9*Cin]
    float* dxp;              // [groups][9][rows][Cin]
    int ldx, ldy, H, W, Cin, Cout, rows, ipg;
    long long wgs;           // group stride of w/m/v
    float step_size, inv_sqrt_bc2, b1, b2, eps;
    const float* hyper;      // optional device {step_size, inv_sqrt_bc2}
};

constexpr int YLD = 65;      // dY tile row stride (floats): conflict-free for both the [m][co] and the transposed fragment reads
constexpr int GLD = 68;      // gradient / old-weight tile row stride

__global__ __launch_bounds__(256) void wgrad_adam_dgrad_kernel(WdArgs p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Xs = smem;                    // [64 rows][64 ci]   im2col rows of this tap (zero rows beyond p.rows / outside the image)
    float* Ys = Xs + 64 * 64;            // [64 rows][YLD]     dY tile of the current output-channel tile
    float* Gs = Ys + 64 * YLD;           // [64 co][GLD]       gradient tile, then the old weight tile

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int r = lane & 31, h = lane >> 5;
    const int g = blockIdx.y;
    const int tiles_ci = p.Cin / 64;
    const int tci = blockIdx.x % tiles_ci;
    const int tap = blockIdx.x / tiles_ci;
    const int kh = tap / 3, kw = tap - kh * 3;
    const int ci0 = tci * 64;
    const int hw = p.H * p.W;
    const int Kpad = 9 * p.Cin;
    const int n_co = p.Cout / 64;
    const int ksteps = (p.rows + 1) >> 1;                    // 32x32x2 MFMA: two reduction rows per instruction

    // ---- this tap's im2col rows of X -> LDS (once)
    {
        const float* xg = p.x + (long long)g * p.rows * p.ldx;
        for (int i = tid; i < 64 * 16; i += 256) {
            const int row = i >> 4, c = (i & 15) * 4;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (row < p.rows) {
                const int img = row / hw;
                const int rem = row - img * hw;
                const int oh = rem / p.W, ow = rem - oh * p.W;
                const int ih = oh - 1 + kh, iw = ow - 1 + kw;
                if (ih >= 0 && ih < p.H && iw >= 0 && iw < p.W)
                    v = *(const f32x4*)(xg + (long long)((img * p.H + ih) * p.W + iw) * p.ldx + ci0 + c);
            }
            *(f32x4*)(Xs + row * 64 + c) = v;
        }
    }
    // ---- w/m/v tile registers: thread (q, rr) owns rows rr + 16u (u = 0..3), 4 consecutive ci
    const int q = tid & 15, rr = tid >> 4;
    const long long gbase = (long long)g * p.wgs + (long long)tap * p.Cin + ci0 + 4 * q;
    f32x4 cw[4], cm[4], cv[4], nw[4], nm[4], nv[4];
    auto load_tile = [&](int t, f32x4* W_, f32x4* M_, f32x4* V_) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long long gi = gbase + (long long)(t * 64 + rr + 16 * u) * Kpad;
            M_[u] = __builtin_nontemporal_load((const f32x4*)(p.m + gi));
            V_[u] = __builtin_nontemporal_load((const f32x4*)(p.v + gi));
            W_[u] = __builtin_nontemporal_load((const f32x4*)(p.w + gi));
        }
    };
    load_tile(0, cw, cm, cv);
    const float step_size = p.hyper ? p.hyper[0] : p.step_size;
    const float inv_sqrt_bc2 = p.hyper ? p.hyper[1] : p.inv_sqrt_bc2;
    const float* dyg = p.dy + (long long)g * p.rows * p.ldy;

    f32x16 accx;                                            // dXp tile of this wave: rows wm*32.., channels wn*32..
#pragma unroll
    for (int e = 0; e < 16; ++e) accx[e] = 0.f;

    // dY tile of the next output-channel tile travels in registers one iteration ahead (its L2 latency under the MFMAs)
    f32x4 ydy[4];
    auto load_dy = [&](int t) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = tid + 256 * k;
            const int row = i >> 4, c = (i & 15) * 4;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (row < p.rows) v = *(const f32x4*)(dyg + (long long)row * p.ldy + t * 64 + c);
            ydy[k] = v;
        }
    };
    load_dy(0);
    for (int t = 0; t < n_co; ++t) {
        const int tn = t + 1 < n_co ? t + 1 : t;             // last iteration: harmless re-load of its own tile (L2 hit)
        load_tile(tn, nw, nm, nv);
        // dY tile -> LDS
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = tid + 256 * k;
            const int row = i >> 4, c = (i & 15) * 4;
            float* d = Ys + row * YLD + c;
            d[0] = ydy[k][0]; d[1] = ydy[k][1]; d[2] = ydy[k][2]; d[3] = ydy[k][3];
        }
        load_dy(tn);
        __syncthreads();
        // dW tile: M = co (64), N = ci (64), K = rows
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
        for (int s = 0; s < ksteps; ++s) {
            const float a = Ys[(2 * s + h) * YLD + wm * 32 + r];
            const float b = Xs[(2 * s + h) * 64 + wn * 32 + r];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int row = (e & 3) + 8 * (e >> 2) + 4 * h;
            Gs[(wm * 32 + row) * GLD + wn * 32 + r] = acc[e];
        }
        __syncthreads();
        // Adam on the tile (same arithmetic as conv_wgrad_kernel<.., ADAM>); keep the old weights
        f32x4 wold[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int row = rr + 16 * u;
            const long long gi = gbase + (long long)(t * 64 + row) * Kpad;
            const f32x4 ge = *(const f32x4*)(Gs + row * GLD + 4 * q);
            wold[u] = cw[u];
            // same epilogue arithmetic as wgrad_adam_rows_kernel's default: packed moment updates, hardware rcp / sqrt (1 ulp)
            mft_adam4_fast(cm[u], cv[u], cw[u], ge, p.b1, p.b2, p.eps, step_size, inv_sqrt_bc2);
            __builtin_nontemporal_store(cm[u], (f32x4*)(p.m + gi));
            __builtin_nontemporal_store(cv[u], (f32x4*)(p.v + gi));
            __builtin_nontemporal_store(cw[u], (f32x4*)(p.w + gi));
        }
        __syncthreads();                                     // every gradient read of Gs is done
#pragma unroll
        for (int u = 0; u < 4; ++u) *(f32x4*)(Gs + (rr + 16 * u) * GLD + 4 * q) = wold[u];
        __syncthreads();
        // dXp tile: M = rows (64), N = ci (64), K = co (64)
        for (int s = 0; s < 32; ++s) {
            const float a = Ys[(wm * 32 + r) * YLD + 2 * s + h];
            const float b = Gs[(2 * s + h) * GLD + wn * 32 + r];
            accx = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, accx, 0, 0, 0);
        }
        __syncthreads();                                     // Ys / Gs are rewritten by the next tile
#pragma unroll
        for (int u = 0; u < 4; ++u) { cw[u] = nw[u]; cm[u] = nm[u]; cv[u] = nv[u]; }
    }
    float* o = p.dxp + (((long long)g * 9 + tap) * p.rows) * p.Cin + ci0 + wn * 32 + r;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int row = wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
        if (row < p.rows) o[(long long)row * p.Cin] = accx[e];
    }
}

// col2im over the 9 per-tap partials + BatchNorm/ReLU backward of the layer in front of the convolution (same formulas and
// reduction order as bn_backward_kernel, csrc/bn.hip): one workgroup per (group, 64 channels).
struct C2iArgs {
    const float* dxp;        // [groups][9][rows][C]
    const float* x_raw; const float* relu_out; float* dx;        // [groups*rows][C]
    int C, rows, H, W;
    const float* mean; const float* rstd; const float* gamma; long long gbs;
    float* dgamma; float* dbeta;
};

constexpr int CI_ROWS = 16;

__global__ __launch_bounds__(256) void col2im_bn_backward_kernel(C2iArgs p) {
    const int cq = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int c = blockIdx.x * 64 + cq * 4;
    const int g = blockIdx.y;
    const long long row0 = (long long)g * p.rows;
    const int hw = p.H * p.W;
    __shared__ f32x4 red1[CI_ROWS][16];
    __shared__ f32x4 red2[CI_ROWS][16];
    const f32x4 mu = *(const f32x4*)(p.mean + (long long)g * p.C + c);
    const f32x4 rs = *(const f32x4*)(p.rstd + (long long)g * p.C + c);
    const f32x4 ga = *(const f32x4*)(p.gamma + g * p.gbs + c);
    // dx of the convolution input at pixel (img, ih, iw): tap (kh,kw) contributes the partial of output pixel (ih-kh+1, iw-kw+1)
    auto dy_at = [&](int rr) {
        const int img = rr / hw;
        const int rem = rr - img * hw;
        const int ih = rem / p.W, iw = rem - ih * p.W;
        f32x4 d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int oh = ih - tap / 3 + 1, ow = iw - tap % 3 + 1;
            if (oh >= 0 && oh < p.H && ow >= 0 && ow < p.W)
                d += *(const f32x4*)(p.dxp + (((long long)g * 9 + tap) * p.rows + (img * hw + oh * p.W + ow)) * p.C + c);
        }
        const f32x4 o = *(const f32x4*)(p.relu_out + (row0 + rr) * p.C + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) d[e] = o[e] > 0.f ? d[e] : 0.f;
        return d;
    };
    f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = s1;
    for (int rr = rl; rr < p.rows; rr += CI_ROWS) {
        const f32x4 d = dy_at(rr);
        const f32x4 xh = (*(const f32x4*)(p.x_raw + (row0 + rr) * p.C + c) - mu) * rs;
        s1 += d;
        s2 += d * xh;
    }
    red1[rl][cq] = s1;
    red2[rl][cq] = s2;
    __syncthreads();
    s1 = red1[0][cq];
    s2 = red2[0][cq];
#pragma unroll
    for (int k = 1; k < CI_ROWS; ++k) {
        s1 += red1[k][cq];
        s2 += red2[k][cq];
    }
    if (rl == 0) {
        *(f32x4*)(p.dgamma + (long long)g * p.C + c) = s2;
        *(f32x4*)(p.dbeta + (long long)g * p.C + c) = s1;
    }
    const float inv = 1.f / (float)p.rows;
    const f32x4 m1 = s1 * inv, m2 = s2 * inv;
    const f32x4 k = ga * rs;
    for (int rr = rl; rr < p.rows; rr += CI_ROWS) {
        const f32x4 d = dy_at(rr);
        const f32x4 xh = (*(const f32x4*)(p.x_raw + (row0 + rr) * p.C + c) - mu) * rs;
        *(f32x4*)(p.dx + (row0 + rr) * p.C + c) = k * (d - m1 - xh * m2);
    }
}

}  // namespace

extern "C" long long mft_conv2d_wgrad_adam_dgrad_ws_floats(int n_img, int H, int W, int Cin) {
    return 9LL * n_img * H * W * Cin;
}

static int wad_launch(const float* x, int ldx, const float* dy, int ldy, float* w, float* m, float* v, float* dx_partials,
                      int n_img, int H, int W, int Cin, int Cout, int imgs_per_group, long long group_stride, int step,
                      const float* hyper, float lr, float beta1, float beta2, float eps, void* stream) {
    if (imgs_per_group <= 0 || n_img % imgs_per_group != 0 || group_stride == 0 || (hyper == nullptr && step < 1)) return MFT_EINVAL;
    const int rows = imgs_per_group * H * W;
    if (rows > 64 || Cin % 64 != 0 || Cout % 64 != 0 || ldx % 4 != 0 || ldy % 4 != 0 || !dx_partials) return MFT_EINVAL;
    if (hyper != nullptr) step = 1;
    WdArgs p;
    p.x = x; p.dy = dy; p.w = w; p.m = m; p.v = v; p.dxp = dx_partials;
    p.ldx = ldx; p.ldy = ldy; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.rows = rows; p.ipg = imgs_per_group;
    p.wgs = group_stride;
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    p.step_size = (float)((double)lr / bc1);
    p.inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
    p.b1 = beta1; p.b2 = beta2; p.eps = eps; p.hyper = hyper;
    const size_t lds = (size_t)(64 * 64 + 64 * YLD + 64 * GLD) * sizeof(float);
    hipLaunchKernelGGL(wgrad_adam_dgrad_kernel, dim3(9 * (Cin / 64), n_img / imgs_per_group), dim3(256), lds, (hipStream_t)stream, p);
    return mft_launch_status();
}

extern "C" int mft_conv2d_wgrad_adam_dgrad_nhwc(const float* x, int ldx, const float* dy, int ldy, float* w, float* m, float* v,
                                                float* dx_partials, int n_img, int H, int W, int Cin, int Cout,
                                                int imgs_per_group, long long group_stride, int step, float lr, float beta1,
                                                float beta2, float eps, void* stream) {
    return wad_launch(x, ldx, dy, ldy, w, m, v, dx_partials, n_img, H, W, Cin, Cout, imgs_per_group, group_stride, step, nullptr,
                      lr, beta1, beta2, eps, stream);
}

extern "C" int mft_conv2d_wgrad_adam_dgrad_nhwc_dev(const float* x, int ldx, const float* dy, int ldy, float* w, float* m,
                                                    float* v, float* dx_partials, int n_img, int H, int W, int Cin, int Cout,
                                                    int imgs_per_group, long long group_stride, const float* hyper, float beta1,
                                                    float beta2, float eps, void* stream) {
    if (hyper == nullptr) return MFT_EINVAL;
    return wad_launch(x, ldx, dy, ldy, w, m, v, dx_partials, n_img, H, W, Cin, Cout, imgs_per_group, group_stride, 1, hyper, 0.f,
                      beta1, beta2, eps, stream);
}

extern "C" int mft_col2im_bn_backward_small(const float* dx_partials, const float* x_raw, const float* relu_out, float* dx,
                                            int n_img, int H, int W, int C, int imgs_per_group, const float* mean,
                                            const float* rstd, const float* gamma, long long gb_group_stride, float* dgamma,
                                            float* dbeta, void* stream) {
    if (imgs_per_group <= 0 || n_img % imgs_per_group != 0 || C % 64 != 0) return MFT_EINVAL;
    C2iArgs p{dx_partials, x_raw, relu_out, dx, C, imgs_per_group * H * W, H, W, mean, rstd, gamma, gb_group_stride, dgamma, dbeta};
    hipLaunchKernelGGL(col2im_bn_backward_kernel, dim3(C / 64, n_img / imgs_per_group), dim3(256), 0, (hipStream_t)stream, p);
    return mft_launch_status();
}
