// Weight gradient + Adam of inner step t that ALSO computes the convolution of inner step t+1 with the weights it has just
// updated (finetune.py:286-299: `output = pretrained_model(z_batch)` of the next iteration is the first reader of what
// `delta_opt.step()` wrote).
//
// Why: in the episode-batched inner loop every episode owns its own trunk.7 weights (14.7 MB) and uses each element for 45
// output pixels per step.  The step moves 7.64 "parameter units" of HBM traffic per episode: Adam reads and writes w, m, v (6),
// the data gradient re-reads C2 (0.64) and the NEXT step's forward reads w once more (1.0) -- only to multiply it with 45 pixel
// rows.  Here the forward read disappears: the workgroup that has just produced a 32 x 128 tile of updated weights multiplies it,
// while it is still in LDS, with the next step's activation rows (which the frozen trunk, running ahead on its own stream, has
// already produced) and accumulates the next step's convolution output of its 32 output channels along the walk.
//
// Structure (one workgroup = one episode x 32 output channels, walking all K tiles of 128 -- (tap, ci) order, i.e. along the
// 18 KB weight rows; 4 waves):
//   per K tile:  G = dY^T . im2col(x_t)            v_mfma_f32_32x32x2_f32, reduction rows 2t+h as wgrad_adam_rows_kernel
//                (w, m, v) <- Adam(G)              row-stream layout, 512 B runs, nontemporal; w' also parked in LDS over G
//                out[48 px][32 co] += w' . im2col(x_{t+1})   v_mfma_f32_16x16x4_f32, each wave a 32-wide k slice of the tile;
//                                                  A = w' fragments (ds_read_b128), B = activation rows straight from L2
//   after the walk: the four waves' partial outputs are summed in fixed order and the layer's epilogue runs on the 32 channels
//   the workgroup owns -- every BatchNorm of the block normalises per channel over the episode's <= 48 pixels:
//     RAW    shortcut 1x1 convolution: raw output only (its BatchNorm is folded into EXIT)
//     ENTRY  C1: BatchNorm1 statistics + affine + ReLU -> r1                      (backbone.py:252-254)
//     EXIT   C2: BatchNorm2 + BatchNorm(shortcut) + add + ReLU + global average pool   (backbone.py:255-261, :438 AvgPool2d)
// The gradient and (w, m, v) are bit-identical to wgrad_adam_rows_kernel (same reduction order, same Adam expressions).
// All request streams are one tile deep: im2col rows, then w/m/v, then next-step activation rows of tile k+1 are requested
// while tile k is being multiplied (the load counter is in order, so the reduction never waits for the big w/m/v requests).
#include "mft_common.h"
#include <math.h>

namespace {

struct WfArgs {
    // weight gradient + Adam of step t
    const float* in; const float* dy; float* dw;
    float* w; float* m; float* v;
    int ldi, ldy;
    int H, W, Cin, OH, OW, Cout, KH, KW, stride, pad;
    int Kpad;                    // KH*KW*Cin: row length of w / m / v
    int rows, ipg;               // reduction rows (= output pixels) per episode, <= 48; images per episode
    int tiles_ci;                // Cin / 128
    int inv_ohw, inv_ow;         // ceil(65536 / (OH*OW)), ceil(65536 / OW): row -> (image, oh, ow) without integer division
    int mma_rows;                // rows that get matrix instructions in the reduction (rows, or 64 = untrimmed)
    long long dwgs;              // group stride of w / m / v / dw
    float step_size, inv_sqrt_bc2, b1, b2, eps;
    const float* hyper;          // optional device {step_size, inv_sqrt_bc2}
    // convolution of step t+1 (null xn: none -- the last inner step)
    const float* xn;             // next step's input activation, same geometry and group layout as `in`
    float* raw;                  // [groups][rows][Cout]  raw convolution output of step t+1
    float* act;                  // ENTRY: ReLU(BN(raw));  EXIT: block output
    const float* gamma; const float* beta; long long gbs;
    float* mean; float* rstd;    // [groups][Cout]
    const float* sc; const float* gs; const float* bs; float* means; float* rstds;      // EXIT: shortcut branch
    float* pooled; int hw;       // EXIT: [groups][ipg][Cout], pixels per image
    float bn_eps;
    int co_blocks;               // Cout / 32: workgroups per episode
    int xcd_groups;              // 1: all workgroups of an episode on ONE XCD (needs groups % 8 == 0)
};

enum { WF_RAW = 0, WF_ENTRY = 1, WF_EXIT = 2 };

__device__ __forceinline__ float lane8_sum(float v) {       // sum over the 8 lanes that share tid >> 3
    v += __shfl_xor(v, 1, 64);
    v += __shfl_xor(v, 2, 64);
    v += __shfl_xor(v, 4, 64);
    return v;
}

// NT: matrix instructions of the reduction (2 rows each; rows beyond the episode's are zeros and change no bit of the sum)
template <int MODE, bool FAST, int NT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void wgrad_adam_fwd_kernel(WfArgs p) {
    constexpr int BM = 32, BN = 128, BLD = BN + 32, GLD = BN + 4, RLD = 36, TLD = 33;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                     // [48][BM]   dY rows of this output-channel tile (resident for the walk)
    float* Bs = smem + 48 * BM;           // [48][BLD]  im2col rows of the current K tile
    float* Gs = Bs + 48 * BLD;            // [32][GLD]  gradient tile, overwritten in place by the updated weight tile
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;            // 32x32x2 fragment coordinates (reduction)
    const int fm = lane & 15, fq = lane >> 4;          // 16x16x4 fragment coordinates (next step's convolution)
    // Workgroup -> (episode, 32 output channels).  The co_blocks workgroups of one episode read the SAME activation rows (x_t for the
    // gradient, x_{t+1} for the next step's forward: 92 KB each for trunk.7.C2, re-read 9 x 16 times through L2).  Workgroup ids
    // are dealt round-robin to the 8 XCDs, each with its own 4 MB L2: in the natural order an episode's 16 workgroups land on all
    // eight XCDs, every L2 holds the rows of all ~32 episodes in flight beside the w / m / v stream, and 43 % of those re-reads
    // missed (PMC: FETCH_SIZE 1.21x the w / m / v bytes).  Remapped: XCD x works through episodes x, x + 8, ... with all of an
    // episode's workgroups, so an L2 holds 4 episodes' rows at a time.
    int g, co0;
    if (p.xcd_groups) {
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        g = (slot / p.co_blocks) * 8 + xcd;
        co0 = (slot % p.co_blocks) * BM;
    } else {
        g = blockIdx.x / p.co_blocks;
        co0 = (blockIdx.x % p.co_blocks) * BM;
    }
    const int ohw = p.OH * p.OW;
    const int rows = p.rows;
    const long long row0 = (long long)g * rows;
    const long long img0 = (long long)g * p.ipg;
    const int n_kt = p.KH * p.KW * p.tiles_ci;
    const bool fwd = p.xn != nullptr;

    const int arow = tid >> 3, acol = (tid & 7) * 4;
    const int brow = tid >> 5, bcol = (tid & 31) * 4;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int m = arow + 32 * j;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (m < rows) v = *(const f32x4*)(p.dy + (row0 + m) * p.ldy + co0 + acol);
        if (m < 48) *(f32x4*)(As + m * BM + acol) = v;
    }
    // pixel geometry, packed (image << 16 | (ih0 + 64) << 8 | (iw0 + 64)); -1 = row beyond the episode's pixels
    auto geom = [&](int m) {
        if (m >= rows) return -1;
        const int img = (m * p.inv_ohw) >> 16, rem = m - img * ohw;
        const int oh = (rem * p.inv_ow) >> 16, ow = rem - oh * p.OW;
        return (img << 16) | ((oh * p.stride - p.pad + 64) << 8) | (ow * p.stride - p.pad + 64);
    };
    int bgeo[6], xgeo[3];
#pragma unroll
    for (int j = 0; j < 6; ++j) bgeo[j] = geom(brow + 8 * j);
#pragma unroll
    for (int nb = 0; nb < 3; ++nb) xgeo[nb] = geom(nb * 16 + fm);
    // element offset of the tap's input pixel inside the episode (always a valid address) and whether the tap is inside the image:
    // the loads are unconditional (a clamped address, then a select) -- no branch per load
    auto pix_off = [&](int geo, int kh, int kw, bool& ok) -> int {
        const int img = (geo >> 16) & 255, ih = ((geo >> 8) & 255) - 64 + kh, iw = (geo & 255) - 64 + kw;
        ok = geo >= 0 && ih >= 0 && ih < p.H && iw >= 0 && iw < p.W;
        return ok ? ((img * p.H + ih) * p.W + iw) * p.ldi : 0;
    };
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

    // (tap, input-channel tile) of the tile the NEXT operand requests are for, carried incrementally (no division per tile)
    int nkh = 0, nkw = 0, nci0 = 0;
    auto advance = [&]() {
        nci0 += BN;
        if (nci0 == p.Cin) {
            nci0 = 0;
            if (++nkw == p.KW) { nkw = 0; ++nkh; }
        }
    };
    f32x4 vb[6];
    const float* const in_e = p.in + (img0 * p.H * p.W) * p.ldi;            // wave-uniform bases; 32-bit lane offsets
    auto load_b = [&](int kh, int kw, int ci0) {
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            bool ok;
            const int o = pix_off(bgeo[j], kh, kw, ok);
            const f32x4 v = *(const f32x4*)(in_e + (o + ci0 + bcol));
            vb[j] = ok ? v : zero4;
        }
    };
    f32x4 xb[3][2];
    const float* const xn_e = p.xn + (img0 * p.H * p.W) * p.ldi;
    auto load_x = [&](int kh, int kw, int ci0) {
        const int xo = ci0 + 32 * wave + 4 * fq;
#pragma unroll
        for (int nb = 0; nb < 3; ++nb) {
            bool ok;
            const int o = pix_off(xgeo[nb], kh, kw, ok);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const f32x4 v = *(const f32x4*)(xn_e + (o + xo + 16 * j));
                xb[nb][j] = ok ? v : zero4;
            }
        }
    };
    // this workgroup's 32 rows of w / m / v: wave-uniform 64-bit bases, 32-bit lane offsets (a row block is < 2^31 floats)
    const int q = tid & 31, rr = tid >> 5;
    const long long tile_base = (long long)g * p.dwgs + (long long)co0 * p.Kpad;
    float* const wg_ = p.w + tile_base;
    float* const mg_ = p.m + tile_base;
    float* const vg_ = p.v + tile_base;
    const int lo0 = rr * p.Kpad + 4 * q;
    // two register sets: while Adam consumes tile k from one, tile k+1 sits (landed or landing) in the other and tile k+2 is
    // requested into the first as soon as Adam is done with it
    f32x4 am[4], av[4], aw[4], bm[4], bv[4], bw[4];
    auto load_wmv = [&](int kt, f32x4 (&M)[4], f32x4 (&V)[4], f32x4 (&W)[4]) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int gi = lo0 + 8 * u * p.Kpad + kt * BN;       // K index of tile kt = kt * 128 (tap-major)
            M[u] = __builtin_nontemporal_load((const f32x4*)(mg_ + gi));
            V[u] = __builtin_nontemporal_load((const f32x4*)(vg_ + gi));
            W[u] = __builtin_nontemporal_load((const f32x4*)(wg_ + gi));
        }
    };
    // request order = order of need (the load counter is in order): operand rows of tile 0, of tile 1, w/m/v of tiles 0 and 1
    load_b(0, 0, 0);
    if (fwd) load_x(0, 0, 0);
    load_wmv(0, am, av, aw);
    if (n_kt > 1) load_wmv(1, bm, bv, bw);
    advance();                            // (nkh, nkw, nci0) = tile 1
    f32x4 accf[3][2];
#pragma unroll
    for (int nb = 0; nb < 3; ++nb)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) accf[nb][cb] = zero4;
    const float step_size = p.hyper ? p.hyper[0] : p.step_size;
    const float inv_sqrt_bc2 = p.hyper ? p.hyper[1] : p.inv_sqrt_bc2;

    auto tile = [&](int kt, f32x4 (&cm)[4], f32x4 (&cv)[4], f32x4 (&cw)[4]) {
        const bool more = kt + 1 < n_kt;
        // ---- gradient tile: G[co][k] = sum_rows dY[row][co] * im2col[row][k]
#pragma unroll
        for (int j = 0; j < 6; ++j) *(f32x4*)(Bs + (brow + 8 * j) * BLD + bcol) = vb[j];
        __syncthreads();                              // (first tile: also the dY rows)
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
        // chunks of 8 instructions: their 16 fragment reads are issued together, no branch inside (a fully unrolled loop lets the
        // scheduler hoist all 2 NT reads and spill)
#pragma unroll 1
        for (int t0 = 0; t0 < NT; t0 += 8) {
            const float* ap = As + (2 * t0 + h) * BM + r;
            const float* bp = Bs + (2 * t0 + h) * BLD + wave * 32 + r;
#pragma unroll
            for (int t = 0; t < 8; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[2 * t * BM], bp[2 * t * BLD], acc, 0, 0, 0);
        }
        if (more) load_b(nkh, nkw, nci0);             // next tile's im2col rows (L2) under the epilogue
#pragma unroll
        for (int e = 0; e < 16; ++e) Gs[((e & 3) + 8 * (e >> 2) + 4 * h) * GLD + wave * 32 + r] = acc[e];
        __syncthreads();
        // ---- Adam on the tile (row-stream layout), updated weights back into the same LDS cells
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int gi = lo0 + 8 * u * p.Kpad + kt * BN;
            float* gcell = Gs + (rr + 8 * u) * GLD + 4 * q;
            const f32x4 ge = *(const f32x4*)gcell;
            if (FAST) mft_adam4_fast(cm[u], cv[u], cw[u], ge, p.b1, p.b2, p.eps, step_size, inv_sqrt_bc2);
            else mft_adam4_exact(cm[u], cv[u], cw[u], ge, p.b1, p.b2, p.eps, step_size, inv_sqrt_bc2);
            __builtin_nontemporal_store(cm[u], (f32x4*)(mg_ + gi));
            __builtin_nontemporal_store(cv[u], (f32x4*)(vg_ + gi));
            __builtin_nontemporal_store(cw[u], (f32x4*)(wg_ + gi));
            if (p.dw) *(f32x4*)(p.dw + tile_base + gi) = ge;
            if (fwd) *(f32x4*)gcell = cw[u];
        }
        if (fwd) {
            __syncthreads();                          // the updated tile is complete
            // ---- step t+1: out[px][co] += w'[co][k] * im2col(x_next)[px][k] over this wave's 32 k of the tile
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const f32x4 a0 = *(const f32x4*)(Gs + fm * GLD + 32 * wave + 16 * j + 4 * fq);
                const f32x4 a1 = *(const f32x4*)(Gs + (16 + fm) * GLD + 32 * wave + 16 * j + 4 * fq);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int nb = 0; nb < 3; ++nb) {
                        accf[nb][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[i], xb[nb][j][i], accf[nb][0], 0, 0, 0);
                        accf[nb][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[i], xb[nb][j][i], accf[nb][1], 0, 0, 0);
                    }
            }
            if (more) load_x(nkh, nkw, nci0);
        }
        // tile k+2's w/m/v into the set Adam has just finished with: requested AFTER everything tile k+1 needs first
        if (kt + 2 < n_kt) load_wmv(kt + 2, cm, cv, cw);
        advance();
        // (the next tile's first barrier separates these fragment reads of Gs from its next overwrite)
    };
    for (int kt = 0; kt < n_kt; kt += 2) {
        tile(kt, am, av, aw);
        if (kt + 1 < n_kt) tile(kt + 1, bm, bv, bw);
    }
    if (!fwd) return;

    // ---- the four k-slices' partial outputs, summed in fixed order
    __syncthreads();
    float* Red = smem;                    // [4][48][RLD], aliases As / Bs
#pragma unroll
    for (int nb = 0; nb < 3; ++nb)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
            *(f32x4*)(Red + ((wave * 48 + nb * 16 + fm) * RLD + cb * 16 + 4 * fq)) = accf[nb][cb];
    __syncthreads();
    const int c = tid >> 3, pl = tid & 7;             // channel co0 + c, pixels pl + 8 i
    float val[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int px = pl + 8 * i;
        val[i] = ((Red[(px)*RLD + c] + Red[(48 + px) * RLD + c]) + Red[(96 + px) * RLD + c]) + Red[(144 + px) * RLD + c];
    }
    float* T0 = Gs;                       // [48][TLD] raw output, [48][TLD] activation (Gs is dead: 2 x 6.3 KB <= 16.9 KB)
    float* T1 = Gs + 48 * TLD;
    const float inv_rows = 1.f / (float)rows;
    if (MODE == WF_RAW) {
#pragma unroll
        for (int i = 0; i < 6; ++i) T0[(pl + 8 * i) * TLD + c] = val[i];
    } else {
        auto stats = [&](const float* x, float& mu, float& rs) {
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < 6; ++i)
                if (pl + 8 * i < rows) s += x[i];
            mu = lane8_sum(s) * inv_rows;
            s = 0.f;
#pragma unroll
            for (int i = 0; i < 6; ++i)
                if (pl + 8 * i < rows) {
                    const float d = x[i] - mu;
                    s += d * d;
                }
            rs = 1.0f / sqrtf(lane8_sum(s) * inv_rows + p.bn_eps);
        };
        const int co = co0 + c;
        float mu, rs;
        stats(val, mu, rs);
        const float ga = p.gamma[g * p.gbs + co], be = p.beta[g * p.gbs + co];
        if (pl == 0) {
            p.mean[(long long)g * p.Cout + co] = mu;
            p.rstd[(long long)g * p.Cout + co] = rs;
        }
        if (MODE == WF_ENTRY) {
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                T0[(pl + 8 * i) * TLD + c] = val[i];
                T1[(pl + 8 * i) * TLD + c] = fmaxf((val[i] - mu) * rs * ga + be, 0.f);
            }
        } else {
            float sv[6];
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                const int px = pl + 8 * i;
                sv[i] = px < rows ? p.sc[(row0 + px) * p.Cout + co] : 0.f;
            }
            float mus, rss;
            stats(sv, mus, rss);
            const float gas = p.gs[g * p.gbs + co], bes = p.bs[g * p.gbs + co];
            if (pl == 0) {
                p.means[(long long)g * p.Cout + co] = mus;
                p.rstds[(long long)g * p.Cout + co] = rss;
            }
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                float o = (val[i] - mu) * rs * ga + be;
                o += (sv[i] - mus) * rss * gas + bes;
                T0[(pl + 8 * i) * TLD + c] = val[i];
                T1[(pl + 8 * i) * TLD + c] = fmaxf(o, 0.f);
            }
        }
    }
    __syncthreads();
    // write-out: one pixel row of the 32 channels = one 128-byte line per half wave
    const int c2 = tid & 31, p2 = tid >> 5;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int px = p2 + 8 * i;
        if (px < rows) {
            p.raw[(row0 + px) * p.Cout + co0 + c2] = T0[px * TLD + c2];
            if (MODE != WF_RAW) p.act[(row0 + px) * p.Cout + co0 + c2] = T1[px * TLD + c2];
        }
    }
    if (MODE == WF_EXIT) {
        const int n_img = rows / p.hw;
        if (p2 < n_img) {                              // global average pool of the block output (<= 8 images per episode)
            float sum = 0.f;
            for (int k2 = 0; k2 < p.hw; ++k2) sum += T1[(p2 * p.hw + k2) * TLD + c2];
            p.pooled[((long long)g * n_img + p2) * p.Cout + co0 + c2] = sum * (1.f / (float)p.hw);
        }
    }
}

int g_wf_exact = 0;      // 1: correctly rounded division / square root in the Adam epilogue (mft_wgrad_fwd_set_exact)
int g_wf_xcd = 1;        // 1: an episode's workgroups share one XCD's L2 (mft_wgrad_fwd_set_xcd; 0 = natural order, for A/B)

}  // namespace

extern "C" void mft_wgrad_fwd_set_exact(int on) { g_wf_exact = on ? 1 : 0; }
extern "C" void mft_wgrad_fwd_set_xcd(int on) { g_wf_xcd = on ? 1 : 0; }

extern "C" int mft_wgrad_adam_next_forward(const float* x, int ldx, const float* dy, int ldy, float* w, float* m, float* v,
                                           float* dw_or_null, int n_img, int H, int W, int Cin, int Cout, int KH, int KW,
                                           int stride, int pad, int imgs_per_group, long long group_stride, int step,
                                           const float* hyper, float lr, float beta1, float beta2, float eps,
                                           const float* x_next, int mode, float* raw, float* act, const float* gamma,
                                           const float* beta, long long gb_group_stride, float* mean, float* rstd,
                                           const float* sc_raw, const float* gamma_s, const float* beta_s, float* mean_s,
                                           float* rstd_s, float* pooled, float bn_eps, void* stream) {
    if (n_img <= 0 || imgs_per_group <= 0 || n_img % imgs_per_group != 0) return MFT_EINVAL;
    if (Cin % 128 != 0 || Cout % 32 != 0 || ldx % 4 != 0 || ldy % 4 != 0) return MFT_EINVAL;
    if (hyper == nullptr && step < 1) return MFT_EINVAL;
    const int OH = (H + 2 * pad - KH) / stride + 1, OW = (W + 2 * pad - KW) / stride + 1;
    const int rows = imgs_per_group * OH * OW;
    if (rows > 48 || OH < 1 || OW < 1 || H > 100 || W > 100 || imgs_per_group > 8) return MFT_EINVAL;
    if (mode < WF_RAW || mode > WF_EXIT) return MFT_EINVAL;
    if (x_next != nullptr) {
        if (raw == nullptr) return MFT_EINVAL;
        if (mode != WF_RAW && (act == nullptr || gamma == nullptr || beta == nullptr || mean == nullptr || rstd == nullptr)) return MFT_EINVAL;
        if (mode == WF_EXIT && (sc_raw == nullptr || gamma_s == nullptr || beta_s == nullptr || mean_s == nullptr || rstd_s == nullptr ||
                                pooled == nullptr)) return MFT_EINVAL;
    }
    WfArgs p = {};
    p.in = x; p.dy = dy; p.dw = dw_or_null; p.w = w; p.m = m; p.v = v;
    p.ldi = ldx; p.ldy = ldy;
    p.H = H; p.W = W; p.Cin = Cin; p.OH = OH; p.OW = OW; p.Cout = Cout; p.KH = KH; p.KW = KW; p.stride = stride; p.pad = pad;
    p.Kpad = KH * KW * Cin;
    p.rows = rows; p.ipg = imgs_per_group;
    p.tiles_ci = Cin / 128;
    p.inv_ohw = (65536 + OH * OW - 1) / (OH * OW);
    p.inv_ow = (65536 + OW - 1) / OW;
    p.mma_rows = rows;
    p.dwgs = group_stride;
    if (hyper != nullptr) step = 1;
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    p.step_size = (float)((double)lr / bc1);
    p.inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
    p.b1 = beta1; p.b2 = beta2; p.eps = eps;
    p.hyper = hyper;
    p.xn = x_next; p.raw = raw; p.act = act; p.gamma = gamma; p.beta = beta; p.gbs = gb_group_stride;
    p.mean = mean; p.rstd = rstd; p.sc = sc_raw; p.gs = gamma_s; p.bs = beta_s; p.means = mean_s; p.rstds = rstd_s;
    p.pooled = pooled; p.hw = OH * OW; p.bn_eps = bn_eps;
    const int groups = n_img / imgs_per_group;
    constexpr int lds = (48 * 32 + 48 * (128 + 32) + 32 * (128 + 4)) * 4;          // 53.8 KB: two workgroups per CU
    p.co_blocks = Cout / 32;
    p.xcd_groups = (g_wf_xcd && groups % 8 == 0) ? 1 : 0;
    const dim3 grid((unsigned)(p.co_blocks * groups), 1, 1), block(256);
    hipStream_t s = (hipStream_t)stream;
    // matrix instructions of the reduction: 2 rows each, in chunks of 8 (<= 32 / <= 48 rows; rows beyond the episode's are zeros)
    const int nt = rows <= 32 ? 16 : 24;
#define WF_LAUNCH2(MODE_, NT_)                                                                            \
    if (g_wf_exact) hipLaunchKernelGGL((wgrad_adam_fwd_kernel<MODE_, false, NT_>), grid, block, lds, s, p);  \
    else hipLaunchKernelGGL((wgrad_adam_fwd_kernel<MODE_, true, NT_>), grid, block, lds, s, p);
#define WF_LAUNCH(MODE_)                                  \
    if (nt == 16) { WF_LAUNCH2(MODE_, 16) }               \
    else { WF_LAUNCH2(MODE_, 24) }
    if (mode == WF_RAW) { WF_LAUNCH(WF_RAW) }
    else if (mode == WF_ENTRY) { WF_LAUNCH(WF_ENTRY) }
    else { WF_LAUNCH(WF_EXIT) }
#undef WF_LAUNCH
#undef WF_LAUNCH2
    return mft_launch_status();
}
