// Backward-only kernels of the meta-training path (loss.backward() in MetaTemplate.train_loop*, meta_template.py:76-109):
// multi-workgroup BatchNorm backward with fused activation derivative, max-pool with saved argmax and its backward,
// column sums (bias gradients), and the GNN head's backward glue (masked-softmax, |x_i-x_j|, graph aggregation, node
// assembly, score gather).  All HBM-bound; reductions are fixed-order (bit-reproducible).
#include "mft_common.h"
#include <type_traits>

namespace {

// (branch-free: `act` is uniform, so the two selects are scalar; with branches per element the loops that call this waited for every
// load before issuing the next one)
__device__ __forceinline__ float act_grad(float y, int act, float slope) {
    const float neg = act == MFT_ACT_RELU ? 0.f : (act == MFT_ACT_LRELU ? slope : 1.f);
    return y > 0.f ? 1.f : neg;
}

// ---------------------------------------------------------------------------------- BN backward, two phase
struct BnBwdArgs {
    const float* x; const float* dy; const float* y_act; float* dx;
    int ldx, lddy, ldya, lddx, C, rows_per_group, rows_per_chunk, chunks;
    const float* mean; const float* rstd; const float* gamma; long long gbs;
    float* dgamma; float* dbeta; float* ws; int act; float slope;
    // meta-training extras (all nullable): the gradients summed over the groups in group order ([C] each: k episodes in lockstep
    // share one set of affine parameters), and the gradient of a bias added in FRONT of this BatchNorm, which is identically zero
    // (a per-channel shift of the input leaves a train-mode BatchNorm's output unchanged): written as exact zeros
    float* dgamma_sum; float* dbeta_sum; float* dbias_zero; int n_groups;
};

// phase 1: per (chunk, 64-channel tile, group) partial sums of dy_eff and dy_eff * xhat
__device__ __forceinline__ void bn_bwd_partial_body(const BnBwdArgs& p, const int bx, const int by, const int bz) {
    const int cq = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int c = by * 64 + cq * 4;
    const int g = bz, chunk = bx;
    const long long row0 = (long long)g * p.rows_per_group;
    __shared__ f32x4 red1[16][16];
    __shared__ f32x4 red2[16][16];
    f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = s1;
    if (c < p.C) {
        const f32x4 mu = *(const f32x4*)(p.mean + (long long)g * p.C + c);
        const f32x4 rs = *(const f32x4*)(p.rstd + (long long)g * p.C + c);
        const int rbeg = chunk * p.rows_per_chunk, rend = min(rbeg + p.rows_per_chunk, p.rows_per_group);
        // (the y_act test is hoisted out of the row walk: with it inside, every unrolled row waited for its own loads)
        auto walk = [&](auto with_act) {
#pragma unroll 4
            for (int rr = rbeg + rl; rr < rend; rr += 16) {           // (the loads of four rows in flight; same order of additions)
                f32x4 d = *(const f32x4*)(p.dy + (row0 + rr) * p.lddy + c);
                if constexpr (decltype(with_act)::value) {
                    const f32x4 ya = *(const f32x4*)(p.y_act + (row0 + rr) * p.ldya + c);
#pragma unroll
                    for (int e = 0; e < 4; ++e) d[e] *= act_grad(ya[e], p.act, p.slope);
                }
                const f32x4 xh = (*(const f32x4*)(p.x + (row0 + rr) * p.ldx + c) - mu) * rs;
                s1 += d;
                s2 += d * xh;
            }
        };
        if (p.y_act) walk(std::true_type{}); else walk(std::false_type{});
    }
    red1[rl][cq] = s1;
    red2[rl][cq] = s2;
    __syncthreads();
    if (rl == 0 && c < p.C) {
#pragma unroll
        for (int k = 1; k < 16; ++k) { s1 += red1[k][cq]; s2 += red2[k][cq]; }
        float* o = p.ws + (((long long)g * p.chunks + chunk) * p.C + c) * 2;
#pragma unroll
        for (int e = 0; e < 4; ++e) { o[2 * e] = s1[e]; o[2 * e + 1] = s2[e]; }
    }
}

__global__ __launch_bounds__(256) void bn_bwd_partial_kernel(BnBwdArgs p) { bn_bwd_partial_body(p, blockIdx.x, blockIdx.y, blockIdx.z); }

// phase 2: fixed-order sum of the partials -> dgamma, dbeta and the two group means used by phase 3
// 256 threads = 16 channels x 16 chunk lanes; each lane sums every 16th partial, then a fixed xor tree over the 16 lanes
// (a serial loop over up to ~340 partials per channel cost 50-250 us per BatchNorm in the meta-training step)
// (LPC = 64: one group with >= 128 chunks -- a whole wave per channel, as in bn_stats_finalize)
template <int LPC>
__device__ __forceinline__ void bn_bwd_finalize_body(const BnBwdArgs& p, float* sums, const int bx, const int by) {
    const int kl = threadIdx.x % LPC;
    const int c = bx * (256 / LPC) + threadIdx.x / LPC;
    // one block row per group -- or, when the sums over the groups are wanted, ONE block row that walks the groups in order
    const bool walk = p.dgamma_sum != nullptr || p.dbeta_sum != nullptr;
    const int g0 = walk ? 0 : by, g1 = walk ? p.n_groups : g0 + 1;
    float t1 = 0.f, t2 = 0.f;
    for (int g = g0; g < g1; ++g) {
        float s1 = 0.f, s2 = 0.f;
        if (c < p.C)
#pragma unroll 4
            for (int k = kl; k < p.chunks; k += LPC) {
                const float* o = p.ws + (((long long)g * p.chunks + k) * p.C + c) * 2;
                s1 += o[0];
                s2 += o[1];
            }
#pragma unroll
        for (int off = LPC / 2; off > 0; off >>= 1) {
            s1 += __shfl_xor(s1, off, 64);
            s2 += __shfl_xor(s2, off, 64);
        }
        if (c >= p.C || kl != 0) continue;
        if (p.dbeta) p.dbeta[(long long)g * p.C + c] = s1;
        if (p.dgamma) p.dgamma[(long long)g * p.C + c] = s2;
        sums[((long long)g * p.C + c) * 2] = s1 / (float)p.rows_per_group;
        sums[((long long)g * p.C + c) * 2 + 1] = s2 / (float)p.rows_per_group;
        t1 += s1;
        t2 += s2;
    }
    if (c >= p.C || kl != 0) return;
    if (p.dbeta_sum) p.dbeta_sum[c] = t1;
    if (p.dgamma_sum) p.dgamma_sum[c] = t2;
    if (p.dbias_zero && g0 == 0) p.dbias_zero[c] = 0.f;
}

template <int LPC>
__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(BnBwdArgs p, float* sums) {
    bn_bwd_finalize_body<LPC>(p, sums, blockIdx.x, blockIdx.y);
}

// phase 3: dx = gamma * rstd * (dy_eff - mean(dy_eff) - xhat * mean(dy_eff * xhat))
__device__ __forceinline__ void bn_bwd_apply_body(const BnBwdArgs& p, const float* sums, int n_groups, const int block, const int n_blocks) {
    const int cq = p.C >> 2;
    const long long total = (long long)n_groups * p.rows_per_group * cq;
    for (long long i = (long long)block * blockDim.x + threadIdx.x; i < total;
         i += (long long)n_blocks * blockDim.x) {
        const long long row = i / cq;
        const int c = (int)(i - row * cq) * 4;
        const int g = (int)(row / p.rows_per_group);
        f32x4 d = *(const f32x4*)(p.dy + row * p.lddy + c);
        if (p.y_act) {
            const f32x4 ya = *(const f32x4*)(p.y_act + row * p.ldya + c);
#pragma unroll
            for (int e = 0; e < 4; ++e) d[e] *= act_grad(ya[e], p.act, p.slope);
        }
        const f32x4 mu = *(const f32x4*)(p.mean + (long long)g * p.C + c);
        const f32x4 rs = *(const f32x4*)(p.rstd + (long long)g * p.C + c);
        const f32x4 ga = *(const f32x4*)(p.gamma + g * p.gbs + c);
        const f32x4 xh = (*(const f32x4*)(p.x + row * p.ldx + c) - mu) * rs;
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float m1 = sums[((long long)g * p.C + c + e) * 2], m2 = sums[((long long)g * p.C + c + e) * 2 + 1];
            o[e] = ga[e] * rs[e] * (d[e] - m1 - xh[e] * m2);
        }
        *(f32x4*)(p.dx + row * p.lddx + c) = o;
    }
}

__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(BnBwdArgs p, const float* sums, int n_groups) {
    bn_bwd_apply_body(p, sums, n_groups, blockIdx.x, gridDim.x);
}

// Several independent BatchNorm backward problems in one launch triple (mft_bn_backward_act_multi): the backward of SimpleBlock's
// BN2 and BNshortcut reads the same upstream gradient and ReLU output (backbone.py:256-260) -- three launches for both instead of
// six.  Block -> job by start tables; inside a job the numbering of its own launches: bit-identical results.
constexpr int BB_MULTI = 8;
struct BnBwdMultiArgs {
    BnBwdArgs job[BB_MULTI];
    float* sums[BB_MULTI];
    int lpc64[BB_MULTI];
    int start_p[BB_MULTI + 1], start_f[BB_MULTI + 1], start_a[BB_MULTI + 1];
    int n;
};
static_assert(sizeof(BnBwdMultiArgs) <= 4000, "kernarg segment");

__global__ __launch_bounds__(256) void bn_bwd_partial_multi_kernel(BnBwdMultiArgs a) {
    int j = 0;
#pragma unroll 1
    while (j + 1 < a.n && (int)blockIdx.x >= a.start_p[j + 1]) ++j;
    const BnBwdArgs p = a.job[j];
    const int local = blockIdx.x - a.start_p[j], cy = (p.C + 63) / 64;
    bn_bwd_partial_body(p, local % p.chunks, (local / p.chunks) % cy, local / (p.chunks * cy));
}

__global__ __launch_bounds__(256) void bn_bwd_finalize_multi_kernel(BnBwdMultiArgs a) {
    int j = 0;
#pragma unroll 1
    while (j + 1 < a.n && (int)blockIdx.x >= a.start_f[j + 1]) ++j;
    const BnBwdArgs p = a.job[j];
    const int local = blockIdx.x - a.start_f[j];
    if (a.lpc64[j]) {
        bn_bwd_finalize_body<64>(p, a.sums[j], local, 0);
    } else {
        const int fx = (p.C + 15) / 16;
        bn_bwd_finalize_body<16>(p, a.sums[j], local % fx, local / fx);
    }
}

__global__ __launch_bounds__(256) void bn_bwd_apply_multi_kernel(BnBwdMultiArgs a) {
    int j = 0;
#pragma unroll 1
    while (j + 1 < a.n && (int)blockIdx.x >= a.start_a[j + 1]) ++j;
    const BnBwdArgs p = a.job[j];
    bn_bwd_apply_body(p, a.sums[j], p.n_groups, blockIdx.x - a.start_a[j], a.start_a[j + 1] - a.start_a[j]);
}

// ---------------------------------------------------------------------------------- elementwise helpers
// dx = dy * act'(y) (+ optionally accumulate into dx)
__global__ __launch_bounds__(256) void act_bwd_kernel(const float* __restrict__ dy, int lddy, const float* __restrict__ y,
                                                      int ldy, float* __restrict__ dx, int lddx, int C, long long rows,
                                                      int act, float slope, int accumulate) {
    const long long total = rows * C;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / C;
        const int c = (int)(i - r * C);
        const float v = dy[r * lddy + c] * act_grad(y[r * ldy + c], act, slope);
        if (accumulate) dx[r * lddx + c] += v;
        else dx[r * lddx + c] = v;
    }
}

// out[c] = sum_r x[r][c], two phase (partials [chunks][C] then fixed-order sum).  V4: float4 loads, 16 row lanes x 16 channel quads
// per block (a quarter of the serial iterations of the scalar form); fixed reduction order either way.
template <bool V4>
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ x, int ldx, int C, long long rows,
                                                             int rows_per_chunk, float* __restrict__ ws) {
    const long long rbeg = (long long)blockIdx.x * rows_per_chunk;
    const long long rend = min(rbeg + rows_per_chunk, rows);
    if (V4) {
        const int cq = threadIdx.x & 15, rl = threadIdx.x >> 4;
        const int c = blockIdx.y * 64 + 4 * cq;
        __shared__ f32x4 red4[16][16];
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        if (c < C)
            for (long long r = rbeg + rl; r < rend; r += 16) s += *(const f32x4*)(x + r * ldx + c);
        red4[rl][cq] = s;
        __syncthreads();
        if (rl == 0 && c < C) {
            f32x4 t = red4[0][cq];
#pragma unroll
            for (int l = 1; l < 16; ++l) t += red4[l][cq];
            *(f32x4*)(ws + (long long)blockIdx.x * C + c) = t;
        }
    } else {
        const int c = blockIdx.y * 64 + (threadIdx.x & 63);
        const int rl = threadIdx.x >> 6;
        __shared__ float red[4][64];
        float s = 0.f;
        if (c < C)
            for (long long r = rbeg + rl; r < rend; r += 4) s += x[r * ldx + c];
        red[rl][threadIdx.x & 63] = s;
        __syncthreads();
        if (rl == 0 && c < C) ws[(long long)blockIdx.x * C + c] = red[0][c & 63] + red[1][c & 63] + red[2][c & 63] + red[3][c & 63];
    }
}

__global__ __launch_bounds__(256) void colsum_final_kernel(const float* __restrict__ ws, int C, int chunks,
                                                           float* __restrict__ out) {
    const int kl = threadIdx.x & 15;
    const int c = blockIdx.x * 16 + (threadIdx.x >> 4);
    float s = 0.f;
    if (c < C)
        for (int k = kl; k < chunks; k += 16) s += ws[(long long)k * C + c];
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    if (c < C && kl == 0) out[c] = s;
}

// ---------------------------------------------------------------------------------- BN backward, SMALL problems: one launch
// The three-phase form above needs two grid-wide hand-offs (partial sums -> sums -> dx): three dependent launches, >= 4.7 us each in a
// replayed meta-training step whatever they do -- and the deep layers of a single 105-image episode (trunk.6: 3,780 rows, trunk.7:
// 945 rows) and the head's BatchNorm1d layers (105 / 480 rows) are a few hundred KB.  Here one workgroup owns FOUR CHANNELS over
// ALL rows of all groups: 256 row lanes, two passes over its 16-byte column (the second out of cache), the sums reduced inside the
// workgroup (wave xor tree, then the four waves in order: fixed) -- no hand-off between workgroups, one launch.  Groups are walked in
// order by the same workgroup (the sums over the groups are formed in group order, as the finalize launch above does).  Several jobs
// per launch (grid.y = job).  Same formulas as the three-phase form; the sums are taken in another (fixed) order.
constexpr int BNS_JOBS = 8;
struct BnBwdSmallArgs {
    BnBwdArgs job[BNS_JOBS];
    int n;
};
static_assert(sizeof(BnBwdSmallArgs) <= 4000, "kernarg segment");

__device__ __forceinline__ void block_sum2_f4(f32x4& a, f32x4& b, float (*red)[4][8]) {       // red[2 buffers][4 waves][8]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            a[e] += __shfl_xor(a[e], off, 64);
            b[e] += __shfl_xor(b[e], off, 64);
        }
    if (lane == 0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { (*red)[wave][e] = a[e]; (*red)[wave][4 + e] = b[e]; }
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        a[e] = (((*red)[0][e] + (*red)[1][e]) + (*red)[2][e]) + (*red)[3][e];
        b[e] = (((*red)[0][4 + e] + (*red)[1][4 + e]) + (*red)[2][4 + e]) + (*red)[3][4 + e];
    }
}

__global__ __launch_bounds__(256) void bn_bwd_small_kernel(BnBwdSmallArgs a) {
    __shared__ float red[2][4][8];
    const BnBwdArgs p = a.job[blockIdx.y];
    const int c = blockIdx.x * 4;
    if (c >= p.C) return;
    const int t = threadIdx.x;
    const f32x4 ga = *(const f32x4*)(p.gamma + c);
    f32x4 t1 = {0.f, 0.f, 0.f, 0.f}, t2 = t1;
    for (int g = 0; g < p.n_groups; ++g) {
        const long long row0 = (long long)g * p.rows_per_group;
        const f32x4 mu = *(const f32x4*)(p.mean + (long long)g * p.C + c), rs = *(const f32x4*)(p.rstd + (long long)g * p.C + c);
        f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = s1;
        auto walk = [&](auto with_act) {
#pragma unroll 4
            for (int r = t; r < p.rows_per_group; r += 256) {
                f32x4 d = *(const f32x4*)(p.dy + (row0 + r) * p.lddy + c);
                if constexpr (decltype(with_act)::value) {
                    const f32x4 ya = *(const f32x4*)(p.y_act + (row0 + r) * p.ldya + c);
#pragma unroll
                    for (int e = 0; e < 4; ++e) d[e] *= act_grad(ya[e], p.act, p.slope);
                }
                const f32x4 xh = (*(const f32x4*)(p.x + (row0 + r) * p.ldx + c) - mu) * rs;
                s1 += d;
                s2 += d * xh;
            }
        };
        if (p.y_act) walk(std::true_type{}); else walk(std::false_type{});
        block_sum2_f4(s1, s2, &red[g & 1]);            // (two buffers: the next group's writes cannot overtake this group's reads)
        if (t == 0) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (p.dbeta) p.dbeta[(long long)g * p.C + c + e] = s1[e];
                if (p.dgamma) p.dgamma[(long long)g * p.C + c + e] = s2[e];
            }
        }
        t1 += s1;
        t2 += s2;
        if (p.dx) {
            f32x4 m1, m2;                                           // (divisions, as the three-phase form's finalize: s / rows)
#pragma unroll
            for (int e = 0; e < 4; ++e) { m1[e] = s1[e] / (float)p.rows_per_group; m2[e] = s2[e] / (float)p.rows_per_group; }
            auto apply = [&](auto with_act) {
#pragma unroll 4
                for (int r = t; r < p.rows_per_group; r += 256) {
                    f32x4 d = *(const f32x4*)(p.dy + (row0 + r) * p.lddy + c);
                    if constexpr (decltype(with_act)::value) {
                        const f32x4 ya = *(const f32x4*)(p.y_act + (row0 + r) * p.ldya + c);
#pragma unroll
                        for (int e = 0; e < 4; ++e) d[e] *= act_grad(ya[e], p.act, p.slope);
                    }
                    const f32x4 xh = (*(const f32x4*)(p.x + (row0 + r) * p.ldx + c) - mu) * rs;
                    f32x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = ga[e] * rs[e] * (d[e] - m1[e] - xh[e] * m2[e]);
                    *(f32x4*)(p.dx + (row0 + r) * p.lddx + c) = o;
                }
            };
            if (p.y_act) apply(std::true_type{}); else apply(std::false_type{});
        }
    }
    if (t == 0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (p.dbeta_sum) p.dbeta_sum[c + e] = t1[e];
            if (p.dgamma_sum) p.dgamma_sum[c + e] = t2[e];
            if (p.dbias_zero) p.dbias_zero[c + e] = 0.f;
        }
    }
}

// rows per group up to which BatchNorm backward runs as the one-launch form (0: never).  512 = the head's BatchNorm1d layers (105 / 480
// rows: 5-8 us against three launches of ~5 us).  Measured at the trunk's deep layers the form LOSES: a workgroup's 16-byte column is an
// eighth of every cache line it touches and 64-128 workgroups walk 945-3,780 rows each -- trunk.7 22-25 us, trunk.6 63 us against ~17.
int g_bn_small_rows = 512;

// ---------------------------------------------------------------------------------- max pool with argmax
__global__ __launch_bounds__(256) void bn_relu_maxpool_arg_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                                  unsigned char* __restrict__ arg, int n_img, int H, int W,
                                                                  int C, int OH, int OW, int imgs_per_group,
                                                                  const float* __restrict__ mean,
                                                                  const float* __restrict__ rstd,
                                                                  const float* __restrict__ gamma,
                                                                  const float* __restrict__ beta) {
    const long long total = (long long)n_img * OH * OW * C;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        long long t = i / C;
        const int ow = (int)(t % OW); t /= OW;
        const int oh = (int)(t % OH);
        const int n = (int)(t / OH);
        const int g = n / imgs_per_group;
        const float mu = mean[(long long)g * C + c], rs = rstd[(long long)g * C + c], ga = gamma[c], be = beta[c];
        float best = -3.4e38f;
        int barg = 0;
        for (int dh = 0; dh < 3; ++dh) {
            const int ih = oh * 2 - 1 + dh;
            if (ih < 0 || ih >= H) continue;
            for (int dw = 0; dw < 3; ++dw) {
                const int iw = ow * 2 - 1 + dw;
                if (iw < 0 || iw >= W) continue;
                const float v = fmaxf((x[(((long long)n * H + ih) * W + iw) * C + c] - mu) * rs * ga + be, 0.f);
                if (v > best) { best = v; barg = dh * 3 + dw; }     // first maximum wins, like ATen's max_pool2d
            }
        }
        y[i] = best;
        arg[i] = (unsigned char)barg;
    }
}

// The same for C % 4 == 0, four channels per thread: the nine taps are loaded unconditionally (coordinates clamped, a tap outside the
// image masked to -inf afterwards), so all nine 16-byte loads are in flight together -- the scalar form above walks its taps through
// branches, one exposed load latency each (31 us for the 105-image stem output; this form: HBM-bound).  Same arithmetic per element,
// same tie rule (first maximum in (dh, dw) order).
__global__ __launch_bounds__(256) void bn_relu_maxpool_arg4_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                                   unsigned char* __restrict__ arg, int n_img, int H, int W,
                                                                   int C, int OH, int OW, int imgs_per_group,
                                                                   const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                   const float* __restrict__ gamma, const float* __restrict__ beta) {
    const int cq = C >> 2;
    const long long total = (long long)n_img * OH * OW * cq;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % cq) * 4;
        long long t = i / cq;
        const int ow = (int)(t % OW); t /= OW;
        const int oh = (int)(t % OH);
        const int n = (int)(t / OH);
        const int g = n / imgs_per_group;
        const f32x4 mu = *(const f32x4*)(mean + (long long)g * C + c), rs = *(const f32x4*)(rstd + (long long)g * C + c);
        const f32x4 ga = *(const f32x4*)(gamma + c), be = *(const f32x4*)(beta + c);
        f32x4 xv[9];
        bool ok[9];
#pragma unroll
        for (int dh = 0; dh < 3; ++dh)
#pragma unroll
            for (int dw = 0; dw < 3; ++dw) {
                const int ih = oh * 2 - 1 + dh, iw = ow * 2 - 1 + dw;
                ok[dh * 3 + dw] = ih >= 0 && ih < H && iw >= 0 && iw < W;
                const int ihc = ih < 0 ? 0 : (ih >= H ? H - 1 : ih), iwc = iw < 0 ? 0 : (iw >= W ? W - 1 : iw);
                xv[dh * 3 + dw] = *(const f32x4*)(x + (((long long)n * H + ihc) * W + iwc) * C + c);
            }
        f32x4 best = {-3.4e38f, -3.4e38f, -3.4e38f, -3.4e38f};
        unsigned barg = 0;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float v = fmaxf((xv[k][e] - mu[e]) * rs[e] * ga[e] + be[e], 0.f);
                if (ok[k] && v > best[e]) { best[e] = v; barg = (barg & ~(0xffu << (8 * e))) | ((unsigned)k << (8 * e)); }
            }
        }
        *(f32x4*)(y + i * 4) = best;
        *(unsigned*)(arg + i * 4) = barg;
    }
}

// gradient w.r.t. the BN output (pre-ReLU): each input pixel gathers from the <= 4 windows that contain it
__global__ __launch_bounds__(256) void maxpool_relu_bwd_kernel(const float* __restrict__ dy,
                                                               const unsigned char* __restrict__ arg,
                                                               const float* __restrict__ y, float* __restrict__ dx,
                                                               int n_img, int H, int W, int C, int OH, int OW) {
    const long long total = (long long)n_img * H * W * C;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        long long t = i / C;
        const int iw = (int)(t % W); t /= W;
        const int ih = (int)(t % H);
        const int n = (int)(t / H);
        float s = 0.f;
        for (int oh = (ih - 1 + 1) / 2; oh <= (ih + 1) / 2; ++oh) {           // windows with oh*2-1 <= ih <= oh*2+1
            if (oh < 0 || oh >= OH) continue;
            const int dh = ih - (oh * 2 - 1);
            if (dh < 0 || dh > 2) continue;
            for (int ow = iw / 2; ow <= (iw + 1) / 2; ++ow) {
                if (ow < 0 || ow >= OW) continue;
                const int dw = iw - (ow * 2 - 1);
                if (dw < 0 || dw > 2) continue;
                const long long o = (((long long)n * OH + oh) * OW + ow) * C + c;
                if (arg[o] == dh * 3 + dw && y[o] > 0.f) s += dy[o];          // y > 0: ReLU derivative
            }
        }
        dx[i] = s;
    }
}

// C % 4 == 0: one thread per 2 x 2 block of input pixels (ih = 2a, 2a+1; iw = 2b, 2b+1) and four channels: the block lies in the windows
// (oh, ow) in {a, a+1} x {b, b+1} only, so four window loads (argmax, y, dy) serve four input pixels -- a per-pixel form loads
// up to four windows for every pixel (427 MB of cache traffic for the 105-image stem output: 27 us against 19).  Per pixel the contributions are added
// in the same (oh, ow)-ascending order: identical results.
__global__ __launch_bounds__(256) void maxpool_relu_bwd4_2x2_kernel(const float* __restrict__ dy, const unsigned char* __restrict__ arg,
                                                                    const float* __restrict__ y, float* __restrict__ dx, int n_img,
                                                                    int H, int W, int C, int OH, int OW) {
    const int cq = C >> 2, HB = (H + 1) >> 1, WB = (W + 1) >> 1;
    const int total = n_img * HB * WB * cq;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int c = (i % cq) * 4;
        int t = i / cq;
        const int wb = t % WB; t /= WB;
        const int hb = t % HB;
        const int n = t / HB;
        unsigned a4[4];
        f32x4 yv[4], dv[4];
        bool wok[4];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int oh = hb + a, ow = wb + b, k = a * 2 + b;
                wok[k] = oh < OH && ow < OW;
                const long long o = (((long long)n * OH + (oh < OH ? oh : OH - 1)) * OW + (ow < OW ? ow : OW - 1)) * C + c;
                a4[k] = *(const unsigned*)(arg + o);
                yv[k] = *(const f32x4*)(y + o);
                dv[k] = *(const f32x4*)(dy + o);
            }
#pragma unroll
        for (int pa = 0; pa < 2; ++pa)
#pragma unroll
            for (int pb = 0; pb < 2; ++pb) {
                const int ih = 2 * hb + pa, iw = 2 * wb + pb;
                if (ih >= H || iw >= W) continue;
                f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int a = 0; a <= pa; ++a)                       // pixel row 2a: window a only; row 2a+1: windows a and a+1
#pragma unroll
                    for (int b = 0; b <= pb; ++b) {
                        const int k = a * 2 + b;
                        const unsigned want = (unsigned)((ih - ((hb + a) * 2 - 1)) * 3 + (iw - ((wb + b) * 2 - 1)));
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (wok[k] && ((a4[k] >> (8 * e)) & 0xffu) == want && yv[k][e] > 0.f) s[e] += dv[k][e];
                    }
                *(f32x4*)(dx + ((((long long)n * H + ih) * W + iw) * C + c)) = s;
            }
    }
}

// ---------------------------------------------------------------------------------- GNN head backward glue
// A = softmax_j(s - 1e8*[i==j]);  ds[b,i,j] = A * (dA - sum_k dA*A)
__global__ __launch_bounds__(256) void masked_softmax_bwd_kernel(const float* __restrict__ A, const float* __restrict__ dA,
                                                                 float* __restrict__ ds, int ldds, int n_graphs, int N) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= (long long)n_graphs * N) return;
    const float* a = A + row * N;
    const float* d = dA + row * N;
    float dot = 0.f;
    for (int j = lane; j < N; j += 64) dot += a[j] * d[j];
    dot = wave_sum(dot);
    for (int j = lane; j < N; j += 64) ds[(row * N + j) * ldds] = a[j] * (d[j] - dot);
}

// d[b,i,j,f] = |x_i - x_j| : dx[b,i,f] = sum_j sign(x_i-x_j) * (dd[b,i,j,f] + dd[b,j,i,f])  (accumulated into dx)
__global__ __launch_bounds__(256) void pair_absdiff_bwd_kernel(const float* __restrict__ x, int ldx,
                                                               const float* __restrict__ dd, int ldd,
                                                               float* __restrict__ dx, int lddx, int n_graphs, int N,
                                                               int F) {
    const long long total = (long long)n_graphs * N * F;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int f = (int)(i % F);
        const long long bi = i / F;
        const int ii = (int)(bi % N);
        const long long b = bi / N;
        const float xi = x[bi * ldx + f];
        float s = 0.f;
        for (int j = 0; j < N; ++j) {
            const float df = xi - x[(b * N + j) * ldx + f];
            const float sg = df > 0.f ? 1.f : (df < 0.f ? -1.f : 0.f);
            s += sg * (dd[((b * N + ii) * N + j) * ldd + f] + dd[((b * N + j) * N + ii) * ldd + f]);
        }
        dx[bi * lddx + f] += s;
    }
}

// y = [x | A x]:  dx[b,i,f] += dy[b,i,f] + sum_j A[b,j,i] * dy[b,j,F+f];   dA[b,i,j] = sum_f dy[b,i,F+f] * x[b,j,f]
__global__ __launch_bounds__(256) void graph_aggregate_bwd_kernel(const float* __restrict__ A,
                                                                  const float* __restrict__ x, int ldx,
                                                                  const float* __restrict__ dy, int lddy,
                                                                  float* __restrict__ dx, int lddx,
                                                                  float* __restrict__ dA, int n_graphs, int N, int F,
                                                                  int accumulate) {
    const long long row = blockIdx.x;     // (b, i)
    const long long b = row / N;
    const int i = (int)(row % N);
    for (int f = threadIdx.x; f < F; f += blockDim.x) {
        float s = dy[row * lddy + f];
#pragma unroll 10
        for (int j = 0; j < N; ++j) s += A[(b * N + j) * N + i] * dy[(b * N + j) * lddy + F + f];      // (ten loads in flight; same order)
        dx[row * lddx + f] = accumulate ? dx[row * lddx + f] + s : s;
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (F <= 256) {
        // dy's row is the same for every j: kept in registers; four j per wave and round -- sixteen loads of x in flight, then four
        // wave sums (per lane the same f-ascending products as the loop below: identical results)
        float dyv[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) dyv[t] = lane + 64 * t < F ? dy[row * lddy + F + lane + 64 * t] : 0.f;
        for (int j0 = wv; j0 < N; j0 += 16) {
            float xv[4][4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j = j0 + 4 * u < N ? j0 + 4 * u : N - 1;
#pragma unroll
                for (int t = 0; t < 4; ++t) xv[u][t] = lane + 64 * t < F ? x[(b * N + j) * ldx + lane + 64 * t] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                float sj = 0.f;
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    if (lane + 64 * t < F) sj += dyv[t] * xv[u][t];
                sj = wave_sum(sj);
                if (lane == 0 && j0 + 4 * u < N) dA[row * N + j0 + 4 * u] = sj;
            }
        }
        return;
    }
    for (int j = wv; j < N; j += 4) {
        float s = 0.f;
#pragma unroll 4
        for (int f = lane; f < F; f += 64) s += dy[row * lddy + F + f] * x[(b * N + j) * ldx + f];
        s = wave_sum(s);
        if (lane == 0) dA[row * N + j] = s;
    }
}

// dz[e, c, slot, :] = sum over the node rows that read it (supports: all n_query graphs; query q: graph q)
__global__ __launch_bounds__(256) void build_nodes_bwd_kernel(const float* __restrict__ dnodes, int ld,
                                                              float* __restrict__ dz, int zf, int n_ep, int n_way, int ns,
                                                              int nq, int fold) {
    const int per_class = (fold ? 2 * ns : ns) + nq;
    const long long total = (long long)n_ep * n_way * per_class * zf;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int col = (int)(i % zf);
        long long t = i / zf;
        const int slot = (int)(t % per_class); t /= per_class;
        const int c = (int)(t % n_way);
        const long long e = t / n_way;
        const int nsup = fold ? 2 * ns : ns;
        float s = 0.f;
        if (slot < nsup) {
            const int sn = fold ? (slot % ns) : slot;
            for (int q = 0; q < nq; ++q)
                s += dnodes[((((e * nq + q) * n_way + c) * (ns + 1)) + sn) * ld + col];
            if (fold) s *= 0.5f;
        } else {
            const int q = slot - nsup;
            s = dnodes[((((e * nq + q) * n_way + c) * (ns + 1)) + ns) * ld + col];
        }
        dz[i] = s;
    }
}

// dout[node(e,q,c,last), k] = dscores[e, c*nq+q, k]; zero elsewhere
__global__ __launch_bounds__(256) void gather_scores_bwd_kernel(const float* __restrict__ dscores, float* __restrict__ dout,
                                                                int ldo, int n_ep, int n_way, int ns, int nq, float* __restrict__ dbias) {
    const long long rows = (long long)n_ep * nq * n_way * (ns + 1);
    const long long total = rows * ldo;
    // dbias[k] (nullable) = column sums of d(out) = of dscores (every other row of d(out) is zero): the gradient of layer_last's
    // fc.bias (gnn.py:43-56; no BatchNorm behind it) by block 0 -- 16 row lanes x 16 columns, lanes combined in lane order (fixed
    // summation order) -- instead of the two column-sum launches the backward otherwise spends on it
    if (dbias && blockIdx.x == 0) {
        __shared__ float red[16][17];
        const int c = threadIdx.x & 15, rl = threadIdx.x >> 4;
        const long long nr = (long long)n_ep * n_way * nq;
        float acc = 0.f;
        if (c < n_way)
            for (long long r = rl; r < nr; r += 16) acc += dscores[r * n_way + c];
        red[rl][c] = acc;
        __syncthreads();
        if (rl == 0 && c < n_way) {
            float t = red[0][c];
#pragma unroll
            for (int j = 1; j < 16; ++j) t += red[j][c];
            dbias[c] = t;
        }
    }
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int k = (int)(i % ldo);
        long long r = i / ldo;
        const int s = (int)(r % (ns + 1)); r /= (ns + 1);
        const int c = (int)(r % n_way); r /= n_way;
        const int q = (int)(r % nq);
        const long long e = r / nq;
        float v = 0.f;
        if (s == ns && k < n_way) v = dscores[((e * n_way + c) * nq + q) * n_way + k];
        dout[i] = v;
    }
}

inline int bgrid(long long total, int cap = 4096) {
    long long b = (total + 255) / 256;
    if (b < 1) b = 1;
    if (b > cap) b = cap;
    return (int)b;
}

inline int bwd_chunks(int rows_per_group, int n_groups, int C) {
    const int tiles = (C + 63) / 64;
    long long want = (1024 + (long long)n_groups * tiles - 1) / ((long long)n_groups * tiles);
    int maxc = (rows_per_group + 63) / 64;
    if (want > maxc) want = maxc;
    if (want < 1) want = 1;
    return (int)want;
}

}  // namespace

// (C++ linkage: not part of the C ABI; reached through mft_debug_set_conv_tile(11000 + rows), include/mft_hip_testing.h)
void mft_bn_small_set_rows(int rows) { g_bn_small_rows = rows; }

// (the one workgroup of a channel group walks the lockstep groups one after the other: beyond two groups' worth of rows the three
// launches, which spread the groups over workgroups, are faster again -- k = 4: 6.66 against 6.58 ms per step)
static bool bn_bwd_small_ok(int C, int rows_per_group, int n_groups, long long gbs) {
    return g_bn_small_rows > 0 && rows_per_group <= g_bn_small_rows && (long long)rows_per_group * n_groups <= 2LL * g_bn_small_rows &&
           C % 4 == 0 && gbs == 0;
}

extern "C" long long mft_bn_backward_ws_floats(int C, int rows_per_group, int n_groups) {
    return 2LL * n_groups * bwd_chunks(rows_per_group, n_groups, C) * C + 2LL * n_groups * C;
}

extern "C" int mft_bn_backward_act(const float* x, int ldx, const float* dy, int lddy, const float* y_act, int ldya,
                                   float* dx, int lddx, int C, int rows_per_group, int n_groups, const float* mean,
                                   const float* rstd, const float* gamma, long long gb_group_stride, float* dgamma,
                                   float* dbeta, int act, float slope, float* ws, float* dgamma_sum, float* dbeta_sum,
                                   float* dbias_zero, void* stream) {
    if (C % 4 != 0 || ldx % 4 != 0 || lddy % 4 != 0 || (dx && lddx % 4 != 0) || (y_act && ldya % 4 != 0)) return MFT_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    BnBwdArgs p;
    p.x = x; p.dy = dy; p.y_act = y_act; p.dx = dx;
    p.ldx = ldx; p.lddy = lddy; p.ldya = ldya; p.lddx = lddx; p.C = C; p.rows_per_group = rows_per_group;
    p.chunks = bwd_chunks(rows_per_group, n_groups, C);
    p.rows_per_chunk = (rows_per_group + p.chunks - 1) / p.chunks;
    p.mean = mean; p.rstd = rstd; p.gamma = gamma; p.gbs = gb_group_stride;
    p.dgamma = dgamma; p.dbeta = dbeta; p.ws = ws; p.act = act; p.slope = slope;
    p.dgamma_sum = dgamma_sum; p.dbeta_sum = dbeta_sum; p.dbias_zero = dbias_zero; p.n_groups = n_groups;
    if (bn_bwd_small_ok(C, rows_per_group, n_groups, gb_group_stride)) {          // a few hundred KB: one launch, no hand-off between workgroups
        BnBwdSmallArgs a = {};
        a.job[0] = p; a.n = 1;
        hipLaunchKernelGGL(bn_bwd_small_kernel, dim3(C / 4, 1), dim3(256), 0, s, a);
        return mft_launch_status();
    }
    const bool walk = dgamma_sum != nullptr || dbeta_sum != nullptr;
    float* sums = ws + 2LL * n_groups * p.chunks * C;
    hipLaunchKernelGGL(bn_bwd_partial_kernel, dim3(p.chunks, (C + 63) / 64, n_groups), dim3(256), 0, s, p);
    if (n_groups == 1 && p.chunks >= 128)
        hipLaunchKernelGGL(bn_bwd_finalize_kernel<64>, dim3((C + 3) / 4, 1), dim3(256), 0, s, p, sums);
    else
        hipLaunchKernelGGL(bn_bwd_finalize_kernel<16>, dim3((C + 15) / 16, walk ? 1 : n_groups), dim3(256), 0, s, p, sums);
    if (dx) {
        const long long total = (long long)n_groups * rows_per_group * (C / 4);
        hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(bgrid(total, 2048)), dim3(256), 0, s, p, (const float*)sums, n_groups);
    }
    return mft_launch_status();
}

extern "C" int mft_bn_backward_act_multi(const MftBnBwdJob* jobs, int n_jobs, void* stream) {
    if (jobs == nullptr || n_jobs < 1 || n_jobs > BB_MULTI) return MFT_EINVAL;
    BnBwdMultiArgs a = {};
    int bp = 0, bf = 0, ba = 0;
    for (int j = 0; j < n_jobs; ++j) {
        const MftBnBwdJob& jb = jobs[j];
        if (jb.C % 4 != 0 || jb.ldx % 4 != 0 || jb.lddy % 4 != 0 || jb.dx == nullptr || jb.lddx % 4 != 0 || (jb.y_act && jb.ldya % 4 != 0) ||
            jb.rows_per_group < 1 || jb.n_groups < 1 || jb.ws == nullptr)
            return MFT_EINVAL;
        BnBwdArgs& p = a.job[j];
        p.x = jb.x; p.dy = jb.dy; p.y_act = jb.y_act; p.dx = jb.dx;
        p.ldx = jb.ldx; p.lddy = jb.lddy; p.ldya = jb.ldya; p.lddx = jb.lddx; p.C = jb.C; p.rows_per_group = jb.rows_per_group;
        p.chunks = bwd_chunks(jb.rows_per_group, jb.n_groups, jb.C);
        p.rows_per_chunk = (jb.rows_per_group + p.chunks - 1) / p.chunks;
        p.mean = jb.mean; p.rstd = jb.rstd; p.gamma = jb.gamma; p.gbs = 0;
        p.dgamma = jb.dgamma; p.dbeta = jb.dbeta; p.ws = jb.ws; p.act = jb.act; p.slope = jb.slope;
        p.dgamma_sum = jb.dgamma_sum; p.dbeta_sum = jb.dbeta_sum; p.dbias_zero = jb.dbias_zero; p.n_groups = jb.n_groups;
        const bool walk = jb.dgamma_sum != nullptr || jb.dbeta_sum != nullptr;
        a.sums[j] = jb.ws + 2LL * jb.n_groups * p.chunks * jb.C;
        a.lpc64[j] = (jb.n_groups == 1 && p.chunks >= 128) ? 1 : 0;
        a.start_p[j] = bp; a.start_f[j] = bf; a.start_a[j] = ba;
        bp += p.chunks * ((jb.C + 63) / 64) * jb.n_groups;
        bf += a.lpc64[j] ? (jb.C + 3) / 4 : ((jb.C + 15) / 16) * (walk ? 1 : jb.n_groups);
        ba += bgrid((long long)jb.n_groups * jb.rows_per_group * (jb.C / 4), 2048);
    }
    a.start_p[n_jobs] = bp; a.start_f[n_jobs] = bf; a.start_a[n_jobs] = ba; a.n = n_jobs;
    hipStream_t s = (hipStream_t)stream;
    {
        bool small = true;
        int cmax = 0;
        for (int j = 0; j < n_jobs; ++j) {
            small = small && bn_bwd_small_ok(jobs[j].C, jobs[j].rows_per_group, jobs[j].n_groups, 0);
            cmax = jobs[j].C > cmax ? jobs[j].C : cmax;
        }
        if (small) {                               // every job is small: one launch for all of them (grid.y = job)
            BnBwdSmallArgs sa = {};
            for (int j = 0; j < n_jobs; ++j) sa.job[j] = a.job[j];
            sa.n = n_jobs;
            hipLaunchKernelGGL(bn_bwd_small_kernel, dim3(cmax / 4, n_jobs), dim3(256), 0, s, sa);
            return mft_launch_status();
        }
    }
    hipLaunchKernelGGL(bn_bwd_partial_multi_kernel, dim3(bp), dim3(256), 0, s, a);
    hipLaunchKernelGGL(bn_bwd_finalize_multi_kernel, dim3(bf), dim3(256), 0, s, a);
    hipLaunchKernelGGL(bn_bwd_apply_multi_kernel, dim3(ba), dim3(256), 0, s, a);
    return mft_launch_status();
}

extern "C" int mft_act_backward(const float* dy, int lddy, const float* y, int ldy, float* dx, int lddx, int C,
                                long long rows, int act, float slope, int accumulate, void* stream) {
    hipLaunchKernelGGL(act_bwd_kernel, dim3(bgrid(rows * C)), dim3(256), 0, (hipStream_t)stream, dy, lddy, y, ldy, dx,
                       lddx, C, rows, act, slope, accumulate);
    return mft_launch_status();
}

extern "C" int mft_colsum(const float* x, int ldx, int C, long long rows, float* out, float* ws, void* stream) {
    // ws: >= ceil(rows/256) * C floats
    hipStream_t s = (hipStream_t)stream;
    const int chunks = (int)((rows + 255) / 256);
    if (C % 4 == 0 && ldx % 4 == 0 && (((unsigned long long)x) & 15) == 0 && (((unsigned long long)ws) & 15) == 0)
        hipLaunchKernelGGL(colsum_partial_kernel<true>, dim3(chunks, (C + 63) / 64), dim3(256), 0, s, x, ldx, C, rows, 256, ws);
    else
        hipLaunchKernelGGL(colsum_partial_kernel<false>, dim3(chunks, (C + 63) / 64), dim3(256), 0, s, x, ldx, C, rows, 256, ws);
    hipLaunchKernelGGL(colsum_final_kernel, dim3((C + 15) / 16), dim3(256), 0, s, (const float*)ws, C, chunks, out);
    return mft_launch_status();
}

extern "C" int mft_bn_relu_maxpool_arg(const float* x, float* y, unsigned char* argmax, int n_img, int H, int W, int C,
                                       int imgs_per_group, const float* mean, const float* rstd, const float* gamma,
                                       const float* beta, void* stream) {
    if (imgs_per_group <= 0) imgs_per_group = n_img;
    const int OH = (H + 2 - 3) / 2 + 1, OW = (W + 2 - 3) / 2 + 1;
    const long long total = (long long)n_img * OH * OW * C;
    if (C % 4 == 0)
        hipLaunchKernelGGL(bn_relu_maxpool_arg4_kernel, dim3(bgrid(total / 4)), dim3(256), 0, (hipStream_t)stream, x, y, argmax,
                           n_img, H, W, C, OH, OW, imgs_per_group, mean, rstd, gamma, beta);
    else
        hipLaunchKernelGGL(bn_relu_maxpool_arg_kernel, dim3(bgrid(total)), dim3(256), 0, (hipStream_t)stream, x, y, argmax,
                           n_img, H, W, C, OH, OW, imgs_per_group, mean, rstd, gamma, beta);
    return mft_launch_status();
}

extern "C" int mft_maxpool_relu_backward(const float* dy, const unsigned char* argmax, const float* y, float* dx,
                                         int n_img, int H, int W, int C, void* stream) {
    const int OH = (H + 2 - 3) / 2 + 1, OW = (W + 2 - 3) / 2 + 1;
    const long long total = (long long)n_img * H * W * C;
    if (C % 4 == 0 && total < 0x7fffffffLL)
        hipLaunchKernelGGL(maxpool_relu_bwd4_2x2_kernel, dim3(bgrid((long long)n_img * ((H + 1) / 2) * ((W + 1) / 2) * (C / 4))), dim3(256), 0,
                           (hipStream_t)stream, dy, argmax, y, dx, n_img, H, W, C, OH, OW);
    else
        hipLaunchKernelGGL(maxpool_relu_bwd_kernel, dim3(bgrid(total)), dim3(256), 0, (hipStream_t)stream, dy, argmax, y, dx,
                           n_img, H, W, C, OH, OW);
    return mft_launch_status();
}

extern "C" int mft_masked_softmax_backward(const float* A, const float* dA, float* ds, int ldds, int n_graphs, int N,
                                           void* stream) {
    const long long rows = (long long)n_graphs * N;
    hipLaunchKernelGGL(masked_softmax_bwd_kernel, dim3((int)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, A, dA,
                       ds, ldds, n_graphs, N);
    return mft_launch_status();
}

extern "C" int mft_pair_absdiff_backward(const float* x, int ldx, const float* dd, int ldd, float* dx, int lddx,
                                         int n_graphs, int N, int F, void* stream) {
    hipLaunchKernelGGL(pair_absdiff_bwd_kernel, dim3(bgrid((long long)n_graphs * N * F)), dim3(256), 0,
                       (hipStream_t)stream, x, ldx, dd, ldd, dx, lddx, n_graphs, N, F);
    return mft_launch_status();
}

extern "C" int mft_graph_aggregate_backward(const float* A, const float* x, int ldx, const float* dy, int lddy, float* dx,
                                            int lddx, float* dA, int n_graphs, int N, int F, int accumulate, void* stream) {
    hipLaunchKernelGGL(graph_aggregate_bwd_kernel, dim3(n_graphs * N), dim3(256), 0, (hipStream_t)stream, A, x, ldx, dy,
                       lddy, dx, lddx, dA, n_graphs, N, F, accumulate);
    return mft_launch_status();
}

extern "C" int mft_build_graph_nodes_backward(const float* dnodes, int ld, float* dz, int zf, int n_episodes, int n_way,
                                              int n_support, int n_query, int fold, void* stream) {
    const long long total = (long long)n_episodes * n_way * ((fold ? 2 * n_support : n_support) + n_query) * zf;
    hipLaunchKernelGGL(build_nodes_bwd_kernel, dim3(bgrid(total)), dim3(256), 0, (hipStream_t)stream, dnodes, ld, dz, zf,
                       n_episodes, n_way, n_support, n_query, fold);
    return mft_launch_status();
}

extern "C" int mft_gather_query_scores_backward(const float* dscores, float* dout, int ldo, int n_episodes, int n_way,
                                                int n_support, int n_query, float* dbias, void* stream) {
    if (dbias && n_way > 16) return MFT_EINVAL;
    const long long total = (long long)n_episodes * n_query * n_way * (n_support + 1) * ldo;
    hipLaunchKernelGGL(gather_scores_bwd_kernel, dim3(bgrid(total)), dim3(256), 0, (hipStream_t)stream, dscores, dout, ldo,
                       n_episodes, n_way, n_support, n_query, dbias);
    return mft_launch_status();
}
