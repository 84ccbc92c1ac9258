// Grouped train-mode BatchNorm (statistics / apply / backward), pooling, ReLU for NHWC fp32 tensors.
//
// Replaces F.batch_norm(training=True), F.relu / F.leaky_relu, max/avg pooling and their autograd
// on the reference hot path (backbone.py:224-261,409-411,427-430; gnn.py:65-102; gnnnet.py:30).
// A "group" is one BatchNorm mini-batch (e.g. the 5 images of one episode's inner step); grouped
// launches process many independent episodes per kernel.  All of this is HBM-bound streaming work:
// every thread reads/writes float4 along the contiguous channel axis, reductions are fixed-order
// (no float atomics) so reruns are bit-identical.
#include "mft_common.h"
#include "bn_fold.h"

namespace {

// ------------------------------------------------------------------------------------------- stats
// Shifted one-pass moments: s = x[first row of group][c]; S1 = sum(x-s), S2 = sum((x-s)^2).
// mean = s + S1/n, var = S2/n - (S1/n)^2  (cancellation-free as long as s is within a few sigma).
// grid (chunks, C/64 tiles... ) : block = 256 threads = 16 channel-quads (64 channels) x 16 row lanes.
constexpr int ST_ROWS = 16;

__device__ __forceinline__ void bn_stats_partial_body(const float* __restrict__ x, int ldx, int C,
                                                      int rows_per_group, int rows_per_chunk, int chunks,
                                                      float* __restrict__ ws, const int bx, const int by, const int bz) {
    const int cq = threadIdx.x & 15;          // channel quad within the 64-channel tile
    const int rl = threadIdx.x >> 4;          // row lane 0..15
    const int c = by * 64 + cq * 4;
    const int g = bz;
    const int chunk = bx;
    const long long row0 = (long long)g * rows_per_group;
    __shared__ f32x4 red1[ST_ROWS][16];
    __shared__ f32x4 red2[ST_ROWS][16];
    f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
    if (c < C) {
        const f32x4 sh = *(const f32x4*)(x + row0 * ldx + c);
        const int rbeg = chunk * rows_per_chunk;
        const int rend = min(rbeg + rows_per_chunk, rows_per_group);
#pragma unroll 4
        for (int rr = rbeg + rl; rr < rend; rr += ST_ROWS) {         // (four loads in flight; same order of additions)
            f32x4 v = *(const f32x4*)(x + (row0 + rr) * ldx + c);
            v -= sh;
            s1 += v;
            s2 += v * v;
        }
    }
    red1[rl][cq] = s1;
    red2[rl][cq] = s2;
    __syncthreads();
    if (rl == 0 && c < C) {
#pragma unroll
        for (int k = 1; k < ST_ROWS; ++k) {
            s1 += red1[k][cq];
            s2 += red2[k][cq];
        }
        float* o = ws + (((long long)g * chunks + chunk) * C + c) * 2;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            o[2 * e] = s1[e];
            o[2 * e + 1] = s2[e];
        }
    }
}

// 256 threads = (256 / LPC) channels x LPC chunk lanes (fixed xor tree over the lanes: deterministic).  LPC = 16 everywhere except
// ONE group with >= 128 chunks (a single 105-image meta-training episode: 199-1024 partials per channel, which 16 lanes walk in
// 13-64 dependent iterations -- 8-23 us per BatchNorm, round 5): there a whole wave sums one channel.
__global__ __launch_bounds__(256) void bn_stats_partial(const float* __restrict__ x, int ldx, int C,
                                                        int rows_per_group, int rows_per_chunk, int chunks,
                                                        float* __restrict__ ws) {
    bn_stats_partial_body(x, ldx, C, rows_per_group, rows_per_chunk, chunks, ws, blockIdx.x, blockIdx.y, blockIdx.z);
}

template <int LPC>
__device__ __forceinline__ void bn_stats_finalize_body(const float* __restrict__ x, int ldx, int C, int rows_per_group,
                                                       int chunks, const float* __restrict__ ws, float eps,
                                                       float* __restrict__ mean, float* __restrict__ rstd,
                                                       float* running_mean, float* running_var, float momentum,
                                                       long long* num_batches_tracked, const int bx, const int by) {
    const int kl = threadIdx.x % LPC;
    const int c = bx * (256 / LPC) + threadIdx.x / LPC;
    const int g = by;
    float s1 = 0.f, s2 = 0.f;
    if (c < C)
#pragma unroll 4
        for (int k = kl; k < chunks; k += LPC) {
            const float* o = ws + (((long long)g * chunks + k) * C + c) * 2;
            s1 += o[0];
            s2 += o[1];
        }
#pragma unroll
    for (int off = LPC / 2; off > 0; off >>= 1) {
        s1 += __shfl_xor(s1, off, 64);
        s2 += __shfl_xor(s2, off, 64);
    }
    if (c >= C || kl != 0) return;
    const float inv = 1.f / (float)rows_per_group;
    const float sh = x[(long long)g * rows_per_group * ldx + c];
    const float d = s1 * inv;
    const float m = sh + d;
    float var = s2 * inv - d * d;
    var = fmaxf(var, 0.f);
    mean[(long long)g * C + c] = m;
    rstd[(long long)g * C + c] = 1.0f / sqrtf(var + eps);
    if (running_mean && g == 0) {          // several groups (k episodes in lockstep): group 0's statistics advance the buffers, as rank 0's
        const float unb = var * ((float)rows_per_group / (float)max(rows_per_group - 1, 1));   // do in an episode-parallel run
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * m;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * unb;
        if (num_batches_tracked && c == 0 && g == 0) *num_batches_tracked += 1;      // nn.BatchNorm2d's counter (one thread of the launch)
    }
}

template <int LPC>
__global__ __launch_bounds__(256) void bn_stats_finalize(const float* __restrict__ x, int ldx, int C, int rows_per_group,
                                                         int chunks, const float* __restrict__ ws, float eps,
                                                         float* __restrict__ mean, float* __restrict__ rstd,
                                                         float* running_mean, float* running_var, float momentum,
                                                         long long* num_batches_tracked) {
    bn_stats_finalize_body<LPC>(x, ldx, C, rows_per_group, chunks, ws, eps, mean, rstd, running_mean, running_var, momentum,
                                num_batches_tracked, blockIdx.x, blockIdx.y);
}

// Several independent statistics problems in one launch pair (mft_bn_stats_multi): SimpleBlock's BN2 and BNshortcut see two
// tensors that exist at the same time (backbone.py:256-259) -- one partial launch and one finalize launch for both instead of two
// each (a dependent launch costs >= 4.7 us in the replayed meta-training step whatever it does).  Block b works for job j with
// start[j] <= b < start[j + 1], numbered inside the job as its own launch would number it: bit-identical results.
constexpr int BS_MULTI = 8;
struct StatsJob {
    const float* x; float* mean; float* rstd; float* ws; float* running_mean; float* running_var; long long* nbt;
    int ldx, C, rows_per_group, rows_per_chunk, chunks, n_groups, lpc64;
    float eps, momentum;
};
struct StatsMultiArgs {
    StatsJob job[BS_MULTI];
    int start_p[BS_MULTI + 1], start_f[BS_MULTI + 1];
    int n;
};

__global__ __launch_bounds__(256) void bn_stats_partial_multi(StatsMultiArgs a) {
    int j = 0;
#pragma unroll 1
    while (j + 1 < a.n && (int)blockIdx.x >= a.start_p[j + 1]) ++j;
    const StatsJob p = a.job[j];
    const int local = blockIdx.x - a.start_p[j], cy = (p.C + 63) / 64;
    bn_stats_partial_body(p.x, p.ldx, p.C, p.rows_per_group, p.rows_per_chunk, p.chunks, p.ws, local % p.chunks, (local / p.chunks) % cy,
                          local / (p.chunks * cy));
}

__global__ __launch_bounds__(256) void bn_stats_finalize_multi(StatsMultiArgs a) {
    int j = 0;
#pragma unroll 1
    while (j + 1 < a.n && (int)blockIdx.x >= a.start_f[j + 1]) ++j;
    const StatsJob p = a.job[j];
    const int local = blockIdx.x - a.start_f[j];
    if (p.lpc64) {
        bn_stats_finalize_body<64>(p.x, p.ldx, p.C, p.rows_per_group, p.chunks, p.ws, p.eps, p.mean, p.rstd, p.running_mean, p.running_var,
                                   p.momentum, p.nbt, local, 0);
    } else {
        const int fx = (p.C + 15) / 16;
        bn_stats_finalize_body<16>(p.x, p.ldx, p.C, p.rows_per_group, p.chunks, p.ws, p.eps, p.mean, p.rstd, p.running_mean, p.running_var,
                                   p.momentum, p.nbt, local % fx, local / fx);
    }
}

// ------------------------------------------------------------------------------------------- apply
struct ApplyArgs {
    const float* x; float* y; int ldx, ldy, C, rows_per_group, n_groups;
    const float* mean; const float* rstd; const float* gamma; const float* beta; long long gbs;
    const float* res; int ldr; const float* rmean; const float* rrstd; const float* rgamma; const float* rbeta;
    int act; float slope;
    unsigned short* planes; long long plane_stride;      // optional bf16x3 planes of y ([3][rows][C], csrc/conv_x3.hip operand)
};

__device__ __forceinline__ float act_f(float v, int act, float slope) {
    if (act == MFT_ACT_RELU) return fmaxf(v, 0.f);
    if (act == MFT_ACT_LRELU) return v > 0.f ? v : v * slope;
    return v;
}

template <bool PLANES>
__device__ __forceinline__ void bn_apply_blocks(const ApplyArgs& p, const int block, const int n_blocks) {
    const int cq = p.C >> 2;
    const long long total = (long long)p.n_groups * p.rows_per_group * cq;
    for (long long i = (long long)block * blockDim.x + threadIdx.x; i < total;
         i += (long long)n_blocks * blockDim.x) {
        const long long row = i / cq;
        const int c = (int)(i - row * cq) * 4;
        const int g = (int)(row / p.rows_per_group);
        const f32x4 v = *(const f32x4*)(p.x + row * p.ldx + c);
        const f32x4 mu = *(const f32x4*)(p.mean + (long long)g * p.C + c);
        const f32x4 rs = *(const f32x4*)(p.rstd + (long long)g * p.C + c);
        const f32x4 ga = *(const f32x4*)(p.gamma + g * p.gbs + c);
        const f32x4 be = *(const f32x4*)(p.beta + g * p.gbs + c);
        const bool folded = (p.act & MFT_BN_AFFINE_FMA) != 0;       // the frozen trunk's arithmetic (bn_fold.h)
        const int act = p.act & 0xff;
        f32x4 sc, sh, o;
        if (folded) {
            mft_bn_fold4(mu, rs, ga, be, sc, sh);
            o = mft_bn_affine4(v, sc, sh);
        } else {
            o = (v - mu) * rs * ga + be;
        }
        if (p.res) {
            f32x4 rv = *(const f32x4*)(p.res + row * p.ldr + c);
            if (p.rmean) {
                const f32x4 rmu = *(const f32x4*)(p.rmean + (long long)g * p.C + c);
                const f32x4 rrs = *(const f32x4*)(p.rrstd + (long long)g * p.C + c);
                const f32x4 rga = *(const f32x4*)(p.rgamma + g * p.gbs + c);
                const f32x4 rbe = *(const f32x4*)(p.rbeta + g * p.gbs + c);
                if (folded) {
                    mft_bn_fold4(rmu, rrs, rga, rbe, sc, sh);
                    rv = mft_bn_affine4(rv, sc, sh);
                } else {
                    rv = (rv - rmu) * rrs * rga + rbe;
                }
            }
            o += rv;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = act_f(o[e], act, p.slope);
        if (!PLANES || p.y) *(f32x4*)(p.y + row * p.ldy + c) = o;
        if constexpr (PLANES) {
            mft_u32x2 p1, p2, p3;
            mft_split4_bf16(o, p1, p2, p3);
            unsigned short* q = p.planes + row * p.C + c;
            *(mft_u32x2*)(q) = p1;
            *(mft_u32x2*)(q + p.plane_stride) = p2;
            *(mft_u32x2*)(q + 2 * p.plane_stride) = p3;
        }
    }
}

template <bool PLANES>
__global__ __launch_bounds__(256) void bn_apply_kernel(ApplyArgs p) {
    bn_apply_blocks<PLANES>(p, blockIdx.x, gridDim.x);
}

// several apply problems in one launch (mft_bn_apply_multi): block b works for the job j with start[j] <= b < start[j + 1], as block
// b - start[j] of that job's own launch
constexpr int BA_MULTI = 16;
struct ApplyMultiArgs {
    ApplyArgs job[BA_MULTI];
    int start[BA_MULTI + 1];
    int n;
};
static_assert(sizeof(ApplyMultiArgs) <= 4000, "kernarg segment");

__global__ __launch_bounds__(256) void bn_apply_multi_kernel(ApplyMultiArgs a) {
    int j = 0;
#pragma unroll 1
    while (j + 1 < a.n && (int)blockIdx.x >= a.start[j + 1]) ++j;
    const ApplyArgs p = a.job[j];
    bn_apply_blocks<false>(p, blockIdx.x - a.start[j], a.start[j + 1] - a.start[j]);
}

// The same apply for the outputs of bf16x3 trunk convolutions whose statistics are still per-tile partials (csrc/conv_x3.hip):
// grid (sub-blocks, groups); each workgroup first merges the partials of ITS group (bn_fold.h, the order the stand-alone finalize
// used) into an LDS table of (scale, shift) per channel -- for the main and, if present, the residual BatchNorm -- then streams
// its share of the group's rows.  One launch instead of finalize + finalize + apply.
struct ApplyWsArgs {
    const float* x; float* y; int ldx, ldy, C, rows_per_group, n_groups, M, BM;
    const float* ws; const float* gamma; const float* beta;
    const float* res; int ldr; const float* rws; const float* rgamma; const float* rbeta;
    int act; float slope, eps;
    float* mean; float* rstd; float* rmean; float* rrstd;      // optional copies of the merged statistics (written by sub-block 0)
    int max_tiles;
};

__global__ __launch_bounds__(256) void bn_apply_ws_kernel(ApplyWsArgs p) {
    extern __shared__ __attribute__((aligned(16))) float tab[];         // [4][C]: scale, shift, residual scale, residual shift
    const int g = blockIdx.y, C = p.C;
    char* stage = reinterpret_cast<char*>(tab + 4 * C);                  // partials + tile weights of this group (bn_fold.h), x2 with a residual BatchNorm
    char* rstage = stage + mft_x3_stage_bytes(p.max_tiles, C);
    mft_x3_stats_stage(p.ws, C, g, p.M, p.rows_per_group, p.BM, p.max_tiles, stage, threadIdx.x, 256);
    if (p.rws) mft_x3_stats_stage(p.rws, C, g, p.M, p.rows_per_group, p.BM, p.max_tiles, rstage, threadIdx.x, 256);
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        float mu, rs, sc, sh;
        mft_x3_stats_staged(stage, p.max_tiles, C, c, g, p.M, p.rows_per_group, p.BM, p.eps, mu, rs);
        mft_bn_fold(mu, rs, p.gamma[c], p.beta[c], sc, sh);
        tab[c] = sc; tab[C + c] = sh;
        if (blockIdx.x == 0 && p.mean) { p.mean[(long long)g * C + c] = mu; p.rstd[(long long)g * C + c] = rs; }
        if (p.rws) {
            mft_x3_stats_staged(rstage, p.max_tiles, C, c, g, p.M, p.rows_per_group, p.BM, p.eps, mu, rs);
            mft_bn_fold(mu, rs, p.rgamma[c], p.rbeta[c], sc, sh);
            tab[2 * C + c] = sc; tab[3 * C + c] = sh;
            if (blockIdx.x == 0 && p.rmean) { p.rmean[(long long)g * C + c] = mu; p.rrstd[(long long)g * C + c] = rs; }
        }
    }
    __syncthreads();
    const int cq = C >> 2;
    const int total = p.rows_per_group * cq;
    const long long row0 = (long long)g * p.rows_per_group;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int rr = i / cq;
        const int c = (i - rr * cq) * 4;
        const long long row = row0 + rr;
        const f32x4 v = *(const f32x4*)(p.x + row * p.ldx + c);
        f32x4 o = mft_bn_affine4(v, *(const f32x4*)(tab + c), *(const f32x4*)(tab + C + c));
        if (p.res) {
            f32x4 rv = *(const f32x4*)(p.res + row * p.ldr + c);
            if (p.rws) rv = mft_bn_affine4(rv, *(const f32x4*)(tab + 2 * C + c), *(const f32x4*)(tab + 3 * C + c));
            o += rv;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = act_f(o[e], p.act, p.slope);
        *(f32x4*)(p.y + row * p.ldy + c) = o;
    }
}

// BN -> ReLU -> MaxPool(3,2,1), NHWC.  relu(max(.)) == max(relu(.)); padding never wins (>= 1 valid tap).
template <bool PLANES>
__global__ __launch_bounds__(256) void bn_relu_maxpool_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                              int n_img, int H, int W, int C, int OH, int OW,
                                                              int imgs_per_group, const float* __restrict__ mean,
                                                              const float* __restrict__ rstd,
                                                              const float* __restrict__ gamma,
                                                              const float* __restrict__ beta,
                                                              const int* __restrict__ src_idx,
                                                              unsigned short* __restrict__ planes, long long plane_stride) {
    const int cq = C >> 2;
    const long long total = (long long)n_img * OH * OW * cq;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % cq) * 4;
        long long t = i / cq;
        const int ow = (int)(t % OW); t /= OW;
        const int oh = (int)(t % OH);
        const int n = (int)(t / OH);
        const int g = n / imgs_per_group;
        const long long ns = src_idx ? (long long)src_idx[n] : (long long)n;   // image slot in the (cached) source
        const f32x4 mu = *(const f32x4*)(mean + (long long)g * C + c);
        const f32x4 rs = *(const f32x4*)(rstd + (long long)g * C + c);
        const f32x4 ga = *(const f32x4*)(gamma + c);
        const f32x4 be = *(const f32x4*)(beta + c);
        f32x4 best = {-3.4e38f, -3.4e38f, -3.4e38f, -3.4e38f};
#pragma unroll
        for (int dh = 0; dh < 3; ++dh) {
            const int ih = oh * 2 - 1 + dh;
            if (ih < 0 || ih >= H) continue;
#pragma unroll
            for (int dw = 0; dw < 3; ++dw) {
                const int iw = ow * 2 - 1 + dw;
                if (iw < 0 || iw >= W) continue;
                const f32x4 v = *(const f32x4*)(x + ((ns * H + ih) * W + iw) * C + c);
                const f32x4 o = __builtin_elementwise_fma((v - mu) * rs, ga, be);
#pragma unroll
                for (int e = 0; e < 4; ++e) best[e] = fmaxf(best[e], o[e]);
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) best[e] = fmaxf(best[e], 0.f);
        *(f32x4*)(y + i * 4) = best;
        if constexpr (PLANES) {
            mft_u32x2 p1, p2, p3;
            mft_split4_bf16(best, p1, p2, p3);
            unsigned short* q = planes + i * 4;
            *(mft_u32x2*)(q) = p1;
            *(mft_u32x2*)(q + plane_stride) = p2;
            *(mft_u32x2*)(q + 2 * plane_stride) = p3;
        }
    }
}

__global__ __launch_bounds__(256) void global_avgpool_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                             int n_img, int HW, int C) {
    const int cq = C >> 2;
    const long long total = (long long)n_img * cq;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int n = (int)(i / cq);
    const int c = (int)(i % cq) * 4;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < HW; ++k) s += *(const f32x4*)(x + ((long long)n * HW + k) * C + c);
    const float inv = 1.f / (float)HW;
    *(f32x4*)(y + (long long)n * C + c) = s * inv;
}

__global__ __launch_bounds__(256) void avgpool_relu_bwd_kernel(const float* __restrict__ dfeat,
                                                               const float* __restrict__ out,
                                                               float* __restrict__ dout, int n_img, int HW, int C) {
    const int cq = C >> 2;
    const long long total = (long long)n_img * HW * cq;
    const float inv = 1.f / (float)HW;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % cq) * 4;
        const long long pix = i / cq;
        const int n = (int)(pix / HW);
        const f32x4 o = *(const f32x4*)(out + pix * C + c);
        const f32x4 d = *(const f32x4*)(dfeat + (long long)n * C + c);
        f32x4 rv;
#pragma unroll
        for (int e = 0; e < 4; ++e) rv[e] = o[e] > 0.f ? d[e] * inv : 0.f;
        *(f32x4*)(dout + pix * C + c) = rv;
    }
}

// ------------------------------------------------------------------------------------------ backward
// One block per (group, 64-channel tile): pass 1 reduces sum(dy) and sum(dy*xhat) over the group's rows
// (fixed order), pass 2 writes dx.  Rows per group on the hot path are 9..245 (L2-resident re-read).
struct BwdArgs {
    const float* x; const float* dy; const float* ro; float* dx;
    int ldx, lddy, ldro, lddx, C, rows_per_group;
    const float* mean; const float* rstd; const float* gamma; long long gbs;
    float* dgamma; float* dbeta;
};

__global__ __launch_bounds__(256) void bn_backward_kernel(BwdArgs p) {
    const int cq = threadIdx.x & 15;
    const int rl = threadIdx.x >> 4;
    const int c = blockIdx.x * 64 + cq * 4;
    const int g = blockIdx.y;
    const long long row0 = (long long)g * p.rows_per_group;
    __shared__ f32x4 red1[ST_ROWS][16];
    __shared__ f32x4 red2[ST_ROWS][16];
    f32x4 mu = {0.f, 0.f, 0.f, 0.f}, rs = mu, ga = mu;
    const bool ok = c < p.C;
    if (ok) {
        mu = *(const f32x4*)(p.mean + (long long)g * p.C + c);
        rs = *(const f32x4*)(p.rstd + (long long)g * p.C + c);
        ga = *(const f32x4*)(p.gamma + g * p.gbs + c);
    }
    f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = s1;
    if (ok) {
        for (int rr = rl; rr < p.rows_per_group; rr += ST_ROWS) {
            f32x4 d = *(const f32x4*)(p.dy + (row0 + rr) * p.lddy + c);
            if (p.ro) {
                const f32x4 o = *(const f32x4*)(p.ro + (row0 + rr) * p.ldro + c);
#pragma unroll
                for (int e = 0; e < 4; ++e) d[e] = o[e] > 0.f ? d[e] : 0.f;
            }
            const f32x4 xh = (*(const f32x4*)(p.x + (row0 + rr) * p.ldx + c) - mu) * rs;
            s1 += d;
            s2 += d * xh;
        }
    }
    red1[rl][cq] = s1;
    red2[rl][cq] = s2;
    __syncthreads();
    s1 = red1[0][cq];
    s2 = red2[0][cq];
#pragma unroll
    for (int k = 1; k < ST_ROWS; ++k) {
        s1 += red1[k][cq];
        s2 += red2[k][cq];
    }
    if (!ok) return;
    if (rl == 0) {
        if (p.dgamma) *(f32x4*)(p.dgamma + (long long)g * p.C + c) = s2;
        if (p.dbeta) *(f32x4*)(p.dbeta + (long long)g * p.C + c) = s1;
    }
    if (!p.dx) return;
    const float inv = 1.f / (float)p.rows_per_group;
    const f32x4 m1 = s1 * inv, m2 = s2 * inv;
    const f32x4 k = ga * rs;
    for (int rr = rl; rr < p.rows_per_group; rr += ST_ROWS) {
        f32x4 d = *(const f32x4*)(p.dy + (row0 + rr) * p.lddy + c);
        if (p.ro) {
            const f32x4 o = *(const f32x4*)(p.ro + (row0 + rr) * p.ldro + c);
#pragma unroll
            for (int e = 0; e < 4; ++e) d[e] = o[e] > 0.f ? d[e] : 0.f;
        }
        const f32x4 xh = (*(const f32x4*)(p.x + (row0 + rr) * p.ldx + c) - mu) * rs;
        *(f32x4*)(p.dx + (row0 + rr) * p.lddx + c) = k * (d - m1 - xh * m2);
    }
}

// ------------------------------------------------------------------------- per-image moments (stem cache)
// The stem convolution of an image does not depend on which mini-batch the image lands in, only the BatchNorm
// statistics do.  Per image and channel we keep (mean_i, M2_i = sum (x - mean_i)^2) over its HW pixels; the
// statistics of any mini-batch of equally sized images follow from Chan's parallel combination
//   mean = avg(mean_i),  M2 = sum M2_i + HW * sum (mean_i - mean)^2,  var = M2 / (k * HW)
// in a fixed order (bit-identical reruns).  One block per (image, 64-channel tile).
__global__ __launch_bounds__(256) void bn_image_moments_kernel(const float* __restrict__ x, int ldx, int C, int rows,
                                                               float* __restrict__ mean_img,
                                                               float* __restrict__ m2_img) {
    const int cq = threadIdx.x & 15;
    const int rl = threadIdx.x >> 4;
    const int c = blockIdx.y * 64 + cq * 4;
    const long long img = blockIdx.x;
    const long long row0 = img * rows;
    __shared__ f32x4 red1[ST_ROWS][16];
    __shared__ f32x4 red2[ST_ROWS][16];
    f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f}, sh = s1;
    if (c < C) {
        sh = *(const f32x4*)(x + row0 * ldx + c);
        for (int rr = rl; rr < rows; rr += ST_ROWS) {
            f32x4 v = *(const f32x4*)(x + (row0 + rr) * ldx + c);
            v -= sh;
            s1 += v;
            s2 += v * v;
        }
    }
    red1[rl][cq] = s1;
    red2[rl][cq] = s2;
    __syncthreads();
    if (rl == 0 && c < C) {
#pragma unroll
        for (int k = 1; k < ST_ROWS; ++k) {
            s1 += red1[k][cq];
            s2 += red2[k][cq];
        }
        const float inv = 1.f / (float)rows;
        f32x4 mu, m2;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float d = s1[e] * inv;
            mu[e] = sh[e] + d;
            m2[e] = fmaxf(s2[e] - s1[e] * d, 0.f);
        }
        *(f32x4*)(mean_img + img * C + c) = mu;
        *(f32x4*)(m2_img + img * C + c) = m2;
    }
}

// batch statistics of imgs_per_group images from their cached per-image (mean, M2): exact pooled variance.
// mean_of(i) / m2_of(i): the channel's moments of the group's i-th image.
template <class MeanOf, class M2Of>
__device__ __forceinline__ void combine_moments_impl(MeanOf mean_of, M2Of m2_of, int rows_per_img, int imgs_per_group, float eps,
                                                     float& mean, float& rstd) {
    float mu = 0.f;
    for (int i = 0; i < imgs_per_group; ++i) mu += mean_of(i);
    mu /= (float)imgs_per_group;
    float m2 = 0.f, dev = 0.f;
    for (int i = 0; i < imgs_per_group; ++i) {
        const float d = mean_of(i) - mu;
        m2 += m2_of(i);
        dev += d * d;
    }
    const float var = (m2 + (float)rows_per_img * dev) / ((float)rows_per_img * (float)imgs_per_group);
    mean = mu;
    rstd = 1.0f / sqrtf(var + eps);
}

__device__ __forceinline__ void combine_moments(const float* __restrict__ mean_img, const float* __restrict__ m2_img,
                                                const int* __restrict__ id, int C, int c, int rows_per_img, int imgs_per_group,
                                                float eps, float& mean, float& rstd) {
    combine_moments_impl([&](int i) { return mean_img[(long long)id[i] * C + c]; },
                         [&](int i) { return m2_img[(long long)id[i] * C + c]; }, rows_per_img, imgs_per_group, eps, mean, rstd);
}

__device__ __forceinline__ void combine_moments_lds(const float* smean, const float* sm2, int C, int c, int rows_per_img,
                                                    int imgs_per_group, float eps, float& mean, float& rstd) {
    combine_moments_impl([&](int i) { return smean[i * C + c]; }, [&](int i) { return sm2[i * C + c]; }, rows_per_img,
                         imgs_per_group, eps, mean, rstd);
}

__global__ void bn_combine_moments_kernel(const float* __restrict__ mean_img, const float* __restrict__ m2_img,
                                          const int* __restrict__ idx, int C, int rows_per_img, int imgs_per_group,
                                          float eps, float* __restrict__ mean, float* __restrict__ rstd) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    const int g = blockIdx.y;
    if (c >= C) return;
    float mu, rs;
    combine_moments(mean_img, m2_img, idx + (long long)g * imgs_per_group, C, c, rows_per_img, imgs_per_group, eps, mu, rs);
    mean[(long long)g * C + c] = mu;
    rstd[(long long)g * C + c] = rs;
}

inline int grid_for(long long total, int block = 256, int cap = 256 * 8) {
    long long b = (total + block - 1) / block;
    if (b < 1) b = 1;
    if (b > cap) b = cap;
    return (int)b;
}

inline int stats_chunks(int rows_per_group, int n_groups, int C) {
    // enough blocks to fill 256 CUs, at least 64 rows per chunk
    const int tiles = (C + 63) / 64;
    long long want = (1024 + (long long)n_groups * tiles - 1) / ((long long)n_groups * tiles);
    int maxc = (rows_per_group + 63) / 64;
    if (want > maxc) want = maxc;
    if (want < 1) want = 1;
    return (int)want;
}

// ------------------------------------------------------------------------------------------- train-mode BatchNorm, SMALL problems
// statistics -> finalize -> apply are three dependent launches (>= 4.7 us each in a replayed meta-training step); the deep layers of
// a single 105-image episode (trunk.6 / trunk.7: 3,780 / 945 rows) and the head's BatchNorm1d layers (105 / 480 rows) are a few hundred
// KB.  One workgroup per (group, four channels): 256 row lanes, pass 1 sums (x - pivot) and its square over the group's rows (the
// pivot-shifted one-pass form of the launches above), the 256 lanes are reduced inside the workgroup (wave xor tree, then the four
// waves in order: fixed), pass 2 re-reads the 16-byte column out of cache and applies.  A residual may go through its OWN BatchNorm
// (SimpleBlock's shortcut, backbone.py:256-260): its statistics are taken in the same pass.  No hand-off between workgroups, one launch.
struct BnFwdSmallArgs {
    const float* x; float* y; const float* gamma; const float* beta; float* mean; float* rstd;
    float* running_mean; float* running_var; long long* nbt;
    const float* res; const float* rgamma; const float* rbeta; float* rmean; float* rrstd;
    float* rrunning_mean; float* rrunning_var; long long* rnbt;
    int ldx, ldy, ldr, C, rows_per_group, n_groups, act;
    float eps, momentum, slope;
    int y_scalar;        // y is not 16-byte aligned (a column window of a wider matrix): four 4-byte stores per thread
};

__device__ __forceinline__ void bn_small_block_sum(f32x4& a, f32x4& b, float (*red)[8]) {     // red[4 waves][8]
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
        for (int e = 0; e < 4; ++e) { red[wave][e] = a[e]; red[wave][4 + e] = b[e]; }
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        a[e] = ((red[0][e] + red[1][e]) + red[2][e]) + red[3][e];
        b[e] = ((red[0][4 + e] + red[1][4 + e]) + red[2][4 + e]) + red[3][4 + e];
    }
}

__device__ __forceinline__ void bn_small_finish(const f32x4 s1, const f32x4 s2, const f32x4 pivot, int rows, float eps, f32x4& mu,
                                                f32x4& rs, f32x4& var) {
    const float inv = 1.f / (float)rows;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float d = s1[e] * inv;
        mu[e] = pivot[e] + d;
        var[e] = fmaxf(s2[e] * inv - d * d, 0.f);
        rs[e] = 1.0f / sqrtf(var[e] + eps);
    }
}

__global__ __launch_bounds__(256) void bn_forward_small_kernel(BnFwdSmallArgs p) {
    __shared__ float red[2][4][8];
    const int c = blockIdx.x * 4, g = blockIdx.y, t = threadIdx.x;
    const long long row0 = (long long)g * p.rows_per_group;
    const bool rbn = p.res != nullptr && p.rgamma != nullptr;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const f32x4 pv = *(const f32x4*)(p.x + row0 * p.ldx + c);
    const f32x4 pr = rbn ? *(const f32x4*)(p.res + row0 * p.ldr + c) : zero4;
    f32x4 s1 = zero4, s2 = zero4, q1 = zero4, q2 = zero4;
    if (rbn) {
#pragma unroll 4
        for (int r = t; r < p.rows_per_group; r += 256) {
            const f32x4 v = *(const f32x4*)(p.x + (row0 + r) * p.ldx + c) - pv;
            const f32x4 w = *(const f32x4*)(p.res + (row0 + r) * p.ldr + c) - pr;
            s1 += v; s2 += v * v;
            q1 += w; q2 += w * w;
        }
    } else {
#pragma unroll 4
        for (int r = t; r < p.rows_per_group; r += 256) {
            const f32x4 v = *(const f32x4*)(p.x + (row0 + r) * p.ldx + c) - pv;
            s1 += v; s2 += v * v;
        }
    }
    bn_small_block_sum(s1, s2, red[0]);
    f32x4 mu, rs, var, rmu = zero4, rrs = zero4, rvar = zero4;
    bn_small_finish(s1, s2, pv, p.rows_per_group, p.eps, mu, rs, var);
    if (rbn) {
        bn_small_block_sum(q1, q2, red[1]);
        bn_small_finish(q1, q2, pr, p.rows_per_group, p.eps, rmu, rrs, rvar);
    }
    if (t == 0) {
        const float unbias = (float)p.rows_per_group / (float)max(p.rows_per_group - 1, 1);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            p.mean[(long long)g * p.C + c + e] = mu[e];
            p.rstd[(long long)g * p.C + c + e] = rs[e];
            if (p.running_mean && g == 0) {      // group 0 advances the buffers (rank 0's episode), as mft_bn_stats
                p.running_mean[c + e] = (1.f - p.momentum) * p.running_mean[c + e] + p.momentum * mu[e];
                p.running_var[c + e] = (1.f - p.momentum) * p.running_var[c + e] + p.momentum * (var[e] * unbias);
            }
            if (rbn) {
                p.rmean[(long long)g * p.C + c + e] = rmu[e];
                p.rrstd[(long long)g * p.C + c + e] = rrs[e];
                if (p.rrunning_mean && g == 0) {
                    p.rrunning_mean[c + e] = (1.f - p.momentum) * p.rrunning_mean[c + e] + p.momentum * rmu[e];
                    p.rrunning_var[c + e] = (1.f - p.momentum) * p.rrunning_var[c + e] + p.momentum * (rvar[e] * unbias);
                }
            }
        }
        if (c == 0 && g == 0) {
            if (p.nbt) *p.nbt += 1;
            if (rbn && p.rnbt) *p.rnbt += 1;
        }
    }
    const f32x4 ga = *(const f32x4*)(p.gamma + c), be = *(const f32x4*)(p.beta + c);
    const f32x4 rga = rbn ? *(const f32x4*)(p.rgamma + c) : zero4, rbe = rbn ? *(const f32x4*)(p.rbeta + c) : zero4;
#pragma unroll 4
    for (int r = t; r < p.rows_per_group; r += 256) {
        const f32x4 v = *(const f32x4*)(p.x + (row0 + r) * p.ldx + c);
        f32x4 o = (v - mu) * rs * ga + be;
        if (p.res) {
            f32x4 rv = *(const f32x4*)(p.res + (row0 + r) * p.ldr + c);
            if (rbn) rv = (rv - rmu) * rrs * rga + rbe;
            o += rv;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = act_f(o[e], p.act, p.slope);
        float* yp = p.y + (row0 + r) * p.ldy + c;
        if (p.y_scalar) { yp[0] = o[0]; yp[1] = o[1]; yp[2] = o[2]; yp[3] = o[3]; }
        else *(f32x4*)yp = o;
    }
}

int g_bn_fwd_small_rows = 512;     // (the head's 105 / 480-row layers: 5-6 us against ~15; trunk.7 / trunk.6 measured 10-18 / 29-47 us against ~15: not taken)

}  // namespace

extern "C" long long mft_bn_stats_ws_floats(int C, int rows_per_group, int n_groups) {
    return 2LL * n_groups * stats_chunks(rows_per_group, n_groups, C) * C;
}

void mft_bn_fwd_small_set_rows(int rows) { g_bn_fwd_small_rows = rows; }      // (C++ linkage; mft_debug_set_conv_tile(11000 + rows))

extern "C" int mft_bn_forward_small_max_rows(void) { return g_bn_fwd_small_rows; }

extern "C" int mft_bn_forward_small(const MftBnFwdJob* jb, void* stream) {
    if (jb == nullptr || jb->C % 4 != 0 || jb->ldx % 4 != 0 || jb->ldy % 4 != 0 || jb->rows_per_group < 1 || jb->n_groups < 1 ||
        jb->x == nullptr || jb->y == nullptr || jb->mean == nullptr || jb->rstd == nullptr || (jb->res && jb->ldr % 4 != 0) ||
        (jb->num_batches_tracked && !jb->running_mean) || (jb->res_gamma && (!jb->res || !jb->res_beta || !jb->res_mean || !jb->res_rstd)) ||
        jb->rows_per_group > 65536)
        return MFT_EINVAL;
    BnFwdSmallArgs p;
    p.x = jb->x; p.y = jb->y; p.gamma = jb->gamma; p.beta = jb->beta; p.mean = jb->mean; p.rstd = jb->rstd;
    p.running_mean = jb->running_mean; p.running_var = jb->running_var; p.nbt = jb->num_batches_tracked;
    p.res = jb->res; p.rgamma = jb->res_gamma; p.rbeta = jb->res_beta; p.rmean = jb->res_mean; p.rrstd = jb->res_rstd;
    p.rrunning_mean = jb->res_running_mean; p.rrunning_var = jb->res_running_var; p.rnbt = jb->res_num_batches_tracked;
    p.ldx = jb->ldx; p.ldy = jb->ldy; p.ldr = jb->ldr; p.C = jb->C; p.rows_per_group = jb->rows_per_group; p.n_groups = jb->n_groups;
    p.act = jb->act; p.eps = jb->eps; p.momentum = jb->momentum; p.slope = jb->slope;
    p.y_scalar = (((unsigned long long)jb->y) & 15) != 0 ? 1 : 0;
    hipLaunchKernelGGL(bn_forward_small_kernel, dim3(jb->C / 4, jb->n_groups), dim3(256), 0, (hipStream_t)stream, p);
    return mft_launch_status();
}

extern "C" int mft_bn_stats(const float* x, int ldx, int C, int rows_per_group, int n_groups, float eps,
                            float* mean, float* rstd, float* ws, float* running_mean, float* running_var,
                            float momentum, long long* num_batches_tracked, void* stream) {
    if (C % 4 != 0 || ldx % 4 != 0 || rows_per_group <= 0 || n_groups <= 0) return MFT_EINVAL;
    if (num_batches_tracked && !running_mean) return MFT_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const int chunks = stats_chunks(rows_per_group, n_groups, C);
    const int rpc = (rows_per_group + chunks - 1) / chunks;
    dim3 grid(chunks, (C + 63) / 64, n_groups);
    hipLaunchKernelGGL(bn_stats_partial, grid, dim3(256), 0, s, x, ldx, C, rows_per_group, rpc, chunks, ws);
    if (n_groups == 1 && chunks >= 128)
        hipLaunchKernelGGL(bn_stats_finalize<64>, dim3((C + 3) / 4, 1, 1), dim3(256), 0, s, x, ldx, C, rows_per_group, chunks, ws, eps,
                           mean, rstd, running_mean, running_var, momentum, num_batches_tracked);
    else
        hipLaunchKernelGGL(bn_stats_finalize<16>, dim3((C + 15) / 16, n_groups, 1), dim3(256), 0, s, x, ldx, C, rows_per_group, chunks, ws,
                           eps, mean, rstd, running_mean, running_var, momentum, num_batches_tracked);
    return mft_launch_status();
}

extern "C" int mft_bn_stats_multi(const MftBnStatsJob* jobs, int n_jobs, void* stream) {
    if (jobs == nullptr || n_jobs < 1 || n_jobs > BS_MULTI) return MFT_EINVAL;
    StatsMultiArgs a = {};
    int bp = 0, bf = 0;
    for (int j = 0; j < n_jobs; ++j) {
        const MftBnStatsJob& jb = jobs[j];
        if (jb.C % 4 != 0 || jb.ldx % 4 != 0 || jb.rows_per_group <= 0 || jb.n_groups <= 0 || jb.ws == nullptr) return MFT_EINVAL;
        if (jb.num_batches_tracked && !jb.running_mean) return MFT_EINVAL;
        StatsJob& p = a.job[j];
        p.x = jb.x; p.mean = jb.mean; p.rstd = jb.rstd; p.ws = jb.ws; p.running_mean = jb.running_mean; p.running_var = jb.running_var;
        p.nbt = jb.num_batches_tracked; p.ldx = jb.ldx; p.C = jb.C; p.rows_per_group = jb.rows_per_group; p.n_groups = jb.n_groups;
        p.eps = jb.eps; p.momentum = jb.momentum;
        p.chunks = stats_chunks(jb.rows_per_group, jb.n_groups, jb.C);
        p.rows_per_chunk = (jb.rows_per_group + p.chunks - 1) / p.chunks;
        p.lpc64 = (jb.n_groups == 1 && p.chunks >= 128) ? 1 : 0;
        a.start_p[j] = bp; a.start_f[j] = bf;
        bp += p.chunks * ((jb.C + 63) / 64) * jb.n_groups;
        bf += p.lpc64 ? (jb.C + 3) / 4 : ((jb.C + 15) / 16) * jb.n_groups;
    }
    a.start_p[n_jobs] = bp; a.start_f[n_jobs] = bf; a.n = n_jobs;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(bn_stats_partial_multi, dim3(bp), dim3(256), 0, s, a);
    hipLaunchKernelGGL(bn_stats_finalize_multi, dim3(bf), dim3(256), 0, s, a);
    return mft_launch_status();
}

extern "C" int mft_bn_apply(const float* x, int ldx, float* y, int ldy, int C, int rows_per_group, int n_groups,
                            const float* mean, const float* rstd, const float* gamma, const float* beta,
                            long long gb_group_stride, const float* res, int ldr, const float* res_mean,
                            const float* res_rstd, const float* res_gamma, const float* res_beta, int act, float slope,
                            void* stream) {
    if (C % 4 != 0 || ldx % 4 != 0 || ldy % 4 != 0 || (res && ldr % 4 != 0)) return MFT_EINVAL;
    ApplyArgs p{x, y, ldx, ldy, C, rows_per_group, n_groups, mean, rstd, gamma, beta, gb_group_stride,
                res, ldr, res_mean, res_rstd, res_gamma, res_beta, act, slope, nullptr, 0};
    const long long total = (long long)n_groups * rows_per_group * (C / 4);
    hipLaunchKernelGGL(bn_apply_kernel<false>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, p);
    return mft_launch_status();
}

extern "C" int mft_bn_apply_multi(const MftBnApplyJob* jobs, int n_jobs, void* stream) {
    if (jobs == nullptr || n_jobs < 1) return MFT_EINVAL;
    for (int j0 = 0; j0 < n_jobs;) {
        ApplyMultiArgs a = {};
        int cnt = 0, blocks = 0;
        for (; j0 < n_jobs && cnt < BA_MULTI; ++j0, ++cnt) {
            const MftBnApplyJob& jb = jobs[j0];
            if (jb.C % 4 != 0 || jb.ldx % 4 != 0 || jb.ldy % 4 != 0 || jb.rows_per_group < 1 || jb.n_groups < 1) return MFT_EINVAL;
            a.job[cnt] = ApplyArgs{jb.x, jb.y, jb.ldx, jb.ldy, jb.C, jb.rows_per_group, jb.n_groups, jb.mean, jb.rstd, jb.gamma, jb.beta, 0,
                                   nullptr, 0, nullptr, nullptr, nullptr, nullptr, jb.act, jb.slope, nullptr, 0};
            a.start[cnt] = blocks;
            blocks += grid_for((long long)jb.n_groups * jb.rows_per_group * (jb.C / 4));
        }
        a.start[cnt] = blocks;
        a.n = cnt;
        hipLaunchKernelGGL(bn_apply_multi_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
    }
    return mft_launch_status();
}

extern "C" int mft_bn_apply_x3ws(const float* x, int ldx, float* y, int ldy, int C, int rows_per_group, int n_groups, const float* ws,
                                 const float* gamma, const float* beta, const float* res, int ldr, const float* res_ws,
                                 const float* res_gamma, const float* res_beta, int act, float slope, float eps, float* mean,
                                 float* rstd, float* res_mean, float* res_rstd, void* stream) {
    if (C % 4 != 0 || C > 1024 || ldx % 4 != 0 || ldy % 4 != 0 || (res && ldr % 4 != 0) || !ws || !y || rows_per_group < 128 ||
        n_groups <= 0 || (res_ws && (!res || !res_gamma || !res_beta)) || (mean == nullptr) != (rstd == nullptr) ||
        (res_mean == nullptr) != (res_rstd == nullptr))
        return MFT_EINVAL;
    const long long M = (long long)n_groups * rows_per_group;
    const long long per_group = (long long)rows_per_group * (C / 4);
    if (M > 0x7fffffffLL) return MFT_EINVAL;
    const int max_tiles = mft_x3_max_group_tiles(rows_per_group, 128);
    const size_t lds = (size_t)4 * C * sizeof(float) + (res_ws ? 2 : 1) * mft_x3_stage_bytes_host(max_tiles, C);
    if (lds > 64 * 1024) return MFT_EINVAL;          // very long groups (one BatchNorm batch of thousands of rows): finalize + mft_bn_apply
    int sub = (int)((per_group + 256 * 16 - 1) / (256 * 16));         // ~16 float4 per thread: the prologue is repeated by every sub-block
    const int cap = (2048 + n_groups - 1) / n_groups;
    if (sub > cap) sub = cap;
    if (sub < 1) sub = 1;
    ApplyWsArgs p{x, y, ldx, ldy, C, rows_per_group, n_groups, (int)M, 128, ws, gamma, beta, res, ldr, res_ws, res_gamma, res_beta,
                  act, slope, eps, mean, rstd, res_mean, res_rstd, max_tiles};
    hipLaunchKernelGGL(bn_apply_ws_kernel, dim3(sub, n_groups), dim3(256), lds, (hipStream_t)stream, p);
    return mft_launch_status();
}

// BatchNorm running statistics after a SEQUENCE of train-mode forwards whose batch statistics were computed in grouped launches
// (the meta-fine-tuning inner loop: the frozen trunk of all ~105 mini-batches of an episode runs as two grouped passes, full and
// ragged mini-batches; gnnnet.py:126-177 runs them one by one): step t used group (order[t] & 0xffffff) of statistics set
// (order[t] >> 24), and torch updates running = (1 - momentum) * running + momentum * batch value, variance unbiased, in step order.
namespace {
__global__ void bn_running_ema_kernel(const float* __restrict__ mean_a, const float* __restrict__ rstd_a, float unb_a,
                                      const float* __restrict__ mean_b, const float* __restrict__ rstd_b, float unb_b,
                                      const int* __restrict__ order, int n_steps, int C, float eps, float momentum,
                                      float* __restrict__ running_mean, float* __restrict__ running_var) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float rm = running_mean[c], rv = running_var[c];
    for (int t = 0; t < n_steps; ++t) {
        const int o = order[t], g = o & 0xffffff;
        const bool b = (o >> 24) != 0;
        const float m = (b ? mean_b : mean_a)[(long long)g * C + c];
        const float rs = (b ? rstd_b : rstd_a)[(long long)g * C + c];
        const float var = fmaxf(1.0f / (rs * rs) - eps, 0.f);
        rm = (1.f - momentum) * rm + momentum * m;
        rv = (1.f - momentum) * rv + momentum * (var * (b ? unb_b : unb_a));
    }
    running_mean[c] = rm;
    running_var[c] = rv;
}
}  // namespace

extern "C" int mft_bn_running_ema(const float* mean_a, const float* rstd_a, int rows_a, const float* mean_b, const float* rstd_b,
                                  int rows_b, const int* order, int n_steps, int C, float eps, float momentum, float* running_mean,
                                  float* running_var, void* stream) {
    if (!mean_a || !rstd_a || rows_a < 1 || !order || n_steps < 1 || C < 1 || !running_mean || !running_var) return MFT_EINVAL;
    const float ua = (float)rows_a / (float)(rows_a > 1 ? rows_a - 1 : 1);
    const float ub = (float)rows_b / (float)(rows_b > 1 ? rows_b - 1 : 1);
    hipLaunchKernelGGL(bn_running_ema_kernel, dim3((C + 63) / 64), dim3(64), 0, (hipStream_t)stream, mean_a, rstd_a, ua, mean_b, rstd_b, ub,
                       order, n_steps, C, eps, momentum, running_mean, running_var);
    return mft_launch_status();
}

extern "C" int mft_bn_apply_x3ws_fits(int C, int rows_per_group, int with_res_bn) {
    if (C % 4 != 0 || C > 1024 || rows_per_group < 128) return 0;
    return (size_t)4 * C * sizeof(float) + (with_res_bn ? 2 : 1) * mft_x3_stage_bytes_host(mft_x3_max_group_tiles(rows_per_group, 128), C) <=
           64 * 1024;
}

extern "C" int mft_bn_apply_planes(const float* x, int ldx, float* y, int ldy, unsigned short* planes, long long plane_stride,
                                   int C, int rows_per_group, int n_groups, const float* mean, const float* rstd,
                                   const float* gamma, const float* beta, long long gb_group_stride, const float* res, int ldr,
                                   const float* res_mean, const float* res_rstd, const float* res_gamma,
                                   const float* res_beta, int act, float slope, void* stream) {
    if (C % 4 != 0 || ldx % 4 != 0 || (y && ldy % 4 != 0) || (res && ldr % 4 != 0) || !planes) return MFT_EINVAL;
    if (plane_stride < (long long)n_groups * rows_per_group * C) return MFT_EINVAL;
    ApplyArgs p{x, y, ldx, ldy, C, rows_per_group, n_groups, mean, rstd, gamma, beta, gb_group_stride,
                res, ldr, res_mean, res_rstd, res_gamma, res_beta, act, slope, planes, plane_stride};
    const long long total = (long long)n_groups * rows_per_group * (C / 4);
    hipLaunchKernelGGL(bn_apply_kernel<true>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, p);
    return mft_launch_status();
}

// ------------------------------------------------------------------- stem cache as per-window (max, min) pairs
// BN -> ReLU -> MaxPool(3,2,1) of a cached raw stem output depends on the mini-batch only through the per-channel affine
// f(v) = (v - mean) * rstd * gamma + beta, and every floating-point step of f is monotone in v: non-decreasing for gamma >= 0,
// non-increasing for gamma < 0.  Hence max_window relu(f(v)) = relu(f(max_window v)) (gamma >= 0) or relu(f(min_window v))
// (gamma < 0) EXACTLY, bit for bit -- the cache keeps the window maxima and minima of the raw convolution output
// (2 x 21 x 21 x 64 instead of 42 x 42 x 64 floats per image) and the per-step gather reads a quarter to a half of the bytes.
namespace {
__global__ __launch_bounds__(256) void pool_window_minmax_kernel(const float* __restrict__ x, float* __restrict__ ymax,
                                                                 float* __restrict__ ymin, long long n_img, int H, int W, int C,
                                                                 int OH, int OW) {
    const int cq = C >> 2;
    const long long total = n_img * OH * OW * cq;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % cq) * 4;
        long long t = i / cq;
        const int ow = (int)(t % OW); t /= OW;
        const int oh = (int)(t % OH);
        const long long n = t / OH;
        f32x4 hi = {-3.4e38f, -3.4e38f, -3.4e38f, -3.4e38f}, lo = {3.4e38f, 3.4e38f, 3.4e38f, 3.4e38f};
#pragma unroll
        for (int dh = 0; dh < 3; ++dh) {
            const int ih = oh * 2 - 1 + dh;
            if (ih < 0 || ih >= H) continue;
#pragma unroll
            for (int dw = 0; dw < 3; ++dw) {
                const int iw = ow * 2 - 1 + dw;
                if (iw < 0 || iw >= W) continue;
                const f32x4 v = *(const f32x4*)(x + ((n * H + ih) * W + iw) * C + c);
#pragma unroll
                for (int e = 0; e < 4; ++e) { hi[e] = fmaxf(hi[e], v[e]); lo[e] = fminf(lo[e], v[e]); }
            }
        }
        *(f32x4*)(ymax + i * 4) = hi;
        *(f32x4*)(ymin + i * 4) = lo;
    }
}

__device__ __forceinline__ f32x4 pooled_bn_relu(const float* __restrict__ pmax, const float* __restrict__ pmin, long long src,
                                                const f32x4 mu, const f32x4 rs, const f32x4 ga, const f32x4 be) {
    const bool all_pos = ga[0] >= 0.f && ga[1] >= 0.f && ga[2] >= 0.f && ga[3] >= 0.f;
    const bool all_neg = ga[0] < 0.f && ga[1] < 0.f && ga[2] < 0.f && ga[3] < 0.f;
    f32x4 v;
    if (all_pos) v = *(const f32x4*)(pmax + src);
    else if (all_neg) v = *(const f32x4*)(pmin + src);
    else {
        const f32x4 a = *(const f32x4*)(pmax + src), b = *(const f32x4*)(pmin + src);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = ga[e] >= 0.f ? a[e] : b[e];
    }
    f32x4 o = __builtin_elementwise_fma((v - mu) * rs, ga, be);
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = fmaxf(o[e], 0.f);
    return o;
}

__global__ __launch_bounds__(256) void bn_relu_pooled_gather_kernel(const float* __restrict__ pmax, const float* __restrict__ pmin,
                                                                    const int* __restrict__ src_idx, float* __restrict__ y,
                                                                    int n_img, int HW, int C, int imgs_per_group,
                                                                    const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                    const float* __restrict__ gamma, const float* __restrict__ beta) {
    const int cq = C >> 2;
    const long long total = (long long)n_img * HW * cq;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % cq) * 4;
        const long long t = i / cq;
        const int pix = (int)(t % HW);
        const int n = (int)(t / HW);
        const int g = n / imgs_per_group;
        const long long src = ((long long)src_idx[n] * HW + pix) * C + c;
        const f32x4 mu = *(const f32x4*)(mean + (long long)g * C + c);
        const f32x4 rs = *(const f32x4*)(rstd + (long long)g * C + c);
        *(f32x4*)(y + i * 4) = pooled_bn_relu(pmax, pmin, src, mu, rs, *(const f32x4*)(gamma + c), *(const f32x4*)(beta + c));
    }
}

// the same gather with the batch statistics combined in the prologue of every workgroup (grid (sub-blocks, groups)) from the
// cached per-image moments: no bn_combine_moments launch in front of it
__global__ __launch_bounds__(256) void bn_relu_pooled_gather_moments_kernel(
    const float* __restrict__ pmax, const float* __restrict__ pmin, const int* __restrict__ src_idx, float* __restrict__ y, int HW,
    int C, int imgs_per_group, const float* __restrict__ mean_img, const float* __restrict__ m2_img, int rows_per_img, float eps,
    const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ mean, float* __restrict__ rstd) {
    extern __shared__ __attribute__((aligned(16))) float tab[];          // [2][C]: mean, rstd of this group; [2][imgs][C] staged moments
    const int g = blockIdx.y;
    const int* id = src_idx + (long long)g * imgs_per_group;
    float* smean = tab + 2 * C;
    float* sm2 = smean + imgs_per_group * C;
    for (int i = threadIdx.x; i < imgs_per_group * C; i += 256) {         // independent loads first, the arithmetic runs out of LDS
        const int im = i / C, c = i - im * C;
        const long long src = (long long)id[im] * C + c;
        smean[i] = mean_img[src];
        sm2[i] = m2_img[src];
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        float mu, rs;
        combine_moments_lds(smean, sm2, C, c, rows_per_img, imgs_per_group, eps, mu, rs);
        tab[c] = mu; tab[C + c] = rs;
        if (blockIdx.x == 0 && mean) { mean[(long long)g * C + c] = mu; rstd[(long long)g * C + c] = rs; }
    }
    __syncthreads();
    const int cq = C >> 2;
    const int total = imgs_per_group * HW * cq;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int c = (i % cq) * 4;
        const int t = i / cq;
        const int pix = t % HW;
        const int n = t / HW;
        const long long src = ((long long)id[n] * HW + pix) * C + c;
        const long long dst = ((long long)g * imgs_per_group * HW * cq + i) * 4;
        *(f32x4*)(y + dst) = pooled_bn_relu(pmax, pmin, src, *(const f32x4*)(tab + c), *(const f32x4*)(tab + C + c),
                                            *(const f32x4*)(gamma + c), *(const f32x4*)(beta + c));
    }
}
}  // namespace

extern "C" int mft_pool_window_minmax(const float* x, float* ymax, float* ymin, long long n_img, int H, int W, int C, void* stream) {
    if (C % 4 != 0 || n_img <= 0) return MFT_EINVAL;
    const int OH = (H + 2 - 3) / 2 + 1, OW = (W + 2 - 3) / 2 + 1;
    const long long total = n_img * OH * OW * (C / 4);
    hipLaunchKernelGGL(pool_window_minmax_kernel, dim3(grid_for(total, 256, 256 * 16)), dim3(256), 0, (hipStream_t)stream, x, ymax,
                       ymin, n_img, H, W, C, OH, OW);
    return mft_launch_status();
}

extern "C" int mft_bn_relu_pooled_gather(const float* pmax, const float* pmin, const int* src_idx, float* y, int n_img, int OH,
                                         int OW, int C, int imgs_per_group, const float* mean, const float* rstd,
                                         const float* gamma, const float* beta, void* stream) {
    if (C % 4 != 0 || !src_idx) return MFT_EINVAL;
    if (imgs_per_group <= 0) imgs_per_group = n_img;
    const long long total = (long long)n_img * OH * OW * (C / 4);
    hipLaunchKernelGGL(bn_relu_pooled_gather_kernel, dim3(grid_for(total, 256, 256 * 16)), dim3(256), 0, (hipStream_t)stream, pmax,
                       pmin, src_idx, y, n_img, OH * OW, C, imgs_per_group, mean, rstd, gamma, beta);
    return mft_launch_status();
}

extern "C" int mft_bn_relu_pooled_gather_moments(const float* pmax, const float* pmin, const int* src_idx, float* y, int n_img,
                                                 int OH, int OW, int C, int imgs_per_group, const float* mean_img,
                                                 const float* m2_img, int rows_per_img, float eps, const float* gamma,
                                                 const float* beta, float* mean, float* rstd, void* stream) {
    if (C % 4 != 0 || C > 1024 || !src_idx || imgs_per_group <= 0 || n_img % imgs_per_group != 0 || rows_per_img <= 0 ||
        (mean == nullptr) != (rstd == nullptr))
        return MFT_EINVAL;
    const int groups = n_img / imgs_per_group;
    const long long per_group = (long long)imgs_per_group * OH * OW * (C / 4);
    if (per_group > 0x7fffffffLL) return MFT_EINVAL;
    int sub = (int)((per_group + 256 * 16 - 1) / (256 * 16));         // ~16 float4 per thread: the prologue is repeated by every sub-block
    const int cap = (2048 + groups - 1) / groups;
    if (sub > cap) sub = cap;
    if (sub < 1) sub = 1;
    const size_t lds = (size_t)(2 + 2 * imgs_per_group) * C * sizeof(float);
    if (lds > 64 * 1024) return MFT_EINVAL;
    hipLaunchKernelGGL(bn_relu_pooled_gather_moments_kernel, dim3(sub, groups), dim3(256), lds, (hipStream_t)stream,
                       pmax, pmin, src_idx, y, OH * OW, C, imgs_per_group, mean_img, m2_img, rows_per_img, eps, gamma, beta, mean,
                       rstd);
    return mft_launch_status();
}

extern "C" int mft_bn_relu_maxpool_gather_planes(const float* x, const int* src_idx, float* y, unsigned short* planes,
                                                 long long plane_stride, int n_img, int H, int W, int C, int imgs_per_group,
                                                 const float* mean, const float* rstd, const float* gamma, const float* beta,
                                                 void* stream) {
    if (C % 4 != 0) return MFT_EINVAL;
    if (imgs_per_group <= 0) imgs_per_group = n_img;
    const int OH = (H + 2 - 3) / 2 + 1, OW = (W + 2 - 3) / 2 + 1;
    const long long total = (long long)n_img * OH * OW * (C / 4);
    if (planes)
        hipLaunchKernelGGL(bn_relu_maxpool_kernel<true>, dim3(grid_for(total, 256, 256 * 16)), dim3(256), 0, (hipStream_t)stream,
                           x, y, n_img, H, W, C, OH, OW, imgs_per_group, mean, rstd, gamma, beta, src_idx, planes, plane_stride);
    else
        hipLaunchKernelGGL(bn_relu_maxpool_kernel<false>, dim3(grid_for(total, 256, 256 * 16)), dim3(256), 0, (hipStream_t)stream,
                           x, y, n_img, H, W, C, OH, OW, imgs_per_group, mean, rstd, gamma, beta, src_idx, planes, plane_stride);
    return mft_launch_status();
}

extern "C" int mft_bn_relu_maxpool_gather(const float* x, const int* src_idx, float* y, int n_img, int H, int W, int C,
                                          int imgs_per_group, const float* mean, const float* rstd, const float* gamma,
                                          const float* beta, void* stream) {
    return mft_bn_relu_maxpool_gather_planes(x, src_idx, y, nullptr, 0, n_img, H, W, C, imgs_per_group, mean, rstd, gamma, beta,
                                             stream);
}

extern "C" int mft_bn_relu_maxpool(const float* x, float* y, int n_img, int H, int W, int C, int imgs_per_group,
                                   const float* mean, const float* rstd, const float* gamma, const float* beta,
                                   void* stream) {
    return mft_bn_relu_maxpool_gather(x, nullptr, y, n_img, H, W, C, imgs_per_group, mean, rstd, gamma, beta, stream);
}

extern "C" int mft_bn_image_moments(const float* x, int ldx, int C, int rows_per_img, long long n_img, float* mean_img,
                                    float* m2_img, void* stream) {
    if (C % 4 != 0 || ldx % 4 != 0 || rows_per_img <= 0 || n_img <= 0 || n_img > 0x7fffffffLL) return MFT_EINVAL;
    dim3 grid((unsigned)n_img, (C + 63) / 64, 1);
    hipLaunchKernelGGL(bn_image_moments_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, ldx, C, rows_per_img,
                       mean_img, m2_img);
    return mft_launch_status();
}

extern "C" int mft_bn_combine_moments(const float* mean_img, const float* m2_img, const int* idx, int C,
                                      int rows_per_img, int imgs_per_group, int n_groups, float eps, float* mean,
                                      float* rstd, void* stream) {
    if (imgs_per_group <= 0 || n_groups <= 0 || rows_per_img <= 0) return MFT_EINVAL;
    dim3 grid((C + 63) / 64, n_groups, 1);
    hipLaunchKernelGGL(bn_combine_moments_kernel, grid, dim3(64), 0, (hipStream_t)stream, mean_img, m2_img, idx, C,
                       rows_per_img, imgs_per_group, eps, mean, rstd);
    return mft_launch_status();
}

extern "C" int mft_global_avgpool(const float* x, float* y, int n_img, int HW, int C, void* stream) {
    if (C % 4 != 0) return MFT_EINVAL;
    const long long total = (long long)n_img * (C / 4);
    hipLaunchKernelGGL(global_avgpool_kernel, dim3((int)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, y,
                       n_img, HW, C);
    return mft_launch_status();
}

extern "C" int mft_avgpool_relu_backward(const float* dfeat, const float* out, float* dout, int n_img, int HW, int C,
                                         void* stream) {
    if (C % 4 != 0) return MFT_EINVAL;
    const long long total = (long long)n_img * HW * (C / 4);
    hipLaunchKernelGGL(avgpool_relu_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, dfeat, out,
                       dout, n_img, HW, C);
    return mft_launch_status();
}

extern "C" int mft_bn_backward(const float* x, int ldx, const float* dy, int lddy, const float* relu_out, int ldro,
                               float* dx, int lddx, int C, int rows_per_group, int n_groups, const float* mean,
                               const float* rstd, const float* gamma, long long gb_group_stride, float* dgamma,
                               float* dbeta, void* stream) {
    if (C % 4 != 0 || ldx % 4 != 0 || lddy % 4 != 0) return MFT_EINVAL;
    BwdArgs p{x, dy, relu_out, dx, ldx, lddy, ldro, lddx, C, rows_per_group, mean, rstd, gamma, gb_group_stride,
              dgamma, dbeta};
    dim3 grid((C + 63) / 64, n_groups, 1);
    hipLaunchKernelGGL(bn_backward_kernel, grid, dim3(256), 0, (hipStream_t)stream, p);
    return mft_launch_status();
}

// ------------------------------------------------------------------------------- fused small-group kernels
// The adapted last block sees 45 rows per episode (5 images x 3x3).  Its BatchNorm work was 2 + 1 launches per
// normalisation (partial sums, finalize, apply) plus pooling -- each a few microseconds of arithmetic behind ~10 us of
// dependent-launch latency on the critical stream.  With <= 64 rows per group a workgroup can hold its (group, 64-channel)
// tile in registers: statistics, normalisation, residual (optionally with its own BatchNorm), activation and the global
// average pool become ONE pass over the data.
namespace {

struct SmallFwdArgs {
    const float* x1; int ld1;
    const float* x2; int ld2;            // optional second branch with its own BatchNorm (shortcut)
    const float* res; int ldr;           // optional plain residual
    float* y; int ldy;
    int C, rows, n_groups;
    const float* g1; const float* b1; const float* g2; const float* b2; long long gbs;
    float* mean1; float* rstd1; float* mean2; float* rstd2;
    int act; float slope; float eps;
    float* pooled; int hw;               // optional [n_groups * rows/hw, C] mean over each image's hw rows
};

constexpr int SM_RPT = 4;                // rows per thread: 16 row lanes x 4 = 64 rows max

__device__ __forceinline__ void small_stats(const f32x4 (&v)[SM_RPT], int rl, int cq, int rows, f32x4 (*red1)[16],
                                            f32x4 (*red2)[16], f32x4& mean, f32x4& rstd, float eps) {
    // shifted one-pass moments, fixed reduction order (same scheme as bn_stats_partial/finalize)
    __shared__ f32x4 s_shift[16];
    if (rl == 0) s_shift[cq] = v[0];
    __syncthreads();
    const f32x4 sh = s_shift[cq];
    f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = s1;
#pragma unroll
    for (int k = 0; k < SM_RPT; ++k)
        if (rl + 16 * k < rows) {
            const f32x4 d = v[k] - sh;
            s1 += d;
            s2 += d * d;
        }
    red1[rl][cq] = s1;
    red2[rl][cq] = s2;
    __syncthreads();
    s1 = red1[0][cq];
    s2 = red2[0][cq];
#pragma unroll
    for (int k = 1; k < ST_ROWS; ++k) {
        s1 += red1[k][cq];
        s2 += red2[k][cq];
    }
    const float inv = 1.f / (float)rows;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float d = s1[e] * inv;
        mean[e] = sh[e] + d;
        rstd[e] = 1.0f / sqrtf(fmaxf(s2[e] * inv - d * d, 0.f) + eps);
    }
    __syncthreads();                      // red1/red2/s_shift may be reused
}

__global__ __launch_bounds__(256) void bn_small_forward_kernel(SmallFwdArgs p) {
    const int cq = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int c = blockIdx.x * 64 + cq * 4;
    const int g = blockIdx.y;
    const long long row0 = (long long)g * p.rows;
    __shared__ f32x4 red1[ST_ROWS][16];
    __shared__ f32x4 red2[ST_ROWS][16];
    __shared__ f32x4 tile[64][16];
    if (c >= p.C) return;                 // C is a multiple of 64 on this path (checked by the launcher)
    f32x4 v1[SM_RPT], v2[SM_RPT];
#pragma unroll
    for (int k = 0; k < SM_RPT; ++k) {
        const int r = rl + 16 * k;
        v1[k] = f32x4{0.f, 0.f, 0.f, 0.f};
        v2[k] = v1[k];
        if (r < p.rows) {
            v1[k] = *(const f32x4*)(p.x1 + (row0 + r) * p.ld1 + c);
            if (p.x2) v2[k] = *(const f32x4*)(p.x2 + (row0 + r) * p.ld2 + c);
        }
    }
    f32x4 m1, r1, m2 = {0.f, 0.f, 0.f, 0.f}, r2 = m2;
    small_stats(v1, rl, cq, p.rows, red1, red2, m1, r1, p.eps);
    if (p.x2) small_stats(v2, rl, cq, p.rows, red1, red2, m2, r2, p.eps);
    if (rl == 0) {
        *(f32x4*)(p.mean1 + (long long)g * p.C + c) = m1;
        *(f32x4*)(p.rstd1 + (long long)g * p.C + c) = r1;
        if (p.x2) {
            *(f32x4*)(p.mean2 + (long long)g * p.C + c) = m2;
            *(f32x4*)(p.rstd2 + (long long)g * p.C + c) = r2;
        }
    }
    const f32x4 ga1 = *(const f32x4*)(p.g1 + g * p.gbs + c), be1 = *(const f32x4*)(p.b1 + g * p.gbs + c);
    f32x4 ga2 = {0.f, 0.f, 0.f, 0.f}, be2 = ga2;
    if (p.x2) {
        ga2 = *(const f32x4*)(p.g2 + g * p.gbs + c);
        be2 = *(const f32x4*)(p.b2 + g * p.gbs + c);
    }
#pragma unroll
    for (int k = 0; k < SM_RPT; ++k) {
        const int r = rl + 16 * k;
        if (r >= p.rows) continue;
        f32x4 o = (v1[k] - m1) * r1 * ga1 + be1;
        if (p.x2) o += (v2[k] - m2) * r2 * ga2 + be2;
        if (p.res) o += *(const f32x4*)(p.res + (row0 + r) * p.ldr + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = act_f(o[e], p.act, p.slope);
        *(f32x4*)(p.y + (row0 + r) * p.ldy + c) = o;
        if (p.pooled) tile[r][cq] = o;
    }
    if (p.pooled) {
        __syncthreads();
        const int n_img = p.rows / p.hw;
        if (rl < n_img) {
            f32x4 s = {0.f, 0.f, 0.f, 0.f};
            for (int k = 0; k < p.hw; ++k) s += tile[rl * p.hw + k][cq];
            const float inv = 1.f / (float)p.hw;
            *(f32x4*)(p.pooled + ((long long)g * n_img + rl) * p.C + c) = s * inv;
        }
    }
}

// two BatchNorm backward passes that share dy (main branch + shortcut branch of a residual block)
__global__ __launch_bounds__(256) void bn_backward2_kernel(BwdArgs a, BwdArgs b) {
    for (int which = 0; which < 2; ++which) {
        const BwdArgs& p = which ? b : a;
        const int cq = threadIdx.x & 15;
        const int rl = threadIdx.x >> 4;
        const int c = blockIdx.x * 64 + cq * 4;
        const int g = blockIdx.y;
        const long long row0 = (long long)g * p.rows_per_group;
        __shared__ f32x4 red1[ST_ROWS][16];
        __shared__ f32x4 red2[ST_ROWS][16];
        const f32x4 mu = *(const f32x4*)(p.mean + (long long)g * p.C + c);
        const f32x4 rs = *(const f32x4*)(p.rstd + (long long)g * p.C + c);
        const f32x4 ga = *(const f32x4*)(p.gamma + g * p.gbs + c);
        f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = s1;
        for (int rr = rl; rr < p.rows_per_group; rr += ST_ROWS) {
            const f32x4 d = *(const f32x4*)(p.dy + (row0 + rr) * p.lddy + c);
            const f32x4 xh = (*(const f32x4*)(p.x + (row0 + rr) * p.ldx + c) - mu) * rs;
            s1 += d;
            s2 += d * xh;
        }
        red1[rl][cq] = s1;
        red2[rl][cq] = s2;
        __syncthreads();
        s1 = red1[0][cq];
        s2 = red2[0][cq];
#pragma unroll
        for (int k = 1; k < ST_ROWS; ++k) {
            s1 += red1[k][cq];
            s2 += red2[k][cq];
        }
        if (rl == 0) {
            *(f32x4*)(p.dgamma + (long long)g * p.C + c) = s2;
            *(f32x4*)(p.dbeta + (long long)g * p.C + c) = s1;
        }
        const float inv = 1.f / (float)p.rows_per_group;
        const f32x4 m1 = s1 * inv, m2 = s2 * inv;
        const f32x4 k = ga * rs;
        for (int rr = rl; rr < p.rows_per_group; rr += ST_ROWS) {
            const f32x4 d = *(const f32x4*)(p.dy + (row0 + rr) * p.lddy + c);
            const f32x4 xh = (*(const f32x4*)(p.x + (row0 + rr) * p.ldx + c) - mu) * rs;
            *(f32x4*)(p.dx + (row0 + rr) * p.lddx + c) = k * (d - m1 - xh * m2);
        }
        __syncthreads();
    }
}

// bn_backward2 with the incoming gradient computed on the fly from the loss: the inner-loop loss is the cross entropy on the
// pooled feature (finetune.py:286-293), so dy[row][c] = out > 0 ? (softmax(feat[img])[c] - onehot[c]) / (k * hw) : 0 needs only
// the k log-sum-exps of the group (recomputed by each of the C/64 workgroups of a group: k x C floats) -- d_out is never written.
// Same arithmetic, in the same order, as ce_pool_backward_kernel followed by bn_backward2_kernel.
struct CeArgs { const float* feat; const int* labels; const float* out; float* loss; int k, hw; };

__global__ __launch_bounds__(256) void bn_backward2_ce_kernel(BwdArgs a, BwdArgs b, CeArgs ce) {
    const int g = blockIdx.y;
    const int C = a.C;
    __shared__ float s_lse[16];
    __shared__ float s_loss[16];
    __shared__ int s_lab[16];
    {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        for (int r = wave; r < ce.k; r += 4) {
            const float* x = ce.feat + ((long long)g * ce.k + r) * C;
            float mx = -3.4e38f;
            for (int c = lane; c < C; c += 64) mx = fmaxf(mx, x[c]);
            mx = wave_max(mx);
            float se = 0.f;
            for (int c = lane; c < C; c += 64) se += __expf(x[c] - mx);
            se = wave_sum(se);
            if (lane == 0) {
                const float lse = mx + __logf(se);
                const int y = ce.labels[(long long)g * ce.k + r];
                s_lse[r] = lse;
                s_lab[r] = y;
                s_loss[r] = lse - x[y];
            }
        }
        __syncthreads();
        if (threadIdx.x == 0 && blockIdx.x == 0 && ce.loss) {
            float s = 0.f;
            for (int r = 0; r < ce.k; ++r) s += s_loss[r];
            ce.loss[g] = s / (float)ce.k;
        }
    }
    const float inv_ce = 1.f / ((float)ce.k * (float)ce.hw);
    const int cq = threadIdx.x & 15;
    const int rl = threadIdx.x >> 4;
    const int c = blockIdx.x * 64 + cq * 4;
    const long long row0 = (long long)g * a.rows_per_group;
    auto dy_at = [&](int rr) {
        const int img = rr / ce.hw;
        const f32x4 o = *(const f32x4*)(ce.out + (row0 + rr) * C + c);
        const f32x4 x = *(const f32x4*)(ce.feat + ((long long)g * ce.k + img) * C + c);
        const int y = s_lab[img];
        const float lse = s_lse[img];
        f32x4 d;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float pr = __expf(x[e] - lse) - ((c + e) == y ? 1.f : 0.f);
            d[e] = o[e] > 0.f ? pr * inv_ce : 0.f;
        }
        return d;
    };
    for (int which = 0; which < 2; ++which) {
        const BwdArgs& p = which ? b : a;
        __shared__ f32x4 red1[ST_ROWS][16];
        __shared__ f32x4 red2[ST_ROWS][16];
        const f32x4 mu = *(const f32x4*)(p.mean + (long long)g * p.C + c);
        const f32x4 rs = *(const f32x4*)(p.rstd + (long long)g * p.C + c);
        const f32x4 ga = *(const f32x4*)(p.gamma + g * p.gbs + c);
        f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = s1;
        for (int rr = rl; rr < p.rows_per_group; rr += ST_ROWS) {
            const f32x4 d = dy_at(rr);
            const f32x4 xh = (*(const f32x4*)(p.x + (row0 + rr) * p.ldx + c) - mu) * rs;
            s1 += d;
            s2 += d * xh;
        }
        red1[rl][cq] = s1;
        red2[rl][cq] = s2;
        __syncthreads();
        s1 = red1[0][cq];
        s2 = red2[0][cq];
#pragma unroll
        for (int k = 1; k < ST_ROWS; ++k) {
            s1 += red1[k][cq];
            s2 += red2[k][cq];
        }
        if (rl == 0) {
            *(f32x4*)(p.dgamma + (long long)g * p.C + c) = s2;
            *(f32x4*)(p.dbeta + (long long)g * p.C + c) = s1;
        }
        const float inv = 1.f / (float)p.rows_per_group;
        const f32x4 m1 = s1 * inv, m2 = s2 * inv;
        const f32x4 k = ga * rs;
        for (int rr = rl; rr < p.rows_per_group; rr += ST_ROWS) {
            const f32x4 d = dy_at(rr);
            const f32x4 xh = (*(const f32x4*)(p.x + (row0 + rr) * p.ldx + c) - mu) * rs;
            *(f32x4*)(p.dx + (row0 + rr) * p.lddx + c) = k * (d - m1 - xh * m2);
        }
        __syncthreads();
    }
}

}  // namespace

extern "C" int mft_bn_small_forward(const float* x1, int ld1, const float* x2, int ld2, const float* res, int ldr, float* y,
                                    int ldy, int C, int rows_per_group, int n_groups, const float* gamma1,
                                    const float* beta1, const float* gamma2, const float* beta2,
                                    long long gb_group_stride, float* mean1, float* rstd1, float* mean2, float* rstd2,
                                    int act, float slope, float eps, float* pooled, int hw, void* stream) {
    if (C % 64 != 0 || rows_per_group < 1 || rows_per_group > 64 || n_groups < 1) return MFT_EINVAL;
    if ((ld1 | ldy) % 4 != 0 || (x2 && ld2 % 4 != 0) || (res && ldr % 4 != 0)) return MFT_EINVAL;
    if (pooled && (hw < 1 || rows_per_group % hw != 0 || rows_per_group / hw > 16)) return MFT_EINVAL;
    SmallFwdArgs p{x1, ld1, x2, ld2, res, ldr, y, ldy, C, rows_per_group, n_groups, gamma1, beta1, gamma2, beta2,
                   gb_group_stride, mean1, rstd1, mean2, rstd2, act, slope, eps, pooled, hw};
    hipLaunchKernelGGL(bn_small_forward_kernel, dim3(C / 64, n_groups), dim3(256), 0, (hipStream_t)stream, p);
    return mft_launch_status();
}

extern "C" int mft_bn_backward2(const float* xa, const float* xb, int ldx, const float* dy, int lddy, float* dxa, float* dxb,
                                int lddx, int C, int rows_per_group, int n_groups, const float* mean_a, const float* rstd_a,
                                const float* gamma_a, const float* mean_b, const float* rstd_b, const float* gamma_b,
                                long long gb_group_stride, float* dgamma_a, float* dbeta_a, float* dgamma_b, float* dbeta_b,
                                void* stream) {
    if (C % 64 != 0 || ldx % 4 != 0 || lddy % 4 != 0 || lddx % 4 != 0) return MFT_EINVAL;
    BwdArgs a{xa, dy, nullptr, dxa, ldx, lddy, 0, lddx, C, rows_per_group, mean_a, rstd_a, gamma_a, gb_group_stride, dgamma_a, dbeta_a};
    BwdArgs b{xb, dy, nullptr, dxb, ldx, lddy, 0, lddx, C, rows_per_group, mean_b, rstd_b, gamma_b, gb_group_stride, dgamma_b, dbeta_b};
    hipLaunchKernelGGL(bn_backward2_kernel, dim3(C / 64, n_groups), dim3(256), 0, (hipStream_t)stream, a, b);
    return mft_launch_status();
}

extern "C" int mft_ce_pool_bn_backward2(const float* feat, const int* labels, int imgs_per_group, int n_groups, int C, int hw,
                                        const float* out, const float* xa, const float* xb, float* dxa, float* dxb,
                                        const float* mean_a, const float* rstd_a, const float* gamma_a, const float* mean_b,
                                        const float* rstd_b, const float* gamma_b, long long gb_group_stride, float* dgamma_a,
                                        float* dbeta_a, float* dgamma_b, float* dbeta_b, float* loss, void* stream) {
    if (C % 64 != 0 || imgs_per_group < 1 || imgs_per_group > 16 || hw < 1) return MFT_EINVAL;
    const int rows = imgs_per_group * hw;
    BwdArgs a{xa, nullptr, nullptr, dxa, C, C, 0, C, C, rows, mean_a, rstd_a, gamma_a, gb_group_stride, dgamma_a, dbeta_a};
    BwdArgs b{xb, nullptr, nullptr, dxb, C, C, 0, C, C, rows, mean_b, rstd_b, gamma_b, gb_group_stride, dgamma_b, dbeta_b};
    CeArgs ce{feat, labels, out, loss, imgs_per_group, hw};
    hipLaunchKernelGGL(bn_backward2_ce_kernel, dim3(C / 64, n_groups), dim3(256), 0, (hipStream_t)stream, a, b, ce);
    return mft_launch_status();
}
