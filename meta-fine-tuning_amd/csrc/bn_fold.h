// BatchNorm pieces shared by the kernels that CONSUME a frozen-trunk convolution (csrc/bn.hip, csrc/conv_x3.hip):
// the statistics of a bf16x3 convolution leave its epilogue as per-tile partial sums; every consumer merges them itself in a
// fixed order (no separate finalize launch, no cross-workgroup hand-off), so all consumers see bit-identical mean / rstd.
// Reference: train-mode F.batch_norm behind every convolution of SimpleBlock (backbone.py:224-227, 251-261).
#pragma once
#include "mft_common.h"

typedef float mft_f32x2 __attribute__((ext_vector_type(2)));

// mean / rstd of one (group g, channel) from the [tiles][2][C][2] partials of mft_conv2d_nhwc_x3*_bnstats: tile t of BM rows
// overlaps group g (rows g*R .. g*R+R-1 of M) in n_t rows; (n_t, mean_t = s1/n_t, M2_t = s2 - s1^2/n_t) are merged in tile order
// with Chan's update
//     mean += (mean_t - mean) * n_t/(n + n_t),   M2 += M2_t + (mean_t - mean)^2 * n*n_t/(n + n_t),   n += n_t.
// The weights 1/n_t, n_t/(n+n_t), n*n_t/(n+n_t) depend on the tile geometry only (n = rows of the group before the tile): a
// workgroup computes them once per tile (mft_x3_tile_weights, three IEEE divisions) and the per-channel chain is five dependent
// FMAs per tile.  Every step is an explicit FMA / multiply, so the stand-alone finalize, the apply launch and the convolution
// loader that run this chain produce the same bits.
struct MftTileSpan { int t0, t1; };
struct MftTileW { float inv_nt, w1, w2, pad; };

__device__ __forceinline__ MftTileSpan mft_x3_group_tiles(int g, int M, int R, int BM) {
    const int r0 = g * R, r1 = min(r0 + R, M);
    return {r0 / BM, (r1 - 1) / BM};
}

__device__ __forceinline__ int mft_x3_tile_seg(int t, int g, int R, int BM) { return ((t * BM) / R == g) ? 0 : 1; }

__device__ __forceinline__ MftTileW mft_x3_tile_weights(int t, int g, int M, int R, int BM) {
    const int r0 = g * R, r1 = min(r0 + R, M);
    const int lo = max(t * BM, r0), hi = min(t * BM + BM, r1);
    const float nt = (float)(hi - lo), n = (float)(lo - r0), tot = (float)(hi - r0);
    return {1.0f / nt, nt / tot, n * nt / tot, 0.f};
}

// get(t, seg): the channel's (sum, sum of squares) of tile t, segment seg (0: the group the tile starts in, 1: the next one);
// weights(t): mft_x3_tile_weights of tile t -- either computed in place or read from an LDS copy.
template <class Get, class Weights>
__device__ __forceinline__ void mft_x3_stats_chain(Get get, Weights weights, int g, int M, int R, int BM, float eps, float& mean,
                                                   float& rstd) {
    const int r0 = g * R, r1 = min(r0 + R, M);
    const MftTileSpan sp = mft_x3_group_tiles(g, M, R, BM);
    float mu = 0.f, m2 = 0.f;
    for (int t = sp.t0; t <= sp.t1; ++t) {
        const MftTileW w = weights(t);
        const mft_f32x2 o = get(t, mft_x3_tile_seg(t, g, R, BM));
        const float mt = o[0] * w.inv_nt;
        const float m2t = fmaxf(__builtin_fmaf(-o[0], mt, o[1]), 0.f);
        const float d = mt - mu;
        m2 += __builtin_fmaf(d * d, w.w2, m2t);
        mu = __builtin_fmaf(d, w.w1, mu);
    }
    mean = mu;
    rstd = 1.0f / sqrtf(m2 / (float)(r1 - r0) + eps);
}

// one thread, straight from global memory (the stand-alone finalize launch; a chain of dependent loads: a memory round trip per tile)
__device__ __forceinline__ void mft_x3_stats_merge(const float* __restrict__ ws, int C, int c, int g, int M, int R, int BM,
                                                   float eps, float& mean, float& rstd) {
    mft_x3_stats_chain([&](int t, int seg) { return *(const mft_f32x2*)(ws + (((long long)t * 2 + seg) * C + c) * 2); },
                       [&](int t) { return mft_x3_tile_weights(t, g, M, R, BM); }, g, M, R, BM, eps, mean, rstd);
}

// workgroup-cooperative form: all ``nthreads`` threads first copy the partials of group g (every channel, every tile of the group)
// into ``stage`` (MftTileW [max_tiles] tile weights, then [tiles of the group][C] float2: independent coalesced loads), the caller
// synchronises, then one thread per channel runs the chain out of LDS with mft_x3_stats_staged.
__device__ __forceinline__ size_t mft_x3_stage_bytes(int max_tiles, int C) {
    return (size_t)max_tiles * (sizeof(MftTileW) + (size_t)C * sizeof(mft_f32x2));
}

__device__ __forceinline__ void mft_x3_stats_stage(const float* __restrict__ ws, int C, int g, int M, int R, int BM, int max_tiles,
                                                   void* stage, int tid, int nthreads) {
    MftTileW* wt = reinterpret_cast<MftTileW*>(stage);
    mft_f32x2* part = reinterpret_cast<mft_f32x2*>(wt + max_tiles);
    const MftTileSpan sp = mft_x3_group_tiles(g, M, R, BM);
    const int nt = sp.t1 - sp.t0 + 1;
    for (int i = tid; i < nt; i += nthreads) wt[i] = mft_x3_tile_weights(sp.t0 + i, g, M, R, BM);
    const int total = nt * C;
    for (int i = tid; i < total; i += nthreads) {
        const int tt = i / C, c = i - tt * C, t = sp.t0 + tt;
        part[i] = *(const mft_f32x2*)(ws + (((long long)t * 2 + mft_x3_tile_seg(t, g, R, BM)) * C + c) * 2);
    }
}

__device__ __forceinline__ void mft_x3_stats_staged(const void* stage, int max_tiles, int C, int c, int g, int M, int R, int BM,
                                                    float eps, float& mean, float& rstd) {
    const MftTileW* wt = reinterpret_cast<const MftTileW*>(stage);
    const mft_f32x2* part = reinterpret_cast<const mft_f32x2*>(wt + max_tiles);
    const int t0 = mft_x3_group_tiles(g, M, R, BM).t0;
    mft_x3_stats_chain([&](int t, int) { return part[(t - t0) * C + c]; }, [&](int t) { return wt[t - t0]; }, g, M, R, BM, eps, mean,
                       rstd);
}

// tiles a group of R rows can overlap (upper bound, for sizing the stage), and the stage's size on the host
inline int mft_x3_max_group_tiles(int R, int BM) { return (R + BM - 1) / BM + 1; }
inline size_t mft_x3_stage_bytes_host(int max_tiles, int C) { return (size_t)max_tiles * (16 + (size_t)C * 8); }

// y = (x - mean) * rstd * gamma + beta as ONE fused multiply-add per element: scale = rstd * gamma, shift = beta - mean * scale
// (the form PyTorch's CPU kernel uses, aten/native/cpu/batch_norm_kernel.cpp).  Every kernel that applies a trunk BatchNorm goes
// through these two helpers with explicit FMAs, so a BatchNorm folded into a convolution loader and the stand-alone apply launch
// round identically.
__device__ __forceinline__ void mft_bn_fold(float mean, float rstd, float gamma, float beta, float& scale, float& shift) {
    scale = rstd * gamma;
    shift = __builtin_fmaf(-mean, scale, beta);
}

__device__ __forceinline__ f32x4 mft_bn_affine4(const f32x4 v, const f32x4 scale, const f32x4 shift) {
    return __builtin_elementwise_fma(v, scale, shift);
}

__device__ __forceinline__ void mft_bn_fold4(const f32x4 mean, const f32x4 rstd, const f32x4 gamma, const f32x4 beta, f32x4& scale,
                                             f32x4& shift) {
    scale = rstd * gamma;
    shift = __builtin_elementwise_fma(-mean, scale, beta);
}
