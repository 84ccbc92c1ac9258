// Fused per-pair MLP of the GNN affinity (gnn.Wcompute.forward, gnn.py:78-132) for gfx950.
//
// The reference materialises |x_i - x_j| for all N x N node pairs of every graph ([B, F, N, N], up to 248 MB per episode)
// and pushes it through 4 x (1x1 conv + BatchNorm2d(batch statistics) + leaky_relu) + a 96 -> 1 conv + diagonal-masked row
// softmax.  BatchNorm's statistics are reductions over ALL B*N*N pair positions of an episode, so a layer cannot start
// before the previous one has finished everywhere -- the chain is five grid-wide phases whatever the kernel structure.
// This file runs each phase as ONE launch over the episode batch and removes everything around the GEMMs:
//
//   * the score is symmetric in (i, j) (|x_i - x_j| is), so only the N(N+1)/2 pairs with i <= j exist anywhere: rows of
//     every intermediate are "upper-triangle rows" p(i, j) = i*N - i(i-1)/2 + (j - i), half the FLOPs and half the bytes;
//     the BatchNorm statistics weigh an off-diagonal row twice (the reference sees (i,j) and (j,i)) and a diagonal row once;
//   * layer 1's A operand is generated in the loader: two rows of x (L2-resident, contiguous in j) -> |a - b| -> LDS; the
//     pair tensor never exists;
//   * layers 2-4 read the previous layer's RAW output and apply BatchNorm + leaky_relu in the loader (one fused multiply-add
//     with per-episode scale / shift, then a select), so no normalised activation is ever written;
//   * every layer's epilogue reduces its output tile (still in registers) to per-channel weighted (mean, M2) and a small
//     finalize launch merges the tiles of an episode with Chan's formula in tile order (deterministic, no float atomics)
//     straight into the next loader's (scale, shift);
//   * the 96 -> 1 layer is a wavefront reduction (8 lanes per pair row) that writes the compact symmetric score, and the
//     masked softmax reads it through p(i, j): A[b, i, :] without a dense N x N score.
//
// HBM traffic per pair row: raw h1..h4 written once and read once (2 x 2304 B) against 13 KB for the unfused sequence,
// on half the rows.  Arithmetic is fp32 MFMA (v_mfma_f32_32x32x2_f32: exact products, fp32 accumulate) as in csrc/conv_igemm.hip.
#include "mft_common.h"
#include <stdlib.h>

namespace {

constexpr int PM_BM = 128;     // pair rows per tile
constexpr int PM_BN = 96;      // output channels per tile (192 = 2 tiles, 96 = 1)
constexpr int PM_BK = 32;
constexpr int PM_LD = 36;      // LDS row stride in floats (conflict-free ds_read_b128 fragment reads)

struct PairArgs {
    const float* in;           // PAIR: node features x [n_graphs*N, ld_in]; BNACT: previous raw layer output [rows, ld_in]
    int ld_in;
    const int* ij;             // [P] (i << 16) | j of upper-triangle row p
    const float* scale_in;     // BNACT: [n_groups, K] rstd*gamma of the previous layer
    const float* shift_in;     //        [n_groups, K] beta - mean*rstd*gamma
    const float* w;            // [Cout, Kpad] packed weights (zero padded)
    const float* bias;         // [Cout]
    float* out;                // [n_groups*rows_per_group, Cout] raw (pre-BatchNorm) output
    int K, Kpad, Cout;
    int N, P, graphs_per_group, rows_per_group;
    int tiles_m, tiles_n;      // per group
    float slope;
    float* ws_mean; float* ws_m2; float* ws_n;     // per (group, m-tile): [.., Cout], [.., Cout], [..]
};

// MODE 0: PAIR loader (layer 1), MODE 1: BNACT loader (layers 2-4).  DB: double-buffered LDS (one barrier per K-step, 64.5 KB:
// two workgroups per CU) or single-buffered (two barriers per K-step, 32 KB: four workgroups per CU -- a tile is only 3-8
// K-steps long, so its load / compute / store phases overlap across workgroups rather than inside one).
template <int MODE, bool DB>
__global__ __launch_bounds__(256) void pair_mlp_layer_kernel(PairArgs p) {
    constexpr int BM = PM_BM, BN = PM_BN, BK = PM_BK, LD = PM_LD;
    constexpr int PA = BM / 32;        // 4 A passes of 32 rows
    constexpr int PB = BN / 32;        // 3 B passes
    constexpr int TN = BN / 32;        // 3 MFMA blocks per wave (wave = 32 rows x 96 channels)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __shared__ float s_wrow[BM];
    __shared__ float s_red[4][BN];
    __shared__ float s_red2[4][BN];
    __shared__ float s_wsum[4];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;

    // XCD-aware linear tile order: workgroup ids are dealt round-robin to the 8 XCDs; XCD x owns one contiguous range of
    // (group, m-tile, n-tile) triples, n fastest, so the n-tiles of an m-tile (same A rows) and neighbouring m-tiles (same x rows)
    // share one L2
    int lin = blockIdx.x;
    {
        const int nwg = gridDim.x, q = nwg >> 3, rmd = nwg & 7;
        const int xcd = lin & 7, slot = lin >> 3;
        lin = xcd * q + (xcd < rmd ? xcd : rmd) + slot;
    }
    const int nt = lin % p.tiles_n;
    const int mt = (lin / p.tiles_n) % p.tiles_m;
    const int g = lin / (p.tiles_n * p.tiles_m);
    const int m0 = mt * BM, n0 = nt * BN;

    const int lrow = tid >> 3, c4 = (tid & 7) * 4;

    // row descriptors
    const float* a_pi[PA];
    const float* a_pj[PA];
    bool a_ok[PA];
#pragma unroll
    for (int j = 0; j < PA; ++j) {
        const int m = m0 + lrow + 32 * j;
        a_ok[j] = m < p.rows_per_group;
        const int mm = a_ok[j] ? m : 0;
        if (MODE == 0) {
            const int b = mm / p.P, pp = mm - b * p.P;
            const int pk = p.ij[pp];
            const long long node0 = ((long long)g * p.graphs_per_group + b) * p.N;
            a_pi[j] = p.in + (node0 + (pk >> 16)) * p.ld_in + c4;
            a_pj[j] = p.in + (node0 + (pk & 0xffff)) * p.ld_in + c4;
        } else {
            a_pi[j] = p.in + ((long long)g * p.rows_per_group + mm) * p.ld_in + c4;
            a_pj[j] = nullptr;
        }
    }
    if (tid < BM) {
        const int m = m0 + tid;
        float wgt = 0.f;
        if (m < p.rows_per_group) {
            const int pk = p.ij[m % p.P];
            wgt = ((pk >> 16) == (pk & 0xffff)) ? 1.f : 2.f;
        }
        s_wrow[tid] = wgt;
    }
    const float* b_ptr[PB];
#pragma unroll
    for (int j = 0; j < PB; ++j) b_ptr[j] = p.w + (long long)(n0 + lrow + 32 * j) * p.Kpad + c4;     // Cout % 96 == 0
    const float* sc_ptr = MODE == 1 ? p.scale_in + (long long)g * p.K + c4 : nullptr;
    const float* sh_ptr = MODE == 1 ? p.shift_in + (long long)g * p.K + c4 : nullptr;

    f32x16 acc[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;

    f32x4 ra[PA], rb[PB];
    const int nk = p.Kpad / BK;

    auto load_tile = [&](int kt) {
        const int k0 = kt * BK;
        if (MODE == 0) {
#pragma unroll
            for (int j = 0; j < PA; ++j) {
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (a_ok[j]) {
                    const f32x4 xi = *(const f32x4*)(a_pi[j] + k0);
                    const f32x4 xj = *(const f32x4*)(a_pj[j] + k0);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = (k0 + c4 + e < p.K) ? fabsf(xi[e] - xj[e]) : 0.f;
                }
                ra[j] = v;
            }
        } else {
            const f32x4 sc = *(const f32x4*)(sc_ptr + k0);
            const f32x4 sh = *(const f32x4*)(sh_ptr + k0);
#pragma unroll
            for (int j = 0; j < PA; ++j) {
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (a_ok[j]) {
                    const f32x4 x = *(const f32x4*)(a_pi[j] + k0);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float y = x[e] * sc[e] + sh[e];
                        v[e] = y > 0.f ? y : y * p.slope;
                    }
                }
                ra[j] = v;
            }
        }
#pragma unroll
        for (int j = 0; j < PB; ++j) rb[j] = *(const f32x4*)(b_ptr[j] + k0);
    };
    auto store_tile = [&](int buf) {
        float* As = smem + (DB ? buf : 0) * (BM + BN) * LD;
        float* Bs = As + BM * LD;
#pragma unroll
        for (int j = 0; j < PA; ++j) *(f32x4*)(As + (lrow + 32 * j) * LD + c4) = ra[j];
#pragma unroll
        for (int j = 0; j < PB; ++j) *(f32x4*)(Bs + (lrow + 32 * j) * LD + c4) = rb[j];
    };

    load_tile(0);
    store_tile(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) load_tile(kt + 1);
        const float* As = smem + (DB ? buf : 0) * (BM + BN) * LD;
        const float* Bs = As + BM * LD;
        f32x4 av[4], bv[TN][4];
        {
            const float* ptr = As + (wave * 32 + r) * LD + h * 16;
#pragma unroll
            for (int q = 0; q < 4; ++q) av[q] = *(const f32x4*)(ptr + 4 * q);
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const float* ptr = Bs + (j * 32 + r) * LD + h * 16;
#pragma unroll
            for (int q = 0; q < 4; ++q) bv[j][q] = *(const f32x4*)(ptr + 4 * q);
        }
#pragma unroll
        for (int t = 0; t < 16; ++t)
#pragma unroll
            for (int j = 0; j < TN; ++j)
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t >> 2][t & 3], bv[j][t >> 2][t & 3], acc[j], 0, 0, 0);
        if (!DB) __syncthreads();
        if (kt + 1 < nk) store_tile(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: bias, raw store, weighted tile statistics.  C/D layout: col = lane&31, row = (e&3) + 8*(e>>2) + 4*h.
    // Statistics in one pass around the bias as pivot: S1 = sum w*(v - bias), S2 = sum w*(v - bias)^2 are sums of the bare
    // accumulators; tile mean = bias + S1/W, M2 = S2 - S1^2/W (cancellation only inside one 128-row tile; tiles are merged with
    // Chan's formula by the finalize launch).
    const long long out_row0 = (long long)g * p.rows_per_group;
    float wr[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) wr[e] = s_wrow[wave * 32 + (e & 3) + 8 * (e >> 2) + 4 * h];
    float wl = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) wl += wr[e];
    wl += __shfl_xor(wl, 32, 64);
    if (lane == 0) s_wsum[wave] = wl;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + j * 32 + r;
        const float bias = p.bias[n];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const float a = acc[j][e];
            const int row = (e & 3) + 8 * (e >> 2) + 4 * h;
            const int m = m0 + wave * 32 + row;
            if (m < p.rows_per_group) p.out[(out_row0 + m) * p.Cout + n] = a + bias;
            s1 += wr[e] * a;
            s2 += wr[e] * a * a;
        }
        s1 += __shfl_xor(s1, 32, 64);
        s2 += __shfl_xor(s2, 32, 64);
        if (h == 0) { s_red[wave][j * 32 + r] = s1; s_red2[wave][j * 32 + r] = s2; }
    }
    __syncthreads();
    const long long trow = (long long)g * p.tiles_m + mt;
    if (tid < BN) {
        const float wtile = s_wsum[0] + s_wsum[1] + s_wsum[2] + s_wsum[3];
        const float s1 = s_red[0][tid] + s_red[1][tid] + s_red[2][tid] + s_red[3][tid];
        const float s2 = s_red2[0][tid] + s_red2[1][tid] + s_red2[2][tid] + s_red2[3][tid];
        p.ws_mean[trow * p.Cout + n0 + tid] = p.bias[n0 + tid] + s1 / wtile;
        p.ws_m2[trow * p.Cout + n0 + tid] = fmaxf(s2 - s1 * (s1 / wtile), 0.f);
        if (tid == 0 && nt == 0) p.ws_n[trow] = wtile;
    }
}

// Chan merge of the m-tiles of one episode -> the next loader's affine: scale = gamma / sqrt(var + eps), shift = beta - mean *
// scale (biased variance over all graphs*N*N pair positions, gnn.py:65-74 BatchNorm2d in train mode).  One workgroup per
// (episode, 16 channels): 16 partitions each merge a contiguous range of tiles in tile order, then the 16 partials are merged
// in partition order -- a fixed tree, bit-identical from run to run.
__global__ __launch_bounds__(256) void pair_stats_finalize_kernel(const float* __restrict__ ws_mean, const float* __restrict__ ws_m2,
                                                                  const float* __restrict__ ws_n, int tiles_m, int C,
                                                                  const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                  float eps, float* __restrict__ scale, float* __restrict__ shift,
                                                                  float* __restrict__ mean_out, float* __restrict__ rstd_out) {
    __shared__ float sn[16][16], sm[16][16], sq[16][16];
    const int cb = C / 16;
    const int g = blockIdx.x / cb, c = (blockIdx.x % cb) * 16 + (threadIdx.x & 15), part = threadIdx.x >> 4;
    const int per = (tiles_m + 15) / 16;
    const int t0 = part * per, t1 = min(t0 + per, tiles_m);
    float n = 0.f, mean = 0.f, m2 = 0.f;
    for (int t = t0; t < t1; ++t) {
        const long long row = (long long)g * tiles_m + t;
        const float nb = ws_n[row], mb = ws_mean[row * C + c], qb = ws_m2[row * C + c];
        const float nn = n + nb, d = mb - mean;
        mean += d * (nb / nn);
        m2 += qb + d * d * (n * nb / nn);
        n = nn;
    }
    sn[part][threadIdx.x & 15] = n; sm[part][threadIdx.x & 15] = mean; sq[part][threadIdx.x & 15] = m2;
    __syncthreads();
    if (part == 0) {
        const int l = threadIdx.x & 15;
        n = 0.f; mean = 0.f; m2 = 0.f;
        for (int q = 0; q < 16; ++q) {
            const float nb = sn[q][l];
            if (nb > 0.f) {
                const float nn = n + nb, d = sm[q][l] - mean;
                mean += d * (nb / nn);
                m2 += sq[q][l] + d * d * (n * nb / nn);
                n = nn;
            }
        }
        const float rstd = 1.f / sqrtf(m2 / n + eps);
        const float sc = rstd * gamma[c];
        scale[(long long)g * C + c] = sc;
        shift[(long long)g * C + c] = beta[c] - mean * sc;
        if (mean_out) { mean_out[(long long)g * C + c] = mean; rstd_out[(long long)g * C + c] = rstd; }
    }
}

// s_ut[row] = b5 + sum_c w5[c] * lrelu(h4[row][c] * scale[c] + shift[c])  -- 8 lanes per row, float4 per lane
__global__ __launch_bounds__(256) void pair_score_kernel(const float* __restrict__ h, int C, const float* __restrict__ scale,
                                                         const float* __restrict__ shift, const float* __restrict__ w5,
                                                         const float* __restrict__ b5, float slope, float* __restrict__ s_ut,
                                                         long long rows_per_group, int n_groups) {
    const int sub = threadIdx.x & 7;
    const long long total = rows_per_group * n_groups;
    const float bias = b5[0];
    for (long long row = (long long)blockIdx.x * 32 + (threadIdx.x >> 3); row < total; row += (long long)gridDim.x * 32) {
        const int g = (int)(row / rows_per_group);
        const float* hr = h + row * C;
        const float* sc = scale + (long long)g * C;
        const float* sh = shift + (long long)g * C;
        float acc = 0.f;
        for (int c = sub * 4; c < C; c += 32) {
            const f32x4 x = *(const f32x4*)(hr + c);
            const f32x4 a = *(const f32x4*)(sc + c), b = *(const f32x4*)(sh + c), w = *(const f32x4*)(w5 + c);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float y = x[e] * a[e] + b[e];
                acc += w[e] * (y > 0.f ? y : y * slope);
            }
        }
        acc += __shfl_xor(acc, 1, 64);
        acc += __shfl_xor(acc, 2, 64);
        acc += __shfl_xor(acc, 4, 64);
        if (sub == 0) s_ut[row] = acc + bias;
    }
}

// A[b, i, j] = softmax_j(s[b, i, j] - 1e8 [i == j]) with s read from the compact symmetric store (gnn.py:105-115)
__global__ __launch_bounds__(256) void masked_softmax_ut_kernel(const float* __restrict__ s_ut, float* __restrict__ A,
                                                                int n_graphs, int N, int P) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= (long long)n_graphs * N) return;
    const int i = (int)(row % N);
    const float* sg = s_ut + (row / N) * P;
    auto at = [&](int j) -> float {
        const int a = i < j ? i : j, b = i < j ? j : i;
        return sg[a * N - (a * (a - 1)) / 2 + (b - a)] - (j == i ? 1e8f : 0.f);
    };
    float mx = -3.4e38f;
    for (int j = lane; j < N; j += 64) mx = fmaxf(mx, at(j));
    mx = wave_max(mx);
    float se = 0.f;
    for (int j = lane; j < N; j += 64) se += __expf(at(j) - mx);
    se = wave_sum(se);
    const float inv = 1.f / se;
    for (int j = lane; j < N; j += 64) A[row * N + j] = __expf(at(j) - mx) * inv;
}

}  // namespace

static int g_pair_db = 0;     // 1: double-buffered LDS form (MFT_PAIR_DB=1 at load time; A/B measurements in DESIGN.md)
__attribute__((constructor)) static void pair_env_init() {
    const char* e = getenv("MFT_PAIR_DB");
    if (e) g_pair_db = atoi(e);
}

extern "C" int mft_pair_mlp_tiles_m(int graphs_per_group, int N) {
    const long long P = (long long)N * (N + 1) / 2;
    return cdiv((long long)graphs_per_group * P, PM_BM);
}

extern "C" int mft_pair_mlp_layer(const float* in, int ld_in, int mode, const int* ij, const float* scale_in, const float* shift_in,
                                  const float* w, int K, int Kpad, const float* bias, float* out, int Cout, int n_groups,
                                  int graphs_per_group, int N, float slope, float* ws_mean, float* ws_m2, float* ws_n,
                                  void* stream) {
    if (Cout % PM_BN != 0 || Kpad % PM_BK != 0 || K > Kpad || ld_in % 4 != 0 || N < 1 || N > 65535 || n_groups < 1 ||
        (mode != 0 && mode != 1) || (mode == 1 && (K != Kpad || !scale_in || !shift_in)) || (mode == 0 && ld_in < Kpad))
        return MFT_EINVAL;
    PairArgs p;
    p.in = in; p.ld_in = ld_in; p.ij = ij; p.scale_in = scale_in; p.shift_in = shift_in; p.w = w; p.bias = bias; p.out = out;
    p.K = K; p.Kpad = Kpad; p.Cout = Cout; p.N = N; p.P = N * (N + 1) / 2; p.graphs_per_group = graphs_per_group;
    const long long rpg = (long long)graphs_per_group * p.P;
    if (rpg > 0x7fffffffLL) return MFT_EINVAL;
    p.rows_per_group = (int)rpg;
    p.tiles_m = cdiv(rpg, PM_BM);
    p.tiles_n = Cout / PM_BN;
    p.slope = slope; p.ws_mean = ws_mean; p.ws_m2 = ws_m2; p.ws_n = ws_n;
    const long long nwg = (long long)p.tiles_m * p.tiles_n * n_groups;
    if (nwg > 0x7fffffffLL) return MFT_EINVAL;
    const size_t lds1 = (PM_BM + PM_BN) * PM_LD * sizeof(float);
    static MftPerDeviceOnce attr_once;
    if (attr_once.need()) {
        hipError_t e = hipFuncSetAttribute((const void*)pair_mlp_layer_kernel<0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * lds1));
        if (e == hipSuccess)
            e = hipFuncSetAttribute((const void*)pair_mlp_layer_kernel<1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * lds1));
        if (e != hipSuccess) return (int)e;
    }
    hipStream_t st = (hipStream_t)stream;
    if (g_pair_db) {
        if (mode == 0) hipLaunchKernelGGL((pair_mlp_layer_kernel<0, true>), dim3((unsigned)nwg), dim3(256), 2 * lds1, st, p);
        else hipLaunchKernelGGL((pair_mlp_layer_kernel<1, true>), dim3((unsigned)nwg), dim3(256), 2 * lds1, st, p);
    } else {
        if (mode == 0) hipLaunchKernelGGL((pair_mlp_layer_kernel<0, false>), dim3((unsigned)nwg), dim3(256), lds1, st, p);
        else hipLaunchKernelGGL((pair_mlp_layer_kernel<1, false>), dim3((unsigned)nwg), dim3(256), lds1, st, p);
    }
    return mft_launch_status();
}

extern "C" int mft_pair_mlp_stats_finalize(const float* ws_mean, const float* ws_m2, const float* ws_n, int n_groups, int tiles_m,
                                           int C, const float* gamma, const float* beta, float eps, float* scale, float* shift,
                                           float* mean_out, float* rstd_out, void* stream) {
    if (n_groups < 1 || tiles_m < 1 || C < 16 || C % 16 != 0) return MFT_EINVAL;
    hipLaunchKernelGGL(pair_stats_finalize_kernel, dim3(n_groups * (C / 16)), dim3(256), 0, (hipStream_t)stream, ws_mean, ws_m2, ws_n,
                       tiles_m, C, gamma, beta, eps, scale, shift, mean_out, rstd_out);
    return mft_launch_status();
}

extern "C" int mft_pair_mlp_score(const float* h, int C, const float* scale, const float* shift, const float* w5, const float* b5,
                                  float slope, float* s_ut, int n_groups, int graphs_per_group, int N, void* stream) {
    if (C % 32 != 0 || n_groups < 1) return MFT_EINVAL;
    const long long rpg = (long long)graphs_per_group * N * (N + 1) / 2;
    long long blocks = (rpg * n_groups + 31) / 32;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(pair_score_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, h, C, scale, shift, w5, b5,
                       slope, s_ut, rpg, n_groups);
    return mft_launch_status();
}

extern "C" int mft_masked_softmax_ut(const float* s_ut, float* A, int n_graphs, int N, void* stream) {
    if (n_graphs < 1 || N < 1) return MFT_EINVAL;
    const long long rows = (long long)n_graphs * N;
    hipLaunchKernelGGL(masked_softmax_ut_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, s_ut, A,
                       n_graphs, N, N * (N + 1) / 2);
    return mft_launch_status();
}
