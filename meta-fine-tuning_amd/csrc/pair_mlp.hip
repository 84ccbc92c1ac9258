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

namespace {

typedef _Float16 pm_f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 pm_f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned pm_u32x2 __attribute__((ext_vector_type(2)));

// "f16x2" (csrc/conv_x3.hip): x = hi + 2^-11 lo, hi = fp16(x), lo = fp16((x - hi) * 2^11): |x - (hi + 2^-11 lo)| <= 2^-22 |x|;
// a b = a_hi b_hi + 2^-11 (a_hi b_lo + a_lo b_hi) + O(2^-22 a b): three fp16 MFMAs, leading / cross terms in two fp32 accumulators.
__device__ __forceinline__ void pm_split4_h2(const f32x4 x, pm_u32x2& p1, pm_u32x2& p2) {
    const pm_f16x4 hi = __builtin_convertvector(x, pm_f16x4);
    const f32x4 r = (x - __builtin_convertvector(hi, f32x4)) * 2048.f;
    const pm_f16x4 lo = __builtin_convertvector(r, pm_f16x4);
    p1 = __builtin_bit_cast(pm_u32x2, hi);
    p2 = __builtin_bit_cast(pm_u32x2, lo);
}

constexpr int PM_BM = 128;     // pair rows per tile
constexpr int PM_BN = 96;      // output channels per tile (192 = 2 tiles, 96 = 1)
constexpr int PM_BK = 32;
constexpr int PM_LD = 36;      // LDS row stride in floats (conflict-free ds_read_b128 fragment reads)
constexpr int PM_RS = 40;      // H2: LDS row stride in fp16 elements (32 + 8: 80-byte rows, 16-byte aligned fragments, as conv_x3's X3_RS)

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
// H2: both operands are split into two fp16 pieces by the loader (A: |x_i - x_j| or the BatchNorm'd activation, as computed above in
// fp32; B: the fp32 weight rows -- no pre-split planes, the weights change every meta-training step) and the tile runs on the fp16
// matrix cores as three products per K-step with fp32 accumulation: fp32-accurate (error vs float64 below the fp32-MFMA form's,
// tests/test_kernels_gpu.py::test_pair_mlp_f16x2_is_fp32_accurate), 18 MFMA issues of 32x32x16 per K-step instead of 48 of 32x32x2.
// Operands beyond fp16's range (|x| >= 65504) become infinities: the result is NaN, loudly.
template <int MODE, bool DB, bool H2 = false>
__global__ __launch_bounds__(256) void pair_mlp_layer_kernel(PairArgs p) {
    static_assert(!(H2 && DB), "f16x2: the single-buffer form only");
    constexpr int BM = PM_BM, BN = PM_BN, BK = PM_BK, LD = PM_LD;
    constexpr int RS = PM_RS, A_PLANE = BM * RS, B_PLANE = BN * RS;
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
    f32x16 acx[H2 ? TN : 1];           // H2: the cross products a_hi b_lo + a_lo b_hi (weight 2^-11)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
#pragma unroll
    for (int j = 0; j < (H2 ? TN : 1); ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acx[j][e] = 0.f;

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
        if constexpr (H2) {
            unsigned short* Ah = reinterpret_cast<unsigned short*>(smem);        // [2][BM][RS] then [2][BN][RS]
            unsigned short* Bh = Ah + 2 * A_PLANE;
#pragma unroll
            for (int j = 0; j < PA; ++j) {
                pm_u32x2 p1, p2;
                pm_split4_h2(ra[j], p1, p2);
                *(pm_u32x2*)(Ah + (lrow + 32 * j) * RS + c4) = p1;
                *(pm_u32x2*)(Ah + A_PLANE + (lrow + 32 * j) * RS + c4) = p2;
            }
#pragma unroll
            for (int j = 0; j < PB; ++j) {
                pm_u32x2 p1, p2;
                pm_split4_h2(rb[j], p1, p2);
                *(pm_u32x2*)(Bh + (lrow + 32 * j) * RS + c4) = p1;
                *(pm_u32x2*)(Bh + B_PLANE + (lrow + 32 * j) * RS + c4) = p2;
            }
            return;
        }
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
        if constexpr (H2) {
            const unsigned short* Ah = reinterpret_cast<const unsigned short*>(smem);
            const unsigned short* Bh = Ah + 2 * A_PLANE;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {                       // two 16-wide halves of the K-step
                pm_f16x8 a[2], b[TN][2];
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) a[pl] = *(const pm_f16x8*)(Ah + pl * A_PLANE + (wave * 32 + r) * RS + kk * 16 + h * 8);
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int pl = 0; pl < 2; ++pl) b[j][pl] = *(const pm_f16x8*)(Bh + pl * B_PLANE + (j * 32 + r) * RS + kk * 16 + h * 8);
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    acx[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], b[j][1], acx[j], 0, 0, 0);
                    acx[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1], b[j][0], acx[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], b[j][0], acc[j], 0, 0, 0);
                }
            }
            __syncthreads();
            if (kt + 1 < nk) store_tile(0);
            __syncthreads();
            continue;
        }
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
    if constexpr (H2) {
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[j][e] = __builtin_fmaf(acx[j][e], 1.0f / 2048.f, acc[j][e]);
    }
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

// "Register-K" form of the layer for SMALL problems (ONE 5-shot episode: 7,440 pair rows -- the meta-training step,
// meta_template.py:76-92).  At that size the 128 x 96 tile above gives 118 workgroups that each walk 3-8 K-steps of
// (global load -> LDS -> 48 MFMAs) back to back on a quarter-filled chip: 24.5 us per layer, nearly all of it exposed latency.
// Here a workgroup owns 32 pair rows x 96 channels over the WHOLE K (<= 256) and its four waves split K in 16-wide units (unit u
// -> wave u mod 4).  The MFMA sums over k, so WHICH k a lane feeds is free as long as A and B agree: with v_mfma_f32_16x16x4_f32
// lane (r, q) = (lane & 15, lane >> 4) feeds row r and k = 16 u + 4 q + e of element e of one float4 -- four neighbouring lanes
// read 64 contiguous bytes of a row (16 cache lines per load instruction; the 32x32x2 form would touch 64).  Every load of the tile
// is issued at once into registers (one exposed latency, no LDS staging, no barrier in the K loop).  The four partial tiles meet in
// LDS (added in wave order: fixed summation order) and waves 0-2 each finish 32 columns: bias, raw store, weighted (mean, M2) of the
// 32-row tile exactly as above, for the same finalize launch (tiles_m = rows / 32).  ~4x the workgroups, each ~1/16 of the work.
template <int MODE, int NV>
__global__ __launch_bounds__(256) void pair_mlp_layer_rk_kernel(PairArgs p) {
    constexpr int BM = 32, BN = PM_BN, RB = BM / 16, CB = BN / 16;
    __shared__ float s_acc[4][RB * CB][4][64];        // 48 KB
    __shared__ float s_wrow[BM];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, q = lane >> 4;

    int lin = blockIdx.x;                              // XCD-aware linear tile order, as above
    {
        const int nwg = gridDim.x, qq = nwg >> 3, rmd = nwg & 7;
        const int xcd = lin & 7, slot = lin >> 3;
        lin = xcd * qq + (xcd < rmd ? xcd : rmd) + slot;
    }
    const int nt = lin % p.tiles_n;
    const int mt = (lin / p.tiles_n) % p.tiles_m;
    const int g = lin / (p.tiles_n * p.tiles_m);
    const int m0 = mt * BM, n0 = nt * BN;

    if (tid < BM) {
        const int m = m0 + tid;
        float wgt = 0.f;
        if (m < p.rows_per_group) {
            const int pk = p.ij[m % p.P];
            wgt = ((pk >> 16) == (pk & 0xffff)) ? 1.f : 2.f;
        }
        s_wrow[tid] = wgt;
    }

    const int units = p.Kpad >> 4;                     // 16-wide K units; this wave takes wave, wave + 4, ...
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 ra[RB][NV], rb[CB][NV];
#pragma unroll
    for (int c = 0; c < CB; ++c) {
        const float* bp = p.w + (long long)(n0 + c * 16 + r) * p.Kpad + 4 * q;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int u = wave + 4 * v;
            rb[c][v] = u < units ? *(const f32x4*)(bp + 16 * u) : zero4;
        }
    }
#pragma unroll
    for (int b = 0; b < RB; ++b) {
        const int m = m0 + b * 16 + r;
        const bool ok = m < p.rows_per_group;
        const int mm = ok ? m : 0;
        if (MODE == 0) {
            const int gb = mm / p.P, pp = mm - gb * p.P;
            const int pk = p.ij[pp];
            const long long node0 = ((long long)g * p.graphs_per_group + gb) * p.N;
            const float* xi = p.in + (node0 + (pk >> 16)) * p.ld_in + 4 * q;
            const float* xj = p.in + (node0 + (pk & 0xffff)) * p.ld_in + 4 * q;
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const int u = wave + 4 * v;
                f32x4 o = zero4;
                if (ok && u < units) {
                    const f32x4 a = *(const f32x4*)(xi + 16 * u), c = *(const f32x4*)(xj + 16 * u);
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = (16 * u + 4 * q + e < p.K) ? fabsf(a[e] - c[e]) : 0.f;
                }
                ra[b][v] = o;
            }
        } else {
            const float* xp = p.in + ((long long)g * p.rows_per_group + mm) * p.ld_in + 4 * q;
            const float* sc = p.scale_in + (long long)g * p.K + 4 * q;
            const float* sh = p.shift_in + (long long)g * p.K + 4 * q;
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const int u = wave + 4 * v;
                f32x4 o = zero4;
                if (ok && u < units) {
                    const f32x4 x = *(const f32x4*)(xp + 16 * u);
                    const f32x4 a = *(const f32x4*)(sc + 16 * u), c = *(const f32x4*)(sh + 16 * u);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float y = x[e] * a[e] + c[e];
                        o[e] = y > 0.f ? y : y * p.slope;
                    }
                }
                ra[b][v] = o;
            }
        }
    }

    f32x4 acc[RB][CB];
#pragma unroll
    for (int b = 0; b < RB; ++b)
#pragma unroll
        for (int c = 0; c < CB; ++c) acc[b][c] = zero4;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        if (wave + 4 * v < units) {                    // wave-uniform
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int b = 0; b < RB; ++b)
#pragma unroll
                    for (int c = 0; c < CB; ++c)
                        acc[b][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(ra[b][v][e], rb[c][v][e], acc[b][c], 0, 0, 0);
        }
    }

#pragma unroll
    for (int b = 0; b < RB; ++b)
#pragma unroll
        for (int c = 0; c < CB; ++c)
#pragma unroll
            for (int e = 0; e < 4; ++e) s_acc[wave][b * CB + c][e][lane] = acc[b][c][e];
    __syncthreads();
    if (wave >= BN / 32) return;

    // ---- epilogue of columns [32 wave, 32 wave + 32): C/D layout of a 16 x 16 block: col = lane & 15, row = 4 * (lane >> 4) + e
    float wr[RB][4];
    float wl = 0.f;
#pragma unroll
    for (int b = 0; b < RB; ++b)
#pragma unroll
        for (int e = 0; e < 4; ++e) { wr[b][e] = s_wrow[b * 16 + 4 * q + e]; wl += wr[b][e]; }
    wl += __shfl_xor(wl, 16, 64);
    wl += __shfl_xor(wl, 32, 64);                      // weight of the whole 32-row tile
    const long long out_row0 = (long long)g * p.rows_per_group;
    const long long trow = (long long)g * p.tiles_m + mt;
#pragma unroll
    for (int cc = 0; cc < 2; ++cc) {
        const int c = wave * 2 + cc;
        const int n = n0 + c * 16 + r;
        const float bias = p.bias[n];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int b = 0; b < RB; ++b)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int blk = b * CB + c;
                const float a = ((s_acc[0][blk][e][lane] + s_acc[1][blk][e][lane]) + s_acc[2][blk][e][lane]) + s_acc[3][blk][e][lane];
                const int mrow = m0 + b * 16 + 4 * q + e;
                if (mrow < p.rows_per_group) p.out[(out_row0 + mrow) * p.Cout + n] = a + bias;
                s1 += wr[b][e] * a;
                s2 += wr[b][e] * a * a;
            }
        s1 += __shfl_xor(s1, 16, 64); s2 += __shfl_xor(s2, 16, 64);
        s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64);
        if (q == 0) {
            p.ws_mean[trow * p.Cout + n] = bias + s1 / wl;
            p.ws_m2[trow * p.Cout + n] = fmaxf(s2 - s1 * (s1 / wl), 0.f);
        }
    }
    if (wave == 0 && nt == 0 && lane == 0) p.ws_n[trow] = wl;
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

// The same merge for the register-K layers' 32-row tiles (233 per 5-shot episode; the 16-partition walk above takes 10 us there:
// 15 dependent Chan merges, two divisions each, per lane).  One WAVE per channel: lane l merges tiles l, l + 64, ... in order, then
// six butterfly levels (lane l absorbs lane l ^ s, s = 1, 2, .. 32; lane 0 ends with the total) -- a fixed tree, 10 merges deep.
__global__ __launch_bounds__(256) void pair_stats_finalize_tree_kernel(const float* __restrict__ ws_mean, const float* __restrict__ ws_m2,
                                                                       const float* __restrict__ ws_n, int tiles_m, int C,
                                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                       float eps, float* __restrict__ scale, float* __restrict__ shift,
                                                                       float* __restrict__ mean_out, float* __restrict__ rstd_out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int cb = C / 4;
    const int g = blockIdx.x / cb, c = (blockIdx.x % cb) * 4 + wave;
    float n = 0.f, mean = 0.f, m2 = 0.f;
    for (int t = lane; t < tiles_m; t += 64) {
        const long long row = (long long)g * tiles_m + t;
        const float nb = ws_n[row], mb = ws_mean[row * C + c], qb = ws_m2[row * C + c];
        const float nn = n + nb, d = mb - mean;
        mean += d * (nb / nn);
        m2 += qb + d * d * (n * nb / nn);
        n = nn;
    }
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) {
        const float nb = __shfl_xor(n, s, 64), mb = __shfl_xor(mean, s, 64), qb = __shfl_xor(m2, s, 64);
        if (nb > 0.f) {
            const float nn = n + nb, d = mb - mean;
            mean += d * (nb / nn);
            m2 += qb + d * d * (n * nb / nn);
            n = nn;
        }
    }
    if (lane == 0) {
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

// ------------------------------------------------------------------------------------------------ backward (meta-training)
// loss.backward() through gnn.Wcompute (gnn.py:78-132 under autograd; meta_template.py:76-92) on the SAME upper-triangle rows the
// forward keeps.  In the reference's N x N formulation the positions (i, j) and (j, i) carry identical activations and their own
// gradients; every backward operator of the chain is linear in the gradient, so a merged row p(i, j) that carries the SUM of the two
// gradients gives the same parameter gradients and the same d x -- with one change: the mean-subtraction terms of the BatchNorm
// backward count a merged off-diagonal row twice (cnt = 2; diagonal rows once), and the means are over all n_graphs * N * N
// positions.  Nothing of shape [B * N * N, F] exists: layer outputs are [B * N(N+1)/2, <= 192], |x_i - x_j| is produced for a
// bounded chunk of rows at a time.

// rd[b, i] = sum_k A[b,i,k] * dA[b,i,k]     (one wave per row)
__global__ __launch_bounds__(256) void softmax_rowdot_kernel(const float* __restrict__ A, const float* __restrict__ dA,
                                                             float* __restrict__ rd, long long rows, int N) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float s = 0.f;
    for (int j = lane; j < N; j += 64) s += A[row * N + j] * dA[row * N + j];
    s = wave_sum(s);
    if (lane == 0) rd[row] = s;
}

// ds_ut[b, p(i,j)] = A_ij (dA_ij - rd_i) + A_ji (dA_ji - rd_j)  (i < j);  diagonal: A_ii (dA_ii - rd_i) (= 0: the masked logit)
// INLINE_RD (small graphs, N <= 64): every thread forms the two row dots <A_i, dA_i>, <A_j, dA_j> itself (2 N multiply-adds out of
// L1 / L2) -- no softmax_rowdot launch in front.  dbias_zero (nullable, one float): the gradient of conv2d_last's bias, which is
// identically zero (the bias shifts every logit of a softmax row alike): written as 0 here instead of a column sum of rounding noise.
template <bool INLINE_RD>
__global__ __launch_bounds__(256) void softmax_ut_backward_kernel(const float* __restrict__ A, const float* __restrict__ dA,
                                                                  const float* __restrict__ rd, const int* __restrict__ ij,
                                                                  float* __restrict__ ds, int ldds, int n_graphs, int N, int P,
                                                                  float* __restrict__ dbias_zero) {
    const long long total = (long long)n_graphs * P;
    if (dbias_zero && blockIdx.x == 0 && threadIdx.x == 0) dbias_zero[0] = 0.f;
    for (long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x; r < total; r += (long long)gridDim.x * blockDim.x) {
        const int b = (int)(r / P), pk = ij[r % P];
        const int i = pk >> 16, j = pk & 0xffff;
        const long long base = (long long)b * N * N, rb = (long long)b * N;
        float rdi, rdj;
        if (INLINE_RD) {
            rdi = 0.f; rdj = 0.f;
            const float* Ai = A + base + (long long)i * N; const float* dAi = dA + base + (long long)i * N;
            const float* Aj = A + base + (long long)j * N; const float* dAj = dA + base + (long long)j * N;
#pragma unroll 10
            for (int k = 0; k < N; ++k) { rdi += Ai[k] * dAi[k]; rdj += Aj[k] * dAj[k]; }
        } else {
            rdi = rd[rb + i]; rdj = rd[rb + j];
        }
        float v = A[base + (long long)i * N + j] * (dA[base + (long long)i * N + j] - rdi);
        if (i != j) v += A[base + (long long)j * N + i] * (dA[base + (long long)j * N + i] - rdj);
        ds[r * ldds] = v;
    }
}

// BatchNorm + leaky_relu backward over merged rows, phase 1: per-channel sums of u = g * lrelu'(y) and u * xhat, y = z * scale +
// shift, xhat = (z - mean) * rstd.  One block per contiguous range of rows of ONE group (grid.y = group = episode: k episodes in
// lockstep have their own BatchNorm statistics, gnn.py:65-102 runs once per episode); fixed-order partials (block, then row lane).
__global__ __launch_bounds__(256) void pair_bwd_stats_kernel(const float* __restrict__ g, int ldg, const float* __restrict__ z, int C,
                                                             const float* __restrict__ scale, const float* __restrict__ shift,
                                                             const float* __restrict__ mean, const float* __restrict__ rstd,
                                                             float slope, long long rows, long long rows_per_block,
                                                             float* __restrict__ ws) {
    __shared__ float red[2][10][192];
    const int q = C / 4, rp = 256 / q;                // float4 per row; rows per pass (5 at C = 192, 10 at C = 96)
    const int c4 = (threadIdx.x % q) * 4, rl = threadIdx.x / q;
    const int grp = blockIdx.y;
    const long long gro = (long long)grp * rows;      // first row of the group
    f32x4 su = {0.f, 0.f, 0.f, 0.f}, sx = {0.f, 0.f, 0.f, 0.f};
    if (rl < rp) {
        const long long go = (long long)grp * C;
        const f32x4 sc = *(const f32x4*)(scale + go + c4), sh = *(const f32x4*)(shift + go + c4);
        const f32x4 mu = *(const f32x4*)(mean + go + c4), rs = *(const f32x4*)(rstd + go + c4);
        const long long r0 = (long long)blockIdx.x * rows_per_block;
        const long long r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
#pragma unroll 4
        for (long long r = r0 + rl; r < r1; r += rp) {
            const f32x4 gv = *(const f32x4*)(g + (gro + r) * ldg + c4), zv = *(const f32x4*)(z + (gro + r) * C + c4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float y = zv[e] * sc[e] + sh[e];
                const float u = y > 0.f ? gv[e] : gv[e] * slope;
                su[e] += u;
                sx[e] += u * ((zv[e] - mu[e]) * rs[e]);
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) { red[0][rl][c4 + e] = su[e]; red[1][rl][c4 + e] = sx[e]; }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * C; i += 256) {
        const int k = i / C, c = i - k * C;
        float s = 0.f;
        for (int l = 0; l < rp; ++l) s += red[k][l][c];
        ws[(((long long)grp * gridDim.x + blockIdx.x) * 2 + k) * C + c] = s;
    }
}

// 16 (sum kind, channel) entries per block, 16 lanes per entry over the block partials (lane l takes partials l, l+16, ...),
// combined in lane order: fixed summation order, and chains of ~8 dependent loads instead of one of ~120 (the serial form took
// 28 us per call -- 12 calls per meta-training step -- for 178 KB of partials).  One block row per group: sums[g][2C] (what phase 2
// subtracts).  Their sum over the groups in group order, dparams[2C] = (d beta | d gamma) of the shared affine parameters, and the
// identically-zero bias gradient dbias_zero [C] are written by workgroup 0 of the phase-2 launch.
__global__ __launch_bounds__(256) void pair_bwd_stats_final_kernel(const float* __restrict__ ws, int nblk, int C, int n_groups,
                                                                   float* __restrict__ sums) {
    __shared__ float red[16][17];
    const int e = threadIdx.x & 15, l = threadIdx.x >> 4;
    const int i = blockIdx.x * 16 + e;                 // over 2 * C
    const int grp = blockIdx.y;                        // one block row per group: k episodes in lockstep are reduced side by side
    float s = 0.f;
    if (i < 2 * C) {
        const int k = i / C, c = i - k * C;
#pragma unroll 8
        for (int b = l; b < nblk; b += 16) s += ws[(((long long)grp * nblk + b) * 2 + k) * C + c];
    }
    red[l][e] = s;
    __syncthreads();
    if (l == 0 && i < 2 * C) {
        float t = red[0][e];
#pragma unroll
        for (int j = 1; j < 16; ++j) t += red[j][e];
        sums[(long long)grp * 2 * C + i] = t;
    }
}

// phase 2: dz = gamma * rstd * (u - cnt * sum_u / n_tot - cnt * xhat * sum_ux / n_tot), cnt = 1 on diagonal rows, 2 elsewhere
// (sums, scale, shift, mean, rstd of the row's group; n_tot = positions of ONE group)
__global__ __launch_bounds__(256) void pair_bwd_dz_kernel(const float* __restrict__ g, int ldg, const float* __restrict__ z, int C,
                                                          const float* __restrict__ scale, const float* __restrict__ shift,
                                                          const float* __restrict__ mean, const float* __restrict__ rstd,
                                                          const float* __restrict__ gamma, const float* __restrict__ sums,
                                                          const int* __restrict__ ij, int P, float inv_n_tot, float slope,
                                                          long long rows, long long rows_per_group, float* __restrict__ dz,
                                                          int n_groups, float* __restrict__ dparams, float* __restrict__ dbias_zero) {
    if (blockIdx.x == 0) {                             // (d beta | d gamma) = the groups' sums added in group order; the zero bias gradient
        for (int i = threadIdx.x; i < 2 * C; i += blockDim.x) {
            float tot = 0.f;
            for (int g = 0; g < n_groups; ++g) tot += sums[(long long)g * 2 * C + i];
            if (dparams) dparams[i] = tot;
            if (dbias_zero && i < C) dbias_zero[i] = 0.f;
        }
    }
    const int q = C / 4;
    const long long total = rows * q;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
        const long long r = t / q;
        const int c4 = (int)(t - r * q) * 4;
        const long long go = (r / rows_per_group) * C;
        const int pk = ij[r % P];
        const float cnt = ((pk >> 16) == (pk & 0xffff)) ? 1.f : 2.f;
        const f32x4 gv = *(const f32x4*)(g + r * ldg + c4), zv = *(const f32x4*)(z + r * C + c4);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int c = c4 + e;
            const float y = zv[e] * scale[go + c] + shift[go + c];
            const float u = y > 0.f ? gv[e] : gv[e] * slope;
            const float xh = (zv[e] - mean[go + c]) * rstd[go + c];
            o[e] = gamma[c] * rstd[go + c] * (u - cnt * (sums[2 * go + c] * inv_n_tot) - cnt * xh * (sums[2 * go + C + c] * inv_n_tot));
        }
        *(f32x4*)(dz + r * C + c4) = o;
    }
}

// d[r][f] = |x_i[f] - x_j[f]| for the upper-triangle rows row0 .. row0 + nrows (zero beyond F up to Kp)
__global__ __launch_bounds__(256) void pair_absdiff_ut_kernel(const float* __restrict__ x, int ldx, const int* __restrict__ ij,
                                                              float* __restrict__ d, int Kp, int F, int N, int P, long long row0,
                                                              long long nrows) {
    const int q = Kp / 4;
    const long long total = nrows * q;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
        const long long rl = t / q, r = row0 + rl;
        const int c4 = (int)(t - rl * q) * 4;
        const int b = (int)(r / P), pk = ij[r % P];
        const float* xi = x + ((long long)b * N + (pk >> 16)) * ldx + c4;
        const float* xj = x + ((long long)b * N + (pk & 0xffff)) * ldx + c4;
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (c4 + e < F) ? fabsf(xi[e] - xj[e]) : 0.f;
        *(f32x4*)(d + rl * Kp + c4) = o;
    }
}

// dX[b, i, f] += sum_j sign(x_i[f] - x_j[f]) * dd[p(i, j)][f] over the pairs whose row lies in [row0, row0 + nrows)
// (one thread per (b, i, f), j ascending: deterministic; d|a - b| / da = sign(a - b), 0 at a == b as torch.abs)
__global__ __launch_bounds__(256) void pair_dx_gather_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ dd, int lddd,
                                                             float* __restrict__ dX, int lddx, int N, int P, int F, long long row0,
                                                             long long nrows) {
    const long long node = blockIdx.x;                 // b * N + i
    const int b = (int)(node / N), i = (int)(node - (long long)b * N);
    for (int f = threadIdx.x; f < F; f += blockDim.x) {
        const float xi = x[node * ldx + f];
        float acc = 0.f;
        for (int j0 = 0; j0 < N; j0 += 6) {            // six j at a time: twelve unconditional loads in flight, then the adds in j order
            float xv[6], dv[6];
            bool okv[6];
#pragma unroll
            for (int u = 0; u < 6; ++u) {
                const int j = j0 + u < N ? j0 + u : N - 1;
                const int a = i < j ? i : j, c = i < j ? j : i;
                const long long r = (long long)b * P + (a * N - (a * (a - 1)) / 2 + (c - a)) - row0;
                okv[u] = j0 + u < N && j != i && r >= 0 && r < nrows;
                xv[u] = x[((long long)b * N + j) * ldx + f];
                dv[u] = dd[(okv[u] ? r : 0) * lddd + f];
            }
#pragma unroll
            for (int u = 0; u < 6; ++u) {
                const float df = xi - xv[u];
                const float sg = df > 0.f ? 1.f : (df < 0.f ? -1.f : 0.f);
                acc += okv[u] ? sg * dv[u] : 0.f;
            }
        }
        dX[node * lddx + f] += acc;
    }
}

}  // namespace

extern "C" int mft_pair_mlp_tiles_m(int graphs_per_group, int N) {
    const long long P = (long long)N * (N + 1) / 2;
    return cdiv((long long)graphs_per_group * P, PM_BM);
}

extern "C" int mft_pair_mlp_layer(const float* in, int ld_in, int mode, const int* ij, const float* scale_in, const float* shift_in,
                                  const float* w, int K, int Kpad, const float* bias, float* out, int Cout, int n_groups,
                                  int graphs_per_group, int N, float slope, float* ws_mean, float* ws_m2, float* ws_n,
                                  int f16x2, void* stream) {
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
    // single-buffered form (32 KB of LDS, four workgroups per CU).  The double-buffered instantiation (DB = true: one barrier per
    // K-step, 64.5 KB) measured the same on the batched final pass (rounds 2-3) and on the single-episode meta-training step
    // (round 5: 3.73 / 3.77 vs 3.74 / 3.78 ms), so it is no longer launched.
    const size_t lds1 = (PM_BM + PM_BN) * PM_LD * sizeof(float);
    hipStream_t st = (hipStream_t)stream;
    if (f16x2) {                   // split-precision form: 35.8 KB of LDS (two fp16 planes per operand, 80-byte rows)
        const size_t ldsh = 2 * (PM_BM + PM_BN) * PM_RS * sizeof(unsigned short);
        if (mode == 0) hipLaunchKernelGGL((pair_mlp_layer_kernel<0, false, true>), dim3((unsigned)nwg), dim3(256), ldsh, st, p);
        else hipLaunchKernelGGL((pair_mlp_layer_kernel<1, false, true>), dim3((unsigned)nwg), dim3(256), ldsh, st, p);
        return mft_launch_status();
    }
    if (mode == 0) hipLaunchKernelGGL((pair_mlp_layer_kernel<0, false>), dim3((unsigned)nwg), dim3(256), lds1, st, p);
    else hipLaunchKernelGGL((pair_mlp_layer_kernel<1, false>), dim3((unsigned)nwg), dim3(256), lds1, st, p);
    return mft_launch_status();
}

/* ---- register-K form for small problems (see pair_mlp_layer_rk_kernel): 32-row tiles ---- */
extern "C" int mft_pair_mlp_tiles_m_rk(int graphs_per_group, int N) {
    const long long P = (long long)N * (N + 1) / 2;
    return cdiv((long long)graphs_per_group * P, 32);
}

template <int MODE>
static int pair_rk_launch(int nv, unsigned nwg, hipStream_t st, const PairArgs& p) {
    switch (nv) {
#define MFT_RK_CASE(V) \
    case V: hipLaunchKernelGGL((pair_mlp_layer_rk_kernel<MODE, V>), dim3(nwg), dim3(256), 0, st, p); break;
        MFT_RK_CASE(1) MFT_RK_CASE(2) MFT_RK_CASE(3) MFT_RK_CASE(4)
#undef MFT_RK_CASE
    default: return MFT_EINVAL;
    }
    return mft_launch_status();
}

extern "C" int mft_pair_mlp_layer_rk(const float* in, int ld_in, int mode, const int* ij, const float* scale_in, const float* shift_in,
                                     const float* w, int K, int Kpad, const float* bias, float* out, int Cout, int n_groups,
                                     int graphs_per_group, int N, float slope, float* ws_mean, float* ws_m2, float* ws_n,
                                     void* stream) {
    if (Cout % PM_BN != 0 || Kpad % 32 != 0 || Kpad < 32 || Kpad > 256 || K > Kpad || ld_in % 4 != 0 || N < 1 || N > 65535 ||
        n_groups < 1 || (mode != 0 && mode != 1) || (mode == 1 && (K != Kpad || !scale_in || !shift_in)) || (mode == 0 && ld_in < Kpad) ||
        ((unsigned long long)in & 15) != 0 || ((unsigned long long)w & 15) != 0)        // 16-byte operand loads
        return MFT_EINVAL;
    PairArgs p;
    p.in = in; p.ld_in = ld_in; p.ij = ij; p.scale_in = scale_in; p.shift_in = shift_in; p.w = w; p.bias = bias; p.out = out;
    p.K = K; p.Kpad = Kpad; p.Cout = Cout; p.N = N; p.P = N * (N + 1) / 2; p.graphs_per_group = graphs_per_group;
    const long long rpg = (long long)graphs_per_group * p.P;
    if (rpg > 0x7fffffffLL) return MFT_EINVAL;
    p.rows_per_group = (int)rpg;
    p.tiles_m = cdiv(rpg, 32);
    p.tiles_n = Cout / PM_BN;
    p.slope = slope; p.ws_mean = ws_mean; p.ws_m2 = ws_m2; p.ws_n = ws_n;
    const long long nwg = (long long)p.tiles_m * p.tiles_n * n_groups;
    if (nwg > 0x7fffffffLL) return MFT_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const int nv = (Kpad / 16 + 3) / 4;            // 16-wide K units per wave
    return mode == 0 ? pair_rk_launch<0>(nv, (unsigned)nwg, st, p) : pair_rk_launch<1>(nv, (unsigned)nwg, st, p);
}

extern "C" int mft_pair_mlp_stats_finalize(const float* ws_mean, const float* ws_m2, const float* ws_n, int n_groups, int tiles_m,
                                           int C, const float* gamma, const float* beta, float eps, float* scale, float* shift,
                                           float* mean_out, float* rstd_out, void* stream) {
    if (n_groups < 1 || tiles_m < 1 || C < 16 || C % 16 != 0) return MFT_EINVAL;
    hipLaunchKernelGGL(pair_stats_finalize_kernel, dim3(n_groups * (C / 16)), dim3(256), 0, (hipStream_t)stream, ws_mean, ws_m2, ws_n,
                       tiles_m, C, gamma, beta, eps, scale, shift, mean_out, rstd_out);
    return mft_launch_status();
}

extern "C" int mft_pair_mlp_stats_finalize_rk(const float* ws_mean, const float* ws_m2, const float* ws_n, int n_groups, int tiles_m,
                                              int C, const float* gamma, const float* beta, float eps, float* scale, float* shift,
                                              float* mean_out, float* rstd_out, void* stream) {
    if (n_groups < 1 || tiles_m < 1 || C < 16 || C % 16 != 0) return MFT_EINVAL;
    hipLaunchKernelGGL(pair_stats_finalize_tree_kernel, dim3(n_groups * (C / 4)), dim3(256), 0, (hipStream_t)stream, ws_mean, ws_m2, ws_n,
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

/* ---- backward over upper-triangle rows (see the kernels' comments) ---- */
extern "C" int mft_pair_softmax_ut_backward(const float* A, const float* dA, const int* ij, float* rowdot_ws, float* ds, int ldds,
                                            int n_graphs, int N, float* dbias_zero, void* stream) {
    if (n_graphs < 1 || N < 1 || ldds < 1) return MFT_EINVAL;
    const long long rows = (long long)n_graphs * N;
    const int P = N * (N + 1) / 2;
    hipStream_t st = (hipStream_t)stream;
    long long blocks = ((long long)n_graphs * P + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    if (N <= 64) {                 // 5-shot graphs (N = 30): one launch, the row dots formed in place
        hipLaunchKernelGGL(softmax_ut_backward_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, st, A, dA, (const float*)nullptr, ij, ds,
                           ldds, n_graphs, N, P, dbias_zero);
        return mft_launch_status();
    }
    hipLaunchKernelGGL(softmax_rowdot_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, A, dA, rowdot_ws, rows, N);
    hipLaunchKernelGGL(softmax_ut_backward_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, st, A, dA, (const float*)rowdot_ws, ij, ds,
                       ldds, n_graphs, N, P, dbias_zero);
    return mft_launch_status();
}

extern "C" long long mft_pair_bwd_stats_ws_floats(long long rows_per_group, int C) {
    long long nblk = (rows_per_group + 63) / 64;
    if (nblk > 1024) nblk = 1024;
    return nblk * 2 * C;                           // per group
}

extern "C" int mft_pair_bn_act_backward(const float* g, int ldg, const float* z, int C, const float* scale, const float* shift,
                                        const float* mean, const float* rstd, const float* gamma, const int* ij, int N,
                                        long long rows_per_group, int n_groups, long long n_tot, float slope, float* ws, float* sums,
                                        float* dparams, float* dbias_zero, float* dz, void* stream) {
    if ((C != 96 && C != 192) || ldg < C || ldg % 4 != 0 || rows_per_group < 1 || n_groups < 1 || n_groups > 65535 || n_tot < rows_per_group)
        return MFT_EINVAL;
    const long long P = (long long)N * (N + 1) / 2;
    if (rows_per_group % P != 0) return MFT_EINVAL;         // a group is whole graphs
    long long nblk = (rows_per_group + 63) / 64;   // ~64 rows per block: enough blocks to fill the CUs at the 5-shot graph size too
    if (nblk > 1024) nblk = 1024;
    const long long rpb = (rows_per_group + nblk - 1) / nblk;
    const long long rows = rows_per_group * n_groups;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(pair_bwd_stats_kernel, dim3((unsigned)nblk, (unsigned)n_groups), dim3(256), 0, st, g, ldg, z, C, scale, shift, mean,
                       rstd, slope, rows_per_group, rpb, ws);
    hipLaunchKernelGGL(pair_bwd_stats_final_kernel, dim3((2 * C + 15) / 16, (unsigned)n_groups), dim3(256), 0, st, (const float*)ws, (int)nblk, C,
                       n_groups, sums);
    long long blocks = (rows * (C / 4) + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(pair_bwd_dz_kernel, dim3((unsigned)blocks), dim3(256), 0, st, g, ldg, z, C, scale, shift, mean, rstd, gamma,
                       (const float*)sums, ij, (int)P, 1.0f / (float)n_tot, slope, rows, rows_per_group, dz, n_groups, dparams, dbias_zero);
    return mft_launch_status();
}

extern "C" int mft_pair_absdiff_ut(const float* x, int ldx, const int* ij, float* d, int Kp, int F, int N, long long row0,
                                   long long nrows, void* stream) {
    if (Kp % 4 != 0 || F > Kp || ldx < Kp || nrows < 1) return MFT_EINVAL;
    long long blocks = (nrows * (Kp / 4) + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(pair_absdiff_ut_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, ldx, ij, d, Kp, F, N,
                       N * (N + 1) / 2, row0, nrows);
    return mft_launch_status();
}

extern "C" int mft_pair_dx_gather(const float* x, int ldx, const float* dd, int lddd, float* dX, int lddx, int n_graphs, int N, int F,
                                  long long row0, long long nrows, void* stream) {
    if (n_graphs < 1 || N < 1 || F < 1 || nrows < 1) return MFT_EINVAL;
    hipLaunchKernelGGL(pair_dx_gather_kernel, dim3((unsigned)((long long)n_graphs * N)), dim3(256), 0, (hipStream_t)stream, x, ldx, dd,
                       lddd, dX, lddx, N, N * (N + 1) / 2, F, row0, nrows);
    return mft_launch_status();
}
