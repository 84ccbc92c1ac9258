// fp32-accurate implicit-GEMM convolution on the bf16 matrix cores ("bf16x3": 6-term split products).
//
// On CDNA4 the f32-input MFMA runs at the vector rate (157 TFLOP/s) while bf16 MFMA is 16x faster.  Every fp32
// value splits EXACTLY into three bf16 pieces x = x1 + x2 + x3 (8 + 8 + 8 significant bits, same exponent range),
// so a product a*b is the sum of nine bf16 x bf16 products, each exact in the fp32 accumulator.  Dropping the three
// terms of order 2^-24 and below (a2*b3, a3*b2, a3*b3) leaves six MFMAs per fp32 MFMA-equivalent whose summed error
// is at the level of ordinary fp32 rounding (measured: 2.5e-7 relative vs 4.1e-7 for an fp32 GEMM at K = 576..4608),
// at 16/6 = 2.67x the fp32-MFMA throughput (419 TFLOP/s equivalent peak).  Used for the FROZEN, shared-weight trunk
// convolutions (backbone.py:221-240 called from finetune.py:286 / gnnnet.py:168): their weights are split once at
// load time; activations are split on the fly in the A-tile loader (v_cvt_pk_bf16_f32 + two subtractions).
//
// Tile: 256 threads = 2 x 2 waves own BM x BN outputs; K walks in 32-element steps.  LDS holds the three bf16 planes
// of the A and B tiles, rows padded to 80 bytes so every ds_read_b128 fragment read (8 bf16 of one row) is
// conflict-free; single LDS buffer, next tile prefetched into registers under the MFMAs.
#include "mft_common.h"
#include "bn_fold.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct X3Args {
    const float* in;
    const unsigned short* w3;      // [3][Cout][Kpad] bf16 planes
    long long plane;               // elements per plane
    float* out;
    int ldi, ldo;
    int H, W, Cin, OH, OW, Cout, KH, KW, stride, pad;
    int Kpad;
    int M;                         // n_img * OH * OW
    int tiles_n;
    unsigned in_bytes, w_bytes;    // buffer-resource extents (< 2^31: out-of-range offsets are used as the zero-fill sentinel)
    int xcd_swizzle;
    int row_swz;                   // staging-row assignment that avoids LDS write bank conflicts
    const unsigned short* in3;     // AP kernels: pre-split input planes [3][n_img*H*W][ldi] bf16 (written by mft_bn_apply_planes)
    unsigned in_plane_bytes;
    float* stats_ws;               // optional [tiles_m][2][Cout][2]: per-tile (sum x, sum x^2) of the two BatchNorm groups a tile can touch
    int rows_per_group;            // >= BM when stats_ws is set
    int s1_rows;                   // conv_x3_s1_kernel: rows of the staged A image (BM + 2 halo pixels + 16 all-zero rows)
    // conv_x3_s1_kernel<.., BNIN = true>: ``in`` is the RAW output of the previous convolution; its train-mode BatchNorm + ReLU is
    // applied by the loader, with the statistics merged from that convolution's per-tile partials in the workgroup prologue
    const float* bn_ws; const float* bn_gamma; const float* bn_beta;
    float bn_eps;
    int bn_max_tiles;              // tiles one group's rows can overlap (stride of the prologue's staging area)
};

constexpr int X3_RS = 40;          // bf16 per LDS row: 32 data + 8 pad (80 B)

__device__ __forceinline__ unsigned pk_bf16(float a, float b) {
    f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}

// x (4 floats) -> three planes of 4 bf16 (2 dwords each), round-to-nearest-even pieces, exact residuals
__device__ __forceinline__ void split4(const f32x4 x, u32x2& p1, u32x2& p2, u32x2& p3) {
    f32x4 r;
    p1[0] = pk_bf16(x[0], x[1]);
    p1[1] = pk_bf16(x[2], x[3]);
    r[0] = x[0] - __builtin_bit_cast(float, p1[0] << 16);
    r[1] = x[1] - __builtin_bit_cast(float, p1[0] & 0xffff0000u);
    r[2] = x[2] - __builtin_bit_cast(float, p1[1] << 16);
    r[3] = x[3] - __builtin_bit_cast(float, p1[1] & 0xffff0000u);
    p2[0] = pk_bf16(r[0], r[1]);
    p2[1] = pk_bf16(r[2], r[3]);
    r[0] -= __builtin_bit_cast(float, p2[0] << 16);
    r[1] -= __builtin_bit_cast(float, p2[0] & 0xffff0000u);
    r[2] -= __builtin_bit_cast(float, p2[1] << 16);
    r[3] -= __builtin_bit_cast(float, p2[1] & 0xffff0000u);
    p3[0] = pk_bf16(r[0], r[1]);
    p3[1] = pk_bf16(r[2], r[3]);
}

// ------------------------------------------------------------------------------------------ "f16x2": two fp16 pieces, three products
// NP = 2 forms of the kernels below.  x = hi + 2^-11 lo with hi = fp16(x) (round to nearest even) and lo = fp16((x - hi) * 2^11):
// the residual x - hi is exact in fp32, the scaled residual has the magnitude of x itself -- so lo is a NORMAL fp16 number
// whenever hi is, whatever the magnitude of x -- and |x - (hi + 2^-11 lo)| <= 2^-22 |x|.  A product is then
//     a b = a_hi b_hi + 2^-11 (a_hi b_lo + a_lo b_hi) + O(2^-22 a b):
// THREE fp16 MFMAs per K-step instead of the six bf16 ones (same matrix rate per instruction: half the matrix work, two LDS
// planes per operand instead of three), with the leading term and the cross terms in two separate fp32 accumulators that the
// epilogue combines (acc0 + 2^-11 acc1) -- which also keeps the small terms from being rounded against the large partial sum.
// Error: 2^-22 relative per PRODUCT (random sign) against the fp32 rounding of every PARTIAL SUM that any fp32-accumulating
// GEMM makes; measured against float64 on the trunk shapes it stays below the fp32-MFMA kernel's error
// (tests/test_kernels_gpu.py::test_conv2d_f16x2_is_fp32_accurate).  Range: the operands must lie inside fp16's (|x| < 65504).  The
// callers are the frozen trunk's convolutions, whose inputs are train-mode BatchNorm outputs, |x| <= |gamma| sqrt(n) + |beta|, and
// whose weights are checked at load time (functional.ResNet10Weights: any tensor outside the bound keeps the bf16x3 kernels).
__device__ __forceinline__ void split4_h2(const f32x4 x, u32x2& p1, u32x2& p2) {
    const f16x4 hi = __builtin_convertvector(x, f16x4);
    const f32x4 r = (x - __builtin_convertvector(hi, f32x4)) * 2048.f;
    const f16x4 lo = __builtin_convertvector(r, f16x4);
    p1 = __builtin_bit_cast(u32x2, hi);
    p2 = __builtin_bit_cast(u32x2, lo);
}
constexpr float X3_H2_LO_SCALE = 1.0f / 2048.f;

// AP = true: the activation arrives already split into its three bf16 planes (the producing BatchNorm-apply / pooling kernel
// split every element ONCE, instead of this loader re-splitting it for each of the 9 taps and each n-tile): the A path is
// then the same plain 16-byte copy into LDS as the weight path, with no VALU work between the loads and the MFMAs.
// DB = true: two LDS buffers and ONE barrier per K-step: the split + LDS store of tile kt+1 goes to the other buffer, so it
// sits in the same basic block as the MFMAs of tile kt and the scheduler can issue it in their shadow (the single-buffer loop
// has barrier - store - barrier between two MFMA blocks; PMC: waves 47 % issue-stalled / 24 % parked, matrix pipes 40 % busy).
// Costs 92 KB of LDS per workgroup (one workgroup per CU instead of three).
// DBG (timing experiments only, results are wrong): bit 0 no operand split (raw bits stored three times), bit 1 no MFMA,
// bit 2 operands loaded once (no global loads in the K loop), bit 3 no LDS stores in the K loop, bit 4 no fragment reads,
// bit 5 no barriers in the K loop
// XS: 64-byte LDS rows (no padding) with an XOR swizzle of the four 16-byte chunks of a row, chunk' = chunk ^ ((row >> 2) & 3):
// fragment reads (16 lanes = 16 consecutive rows, one chunk) and staging writes (consecutive rows, whole rows) stay
// conflict-free, and the tile takes 36 KB instead of 46 KB of LDS -> FOUR workgroups per CU instead of three.
template <int BM, int BN, bool AP, bool DB, int HOIST = 0, int DBG = 0, bool XS = false, int NP = 3>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(NP == 2 ? 3 : 1))) void conv_x3_kernel(X3Args p) {
    static_assert(NP == 3 || (NP == 2 && !AP && !DB && HOIST == 0 && DBG == 0 && !XS), "f16x2: the plain single-buffer form only");
    constexpr int NACC = NP == 3 ? 1 : 2;
    constexpr int TM = BM / 64;           // 32-row blocks per wave (waves 2 x 2)
    constexpr int TN = BN / 64;
    constexpr int PA = AP ? BM / 64 : BM / 32;   // A passes: fp32: 32 rows x 8 threads x float4; planes: 64 rows x 4 threads x 16 B
    constexpr int PB = BN / 64;           // B passes: 64 rows per pass, 4 threads x 16 B per row and plane
    constexpr int RS = XS ? 32 : X3_RS;
    constexpr int A_PLANE = BM * RS;      // bf16 elements
    constexpr int B_PLANE = BN * RS;
    auto lo = [](int row, int c) -> int {          // element offset of bf16 column c (multiple of 4) of a row inside a plane
        if constexpr (XS) return row * 32 + ((((c >> 3) ^ (row >> 2)) & 3) << 3) + (c & 7);
        else return row * X3_RS + c;
    };
    extern __shared__ __attribute__((aligned(16))) unsigned short smem[];
    unsigned short* As = smem;                    // [NP][BM][RS]
    unsigned short* Bs = smem + NP * A_PLANE;     // [NP][BN][RS]
    constexpr int BUF = NP * (A_PLANE + B_PLANE);  // elements per LDS buffer (DB: two of them)

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int r = lane & 31, h = lane >> 5;
    const bool g_swz = !XS && p.row_swz != 0;
    // XCD-aware tile order.  Workgroup ids are dealt round-robin to the 8 XCDs, each with its own 4 MB L2.  With the
    // natural order the n-tiles of one m-tile (which read the SAME im2col rows) and neighbouring m-tiles (which share
    // halo rows) land on different XCDs and every L2 fetches the activation separately -- PMC showed ~6x the input size
    // in fabric reads.  Remap: XCD x works through one contiguous range of tiles (n fastest), so operand re-use stays
    // inside one L2.
    int tile_id = blockIdx.x;
    if (p.xcd_swizzle) {
        const int nwg = gridDim.x, q = nwg >> 3, rmd = nwg & 7;
        const int xcd = tile_id & 7, slot = tile_id >> 3;
        tile_id = xcd * q + (xcd < rmd ? xcd : rmd) + slot;
    }
    const int nt = tile_id % p.tiles_n;
    const int mt = tile_id / p.tiles_n;
    const int m0 = mt * BM, n0 = nt * BN;

    // Row owned by a thread in the staging passes.  LDS rows are 80 B = 20 banks apart, so four CONSECUTIVE rows written by one
    // 32-lane (ds_write_b64) or 16-lane (ds_write_b128) group wrap around the 64 banks and collide 2-way; rows 4 apart start
    // at banks 0/16/32/48.  Each group therefore owns rows {q, q+4, q+8, q+12} (PMC: 1/3 of the LDS-active cycles were conflicts).
    const int g_id = AP ? tid >> 4 : tid >> 5;                 // 16-lane groups of 4 threads/row, 32-lane groups of 8 threads/row
    const int g_rr = AP ? (tid >> 2) & 3 : (tid >> 3) & 3;
    const int lrow = g_swz ? g_rr * 4 + (g_id & 3) + 16 * (g_id >> 2) : (AP ? tid >> 2 : tid >> 3);
    const int c4 = AP ? (tid & 3) * 8 : (tid & 7) * 4;         // first channel of this thread's 16-byte piece
    constexpr int RPP = AP ? 64 : 32;                          // rows per A pass
    constexpr int ESZ = AP ? 2 : 4;                            // bytes per stored activation element
    const int ohw = p.OH * p.OW;
    // Operands are fetched with raw buffer loads: one 32-bit byte offset per lane, and an offset beyond the buffer
    // (the sentinel 0x80000000) returns zeros -- that IS the zero padding of the convolution and of ragged tiles, so the
    // im2col gather costs a compare + select per row instead of 64-bit address arithmetic under divergent branches.
    const __amdgpu_buffer_rsrc_t rA = AP ? __builtin_amdgcn_make_buffer_rsrc((void*)p.in3, 0, 3 * p.in_plane_bytes, 0x00020000)
                                         : __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, p.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc((void*)p.w3, 0, p.w_bytes, 0x00020000);
    int a_off[PA];                 // byte offset of (img, ih0, iw0, channel c4); meaningful only with a valid tap
    int a_ih0[PA], a_iw0[PA];
    bool a_ok[PA];
#pragma unroll
    for (int j = 0; j < PA; ++j) {
        const int m = m0 + lrow + RPP * j;
        a_ok[j] = m < p.M;
        const int mm = a_ok[j] ? m : 0;
        const int img = mm / ohw;
        const int rem = mm - img * ohw;
        const int oh = rem / p.OW, ow = rem - oh * p.OW;
        a_ih0[j] = oh * p.stride - p.pad;
        a_iw0[j] = ow * p.stride - p.pad;
        a_off[j] = (((img * p.H + a_ih0[j]) * p.W + a_iw0[j]) * p.ldi + c4) * ESZ;
    }
    const int bseg = tid & 3;
    const int brow = g_swz ? ((tid >> 2) & 3) * 4 + ((tid >> 4) & 3) + 16 * (tid >> 6) : tid >> 2;
    int b_off[PB];
#pragma unroll
    for (int j = 0; j < PB; ++j) b_off[j] = ((n0 + brow + 64 * j) * p.Kpad + bseg * 8) * 2;
    const int plane_bytes = (int)(p.plane * 2);

    f32x16 accs[NACC][TM][TN];
#pragma unroll
    for (int q = 0; q < NACC; ++q)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) accs[q][i][j][e] = 0.f;
    auto& acc = accs[0];
    auto val = [&](int i, int j, int e) -> float {      // the finished output element
        if constexpr (NP == 3) return accs[0][i][j][e];
        else return __builtin_fmaf(accs[NACC - 1][i][j][e], X3_H2_LO_SCALE, accs[0][i][j][e]);
    };

    // register stage: the operands of K-step kt+1 are in flight while kt is multiplied
    struct Stage {
        f32x4 ra[AP ? 1 : PA];
        u32x4 ra3[AP ? PA : 1][3];
        u32x4 rb[PB][NP];
    };
    Stage st0;
    const int nk = p.Kpad / 32;

    // tap bookkeeping of the K walk, carried incrementally (the tiles are requested in order): k0 = kt*32 -> (kh, kw, ci0).
    // The divisions k0 / Cin and khkw / KW cost ~40 dependent scalar instructions at the top of every K-step (PMC: 2.5 SALU
    // instructions per MFMA) in front of the operand requests.
    int t_kt = 0, t_ci0 = 0, t_kh = 0, t_kw = 0;
    auto load_tile = [&](int kt, Stage& S) {
        const int k0 = kt * 32;
        while (t_kt < kt) {                                // uniform; one iteration per call in the K loop
            ++t_kt;
            t_ci0 += 32;
            if (t_ci0 == p.Cin) { t_ci0 = 0; if (++t_kw == p.KW) { t_kw = 0; ++t_kh; } }
        }
        const int ci0 = t_ci0, kh = t_kh, kw = t_kw;
        const int tap_off = ((kh * p.W + kw) * p.ldi + ci0) * ESZ;
#pragma unroll
        for (int j = 0; j < PA; ++j) {
            const bool ok = a_ok[j] && (unsigned)(a_ih0[j] + kh) < (unsigned)p.H && (unsigned)(a_iw0[j] + kw) < (unsigned)p.W;
            const unsigned voff = ok ? (unsigned)(a_off[j] + tap_off) : 0x80000000u;
            if constexpr (AP) {
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
                    S.ra3[j][pl] = __builtin_amdgcn_raw_buffer_load_b128(rA, voff, pl * (int)p.in_plane_bytes, 0);
            } else {
                S.ra[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rA, voff, 0, 0));
            }
        }
#pragma unroll
        for (int j = 0; j < PB; ++j)
#pragma unroll
            for (int pl = 0; pl < NP; ++pl)
                S.rb[j][pl] = __builtin_amdgcn_raw_buffer_load_b128(rB, b_off[j] + pl * plane_bytes, k0 * 2, 0);
    };
    auto store_tile = [&](const Stage& S, int buf) {
        unsigned short* As = smem + buf * BUF;
        unsigned short* Bs = As + NP * A_PLANE;
#pragma unroll
        for (int j = 0; j < PA; ++j) {
            const int off = lo(lrow + RPP * j, c4);
            if constexpr (NP == 2) {
                u32x2 p1, p2;
                split4_h2(S.ra[j], p1, p2);
                *(u32x2*)(As + off) = p1;
                *(u32x2*)(As + A_PLANE + off) = p2;
            } else if constexpr (AP) {
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) *(u32x4*)(As + pl * A_PLANE + off) = S.ra3[j][pl];
            } else {
                u32x2 p1, p2, p3;
                if constexpr (DBG & 1) {
                    const u32x4 raw = __builtin_bit_cast(u32x4, S.ra[j]);
                    p1[0] = raw[0]; p1[1] = raw[1]; p2[0] = raw[2]; p2[1] = raw[3]; p3 = p1;
                } else {
                    split4(S.ra[j], p1, p2, p3);
                }
                *(u32x2*)(As + off) = p1;
                *(u32x2*)(As + A_PLANE + off) = p2;
                *(u32x2*)(As + 2 * A_PLANE + off) = p3;
            }
        }
#pragma unroll
        for (int j = 0; j < PB; ++j)
#pragma unroll
            for (int pl = 0; pl < NP; ++pl)
                *(u32x4*)(Bs + pl * B_PLANE + lo(brow + 64 * j, bseg * 8)) = S.rb[j][pl];
    };

    auto compute = [&](int buf) {
        const unsigned short* As = smem + buf * BUF;
        const unsigned short* Bs = As + NP * A_PLANE;
        if constexpr (NP == 2) {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                f16x8 a[TM][2], b[TN][2];
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int pl = 0; pl < 2; ++pl)
                        a[i][pl] = *(const f16x8*)(As + pl * A_PLANE + lo(wm * (BM / 2) + i * 32 + r, kk * 16 + h * 8));
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int pl = 0; pl < 2; ++pl)
                        b[j][pl] = *(const f16x8*)(Bs + pl * B_PLANE + lo(wn * (BN / 2) + j * 32 + r, kk * 16 + h * 8));
                constexpr int TA[3] = {0, 1, 0};
                constexpr int TB[3] = {1, 0, 0};
                constexpr int TQ[3] = {1, 1, 0};          // cross products -> accumulator 1, leading product -> accumulator 0
#pragma unroll
                for (int t = 0; t < 3; ++t)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            accs[TQ[t] ? NACC - 1 : 0][i][j] =
                                __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i][TA[t]], b[j][TB[t]], accs[TQ[t] ? NACC - 1 : 0][i][j], 0, 0, 0);
            }
            return;
        }
        // ALL fragment reads of the K-step (both 16-wide halves: 2 x 3 x (TM + TN) ds_read_b128) are issued before the first
        // MFMA and pinned there (sched_barrier): left to itself the scheduler sinks each read to just before the MFMA that
        // consumes it to save registers, which exposes one LDS round trip per MFMA (ISA of round 1: "ds_read, s_waitcnt, mfma"
        // six times per half -- the matrix pipes measured 43 % busy).  In-order lgkmcnt lets the first MFMA start as soon as ITS
        // two fragments are in while the remaining reads complete under the matrix work.
        if constexpr (HOIST == 0) {
            // round-1 form: fragments read per 16-wide half, scheduling left to the compiler
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                bf16x8 a[TM][3], b[TN][3];
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl)
                        a[i][pl] = *(const bf16x8*)(As + pl * A_PLANE + lo(wm * (BM / 2) + i * 32 + r, kk * 16 + h * 8));
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl)
                        b[j][pl] = *(const bf16x8*)(Bs + pl * B_PLANE + lo(wn * (BN / 2) + j * 32 + r, kk * 16 + h * 8));
                constexpr int TA[6] = {2, 0, 1, 1, 0, 0};
                constexpr int TB[6] = {0, 2, 1, 0, 1, 0};
                if constexpr (DBG & 2) {
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
#pragma unroll
                            for (int pl = 0; pl < 3; ++pl)
                                acc[i][j][pl] += (float)a[i][pl][0] + (float)b[j][pl][1];      // keeps the reads alive
                } else {
#pragma unroll
                    for (int t = 0; t < 6; ++t)
#pragma unroll
                        for (int i = 0; i < TM; ++i)
#pragma unroll
                            for (int j = 0; j < TN; ++j)
                                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][TA[t]], b[j][TB[t]], acc[i][j], 0, 0, 0);
                }
            }
            return;
        }
        bf16x8 a[2][TM][3], b[2][TN][3];
        if constexpr (HOIST == 1) {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
                    for (int i = 0; i < TM; ++i)
                        a[kk][i][pl] = *(const bf16x8*)(As + pl * A_PLANE + lo(wm * (BM / 2) + i * 32 + r, kk * 16 + h * 8));
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        b[kk][j][pl] = *(const bf16x8*)(Bs + pl * B_PLANE + lo(wn * (BN / 2) + j * 32 + r, kk * 16 + h * 8));
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            constexpr int TA[6] = {0, 0, 1, 1, 0, 2};
            constexpr int TB[6] = {0, 1, 0, 1, 2, 0};
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int t = 0; t < 6; ++t)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[kk][i][TA[t]], b[kk][j][TB[t]], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        } else {
            // HOIST == 2: half 0's nine reads, then half 1's nine reads issued in the shadow of half 0's MFMAs
            auto rd = [&](int kk) {
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
                    for (int i = 0; i < TM; ++i)
                        a[kk][i][pl] = *(const bf16x8*)(As + pl * A_PLANE + lo(wm * (BM / 2) + i * 32 + r, kk * 16 + h * 8));
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        b[kk][j][pl] = *(const bf16x8*)(Bs + pl * B_PLANE + lo(wn * (BN / 2) + j * 32 + r, kk * 16 + h * 8));
                }
            };
            constexpr int TA[6] = {0, 0, 1, 1, 0, 2};
            constexpr int TB[6] = {0, 1, 0, 1, 2, 0};
            auto mm = [&](int kk) {
#pragma unroll
                for (int t = 0; t < 6; ++t)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[kk][i][TA[t]], b[kk][j][TB[t]], acc[i][j], 0, 0, 0);
            };
            rd(0);
            __builtin_amdgcn_sched_barrier(0);
            rd(1);
            mm(0);
            // one MFMA, then one or two reads, ... : the second half's reads ride under the first half's matrix work
            for (int q = 0; q < 6 * TM * TN; ++q) {
                __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, (3 * (TM + TN) + 6 * TM * TN - 1) / (6 * TM * TN), 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            mm(1);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // Rejected variants of this loop (measured, removed again): two K-steps of operands in flight (second register stage):
    // neutral in situ, +3 % standalone time from the extra registers; s_setprio around the MFMA block or around the split/store
    // block: no gain from the priority, and the run-time branches alone cost 40 % -- keep the K-step body branch-free.
    if constexpr (DB) {
        load_tile(0, st0);
        store_tile(st0, 0);
        __syncthreads();
        load_tile(nk > 1 ? 1 : 0, st0);
        for (int kt = 0; kt + 1 < nk; ++kt) {      // branch-free body: the last tile is simply requested twice
            compute(kt & 1);
            store_tile(st0, (kt + 1) & 1);
            load_tile(kt + 2 < nk ? kt + 2 : nk - 1, st0);
            __syncthreads();
        }
        compute((nk - 1) & 1);
    } else {
        load_tile(0, st0);
        store_tile(st0, 0);
        __syncthreads();
        for (int kt = 0; kt < nk; ++kt) {
            if constexpr (!(DBG & 4)) { if (kt + 1 < nk) load_tile(kt + 1, st0); }
            if constexpr (!(DBG & 16)) compute(0);
            if constexpr (!(DBG & 32)) __syncthreads();
            if constexpr (!(DBG & 8)) { if (kt + 1 < nk) store_tile(st0, 0); }
            if constexpr (!(DBG & 32)) __syncthreads();
        }
    }

    // epilogue: C/D layout of the 32x32 MFMA: col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn * (BN / 2) + j * 32 + r;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = (e & 3) + 8 * (e >> 2) + 4 * h;
                const int m = m0 + wm * (BM / 2) + i * 32 + row;
                if (m < p.M) p.out[(long long)m * p.ldo + n] = val(i, j, e);
            }
        }
    // Fused BatchNorm statistics (the consumer of every trunk convolution is a train-mode BatchNorm, backbone.py:224-227):
    // the tile is still in registers, so its per-channel (sum x, sum x^2) cost no memory traffic -- the separate statistics
    // pass re-read every convolution output (1.1 GB per lockstep step at E = 128).  A tile of BM <= rows_per_group rows
    // touches at most two groups: segment 0 = rows of the group the tile starts in, segment 1 = rows of the next group.
    // Partials are combined by x3_stats_finalize_kernel with Chan's formula in a fixed order (deterministic).
    if (p.stats_ws != nullptr) {
        static_assert(TN == 1 || TN == 2, "stats epilogue");
        const int split = (m0 / p.rows_per_group + 1) * p.rows_per_group;     // first row of the next group
        float* sred = reinterpret_cast<float*>(smem);                         // [2 wm][2 wn][TN][32 r][4]
        __syncthreads();                                                      // all fragment reads of the last K-step are done
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            float a1 = 0.f, a2 = 0.f, b1 = 0.f, b2 = 0.f;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int m = m0 + wm * (BM / 2) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                    const float v = val(i, j, e);
                    if (m < p.M) {
                        if (m < split) { a1 += v; a2 += v * v; }
                        else { b1 += v; b2 += v * v; }
                    }
                }
            a1 += __shfl_xor(a1, 32, 64); a2 += __shfl_xor(a2, 32, 64);
            b1 += __shfl_xor(b1, 32, 64); b2 += __shfl_xor(b2, 32, 64);
            if (h == 0) {
                float* o = sred + ((((wm * 2 + wn) * TN + j) * 32 + r) << 2);
                o[0] = a1; o[1] = a2; o[2] = b1; o[3] = b2;
            }
        }
        __syncthreads();
        if (wm == 0 && h == 0) {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const float* o0 = sred + ((((0 * 2 + wn) * TN + j) * 32 + r) << 2);
                const float* o1 = sred + ((((1 * 2 + wn) * TN + j) * 32 + r) << 2);
                const int n = n0 + wn * (BN / 2) + j * 32 + r;
                float* w0 = p.stats_ws + (((long long)mt * 2 + 0) * p.Cout + n) * 2;
                float* w1 = p.stats_ws + (((long long)mt * 2 + 1) * p.Cout + n) * 2;
                w0[0] = o0[0] + o1[0]; w0[1] = o0[1] + o1[1];
                w1[0] = o0[2] + o1[2]; w1[1] = o0[3] + o1[3];
            }
        }
    }
}

#ifdef MFT_EXPERIMENTS      // measured slower than the default kernels; built only for tools/ (MFT_EXPERIMENTS=1)
#include "../../tools/experiments/conv_x3_pingpong.inc"
#endif  // MFT_EXPERIMENTS

// ------------------------------------------------------------------------------------------ 3x3 / stride 1 / pad 1: shared taps
// The implicit-GEMM kernel above fetches, splits and stores the A tile of every tap separately, although the three kw taps of one
// kernel row read the SAME pixels shifted by one: with OW == W the flat output index m is also the flat input pixel index, and tap
// (kh, kw) of output m needs pixel m + (kh-1) W + (kw-1) -- or zero where ow + kw - 1 leaves the image row.  This form stages, per
// (kh, 32-channel slice), ONE image of the BM pixels m0 .. m0+BM-1 shifted by (kh-1) rows, plus one halo pixel at each end and
// sixteen all-zero rows (LDS row of pixel m: (m - m0) + 1), and the three kw taps read their fragments from it at row offsets
// -1 / 0 / +1 -- except that a lane whose output pixel is the first / last of its image row reads a zero row for kw = 0 / kw = 2:
// that IS the horizontal padding (one address select per lane and tap, no masking of data).  Global loads, bf16x3 splits and LDS stores of the A operand drop to a third; the weights (B) are
// staged per tap as before.  K is walked (kh, ci, kw) instead of (kh, kw, ci): the same products, summed in another order.
// The launch is power-limited: the saving is in joules first, in issue slots second.
// BNIN: the input is the raw output c of the previous 3x3 convolution and this kernel consumes relu(BatchNorm(c)) (SimpleBlock:
// C1 -> BN1 -> ReLU -> C2, backbone.py:251-256): the input has the same pixel grid as the output (stride 1), so its BatchNorm groups
// are the output's (rows_per_group), a tile touches at most two of them, and the prologue merges their statistics from C1's
// partials (bn_fold.h) into an LDS table of (scale, shift) per (group, channel).  The loader then applies one FMA + max per
// element in front of the bf16x3 split; padding and zero rows stay exact zeros.  Removes BN1's finalize and apply launches and
// the activation they wrote and re-read.
template <int BM, int BN, bool BNIN, int NP = 3, bool BDB = true>
__device__ __forceinline__ void conv_x3_s1_body(const X3Args& p) {
    static_assert(NP == 3 || NP == 2, "3 bf16 pieces (six products) or 2 fp16 pieces (three products)");
    constexpr int NACC = NP == 3 ? 1 : 2;           // f16x2: leading products / cross products in separate accumulators
    constexpr int TM = BM / 64, TN = BN / 64;
    constexpr int PA = BM / 32, PB = BN / 64;
    constexpr int RS = X3_RS;
    constexpr int B_PLANE = BN * RS;
    const int A_PLANE = p.s1_rows * RS;
    extern __shared__ __attribute__((aligned(16))) unsigned short smem[];
    // f16x2 (NP == 2): TWO weight-tile buffers, so that the next tap's tile is stored while slower waves still multiply the
    // current one -- one barrier per tap instead of two (round 5); the bf16x3 form keeps one (a second would cost its third
    // workgroup per CU)
    constexpr int NBB = (NP == 2 && BDB) ? 2 : 1;        // (BDB = false: the single-buffer form, mft_debug_set_x3_tile(30), for A/B)
    unsigned short* As = smem;                       // [NP][s1_rows][RS]
    unsigned short* Bs = smem + NP * A_PLANE;        // [NBB][NP][BN][RS]
    float* bn_tab = reinterpret_cast<float*>(Bs + NBB * NP * B_PLANE);     // BNIN: [2 groups][scale | shift][Cin]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int r = lane & 31, h = lane >> 5;
    int tile_id = blockIdx.x;
    if (p.xcd_swizzle) {
        const int nwg = gridDim.x, q = nwg >> 3, rmd = nwg & 7;
        const int xcd = tile_id & 7, slot = tile_id >> 3;
        tile_id = xcd * q + (xcd < rmd ? xcd : rmd) + slot;
    }
    const int nt = tile_id % p.tiles_n;
    const int mt = tile_id / p.tiles_n;
    const int m0 = mt * BM, n0 = nt * BN;
    const int W = p.W, H = p.H;
    const int fr0 = m0 / W;                          // flat image-row index (img * H + oh) of the tile's first pixel
    const int g0 = BNIN ? m0 / p.rows_per_group : 0; // first BatchNorm group of the tile
    const bool bn_two = BNIN && (m0 + (BM < p.M - m0 ? BM : p.M - m0) - 1) / p.rows_per_group > g0;      // the tile straddles two groups
    if constexpr (BNIN) {
        // the partials of the tile's (one or two) groups are first copied into LDS with independent loads -- the stage aliases the
        // A planes, which are not written yet -- then 2*Cin threads run the merge chains out of LDS
        char* stage = reinterpret_cast<char*>(smem);
        const size_t gstride = mft_x3_stage_bytes(p.bn_max_tiles, p.Cin);
        const bool two = bn_two;
        mft_x3_stats_stage(p.bn_ws, p.Cin, g0, p.M, p.rows_per_group, BM, p.bn_max_tiles, stage, tid, 256);
        if (two) mft_x3_stats_stage(p.bn_ws, p.Cin, g0 + 1, p.M, p.rows_per_group, BM, p.bn_max_tiles, stage + gstride, tid, 256);
        __syncthreads();
        for (int i = tid; i < 2 * p.Cin; i += 256) {
            const int gi = i >= p.Cin ? 1 : 0, c = i - gi * p.Cin;
            if (gi == 0 || two) {
                float mu, rs, sc, sh;
                mft_x3_stats_staged(stage + gi * gstride, p.bn_max_tiles, p.Cin, c, g0 + gi, p.M, p.rows_per_group, BM, p.bn_eps, mu, rs);
                mft_bn_fold(mu, rs, p.bn_gamma[c], p.bn_beta[c], sc, sh);
                bn_tab[(gi * 2 + 0) * p.Cin + c] = sc;
                bn_tab[(gi * 2 + 1) * p.Cin + c] = sh;
            }
        }
        __syncthreads();                                  // the zero fill below overwrites the stage
    }
    const int g_id = tid >> 5, g_rr = (tid >> 3) & 3;
    const int lrow = p.row_swz ? g_rr * 4 + (g_id & 3) + 16 * (g_id >> 2) : tid >> 3;
    const int c4 = (tid & 7) * 4;
    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, p.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc((void*)p.w3, 0, p.w_bytes, 0x00020000);
    int a_off[PA], a_oh[PA], a_lds[PA], a_tab[PA];
    bool a_ok[PA];
#pragma unroll
    for (int j = 0; j < PA; ++j) {
        const int t = lrow + 32 * j, m = m0 + t;
        const int fr = m / W;
        a_ok[j] = m < p.M;
        a_oh[j] = fr - (fr / H) * H;
        a_off[j] = ((a_ok[j] ? m : 0) * p.ldi + c4) * 4;           // (rows beyond M are never requested; keep the product in range)
        a_lds[j] = (t + 1) * RS + c4;
        a_tab[j] = BNIN ? ((a_ok[j] ? m / p.rows_per_group - g0 : 0) * 2 * p.Cin + c4) : 0;
    }
    // halo pixels m0 - 1 and m0 + BM (threads 0-7 / 8-15): real only when they lie in the same image row as their neighbour
    const int hside = (tid >> 3) & 1;
    const int mh = hside ? m0 + BM : m0 - 1;
    const int frh = (mh > 0 ? mh : 0) / W;
    const bool h_ok = tid < 16 && (hside ? (mh < p.M && mh - frh * W != 0) : (m0 > 0 && m0 - fr0 * W != 0));
    const int h_oh = frh - (frh / H) * H;
    const int h_off = ((h_ok ? mh : 0) * p.ldi + c4) * 4;
    const int h_lds = (hside ? BM + 1 : 0) * RS + c4;
    // a real halo pixel lies in the same image (hence the same group) as its neighbour inside the tile
    const int h_tab = BNIN ? (((hside && h_ok) ? (m0 + BM - 1) / p.rows_per_group - g0 : 0) * 2 * p.Cin + c4) : 0;

    const int bseg = tid & 3;
    const int brow = p.row_swz ? ((tid >> 2) & 3) * 4 + ((tid >> 4) & 3) + 16 * (tid >> 6) : tid >> 2;
    int b_off[PB];
#pragma unroll
    for (int j = 0; j < PB; ++j) b_off[j] = ((n0 + brow + 64 * j) * p.Kpad + bseg * 8) * 2;
    const int plane_bytes = (int)(p.plane * 2);

    // fragment rows of this lane inside the staged image
    // LDS row of pixel m0 + t is t + 1 (rows 0 and BM + 1: the halo pixels; rows BM + 2 .. BM + 17: zeros, never written after the
    // fill).  A lane whose output pixel sits at the left / right end of its image row reads a ZERO row for the kw = 0 / kw = 2 tap --
    // that is the horizontal padding -- namely the one of the sixteen that shares its banks with the row it replaces (rows 16 apart
    // alias at the 80-byte pitch), so the 16 lanes of a ds_read_b128 still hit 64 distinct banks.  (Round 2's layout put one zero
    // row at every image-row boundary INSIDE the image instead: the 16 lanes then spanned 17-19 LDS rows and 32 % of the LDS cycles
    // of these kernels were bank conflicts -- PMC, profiles/r04_g_lds_conflicts.txt.)
    int fa[TM], fl[TM], fr_[TM];
    constexpr int Z0 = BM + 2;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int t = wm * (BM / 2) + i * 32 + r;
        const int ow = (m0 + t) - ((m0 + t) / W) * W;
        fa[i] = (t + 1) * RS;
        fl[i] = ow == 0 ? (Z0 + ((t - Z0) & 15)) * RS : fa[i] - RS;              // replaces row t
        fr_[i] = ow == W - 1 ? (Z0 + ((t + 2 - Z0) & 15)) * RS : fa[i] + RS;      // replaces row t + 2
    }

    f32x16 acc[NACC][TM][TN];
#pragma unroll
    for (int q = 0; q < NACC; ++q)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[q][i][j][e] = 0.f;
    auto val = [&](int i, int j, int e) -> float {      // the finished output element
        if constexpr (NP == 3) return acc[0][i][j][e];
        else return __builtin_fmaf(acc[1][i][j][e], X3_H2_LO_SCALE, acc[0][i][j][e]);
    };

    struct Stage {
        f32x4 ra[PA];
        f32x4 rh;
        u32x4 rb[PB][NP];
    };
    Stage st;
    const int n_ci = p.Cin / 32;
    const int n_ms = 3 * n_ci;                       // macro steps: (kh, 32-channel slice), three kw taps each
    auto load_a = [&](int ms, Stage& S) {
        const int kh = ms / n_ci, ci0 = (ms - kh * n_ci) * 32;
        const int shift = ((kh - 1) * W * p.ldi + ci0) * 4;
#pragma unroll
        for (int j = 0; j < PA; ++j) {
            const bool ok = a_ok[j] && (unsigned)(a_oh[j] + kh - 1) < (unsigned)H;
            S.ra[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rA, ok ? (unsigned)(a_off[j] + shift) : 0x80000000u, 0, 0));
        }
        const bool okh = h_ok && (unsigned)(h_oh + kh - 1) < (unsigned)H;
        S.rh = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rA, okh ? (unsigned)(h_off + shift) : 0x80000000u, 0, 0));
    };
    auto load_b = [&](int ms, int kw, Stage& S) {
        const int kh = ms / n_ci, ci0 = (ms - kh * n_ci) * 32;
        const int k0 = (kh * 3 + kw) * p.Cin + ci0;
#pragma unroll
        for (int j = 0; j < PB; ++j)
#pragma unroll
            for (int pl = 0; pl < NP; ++pl)
                S.rb[j][pl] = __builtin_amdgcn_raw_buffer_load_b128(rB, b_off[j] + pl * plane_bytes, k0 * 2, 0);
    };
    auto put_a = [&](const f32x4 v, int off) {
        if constexpr (NP == 3) {
            u32x2 p1, p2, p3;
            split4(v, p1, p2, p3);
            *(u32x2*)(As + off) = p1;
            *(u32x2*)(As + A_PLANE + off) = p2;
            *(u32x2*)(As + 2 * A_PLANE + off) = p3;
        } else {
            u32x2 p1, p2;
            split4_h2(v, p1, p2);
            *(u32x2*)(As + off) = p1;
            *(u32x2*)(As + A_PLANE + off) = p2;
        }
    };
    auto bn_relu = [&](const f32x4 v, const f32x4 sc, const f32x4 sh, bool ok) {      // relu(BatchNorm(v)); masked (padding) elements stay 0
        f32x4 o = mft_bn_affine4(v, sc, sh);
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = ok ? fmaxf(o[e], 0.f) : 0.f;
        return o;
    };
    auto store_a = [&](const Stage& S, int ms) {
        if constexpr (BNIN) {
            const int kh = ms / n_ci, ci0 = (ms - kh * n_ci) * 32;
            const bool okh = h_ok && (unsigned)(h_oh + kh - 1) < (unsigned)H;
            if (!bn_two) {                      // (most tiles lie inside one group: one table row per K-slice for every pixel)
                const f32x4 sc = *(const f32x4*)(bn_tab + ci0 + c4), sh = *(const f32x4*)(bn_tab + p.Cin + ci0 + c4);
#pragma unroll
                for (int j = 0; j < PA; ++j)
                    put_a(bn_relu(S.ra[j], sc, sh, a_ok[j] && (unsigned)(a_oh[j] + kh - 1) < (unsigned)H), a_lds[j]);
                if (tid < 16) put_a(bn_relu(S.rh, sc, sh, okh), h_lds);
            } else {
#pragma unroll
                for (int j = 0; j < PA; ++j)
                    put_a(bn_relu(S.ra[j], *(const f32x4*)(bn_tab + a_tab[j] + ci0), *(const f32x4*)(bn_tab + a_tab[j] + p.Cin + ci0),
                                  a_ok[j] && (unsigned)(a_oh[j] + kh - 1) < (unsigned)H), a_lds[j]);
                if (tid < 16)
                    put_a(bn_relu(S.rh, *(const f32x4*)(bn_tab + h_tab + ci0), *(const f32x4*)(bn_tab + h_tab + p.Cin + ci0), okh), h_lds);
            }
        } else {
#pragma unroll
            for (int j = 0; j < PA; ++j) put_a(S.ra[j], a_lds[j]);
            if (tid < 16) put_a(S.rh, h_lds);
        }
    };
    auto store_b = [&](const Stage& S, int bb = 0) {
#pragma unroll
        for (int j = 0; j < PB; ++j)
#pragma unroll
            for (int pl = 0; pl < NP; ++pl)
                *(u32x4*)(Bs + (bb * NP + pl) * B_PLANE + (brow + 64 * j) * RS + bseg * 8) = S.rb[j][pl];
    };
    auto compute = [&](int kw, int bb = 0) {
        const unsigned short* const Bc = Bs + bb * NP * B_PLANE;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            if constexpr (NP == 3) {
                bf16x8 a[TM][3], b[TN][3];
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl)
                        a[i][pl] = *(const bf16x8*)(As + pl * A_PLANE + (kw == 0 ? fl[i] : (kw == 1 ? fa[i] : fr_[i])) + kk * 16 + h * 8);
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl)
                        b[j][pl] = *(const bf16x8*)(Bc + pl * B_PLANE + (wn * (BN / 2) + j * 32 + r) * RS + kk * 16 + h * 8);
                constexpr int TA[6] = {2, 0, 1, 1, 0, 0};
                constexpr int TB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
                for (int t = 0; t < 6; ++t)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[0][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][TA[t]], b[j][TB[t]], acc[0][i][j], 0, 0, 0);
            } else {
                f16x8 a[TM][2], b[TN][2];
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int pl = 0; pl < 2; ++pl)
                        a[i][pl] = *(const f16x8*)(As + pl * A_PLANE + (kw == 0 ? fl[i] : (kw == 1 ? fa[i] : fr_[i])) + kk * 16 + h * 8);
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int pl = 0; pl < 2; ++pl)
                        b[j][pl] = *(const f16x8*)(Bc + pl * B_PLANE + (wn * (BN / 2) + j * 32 + r) * RS + kk * 16 + h * 8);
                // cross products first (hi x lo, lo x hi -> acc 1), leading product last (-> acc 0): neighbouring MFMAs of one
                // (i, j) alternate between the two accumulators, so none waits for its predecessor's result
                constexpr int TA[3] = {0, 1, 0};
                constexpr int TB[3] = {1, 0, 0};
                constexpr int TQ[3] = {1, 1, 0};
#pragma unroll
                for (int t = 0; t < 3; ++t)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[TQ[t]][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i][TA[t]], b[j][TB[t]], acc[TQ[t]][i][j], 0, 0, 0);
            }
        }
    };

    // the zero rows (image-row boundaries, halo rows that fall outside their image row) are written once and never touched again
    for (int i = tid; i < NP * A_PLANE / 8; i += 256) ((u32x4*)As)[i] = (u32x4){0u, 0u, 0u, 0u};
    load_a(0, st);
    load_b(0, 0, st);
    __syncthreads();
    store_a(st, 0);
    store_b(st);
    __syncthreads();
    if constexpr (NBB == 2) {
        // Tap s multiplies weight buffer s & 1 while tap s + 1's tile goes into the other one: that buffer was last read by tap
        // s - 1, which every wave finished before the barrier that closed it.  The staged image A is single: its refill waits
        // for the third tap of all waves (two barriers there) -- four barriers per (kh, channel slice) instead of six.
        int cur = 0;
        for (int ms = 0; ms < n_ms; ++ms) {
            const int msn = ms + 1 < n_ms ? ms + 1 : ms;      // branch-free body: the last image is simply requested twice
            load_b(ms, 1, st);
            compute(0, cur);
            store_b(st, cur ^ 1);
            __syncthreads();
            load_b(ms, 2, st);
            compute(1, cur ^ 1);
            store_b(st, cur);
            __syncthreads();
            load_a(msn, st);
            load_b(msn, 0, st);
            compute(2, cur);
            store_b(st, cur ^ 1);
            __syncthreads();
            store_a(st, msn);
            __syncthreads();
            cur ^= 1;
        }
    } else {
        for (int ms = 0; ms < n_ms; ++ms) {
            const int msn = ms + 1 < n_ms ? ms + 1 : ms;      // branch-free body: the last image is simply requested twice
            load_b(ms, 1, st);
            compute(0);
            __syncthreads();
            store_b(st);
            __syncthreads();
            load_b(ms, 2, st);
            compute(1);
            __syncthreads();
            store_b(st);
            __syncthreads();
            load_a(msn, st);
            load_b(msn, 0, st);
            compute(2);
            __syncthreads();
            store_a(st, msn);
            store_b(st);
            __syncthreads();
        }
    }

    // epilogue (as conv_x3_kernel): C/D layout of the 32x32 MFMA: col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn * (BN / 2) + j * 32 + r;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = (e & 3) + 8 * (e >> 2) + 4 * h;
                const int m = m0 + wm * (BM / 2) + i * 32 + row;
                if (m < p.M) p.out[(long long)m * p.ldo + n] = val(i, j, e);
            }
        }
    if (p.stats_ws != nullptr) {
        static_assert(TN == 1 || TN == 2, "stats epilogue");
        const int split = (m0 / p.rows_per_group + 1) * p.rows_per_group;
        float* sred = reinterpret_cast<float*>(smem);
        __syncthreads();
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            float a1 = 0.f, a2 = 0.f, b1 = 0.f, b2 = 0.f;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int m = m0 + wm * (BM / 2) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                    const float v = val(i, j, e);
                    if (m < p.M) {
                        if (m < split) { a1 += v; a2 += v * v; }
                        else { b1 += v; b2 += v * v; }
                    }
                }
            a1 += __shfl_xor(a1, 32, 64); a2 += __shfl_xor(a2, 32, 64);
            b1 += __shfl_xor(b1, 32, 64); b2 += __shfl_xor(b2, 32, 64);
            if (h == 0) {
                float* o = sred + ((((wm * 2 + wn) * TN + j) * 32 + r) << 2);
                o[0] = a1; o[1] = a2; o[2] = b1; o[3] = b2;
            }
        }
        __syncthreads();
        if (wm == 0 && h == 0) {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const float* o0 = sred + ((((0 * 2 + wn) * TN + j) * 32 + r) << 2);
                const float* o1 = sred + ((((1 * 2 + wn) * TN + j) * 32 + r) << 2);
                const int n = n0 + wn * (BN / 2) + j * 32 + r;
                float* w0 = p.stats_ws + (((long long)mt * 2 + 0) * p.Cout + n) * 2;
                float* w1 = p.stats_ws + (((long long)mt * 2 + 1) * p.Cout + n) * 2;
                w0[0] = o0[0] + o1[0]; w0[1] = o0[1] + o1[1];
                w1[0] = o0[2] + o1[2]; w1[1] = o0[3] + o1[3];
            }
        }
    }
}

template <int BM, int BN, int NP = 3, bool BDB = true>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(NP == 2 ? 3 : 1))) void conv_x3_s1_kernel(X3Args p) {
    conv_x3_s1_body<BM, BN, false, NP, BDB>(p);
}

// (three waves per SIMD is what the trunk is tuned for: the loader-side BatchNorm must fit the same 168 registers)
template <int BM, int BN, int NP = 3, bool BDB = true>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void conv_x3_s1_bnin_kernel(X3Args p) {
    conv_x3_s1_body<BM, BN, true, NP, BDB>(p);
}

// mean / rstd of every (group, channel) from the per-tile partials: tile t of BM rows overlaps group g in n_t rows;
// (n_t, mean_t = s1/n_t, M2_t = s2 - s1^2/n_t) are merged in tile order with Chan's update.
__global__ void x3_stats_finalize_kernel(const float* __restrict__ ws, int C, int M, int R, int BM, float eps,
                                         float* __restrict__ mean, float* __restrict__ rstd) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    const int g = blockIdx.y;
    if (c >= C) return;
    float mu, rs;
    mft_x3_stats_merge(ws, C, c, g, M, R, BM, eps, mu, rs);
    mean[(long long)g * C + c] = mu;
    rstd[(long long)g * C + c] = rs;
}

__global__ __launch_bounds__(256) void split_bf16x3_kernel(const float* __restrict__ w, unsigned short* __restrict__ out,
                                                           long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float x = w[i];
        const unsigned q1 = pk_bf16(x, 0.f) & 0xffffu;
        const float r1 = x - __builtin_bit_cast(float, q1 << 16);
        const unsigned q2 = pk_bf16(r1, 0.f) & 0xffffu;
        const float r2 = r1 - __builtin_bit_cast(float, q2 << 16);
        const unsigned q3 = pk_bf16(r2, 0.f) & 0xffffu;
        out[i] = (unsigned short)q1;
        out[n + i] = (unsigned short)q2;
        out[2 * n + i] = (unsigned short)q3;
    }
}

// Several weight matrices split in ONE launch (meta-training: the weights change every step, so the planes of every layer that
// runs on these kernels are refreshed per step; round 5).  Job = {packed fp32 source [Cout][taps][Cin], planes [3][n], n = Cout *
// taps * Cin, Cout, Cin, taps, transposed, first element}; ``transposed``: the planes hold the DATA-GRADIENT operand of a stride-1
// convolution, wt[ci][taps - 1 - tap][co] = w[co][tap][ci] (the tap-flipped, channel-swapped weights: dx = conv(dy, wt), same
// padding) -- writes coalesced along co, reads strided (the matrices are L2-sized).  Same arithmetic as split_bf16x3_kernel.
struct SplitJob { const float* src; unsigned short* dst; long long n, Cout, Cin, taps, transposed, start; };

__global__ __launch_bounds__(256) void split_bf16x3_multi_kernel(const SplitJob* __restrict__ jobs, int n_jobs, long long total) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        int lo = 0, hi = n_jobs - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (jobs[mid].start <= i) lo = mid; else hi = mid - 1;
        }
        const SplitJob j = jobs[lo];
        const long long l = i - j.start;
        long long src = l;
        if (j.transposed) {
            const long long co = l % j.Cout, t = l / j.Cout;
            const long long tp = t % j.taps, ci = t / j.taps;
            src = (co * j.taps + (j.taps - 1 - tp)) * j.Cin + ci;
        }
        const float x = j.src[src];
        const unsigned q1 = pk_bf16(x, 0.f) & 0xffffu;
        const float r1 = x - __builtin_bit_cast(float, q1 << 16);
        const unsigned q2 = pk_bf16(r1, 0.f) & 0xffffu;
        const float r2 = r1 - __builtin_bit_cast(float, q2 << 16);
        const unsigned q3 = pk_bf16(r2, 0.f) & 0xffffu;
        j.dst[l] = (unsigned short)q1;
        j.dst[j.n + l] = (unsigned short)q2;
        j.dst[2 * j.n + l] = (unsigned short)q3;
    }
}

// weights for the f16x2 kernels: [2][n] fp16 planes (hi, lo * 2^11), the same arithmetic as split4_h2
__global__ __launch_bounds__(256) void split_f16x2_kernel(const float* __restrict__ w, unsigned short* __restrict__ out, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float x = w[i];
        const _Float16 hi = (_Float16)x;
        const _Float16 lo = (_Float16)((x - (float)hi) * 2048.f);
        out[i] = __builtin_bit_cast(unsigned short, hi);
        out[n + i] = __builtin_bit_cast(unsigned short, lo);
    }
}

int g_x3_db = 0;           // double-buffered LDS form of the 128x64 kernel (mft_debug_set_x3_tile(60/61))
int g_x3_dbg = 0;          // timing experiments (wrong results): mft_debug_set_x3_tile(200 + bits), see conv_x3_kernel
int g_x3_hoist = 0;        // fragment-read schedule of the K-step: 0 compiler's (default), 1 all 18 reads pinned before the MFMAs, 2 second half's reads
                           // under the first half's MFMAs (mft_debug_set_x3_tile(80 + v)).  Measured standalone over the five trunk shapes, one
                           // process: 608 / 609 / 600 us -- with three waves per SIMD the exposed LDS round trips of one wave are covered by
                           // the others; the schedule inside a wave is not what holds the matrix pipes at 43 %.
int g_x3_row_swz = 1;      // LDS layout (mft_debug_set_x3_tile(40 + v)): 0 80-byte rows, natural staging rows; 1 80-byte rows, conflict-free
                           // staging-row assignment (default); 2 64-byte rows with an XOR chunk swizzle: 36 KB per tile, 4 workgroups per CU
                           // instead of 3 (bit-identical; measured 643 vs 634 us over the five trunk shapes and 72.8 vs 73.4 episodes/s:
                           // occupancy is not the limiter either)
int g_x3_xcd = 1;          // XCD-aware tile order (mft_debug_set_x3_tile(20/21))
int g_x3_min_lds_kb = 0;   // throttle: pad the workgroup's LDS so fewer fit per CU (mft_debug_set_x3_tile(100 + KB))

template <int BM, int BN, bool AP, bool DB>
int launch_x3(X3Args p, hipStream_t s, int np = 3) {
    const int tiles_m = (p.M + BM - 1) / BM;
    p.tiles_n = p.Cout / BN;
    p.xcd_swizzle = g_x3_xcd;
    p.row_swz = g_x3_row_swz;
    if (np == 2) {                                  // f16x2: two planes per operand, the plain single-buffer kernel
        if constexpr (!AP && !DB) {
            p.row_swz = g_x3_row_swz != 0;
            const size_t lds2 = (size_t)2 * (BM + BN) * X3_RS * sizeof(unsigned short);
            hipLaunchKernelGGL((conv_x3_kernel<BM, BN, false, false, 0, 0, false, 2>), dim3((unsigned)(tiles_m * p.tiles_n)), dim3(256), lds2, s, p);
            return mft_launch_status();
        }
        return MFT_EINVAL;
    }
#ifdef MFT_EXPERIMENTS
    const bool xs = g_x3_row_swz == 2 && !AP && !DB && g_x3_hoist == 0 && g_x3_dbg == 0;
#else
    constexpr bool xs = false;
#endif
    size_t lds = (size_t)(DB ? 2 : 1) * 3 * (BM + BN) * (xs ? 32 : X3_RS) * sizeof(unsigned short);
#ifdef MFT_EXPERIMENTS
    if ((size_t)g_x3_min_lds_kb * 1024 > lds) lds = (size_t)g_x3_min_lds_kb * 1024;
    if (xs) {
        if constexpr (!AP && !DB) {
            hipLaunchKernelGGL((conv_x3_kernel<BM, BN, false, false, 0, 0, true>), dim3((unsigned)(tiles_m * (p.Cout / BN))), dim3(256), lds, s,
                               ([&] { X3Args q = p; q.tiles_n = p.Cout / BN; q.xcd_swizzle = g_x3_xcd; q.row_swz = 0; return q; })());
            return mft_launch_status();
        }
    }
    auto kern = g_x3_hoist == 1 ? conv_x3_kernel<BM, BN, AP, DB, 1> : (g_x3_hoist == 2 ? conv_x3_kernel<BM, BN, AP, DB, 2> : conv_x3_kernel<BM, BN, AP, DB, 0>);
    if (g_x3_dbg && !AP && !DB) {
        switch (g_x3_dbg) {
            case 1: kern = conv_x3_kernel<BM, BN, AP, DB, 0, 1>; break;
            case 2: kern = conv_x3_kernel<BM, BN, AP, DB, 0, 2>; break;
            case 4: kern = conv_x3_kernel<BM, BN, AP, DB, 0, 4>; break;
            case 8: kern = conv_x3_kernel<BM, BN, AP, DB, 0, 8>; break;
            case 12: kern = conv_x3_kernel<BM, BN, AP, DB, 0, 12>; break;
            case 13: kern = conv_x3_kernel<BM, BN, AP, DB, 0, 13>; break;
            case 16: kern = conv_x3_kernel<BM, BN, AP, DB, 0, 16>; break;
            case 28: kern = conv_x3_kernel<BM, BN, AP, DB, 0, 28>; break;
            case 32: kern = conv_x3_kernel<BM, BN, AP, DB, 0, 32>; break;
            case 34: kern = conv_x3_kernel<BM, BN, AP, DB, 0, 34>; break;
            case 36: kern = conv_x3_kernel<BM, BN, AP, DB, 0, 36>; break;
            case 33: kern = conv_x3_kernel<BM, BN, AP, DB, 0, 33>; break;
            default: break;
        }
    }
#else
    auto kern = conv_x3_kernel<BM, BN, AP, DB, 0>;
#endif
    if (lds > 64 * 1024) {                  // opt-in double-buffered / throttled forms only
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)(tiles_m * p.tiles_n)), dim3(256), lds, s, p);
    return mft_launch_status();
}

int g_x3_s1 = 1;           // 3x3 / stride 1 / pad 1 layers: A image staged once per (kh, channel slice) for its three kw taps
                           // (conv_x3_s1_kernel; mft_debug_set_x3_tile(90/91))

constexpr size_t X3_S1_LDS_3PER_CU = 160 * 1024 / 3;       // three workgroups per CU: the occupancy the trunk convolutions are tuned for

int g_x3_bdb = 1;    // f16x2 shared-tap kernels: two weight-tile buffers / one barrier per tap (mft_debug_set_x3_tile(30 | 31); 0 = round 4's form)

inline size_t x3_s1_lds(int BM, int BN, int W, int bn_cin, int np = 3) {
    (void)W;
    return (size_t)np * (BM + 18 + ((np == 2 && g_x3_bdb) ? 2 : 1) * BN) * X3_RS * sizeof(unsigned short) + (size_t)4 * bn_cin * sizeof(float);
}

template <int BM, int BN>
int launch_x3_s1(X3Args p, hipStream_t s, int np = 3) {
    const int tiles_m = (p.M + BM - 1) / BM;
    p.tiles_n = p.Cout / BN;
    p.xcd_swizzle = g_x3_xcd;
    p.row_swz = g_x3_row_swz != 0;
    p.s1_rows = BM + 18;                             // BM pixels + two halo pixels + sixteen zero rows (one per bank phase)
    const size_t lds = x3_s1_lds(BM, BN, p.W, p.bn_ws ? p.Cin : 0, np);
    const dim3 grid((unsigned)(tiles_m * p.tiles_n));
    if (np == 2 && g_x3_bdb) {
        if (p.bn_ws) hipLaunchKernelGGL((conv_x3_s1_bnin_kernel<BM, BN, 2, true>), grid, dim3(256), lds, s, p);
        else hipLaunchKernelGGL((conv_x3_s1_kernel<BM, BN, 2, true>), grid, dim3(256), lds, s, p);
    } else if (np == 2) {
        if (p.bn_ws) hipLaunchKernelGGL((conv_x3_s1_bnin_kernel<BM, BN, 2, false>), grid, dim3(256), lds, s, p);
        else hipLaunchKernelGGL((conv_x3_s1_kernel<BM, BN, 2, false>), grid, dim3(256), lds, s, p);
    } else if (p.bn_ws) {
        hipLaunchKernelGGL((conv_x3_s1_bnin_kernel<BM, BN>), grid, dim3(256), lds, s, p);
    } else {
        hipLaunchKernelGGL((conv_x3_s1_kernel<BM, BN>), grid, dim3(256), lds, s, p);
    }
    return mft_launch_status();
}

int g_x3_pp = 0;           // (MFT_EXPERIMENTS builds) 512-thread ping-pong form of the 128x64 kernel, mft_debug_set_x3_tile(70/71)
#ifdef MFT_EXPERIMENTS
// Measured, one process: 781 us
                           // over the five trunk shapes against 618 us for conv_x3_kernel, and 64.5 vs 76-77 episodes/s in the bench: with one
                           // 8-wave workgroup per CU the stage and multiply halves do overlap inside the workgroup, but the CU then runs 2
                           // waves per SIMD instead of 3 and every barrier stalls all of them -- off.

template <int BM, int BN>
int launch_x3_pp(X3Args p, hipStream_t s) {
    const int tiles_m = (p.M + BM - 1) / BM;
    p.tiles_n = p.Cout / BN;
    p.xcd_swizzle = g_x3_xcd;
    p.row_swz = 1;
    const size_t lds = (size_t)2 * 3 * (BM + BN) * X3_RS * sizeof(unsigned short);       // 92 KB: one workgroup (8 waves) per CU
    static MftPerDeviceOnce attr_once;
    if (attr_once.need()) {
        hipError_t e = hipFuncSetAttribute((const void*)conv_x3_pp_kernel<BM, BN>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return (int)e;
        attr_once.mark();
    }
    hipLaunchKernelGGL((conv_x3_pp_kernel<BM, BN>), dim3((unsigned)(tiles_m * p.tiles_n)), dim3(512), lds, s, p);
    return mft_launch_status();
}

#endif  // MFT_EXPERIMENTS

#ifdef MFT_EXPERIMENTS
#include "../../tools/experiments/conv3x3_patch_x3.inc"
#endif  // MFT_EXPERIMENTS
// Measured (tools/x3_tune.py, tools/phase_times.py, E=128): standalone the patch form wins only on 11x11 maps (157 vs 173 us)
// and loses on 21x21 / 6x6 (tile utilisation 86 / 84 %); beside the last-block stream it is slower everywhere (61.0 / 59.7
// vs 62.0 episodes/s) because both forms are bound by VALU + LDS issue, not by operand re-reads.  Kept as an opt-in.
int g_x3_patch = 0;  // mft_debug_set_x3_tile(10 + v): 0 never use the patch form (default), 1 only when >= 90 % of a tile's
                     // rows are useful (11x11 maps), 2 whenever it applies
int g_x3_tile = 0;   // 0 auto; 1: 128x64; 2: 128x128

}  // namespace

extern "C" int mft_split_bf16x3(const float* w, unsigned short* planes, long long n, void* stream) {
    if (n <= 0) return MFT_EINVAL;
    long long blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(split_bf16x3_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, w, planes, n);
    return mft_launch_status();
}

extern "C" int mft_split_bf16x3_multi(const void* jobs, int n_jobs, long long total_elements, void* stream) {
    if (n_jobs < 1 || total_elements < 1) return MFT_EINVAL;
    long long blocks = (total_elements + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(split_bf16x3_multi_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const SplitJob*)jobs, n_jobs,
                       total_elements);
    return mft_launch_status();
}

extern "C" int mft_debug_set_x3_tile(int t) {
#ifndef MFT_EXPERIMENTS
    // product build: only the knobs that select VALIDATED alternative paths exist (tile shape 0-3, XCD order 20/21, staging-row
    // assignment 40/41, one / two weight-tile buffers 30/31, shared-tap vs per-tap kernel 90/91); the measured-slower experiment kernels are not compiled in
    const bool off = (t == 10 || t == 60 || t == 70 || t == 80 || t == 100 || t == 200);      // "experiment off" codes are no-ops
    if (!off && ((t >= 10 && t < 20) || t == 42 || (t >= 60 && t < 90) || t >= 100)) return MFT_EINVAL;
#endif
    if (t >= 200) g_x3_dbg = t - 200;
    else if (t >= 100) g_x3_min_lds_kb = t - 100;
    else if (t >= 90) g_x3_s1 = t - 90;
    else if (t >= 80) g_x3_hoist = t - 80;
    else if (t >= 70) g_x3_pp = t - 70;
    else if (t >= 60) g_x3_db = t - 60;
    else if (t >= 40) g_x3_row_swz = t - 40;
    else if (t >= 30) g_x3_bdb = t - 30;
    else if (t >= 20) g_x3_xcd = t - 20;
    else if (t >= 10) g_x3_patch = t - 10;
    else g_x3_tile = t;
    return 0;
}

static int x3_dispatch(const float* in, int ldi, const unsigned short* w3, long long plane_elems, float* out, int ldo,
                       int n_img, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad, float* stats_ws,
                       int rows_per_group, void* stream, const unsigned short* in3 = nullptr, long long in_plane_elems = 0,
                       const float* bn_ws = nullptr, const float* bn_gamma = nullptr, const float* bn_beta = nullptr,
                       float bn_eps = 0.f, int np = 3) {
    if (n_img <= 0 || Cin % 32 != 0 || Cout % 64 != 0 || ldi % 4 != 0 || (np != 3 && np != 2)) return MFT_EINVAL;
    if (np == 2 && (in3 != nullptr || g_x3_tile > 1 || g_x3_db || g_x3_pp || g_x3_dbg || g_x3_hoist || g_x3_row_swz == 2 || g_x3_min_lds_kb ||
                    g_x3_patch))
        return MFT_EINVAL;                     // f16x2 exists for the default 128x64 kernels only
    X3Args p;
    p.bn_ws = bn_ws; p.bn_gamma = bn_gamma; p.bn_beta = bn_beta; p.bn_eps = bn_eps;
    p.bn_max_tiles = bn_ws ? mft_x3_max_group_tiles(rows_per_group, 128) : 0;
    if (bn_ws != nullptr && (in3 != nullptr || !bn_gamma || !bn_beta || rows_per_group < 128 || KH != 3 || KW != 3 || stride != 1 ||
                             pad != 1 || x3_s1_lds(128, 64, W, Cin, np) > X3_S1_LDS_3PER_CU ||
                             2 * mft_x3_stage_bytes_host(p.bn_max_tiles, Cin) >
                                 (size_t)np * (128 + 18) * X3_RS * sizeof(unsigned short)))
        return MFT_EINVAL;                     // the loader-side BatchNorm exists in the shared-tap kernel only, at full occupancy
    p.in3 = in3;
    p.in_plane_bytes = 0;
    p.stats_ws = stats_ws;
    p.rows_per_group = rows_per_group;
    p.in = in; p.w3 = w3; p.plane = plane_elems; p.out = out; p.ldi = ldi; p.ldo = ldo;
    p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.KH = KH; p.KW = KW; p.stride = stride; p.pad = pad;
    p.OH = (H + 2 * pad - KH) / stride + 1;
    p.OW = (W + 2 * pad - KW) / stride + 1;
    p.Kpad = KH * KW * Cin;
    const long long M = (long long)n_img * p.OH * p.OW;
    if (M > 0x7fffffffLL) return MFT_EINVAL;
    p.M = (int)M;
    p.tiles_n = 0;
    hipStream_t s = (hipStream_t)stream;
    int G = 0, R = 0;
    double eff = 0.0;
    if (in3 != nullptr) {
        // pre-split input planes: one launch, 32-bit byte offsets over the three planes
        const long long need = (long long)n_img * H * W * ldi;
        if (ldi % 8 != 0 || in_plane_elems < need || 3 * in_plane_elems * 2 >= 0x7fff0000LL || 3 * plane_elems * 2 >= 0x7fff0000LL)
            return MFT_EINVAL;
        if (stats_ws != nullptr && rows_per_group < 128) return MFT_EINVAL;
        p.in_plane_bytes = (unsigned)(in_plane_elems * 2);
        p.in_bytes = 0;
        p.w_bytes = (unsigned)(3 * plane_elems * 2);
        return launch_x3<128, 64, true, false>(p, (hipStream_t)stream);
    }
#ifdef MFT_EXPERIMENTS
    if (stats_ws == nullptr && g_x3_patch && KH == 3 && KW == 3 && stride == 1 && pad == 1 && ldi == Cin && ldo == Cout &&
        patch_geometry(H, W, &G, &R, &eff) && (g_x3_patch >= 2 || eff >= 0.9)) {
        P3Args q;
        q.in = in; q.w3 = w3; q.plane = plane_elems; q.out = out;
        q.n_img = n_img; q.H = H; q.W = W; q.Cin = Cin; q.Cout = Cout; q.G = G; q.R = R;
        q.row_blocks = (H + R - 1) / R;
        q.tiles_n = Cout / 64;
        const int img_groups = (n_img + G - 1) / G;
        const size_t lds = (size_t)3 * G * (R + 2) * (W + 2) * X3_RS * sizeof(unsigned short);
        static MftPerDeviceOnce attr_once;
        if (attr_once.need()) {
            hipError_t e = hipFuncSetAttribute((const void*)conv3x3_patch_x3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                               64 * 1024);
            if (e != hipSuccess) return (int)e;
            attr_once.mark();
        }
        if (lds <= 64 * 1024) {
            hipLaunchKernelGGL(conv3x3_patch_x3_kernel, dim3((unsigned)(img_groups * q.row_blocks * q.tiles_n)), dim3(256), lds, s, q);
            return mft_launch_status();
        }
    }
#else
    (void)G; (void)R; (void)eff;
#endif
    int tile = g_x3_tile;
    if (tile == 0) tile = 1;       // 128x64 beats 128x128 on every trunk shape (3 vs 2 workgroups per CU)
    p.w_bytes = (unsigned)(np * plane_elems * 2);
    // buffer extents must stay below 2^31 bytes: split the batch over images when the input is larger
    const long long img_bytes = (long long)H * W * ldi * 4;
    const long long max_imgs = 0x7fff0000LL / img_bytes;
    if (max_imgs < 1 || 3 * plane_elems * 2 >= 0x7fff0000LL) return MFT_EINVAL;
    if ((stats_ws != nullptr || bn_ws != nullptr) && (max_imgs < n_img || rows_per_group < 128)) return MFT_EINVAL;   // tile numbering needs one launch
    if (stats_ws != nullptr && tile != 3) tile = 1;
    if (stats_ws != nullptr && tile == 3 && rows_per_group < 64) return MFT_EINVAL;
    for (long long i0 = 0; i0 < n_img; i0 += max_imgs) {
        const long long ni = (n_img - i0 < max_imgs) ? (n_img - i0) : max_imgs;
        X3Args q = p;
        q.in = in + i0 * H * W * ldi;
        q.out = out + i0 * p.OH * p.OW * ldo;
        q.M = (int)(ni * p.OH * p.OW);
        q.in_bytes = (unsigned)(ni * img_bytes);
        const bool s1 = g_x3_s1 && tile == 1 && KH == 3 && KW == 3 && stride == 1 && pad == 1 && !g_x3_db && !g_x3_pp && !g_x3_dbg &&
                        g_x3_hoist == 0 && g_x3_row_swz != 2 && g_x3_min_lds_kb == 0 &&
                        x3_s1_lds(128, 64, W, 0, np) <= 64 * 1024;
        if (bn_ws != nullptr && !s1) return MFT_EINVAL;
        const int rc = s1 ? launch_x3_s1<128, 64>(q, s, np)
                       : np == 2 ? launch_x3<128, 64, false, false>(q, s, 2)
                       : (tile == 2 && Cout % 128 == 0) ? launch_x3<128, 128, false, false>(q, s)
                       : (tile == 3)                  ? launch_x3<64, 64, false, false>(q, s)
#ifdef MFT_EXPERIMENTS
                       : g_x3_db                      ? launch_x3<128, 64, false, true>(q, s)
                       : (g_x3_pp && !g_x3_dbg)       ? launch_x3_pp<128, 64>(q, s)
#endif
                                                      : launch_x3<128, 64, false, false>(q, s);
        if (rc != 0) return rc;
    }
    return 0;
}

extern "C" int mft_conv2d_nhwc_x3(const float* in, int ldi, const unsigned short* w3, long long plane_elems, float* out,
                                  int ldo, int n_img, int H, int W, int Cin, int Cout, int KH, int KW, int stride,
                                  int pad, void* stream) {
    return x3_dispatch(in, ldi, w3, plane_elems, out, ldo, n_img, H, W, Cin, Cout, KH, KW, stride, pad, nullptr, 0, stream);
}

extern "C" int mft_split_f16x2(const float* w, unsigned short* planes, long long n, void* stream) {
    if (n <= 0) return MFT_EINVAL;
    long long blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(split_f16x2_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, w, planes, n);
    return mft_launch_status();
}

extern "C" int mft_conv2d_nhwc_h2(const float* in, int ldi, const unsigned short* w2, long long plane_elems, float* out,
                                  int ldo, int n_img, int H, int W, int Cin, int Cout, int KH, int KW, int stride,
                                  int pad, void* stream) {
    return x3_dispatch(in, ldi, w2, plane_elems, out, ldo, n_img, H, W, Cin, Cout, KH, KW, stride, pad, nullptr, 0, stream, nullptr, 0,
                       nullptr, nullptr, nullptr, 0.f, 2);
}

extern "C" long long mft_conv2d_x3_stats_ws_floats(int n_img, int H, int W, int Cout, int KH, int KW, int stride, int pad) {
    const long long OH = (H + 2 * pad - KH) / stride + 1, OW = (W + 2 * pad - KW) / stride + 1;
    return ((long long)n_img * OH * OW + 63) / 64 * 2 * Cout * 2;          // sized for the smallest tile (64 rows)
}

static int x3_bnstats(int np, const float* in, int ldi, const unsigned short* w3, long long plane_elems,
                      float* out, int ldo, int n_img, int H, int W, int Cin, int Cout, int KH, int KW,
                      int stride, int pad, int imgs_per_group, float eps, float* stats_ws,
                      float* mean, float* rstd, void* stream) {
    if (imgs_per_group <= 0 || n_img % imgs_per_group != 0 || stats_ws == nullptr) return MFT_EINVAL;
    const int OH = (H + 2 * pad - KH) / stride + 1, OW = (W + 2 * pad - KW) / stride + 1;
    const int R = imgs_per_group * OH * OW;
    const int rc = x3_dispatch(in, ldi, w3, plane_elems, out, ldo, n_img, H, W, Cin, Cout, KH, KW, stride, pad, stats_ws, R,
                               stream, nullptr, 0, nullptr, nullptr, nullptr, 0.f, np);
    if (rc != 0) return rc;
    if (mean == nullptr && rstd == nullptr) return g_x3_tile == 3 ? MFT_EINVAL : 0;     // partials only: the consumer merges them (128-row tiles)
    const int groups = n_img / imgs_per_group;
    hipLaunchKernelGGL(x3_stats_finalize_kernel, dim3((Cout + 63) / 64, groups), dim3(64), 0, (hipStream_t)stream,
                       (const float*)stats_ws, Cout, n_img * OH * OW, R, g_x3_tile == 3 ? 64 : 128, eps, mean, rstd);
    return mft_launch_status();
}

extern "C" int mft_conv2d_nhwc_x3_bnstats(const float* in, int ldi, const unsigned short* w3, long long plane_elems,
                                          float* out, int ldo, int n_img, int H, int W, int Cin, int Cout, int KH, int KW,
                                          int stride, int pad, int imgs_per_group, float eps, float* stats_ws,
                                          float* mean, float* rstd, void* stream) {
    return x3_bnstats(3, in, ldi, w3, plane_elems, out, ldo, n_img, H, W, Cin, Cout, KH, KW, stride, pad, imgs_per_group, eps, stats_ws, mean,
                      rstd, stream);
}

extern "C" int mft_conv2d_nhwc_h2_bnstats(const float* in, int ldi, const unsigned short* w2, long long plane_elems,
                                          float* out, int ldo, int n_img, int H, int W, int Cin, int Cout, int KH, int KW,
                                          int stride, int pad, int imgs_per_group, float eps, float* stats_ws,
                                          float* mean, float* rstd, void* stream) {
    return x3_bnstats(2, in, ldi, w2, plane_elems, out, ldo, n_img, H, W, Cin, Cout, KH, KW, stride, pad, imgs_per_group, eps, stats_ws, mean,
                      rstd, stream);
}

// 3x3 / stride 1 / pad 1 convolution of relu(BatchNorm(in)) where ``in`` is the raw output of the previous bf16x3 convolution and
// ``in_ws`` its statistics partials (SimpleBlock's C1 -> BN1 -> ReLU -> C2, backbone.py:251-256, in one launch + this
// convolution's own partials).  MFT_EINVAL outside the shared-tap kernel's domain (the caller then runs apply + convolution).
static int x3_bnin_bnstats(int np, const float* in, int ldi, const float* in_ws, const float* in_gamma,
                           const float* in_beta, const unsigned short* w3, long long plane_elems, float* out,
                           int ldo, int n_img, int H, int W, int Cin, int Cout, int imgs_per_group, float eps,
                           float* stats_ws, float* mean, float* rstd, void* stream) {
    if (imgs_per_group <= 0 || n_img % imgs_per_group != 0 || stats_ws == nullptr || in_ws == nullptr ||
        (mean == nullptr) != (rstd == nullptr) || g_x3_tile == 3)
        return MFT_EINVAL;
    const int R = imgs_per_group * H * W;
    const int rc = x3_dispatch(in, ldi, w3, plane_elems, out, ldo, n_img, H, W, Cin, Cout, 3, 3, 1, 1, stats_ws, R, stream, nullptr, 0,
                               in_ws, in_gamma, in_beta, eps, np);
    if (rc != 0 || mean == nullptr) return rc;
    hipLaunchKernelGGL(x3_stats_finalize_kernel, dim3((Cout + 63) / 64, n_img / imgs_per_group), dim3(64), 0, (hipStream_t)stream,
                       (const float*)stats_ws, Cout, n_img * H * W, R, 128, eps, mean, rstd);
    return mft_launch_status();
}

extern "C" int mft_conv2d_nhwc_x3_bnin_bnstats(const float* in, int ldi, const float* in_ws, const float* in_gamma,
                                               const float* in_beta, const unsigned short* w3, long long plane_elems, float* out,
                                               int ldo, int n_img, int H, int W, int Cin, int Cout, int imgs_per_group, float eps,
                                               float* stats_ws, float* mean, float* rstd, void* stream) {
    return x3_bnin_bnstats(3, in, ldi, in_ws, in_gamma, in_beta, w3, plane_elems, out, ldo, n_img, H, W, Cin, Cout, imgs_per_group, eps,
                           stats_ws, mean, rstd, stream);
}

extern "C" int mft_conv2d_nhwc_h2_bnin_bnstats(const float* in, int ldi, const float* in_ws, const float* in_gamma,
                                               const float* in_beta, const unsigned short* w2, long long plane_elems, float* out,
                                               int ldo, int n_img, int H, int W, int Cin, int Cout, int imgs_per_group, float eps,
                                               float* stats_ws, float* mean, float* rstd, void* stream) {
    return x3_bnin_bnstats(2, in, ldi, in_ws, in_gamma, in_beta, w2, plane_elems, out, ldo, n_img, H, W, Cin, Cout, imgs_per_group, eps,
                           stats_ws, mean, rstd, stream);
}

extern "C" int mft_conv2d_nhwc_x3p_bnstats(const unsigned short* in_planes, long long in_plane_elems, int ldi,
                                           const unsigned short* w3, long long plane_elems, float* out, int ldo, int n_img,
                                           int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad,
                                           int imgs_per_group, float eps, float* stats_ws, float* mean, float* rstd,
                                           void* stream) {
    if (imgs_per_group <= 0 || n_img % imgs_per_group != 0 || stats_ws == nullptr || in_planes == nullptr) return MFT_EINVAL;
    const int OH = (H + 2 * pad - KH) / stride + 1, OW = (W + 2 * pad - KW) / stride + 1;
    const int R = imgs_per_group * OH * OW;
    const int rc = x3_dispatch(nullptr, ldi, w3, plane_elems, out, ldo, n_img, H, W, Cin, Cout, KH, KW, stride, pad, stats_ws, R,
                               stream, in_planes, in_plane_elems);
    if (rc != 0) return rc;
    const int groups = n_img / imgs_per_group;
    hipLaunchKernelGGL(x3_stats_finalize_kernel, dim3((Cout + 63) / 64, groups), dim3(64), 0, (hipStream_t)stream,
                       (const float*)stats_ws, Cout, n_img * OH * OW, R, 128, eps, mean, rstd);
    return mft_launch_status();
}
