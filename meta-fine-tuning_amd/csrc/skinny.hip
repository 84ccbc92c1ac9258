// Weight-streaming "skinny" convolutions of the adapted last block (trunk.7, backbone.py:216-261 called from
// finetune.py:286 / gnnnet.py:168, and its data gradient in loss.backward(), finetune.py:293).
//
// In the episode-batched inner loop every episode owns its own trunk.7 weights (14.7 MB) but contributes only
// 5 images = 45 output pixels: each weight element is used for 45 rows and never again in that step.  The generic
// implicit-GEMM kernel stages weights through LDS tile by tile with one K-step in flight and reaches ~3 TB/s.
// Here the roles are swapped:
//   * the episode's ACTIVATIONS (<= 96 KB per channel slice) are staged once in LDS, rows padded by 8 floats so that
//     the 16-byte fragment reads of 16 different pixels are conflict-free;
//   * the WEIGHTS go straight from HBM into MFMA operand registers: with v_mfma_f32_16x16x4_f32 the weight matrix is
//     the "A" operand (16 output channels x 4 k per instruction), lane (m, kq) loads 16 contiguous bytes of its
//     channel's row and uses one element per MFMA step; 8 such loads per lane are in flight ahead of the MFMAs;
//   * the 45 pixels are the "B"/N side: 3 blocks of 16 (M padding 48 instead of 64), fp32 MFMA throughout.
// One 1024-thread workgroup (16 waves) handles one episode x 256 output channels, so a launch over 128 episodes is
// one workgroup per CU with 64+ KB of weight loads in flight per CU.
#include "mft_common.h"

namespace {

typedef float f32x4v __attribute__((ext_vector_type(4)));

struct SkinnyArgs {
    const float* act;      // [groups][rows_in][lda]
    const float* w;        // [groups][Cout][K]
    float* out;            // [groups][rows_out][ldo]
    int lda, ldo;
    int H, W, Cin, OH, OW, Cout, KH, KW, stride, pad;
    int ipg, rows_in, rows_out;
    long long wgs;
    int K;
    int CS;                // channel slice staged in LDS (multiple of 128, divides Cin)
};

constexpr int SK_PADF = 8;     // row padding (floats)
constexpr int SK_U = 8;        // k16 groups per chunk (loads in flight per lane)

// forward: out[ro][co] = sum_{tap,ci} act[pix(ro,tap)][ci] * w[co][tap][ci]
__global__ __launch_bounds__(1024) void skinny_conv_fwd_kernel(SkinnyArgs p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int m = lane & 15, kq = lane >> 4;
    const int g = blockIdx.y;
    const int co0 = blockIdx.x * 256 + wave * 16;
    const int RS = p.CS + SK_PADF;
    const int ohw = p.OH * p.OW;
    const int gpt = p.CS / 16;                    // k16 groups per tap within a slice
    const int taps = p.KH * p.KW;

    // the three pixel blocks of this lane's N index
    int n_img[3], n_ih0[3], n_iw0[3];
    bool n_ok[3];
#pragma unroll
    for (int nb = 0; nb < 3; ++nb) {
        const int ro = nb * 16 + m;
        n_ok[nb] = ro < p.rows_out;
        const int rr = n_ok[nb] ? ro : 0;
        const int img = rr / ohw;
        const int rem = rr - img * ohw;
        const int oh = rem / p.OW, ow = rem - oh * p.OW;
        n_img[nb] = img;
        n_ih0[nb] = oh * p.stride - p.pad;
        n_iw0[nb] = ow * p.stride - p.pad;
    }
    const float* wrow = p.w + (long long)g * p.wgs + (long long)(co0 + m) * p.K + 4 * kq;
    const float* actg = p.act + (long long)g * p.rows_in * p.lda;

    f32x4v acc[3];
#pragma unroll
    for (int nb = 0; nb < 3; ++nb) acc[nb] = f32x4v{0.f, 0.f, 0.f, 0.f};

    const int n_slices = p.Cin / p.CS;
    for (int sl = 0; sl < n_slices; ++sl) {
        __syncthreads();                           // previous slice fully consumed
        // stage act[:, sl*CS : (sl+1)*CS] -> lds[rows_in][RS]
        const int q4 = p.CS / 4;
        for (int i = tid; i < p.rows_in * q4; i += 1024) {
            const int r = i / q4, c = (i - r * q4) * 4;
            *(f32x4v*)(lds + r * RS + c) = *(const f32x4v*)(actg + (long long)r * p.lda + sl * p.CS + c);
        }
        __syncthreads();
        const int n_groups = taps * gpt;           // k16 groups of this slice; chunks never straddle a tap (gpt % SK_U == 0)
        f32x4v a_cur[SK_U], a_nxt[SK_U];
        auto load_chunk = [&](int q0, f32x4v* dst) {
            const int tap = q0 / gpt;
            const int cg0 = q0 - tap * gpt;
            const float* src = wrow + tap * p.Cin + sl * p.CS + cg0 * 16;
#pragma unroll
            for (int u = 0; u < SK_U; ++u) dst[u] = __builtin_nontemporal_load((const f32x4v*)(src + u * 16));
        };
        load_chunk(0, a_cur);
        for (int q0 = 0; q0 < n_groups; q0 += SK_U) {
            if (q0 + SK_U < n_groups) load_chunk(q0 + SK_U, a_nxt);
            const int tap = q0 / gpt;
            const int cg0 = q0 - tap * gpt;
            const int kh = tap / p.KW, kw = tap - kh * p.KW;
            int boff[3];
#pragma unroll
            for (int nb = 0; nb < 3; ++nb) {
                const int ih = n_ih0[nb] + kh, iw = n_iw0[nb] + kw;
                const bool ok = n_ok[nb] && ih >= 0 && ih < p.H && iw >= 0 && iw < p.W;
                boff[nb] = ok ? ((n_img[nb] * p.H + ih) * p.W + iw) * RS + 4 * kq + cg0 * 16 : -1;
            }
#pragma unroll
            for (int u = 0; u < SK_U; ++u) {
                f32x4v b4[3];
#pragma unroll
                for (int nb = 0; nb < 3; ++nb) {
                    b4[nb] = f32x4v{0.f, 0.f, 0.f, 0.f};
                    if (boff[nb] >= 0) b4[nb] = *(const f32x4v*)(lds + boff[nb] + u * 16);
                }
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int nb = 0; nb < 3; ++nb)
                        acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[u][t], b4[nb][t], acc[nb], 0, 0, 0);
            }
#pragma unroll
            for (int u = 0; u < SK_U; ++u) a_cur[u] = a_nxt[u];
        }
    }
    // D layout of the 16x16 MFMA: lane holds D[i = 4*(lane/16) + e][j = lane%16]; i = output channel, j = pixel
    float* outg = p.out + (long long)g * p.rows_out * p.ldo;
#pragma unroll
    for (int nb = 0; nb < 3; ++nb) {
        const int ro = nb * 16 + m;
        if (ro < p.rows_out) *(f32x4v*)(outg + (long long)ro * p.ldo + co0 + 4 * kq) = acc[nb];
    }
}

// ---- bf16x3 forms ----------------------------------------------------------------------------------------------------------
// At 45 pixels per episode the fp32 MFMA work of these layers (2 x 48 x 512 x 4608 FLOP per episode, 157 TFLOP/s peak) takes
// about as long as streaming the weights (9.4 MB per episode), so the fp32 kernels above are co-limited by the matrix pipe and
// HBM and stop at ~4 TB/s.  The x3 forms do the same product with six bf16 MFMAs per fp32-equivalent (csrc/conv_x3.hip: every
// fp32 value is the exact sum of three bf16 pieces; the three lowest-order cross terms are dropped; error at the level of fp32
// rounding) at 16/6 of the fp32 rate: weights are split in registers right after their load, the activations are split ONCE
// while they are staged into three bf16 LDS planes.  K order inside a 32-element group is permuted so that a lane's eight
// operand elements are the two float4 it already loads (k = 16*uu + 4*kq + t -> slot 8*kq + 4*uu + t).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));
typedef float f32x2s __attribute__((ext_vector_type(2)));
typedef unsigned u32x2v __attribute__((ext_vector_type(2)));
typedef unsigned u32x4v __attribute__((ext_vector_type(4)));

constexpr int SK_PADH = 16;    // row padding of the bf16 planes (elements): row stride = 32 B mod 256 B like the fp32 tiles

__device__ __forceinline__ unsigned sk_pk_bf16(float a, float b) {
    f32x2s v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2v));
}

// 4 floats -> three planes of 4 bf16 (round-to-nearest-even pieces, exact residuals); same split as csrc/conv_x3.hip
__device__ __forceinline__ void sk_split4(const f32x4v x, u32x2v& p1, u32x2v& p2, u32x2v& p3) {
    f32x4v r;
    p1[0] = sk_pk_bf16(x[0], x[1]);
    p1[1] = sk_pk_bf16(x[2], x[3]);
    r[0] = x[0] - __builtin_bit_cast(float, p1[0] << 16);
    r[1] = x[1] - __builtin_bit_cast(float, p1[0] & 0xffff0000u);
    r[2] = x[2] - __builtin_bit_cast(float, p1[1] << 16);
    r[3] = x[3] - __builtin_bit_cast(float, p1[1] & 0xffff0000u);
    p2[0] = sk_pk_bf16(r[0], r[1]);
    p2[1] = sk_pk_bf16(r[2], r[3]);
    r[0] -= __builtin_bit_cast(float, p2[0] << 16);
    r[1] -= __builtin_bit_cast(float, p2[0] & 0xffff0000u);
    r[2] -= __builtin_bit_cast(float, p2[1] << 16);
    r[3] -= __builtin_bit_cast(float, p2[1] & 0xffff0000u);
    p3[0] = sk_pk_bf16(r[0], r[1]);
    p3[1] = sk_pk_bf16(r[2], r[3]);
}

// slot of channel c (multiple of 4) inside its row of a permuted bf16 plane
__device__ __forceinline__ int sk_perm(int c) {
    const int kl = c & 31;
    return (c & ~31) + ((kl & 15) >> 2) * 8 + (kl >> 4) * 4;
}

// split one staged float4 and store its pieces into the three planes (plane stride PL elements)
__device__ __forceinline__ void sk_store3(unsigned short* L, int off, int PL, const f32x4v v) {
    u32x2v p1, p2, p3;
    sk_split4(v, p1, p2, p3);
    *(u32x2v*)(L + off) = p1;
    *(u32x2v*)(L + PL + off) = p2;
    *(u32x2v*)(L + 2 * PL + off) = p3;
}

// two weight float4 (k16 groups u, u+1 of one lane) -> the three bf16x8 A operands
__device__ __forceinline__ void sk_split_a(const f32x4v w0, const f32x4v w1, bf16x8 a[3]) {
    u32x2v p0[3], p1[3];
    sk_split4(w0, p0[0], p0[1], p0[2]);
    sk_split4(w1, p1[0], p1[1], p1[2]);
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
        const u32x4v q = {p0[pl][0], p0[pl][1], p1[pl][0], p1[pl][1]};
        a[pl] = __builtin_bit_cast(bf16x8, q);
    }
}

// Full-line weight loads.  In MFMA fragment order lane (m, kq) reads 16 B of row m, so one wave-instruction touches 16 rows x 64 B:
// HALF cache lines, whose other halves follow in the next instruction.  A read-only stream of that shape tops out at 5.5 TB/s where
// whole 128-B lines per instruction reach 6.95 (tools/microbench/read_pattern.hip; the 64-B run is what costs, not the lane order).
// With LN the two loads of a k32 group are re-dealt: the first reads rows 0-7 of the wave's 16 (lanes m < 8: floats 4 kq of the
// group, lanes m >= 8: floats 16 + 4 kq of row m - 8), the second rows 8-15 the same way -- 8 rows x 128 B each -- and two DPP
// row rotations by 8 lanes hand every lane the pieces the fragment order wants (8 v_mov_dpp per 32 B of weights per lane).
// Same registers, same values, same arithmetic as the fragment-order loads: results are bit-identical.
template <bool LN>
__device__ __forceinline__ const float* sk_wbase(const float* w_co0, int m, int kq, int K) {
    if constexpr (LN) return w_co0 + (long long)(m & 7) * K + 4 * kq + 16 * (m >> 3);
    else return w_co0 + (long long)m * K + 4 * kq;
}
// pointer of load number qq (units of 16 floats; pairs (even, odd) belong to one k32 group) relative to a lane base from sk_wbase
template <bool LN>
__device__ __forceinline__ const float* sk_wptr(const float* base, int qq, int K) {
    if constexpr (LN) return base + (long long)(qq & ~1) * 16 + ((qq & 1) ? (long long)8 * K : 0);
    else return base + (long long)qq * 16;
}
template <bool LN>
__device__ __forceinline__ void sk_wfix(f32x4v& w0, f32x4v& w1) {
    if constexpr (LN) {
        f32x4v f0, f1;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            // (element copies first: __builtin_bit_cast applied to a vector ELEMENT expression reads element 0 with this compiler)
            const float ta = w0[e], tb = w1[e];
            const int a = __builtin_bit_cast(int, ta), b = __builtin_bit_cast(int, tb);
            // row_ror:8 = 0x128; bank_mask 0xc: lanes 8-15 of every row of 16 take the rotated source, 0x3: lanes 0-7
            f0[e] = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(a, b, 0x128, 0xf, 0xc, false));
            f1[e] = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(b, a, 0x128, 0xf, 0x3, false));
        }
        w0 = f0; w1 = f1;
    }
}

// six piece products, smallest first (same order as conv_x3_kernel); weights are the A side
#define SK_X3_MFMA(ACC, A, B)                                                              \
    {                                                                                      \
        constexpr int TA_[6] = {2, 0, 1, 1, 0, 0};                                         \
        constexpr int TB_[6] = {0, 2, 1, 0, 1, 0};                                         \
        _Pragma("unroll") for (int t_ = 0; t_ < 6; ++t_)                                   \
            _Pragma("unroll") for (int nb_ = 0; nb_ < 3; ++nb_)                            \
                ACC[nb_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[TA_[t_]], B[nb_][TB_[t_]], ACC[nb_], 0, 0, 0); \
    }

// sum over the 16 lanes that share a kq (the 16 pixels of one MFMA column block)
__device__ __forceinline__ float sk_row16_sum(float v) {
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// mean / rstd of 4 channels over the episode's valid pixels held as v[nb] (pixel nb*16 + m) by the 16 lanes of a kq group
__device__ __forceinline__ void sk_stats(const f32x4v v[3], int m, int rows, float eps, f32x4v& mu, f32x4v& rs) {
    const float inv = 1.f / (float)rows;
    f32x4v s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int nb = 0; nb < 3; ++nb)
        if (nb * 16 + m < rows) s += v[nb];
#pragma unroll
    for (int e = 0; e < 4; ++e) mu[e] = sk_row16_sum(s[e]) * inv;
    s = f32x4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int nb = 0; nb < 3; ++nb)
        if (nb * 16 + m < rows) {
            const f32x4v d = v[nb] - mu;
            s += d * d;
        }
#pragma unroll
    for (int e = 0; e < 4; ++e) rs[e] = 1.0f / sqrtf(sk_row16_sum(s[e]) * inv + eps);
}

// EXIT: second half of the residual block in the same launch (SimpleBlock.forward, backbone.py:256-261, + the AvgPool2d of
// ResNet.forward): out = ReLU(BN2(c2) + BNshortcut(sc)) with per-episode statistics of both branches, and the global average
// pool of `out` per image -- the wave already holds all pixels of its 16 channels.
struct SkinnyExitArgs {
    const float* sc;           // raw shortcut conv output [groups][rows_out][ldo]
    float* y;                  // block output [groups][rows_out][ldo]
    float* pooled;             // [groups][ipg][Cout]
    const float* g2; const float* b2; const float* gs; const float* bs; long long gbs;
    float* mean2; float* rstd2; float* means; float* rstds;
    float eps;
    int hw;                    // pixels per image
};

// forward, whole activation in LDS (trunk.7.C2: 45 input pixels x 512 channels = 3 x 46 KB of bf16 planes)
// NW = waves per workgroup (16 output channels each): 16 at E >= 128; fewer, so that a small episode batch still fills the CUs
template <bool EXIT, int NW, bool LN>
__global__ __launch_bounds__(64 * NW) void skinny_conv_fwd_x3_kernel(SkinnyArgs p, SkinnyExitArgs x) {
    extern __shared__ __attribute__((aligned(16))) unsigned short ldh[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int m = lane & 15, kq = lane >> 4;
    const int g = blockIdx.y;
    const int co0 = blockIdx.x * (16 * NW) + wave * 16;
    const int RS = p.Cin + SK_PADH;
    const int PL = p.rows_in * RS;
    const int ohw = p.OH * p.OW;
    const int gpt = p.Cin / 16;
    const int taps = p.KH * p.KW;

    int n_img[3], n_ih0[3], n_iw0[3];
    bool n_ok[3];
#pragma unroll
    for (int nb = 0; nb < 3; ++nb) {
        const int ro = nb * 16 + m;
        n_ok[nb] = ro < p.rows_out;
        const int rr = n_ok[nb] ? ro : 0;
        const int img = rr / ohw;
        const int rem = rr - img * ohw;
        const int oh = rem / p.OW, ow = rem - oh * p.OW;
        n_img[nb] = img;
        n_ih0[nb] = oh * p.stride - p.pad;
        n_iw0[nb] = ow * p.stride - p.pad;
    }
    const float* wrow = sk_wbase<LN>(p.w + (long long)g * p.wgs + (long long)co0 * p.K, m, kq, p.K);
    const float* actg = p.act + (long long)g * p.rows_in * p.lda;

    // weights of one output channel are contiguous over (tap, ci); SK_U float4 per lane stay in flight: a register pair is
    // re-issued for the next chunk as soon as its pieces have been split off
    f32x4v a_cur[SK_U];
#pragma unroll
    for (int u = 0; u < SK_U; ++u) a_cur[u] = __builtin_nontemporal_load((const f32x4v*)sk_wptr<LN>(wrow, u, p.K));
    {
        const int q4 = p.Cin / 4;
        for (int i = tid; i < p.rows_in * q4; i += 64 * NW) {
            const int r = i / q4, c = (i - r * q4) * 4;
            sk_store3(ldh, r * RS + sk_perm(c), PL, *(const f32x4v*)(actg + (long long)r * p.lda + c));
        }
    }
    __syncthreads();

    f32x4v acc[3];
#pragma unroll
    for (int nb = 0; nb < 3; ++nb) acc[nb] = f32x4v{0.f, 0.f, 0.f, 0.f};
    const int n_groups = taps * gpt;               // chunks never straddle a tap (gpt % SK_U == 0)
    for (int q0 = 0; q0 < n_groups; q0 += SK_U) {
        const bool more = q0 + SK_U < n_groups;
        const int tap = q0 / gpt;
        const int cg0 = q0 - tap * gpt;
        const int kh = tap / p.KW, kw = tap - kh * p.KW;
        int boff[3];
#pragma unroll
        for (int nb = 0; nb < 3; ++nb) {
            const int ih = n_ih0[nb] + kh, iw = n_iw0[nb] + kw;
            const bool ok = n_ok[nb] && ih >= 0 && ih < p.H && iw >= 0 && iw < p.W;
            boff[nb] = ok ? ((n_img[nb] * p.H + ih) * p.W + iw) * RS + 8 * kq + cg0 * 16 : -1;
        }
#pragma unroll
        for (int u = 0; u < SK_U; u += 2) {
            bf16x8 a[3], b[3][3];
            sk_wfix<LN>(a_cur[u], a_cur[u + 1]);
            sk_split_a(a_cur[u], a_cur[u + 1], a);
            if (more) {
                a_cur[u] = __builtin_nontemporal_load((const f32x4v*)sk_wptr<LN>(wrow, q0 + SK_U + u, p.K));
                a_cur[u + 1] = __builtin_nontemporal_load((const f32x4v*)sk_wptr<LN>(wrow, q0 + SK_U + u + 1, p.K));
            }
#pragma unroll
            for (int nb = 0; nb < 3; ++nb)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) {
                    u32x4v z = {0u, 0u, 0u, 0u};
                    if (boff[nb] >= 0) z = *(const u32x4v*)(ldh + pl * PL + boff[nb] + u * 16);
                    b[nb][pl] = __builtin_bit_cast(bf16x8, z);
                }
            SK_X3_MFMA(acc, a, b)
        }
    }
    float* outg = p.out + (long long)g * p.rows_out * p.ldo;
#pragma unroll
    for (int nb = 0; nb < 3; ++nb) {
        const int ro = nb * 16 + m;
        if (ro < p.rows_out) *(f32x4v*)(outg + (long long)ro * p.ldo + co0 + 4 * kq) = acc[nb];
    }
    if constexpr (EXIT) {
        const int co = co0 + 4 * kq;
        const long long ob = (long long)g * p.rows_out * p.ldo;
        f32x4v sv[3];
#pragma unroll
        for (int nb = 0; nb < 3; ++nb) {
            const int ro = nb * 16 + m;
            sv[nb] = f32x4v{0.f, 0.f, 0.f, 0.f};
            if (ro < p.rows_out) sv[nb] = *(const f32x4v*)(x.sc + ob + (long long)ro * p.ldo + co);
        }
        f32x4v m2, r2, ms, rs;
        sk_stats(acc, m, p.rows_out, x.eps, m2, r2);
        sk_stats(sv, m, p.rows_out, x.eps, ms, rs);
        if (m == 0) {
            *(f32x4v*)(x.mean2 + (long long)g * p.Cout + co) = m2;
            *(f32x4v*)(x.rstd2 + (long long)g * p.Cout + co) = r2;
            *(f32x4v*)(x.means + (long long)g * p.Cout + co) = ms;
            *(f32x4v*)(x.rstds + (long long)g * p.Cout + co) = rs;
        }
        const f32x4v ga2 = *(const f32x4v*)(x.g2 + g * x.gbs + co), be2 = *(const f32x4v*)(x.b2 + g * x.gbs + co);
        const f32x4v gas = *(const f32x4v*)(x.gs + g * x.gbs + co), bes = *(const f32x4v*)(x.bs + g * x.gbs + co);
        __syncthreads();                           // every wave is done with the activation planes: reuse them for the pool
        float* T = (float*)ldh + wave * 48 * 16;   // this wave's [48 pixels][16 channels] output tile
#pragma unroll
        for (int nb = 0; nb < 3; ++nb) {
            const int ro = nb * 16 + m;
            if (ro >= p.rows_out) continue;
            f32x4v o = (acc[nb] - m2) * r2 * ga2 + be2;
            o += (sv[nb] - ms) * rs * gas + bes;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = fmaxf(o[e], 0.f);
            *(f32x4v*)(x.y + ob + (long long)ro * p.ldo + co) = o;
            *(f32x4v*)(T + ro * 16 + 4 * kq) = o;
        }
        __syncthreads();
        const int n_img = p.rows_out / x.hw;
        const float invp = 1.f / (float)x.hw;
        for (int i0 = 0; i0 < n_img; i0 += 4) {
            const int img = i0 + (lane >> 4);
            if (img < n_img) {
                float sum = 0.f;
                for (int k2 = 0; k2 < x.hw; ++k2) sum += T[(img * x.hw + k2) * 16 + m];
                x.pooled[((long long)g * n_img + img) * p.Cout + co0 + m] = sum * invp;
            }
        }
    }
}

// forward, per-tap staging: for strided / wide-input layers the episode's whole activation does not fit LDS (trunk.7.C1:
// 5 x 6 x 6 pixels x 256 channels = 184 KB), but one TAP's im2col rows do (48 x Cin floats): they are gathered into a dense
// [48][Cin + 8] LDS tile (zero rows where the tap falls into the padding), double buffered so the gather of tap t+1
// overlaps the MFMAs of tap t.  The weight stream is then fully sequential per output channel (K order = (tap, ci)).
__global__ __launch_bounds__(1024) void skinny_conv_fwd_tap_kernel(SkinnyArgs p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int m = lane & 15, kq = lane >> 4;
    const int g = blockIdx.y;
    const int co0 = blockIdx.x * 256 + wave * 16;
    const int RS = p.Cin + SK_PADF;
    const int ohw = p.OH * p.OW;
    const int taps = p.KH * p.KW;
    const int gpt = p.Cin / 16;
    const int q4 = p.Cin / 4;                      // float4 per staged row
    constexpr int NST = 3;                         // float4 slots per thread: 48 rows x (Cin <= 256)/4 <= 3072
    const float* wrow = p.w + (long long)g * p.wgs + (long long)(co0 + m) * p.K + 4 * kq;
    const float* actg = p.act + (long long)g * p.rows_in * p.lda;

    // staging descriptors: slot k of this thread covers row sr[k], channels sc[k]..+3
    int s_row[NST], s_c[NST], s_img[NST], s_ih0[NST], s_iw0[NST];
    bool s_use[NST], s_rowok[NST];
#pragma unroll
    for (int k = 0; k < NST; ++k) {
        const int i = tid + k * 1024;
        s_use[k] = i < 48 * q4;
        const int r = s_use[k] ? i / q4 : 0;
        s_row[k] = r;
        s_c[k] = (i - r * q4) * 4;
        s_rowok[k] = s_use[k] && r < p.rows_out;
        const int rr = s_rowok[k] ? r : 0;
        const int img = rr / ohw;
        const int rem = rr - img * ohw;
        const int oh = rem / p.OW, ow = rem - oh * p.OW;
        s_img[k] = img;
        s_ih0[k] = oh * p.stride - p.pad;
        s_iw0[k] = ow * p.stride - p.pad;
    }
    f32x4v st[NST];
    auto gather = [&](int tap) {
        const int kh = tap / p.KW, kw = tap - kh * p.KW;
#pragma unroll
        for (int k = 0; k < NST; ++k) {
            f32x4v v = {0.f, 0.f, 0.f, 0.f};
            const int ih = s_ih0[k] + kh, iw = s_iw0[k] + kw;
            if (s_rowok[k] && ih >= 0 && ih < p.H && iw >= 0 && iw < p.W)
                v = *(const f32x4v*)(actg + (long long)((s_img[k] * p.H + ih) * p.W + iw) * p.lda + s_c[k]);
            st[k] = v;
        }
    };
    auto scatter = [&](int buf) {
        float* L = lds + buf * 48 * RS;
#pragma unroll
        for (int k = 0; k < NST; ++k)
            if (s_use[k]) *(f32x4v*)(L + s_row[k] * RS + s_c[k]) = st[k];
    };

    f32x4v acc[3];
#pragma unroll
    for (int nb = 0; nb < 3; ++nb) acc[nb] = f32x4v{0.f, 0.f, 0.f, 0.f};
    f32x4v a_cur[SK_U], a_nxt[SK_U];
    auto load_chunk = [&](int q0, f32x4v* dst) {
#pragma unroll
        for (int u = 0; u < SK_U; ++u) dst[u] = __builtin_nontemporal_load((const f32x4v*)(wrow + (long long)(q0 + u) * 16));
    };
    const int n_groups = taps * gpt;               // weights of one output channel are contiguous over (tap, ci)
    gather(0);
    scatter(0);
    load_chunk(0, a_cur);
    __syncthreads();
    for (int tap = 0; tap < taps; ++tap) {
        if (tap + 1 < taps) gather(tap + 1);
        const float* L = lds + (tap & 1) * 48 * RS + 4 * kq;
        for (int cg0 = 0; cg0 < gpt; cg0 += SK_U) {
            const int q0 = tap * gpt + cg0;
            if (q0 + SK_U < n_groups) load_chunk(q0 + SK_U, a_nxt);
#pragma unroll
            for (int u = 0; u < SK_U; ++u) {
                f32x4v b4[3];
#pragma unroll
                for (int nb = 0; nb < 3; ++nb) b4[nb] = *(const f32x4v*)(L + (nb * 16 + m) * RS + (cg0 + u) * 16);
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int nb = 0; nb < 3; ++nb)
                        acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[u][t], b4[nb][t], acc[nb], 0, 0, 0);
            }
#pragma unroll
            for (int u = 0; u < SK_U; ++u) a_cur[u] = a_nxt[u];
        }
        if (tap + 1 < taps) scatter((tap + 1) & 1);
        __syncthreads();
    }
    float* outg = p.out + (long long)g * p.rows_out * p.ldo;
#pragma unroll
    for (int nb = 0; nb < 3; ++nb) {
        const int ro = nb * 16 + m;
        if (ro < p.rows_out) *(f32x4v*)(outg + (long long)ro * p.ldo + co0 + 4 * kq) = acc[nb];
    }
}

// data gradient of a stride-1 convolution: dx[pix][ci] = sum_{tap,co} dy[pix + pad - (kh,kw)][co] * w[co][tap][ci]
// Weights are read in their forward layout; a lane loads 8 bytes (2 consecutive ci) of row co(kq,t): M index = ci,
// two 16-wide M blocks per wave with the interleaved assignment ci = ci0 + 2*m + b.
__global__ __launch_bounds__(512) void skinny_conv_dgrad_kernel(SkinnyArgs p) {
    // here: act = dy [groups][rows_in][lda] (rows_in = ipg*H*W, channels = Cout of the forward conv = p.Cin field),
    //       out = dx [groups][rows_out = rows_in][ldo] with p.Cout = forward Cin channels
    extern __shared__ __attribute__((aligned(16))) float lds[];
    typedef float f32x2v __attribute__((ext_vector_type(2)));
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int m = lane & 15, kq = lane >> 4;
    const int g = blockIdx.y;
    const int ci0 = blockIdx.x * 256 + wave * 32;       // 8 waves x 32 input channels
    const int RS = p.CS + SK_PADF;
    const int hw = p.H * p.W;
    const int taps = p.KH * p.KW;
    const int Cdy = p.Cin;                          // reduction channels (forward Cout)
    const int Cdx = p.Cout;                         // output channels (forward Cin)
    const long long co_stride = (long long)taps * Cdx;   // floats between consecutive co rows of the forward pack

    int n_img[3], n_h[3], n_w[3];
    bool n_ok[3];
#pragma unroll
    for (int nb = 0; nb < 3; ++nb) {
        const int ro = nb * 16 + m;
        n_ok[nb] = ro < p.rows_out;
        const int rr = n_ok[nb] ? ro : 0;
        const int img = rr / hw;
        const int rem = rr - img * hw;
        n_img[nb] = img;
        n_h[nb] = rem / p.W;
        n_w[nb] = rem - n_h[nb] * p.W;
    }
    const float* wbase = p.w + (long long)g * p.wgs + ci0 + 2 * m;
    const float* actg = p.act + (long long)g * p.rows_in * p.lda;

    f32x4v acc[2][3];
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int nb = 0; nb < 3; ++nb) acc[b][nb] = f32x4v{0.f, 0.f, 0.f, 0.f};

    // dy is staged in channel slices of CS = p.CS reduction channels (LDS footprint rows x (CS+8) x 4 B: 47 KB at CS = 256
    // leaves room for the other stream's workgroups on the CU; one 92 KB slice does not)
    const int CS = p.CS;
    const int gpt = CS / 16;                        // k16 groups (16 co each) per tap and slice
    const int n_groups = taps * gpt;
    constexpr int U = 4;                            // groups per chunk: 4 groups x 4 steps = 16 loads of 8 B in flight
    f32x2v a_cur[U][4], a_nxt[U][4];
    for (int sl = 0; sl < Cdy / CS; ++sl) {
        __syncthreads();
        {
            const int q4 = CS / 4;
            for (int i = tid; i < p.rows_in * q4; i += 512) {
                const int r = i / q4, c = (i - r * q4) * 4;
                *(f32x4v*)(lds + r * RS + c) = *(const f32x4v*)(actg + (long long)r * p.lda + sl * CS + c);
            }
        }
        __syncthreads();
        auto load_chunk = [&](int q0, f32x2v (*dst)[4]) {
            const int tap = q0 / gpt;
            const int cg0 = q0 - tap * gpt;
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int co = sl * CS + (cg0 + u) * 16 + 4 * kq + t;
                    dst[u][t] = __builtin_nontemporal_load((const f32x2v*)(wbase + (long long)co * co_stride + (long long)tap * Cdx));
                }
        };
        load_chunk(0, a_cur);
        for (int q0 = 0; q0 < n_groups; q0 += U) {
            if (q0 + U < n_groups) load_chunk(q0 + U, a_nxt);
            const int tap = q0 / gpt;
            const int cg0 = q0 - tap * gpt;
            const int kh = tap / p.KW, kw = tap - kh * p.KW;
            int boff[3];
#pragma unroll
            for (int nb = 0; nb < 3; ++nb) {
                const int ih = n_h[nb] + p.pad - kh, iw = n_w[nb] + p.pad - kw;     // dy pixel feeding dx(h,w) through tap (kh,kw)
                const bool ok = n_ok[nb] && ih >= 0 && ih < p.H && iw >= 0 && iw < p.W;
                boff[nb] = ok ? ((n_img[nb] * p.H + ih) * p.W + iw) * RS + 4 * kq + cg0 * 16 : -1;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                f32x4v b4[3];
#pragma unroll
                for (int nb = 0; nb < 3; ++nb) {
                    b4[nb] = f32x4v{0.f, 0.f, 0.f, 0.f};
                    if (boff[nb] >= 0) b4[nb] = *(const f32x4v*)(lds + boff[nb] + u * 16);
                }
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int b = 0; b < 2; ++b)
#pragma unroll
                        for (int nb = 0; nb < 3; ++nb)
                            acc[b][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[u][t][b], b4[nb][t], acc[b][nb], 0, 0, 0);
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int t = 0; t < 4; ++t) a_cur[u][t] = a_nxt[u][t];
        }
    }
    // D[i][j]: i = 4*kq + e -> M row -> ci = ci0 + 2*i + b ; j = m -> pixel
    float* outg = p.out + (long long)g * p.rows_out * p.ldo;
#pragma unroll
    for (int nb = 0; nb < 3; ++nb) {
        const int ro = nb * 16 + m;
        if (ro >= p.rows_out) continue;
        float* o = outg + (long long)ro * p.ldo + ci0 + 8 * kq;     // rows i = 4kq..4kq+3 -> ci0 + 8kq + {0..7}
        f32x4v lo, hi;
        lo[0] = acc[0][nb][0]; lo[1] = acc[1][nb][0]; lo[2] = acc[0][nb][1]; lo[3] = acc[1][nb][1];
        hi[0] = acc[0][nb][2]; hi[1] = acc[1][nb][2]; hi[2] = acc[0][nb][3]; hi[3] = acc[1][nb][3];
        *(f32x4v*)(o) = lo;
        *(f32x4v*)(o + 4) = hi;
    }
}

// forward, per-tap staging, bf16x3 (trunk.7.C1 / shortcut): the gathered [48][Cin] im2col rows of one tap are split while
// they are scattered into the three bf16 planes (double buffered: 2 x 3 x 48 x (Cin + 16) x 2 B = 153 KB at Cin = 256)
template <bool LN>
__global__ __launch_bounds__(1024) void skinny_conv_fwd_tap_x3_kernel(SkinnyArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned short ldh[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int m = lane & 15, kq = lane >> 4;
    const int g = blockIdx.y;
    const int co0 = blockIdx.x * 256 + wave * 16;
    const int RS = p.Cin + SK_PADH;
    const int PL = 48 * RS;
    const int ohw = p.OH * p.OW;
    const int taps = p.KH * p.KW;
    const int gpt = p.Cin / 16;
    const int q4 = p.Cin / 4;
    constexpr int NST = 3;
    const float* wrow = sk_wbase<LN>(p.w + (long long)g * p.wgs + (long long)co0 * p.K, m, kq, p.K);
    const float* actg = p.act + (long long)g * p.rows_in * p.lda;

    int s_off[NST], s_c[NST], s_img[NST], s_ih0[NST], s_iw0[NST];
    bool s_use[NST], s_rowok[NST];
#pragma unroll
    for (int k = 0; k < NST; ++k) {
        const int i = tid + k * 1024;
        s_use[k] = i < 48 * q4;
        const int r = s_use[k] ? i / q4 : 0;
        s_c[k] = (i - r * q4) * 4;
        s_off[k] = r * RS + sk_perm(s_c[k]);
        s_rowok[k] = s_use[k] && r < p.rows_out;
        const int rr = s_rowok[k] ? r : 0;
        const int img = rr / ohw;
        const int rem = rr - img * ohw;
        const int oh = rem / p.OW, ow = rem - oh * p.OW;
        s_img[k] = img;
        s_ih0[k] = oh * p.stride - p.pad;
        s_iw0[k] = ow * p.stride - p.pad;
    }
    f32x4v st[NST];
    auto gather = [&](int tap) {
        const int kh = tap / p.KW, kw = tap - kh * p.KW;
#pragma unroll
        for (int k = 0; k < NST; ++k) {
            f32x4v v = {0.f, 0.f, 0.f, 0.f};
            const int ih = s_ih0[k] + kh, iw = s_iw0[k] + kw;
            if (s_rowok[k] && ih >= 0 && ih < p.H && iw >= 0 && iw < p.W)
                v = *(const f32x4v*)(actg + (long long)((s_img[k] * p.H + ih) * p.W + iw) * p.lda + s_c[k]);
            st[k] = v;
        }
    };
    auto scatter = [&](int buf) {
        unsigned short* L = ldh + buf * 3 * PL;
#pragma unroll
        for (int k = 0; k < NST; ++k)
            if (s_use[k]) sk_store3(L, s_off[k], PL, st[k]);
    };

    f32x4v acc[3];
#pragma unroll
    for (int nb = 0; nb < 3; ++nb) acc[nb] = f32x4v{0.f, 0.f, 0.f, 0.f};
    f32x4v a_cur[SK_U];
#pragma unroll
    for (int u = 0; u < SK_U; ++u) a_cur[u] = __builtin_nontemporal_load((const f32x4v*)sk_wptr<LN>(wrow, u, p.K));
    const int n_groups = taps * gpt;
    gather(0);
    scatter(0);
    __syncthreads();
    for (int tap = 0; tap < taps; ++tap) {
        if (tap + 1 < taps) gather(tap + 1);
        const unsigned short* L = ldh + (tap & 1) * 3 * PL + 8 * kq;
        for (int cg0 = 0; cg0 < gpt; cg0 += SK_U) {
            const int q0 = tap * gpt + cg0;
            const bool more = q0 + SK_U < n_groups;
#pragma unroll
            for (int u = 0; u < SK_U; u += 2) {
                bf16x8 a[3], b[3][3];
                sk_wfix<LN>(a_cur[u], a_cur[u + 1]);
                sk_split_a(a_cur[u], a_cur[u + 1], a);
                if (more) {
                    a_cur[u] = __builtin_nontemporal_load((const f32x4v*)sk_wptr<LN>(wrow, q0 + SK_U + u, p.K));
                    a_cur[u + 1] = __builtin_nontemporal_load((const f32x4v*)sk_wptr<LN>(wrow, q0 + SK_U + u + 1, p.K));
                }
#pragma unroll
                for (int nb = 0; nb < 3; ++nb)
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl)
                        b[nb][pl] = __builtin_bit_cast(bf16x8, *(const u32x4v*)(L + pl * PL + (nb * 16 + m) * RS + (cg0 + u) * 16));
                SK_X3_MFMA(acc, a, b)
            }
        }
        if (tap + 1 < taps) scatter((tap + 1) & 1);
        __syncthreads();
    }
    float* outg = p.out + (long long)g * p.rows_out * p.ldo;
#pragma unroll
    for (int nb = 0; nb < 3; ++nb) {
        const int ro = nb * 16 + m;
        if (ro < p.rows_out) *(f32x4v*)(outg + (long long)ro * p.ldo + co0 + 4 * kq) = acc[nb];
    }
}

// Entry of a down-sampling residual block for one episode in ONE launch (SimpleBlock.forward, backbone.py:251-261, first
// half): c1 = C1(x) (3x3, stride s, pad 1), r1 = ReLU(BN1(c1)) with the episode's own mini-batch statistics, and
// sc = shortcut(x) (1x1, stride s, pad 0).  The shortcut samples exactly the pixels of C1's centre tap, so it runs as a tenth
// "tap" pass over the re-gathered centre tile with its own weight stream and accumulators; every wave holds all <= 48 pixels
// of its 16 output channels, so the BatchNorm statistics are an in-register reduction (two-pass mean / variance).
struct SkinnyEntryArgs {
    SkinnyArgs c;              // C1 geometry; c.out = raw c1
    const float* w_sc;         // [groups][Cout][Cin]
    long long wscs;            // group stride of w_sc
    float* sc;                 // raw shortcut output [groups][rows_out][ldo]
    float* r1;                 // ReLU(BN1(c1))
    const float* gamma; const float* beta; long long gbs;
    float* mean; float* rstd;  // [groups][Cout]
    float eps;
};

template <int NW, bool LN>
__global__ __launch_bounds__(64 * NW) void skinny_block_entry_x3_kernel(SkinnyEntryArgs q) {
    const SkinnyArgs& p = q.c;
    extern __shared__ __attribute__((aligned(16))) unsigned short ldh[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int m = lane & 15, kq = lane >> 4;
    const int g = blockIdx.y;
    const int co0 = blockIdx.x * (16 * NW) + wave * 16;
    const int RS = p.Cin + SK_PADH;
    const int PL = 48 * RS;
    const int ohw = p.OH * p.OW;
    const int taps = p.KH * p.KW;
    const int gpt = p.Cin / 16;
    const int q4 = p.Cin / 4;
    constexpr int NST = 48 / NW;                   // float4 staging slots per thread: 48 rows x (Cin <= 256)/4 = 3072 slots
    const float* wrow = sk_wbase<LN>(p.w + (long long)g * p.wgs + (long long)co0 * p.K, m, kq, p.K);
    const float* wsrow = sk_wbase<LN>(q.w_sc + (long long)g * q.wscs + (long long)co0 * p.Cin, m, kq, p.Cin);
    const float* actg = p.act + (long long)g * p.rows_in * p.lda;

    const int centre = (p.KH / 2) * p.KW + p.KW / 2;
    f32x4v st[NST];
    // staging slot k of this thread: row (tid + 64 NW k) / q4 of the 48-row im2col tile, 4 channels; the descriptors are
    // recomputed per pass (10 times per kernel) instead of living in registers through the MFMA loop
    auto gather = [&](int pass) {
        const int tap = pass < taps ? pass : centre;
        const int kh = tap / p.KW, kw = tap - kh * p.KW;
#pragma unroll
        for (int k = 0; k < NST; ++k) {
            const int i = tid + k * 64 * NW;
            const int r = i / q4;
            const int c = (i - r * q4) * 4;
            f32x4v v = {0.f, 0.f, 0.f, 0.f};
            if (r < p.rows_out) {
                const int img = r / ohw;
                const int rem = r - img * ohw;
                const int oh = rem / p.OW, ow = rem - oh * p.OW;
                const int ih = oh * p.stride - p.pad + kh, iw = ow * p.stride - p.pad + kw;
                if (ih >= 0 && ih < p.H && iw >= 0 && iw < p.W)
                    v = *(const f32x4v*)(actg + (long long)((img * p.H + ih) * p.W + iw) * p.lda + c);
            }
            st[k] = v;
        }
    };
    auto scatter = [&](int buf) {
        unsigned short* L = ldh + buf * 3 * PL;
#pragma unroll
        for (int k = 0; k < NST; ++k) {
            const int i = tid + k * 64 * NW;
            const int r = i / q4;
            if (r < 48) sk_store3(L, r * RS + sk_perm((i - r * q4) * 4), PL, st[k]);
        }
    };

    f32x4v acc[3], acs[3];
#pragma unroll
    for (int nb = 0; nb < 3; ++nb) acc[nb] = acs[nb] = f32x4v{0.f, 0.f, 0.f, 0.f};
    const int n_groups = taps * gpt;               // C1's k16 groups; the shortcut's gpt groups follow as pass `taps`
    auto wptr = [&](int qq) { return qq < n_groups ? sk_wptr<LN>(wrow, qq, p.K) : sk_wptr<LN>(wsrow, qq - n_groups, p.Cin); };
    f32x4v a_cur[SK_U];
#pragma unroll
    for (int u = 0; u < SK_U; ++u) a_cur[u] = __builtin_nontemporal_load((const f32x4v*)sk_wptr<LN>(wrow, u, p.K));
    gather(0);
    scatter(0);
    __syncthreads();
    auto pass_body = [&](int pass, f32x4v (&A)[3]) {
        const unsigned short* L = ldh + (pass & 1) * 3 * PL + 8 * kq;
        for (int cg0 = 0; cg0 < gpt; cg0 += SK_U) {
            const int q0 = pass * gpt + cg0;
            const bool more = q0 + SK_U < n_groups + gpt;
#pragma unroll
            for (int u = 0; u < SK_U; u += 2) {
                bf16x8 a[3], b[3][3];
                sk_wfix<LN>(a_cur[u], a_cur[u + 1]);
                sk_split_a(a_cur[u], a_cur[u + 1], a);
                if (more) {
                    a_cur[u] = __builtin_nontemporal_load((const f32x4v*)wptr(q0 + SK_U + u));
                    a_cur[u + 1] = __builtin_nontemporal_load((const f32x4v*)wptr(q0 + SK_U + u + 1));
                }
#pragma unroll
                for (int nb = 0; nb < 3; ++nb)
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl)
                        b[nb][pl] = __builtin_bit_cast(bf16x8, *(const u32x4v*)(L + pl * PL + (nb * 16 + m) * RS + (cg0 + u) * 16));
                SK_X3_MFMA(A, a, b)
            }
        }
    };
    for (int pass = 0; pass < taps; ++pass) {
        gather(pass + 1);
        pass_body(pass, acc);
        scatter((pass + 1) & 1);
        __syncthreads();
    }
    pass_body(taps, acs);

    // ---- epilogue: raw outputs, BatchNorm statistics of c1 over the episode's pixels, r1 = ReLU(BN1(c1))
    const float inv = 1.f / (float)p.rows_out;
    f32x4v s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int nb = 0; nb < 3; ++nb)
        if (nb * 16 + m < p.rows_out) s += acc[nb];
    f32x4v mu, rs;
#pragma unroll
    for (int e = 0; e < 4; ++e) mu[e] = sk_row16_sum(s[e]) * inv;
    s = f32x4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int nb = 0; nb < 3; ++nb)
        if (nb * 16 + m < p.rows_out) {
            const f32x4v d = acc[nb] - mu;
            s += d * d;
        }
#pragma unroll
    for (int e = 0; e < 4; ++e) rs[e] = 1.0f / sqrtf(sk_row16_sum(s[e]) * inv + q.eps);
    const int co = co0 + 4 * kq;
    const f32x4v ga = *(const f32x4v*)(q.gamma + g * q.gbs + co), be = *(const f32x4v*)(q.beta + g * q.gbs + co);
    if (m == 0) {
        *(f32x4v*)(q.mean + (long long)g * p.Cout + co) = mu;
        *(f32x4v*)(q.rstd + (long long)g * p.Cout + co) = rs;
    }
    const long long ob = (long long)g * p.rows_out * p.ldo;
#pragma unroll
    for (int nb = 0; nb < 3; ++nb) {
        const int ro = nb * 16 + m;
        if (ro >= p.rows_out) continue;
        const long long o = ob + (long long)ro * p.ldo + co;
        *(f32x4v*)(p.out + o) = acc[nb];
        *(f32x4v*)(q.sc + o) = acs[nb];
        f32x4v y = (acc[nb] - mu) * rs * ga + be;
#pragma unroll
        for (int e = 0; e < 4; ++e) y[e] = fmaxf(y[e], 0.f);
        *(f32x4v*)(q.r1 + o) = y;
    }
}

// data gradient, bf16x3: dy is split once into the three LDS planes (reduction index = forward output channel, same slot
// permutation); a lane's four 8-byte weight loads of a k16 group give it 4 consecutive reduction rows of 2 input channels.
//
// BNB: the BatchNorm + ReLU in front of the convolution is differentiated in the epilogue (SimpleBlock: r1 = ReLU(BN1(c1)) feeds
// C2, backbone.py:253-255): with g = dx * (r1 > 0) and xhat = (c1 - mean) * rstd the kernel writes
// dc1 = gamma * rstd * (g - mean_rows(g) - xhat * mean_rows(g * xhat)), dgamma = sum_rows(g * xhat), dbeta = sum_rows(g)
// over the episode's rows, all of which sit in this wave's accumulators (16 lanes x 3 blocks per channel).
struct SkinnyBnArgs {
    const float* x_raw;        // c1 [groups][rows][ldo]
    const float* relu_out;     // r1 [groups][rows][ldo]
    const float* mean; const float* rstd; const float* gamma; long long gbs;
    float* dgamma; float* dbeta;                   // [groups][C]
};

template <bool BNB, int NW>
__global__ __launch_bounds__(64 * NW) void skinny_conv_dgrad_x3_kernel(SkinnyArgs p, SkinnyBnArgs bn) {
    extern __shared__ __attribute__((aligned(16))) unsigned short ldh[];
    typedef float f32x2v __attribute__((ext_vector_type(2)));
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int m = lane & 15, kq = lane >> 4;
    const int g = blockIdx.y;
    const int ci0 = blockIdx.x * (32 * NW) + wave * 32;
    const int Cdy = p.Cin, Cdx = p.Cout;
    const int RS = Cdy + SK_PADH;
    const int PL = p.rows_in * RS;
    const int hw = p.H * p.W;
    const int taps = p.KH * p.KW;
    const long long co_stride = (long long)taps * Cdx;

    int n_img[3], n_h[3], n_w[3];
    bool n_ok[3];
#pragma unroll
    for (int nb = 0; nb < 3; ++nb) {
        const int ro = nb * 16 + m;
        n_ok[nb] = ro < p.rows_out;
        const int rr = n_ok[nb] ? ro : 0;
        const int img = rr / hw;
        const int rem = rr - img * hw;
        n_img[nb] = img;
        n_h[nb] = rem / p.W;
        n_w[nb] = rem - n_h[nb] * p.W;
    }
    const float* wbase = p.w + (long long)g * p.wgs + ci0 + 2 * m;
    const float* actg = p.act + (long long)g * p.rows_in * p.lda;

    constexpr int U = 4;                            // k16 groups per chunk: 16 loads of 8 B per lane in flight
    const int gpt = Cdy / 16;
    const int n_groups = taps * gpt;
    f32x2v a_cur[U][4];
    auto load_group = [&](int q, f32x2v* dst) {
        const int tap = q / gpt;
        const int cg = q - tap * gpt;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int co = cg * 16 + 4 * kq + t;
            dst[t] = __builtin_nontemporal_load((const f32x2v*)(wbase + (long long)co * co_stride + (long long)tap * Cdx));
        }
    };
#pragma unroll
    for (int u = 0; u < U; ++u) load_group(u, a_cur[u]);
    {
        const int q4 = Cdy / 4;
        for (int i = tid; i < p.rows_in * q4; i += 64 * NW) {
            const int r = i / q4, c = (i - r * q4) * 4;
            sk_store3(ldh, r * RS + sk_perm(c), PL, *(const f32x4v*)(actg + (long long)r * p.lda + c));
        }
    }
    __syncthreads();

    f32x4v acc[2][3];
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int nb = 0; nb < 3; ++nb) acc[b][nb] = f32x4v{0.f, 0.f, 0.f, 0.f};

    for (int q0 = 0; q0 < n_groups; q0 += U) {
        const bool more = q0 + U < n_groups;
        const int tap = q0 / gpt;
        const int cg0 = q0 - tap * gpt;
        const int kh = tap / p.KW, kw = tap - kh * p.KW;
        int boff[3];
#pragma unroll
        for (int nb = 0; nb < 3; ++nb) {
            const int ih = n_h[nb] + p.pad - kh, iw = n_w[nb] + p.pad - kw;
            const bool ok = n_ok[nb] && ih >= 0 && ih < p.H && iw >= 0 && iw < p.W;
            boff[nb] = ok ? ((n_img[nb] * p.H + ih) * p.W + iw) * RS + 8 * kq + cg0 * 16 : -1;
        }
#pragma unroll
        for (int u = 0; u < U; u += 2) {
            bf16x8 a[2][3], b[3][3];
#pragma unroll
            for (int bb = 0; bb < 2; ++bb) {
                const f32x4v w0 = {a_cur[u][0][bb], a_cur[u][1][bb], a_cur[u][2][bb], a_cur[u][3][bb]};
                const f32x4v w1 = {a_cur[u + 1][0][bb], a_cur[u + 1][1][bb], a_cur[u + 1][2][bb], a_cur[u + 1][3][bb]};
                sk_split_a(w0, w1, a[bb]);
            }
            if (more) {
                load_group(q0 + U + u, a_cur[u]);
                load_group(q0 + U + u + 1, a_cur[u + 1]);
            }
#pragma unroll
            for (int nb = 0; nb < 3; ++nb)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) {
                    u32x4v z = {0u, 0u, 0u, 0u};
                    if (boff[nb] >= 0) z = *(const u32x4v*)(ldh + pl * PL + boff[nb] + u * 16);
                    b[nb][pl] = __builtin_bit_cast(bf16x8, z);
                }
            SK_X3_MFMA(acc[0], a[0], b)
            SK_X3_MFMA(acc[1], a[1], b)
        }
    }
    float* outg = p.out + (long long)g * p.rows_out * p.ldo;
    // channels of this lane: ci0 + 8 kq + j, j = 2 e + b  <->  acc[b][nb][e]
    f32x4v d[3][2];
#pragma unroll
    for (int nb = 0; nb < 3; ++nb) {
        d[nb][0][0] = acc[0][nb][0]; d[nb][0][1] = acc[1][nb][0]; d[nb][0][2] = acc[0][nb][1]; d[nb][0][3] = acc[1][nb][1];
        d[nb][1][0] = acc[0][nb][2]; d[nb][1][1] = acc[1][nb][2]; d[nb][1][2] = acc[0][nb][3]; d[nb][1][3] = acc[1][nb][3];
    }
    const int cc = ci0 + 8 * kq;
    if constexpr (BNB) {
        const long long gb = (long long)g * p.rows_out * p.ldo;
        f32x4v mu[2], rs[2], ga[2], xh[3][2], s1[2], s2[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            mu[h] = *(const f32x4v*)(bn.mean + (long long)g * Cdx + cc + 4 * h);
            rs[h] = *(const f32x4v*)(bn.rstd + (long long)g * Cdx + cc + 4 * h);
            ga[h] = *(const f32x4v*)(bn.gamma + g * bn.gbs + cc + 4 * h);
            s1[h] = s2[h] = f32x4v{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int nb = 0; nb < 3; ++nb) {
            const int ro = nb * 16 + m;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                xh[nb][h] = f32x4v{0.f, 0.f, 0.f, 0.f};
                if (ro < p.rows_out) {
                    const long long o = gb + (long long)ro * p.ldo + cc + 4 * h;
                    const f32x4v r = *(const f32x4v*)(bn.relu_out + o);
                    xh[nb][h] = (*(const f32x4v*)(bn.x_raw + o) - mu[h]) * rs[h];
#pragma unroll
                    for (int e = 0; e < 4; ++e) d[nb][h][e] = r[e] > 0.f ? d[nb][h][e] : 0.f;
                    s1[h] += d[nb][h];
                    s2[h] += d[nb][h] * xh[nb][h];
                } else {
                    d[nb][h] = f32x4v{0.f, 0.f, 0.f, 0.f};
                }
            }
        }
        const float inv = 1.f / (float)p.rows_out;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                s1[h][e] = sk_row16_sum(s1[h][e]);
                s2[h][e] = sk_row16_sum(s2[h][e]);
            }
            if (m == 0) {
                *(f32x4v*)(bn.dgamma + (long long)g * Cdx + cc + 4 * h) = s2[h];
                *(f32x4v*)(bn.dbeta + (long long)g * Cdx + cc + 4 * h) = s1[h];
            }
            const f32x4v m1 = s1[h] * inv, m2 = s2[h] * inv, kk = ga[h] * rs[h];
#pragma unroll
            for (int nb = 0; nb < 3; ++nb) d[nb][h] = kk * (d[nb][h] - m1 - xh[nb][h] * m2);
        }
    }
#pragma unroll
    for (int nb = 0; nb < 3; ++nb) {
        const int ro = nb * 16 + m;
        if (ro >= p.rows_out) continue;
        float* o = outg + (long long)ro * p.ldo + cc;
        *(f32x4v*)(o) = d[nb][0];
        *(f32x4v*)(o + 4) = d[nb][1];
    }
}

int g_skinny_x3 = 1;             // bf16x3 forms of the weight-streaming kernels (mft_debug_set_conv_tile(8000/8001))
int g_skinny_tap = 1;            // per-tap staged forward for shapes whose activation exceeds LDS (mft_debug_set_conv_tile(7000/7001))
int g_skinny_dgrad_slices = 1;   // reduction-channel slices of the data-gradient kernel (mft_debug_set_conv_tile(6000 + n))

int g_skinny_lines = 1;         // full-line weight loads + DPP fix-up in the bf16x3 forward kernels (mft_debug_set_conv_tile(9700/9701))
int g_skinny_nw = 0;            // waves per workgroup of the per-episode bf16x3 kernels: 0 = widest (mft_debug_set_conv_tile(9100 + nw); 9199 = by episode count)

// Workgroups of W waves each own 16*W (forward) or 32*W (data gradient) channels of one episode.  Narrower workgroups fill the
// CUs at small episode batches and are 1.3-2x faster STANDALONE there (E = 32: block entry 122 -> 70 us, exit 204 -> 143 us,
// data gradient 201 -> 156 us; tools/small_e_skinny.py), but every one of them holds ~140 KB of LDS, so a launch that covers all
// 256 CUs leaves no CU on which the trunk stream's convolutions fit: in the two-stream engine that choice is SLOWER
// (50-shot, E = 48: 5.4 vs 6.0 episodes/s; 20-shot, E = 64: 16.3 vs 16.5).  Default: the widest workgroups.
inline int pick_nw(int groups, int wgs_per_group_at_max, int nw_max) {
    if (g_skinny_nw == 0) return nw_max;
    if (g_skinny_nw != 99) return g_skinny_nw <= nw_max ? (g_skinny_nw >= nw_max / 4 ? g_skinny_nw : nw_max / 4) : nw_max;
    int nw = nw_max;
    while (nw > nw_max / 4 && (long long)groups * wgs_per_group_at_max * (nw_max / nw) < 256) nw >>= 1;
    return nw;
}

template <typename K, typename... A>
int launch_big_lds(K kern, size_t max_lds, dim3 grid, dim3 block, size_t lds, hipStream_t s, A... args) {
    // the dynamic-LDS limit is raised once per kernel instantiation (instantiations of one template share the pointer TYPE, so
    // the "done" set is keyed by the function address)
    // (per device: the attribute belongs to the current device's copy of the code object)
    // Entries are added only after the attribute call succeeded (a failure is retried and reported by the next launch), under a
    // mutex: engines on several devices may launch from their own threads.
    static std::mutex mu;
    static const void* done[256];
    static int done_dev[256];
    static int n_done = 0;
    int dev = 0;
    (void)hipGetDevice(&dev);
    bool seen = false;
    {
        std::lock_guard<std::mutex> lock(mu);
        for (int i = 0; i < n_done; ++i) seen = seen || (done[i] == (const void*)kern && done_dev[i] == dev);
    }
    if (!seen) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)max_lds);
        if (e != hipSuccess) return (int)e;
        std::lock_guard<std::mutex> lock(mu);
        bool dup = false;
        for (int i = 0; i < n_done; ++i) dup = dup || (done[i] == (const void*)kern && done_dev[i] == dev);
        if (!dup && n_done < 256) { done[n_done] = (const void*)kern; done_dev[n_done] = dev; ++n_done; }
    }
    hipLaunchKernelGGL(kern, grid, block, lds, s, args...);
    return mft_launch_status();
}

int pick_slice(int rows_in, int Cin) {
    for (int cs = Cin; cs >= 128; cs /= 2) {
        if (cs % 128 != 0 || Cin % cs != 0) continue;
        if ((long long)rows_in * (cs + SK_PADF) * 4 <= 100 * 1024) return cs;
    }
    return 0;
}

}  // namespace

void mft_skinny_set_dgrad_slices(int n) { g_skinny_dgrad_slices = n; }
void mft_skinny_set_tap(int v) { g_skinny_tap = v; }
void mft_skinny_set_x3(int v) { g_skinny_x3 = v; }
void mft_skinny_set_nw(int v) { g_skinny_nw = v; }
void mft_skinny_set_lines(int v) { g_skinny_lines = v; }

// Returns MFT_EINVAL when the shape is outside the skinny kernel's domain (callers fall back to the generic kernel).
static int skinny_fwd_impl(const float* in, int ldi, const float* w, float* out, int ldo, int n_img, int H, int W,
                           int Cin, int Cout, int KH, int KW, int stride, int pad, int imgs_per_group,
                           long long w_group_stride, const SkinnyExitArgs* ex, hipStream_t s) {
    if (imgs_per_group <= 0 || w_group_stride == 0 || n_img % imgs_per_group != 0) return MFT_EINVAL;
    const int OH = (H + 2 * pad - KH) / stride + 1, OW = (W + 2 * pad - KW) / stride + 1;
    const int rows_out = imgs_per_group * OH * OW, rows_in = imgs_per_group * H * W;
    if (rows_out > 48 || Cout % 256 != 0 || Cin % 128 != 0 || ldi % 4 != 0 || ldo % 4 != 0) return MFT_EINVAL;
    const int cs = pick_slice(rows_in, Cin);
    SkinnyArgs p;
    p.act = in; p.w = w; p.out = out; p.lda = ldi; p.ldo = ldo;
    p.H = H; p.W = W; p.Cin = Cin; p.OH = OH; p.OW = OW; p.Cout = Cout; p.KH = KH; p.KW = KW; p.stride = stride; p.pad = pad;
    p.ipg = imgs_per_group; p.rows_in = rows_in; p.rows_out = rows_out; p.wgs = w_group_stride;
    p.K = KH * KW * Cin; p.CS = cs;
    dim3 grid(Cout / 256, n_img / imgs_per_group, 1);
    if (cs != Cin) {
        if (ex != nullptr) return MFT_EINVAL;
        // the whole activation does not fit one LDS slice (trunk.7.C1 / shortcut: 180 input pixels): per-tap staging
        if (Cin > 256 || Cin % (16 * SK_U) != 0 || g_skinny_tap == 0) return MFT_EINVAL;
        if (g_skinny_x3) {
            const size_t lds_h = (size_t)2 * 3 * 48 * (Cin + SK_PADH) * 2;
            static bool attr_tx = false;
            if (!attr_tx) {
                hipError_t e = hipFuncSetAttribute((const void*)skinny_conv_fwd_tap_x3_kernel<true>,
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024);
                if (e == hipSuccess)
                    e = hipFuncSetAttribute((const void*)skinny_conv_fwd_tap_x3_kernel<false>,
                                            hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024);
                if (e != hipSuccess) return (int)e;
                attr_tx = true;
            }
            if (g_skinny_lines) hipLaunchKernelGGL(skinny_conv_fwd_tap_x3_kernel<true>, grid, dim3(1024), lds_h, s, p);
            else hipLaunchKernelGGL(skinny_conv_fwd_tap_x3_kernel<false>, grid, dim3(1024), lds_h, s, p);
            return mft_launch_status();
        }
        const size_t lds_t = (size_t)2 * 48 * (Cin + SK_PADF) * sizeof(float);
        static bool attr_t = false;
        if (!attr_t) {
            hipError_t e = hipFuncSetAttribute((const void*)skinny_conv_fwd_tap_kernel,
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 110 * 1024);
            if (e != hipSuccess) return (int)e;
            attr_t = true;
        }
        hipLaunchKernelGGL(skinny_conv_fwd_tap_kernel, grid, dim3(1024), lds_t, s, p);
        return mft_launch_status();
    }
    if (g_skinny_x3 && (size_t)3 * rows_in * (Cin + SK_PADH) * 2 <= 150 * 1024 && Cin % (16 * SK_U) == 0) {
        size_t lds3 = (size_t)3 * rows_in * (Cin + SK_PADH) * 2;
        if (ex != nullptr && lds3 < (size_t)16 * 48 * 16 * 4) lds3 = (size_t)16 * 48 * 16 * 4;    // the pool's per-wave output tiles
        const int groups = n_img / imgs_per_group;
        const int nw = pick_nw(groups, Cout / 256, 16);
        const dim3 g2(Cout / (16 * nw), groups, 1);
        const SkinnyExitArgs e0 = ex ? *ex : SkinnyExitArgs{};
#define SK_FWD(EX, NW) (g_skinny_lines ? launch_big_lds(skinny_conv_fwd_x3_kernel<EX, NW, true>, 150 * 1024, g2, dim3(64 * NW), lds3, s, p, e0) \
                                       : launch_big_lds(skinny_conv_fwd_x3_kernel<EX, NW, false>, 150 * 1024, g2, dim3(64 * NW), lds3, s, p, e0))
        if (ex != nullptr) return nw == 16 ? SK_FWD(true, 16) : nw == 8 ? SK_FWD(true, 8) : SK_FWD(true, 4);
        return nw == 16 ? SK_FWD(false, 16) : nw == 8 ? SK_FWD(false, 8) : SK_FWD(false, 4);
#undef SK_FWD
    }
    if (ex != nullptr) return MFT_EINVAL;          // the fused block exit exists in the bf16x3 form only
    const size_t lds = (size_t)rows_in * (cs + SK_PADF) * sizeof(float);
    static MftPerDeviceOnce attr_once;
    if (attr_once.need()) {
        hipError_t e = hipFuncSetAttribute((const void*)skinny_conv_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           110 * 1024);
        if (e != hipSuccess) return (int)e;
        attr_once.mark();
    }
    hipLaunchKernelGGL(skinny_conv_fwd_kernel, grid, dim3(1024), lds, s, p);
    return mft_launch_status();
}

static int skinny_dgrad_impl(const float* dy, int ldy, const float* w, float* dx, int ldx, int n_img, int H, int W,
                             int Cin, int Cout, int KH, int KW, int stride, int pad, int imgs_per_group,
                             long long w_group_stride, const SkinnyBnArgs* bn, hipStream_t s) {
    // forward conv: Cin -> Cout, stride 1, "same" padding: dy and dx have the same H x W
    if (imgs_per_group <= 0 || w_group_stride == 0 || n_img % imgs_per_group != 0 || stride != 1) return MFT_EINVAL;
    if (2 * pad != KH - 1 || 2 * pad != KW - 1) return MFT_EINVAL;
    const int rows = imgs_per_group * H * W;
    if (rows > 48 || Cin % 256 != 0 || Cout % 64 != 0 || ldy % 4 != 0 || ldx % 4 != 0) return MFT_EINVAL;
    int cs = Cout;
    if (g_skinny_dgrad_slices > 1 && Cout % (64 * g_skinny_dgrad_slices) == 0) cs = Cout / g_skinny_dgrad_slices;
    if ((long long)rows * (cs + SK_PADF) * 4 > 100 * 1024) return MFT_EINVAL;
    SkinnyArgs p;
    p.act = dy; p.w = w; p.out = dx; p.lda = ldy; p.ldo = ldx;
    p.H = H; p.W = W; p.Cin = Cout; p.OH = H; p.OW = W; p.Cout = Cin; p.KH = KH; p.KW = KW; p.stride = 1; p.pad = pad;
    p.ipg = imgs_per_group; p.rows_in = rows; p.rows_out = rows; p.wgs = w_group_stride;
    p.K = KH * KW * Cin; p.CS = cs;
    dim3 grid(Cin / 256, n_img / imgs_per_group, 1);
    if (g_skinny_x3 && g_skinny_dgrad_slices <= 1 && (size_t)3 * rows * (Cout + SK_PADH) * 2 <= 150 * 1024 && Cout % 64 == 0) {
        const size_t lds3 = (size_t)3 * rows * (Cout + SK_PADH) * 2;
        const int groups = n_img / imgs_per_group;
        int nw = pick_nw(groups, Cin / 256, 8);
        if (nw > 8) nw = 8;
        const dim3 g2(Cin / (32 * nw), groups, 1);
        const SkinnyBnArgs b0 = bn ? *bn : SkinnyBnArgs{};
#define SK_DG(BB, NW) launch_big_lds(skinny_conv_dgrad_x3_kernel<BB, NW>, 150 * 1024, g2, dim3(64 * NW), lds3, s, p, b0)
        if (bn != nullptr) return nw == 8 ? SK_DG(true, 8) : nw == 4 ? SK_DG(true, 4) : SK_DG(true, 2);
        return nw == 8 ? SK_DG(false, 8) : nw == 4 ? SK_DG(false, 4) : SK_DG(false, 2);
#undef SK_DG
    }
    if (bn != nullptr) return MFT_EINVAL;          // the fused BatchNorm epilogue exists in the bf16x3 form only
    const size_t lds = (size_t)rows * (cs + SK_PADF) * sizeof(float);
    static MftPerDeviceOnce attr_once;
    if (attr_once.need()) {
        hipError_t e = hipFuncSetAttribute((const void*)skinny_conv_dgrad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           110 * 1024);
        if (e != hipSuccess) return (int)e;
        attr_once.mark();
    }
    hipLaunchKernelGGL(skinny_conv_dgrad_kernel, grid, dim3(512), lds, s, p);
    return mft_launch_status();
}

int mft_skinny_fwd_dispatch(const float* in, int ldi, const float* w, float* out, int ldo, int n_img, int H, int W,
                            int Cin, int Cout, int KH, int KW, int stride, int pad, int imgs_per_group,
                            long long w_group_stride, hipStream_t s) {
    return skinny_fwd_impl(in, ldi, w, out, ldo, n_img, H, W, Cin, Cout, KH, KW, stride, pad, imgs_per_group, w_group_stride,
                           nullptr, s);
}

extern "C" int mft_block_exit_small_forward(const float* r1, const float* w_c2, long long w_group_stride, const float* sc,
                                            float* c2, float* out, float* pooled, int n_img, int H, int W, int C,
                                            int imgs_per_group, const float* gamma2, const float* beta2, const float* gamma_sc,
                                            const float* beta_sc, long long gb_group_stride, float* mean2, float* rstd2,
                                            float* mean_sc, float* rstd_sc, float eps, void* stream) {
    if (!sc || !out || !pooled || imgs_per_group <= 0) return MFT_EINVAL;
    SkinnyExitArgs ex{sc, out, pooled, gamma2, beta2, gamma_sc, beta_sc, gb_group_stride, mean2, rstd2, mean_sc, rstd_sc, eps,
                      H * W};
    return skinny_fwd_impl(r1, C, w_c2, c2, C, n_img, H, W, C, C, 3, 3, 1, 1, imgs_per_group, w_group_stride, &ex,
                           (hipStream_t)stream);
}

int mft_skinny_dgrad_dispatch(const float* dy, int ldy, const float* w, float* dx, int ldx, int n_img, int H, int W,
                              int Cin, int Cout, int KH, int KW, int stride, int pad, int imgs_per_group,
                              long long w_group_stride, hipStream_t s) {
    return skinny_dgrad_impl(dy, ldy, w, dx, ldx, n_img, H, W, Cin, Cout, KH, KW, stride, pad, imgs_per_group, w_group_stride,
                             nullptr, s);
}

extern "C" int mft_conv2d_dgrad_bn_backward_small(const float* dy, int ldy, const float* w, float* dx, int ldx, int n_img, int H,
                                                  int W, int Cin, int Cout, int KH, int KW, int pad, int imgs_per_group,
                                                  long long w_group_stride, const float* x_raw, const float* relu_out,
                                                  const float* mean, const float* rstd, const float* gamma,
                                                  long long gb_group_stride, float* dgamma, float* dbeta, void* stream) {
    if (!x_raw || !relu_out || !mean || !rstd || !gamma || !dgamma || !dbeta || ldx != Cin) return MFT_EINVAL;
    SkinnyBnArgs bn{x_raw, relu_out, mean, rstd, gamma, gb_group_stride, dgamma, dbeta};
    return skinny_dgrad_impl(dy, ldy, w, dx, ldx, n_img, H, W, Cin, Cout, KH, KW, 1, pad, imgs_per_group, w_group_stride, &bn,
                             (hipStream_t)stream);
}

extern "C" int mft_block_entry_small_forward(const float* x, int ldx, const float* w_c1, long long w1_group_stride,
                                             const float* w_sc, long long wsc_group_stride, float* c1, float* r1, float* sc,
                                             int n_img, int H, int W, int Cin, int Cout, int stride, int imgs_per_group,
                                             const float* gamma1, const float* beta1, long long gb_group_stride, float* mean1,
                                             float* rstd1, float eps, void* stream) {
    if (imgs_per_group <= 0 || n_img % imgs_per_group != 0 || w1_group_stride == 0 || wsc_group_stride == 0) return MFT_EINVAL;
    const int OH = (H + 2 - 3) / stride + 1, OW = (W + 2 - 3) / stride + 1;
    // the 1x1 / stride s / pad 0 shortcut must sample the centre tap of the 3x3 / stride s / pad 1 convolution
    if ((H - 1) / stride + 1 != OH || (W - 1) / stride + 1 != OW) return MFT_EINVAL;
    const int rows_out = imgs_per_group * OH * OW;
    if (!g_skinny_x3 || rows_out > 48 || Cout % 256 != 0 || Cin % (16 * SK_U) != 0 || Cin > 256 || ldx % 4 != 0) return MFT_EINVAL;
    SkinnyEntryArgs q;
    SkinnyArgs& p = q.c;
    p.act = x; p.w = w_c1; p.out = c1; p.lda = ldx; p.ldo = Cout;
    p.H = H; p.W = W; p.Cin = Cin; p.OH = OH; p.OW = OW; p.Cout = Cout; p.KH = 3; p.KW = 3; p.stride = stride; p.pad = 1;
    p.ipg = imgs_per_group; p.rows_in = imgs_per_group * H * W; p.rows_out = rows_out; p.wgs = w1_group_stride;
    p.K = 9 * Cin; p.CS = Cin;
    q.w_sc = w_sc; q.wscs = wsc_group_stride; q.sc = sc; q.r1 = r1; q.gamma = gamma1; q.beta = beta1; q.gbs = gb_group_stride;
    q.mean = mean1; q.rstd = rstd1; q.eps = eps;
    const size_t lds_h = (size_t)2 * 3 * 48 * (Cin + SK_PADH) * 2;
    const int groups = n_img / imgs_per_group;
    const int nw = pick_nw(groups, Cout / 256, 16);
    const dim3 g2(Cout / (16 * nw), groups, 1);
#define SK_EN(NW) (g_skinny_lines ? launch_big_lds(skinny_block_entry_x3_kernel<NW, true>, 156 * 1024, g2, dim3(64 * NW), lds_h, (hipStream_t)stream, q) \
                                  : launch_big_lds(skinny_block_entry_x3_kernel<NW, false>, 156 * 1024, g2, dim3(64 * NW), lds_h, (hipStream_t)stream, q))
    return nw == 16 ? SK_EN(16) : nw == 8 ? SK_EN(8) : SK_EN(4);
#undef SK_EN
}
