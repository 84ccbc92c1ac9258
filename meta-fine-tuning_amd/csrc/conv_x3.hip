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

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
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

template <int BM, int BN>
__global__ __launch_bounds__(256) void conv_x3_kernel(X3Args p) {
    constexpr int TM = BM / 64;           // 32-row blocks per wave (waves 2 x 2)
    constexpr int TN = BN / 64;
    constexpr int PA = BM / 32;           // A passes: 32 rows per pass, 8 threads x float4 per row
    constexpr int PB = BN / 64;           // B passes: 64 rows per pass, 4 threads x 16 B per row and plane
    constexpr int A_PLANE = BM * X3_RS;   // bf16 elements
    constexpr int B_PLANE = BN * X3_RS;
    extern __shared__ __attribute__((aligned(16))) unsigned short smem[];
    unsigned short* As = smem;                    // [3][BM][RS]
    unsigned short* Bs = smem + 3 * A_PLANE;      // [3][BN][RS]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int r = lane & 31, h = lane >> 5;
    const int nt = blockIdx.x % p.tiles_n;
    const int mt = blockIdx.x / p.tiles_n;
    const int m0 = mt * BM, n0 = nt * BN;

    const int lrow = tid >> 3;
    const int c4 = (tid & 7) * 4;
    const int ohw = p.OH * p.OW;
    long long a_base[PA];
    int a_ih0[PA], a_iw0[PA];
    bool a_ok[PA];
#pragma unroll
    for (int j = 0; j < PA; ++j) {
        const int m = m0 + lrow + 32 * j;
        a_ok[j] = m < p.M;
        const int mm = a_ok[j] ? m : 0;
        const int img = mm / ohw;
        const int rem = mm - img * ohw;
        const int oh = rem / p.OW, ow = rem - oh * p.OW;
        a_ih0[j] = oh * p.stride - p.pad;
        a_iw0[j] = ow * p.stride - p.pad;
        a_base[j] = (long long)img * p.H * p.W;
    }
    const int brow = tid >> 2, bseg = tid & 3;
    const unsigned short* b_ptr[PB];
#pragma unroll
    for (int j = 0; j < PB; ++j) b_ptr[j] = p.w3 + (long long)(n0 + brow + 64 * j) * p.Kpad + bseg * 8;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    f32x4 ra[PA];
    u32x4 rb[PB][3];
    const int nk = p.Kpad / 32;

    auto load_tile = [&](int kt) {
        const int k0 = kt * 32;
        const int khkw = k0 / p.Cin;
        const int ci0 = k0 - khkw * p.Cin;
        const int kh = khkw / p.KW, kw = khkw - kh * p.KW;
#pragma unroll
        for (int j = 0; j < PA; ++j) {
            const int ih = a_ih0[j] + kh, iw = a_iw0[j] + kw;
            const bool ok = a_ok[j] && ih >= 0 && ih < p.H && iw >= 0 && iw < p.W;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (ok) v = *(const f32x4*)(p.in + (a_base[j] + (long long)ih * p.W + iw) * p.ldi + ci0 + c4);
            ra[j] = v;
        }
#pragma unroll
        for (int j = 0; j < PB; ++j)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) rb[j][pl] = *(const u32x4*)(b_ptr[j] + pl * p.plane + k0);
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int j = 0; j < PA; ++j) {
            u32x2 p1, p2, p3;
            split4(ra[j], p1, p2, p3);
            const int off = (lrow + 32 * j) * X3_RS + c4;
            *(u32x2*)(As + off) = p1;
            *(u32x2*)(As + A_PLANE + off) = p2;
            *(u32x2*)(As + 2 * A_PLANE + off) = p3;
        }
#pragma unroll
        for (int j = 0; j < PB; ++j)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
                *(u32x4*)(Bs + pl * B_PLANE + (brow + 64 * j) * X3_RS + bseg * 8) = rb[j][pl];
    };

    load_tile(0);
    store_tile();
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) load_tile(kt + 1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 a[TM][3], b[TN][3];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
                    a[i][pl] = *(const bf16x8*)(As + pl * A_PLANE + (wm * (BM / 2) + i * 32 + r) * X3_RS + kk * 16 + h * 8);
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
                    b[j][pl] = *(const bf16x8*)(Bs + pl * B_PLANE + (wn * (BN / 2) + j * 32 + r) * X3_RS + kk * 16 + h * 8);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    // smallest terms first; (1,1) last
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][2], b[j][0], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][2], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][1], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][0], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][1], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][0], acc[i][j], 0, 0, 0);
                }
        }
        __syncthreads();
        if (kt + 1 < nk) store_tile();
        __syncthreads();
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
                if (m < p.M) p.out[(long long)m * p.ldo + n] = acc[i][j][e];
            }
        }
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

template <int BM, int BN>
int launch_x3(X3Args p, hipStream_t s) {
    const int tiles_m = (p.M + BM - 1) / BM;
    p.tiles_n = p.Cout / BN;
    const size_t lds = (size_t)3 * (BM + BN) * X3_RS * sizeof(unsigned short);
    auto kern = conv_x3_kernel<BM, BN>;
    if (lds > 64 * 1024) {
        static bool attr_done = false;
        if (!attr_done) {
            hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return (int)e;
            attr_done = true;
        }
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)(tiles_m * p.tiles_n)), dim3(256), lds, s, p);
    return mft_launch_status();
}

int g_x3_tile = 0;   // 0 auto; 1: 128x64; 2: 128x128

}  // namespace

extern "C" int mft_split_bf16x3(const float* w, unsigned short* planes, long long n, void* stream) {
    if (n <= 0) return MFT_EINVAL;
    long long blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(split_bf16x3_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, w, planes, n);
    return mft_launch_status();
}

extern "C" int mft_debug_set_x3_tile(int t) {
    g_x3_tile = t;
    return 0;
}

extern "C" int mft_conv2d_nhwc_x3(const float* in, int ldi, const unsigned short* w3, long long plane_elems, float* out,
                                  int ldo, int n_img, int H, int W, int Cin, int Cout, int KH, int KW, int stride,
                                  int pad, void* stream) {
    if (n_img <= 0 || Cin % 32 != 0 || Cout % 64 != 0 || ldi % 4 != 0) return MFT_EINVAL;
    X3Args p;
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
    int tile = g_x3_tile;
    if (tile == 0) tile = (Cout % 128 == 0) ? 2 : 1;
    if (tile == 2 && Cout % 128 == 0) return launch_x3<128, 128>(p, s);
    return launch_x3<128, 64>(p, s);
}
