// NHWC implicit-GEMM convolution / dense GEMM and conv weight-gradient on fp32 MFMA
// (v_mfma_f32_32x32x2_f32: exact f32 products, f32 accumulate) for gfx950.
//
// Replaces the cuDNN/cuBLAS dispatches behind nn.Conv2d / nn.Linear on the reference hot path
// (backbone.py:221-240,408; gnnnet.py:30; gnn.py:38,64-76) and conv weight gradients of
// loss.backward() (finetune.py:293; gnnnet.py:174).
//
// Forward tiling: a 256-thread workgroup (4 waves) owns a BM x BN output tile; K is walked in
// 32-wide steps whose A rows are gathered straight from the NHWC input (im2col is never
// materialised: for a fixed (kh,kw) 32 input channels are one 128-byte run) and whose B rows are
// weight rows packed [Cout][KH][KW][Cin].  Global -> registers -> LDS staging with the next
// K-step's loads in flight during the MFMAs of the current one, double-buffered LDS, one barrier
// per K-step.  LDS rows are padded to 36 floats so the ds_read_b128 fragment reads are
// conflict-free.  Each lane reads 16 contiguous k of its row; MFMA step t pairs k=t (lanes 0-31)
// with k=16+t (lanes 32-63) -- a fixed permutation of the reduction order shared by A and B.
#include "mft_common.h"
#include "mft_hip_testing.h"      // the form-selection hooks defined at the end of this file (tests / A-B tools only)

int mft_stem_conv_dispatch(const float* in, const float* w, float* out, int n_img, int H, int W, int w_ld,
                           hipStream_t s);   // csrc/stem.hip
int mft_skinny_fwd_dispatch(const float* in, int ldi, const float* w, float* out, int ldo, int n_img, int H, int W,
                            int Cin, int Cout, int KH, int KW, int stride, int pad, int imgs_per_group,
                            long long w_group_stride, hipStream_t s);   // csrc/skinny.hip
void mft_skinny_set_dgrad_slices(int n);
void mft_skinny_set_tap(int v);
void mft_skinny_set_x3(int v);
void mft_skinny_set_nw(int v);
void mft_bn_small_set_rows(int rows);      // csrc/backward.hip / csrc/bn.hip: one-launch BatchNorm forms for small problems
void mft_bn_fwd_small_set_rows(int rows);
void mft_skinny_set_lines(int v);
int mft_skinny_dgrad_dispatch(const float* dy, int ldy, const float* w, float* dx, int ldx, int n_img, int H, int W,
                              int Cin, int Cout, int KH, int KW, int stride, int pad, int imgs_per_group,
                              long long w_group_stride, hipStream_t s);

namespace {

struct ConvArgs {
    const float* in;
    const float* w;
    const float* bias;
    float* out;
    int ldi, ldo;
    int H, W, Cin, OH, OW, Cout, KH, KW, stride, pad;
    int Kpad;            // weight row length in floats (multiple of 32)
    int Ktot;            // KH*KW*Cin
    int rows_per_group;  // imgs_per_group * OH * OW
    int imgs_per_group;
    int tiles_n;
    long long wgs;       // weight group stride (floats), 0 = shared
    int bt_stride;       // BT (data-gradient) mode: stride of the forward convolution
    int ksplit;          // > 1: grid.z slices of the K walk, partial tiles to ws[z][rows][Cout] (summed in slice order afterwards)
    float* ws;
    int parity;          // BT with bt_stride == 2, one weight set: blockIdx.y = parity class (h & 1, w & 1) of the dx pixels of the tile
};

constexpr int BK = 32;
constexpr int LDS_LD = 36;
int g_dgrad_parity = 1; // stride-2 data gradients by pixel parity class (mft_debug_set_conv_tile(9800/9801))
int g_wgrad_tile = 64; // 64 (default: 17 KB LDS lets conv workgroups of the other stream co-reside) or 128
int g_conv_tile = 0;   // 0 = automatic; 1..5 force a tile (mft_debug_set_conv_tile, tuning only)
int g_wgrad_pol = 7;          // w/m/v cache policy: bit 0 nontemporal loads, bit 1 nontemporal stores; bit 2 (wgrad_adam_rows_kernel):
                              // hardware v_rcp_f32 / v_sqrt_f32 + packed fp32 moment updates (mft_debug_set_conv_tile(9000 + pol))
int g_wgrad_trim = 1;         // 1: no matrix instructions for the zero rows beyond rows_per_group (mft_debug_set_conv_tile(9600/9601))
int g_wgrad_rows = 1;         // 1: <= 64 reduction rows use the 32 x 128 stream-shaped kernel (mft_debug_set_conv_tile(9500/9501))
int g_wgrad_early = 1;        // 1: issue the tile's w/m/v loads before the reduction (mft_debug_set_conv_tile(5000/5001))
int g_wgrad_min_lds_kb = 0;   // experiment: pad the fused wgrad+Adam workgroup's LDS to cap its occupancy (4000 + KB)
int g_skinny = 1;      // 0: per-episode-weight launches use the generic tiles (mft_debug_set_conv_tile(3000/3001))
int g_stem_fast = 1;   // 0: route the stem through the generic gather kernel (mft_debug_set_conv_tile(2000/2001))

// BT == true is the data-gradient form: dx[h][w][ci] = sum_{kh,kw,co} dy[(h+pad-kh)/s][(w+pad-kw)/s][co] *
// w[co][kh][kw][ci] (taps whose offset is not divisible by the stride contribute nothing).  The B operand is
// read straight from the *forward* weight pack as B[n=ci][k=(kh,kw,co)], so no per-step weight transpose.  Its B tile is staged
// k-major ([32 co][BN ci], coalesced 16-byte loads along ci) and the fragments are fetched with
// conflict-free ds_read_b32 using the same k = 16*h + t mapping as the A operand.
template <int BM, int BN, int WM, int WN, bool STEM, bool BT>
__global__ __launch_bounds__(256) void conv_igemm_kernel(ConvArgs p) {
    constexpr int TM = BM / WM / 32;
    constexpr int TN = BN / WN / 32;
    constexpr int PA = BM / 32;   // A passes: 32 rows per pass (8 threads x float4 per row)
    constexpr int PB = BN / 32;
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int r = lane & 31, h = lane >> 5;

    // XCD-aware tile order for single-group launches (see csrc/conv_x3.hip): workgroup ids are dealt round-robin to the 8
    // XCDs; give each XCD one contiguous range of tiles (n fastest) so that operand re-use stays inside one L2
    int tile_id = blockIdx.x;
    if (gridDim.y == 1) {
        const int nwg = gridDim.x, q = nwg >> 3, rmd = nwg & 7;
        const int xcd = tile_id & 7, slot = tile_id >> 3;
        tile_id = xcd * q + (xcd < rmd ? xcd : rmd) + slot;
    }
    const int nt = tile_id % p.tiles_n;
    const int mt = tile_id / p.tiles_n;
    // Data gradient of a stride-2 convolution: dx pixel (h, w) receives only the taps with kh = (h + pad) mod 2, kw = (w + pad) mod 2
    // (1, 2 or 4 of 9 for 3x3 / pad 1; 1 or 0 for the 1x1 shortcut) -- walking all nine with zero rows wastes 3/4 of the matrix work.
    // In parity mode a tile holds dx pixels of ONE class (ph, pw) = blockIdx.y, numbered (img, i, j) with h = 2i + ph, w = 2j + pw,
    // and its K walk covers that class's taps only.
    const bool par = BT && p.parity != 0;
    const int g = par ? 0 : blockIdx.y;
    const int ph = par ? (int)(blockIdx.y >> 1) : 0, pw = par ? (int)(blockIdx.y & 1) : 0;
    const int Hc = par ? (p.OH - ph + 1) / 2 : p.OH, Wc = par ? (p.OW - pw + 1) / 2 : p.OW;
    const int rows = par ? p.imgs_per_group * Hc * Wc : p.rows_per_group;      // rows of this launch slice (class or group)
    const int m0 = mt * BM, n0 = nt * BN;
    if (par && m0 >= rows) return;
    const int kh0 = par ? ((ph - p.pad) & 1) : 0, kw0 = par ? ((pw - p.pad) & 1) : 0;         // p.pad = -pad_fwd in BT mode
    const int nkh = par ? (p.KH - kh0 + 1) / 2 : p.KH, nkw = par ? (p.KW - kw0 + 1) / 2 : p.KW;

    const int lrow = tid >> 3;        // 0..31
    const int c4 = (tid & 7) * 4;     // 0..28
    const int ohw = Hc * Wc;

    // per-thread A row descriptors
    long long a_base[PA];
    int a_ih0[PA], a_iw0[PA];
    bool a_ok[PA];
#pragma unroll
    for (int j = 0; j < PA; ++j) {
        int m = m0 + lrow + 32 * j;
        a_ok[j] = m < rows;
        int mm = a_ok[j] ? m : 0;
        int img = mm / ohw;
        int rem = mm - img * ohw;
        int oh = rem / Wc, ow = rem - oh * Wc;
        if (par) { oh = 2 * oh + ph; ow = 2 * ow + pw; }
        a_ih0[j] = oh * p.stride - p.pad;
        a_iw0[j] = ow * p.stride - p.pad;
        a_base[j] = (long long)(g * p.imgs_per_group + img) * p.H * p.W;
    }
    const float* wg = p.w + (long long)g * p.wgs;
    bool b_ok[PB];
    const float* b_ptr[PB];
    constexpr int QB = BN / 4;            // BT: float4 per k-row of the B tile
    constexpr int RB = 256 / QB;          // BT: k-rows per pass
    const int bt_q = tid % QB, bt_k = tid / QB;
#pragma unroll
    for (int j = 0; j < PB; ++j) {
        if (!BT) {
            int n = n0 + lrow + 32 * j;
            b_ok[j] = n < p.Cout;
            b_ptr[j] = wg + (long long)(b_ok[j] ? n : 0) * p.Kpad + c4;
        } else {
            // row (co) stride of the forward pack = KH*KW*Cout_here (Cout_here = forward Cin)
            b_ok[j] = (n0 + 4 * bt_q) < p.Cout;
            b_ptr[j] = wg + (long long)(bt_k + j * RB) * ((long long)p.KH * p.KW * p.Cout) + n0 + 4 * bt_q;
        }
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    f32x4 ra[PA], rb[PB];
    const int nk_all = par ? nkh * nkw * (p.Cin / BK) : p.Kpad / BK;
    const int kt0 = p.ksplit > 1 ? (int)((long long)blockIdx.z * nk_all / p.ksplit) : 0;
    const int nk = p.ksplit > 1 ? (int)((long long)(blockIdx.z + 1) * nk_all / p.ksplit) : nk_all;

    auto load_tile = [&](int kt) {
        const int k0 = kt * BK;
        if (!STEM) {
            const int khkw = k0 / p.Cin;
            const int ci0 = k0 - khkw * p.Cin;
            int kh = khkw / nkw, kw = khkw - kh * nkw;
            if (par) { kh = kh0 + 2 * kh; kw = kw0 + 2 * kw; }
#pragma unroll
            for (int j = 0; j < PA; ++j) {
                int ih, iw;
                bool ok;
                if (!BT) {
                    ih = a_ih0[j] + kh; iw = a_iw0[j] + kw;
                    ok = a_ok[j] && ih >= 0 && ih < p.H && iw >= 0 && iw < p.W;
                } else {
                    // dx pixel (h,w) receives dy[(h+pad-kh)/s][(w+pad-kw)/s] * w[co][kh][kw][ci] when divisible
                    const int th = a_ih0[j] - kh, tw = a_iw0[j] - kw;        // a_ih0 = h + pad_fwd
                    ih = th / p.bt_stride; iw = tw / p.bt_stride;
                    ok = a_ok[j] && th >= 0 && tw >= 0 && ih * p.bt_stride == th && iw * p.bt_stride == tw &&
                         ih < p.H && iw < p.W;
                }
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (ok) v = *(const f32x4*)(p.in + (a_base[j] + (long long)ih * p.W + iw) * p.ldi + ci0 + c4);
                ra[j] = v;
            }
        } else {
            // stem: Cin == 3, k = (kh*KW + kw)*3 + ci, scalar gather (K = 147 -> 5 K-steps)
#pragma unroll
            for (int j = 0; j < PA; ++j) {
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    int k = k0 + c4 + e;
                    if (a_ok[j] && k < p.Ktot) {
                        int khkw = k / 3, ci = k - khkw * 3;
                        int kh = khkw / p.KW, kw = khkw - kh * p.KW;
                        int ih = a_ih0[j] + kh, iw = a_iw0[j] + kw;
                        if (ih >= 0 && ih < p.H && iw >= 0 && iw < p.W)
                            v[e] = p.in[(a_base[j] + (long long)ih * p.W + iw) * p.ldi + ci];
                    }
                }
                ra[j] = v;
            }
        }
        if (!BT) {
#pragma unroll
            for (int j = 0; j < PB; ++j) {
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (b_ok[j]) v = *(const f32x4*)(b_ptr[j] + k0);
                rb[j] = v;
            }
        } else {
            int khkw = k0 / p.Cin;                       // p.Cin = forward Cout (reduction channels)
            const int co0 = k0 - khkw * p.Cin;
            if (par) { const int a = khkw / nkw; khkw = (kh0 + 2 * a) * p.KW + kw0 + 2 * (khkw - a * nkw); }
            const long long off = (long long)co0 * ((long long)p.KH * p.KW * p.Cout) + (long long)khkw * p.Cout;
#pragma unroll
            for (int j = 0; j < PB; ++j) {
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (b_ok[j]) v = *(const f32x4*)(b_ptr[j] + off);
                rb[j] = v;
            }
        }
    };
    auto store_tile = [&](int buf) {
        float* As = smem + buf * (BM + BN) * LDS_LD;
        float* Bs = As + BM * LDS_LD;
#pragma unroll
        for (int j = 0; j < PA; ++j) *(f32x4*)(As + (lrow + 32 * j) * LDS_LD + c4) = ra[j];
        if (!BT) {
#pragma unroll
            for (int j = 0; j < PB; ++j) *(f32x4*)(Bs + (lrow + 32 * j) * LDS_LD + c4) = rb[j];
        } else {
#pragma unroll
            for (int j = 0; j < PB; ++j) *(f32x4*)(Bs + (bt_k + j * RB) * BN + 4 * bt_q) = rb[j];
        }
    };

    if (kt0 < nk) {
        load_tile(kt0);
        store_tile(0);
    }
    __syncthreads();

    for (int kt = kt0; kt < nk; ++kt) {
        const int buf = (kt - kt0) & 1;
        if (kt + 1 < nk) load_tile(kt + 1);
        const float* As = smem + buf * (BM + BN) * LDS_LD;
        const float* Bs = As + BM * LDS_LD;
        f32x4 av[TM][4], bv[TN][4];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const float* ptr = As + (wm * (BM / WM) + i * 32 + r) * LDS_LD + h * 16;
#pragma unroll
            for (int q = 0; q < 4; ++q) av[i][q] = *(const f32x4*)(ptr + 4 * q);
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            if (!BT) {
                const float* ptr = Bs + (wn * (BN / WN) + j * 32 + r) * LDS_LD + h * 16;
#pragma unroll
                for (int q = 0; q < 4; ++q) bv[j][q] = *(const f32x4*)(ptr + 4 * q);
            } else {
                const float* ptr = Bs + (h * 16) * BN + wn * (BN / WN) + j * 32 + r;
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int e = 0; e < 4; ++e) bv[j][q][e] = ptr[(4 * q + e) * BN];
            }
        }
#pragma unroll
        for (int t = 0; t < 16; ++t)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i][t >> 2][t & 3], bv[j][t >> 2][t & 3],
                                                                     acc[i][j], 0, 0, 0);
        if (kt + 1 < nk) store_tile(buf ^ 1);
        __syncthreads();
    }

    // epilogue: C/D layout of 32x32 MFMA: col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5)
    const long long out_row0 = (long long)g * p.rows_per_group;
    if (p.ksplit > 1) {                    // partial tile of this K slice: ws[slice][all groups' rows][Cout]
        // (parity mode: one group, grid.y = class; a class's rows land at their dx pixel's row, so every row of a slice is written by
        // exactly one class -- classes without taps write zeros -- and the slice sum below needs no parity logic)
        float* part = p.ws + ((long long)blockIdx.z * (par ? 1 : (int)gridDim.y) * p.rows_per_group + out_row0) * p.Cout;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = n0 + wn * (BN / WN) + j * 32 + r;
                if (n >= p.Cout) continue;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int m = m0 + wm * (BM / WM) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                    if (m >= rows) continue;
                    long long orow = m;
                    if (par) {
                        const int img = m / ohw, rem = m - img * ohw;
                        const int ii = rem / Wc, jj = rem - ii * Wc;
                        orow = ((long long)img * p.OH + 2 * ii + ph) * p.OW + 2 * jj + pw;
                    }
                    part[orow * p.Cout + n] = acc[i][j][e];
                }
            }
        return;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn * (BN / WN) + j * 32 + r;
            if (n >= p.Cout) continue;
            const float bias = p.bias ? p.bias[n] : 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = (e & 3) + 8 * (e >> 2) + 4 * h;
                const int m = m0 + wm * (BM / WM) + i * 32 + row;
                if (m >= rows) continue;
                long long orow = out_row0 + m;
                if (par) {                                   // class-local (img, i, j) -> dx pixel (img, 2i + ph, 2j + pw)
                    const int img = m / ohw, rem = m - img * ohw;
                    const int ii = rem / Wc, jj = rem - ii * Wc;
                    orow = ((long long)img * p.OH + 2 * ii + ph) * p.OW + 2 * jj + pw;
                }
                p.out[orow * p.ldo + n] = acc[i][j][e] + bias;
            }
        }
}

// out[m][n] = sum over the K slices (in slice order: deterministic) of the partial tiles + bias
__global__ __launch_bounds__(256) void reduce_ksplit_kernel(const float* __restrict__ ws, float* __restrict__ out,
                                                            const float* __restrict__ bias, int rows, int C, int ldo, int splits) {
    const int cq = C >> 2;
    const long long total = (long long)rows * cq;
    const long long slice = (long long)rows * C;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long m = i / cq;
        const int c = (int)(i - m * cq) * 4;
        f32x4 s = *(const f32x4*)(ws + m * C + c);
#pragma unroll 4
        for (int z = 1; z < splits; ++z) s += *(const f32x4*)(ws + z * slice + m * C + c);
        if (bias) s += *(const f32x4*)(bias + c);
        *(f32x4*)(out + m * ldo + c) = s;
    }
}

// K slices for a single-group launch of `rows` x `cols` outputs with a reduction of K: a meta-training step runs ONE episode
// (105 images), so the deep layers are a handful of 64 x 64 tiles with 72-144 K-steps each (trunk.7: 120 tiles on 256 CUs,
// 70 us of dependent MFMAs per tile); slicing K puts ~3 workgroups on every CU.  1 = no slicing.
inline int conv_ksplit(long long rows, int cols, int K, int groups = 1) {
    if (cols % 4 != 0) return 1;
    const long long tiles = (long long)groups * ((rows + 63) / 64) * ((cols + 63) / 64);      // rows: per group
    const int nk = K / BK;
    if (tiles >= 512 || nk < 16) return 1;
    long long s = (768 + tiles - 1) / tiles;
    if (s > nk / 8) s = nk / 8;
    if (s > 16) s = 16;
    return s < 2 ? 1 : (int)s;
}

template <int BM, int BN, int WM, int WN, bool STEM, bool BT = false>
int launch_conv(const ConvArgs& a, int groups, hipStream_t s) {
    const int tiles_m = cdiv(a.rows_per_group, BM);
    ConvArgs p = a;
    p.tiles_n = cdiv(a.Cout, BN);
    const size_t lds = 2 * (BM + BN) * LDS_LD * sizeof(float);
    auto kern = conv_igemm_kernel<BM, BN, WM, WN, STEM, BT>;
    if (lds > 64 * 1024) {
        static MftPerDeviceOnce attr_once;
        if (attr_once.need()) {
            hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return (int)e;
            attr_once.mark();
        }
    }
    if (p.ksplit > 1 && p.ws == nullptr) return MFT_EINVAL;
    if (p.parity) {
        if (!BT || groups != 1 || p.bt_stride != 2) return MFT_EINVAL;
        const int tm = cdiv(p.imgs_per_group * ((p.OH + 1) / 2) * ((p.OW + 1) / 2), BM);       // the largest class
        hipLaunchKernelGGL(kern, dim3(tm * p.tiles_n, 4, p.ksplit > 1 ? p.ksplit : 1), dim3(256), lds, s, p);
    } else {
        dim3 grid(tiles_m * p.tiles_n, groups, p.ksplit > 1 ? p.ksplit : 1);
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, p);
    }
    if (p.ksplit > 1) {
        const long long total = (long long)groups * p.rows_per_group * (p.Cout / 4);
        long long blocks = (total + 255) / 256;
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(reduce_ksplit_kernel, dim3((unsigned)blocks), dim3(256), 0, s, (const float*)p.ws, p.out, p.bias,
                           groups * p.rows_per_group, p.Cout, p.ldo, p.ksplit);
    }
    return mft_launch_status();
}

// ----------------------------------------------------------------------------------------- wgrad
struct WgradArgs {
    const float* in;
    const float* dy;
    float* dw;           // gradient output (may be null when the Adam epilogue is used)
    int ldi, ldy;
    int H, W, Cin, OH, OW, Cout, KH, KW, stride, pad;
    int Kpad;
    int rows_per_group, imgs_per_group;
    int tiles_ci;        // Cin / BN
    int tiles_co;        // Cout / BM
    long long dwgs;
    // fused Adam epilogue (ADAM == true): the gradient tile never reaches HBM
    float* w; float* m; float* v;
    float step_size, inv_sqrt_bc2, b1, b2, eps;
    const float* hyper;  // optional device pointer {step_size, inv_sqrt_bc2}: lets a captured hipGraph replay with advancing steps
    // split-M: grid.z chunks of chunk_rows rows each write partial gradients to ws (reduced afterwards)
    int chunk_rows, chunks;
    float* ws;
    int oihw;            // plain weight gradient: partials always go through ws and the chunk sum writes torch's [Cout][Cin][KH][KW] layout
    int ws_inv_ow;       // wgrad_adam_rows_kernel: ceil(65536 / OW) (chunk_rows then holds ceil(65536 / (OH*OW)))
    int mma_rows;        // wgrad_adam_rows_kernel: reduction rows that get matrix instructions (rows_per_group, or 64 = the padded form)
    int cin_out, cout_out;   // oihw: the gradient handed back is the first cout_out x cin_out of the Cout x Cin computed (0 = all)
};

// dw[co][(kh,kw,ci)] = sum_m dy[m][co] * in[pix(m,kh,kw)][ci]; reduction index m is the slow memory
// index of both operands, so tiles are staged k-major ([m][co], [m][ci]) and the MFMA fragments are
// fetched with conflict-free ds_read_b32 (lane -> consecutive co / ci); MFMA step t uses m = 2t + h.
// ADAM == true applies torch.optim.Adam (finetune.py:255,299) in the epilogue: the gradient tile is parked
// in LDS ([BM][BN+4]) and the w/m/v update streams with 16 B per lane and 4*BN contiguous bytes per row:
// 3 reads + 3 writes per parameter instead of a gradient write plus Adam's 4 reads + 3 writes.
// STEMW: the 7x7x3 stem (Cin == 3): the "ci" axis of the tile is the flattened k = (kh*KW+kw)*3+ci (147 -> 160).
// POL: cache policy of the w/m/v stream (bit 0: nontemporal loads, bit 1: nontemporal stores); 3 is the default
template <int BM, int BN, bool ADAM, bool STEMW, bool EARLYT = false, int POL = 3>
__device__ __forceinline__ void conv_wgrad_tile(const WgradArgs& p, const int bid_x, const int bid_y, const int bid_z) {
    auto ldp = [](const f32x4* q) { return (POL & 1) ? __builtin_nontemporal_load(q) : *q; };
    auto stp = [](const f32x4 v, f32x4* q) { if (POL & 2) __builtin_nontemporal_store(v, q); else *q = v; };
    constexpr int TM = BM / 64, TN = BN / 64;   // waves 2 x 2, wave tile (BM/2) x (BN/2)
    constexpr int QA = BM / 4;                  // float4 per A row
    constexpr int QB = BN / 4;
    constexpr int PA = (32 * QA) / 256;         // passes over the 32-row A tile
    constexpr int PB = (32 * QB) / 256;
    constexpr int RPA = 256 / QA;               // rows per pass
    constexpr int RPB = 256 / QB;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                 // [32][BM]
    float* Bs = smem + 32 * BM;       // [32][BN]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int r = lane & 31, h = lane >> 5;
    const int g = bid_y;

    int bx = bid_x;
    const int tci = bx % p.tiles_ci;
    bx /= p.tiles_ci;
    const int tco = bx % p.tiles_co;
    const int khkw = bx / p.tiles_co;
    const int kh = khkw / p.KW, kw = khkw - kh * p.KW;
    const int co0 = tco * BM, ci0 = tci * BN;
    const int ohw = p.OH * p.OW;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int arow = tid / QA, acol = (tid % QA) * 4;
    const int brow = tid / QB, bcol = (tid % QB) * 4;
    const long long row0 = (long long)g * p.rows_per_group;
    const long long img0 = (long long)g * p.imgs_per_group;

    // ADAM, 64x64 tile: issue the tile's w/m/v loads (12 x 16 B per lane, nontemporal) BEFORE the reduction so that their
    // HBM latency is covered by the MFMA phase of this same workgroup instead of by other resident workgroups -- the kernel
    // then keeps its bandwidth when the co-running trunk stream takes CU slots away.
    constexpr bool EARLY = EARLYT && ADAM && BM == 64 && BN == 64;
    f32x4 e_m[4], e_v[4], e_w[4];
    long long e_gi[4];
    if (EARLY) {
        const int q = tid % 16, rr = tid / 16;
        const long long gbase = (long long)g * p.dwgs + (long long)khkw * p.Cin + ci0 + 4 * q;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            e_gi[u] = gbase + (long long)(co0 + rr + u * 16) * p.Kpad;
            e_m[u] = ldp((const f32x4*)(p.m + e_gi[u]));
            e_v[u] = ldp((const f32x4*)(p.v + e_gi[u]));
            e_w[u] = ldp((const f32x4*)(p.w + e_gi[u]));
        }
    }
    const int m_begin = bid_z * p.chunk_rows;
    const int m_end = min(m_begin + p.chunk_rows, p.rows_per_group);
    const bool a_col_ok = (co0 + acol) < p.Cout;
    const bool b_col_ok = STEMW ? true : (ci0 + bcol) < p.Cin;
    for (int mk = m_begin; mk < m_end; mk += 32) {
        f32x4 va[PA], vb[PB];
#pragma unroll
        for (int j = 0; j < PA; ++j) {
            int m = mk + arow + j * RPA;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (m < m_end && a_col_ok) v = *(const f32x4*)(p.dy + (row0 + m) * p.ldy + co0 + acol);
            va[j] = v;
        }
#pragma unroll
        for (int j = 0; j < PB; ++j) {
            int m = mk + brow + j * RPB;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (m < m_end && b_col_ok) {
                int img = m / ohw;
                int rem = m - img * ohw;
                int oh = rem / p.OW, ow = rem - oh * p.OW;
                if (!STEMW) {
                    int ih = oh * p.stride - p.pad + kh, iw = ow * p.stride - p.pad + kw;
                    if (ih >= 0 && ih < p.H && iw >= 0 && iw < p.W)
                        v = *(const f32x4*)(p.in + ((img0 + img) * p.H * p.W + (long long)ih * p.W + iw) * p.ldi + ci0 + bcol);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int k = ci0 + bcol + e;
                        if (k < p.KH * p.KW * 3) {
                            const int tap = k / 3, ci = k - tap * 3;
                            const int skh = tap / p.KW, skw = tap - skh * p.KW;
                            const int ih = oh * p.stride - p.pad + skh, iw = ow * p.stride - p.pad + skw;
                            if (ih >= 0 && ih < p.H && iw >= 0 && iw < p.W)
                                v[e] = p.in[((img0 + img) * p.H * p.W + (long long)ih * p.W + iw) * p.ldi + ci];
                        }
                    }
                }
            }
            vb[j] = v;
        }
        __syncthreads();   // previous iteration's fragment reads are done
#pragma unroll
        for (int j = 0; j < PA; ++j) *(f32x4*)(As + (arow + j * RPA) * BM + acol) = va[j];
#pragma unroll
        for (int j = 0; j < PB; ++j) *(f32x4*)(Bs + (brow + j * RPB) * BN + bcol) = vb[j];
        __syncthreads();
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            float a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = As[(2 * t + h) * BM + wm * (BM / 2) + i * 32 + r];
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = Bs[(2 * t + h) * BN + wn * (BN / 2) + j * 32 + r];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    }
    if (!ADAM) {
        float* dwg = (p.chunks > 1 || p.oihw) ? p.ws + ((long long)g * p.chunks + bid_z) * p.dwgs
                                              : p.dw + (long long)g * p.dwgs;
        const int ci_lim = STEMW ? p.Kpad : p.Cin;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int ci = ci0 + wn * (BN / 2) + j * 32 + r;
                if (ci >= ci_lim) continue;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int row = (e & 3) + 8 * (e >> 2) + 4 * h;
                    const int co = co0 + wm * (BM / 2) + i * 32 + row;
                    if (co < p.Cout) dwg[(long long)co * p.Kpad + (long long)khkw * p.Cin + ci] = acc[i][j][e];
                }
            }
    } else {
        constexpr int GLD = BN + 4;
        float* Gs = smem;             // aliases As/Bs
        __syncthreads();              // all fragment reads of As/Bs are done
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int row = (e & 3) + 8 * (e >> 2) + 4 * h;
                    Gs[(wm * (BM / 2) + i * 32 + row) * GLD + wn * (BN / 2) + j * 32 + r] = acc[i][j][e];
                }
        __syncthreads();
        constexpr int Q = BN / 4;             // float4 per row
        constexpr int RP = 256 / Q;           // rows per pass
        const int q = tid % Q, rr = tid / Q;
        const long long gbase = (long long)g * p.dwgs + (long long)khkw * p.Cin + ci0 + 4 * q;
        // w, m, v are touched exactly once per step and the slabs (GBs) dwarf every cache: stream them with
        // nontemporal loads/stores, four rows (12 x 16 B per lane) in flight before the first use.
        constexpr int NR = BM / RP;           // rows per thread
        static_assert(NR % 4 == 0, "row blocking");
        const float step_size = p.hyper ? p.hyper[0] : p.step_size;
        const float inv_sqrt_bc2 = p.hyper ? p.hyper[1] : p.inv_sqrt_bc2;
#pragma unroll
        for (int r0 = 0; r0 < NR; r0 += 4) {
            f32x4 mm[4], vv[4], ww[4];
            long long gi[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int row = rr + (r0 + u) * RP;
                if (EARLY) {
                    gi[u] = e_gi[u]; mm[u] = e_m[u]; vv[u] = e_v[u]; ww[u] = e_w[u];
                } else {
                    gi[u] = gbase + (long long)(co0 + row) * p.Kpad;
                    mm[u] = ldp((const f32x4*)(p.m + gi[u]));
                    vv[u] = ldp((const f32x4*)(p.v + gi[u]));
                    ww[u] = ldp((const f32x4*)(p.w + gi[u]));
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int row = rr + (r0 + u) * RP;
                const f32x4 ge = *(const f32x4*)(Gs + row * GLD + 4 * q);
                mft_adam4_exact(mm[u], vv[u], ww[u], ge, p.b1, p.b2, p.eps, step_size, inv_sqrt_bc2);
                stp(mm[u], (f32x4*)(p.m + gi[u]));
                stp(vv[u], (f32x4*)(p.v + gi[u]));
                stp(ww[u], (f32x4*)(p.w + gi[u]));
                if (p.dw) *(f32x4*)(p.dw + gi[u]) = ge;
            }
        }
    }
}

template <int BM, int BN, bool ADAM, bool STEMW, bool EARLYT = false, int POL = 3>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(WgradArgs p) {
    conv_wgrad_tile<BM, BN, ADAM, STEMW, EARLYT, POL>(p, blockIdx.x, blockIdx.y, blockIdx.z);
}

// Several independent weight-gradient problems in ONE launch (mft_conv2d_wgrad_oihw_multi): the meta-training backward defers every
// layer's weight gradient to the end of its pass (nothing downstream reads them) and runs them together -- 31 launches + 31 partial
// sums per step become 2 + 2 (a dependent kernel costs >= 4.7 us in a replayed step whatever it does).  Workgroup b belongs to the
// job j with start[j] <= b < start[j + 1]; inside a job the numbering is the single launch's (x = tile and tap, z = row chunk), so
// every job computes bit for bit what its own launch would.
constexpr int WG_MULTI = 16;
struct WgradMultiArgs {
    WgradArgs job[WG_MULTI];
    int start[WG_MULTI + 1];
    int nx[WG_MULTI];          // grid.x of the job's own launch
    int n;
};

static_assert(sizeof(WgradMultiArgs) <= 4000, "kernel arguments are passed by value: keep them inside the 4 KB kernarg segment");

__global__ __launch_bounds__(256) void conv_wgrad_multi_kernel(WgradMultiArgs a) {
    const int b = blockIdx.x;
    int j = 0;
#pragma unroll 1
    while (j + 1 < a.n && b >= a.start[j + 1]) ++j;
    const WgradArgs p = a.job[j];
    const int local = b - a.start[j];
    conv_wgrad_tile<64, 64, false, false, false, 3>(p, local % a.nx[j], 0, local / a.nx[j]);
}

// ---------------------------------------------------------------------------- weight gradient + Adam, <= 64 reduction rows
// The inner-loop regime (finetune.py:286-299): per episode the weight gradient is a reduction over 5 images x 3 x 3 = 45 pixel
// rows, i.e. the launch is a pure w/m/v stream (6 x 4 B per parameter) with a sliver of matrix work.  Measured on this part
// (tools/microbench/adam_tiles.hip): a 3-read/3-write stream walked in 64 x 64-float tiles (256-byte runs, the kernel above)
// tops out at 5.9 TB/s, in 32 x 128-float tiles (512-byte runs) at 6.3 TB/s -- DRAM-page locality, not bytes in flight.  This
// kernel is the 32 (co) x 128 (ci) form, shaped for the stream:
//   * every operand row (<= 64 rows of dy and of the im2col slice) is requested FIRST, then the tile's w/m/v (12 x 16 B per
//     lane, nontemporal): the in-order load counter lets the reduction start as soon as the operands are in while the 48 KB of
//     w/m/v per workgroup stay in flight behind them;
//   * one wave per 32 x 32 block (v_mfma_f32_32x32x2_f32, reduction order m = 2t + h exactly as conv_wgrad_kernel: the
//     gradient and therefore Adam's result are bit-identical to it);
//   * the gradient tile goes through LDS once ([32][132]) and Adam streams rows of 512 contiguous bytes.
template <int POL>
__global__ __launch_bounds__(256) void wgrad_adam_rows_kernel(WgradArgs p) {
    constexpr int BM = 32, BN = 128, BLD = BN + 32, GLD = BN + 4;
    auto ldp = [](const f32x4* q) { return (POL & 1) ? __builtin_nontemporal_load(q) : *q; };
    auto stp = [](const f32x4 v, f32x4* q) { if (POL & 2) __builtin_nontemporal_store(v, q); else *q = v; };
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                 // [32][BM]   one half of the reduction rows at a time
    float* Bs = smem + 32 * BM;       // [32][BLD]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int g = blockIdx.y;
    // tile order inside an episode: the whole K extent of 32 output-channel rows (ci-tile, then tap: 18 KB contiguous per row
    // for trunk.7.C2) before the next 32 rows -- consecutive workgroups then walk whole DRAM pages of w, m and v
    int bx = blockIdx.x;
    int tci, tco, khkw;
    if (p.chunks == 1) {                                     // (p.chunks doubles as the order switch: 1 = row-major walk)
        tci = bx % p.tiles_ci; bx /= p.tiles_ci;
        khkw = bx % (p.KH * p.KW);
        tco = bx / (p.KH * p.KW);
    } else {
        tci = bx % p.tiles_ci; bx /= p.tiles_ci;
        tco = bx % p.tiles_co;
        khkw = bx / p.tiles_co;
    }
    const int kh = khkw / p.KW, kw = khkw - kh * p.KW;
    const int co0 = tco * BM, ci0 = tci * BN;
    const int ohw = p.OH * p.OW;
    const int rows = p.rows_per_group;                       // <= 64
    const long long row0 = (long long)g * rows;
    const long long img0 = (long long)g * p.imgs_per_group;

    // operand requests: A = dy[m][co0 .. co0+32) (8 float4 per row, 32 rows per pass), B = im2col[m][ci0 .. ci0+128) (8 rows per pass)
    f32x4 va[2], vb[8];
    const int arow = tid >> 3, acol = (tid & 7) * 4;
    const int brow = tid >> 5, bcol = (tid & 31) * 4;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int m = arow + 32 * j;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (m < rows) v = *(const f32x4*)(p.dy + (row0 + m) * p.ldy + co0 + acol);
        va[j] = v;
    }
    // m -> (image, oh, ow) by multiplication with host-made reciprocals (exact for m < 64): the compiler's general 32-bit
    // division is ~30 VALU instructions each, 16 of them per lane here -- more issue slots than the whole Adam epilogue
    const float* in_g = p.in + (img0 * p.H * p.W) * p.ldi + ci0 + bcol;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int m = brow + 8 * j;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (m < rows) {
            const int img = (m * p.chunk_rows) >> 16, rem = m - img * ohw;          // chunk_rows / tiles_co carry the reciprocals
            const int oh = (rem * p.ws_inv_ow) >> 16, ow = rem - oh * p.OW;
            const int ih = oh * p.stride - p.pad + kh, iw = ow * p.stride - p.pad + kw;
            if (ih >= 0 && ih < p.H && iw >= 0 && iw < p.W)
                v = *(const f32x4*)(in_g + ((img * p.H + ih) * p.W + iw) * p.ldi);
        }
        vb[j] = v;
    }
    // w/m/v requests: 32 float4 per row, 8 rows per pass, 4 passes
    const int q = tid & 31, rr = tid >> 5;
    f32x4 mm[4], vv[4], ww[4];
    long long gi[4];
    const long long gbase = (long long)g * p.dwgs + (long long)(co0 + rr) * p.Kpad + khkw * p.Cin + ci0 + 4 * q;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        gi[u] = gbase + (long long)(8 * u * p.Kpad);
        mm[u] = ldp((const f32x4*)(p.m + gi[u]));
        vv[u] = ldp((const f32x4*)(p.v + gi[u]));
        ww[u] = ldp((const f32x4*)(p.w + gi[u]));
    }
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        if (half * 32 < rows) {                               // wave-uniform
            if (half) __syncthreads();                        // fragment reads of the first half are done
            *(f32x4*)(As + arow * BM + acol) = va[half];
#pragma unroll
            for (int j = 0; j < 4; ++j) *(f32x4*)(Bs + (brow + 8 * j) * BLD + bcol) = vb[4 * half + j];
            __syncthreads();
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                // rows beyond the group's (45 of 64 in the inner loop) are zeros: skipping their MFMAs changes no bit of the sum
                // and saves 9 of 32 matrix instructions -- the launch is power-limited together with the trunk stream
                if (32 * half + 2 * t < p.mma_rows) {            // wave-uniform
                    const float a = As[(2 * t + h) * BM + r];
                    const float b = Bs[(2 * t + h) * BLD + wave * 32 + r];
#ifdef MFT_EXPERIMENTS
                    if constexpr (POL & 8) acc[t] += a + b;      // measurement aid (power / time without the matrix work)
                    else
#endif
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
                }
            }
        }
    }
    float* Gs = smem;                 // [32][GLD], aliases As/Bs
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 16; ++e) Gs[((e & 3) + 8 * (e >> 2) + 4 * h) * GLD + wave * 32 + r] = acc[e];
    __syncthreads();
    const float step_size = p.hyper ? p.hyper[0] : p.step_size;
    const float inv_sqrt_bc2 = p.hyper ? p.hyper[1] : p.inv_sqrt_bc2;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const f32x4 ge = *(const f32x4*)(Gs + (rr + 8 * u) * GLD + 4 * q);
#ifdef MFT_EXPERIMENTS
        if constexpr ((POL & 16) != 0) {                         // measurement aid: no Adam arithmetic, the stream only
            mm[u] += ge; vv[u] += ge; ww[u] += ge;
        } else
#endif
        if (POL & 4) {
            // hardware v_sqrt_f32 / v_rcp_f32 (1 ulp each) instead of the correctly rounded sequences, and the moment updates as
            // packed fp32 operations (v_pk_mul_f32 / v_pk_fma_f32): the epilogue's VALU work competes with the co-running trunk
            // convolutions for issue slots (A/B in DESIGN.md: +2 % end to end)
            mft_adam4_fast(mm[u], vv[u], ww[u], ge, p.b1, p.b2, p.eps, step_size, inv_sqrt_bc2);
        } else {
            mft_adam4_exact(mm[u], vv[u], ww[u], ge, p.b1, p.b2, p.eps, step_size, inv_sqrt_bc2);
        }
        stp(mm[u], (f32x4*)(p.m + gi[u]));
        stp(vv[u], (f32x4*)(p.v + gi[u]));
        stp(ww[u], (f32x4*)(p.w + gi[u]));
        if (p.dw) *(f32x4*)(p.dw + gi[u]) = ge;
    }
}

#ifdef MFT_EXPERIMENTS      // walking forms of the kernel above: measured no faster; built only for tools/
#include "../../tools/experiments/wgrad_adam_walk.inc"
#endif  // MFT_EXPERIMENTS

int launch_wgrad_adam_rows(const WgradArgs& a, int taps, int groups, hipStream_t s) {
    WgradArgs p = a;
    p.tiles_ci = a.Cin / 128;
    p.tiles_co = a.Cout / 32;
    constexpr int lds = (32 * 32 + 32 * (128 + 32)) * 4;       // 24.6 KB (the gradient tile [32][132] aliases it)
    p.chunks = g_wgrad_rows == 2 ? 2 : 1;
    p.chunk_rows = (65536 + a.OH * a.OW - 1) / (a.OH * a.OW);
    p.ws_inv_ow = (65536 + a.OW - 1) / a.OW;
    p.mma_rows = g_wgrad_trim ? a.rows_per_group : 64;
#ifdef MFT_EXPERIMENTS
    if (g_wgrad_rows == 5 && p.tiles_ci * taps >= 16) {      // (a 1x1 layer has too few K tiles to fill the CUs with walkers)
        // rows resident in LDS: 48 when the group has <= 48 reduction rows (the inner loop's 45): 47.6 KB = three workgroups per CU
        const int RP = a.rows_per_group <= 48 ? 48 : 64;
        const int lds_co = (RP * (128 + 32) + (RP * 32 > 32 * (128 + 4) ? RP * 32 : 32 * (128 + 4))) * 4;
        p.chunks = RP;
        dim3 grid_co(p.tiles_ci * taps, groups, 1);
        if (g_wgrad_pol == 3) hipLaunchKernelGGL(wgrad_adam_cowalk_kernel<3>, grid_co, dim3(256), lds_co, s, p);
        else hipLaunchKernelGGL(wgrad_adam_cowalk_kernel<7>, grid_co, dim3(256), lds_co, s, p);
        return mft_launch_status();
    }
    if (g_wgrad_rows == 4) {
        constexpr int lds_walk = (64 * 32 + 32 * (128 + 32) + 32 * (128 + 4)) * 4;      // 45.6 KB
        hipLaunchKernelGGL(wgrad_adam_walk_kernel<7>, dim3(p.tiles_co, groups, 1), dim3(256), lds_walk, s, p);
        return mft_launch_status();
    }
#endif
    dim3 grid(p.tiles_ci * p.tiles_co * taps, groups, 1);
    if (g_wgrad_pol == 7) hipLaunchKernelGGL(wgrad_adam_rows_kernel<7>, grid, dim3(256), lds, s, p);
#ifdef MFT_EXPERIMENTS
    else if (g_wgrad_pol == 15) hipLaunchKernelGGL(wgrad_adam_rows_kernel<15>, grid, dim3(256), lds, s, p);     // timing / power aids
    else if (g_wgrad_pol == 23) hipLaunchKernelGGL(wgrad_adam_rows_kernel<23>, grid, dim3(256), lds, s, p);
    else if (g_wgrad_pol == 31) hipLaunchKernelGGL(wgrad_adam_rows_kernel<31>, grid, dim3(256), lds, s, p);
    else if (g_wgrad_pol == 6) hipLaunchKernelGGL(wgrad_adam_rows_kernel<6>, grid, dim3(256), lds, s, p);
    else if (g_wgrad_pol == 5) hipLaunchKernelGGL(wgrad_adam_rows_kernel<5>, grid, dim3(256), lds, s, p);
    else if (g_wgrad_pol == 4) hipLaunchKernelGGL(wgrad_adam_rows_kernel<4>, grid, dim3(256), lds, s, p);
#endif
    else if (g_wgrad_pol == 3) hipLaunchKernelGGL(wgrad_adam_rows_kernel<3>, grid, dim3(256), lds, s, p);
#ifdef MFT_EXPERIMENTS
    else if (g_wgrad_pol == 2) hipLaunchKernelGGL(wgrad_adam_rows_kernel<2>, grid, dim3(256), lds, s, p);
    else if (g_wgrad_pol == 1) hipLaunchKernelGGL(wgrad_adam_rows_kernel<1>, grid, dim3(256), lds, s, p);
    else hipLaunchKernelGGL(wgrad_adam_rows_kernel<0>, grid, dim3(256), lds, s, p);
#else
    else hipLaunchKernelGGL(wgrad_adam_rows_kernel<3>, grid, dim3(256), lds, s, p);       // (any other policy code: the exact epilogue)
#endif
    return mft_launch_status();
}

// Sum of the split-M partial gradients.  Eight lanes per group of four consecutive elements: lane l adds chunks l, l+8, ... (float4
// loads), the eight partial sums are combined in lane order through LDS -- a fixed summation order with chains of chunks/8
// dependent loads (one thread per element walked all chunks serially: 14-60 us per call for 30-250 chunks).
template <bool OIHW>
__global__ __launch_bounds__(256) void reduce_chunks_kernel(const float* __restrict__ ws, float* __restrict__ dw, long long n, int chunks,
                                                            long long dwgs, int Cin, int taps, int Kpad, int cin_out, int cout_out) {
    __shared__ f32x4 red[8][33];
    const int g = blockIdx.y;
    const int q = threadIdx.x & 31, l = threadIdx.x >> 5;
    const long long nq = n >> 2;                                   // n = Cout * Kpad, Kpad % 32 == 0
    for (long long i0 = (long long)blockIdx.x * 32; i0 < nq; i0 += (long long)gridDim.x * 32) {
        const long long i = i0 + q;
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        if (i < nq)
#pragma unroll 4
            for (int c = l; c < chunks; c += 8) s += *(const f32x4*)(ws + ((long long)g * chunks + c) * dwgs + i * 4);
        red[l][q] = s;
        __syncthreads();
        if (l == 0 && i < nq) {
            f32x4 t = red[0][q];
#pragma unroll
            for (int j = 1; j < 8; ++j) t += red[j][q];
            if constexpr (!OIHW) {
                *(f32x4*)(dw + (long long)g * dwgs + i * 4) = t;
            } else {                                               // packed [Cout][(tap, ci)] -> torch's [Cout][Cin][KH][KW]
                const long long e0 = i * 4;                        // (the first cout_out x cin_out of it: zero-padded operand rows / columns)
                const int co = (int)(e0 / Kpad), k0 = (int)(e0 - (long long)co * Kpad);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int k = k0 + e;
                    if (k < taps * Cin && co < cout_out) {
                        const int tap = k / Cin, ci = k - tap * Cin;
                        if (ci < cin_out) dw[((long long)co * cin_out + ci) * taps + tap] = t[e];
                    }
                }
            }
        }
        __syncthreads();
    }
}

struct ReduceMultiArgs {
    const float* ws[WG_MULTI]; float* dw[WG_MULTI];
    long long n[WG_MULTI], dwgs[WG_MULTI];
    int chunks[WG_MULTI], Cin[WG_MULTI], taps[WG_MULTI], Kpad[WG_MULTI], cin_out[WG_MULTI], cout_out[WG_MULTI];
    int start[WG_MULTI + 1];
    int cnt;
};

static_assert(sizeof(ReduceMultiArgs) <= 4000, "kernarg segment");

// reduce_chunks_kernel<true> of several jobs in one launch (same lanes, same summation order per element as the single launch)
__global__ __launch_bounds__(256) void reduce_chunks_multi_kernel(ReduceMultiArgs a) {
    __shared__ f32x4 red[8][33];
    int j = 0;
#pragma unroll 1
    while (j + 1 < a.cnt && (int)blockIdx.x >= a.start[j + 1]) ++j;
    const float* __restrict__ ws = a.ws[j];
    float* __restrict__ dw = a.dw[j];
    const long long n = a.n[j], dwgs = a.dwgs[j];
    const int chunks = a.chunks[j], Cin = a.Cin[j], taps = a.taps[j], Kpad = a.Kpad[j], cin_out = a.cin_out[j], cout_out = a.cout_out[j];
    const int lb = blockIdx.x - a.start[j], nb = a.start[j + 1] - a.start[j];
    const int q = threadIdx.x & 31, l = threadIdx.x >> 5;
    const long long nq = n >> 2;
    if (chunks <= 8 && taps == 9 && cin_out == Cin && (Cin & 3) == 0 && Kpad == 9 * Cin && ((unsigned long long)dw & 15) == 0) {
        // Few chunks, 3x3 layer (the deep trunk layers: 2-6 chunks, 88 % of the trunk's weights).  One thread per (output channel,
        // four input channels): the nine taps of its four channels are 36 CONSECUTIVE floats of torch's [Cout][Cin][3][3] -- nine
        // 16-byte stores per thread, whole cache lines per wave (the element-wise form below writes every float on its own, 36
        // bytes from its neighbour's).  Chunks added in order: the sum the lane form gives for <= 8 chunks.
        const int cq = Cin >> 2;
        const long long groups4 = (long long)cout_out * cq;
        for (long long i = (long long)lb * 256 + threadIdx.x; i < groups4; i += (long long)nb * 256) {
            const int co = (int)(i / cq), c4 = (int)(i - (long long)co * cq) * 4;
            const float* src = ws + (long long)co * Kpad + c4;
            f32x4 t[9];
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) t[tap] = *(const f32x4*)(src + tap * Cin);
            for (int c = 1; c < chunks; ++c) {
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) t[tap] += *(const f32x4*)(src + (long long)c * dwgs + tap * Cin);
            }
            float o[36];
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) o[e * 9 + tap] = t[tap][e];
            float* dst = dw + ((long long)co * Cin + c4) * 9;
#pragma unroll
            for (int v = 0; v < 9; ++v) {
                const f32x4 w4 = {o[4 * v], o[4 * v + 1], o[4 * v + 2], o[4 * v + 3]};
                *(f32x4*)(dst + 4 * v) = w4;
            }
        }
        return;
    }
    if (chunks <= 8) {
        // Few chunks, any other shape: with eight lanes per element group most lanes would hold no chunk at all.  One thread per
        // group of four elements, chunks added in order -- the same sum as the lane form produces for <= 8 chunks (lane l holds
        // chunk l alone; the lanes are combined in lane order).
        for (long long i = (long long)lb * 256 + threadIdx.x; i < nq; i += (long long)nb * 256) {
            f32x4 t = *(const f32x4*)(ws + i * 4);
            for (int c = 1; c < chunks; ++c) t += *(const f32x4*)(ws + (long long)c * dwgs + i * 4);
            const long long e0 = i * 4;
            const int co = (int)(e0 / Kpad), k0 = (int)(e0 - (long long)co * Kpad);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int k = k0 + e;
                if (k < taps * Cin && co < cout_out) {
                    const int tap = k / Cin, ci = k - tap * Cin;
                    if (ci < cin_out) dw[((long long)co * cin_out + ci) * taps + tap] = t[e];
                }
            }
        }
        return;
    }
    for (long long i0 = (long long)lb * 32; i0 < nq; i0 += (long long)nb * 32) {
        const long long i = i0 + q;
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        if (i < nq)
#pragma unroll 4
            for (int c = l; c < chunks; c += 8) s += *(const f32x4*)(ws + (long long)c * dwgs + i * 4);
        red[l][q] = s;
        __syncthreads();
        if (l == 0 && i < nq) {
            f32x4 t = red[0][q];
#pragma unroll
            for (int jj = 1; jj < 8; ++jj) t += red[jj][q];
            const long long e0 = i * 4;
            const int co = (int)(e0 / Kpad), k0 = (int)(e0 - (long long)co * Kpad);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int k = k0 + e;
                if (k < taps * Cin && co < cout_out) {
                    const int tap = k / Cin, ci = k - tap * Cin;
                    if (ci < cin_out) dw[((long long)co * cin_out + ci) * taps + tap] = t[e];
                }
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------ stem weight gradient by strips
// trunk.0's weight gradient (7x7 / stride 2 / pad 3, Cin = 3, Cout = 64; meta_template.py:76-92 trains the whole backbone):
// dW[co][kh][kw][ci] = sum over output pixels m of dy[m][co] * x[2 oh - 3 + kh][2 ow - 3 + kw][ci].  The generic kernel above
// (STEMW) gathers its im2col operand element by element: four scalar loads per thread and 32-row step, each behind two integer
// divisions -- 84 us for the 105-image episode (41 TFLOP/s), 283 us for four.  For a fixed kernel row kh the 21 values (kw, ci) of
// an im2col row are CONTIGUOUS in the image row, and consecutive output pixels are 6 floats apart (csrc/stem.hip uses the same fact
// forward): a workgroup stages, per strip of S <= 64 output pixels of one output row, dy [S][64] and the seven image rows (6 S + 32
// floats each, zero beyond the image) in LDS; the reduction runs over the strip's pixels in pairs, the MFMA's A operand is dy (rows =
// co), its B operand the image row at offset 6 m + j (columns = j = kw * 3 + ci, 21 of 32 used): 7 x 2 accumulator tiles (kh, co
// half) dealt to the eight waves of a 512-thread workgroup, kept in registers over all strips of the workgroup and written once as the partial gradient of
// "chunk" blockIdx.x -- the split-M reducer above sums the chunks in order, as for every other layer.  The next strip's global
// loads are issued before the current strip's MFMAs (registers), so one workgroup per CU suffices.
struct StemWgArgs {
    const float* in; const float* dy; float* ws;
    int n_img, H, W, ldi, ldy, OH, OW, S, nseg, Kpad;
    long long dwgs;
};

__global__ __launch_bounds__(512) void stem_wgrad_strip_kernel(StemWgArgs p) {
    constexpr int NT = 512, NWAVE = NT / 64, NQ = (14 + NWAVE - 1) / NWAVE;       // eight waves, two accumulator tiles each (waves 6, 7: one)
    constexpr int PR = 6 * 64 + 32;                   // floats per staged image row
    __shared__ float dy_s[64 * 64];
    __shared__ float prow[7 * PR];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int S2 = (p.S + 1) & ~1;                     // reduction length per strip (pairs of output pixels)
    const int n_strips = p.n_img * p.OH * p.nseg;
    const int plen = 6 * p.S + 32;                     // staged floats per image row
    // this wave's accumulator tiles: combos c = wave, wave + 8 (< 14); c -> (kh = c >> 1, co half = c & 1 = wave & 1)
    const int cb = wave & 1;
    f32x16 acc[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[q][e] = 0.f;

    constexpr int NDY = (64 * 64 / 4 + NT - 1) / NT;  // float4 of dy per thread (2)
    constexpr int NPR = (7 * PR + NT - 1) / NT;       // floats of the image rows per thread (6)
    // what a thread stages does not depend on the strip: decoded once (image row kh, pixel offset, channel, LDS slot; -1: nothing)
    int e_kh[NPR], e_q3[NPR], e_ci[NPR], e_lds[NPR];
#pragma unroll
    for (int u = 0; u < NPR; ++u) {
        const int i = tid + NT * u;
        const int kh = i / plen, idx = i - kh * plen;
        e_kh[u] = kh; e_q3[u] = idx / 3; e_ci[u] = idx - e_q3[u] * 3;
        e_lds[u] = i < 7 * plen ? kh * PR + idx : -1;
    }
    f32x4 rdy[NDY];
    float rpr[NPR];
    auto load_strip = [&](const int s) {
        const int seg = s % p.nseg, row = s / p.nseg;                  // row = img * OH + oh
        const int img = row / p.OH, oh = row - img * p.OH;
        const int ow0 = seg * p.S;
        const int sv = min(p.S, p.OW - ow0);
#pragma unroll
        for (int u = 0; u < NDY; ++u) {
            const int i = tid + NT * u, m = i >> 4, c4 = (i & 15) * 4;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (m < sv) v = *(const f32x4*)(p.dy + ((long long)row * p.OW + ow0 + m) * p.ldy + c4);
            rdy[u] = v;
        }
        const long long ibase = (long long)img * p.H * p.W;
#pragma unroll
        for (int u = 0; u < NPR; ++u) {
            const int ih = 2 * oh - 3 + e_kh[u], iw = 2 * ow0 - 3 + e_q3[u];
            float v = 0.f;
            if (e_lds[u] >= 0 && ih >= 0 && ih < p.H && iw >= 0 && iw < p.W) v = p.in[(ibase + (long long)ih * p.W + iw) * p.ldi + e_ci[u]];
            rpr[u] = v;
        }
    };
    auto store_strip = [&]() {
#pragma unroll
        for (int u = 0; u < NDY; ++u) {
            const int i = tid + NT * u;
            *(f32x4*)(dy_s + (i >> 4) * 64 + (i & 15) * 4) = rdy[u];
        }
#pragma unroll
        for (int u = 0; u < NPR; ++u)
            if (e_lds[u] >= 0) prow[e_lds[u]] = rpr[u];
    };

    int kh_q[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) kh_q[q] = wave + NWAVE * q < 14 ? (wave + NWAVE * q) >> 1 : 0;
    int s = blockIdx.x;
    if (s < n_strips) load_strip(s);
    for (; s < n_strips; s += gridDim.x) {
        __syncthreads();                               // the previous strip's fragment reads are done
        store_strip();
        __syncthreads();
        if (s + (int)gridDim.x < n_strips) load_strip(s + gridDim.x);
        const float* ap = dy_s + h * 64 + cb * 32 + r;
        const float* bp = prow + 6 * h + r;
        // (branch-free body, unrolled: the LDS reads of several pixel pairs are in flight ahead of their MFMAs.  Waves 6 and 7 have
        // one real tile; their second accumulator multiplies image row 0 again and is never written out)
        const int np = S2 >> 1;                        // pixel pairs
        int tp = 0;
        for (; tp + 4 <= np; tp += 4) {
            float a[4], b[4][NQ];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                a[u] = ap[(tp + u) * 128];
#pragma unroll
                for (int q = 0; q < NQ; ++q) b[u][q] = bp[kh_q[q] * PR + 12 * (tp + u)];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int q = 0; q < NQ; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u][q], acc[q], 0, 0, 0);
        }
        for (; tp < np; ++tp) {
            const float a = ap[tp * 128];
#pragma unroll
            for (int q = 0; q < NQ; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bp[kh_q[q] * PR + 12 * tp], acc[q], 0, 0, 0);
        }
    }
    // partial gradient of this workgroup: ws[chunk][co][kh * 21 + j], j = kw * 3 + ci < 21 (C/D layout: col = lane & 31, row = (e&3) + 8 (e>>2) + 4 h)
    float* out = p.ws + (long long)blockIdx.x * p.dwgs;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int c = wave + NWAVE * q;
        if (c >= 14 || r >= 21) continue;
        const int kh = c >> 1;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int co = cb * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
            out[(long long)co * p.Kpad + kh * 21 + r] = acc[q][e];
        }
    }
}

int g_stem_wgrad_strips = 1;       // 0: the generic gather kernel for the stem's weight gradient (mft_debug_set_conv_tile(2100 / 2101))

template <int BM, int BN, bool ADAM, bool STEMW = false, bool EARLYT = false, int POL = 3>
int launch_wgrad(const WgradArgs& a, int taps, int groups, hipStream_t s) {
    constexpr int lds_mm = 32 * (BM + BN) * 4;
    constexpr int lds_ad = BM * (BN + 4) * 4;
    constexpr int lds0 = ADAM ? (lds_ad > lds_mm ? lds_ad : lds_mm) : lds_mm;
    int lds = lds0;
    if (ADAM && g_wgrad_min_lds_kb * 1024 > lds) lds = g_wgrad_min_lds_kb * 1024;
    auto kern = conv_wgrad_kernel<BM, BN, ADAM, STEMW, EARLYT, POL>;
    if (lds > 64 * 1024) {
        static MftPerDeviceOnce attr_once;
        if (attr_once.need()) {
            hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            if (e != hipSuccess) return (int)e;
            attr_once.mark();
        }
    }
    WgradArgs p = a;
    p.tiles_ci = ((STEMW ? a.Kpad : a.Cin) + BN - 1) / BN;
    p.tiles_co = (a.Cout + BM - 1) / BM;
    dim3 grid(p.tiles_ci * p.tiles_co * taps, groups, p.chunks);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, p);
    if (p.oihw || p.chunks > 1) {
        const long long n = (long long)a.Cout * a.Kpad;
        int blocks = (int)((n / 4 + 31) / 32);
        if (blocks > 4096) blocks = 4096;
        if (p.oihw)
            hipLaunchKernelGGL(reduce_chunks_kernel<true>, dim3(blocks, 1), dim3(256), 0, s, (const float*)p.ws, p.dw, n, p.chunks, p.dwgs,
                               a.Cin, a.KH * a.KW, a.Kpad, a.cin_out > 0 ? a.cin_out : a.Cin, a.cout_out > 0 ? a.cout_out : a.Cout);
        else
            hipLaunchKernelGGL(reduce_chunks_kernel<false>, dim3(blocks, groups), dim3(256), 0, s, (const float*)p.ws, p.dw, n, p.chunks,
                               p.dwgs, a.Cin, a.KH * a.KW, a.Kpad, a.Cin, a.Cout);
    }
    return mft_launch_status();
}

constexpr int WGRAD_CHUNK_ROWS = 1024;

// split-M of the plain (no Adam) weight gradient: one workgroup per (64 x 64 tile, tap, group, chunk of rows), partial
// gradients summed in chunk order by reduce_chunks_kernel (deterministic).  A meta-training step has ONE 105-image episode in
// flight (train.py:28, meta_template.py:76-92), so every launch is a small GEMM with a long reduction: the chunk length is
// chosen so that the launch has ~3 workgroups per CU -- with the 1024-row chunks of round 2 the GNN's 1x1 layers (7,440 pair
// rows, 3-9 tiles) ran on 24-72 workgroups for 51-53 us each and trunk.7 (945 rows) on one chunk.
inline void wgrad_chunking(long long rows, long long wgs_per_chunk, int* chunk_rows, int* chunks) {
    *chunk_rows = (int)rows;
    *chunks = 1;
    if (rows <= 256) return;
    const long long want = (768 + wgs_per_chunk - 1) / wgs_per_chunk;
    long long cr = ((rows + want - 1) / want + 31) / 32 * 32;
    if (cr < 128) cr = 128;
    if (cr > WGRAD_CHUNK_ROWS) cr = WGRAD_CHUNK_ROWS;
    const long long n = (rows + cr - 1) / cr;
    if (n <= 1) return;
    *chunk_rows = (int)cr;
    *chunks = (int)n;
}

inline long long wgrad_wgs_per_chunk(int Cin, int Cout, int KH, int KW, long long groups) {
    if (Cin == 3) return (long long)(((KH * KW * Cin + 31) / 32 * 32 + 63) / 64) * ((Cout + 63) / 64) * groups;       // stem: K-major tiles
    return (long long)((Cin + 63) / 64) * ((Cout + 63) / 64) * KH * KW * groups;
}

int wgrad_dispatch(WgradArgs a, int n_img, int imgs_per_group, bool adam, float* ws, hipStream_t s) {
    if (imgs_per_group <= 0) imgs_per_group = n_img;
    if (n_img % imgs_per_group != 0) return MFT_EINVAL;
    const bool stem = (a.Cin == 3);
    if ((!stem && a.Cin % 4 != 0) || a.Cout % 4 != 0 || (!stem && a.ldi % 4 != 0) || a.ldy % 4 != 0) return MFT_EINVAL;
    const int groups = n_img / imgs_per_group;
    a.OH = (a.H + 2 * a.pad - a.KH) / a.stride + 1;
    a.OW = (a.W + 2 * a.pad - a.KW) / a.stride + 1;
    a.Kpad = (a.KH * a.KW * a.Cin + 31) / 32 * 32;
    a.imgs_per_group = imgs_per_group;
    a.rows_per_group = imgs_per_group * a.OH * a.OW;
    a.chunk_rows = a.rows_per_group;
    a.chunks = 1;
    a.ws = ws;
    if (a.oihw && (adam || ws == nullptr || groups != 1)) return MFT_EINVAL;
    if (!adam && ws != nullptr) wgrad_chunking(a.rows_per_group, wgrad_wgs_per_chunk(a.Cin, a.Cout, a.KH, a.KW, groups), &a.chunk_rows, &a.chunks);
    const int taps = a.KH * a.KW;
    if (stem && !adam && g_stem_wgrad_strips && a.oihw && a.KH == 7 && a.KW == 7 && a.stride == 2 && a.pad == 3 && a.Cout == 64 && groups == 1 &&
        a.chunks > 1 && a.Kpad == 160 && (((unsigned long long)a.dy) & 15) == 0 && (a.cin_out <= 0 || a.cin_out == a.Cin) &&
        (a.cout_out <= 0 || a.cout_out == a.Cout)) {
        StemWgArgs q;
        q.in = a.in; q.dy = a.dy; q.ws = a.ws; q.n_img = n_img; q.H = a.H; q.W = a.W; q.ldi = a.ldi; q.ldy = a.ldy; q.OH = a.OH; q.OW = a.OW;
        q.nseg = (a.OW + 63) / 64;
        q.S = (a.OW + q.nseg - 1) / q.nseg;
        q.Kpad = a.Kpad; q.dwgs = a.dwgs;
        int wgs = a.chunks;                            // (the workspace holds `chunks` partial gradients; the reducer sums that many)
        const long long strips = (long long)n_img * a.OH * q.nseg;
        if (strips < wgs) return launch_wgrad<64, 64, false, true>(a, 1, groups, s);
        hipLaunchKernelGGL(stem_wgrad_strip_kernel, dim3(wgs), dim3(512), 0, s, q);
        const long long n = (long long)a.Cout * a.Kpad;
        int blocks = (int)((n / 4 + 31) / 32);
        hipLaunchKernelGGL(reduce_chunks_kernel<true>, dim3(blocks, 1), dim3(256), 0, s, (const float*)a.ws, a.dw, n, a.chunks, a.dwgs, a.Cin,
                           a.KH * a.KW, a.Kpad, a.Cin, a.Cout);
        return mft_launch_status();
    }
    if (stem) return adam ? MFT_EINVAL : launch_wgrad<64, 64, false, true>(a, 1, groups, s);
    if (adam) {
        if (a.Cin % 64 != 0 || a.Cout % 64 != 0) return MFT_EINVAL;
        if (g_wgrad_rows && a.rows_per_group <= 64 && a.Cin % 128 == 0 && a.Cout % 32 == 0)
            return launch_wgrad_adam_rows(a, taps, groups, s);
        if (a.Cin % 128 == 0 && a.Cout % 128 == 0 && g_wgrad_tile != 64) return launch_wgrad<128, 128, true>(a, taps, groups, s);
        if (g_wgrad_early && (g_wgrad_pol & 3) == 0) return launch_wgrad<64, 64, true, false, true, 0>(a, taps, groups, s);
        if (g_wgrad_early && (g_wgrad_pol & 3) == 1) return launch_wgrad<64, 64, true, false, true, 1>(a, taps, groups, s);
        if (g_wgrad_early && (g_wgrad_pol & 3) == 2) return launch_wgrad<64, 64, true, false, true, 2>(a, taps, groups, s);
        if (g_wgrad_early) return launch_wgrad<64, 64, true, false, true>(a, taps, groups, s);
        return launch_wgrad<64, 64, true>(a, taps, groups, s);
    }
    return launch_wgrad<64, 64, false>(a, taps, groups, s);
}

}  // namespace

static int conv2d_impl(const float* in, int ldi, const float* w, const float* bias, float* out, int ldo,
                       int n_img, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad,
                       int imgs_per_group, long long w_group_stride, float* ws, void* stream) {
    if (n_img <= 0 || Cout <= 0) return MFT_EINVAL;
    const bool stem = (Cin == 3);
    if (!stem && (Cin % 32 != 0 || ldi % 4 != 0)) return MFT_EINVAL;
    if (imgs_per_group <= 0) imgs_per_group = n_img;
    if (n_img % imgs_per_group != 0) return MFT_EINVAL;
    const int groups = n_img / imgs_per_group;
    ConvArgs a;
    a.in = in; a.w = w; a.bias = bias; a.out = out;
    a.ldi = ldi; a.ldo = ldo;
    a.H = H; a.W = W; a.Cin = Cin; a.KH = KH; a.KW = KW; a.stride = stride; a.pad = pad;
    a.OH = (H + 2 * pad - KH) / stride + 1;
    a.OW = (W + 2 * pad - KW) / stride + 1;
    a.Cout = Cout;
    a.Ktot = KH * KW * Cin;
    a.Kpad = (a.Ktot + 31) / 32 * 32;
    a.imgs_per_group = imgs_per_group;
    a.rows_per_group = imgs_per_group * a.OH * a.OW;
    a.wgs = (groups > 1) ? w_group_stride : 0;
    a.tiles_n = 0;
    a.bt_stride = 1;
    a.ksplit = 1; a.ws = nullptr; a.parity = 0;
    if (ws != nullptr && !stem && ldo % 4 == 0) {
        a.ksplit = conv_ksplit(a.rows_per_group, Cout, a.Kpad, groups);
        a.ws = ws;
    }
    hipStream_t s = (hipStream_t)stream;
    if (stem) {
        // 7x7/2 pad 3 -> 64 channels on 84x84 / 224x224 inputs: LDS-patch kernel (csrc/stem.hip)
        if (g_stem_fast && KH == 7 && KW == 7 && stride == 2 && pad == 3 && Cout == 64 && ldi == 3 && ldo == 64 &&
            bias == nullptr && (W == 84 || W == 224)) {
            const int rc = mft_stem_conv_dispatch(in, w, out, n_img, H, W, a.Kpad, s);
            if (rc != MFT_EINVAL) return rc;
        }
        return launch_conv<128, 64, 2, 2, true>(a, groups, s);
    }
    if (g_skinny && groups > 1 && w_group_stride != 0 && bias == nullptr && ws == nullptr) {
        // per-episode weights, <= 48 output pixels per episode: weight-streaming skinny kernel (csrc/skinny.hip)
        const int rc = mft_skinny_fwd_dispatch(in, ldi, w, out, ldo, n_img, H, W, Cin, Cout, KH, KW, stride, pad,
                                               imgs_per_group, w_group_stride, s);
        if (rc != MFT_EINVAL) return rc;
    }
    // tile choice (tools/conv_tune.py, MI355X): with fp32 MFMA the 64x64 tile (58 VGPRs, 36.9 KB LDS, 4 workgroups
    // per CU) beats every larger tile on all ResNet10 shapes (84-97 vs 61-88 TFLOP/s): latency hiding through
    // occupancy matters more than operand reuse, LDS bandwidth is not a constraint at 2 floats per 64-cycle MFMA.
    int tile = g_conv_tile;
    if (tile == 0) tile = (Cout >= 64) ? 4 : 5;
    switch (tile) {
        case 1: return launch_conv<128, 128, 2, 2, false>(a, groups, s);
        case 2: return launch_conv<128, 64, 2, 2, false>(a, groups, s);
        case 3: return launch_conv<64, 128, 2, 2, false>(a, groups, s);
        case 4: return launch_conv<64, 64, 2, 2, false>(a, groups, s);
        default: return launch_conv<128, 32, 4, 1, false>(a, groups, s);
    }
}

extern "C" int mft_conv2d_nhwc(const float* in, int ldi, const float* w, const float* bias, float* out, int ldo,
                               int n_img, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad,
                               int imgs_per_group, long long w_group_stride, void* stream) {
    return conv2d_impl(in, ldi, w, bias, out, ldo, n_img, H, W, Cin, Cout, KH, KW, stride, pad, imgs_per_group, w_group_stride,
                       nullptr, stream);
}

// floats of workspace the K-sliced forms want for a single-group launch producing rows x cols outputs over a reduction of K
// (0: the launch is not sliced)
extern "C" long long mft_conv_ksplit_ws_floats(long long rows, int cols, int K) {
    const int s = conv_ksplit(rows, cols, K);
    return s > 1 ? (long long)s * rows * cols : 0;
}

// The K-sliced forms for a FEW weight sets (2 <= episodes <= 8 in lockstep: grid.y = episode): the weight-streaming per-episode
// kernels put one workgroup set per episode on the GPU, which leaves most CUs idle below ~16 episodes.
extern "C" long long mft_conv_ksplit_grouped_ws_floats(long long rows_per_group, int cols, int K, int groups) {
    const int s = conv_ksplit(rows_per_group, cols, K, groups);
    return s > 1 ? (long long)s * groups * rows_per_group * cols : 0;
}

extern "C" int mft_conv2d_nhwc_ksplit_grouped(const float* in, int ldi, const float* w, float* out, int ldo, int n_img, int H, int W,
                                              int Cin, int Cout, int KH, int KW, int stride, int pad, int imgs_per_group,
                                              long long w_group_stride, float* ws, void* stream) {
    if (ws == nullptr || imgs_per_group <= 0) return MFT_EINVAL;
    return conv2d_impl(in, ldi, w, nullptr, out, ldo, n_img, H, W, Cin, Cout, KH, KW, stride, pad, imgs_per_group, w_group_stride, ws,
                       stream);
}

extern "C" int mft_conv2d_nhwc_ksplit(const float* in, int ldi, const float* w, const float* bias, float* out, int ldo,
                                      int n_img, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad, float* ws,
                                      void* stream) {
    return conv2d_impl(in, ldi, w, bias, out, ldo, n_img, H, W, Cin, Cout, KH, KW, stride, pad, 0, 0, ws, stream);
}

extern "C" int mft_has_experiments(void) {
#ifdef MFT_EXPERIMENTS
    return 1;
#else
    return 0;
#endif
}

extern "C" int mft_debug_set_conv_tile(int tile) {
#ifndef MFT_EXPERIMENTS
    // product build: the knobs select VALIDATED alternative paths only (tile shapes, generic vs weight-streaming per-episode kernels
    // and their fp32 / per-tap / fragment-order forms, 64 x 64 vs 32 x 128 weight-gradient kernel, exact vs fast Adam epilogue,
    // trimmed vs padded reduction).  The measured-slower experiment kernels and the measurement aids are not compiled in.
    if (tile == 9502 || tile == 9504 || tile == 9505) return MFT_EINVAL;                       // order switch, walking forms
    if (tile >= 9000 && tile < 9100 && tile != 9003 && tile != 9007) return MFT_EINVAL;        // cache-policy / ablation variants
    if (tile > 4000 && tile < 5000) return MFT_EINVAL;                                         // occupancy throttle
#endif
    if (tile >= 11000) { mft_bn_small_set_rows(tile - 11000); mft_bn_fwd_small_set_rows(tile - 11000); }   // 11000: the multi-launch BatchNorm forms always; 11000 + r: one launch up to r rows per group
    else if (tile >= 9800) g_dgrad_parity = tile - 9800;          // 9800 / 9801: stride-2 data gradient over all taps / by pixel parity class
    else if (tile >= 9700) mft_skinny_set_lines(tile - 9700);
    else if (tile >= 9600) g_wgrad_trim = tile - 9600;
    else if (tile >= 9500) g_wgrad_rows = tile - 9500;
    else if (tile >= 9100) mft_skinny_set_nw(tile - 9100);
    else if (tile >= 9000) g_wgrad_pol = tile - 9000;
    else if (tile >= 8000) mft_skinny_set_x3(tile - 8000);
    else if (tile >= 7000) mft_skinny_set_tap(tile - 7000);
    else if (tile >= 6000) mft_skinny_set_dgrad_slices(tile - 6000);
    else if (tile >= 5000) g_wgrad_early = tile - 5000;
    else if (tile >= 4000) g_wgrad_min_lds_kb = tile - 4000;
    else if (tile >= 3000) g_skinny = tile - 3000;          // 3000 / 3001: generic / skinny per-episode kernels
    else if (tile >= 2100) g_stem_wgrad_strips = tile - 2100;   // 2100 / 2101: generic gather / strip kernel for the stem's weight gradient
    else if (tile >= 2000) g_stem_fast = tile - 2000;       // 2000 / 2001: generic / LDS-patch stem kernel
    else if (tile >= 1000) g_wgrad_tile = tile - 1000; // 1064 / 1128: choose the wgrad tile
    else g_conv_tile = tile;
    return 0;
}

extern "C" int mft_debug_set_x3_tile(int t);

// Every tuning knob back to its default (tests call this from an always-run fixture: a failed assert between a set and its
// hand-written restore must not leave later tests on a different kernel variant).
extern "C" void mft_wgrad_fwd_set_exact(int on);
extern "C" int mft_debug_reset(void) {
    mft_wgrad_fwd_set_exact(0);
    mft_wgrad_fwd_set_xcd(1);
    g_dgrad_parity = 1; g_wgrad_trim = 1; g_wgrad_rows = 1; g_wgrad_tile = 64; g_conv_tile = 0; g_wgrad_pol = 7; g_wgrad_early = 1; g_wgrad_min_lds_kb = 0; g_skinny = 1; g_stem_fast = 1; g_stem_wgrad_strips = 1;
    mft_skinny_set_lines(1); mft_skinny_set_nw(0); mft_skinny_set_x3(1); mft_skinny_set_tap(1); mft_skinny_set_dgrad_slices(1);
    mft_debug_set_x3_tile(0); mft_debug_set_x3_tile(10); mft_debug_set_x3_tile(21); mft_debug_set_x3_tile(31); mft_debug_set_x3_tile(41);
    mft_bn_small_set_rows(512); mft_bn_fwd_small_set_rows(512);
    mft_debug_set_x3_tile(60); mft_debug_set_x3_tile(70); mft_debug_set_x3_tile(80); mft_debug_set_x3_tile(91); mft_debug_set_x3_tile(100); mft_debug_set_x3_tile(200);
    return 0;
}

static int dgrad_impl(const float* dy, int ldy, const float* w, float* dx, int ldx, int n_img, int H,
                      int W, int Cin, int Cout, int KH, int KW, int stride, int pad,
                      int imgs_per_group, long long w_group_stride, float* ws, void* stream) {
    // H, W: spatial size of dx (= forward input); dy is [(H+2p-KH)/s+1, (W+2p-KW)/s+1]
    if (Cout % 32 != 0 || Cin % 4 != 0 || ldy % 4 != 0 || stride < 1) return MFT_EINVAL;
    if (imgs_per_group <= 0) imgs_per_group = n_img;
    if (n_img % imgs_per_group != 0) return MFT_EINVAL;
    const int groups = n_img / imgs_per_group;
    if (g_skinny && groups > 1 && w_group_stride != 0 && ws == nullptr) {
        const int rc = mft_skinny_dgrad_dispatch(dy, ldy, w, dx, ldx, n_img, H, W, Cin, Cout, KH, KW, stride, pad,
                                                 imgs_per_group, w_group_stride, (hipStream_t)stream);
        if (rc != MFT_EINVAL) return rc;
    }
    ConvArgs a;
    a.in = dy; a.w = w; a.bias = nullptr; a.out = dx;
    a.ldi = ldy; a.ldo = ldx;
    a.H = (H + 2 * pad - KH) / stride + 1;       // dy spatial size (the "input" of this gather)
    a.W = (W + 2 * pad - KW) / stride + 1;
    a.Cin = Cout; a.KH = KH; a.KW = KW;
    a.stride = 1; a.pad = -pad;                  // row setup yields a_ih0 = h + pad
    a.bt_stride = stride;
    a.OH = H; a.OW = W;
    a.Cout = Cin;
    a.Ktot = KH * KW * Cout;
    a.Kpad = a.Ktot;
    a.imgs_per_group = imgs_per_group;
    a.rows_per_group = imgs_per_group * H * W;
    a.wgs = (groups > 1) ? w_group_stride : 0;
    a.tiles_n = 0;
    a.ksplit = 1; a.ws = nullptr;
    a.parity = (groups == 1 && stride == 2 && g_dgrad_parity) ? 1 : 0;
    if (ws != nullptr && ldx % 4 == 0 && Cin % 64 == 0) {
        const int s_full = conv_ksplit(a.rows_per_group, Cin, a.Kpad, groups);      // what mft_conv_ksplit_ws_floats sized ws for
        if (!a.parity) {
            a.ksplit = s_full;
            a.ws = ws;
        } else if (s_full > 1) {
            // parity classes: the class with the most taps walks ceil(KH/2) * ceil(KW/2) * Cout (trunk.7.C1's data gradient: 64
            // K-steps on 240 tiles = 86 us of dependent steps, round 5); sliced like a four-group launch of the largest class
            int sp = conv_ksplit((long long)imgs_per_group * ((H + 1) / 2) * ((W + 1) / 2), Cin, ((KH + 1) / 2) * ((KW + 1) / 2) * Cout, 4);
            if (sp > s_full) sp = s_full;
            if (sp > 1) { a.ksplit = sp; a.ws = ws; }
        }
    }
    hipStream_t s = (hipStream_t)stream;
    if (Cin % 64 == 0) return launch_conv<64, 64, 2, 2, false, true>(a, groups, s);
    return launch_conv<128, 32, 4, 1, false, true>(a, groups, s);
}

extern "C" int mft_conv2d_dgrad_nhwc(const float* dy, int ldy, const float* w, float* dx, int ldx, int n_img, int H,
                                     int W, int Cin, int Cout, int KH, int KW, int stride, int pad,
                                     int imgs_per_group, long long w_group_stride, void* stream) {
    return dgrad_impl(dy, ldy, w, dx, ldx, n_img, H, W, Cin, Cout, KH, KW, stride, pad, imgs_per_group, w_group_stride, nullptr,
                      stream);
}

extern "C" int mft_conv2d_dgrad_nhwc_ksplit_grouped(const float* dy, int ldy, const float* w, float* dx, int ldx, int n_img, int H,
                                                    int W, int Cin, int Cout, int KH, int KW, int stride, int pad, int imgs_per_group,
                                                    long long w_group_stride, float* ws, void* stream) {
    if (ws == nullptr || imgs_per_group <= 0 || stride != 1) return MFT_EINVAL;
    return dgrad_impl(dy, ldy, w, dx, ldx, n_img, H, W, Cin, Cout, KH, KW, stride, pad, imgs_per_group, w_group_stride, ws, stream);
}

// the same data gradient with the reduction over (tap, output channel) sliced across workgroups (workspace:
// mft_conv_ksplit_ws_floats(n_img*H*W, Cin, KH*KW*Cout) floats; single weight set)
extern "C" int mft_conv2d_dgrad_nhwc_ksplit(const float* dy, int ldy, const float* w, float* dx, int ldx, int n_img, int H, int W,
                                            int Cin, int Cout, int KH, int KW, int stride, int pad, float* ws, void* stream) {
    return dgrad_impl(dy, ldy, w, dx, ldx, n_img, H, W, Cin, Cout, KH, KW, stride, pad, 0, 0, ws, stream);
}

extern "C" long long mft_conv2d_wgrad_ws_floats(int n_img, int H, int W, int Cin, int Cout, int KH, int KW, int stride,
                                                 int pad, int imgs_per_group) {
    if (imgs_per_group <= 0) imgs_per_group = n_img;
    const int OH = (H + 2 * pad - KH) / stride + 1, OW = (W + 2 * pad - KW) / stride + 1;
    const long long rows = (long long)imgs_per_group * OH * OW;
    int chunk_rows, chunks;
    wgrad_chunking(rows, wgrad_wgs_per_chunk(Cin, Cout, KH, KW, n_img / imgs_per_group), &chunk_rows, &chunks);
    if (chunks <= 1) return 0;
    const long long kpad = (KH * KW * Cin + 31) / 32 * 32;
    return (long long)(n_img / imgs_per_group) * chunks * Cout * kpad;
}

extern "C" int mft_conv2d_wgrad_nhwc(const float* in, int ldi, const float* dy, int ldy, float* dw,
                                     int n_img, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad,
                                     int imgs_per_group, long long dw_group_stride, float* ws, void* stream) {
    WgradArgs a = {};
    a.in = in; a.dy = dy; a.dw = dw; a.ldi = ldi; a.ldy = ldy;
    a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.KH = KH; a.KW = KW; a.stride = stride; a.pad = pad;
    a.dwgs = dw_group_stride;
    return wgrad_dispatch(a, n_img, imgs_per_group, false, ws, (hipStream_t)stream);
}

// mft_conv2d_wgrad_nhwc for ONE weight set with the gradient written in torch's [Cout][Cin][KH][KW] layout (what
// Conv2d.weight.grad holds): the partial-gradient sum does the permutation.  ws: mft_conv2d_wgrad_oihw_ws_floats floats.
extern "C" long long mft_conv2d_wgrad_oihw_ws_floats(int n_img, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad) {
    const int OH = (H + 2 * pad - KH) / stride + 1, OW = (W + 2 * pad - KW) / stride + 1;
    int chunk_rows, chunks;
    wgrad_chunking((long long)n_img * OH * OW, wgrad_wgs_per_chunk(Cin, Cout, KH, KW, 1), &chunk_rows, &chunks);
    return (long long)chunks * Cout * ((KH * KW * Cin + 31) / 32 * 32);
}

extern "C" int mft_conv2d_wgrad_oihw(const float* in, int ldi, const float* dy, int ldy, float* dw_oihw, int n_img, int H, int W,
                                     int Cin, int Cout, int KH, int KW, int stride, int pad, int cin_valid, int cout_valid, float* ws,
                                     void* stream) {
    if (ws == nullptr || cin_valid < 0 || cin_valid > Cin || cout_valid < 0 || cout_valid > Cout) return MFT_EINVAL;
    WgradArgs a = {};
    a.in = in; a.dy = dy; a.dw = dw_oihw; a.ldi = ldi; a.ldy = ldy;
    a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.KH = KH; a.KW = KW; a.stride = stride; a.pad = pad;
    a.oihw = 1;
    a.cin_out = cin_valid; a.cout_out = cout_valid;
    a.dwgs = (long long)Cout * ((KH * KW * Cin + 31) / 32 * 32);
    return wgrad_dispatch(a, n_img, 0, false, ws, (hipStream_t)stream);
}

// Host side of the multi-problem launch: each job is prepared exactly as mft_conv2d_wgrad_oihw prepares it (same chunking, same
// workspace layout), then up to WG_MULTI jobs share a launch; the stem (Cin == 3: its own K-major kernel) keeps its own launch.
extern "C" int mft_conv2d_wgrad_oihw_multi(const MftWgradJob* jobs, int n_jobs, void* stream) {
    if (jobs == nullptr || n_jobs < 1) return MFT_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    for (int j0 = 0; j0 < n_jobs;) {
        WgradMultiArgs wa = {};
        ReduceMultiArgs ra = {};
        int cnt = 0, wgs = 0, rblocks = 0;
        for (; j0 < n_jobs && cnt < WG_MULTI; ++j0) {
            const MftWgradJob& jb = jobs[j0];
            if (jb.ws == nullptr || jb.cin_valid < 0 || jb.cin_valid > jb.Cin || jb.cout_valid < 0 || jb.cout_valid > jb.Cout) return MFT_EINVAL;
            if (jb.Cin == 3) {                                        // the stem: single launch (STEMW kernel)
                int rc = mft_conv2d_wgrad_oihw(jb.in, jb.ldi, jb.dy, jb.ldy, jb.dw_oihw, jb.n_img, jb.H, jb.W, jb.Cin, jb.Cout, jb.KH, jb.KW,
                                               jb.stride, jb.pad, jb.cin_valid, jb.cout_valid, jb.ws, stream);
                if (rc != 0) return rc;
                continue;
            }
            if (jb.Cin % 4 != 0 || jb.Cout % 4 != 0 || jb.ldi % 4 != 0 || jb.ldy % 4 != 0 || jb.n_img < 1) return MFT_EINVAL;
            WgradArgs a = {};
            a.in = jb.in; a.dy = jb.dy; a.dw = jb.dw_oihw; a.ldi = jb.ldi; a.ldy = jb.ldy;
            a.H = jb.H; a.W = jb.W; a.Cin = jb.Cin; a.Cout = jb.Cout; a.KH = jb.KH; a.KW = jb.KW; a.stride = jb.stride; a.pad = jb.pad;
            a.oihw = 1; a.cin_out = jb.cin_valid; a.cout_out = jb.cout_valid;
            a.OH = (a.H + 2 * a.pad - a.KH) / a.stride + 1;
            a.OW = (a.W + 2 * a.pad - a.KW) / a.stride + 1;
            a.Kpad = (a.KH * a.KW * a.Cin + 31) / 32 * 32;
            a.dwgs = (long long)a.Cout * a.Kpad;
            a.imgs_per_group = jb.n_img;
            a.rows_per_group = jb.n_img * a.OH * a.OW;
            a.ws = jb.ws;
            wgrad_chunking(a.rows_per_group, wgrad_wgs_per_chunk(a.Cin, a.Cout, a.KH, a.KW, 1), &a.chunk_rows, &a.chunks);
            a.tiles_ci = (a.Cin + 63) / 64;
            a.tiles_co = (a.Cout + 63) / 64;
            const int nx = a.tiles_ci * a.tiles_co * a.KH * a.KW;
            wa.job[cnt] = a; wa.start[cnt] = wgs; wa.nx[cnt] = nx;
            wgs += nx * a.chunks;
            const long long n = (long long)a.Cout * a.Kpad;
            int blocks = (int)((n / 4 + 31) / 32);
            if (blocks > 4096) blocks = 4096;
            ra.ws[cnt] = a.ws; ra.dw[cnt] = a.dw; ra.n[cnt] = n; ra.dwgs[cnt] = a.dwgs; ra.chunks[cnt] = a.chunks; ra.Cin[cnt] = a.Cin;
            ra.taps[cnt] = a.KH * a.KW; ra.Kpad[cnt] = a.Kpad; ra.cin_out[cnt] = a.cin_out > 0 ? a.cin_out : a.Cin;
            ra.cout_out[cnt] = a.cout_out > 0 ? a.cout_out : a.Cout; ra.start[cnt] = rblocks;
            rblocks += blocks;
            ++cnt;
        }
        if (cnt == 0) continue;
        wa.start[cnt] = wgs; wa.n = cnt;
        ra.start[cnt] = rblocks; ra.cnt = cnt;
        constexpr int lds = 32 * (64 + 64) * 4;
        hipLaunchKernelGGL(conv_wgrad_multi_kernel, dim3(wgs), dim3(256), lds, s, wa);
        hipLaunchKernelGGL(reduce_chunks_multi_kernel, dim3(rblocks), dim3(256), 0, s, ra);
    }
    return mft_launch_status();
}

static int wgrad_adam_impl(const float* in, int ldi, const float* dy, int ldy, float* w, float* m, float* v,
                           float* dw_or_null, int n_img, int H, int W, int Cin, int Cout, int KH, int KW, int stride,
                           int pad, int imgs_per_group, long long group_stride, int step, const float* hyper, float lr,
                           float beta1, float beta2, float eps, void* stream) {
    if ((hyper == nullptr && step < 1) || (KH * KW * Cin) % 32 != 0) return MFT_EINVAL;
    if (hyper != nullptr) step = 1;
    WgradArgs a = {};
    a.in = in; a.dy = dy; a.dw = dw_or_null; a.ldi = ldi; a.ldy = ldy;
    a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.KH = KH; a.KW = KW; a.stride = stride; a.pad = pad;
    a.dwgs = group_stride;
    a.w = w; a.m = m; a.v = v;
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    a.step_size = (float)((double)lr / bc1);
    a.inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
    a.b1 = beta1; a.b2 = beta2; a.eps = eps;
    a.hyper = hyper;
    return wgrad_dispatch(a, n_img, imgs_per_group, true, nullptr, (hipStream_t)stream);
}

extern "C" int mft_conv2d_wgrad_adam_nhwc(const float* in, int ldi, const float* dy, int ldy, float* w, float* m,
                                          float* v, float* dw_or_null, int n_img, int H, int W, int Cin, int Cout,
                                          int KH, int KW, int stride, int pad, int imgs_per_group,
                                          long long group_stride, int step, float lr, float beta1, float beta2,
                                          float eps, void* stream) {
    return wgrad_adam_impl(in, ldi, dy, ldy, w, m, v, dw_or_null, n_img, H, W, Cin, Cout, KH, KW, stride, pad,
                           imgs_per_group, group_stride, step, nullptr, lr, beta1, beta2, eps, stream);
}

extern "C" int mft_conv2d_wgrad_adam_nhwc_dev(const float* in, int ldi, const float* dy, int ldy, float* w, float* m,
                                              float* v, float* dw_or_null, int n_img, int H, int W, int Cin, int Cout,
                                              int KH, int KW, int stride, int pad, int imgs_per_group,
                                              long long group_stride, const float* hyper, float beta1, float beta2,
                                              float eps, void* stream) {
    if (hyper == nullptr) return MFT_EINVAL;
    return wgrad_adam_impl(in, ldi, dy, ldy, w, m, v, dw_or_null, n_img, H, W, Cin, Cout, KH, KW, stride, pad,
                           imgs_per_group, group_stride, 1, hyper, 0.f, beta1, beta2, eps, stream);
}
