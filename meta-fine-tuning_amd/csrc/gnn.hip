// GNN few-shot head glue kernels: pairwise |x_i - x_j| features, diagonal-masked row softmax, graph
// aggregation cat(x, A x) and strided column copies.  The per-pair MLP GEMMs run on mft_conv2d_nhwc
// (1x1 conv == GEMM) and the BatchNorm/leaky-relu on the grouped BN kernels.
//
// Replaces the tensor ops of gnn.Wcompute.forward / gmul / GNN_nl.forward (gnn.py:16-28,78-132,154-166).
#include "mft_common.h"

namespace {

__global__ __launch_bounds__(256) void pair_absdiff_kernel(const float* __restrict__ x, int ldx,
                                                           float* __restrict__ d, int ldd, int n_graphs, int N,
                                                           int F) {
    const int q = ldd >> 2;
    const long long total = (long long)n_graphs * N * N * q;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % q) * 4;
        long long t = i / q;
        const int j = (int)(t % N); t /= N;
        const int ii = (int)(t % N);
        const int b = (int)(t / N);
        const float* xi = x + ((long long)b * N + ii) * ldx + c;
        const float* xj = x + ((long long)b * N + j) * ldx + c;
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (c + e < F) ? fabsf(xi[e] - xj[e]) : 0.f;
        *(f32x4*)(d + i * 4) = o;
    }
}

// one wave per (b,i) row; N <= 256 handled by striding
__global__ __launch_bounds__(256) void masked_softmax_kernel(const float* __restrict__ s, int lds_,
                                                             float* __restrict__ A, int n_graphs, int N) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long long rows = (long long)n_graphs * N;
    if (row >= rows) return;
    const int i = (int)(row % N);
    const float* sr = s + row * N * lds_;
    float mx = -3.4e38f;
    for (int j = lane; j < N; j += 64) {
        float v = sr[(long long)j * lds_] - (j == i ? 1e8f : 0.f);
        mx = fmaxf(mx, v);
    }
    mx = wave_max(mx);
    float se = 0.f;
    for (int j = lane; j < N; j += 64) se += __expf(sr[(long long)j * lds_] - (j == i ? 1e8f : 0.f) - mx);
    se = wave_sum(se);
    const float inv = 1.f / se;
    for (int j = lane; j < N; j += 64)
        A[row * N + j] = __expf(sr[(long long)j * lds_] - (j == i ? 1e8f : 0.f) - mx) * inv;
}

// y[b,i,0:F] = x[b,i,0:F]; y[b,i,F:2F] = sum_j A[b,i,j] x[b,j,0:F]; y[b,i,2F:ldy] = 0
__global__ __launch_bounds__(256) void graph_aggregate_kernel(const float* __restrict__ A,
                                                              const float* __restrict__ x, int ldx,
                                                              float* __restrict__ y, int ldy, int n_graphs, int N,
                                                              int F) {
    const long long row = blockIdx.x;      // (b,i)
    const int b = (int)(row / N);
    const float* Ar = A + row * N;
    const float* xb = x + (long long)b * N * ldx;
    const float* xi = x + row * ldx;
    float* yr = y + row * ldy;
    for (int c = threadIdx.x; c < ldy; c += blockDim.x) {
        float o = 0.f;
        if (c < F) o = xi[c];
        else if (c < 2 * F) {
            const int f = c - F;
            float acc = 0.f;
#pragma unroll 10
            for (int j = 0; j < N; ++j) acc += Ar[j] * xb[(long long)j * ldx + f];      // (unrolled: ten loads in flight, same order of additions)
            o = acc;
        }
        yr[c] = o;
    }
}

// Same contract, one workgroup per GRAPH: x[b] (N x F floats, <= 119 KB at N = 130, F = 229) is staged in LDS once and every
// thread owns one feature column, so each x element is fetched from L2 once per graph instead of once per node row (the
// row-per-workgroup form above re-reads x[b] N times: 21 GB of L2 traffic per launch at 128 episodes x 15 graphs x N = 105).
// Four node rows share one pass over j: 1 LDS read + 4 FMAs with wave-uniform A values.
__global__ __launch_bounds__(256) void graph_aggregate_lds_kernel(const float* __restrict__ A, const float* __restrict__ x, int ldx,
                                                                  float* __restrict__ y, int ldy, int N, int F) {
    extern __shared__ __attribute__((aligned(16))) float xs[];          // [N][F]
    const int b = blockIdx.x;
    const float* xb = x + (long long)b * N * ldx;
    const float* Ab = A + (long long)b * N * N;
    float* yb = y + (long long)b * N * ldy;
    for (int i = threadIdx.x; i < N * F; i += 256) {
        const int j = i / F, f = i - j * F;
        xs[i] = xb[(long long)j * ldx + f];
    }
    __syncthreads();
    for (int f = threadIdx.x; f < F; f += 256) {
        for (int i0 = 0; i0 < N; i0 += 4) {
            float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
            const float* r0 = Ab + (long long)i0 * N;
            const float* r1 = Ab + (long long)min(i0 + 1, N - 1) * N;
            const float* r2 = Ab + (long long)min(i0 + 2, N - 1) * N;
            const float* r3 = Ab + (long long)min(i0 + 3, N - 1) * N;
#pragma unroll 4
            for (int j = 0; j < N; ++j) {
                const float v = xs[j * F + f];
                a0 += r0[j] * v; a1 += r1[j] * v; a2 += r2[j] * v; a3 += r3[j] * v;
            }
            yb[(long long)i0 * ldy + F + f] = a0;
            if (i0 + 1 < N) yb[(long long)(i0 + 1) * ldy + F + f] = a1;
            if (i0 + 2 < N) yb[(long long)(i0 + 2) * ldy + F + f] = a2;
            if (i0 + 3 < N) yb[(long long)(i0 + 3) * ldy + F + f] = a3;
        }
    }
    for (int i = threadIdx.x; i < N * ldy; i += 256) {
        const int r = i / ldy, c = i - r * ldy;
        if (c < F) yb[i] = xs[r * F + c];
        else if (c >= 2 * F) yb[i] = 0.f;
    }
}

__global__ __launch_bounds__(256) void copy_cols_kernel(const float* __restrict__ x, int ldx, float* __restrict__ y,
                                                        int ldy, int col_off, int C, long long rows, int act,
                                                        float slope) {
    const long long total = rows * C;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / C;
        const int c = (int)(i - r * C);
        float v = x[r * ldx + c];
        if (act == MFT_ACT_RELU) v = fmaxf(v, 0.f);
        else if (act == MFT_ACT_LRELU) v = v > 0.f ? v : v * slope;
        y[r * ldy + col_off + c] = v;
    }
}


// nodes[((e*nq+i)*n_way+c)*(ns+1)+s] = cat(z[e, c, s<ns ? s : ns+i, :], onehot(c) if s<ns else 0), zero padded to ld.
// fold != 0: the 2*ns supports per class are averaged pairwise (k with k+ns) first (gnnnet_copy.py:67-72).
__global__ __launch_bounds__(256) void build_nodes_kernel(const float* __restrict__ z, int zf, float* __restrict__ nodes,
                                                          int ld, int n_ep, int n_way, int ns, int nq, int fold) {
    const long long rows = (long long)n_ep * nq * n_way * (ns + 1);
    const int per_class = (fold ? 2 * ns : ns) + nq;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < rows * ld;
         i += (long long)gridDim.x * blockDim.x) {
        const int col = (int)(i % ld);
        long long r = i / ld;
        const int s = (int)(r % (ns + 1)); r /= (ns + 1);
        const int c = (int)(r % n_way); r /= n_way;
        const int q = (int)(r % nq);
        const int e = (int)(r / nq);
        float v = 0.f;
        const float* zc = z + ((long long)e * n_way + c) * per_class * zf;
        if (col < zf) {
            if (s < ns) v = fold ? 0.5f * (zc[(long long)s * zf + col] + zc[(long long)(s + ns) * zf + col]) : zc[(long long)s * zf + col];
            else v = zc[(long long)((fold ? 2 * ns : ns) + q) * zf + col];
        } else if (col < zf + n_way) {
            v = (s < ns && col - zf == c) ? 1.f : 0.f;
        }
        nodes[i] = v;
    }
}

// scores[e, c*nq+q, :] = out[((e*nq+q)*n_way+c)*(ns+1)+ns, :n_way]   (gnnnet.py:215-216)
__global__ void gather_scores_kernel(const float* __restrict__ out, int ldo, float* __restrict__ scores, int n_ep,
                                     int n_way, int ns, int nq) {
    const long long total = (long long)n_ep * n_way * nq * n_way;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int k = (int)(i % n_way);
    long long r = i / n_way;
    const int q = (int)(r % nq); r /= nq;
    const int c = (int)(r % n_way);
    const int e = (int)(r / n_way);
    const long long node = (((long long)e * nq + q) * n_way + c) * (ns + 1) + ns;
    scores[i] = out[node * ldo + k];
}

__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ src, const int* __restrict__ idx,
                                                          float* __restrict__ dst, int n_rows, long long row_f4) {
    const long long total = (long long)n_rows * row_f4;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / row_f4;
        const long long c = i - r * row_f4;
        ((f32x4*)dst)[i] = ((const f32x4*)src)[(long long)idx[r] * row_f4 + c];
    }
}

inline int ggrid(long long total) {
    long long b = (total + 255) / 256;
    if (b < 1) b = 1;
    if (b > 4096) b = 4096;
    return (int)b;
}

// Skinny GEMM of the GNN head's linear layers in a meta-training step (gnnnet.py:44 fc 512 -> 128 on 105 feature rows; gnn.py:134-166
// Gconv 2F -> 48 / n_way on 480 node rows): out[m][n] = sum_k a[m][k] w[n][k] + bias[n].  The tile kernel of csrc/conv_igemm.hip
// runs these on 4-8 workgroups that walk 9-16 K-steps back to back: 16-21 us of exposed latency each.  Register-K form (as
// pair_mlp_layer_rk_kernel, csrc/pair_mlp.hip): a workgroup owns 16 rows x 16 CB columns over the whole K (<= 512); its four waves
// take the 16-wide K units u = wave, wave + 4, ...; lane (r, q) feeds row r and k = 16 u + 4 q + e of v_mfma_f32_16x16x4_f32 -- every
// load of the tile is issued at once, no LDS staging; the four partial tiles are added in wave order in LDS and wave c finishes
// column block c.
template <int NV, int CB>
__global__ __launch_bounds__(256) void gemm_rk_kernel(const float* __restrict__ a, int lda, const float* __restrict__ w, int w_rows,
                                                      int K, const float* __restrict__ bias, float* __restrict__ out, int ldo, int M,
                                                      int N, int tiles_n) {
    __shared__ float s_acc[4][CB][4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, q = lane >> 4;
    const int nt = blockIdx.x % tiles_n, mt = blockIdx.x / tiles_n;
    const int m0 = mt * 16, n0 = nt * 16 * CB;
    const int units = K >> 4;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 ra[NV], rb[CB][NV];
    {
        const int m = m0 + r;
        const float* ap = a + (long long)(m < M ? m : 0) * lda + 4 * q;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int u = wave + 4 * v;
            ra[v] = (m < M && u < units) ? *(const f32x4*)(ap + 16 * u) : zero4;
        }
    }
#pragma unroll
    for (int c = 0; c < CB; ++c) {
        const int n = n0 + c * 16 + r;
        const float* bp = w + (long long)(n < w_rows ? n : 0) * K + 4 * q;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int u = wave + 4 * v;
            rb[c][v] = (n < w_rows && u < units) ? *(const f32x4*)(bp + 16 * u) : zero4;
        }
    }
    f32x4 acc[CB];
#pragma unroll
    for (int c = 0; c < CB; ++c) acc[c] = zero4;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        if (wave + 4 * v < units) {                    // wave-uniform
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int c = 0; c < CB; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(ra[v][e], rb[c][v][e], acc[c], 0, 0, 0);
        }
    }
#pragma unroll
    for (int c = 0; c < CB; ++c)
#pragma unroll
        for (int e = 0; e < 4; ++e) s_acc[wave][c][e][lane] = acc[c][e];
    __syncthreads();
    if (wave >= CB) return;
    // column block `wave`: C/D layout of a 16 x 16 block: col = lane & 15, row = 4 * (lane >> 4) + e
    const int n = n0 + wave * 16 + r;
    const float bv = (bias && n < N) ? bias[n] : 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float t = ((s_acc[0][wave][e][lane] + s_acc[1][wave][e][lane]) + s_acc[2][wave][e][lane]) + s_acc[3][wave][e][lane];
        const int m = m0 + 4 * q + e;
        if (m < M && n < N) out[(long long)m * ldo + n] = t + bv;
    }
}

template <int CB>
int gemm_rk_launch(int nv, unsigned nwg, hipStream_t st, const float* a, int lda, const float* w, int w_rows, int K, const float* bias,
                   float* out, int ldo, int M, int N, int tiles_n) {
    switch (nv) {
#define MFT_GRK_CASE(V) \
    case V: hipLaunchKernelGGL((gemm_rk_kernel<V, CB>), dim3(nwg), dim3(256), 0, st, a, lda, w, w_rows, K, bias, out, ldo, M, N, tiles_n); break;
        MFT_GRK_CASE(1) MFT_GRK_CASE(2) MFT_GRK_CASE(3) MFT_GRK_CASE(4) MFT_GRK_CASE(5) MFT_GRK_CASE(6) MFT_GRK_CASE(7) MFT_GRK_CASE(8)
#undef MFT_GRK_CASE
    default: return MFT_EINVAL;
    }
    return mft_launch_status();
}

}  // namespace

extern "C" int mft_pair_absdiff(const float* x, int ldx, float* d, int ldd, int n_graphs, int N, int F, void* stream) {
    if (ldd % 4 != 0 || ldx < ldd) return MFT_EINVAL;   // reads up to column ldd-1 of x (finite padding)
    const long long total = (long long)n_graphs * N * N * (ldd / 4);
    hipLaunchKernelGGL(pair_absdiff_kernel, dim3(ggrid(total)), dim3(256), 0, (hipStream_t)stream, x, ldx, d, ldd,
                       n_graphs, N, F);
    return mft_launch_status();
}

extern "C" int mft_masked_softmax(const float* s, int lds_, float* A, int n_graphs, int N, void* stream) {
    const long long rows = (long long)n_graphs * N;
    hipLaunchKernelGGL(masked_softmax_kernel, dim3((int)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, s, lds_,
                       A, n_graphs, N);
    return mft_launch_status();
}

/* out[m][n] = sum_k a[m][k] * w[n][k] + bias[n]  (m < M, n < N) -- the skinny register-K form for the head's linear layers in a
 * meta-training step (see gemm_rk_kernel).  w: packed [w_rows >= N][K] (rows beyond N are read as zero padding when present), K % 16
 * == 0, K <= 512, lda % 4 == 0, lda >= K.  Columns N..ldo-1 of `out` are not written. */
extern "C" int mft_gemm_rk(const float* a, int lda, const float* w, int w_rows, int K, const float* bias, float* out, int ldo, int M,
                           int N, void* stream) {
    if (M < 1 || N < 1 || K < 16 || K % 16 != 0 || K > 512 || lda % 4 != 0 || lda < K || w_rows < N || ldo < N ||
        ((unsigned long long)a & 15) != 0 || ((unsigned long long)w & 15) != 0)        // 16-byte operand loads
        return MFT_EINVAL;
    const int cbs = (N + 15) / 16;
    const int CB = cbs >= 4 ? 4 : cbs;
    const int tiles_n = (N + 16 * CB - 1) / (16 * CB);
    const long long nwg = (long long)((M + 15) / 16) * tiles_n;
    if (nwg > 0x7fffffffLL) return MFT_EINVAL;
    const int nv = (K / 16 + 3) / 4;
    hipStream_t st = (hipStream_t)stream;
    switch (CB) {
    case 1: return gemm_rk_launch<1>(nv, (unsigned)nwg, st, a, lda, w, w_rows, K, bias, out, ldo, M, N, tiles_n);
    case 2: return gemm_rk_launch<2>(nv, (unsigned)nwg, st, a, lda, w, w_rows, K, bias, out, ldo, M, N, tiles_n);
    case 3: return gemm_rk_launch<3>(nv, (unsigned)nwg, st, a, lda, w, w_rows, K, bias, out, ldo, M, N, tiles_n);
    default: return gemm_rk_launch<4>(nv, (unsigned)nwg, st, a, lda, w, w_rows, K, bias, out, ldo, M, N, tiles_n);
    }
}

extern "C" int mft_graph_aggregate(const float* A, const float* x, int ldx, float* y, int ldy, int n_graphs, int N,
                                   int F, void* stream) {
    if (ldy < 2 * F) return MFT_EINVAL;
    const size_t lds = (size_t)N * F * sizeof(float);
    if (n_graphs >= 128 && lds <= 150 * 1024) {              // enough graphs to fill the chip with one workgroup each
        static MftPerDeviceOnce attr_once;
        if (attr_once.need()) {
            hipError_t e = hipFuncSetAttribute((const void*)graph_aggregate_lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                               150 * 1024);
            if (e != hipSuccess) return (int)e;
            attr_once.mark();
        }
        hipLaunchKernelGGL(graph_aggregate_lds_kernel, dim3(n_graphs), dim3(256), lds, (hipStream_t)stream, A, x, ldx, y, ldy, N, F);
        return mft_launch_status();
    }
    hipLaunchKernelGGL(graph_aggregate_kernel, dim3(n_graphs * N), dim3(256), 0, (hipStream_t)stream, A, x, ldx, y,
                       ldy, n_graphs, N, F);
    return mft_launch_status();
}

extern "C" int mft_copy_cols(const float* x, int ldx, float* y, int ldy, int col_off, int C, int rows, int act,
                             float slope, void* stream) {
    hipLaunchKernelGGL(copy_cols_kernel, dim3(ggrid((long long)rows * C)), dim3(256), 0, (hipStream_t)stream, x, ldx,
                       y, ldy, col_off, C, (long long)rows, act, slope);
    return mft_launch_status();
}

extern "C" int mft_build_graph_nodes(const float* z, int zf, float* nodes, int ld, int n_episodes, int n_way,
                                     int n_support, int n_query, int fold, void* stream) {
    if (ld < zf + n_way) return MFT_EINVAL;
    const long long total = (long long)n_episodes * n_query * n_way * (n_support + 1) * ld;
    hipLaunchKernelGGL(build_nodes_kernel, dim3(ggrid(total)), dim3(256), 0, (hipStream_t)stream, z, zf, nodes, ld,
                       n_episodes, n_way, n_support, n_query, fold);
    return mft_launch_status();
}

extern "C" int mft_gather_query_scores(const float* out, int ldo, float* scores, int n_episodes, int n_way,
                                       int n_support, int n_query, void* stream) {
    const long long total = (long long)n_episodes * n_way * n_query * n_way;
    hipLaunchKernelGGL(gather_scores_kernel, dim3((int)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, out,
                       ldo, scores, n_episodes, n_way, n_support, n_query);
    return mft_launch_status();
}

extern "C" int mft_gather_rows(const float* src, const int* idx, float* dst, int n_rows, long long row_floats,
                               void* stream) {
    if (row_floats % 4 != 0) return MFT_EINVAL;
    hipLaunchKernelGGL(gather_rows_kernel, dim3(ggrid((long long)n_rows * (row_floats / 4))), dim3(256), 0,
                       (hipStream_t)stream, src, idx, dst, n_rows, row_floats / 4);
    return mft_launch_status();
}
