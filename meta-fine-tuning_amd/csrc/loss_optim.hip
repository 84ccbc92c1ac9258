// Cross-entropy (forward + backward), row softmax, fused Adam / SGD updates and the first-order-MAML
// weight bookkeeping.  HBM-bound streaming kernels (float4 per lane, grid-stride).
//
// Replaces nn.CrossEntropyLoss + autograd (finetune.py:291-293; gnnnet.py:170-174,219-231),
// F.softmax (finetune.py:317), torch.optim.Adam.step (finetune.py:255,299; gnnnet.py:128,177; train.py:28),
// torch.optim.SGD.step (meta_template.py:166) and GnnNet.MAML_update (gnnnet.py:90-103).
#include "mft_common.h"

namespace {

// one wave per row; C <= 4096
__global__ __launch_bounds__(256) void cross_entropy_kernel(const float* __restrict__ logits, int ld,
                                                            const int* __restrict__ labels, int C,
                                                            int rows_per_group, int n_groups,
                                                            float* __restrict__ row_loss,
                                                            float* __restrict__ dlogits) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long long rows = (long long)rows_per_group * n_groups;
    if (row >= rows) return;
    const float* x = logits + row * ld;
    float mx = -3.4e38f;
    for (int c = lane; c < C; c += 64) mx = fmaxf(mx, x[c]);
    mx = wave_max(mx);
    float se = 0.f;
    for (int c = lane; c < C; c += 64) se += __expf(x[c] - mx);
    se = wave_sum(se);
    const int y = labels[row];
    const float lse = mx + __logf(se);
    if (lane == 0) row_loss[row] = lse - x[y];
    if (dlogits) {
        const float inv = 1.f / (float)rows_per_group;
        float* d = dlogits + row * ld;
        for (int c = lane; c < C; c += 64) {
            float pr = __expf(x[c] - lse);
            d[c] = (pr - (c == y ? 1.f : 0.f)) * inv;
        }
    }
}

__global__ void group_mean_kernel(const float* __restrict__ row_loss, int rows_per_group, int n_groups,
                                  float* __restrict__ loss) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n_groups) return;
    float s = 0.f;
    for (int r = 0; r < rows_per_group; ++r) s += row_loss[(long long)g * rows_per_group + r];
    loss[g] = s / (float)rows_per_group;
}

__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ x, int ldx,
                                                           float* __restrict__ y, int ldy, int C, int rows) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + row * ldx;
    float mx = -3.4e38f;
    for (int c = lane; c < C; c += 64) mx = fmaxf(mx, xr[c]);
    mx = wave_max(mx);
    float se = 0.f;
    for (int c = lane; c < C; c += 64) se += __expf(xr[c] - mx);
    se = wave_sum(se);
    const float inv = 1.f / se;
    for (int c = lane; c < C; c += 64) y[row * ldy + c] = __expf(xr[c] - mx) * inv;
}

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, long long n4,
                                                   long long n, float step_size, float inv_sqrt_bc2, float b1,
                                                   float b2, float eps, float wd, const float* __restrict__ hyper) {
    if (hyper) { step_size = hyper[0]; inv_sqrt_bc2 = hyper[1]; }
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4;
         i += (long long)gridDim.x * blockDim.x) {
        f32x4 pp = *(const f32x4*)(p + 4 * i);
        f32x4 gg = *(const f32x4*)(g + 4 * i);
        f32x4 mm = *(const f32x4*)(m + 4 * i);
        f32x4 vv = *(const f32x4*)(v + 4 * i);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float ge = gg[e] + wd * pp[e];
            mm[e] = b1 * mm[e] + (1.f - b1) * ge;
            vv[e] = b2 * vv[e] + (1.f - b2) * ge * ge;
            const float denom = sqrtf(vv[e]) * inv_sqrt_bc2 + eps;
            pp[e] -= step_size * (mm[e] / denom);
        }
        *(f32x4*)(p + 4 * i) = pp;
        *(f32x4*)(m + 4 * i) = mm;
        *(f32x4*)(v + 4 * i) = vv;
    }
    // tail (n % 4)
    const long long t = 4 * n4 + (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (blockIdx.x == 0 && t < n) {
        float ge = g[t] + wd * p[t];
        float mm = b1 * m[t] + (1.f - b1) * ge;
        float vv = b2 * v[t] + (1.f - b2) * ge * ge;
        m[t] = mm;
        v[t] = vv;
        p[t] -= step_size * (mm / (sqrtf(vv) * inv_sqrt_bc2 + eps));
    }
}

__global__ __launch_bounds__(256) void sgd_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                  float* __restrict__ buf, long long n, int first, float lr,
                                                  float mom, float damp, float wd) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x) {
        const float ge = g[i] + wd * p[i];
        const float b = first ? ge : mom * buf[i] + (1.f - damp) * ge;
        buf[i] = b;
        p[i] -= lr * b;
    }
}

__global__ __launch_bounds__(256) void maml_delta_kernel(float* __restrict__ p, const float* __restrict__ p2,
                                                         const float* __restrict__ p3, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x)
        p[i] = p[i] - (p3[i] - p2[i]);
}

inline int sgrid(long long total) {
    long long b = (total + 255) / 256;
    if (b < 1) b = 1;
    if (b > 2048) b = 2048;
    return (int)b;
}

}  // namespace

extern "C" int mft_cross_entropy(const float* logits, int ld, const int* labels, int C, int rows_per_group,
                                 int n_groups, float* loss, float* dlogits, void* stream) {
    // loss doubles as scratch: needs rows_per_group*n_groups + n_groups floats; layout [n_groups | rows]
    hipStream_t s = (hipStream_t)stream;
    const long long rows = (long long)rows_per_group * n_groups;
    float* row_loss = loss + n_groups;
    hipLaunchKernelGGL(cross_entropy_kernel, dim3((int)((rows + 3) / 4)), dim3(256), 0, s, logits, ld, labels, C,
                       rows_per_group, n_groups, row_loss, dlogits);
    hipLaunchKernelGGL(group_mean_kernel, dim3((n_groups + 63) / 64), dim3(64), 0, s, row_loss, rows_per_group,
                       n_groups, loss);
    return mft_launch_status();
}

// ---- nn.CrossEntropyLoss(reduction='mean') of the meta-training / pre-training losses (gnnnet.py:219-231: [80,5] scores;
// baselinetrain.py:38-45: [16, num_class]) as ONE launch forward and ONE backward: labels are read as torch hands them over
// (int64, or int32), the mean is formed in a fixed order inside one workgroup (rerun- and replay-identical), and the backward
// takes the upstream gradient as a DEVICE scalar (what autograd passes; a hipGraph replay cannot bake a host value in).
namespace {
__device__ __forceinline__ long long ce_label(const void* labels, int i64, long long row) {
    return i64 ? ((const long long*)labels)[row] : (long long)((const int*)labels)[row];
}

__global__ __launch_bounds__(256) void ce_mean_kernel(const float* __restrict__ logits, int ld, const void* __restrict__ labels,
                                                      int i64, int C, int rows, float* __restrict__ loss, double* loss_sum) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __shared__ float part[256];
    float acc = 0.f;
    if (C <= 16) {
        // few classes (the episode loss: 5): a thread per row -- every row's loads are in flight together instead of 20 dependent
        // row visits per wave (20.6 us for [80, 5] in round 6's first form)
        for (int row = threadIdx.x; row < rows; row += 256) {
            const float* x = logits + (long long)row * ld;
            float v[16];
            float mx = -3.4e38f;
#pragma unroll
            for (int c = 0; c < 16; ++c) { v[c] = c < C ? x[c] : -3.4e38f; mx = fmaxf(mx, v[c]); }
            float se = 0.f;
#pragma unroll
            for (int c = 0; c < 16; ++c) se += c < C ? __expf(v[c] - mx) : 0.f;
            const long long y = ce_label(labels, i64, row);
            const float xy = (y >= 0 && y < C) ? x[y] : __builtin_nanf("");      // an out-of-range label poisons the loss (torch asserts)
            acc += (mx + __logf(se)) - xy;
        }
    } else {                                      // a wave per row (row = wave, wave + 4, ...); lane 0 carries the wave's sum
        for (int row = wave; row < rows; row += 4) {
            const float* x = logits + (long long)row * ld;
            float mx = -3.4e38f;
            for (int c = lane; c < C; c += 64) mx = fmaxf(mx, x[c]);
            mx = wave_max(mx);
            float se = 0.f;
            for (int c = lane; c < C; c += 64) se += __expf(x[c] - mx);
            se = wave_sum(se);
            const long long y = ce_label(labels, i64, row);
            const float xy = (y >= 0 && y < C) ? x[y] : __builtin_nanf("");
            if (lane == 0) acc += (mx + __logf(se)) - xy;
        }
    }
    part[threadIdx.x] = acc;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {     // fixed tree: rerun- and replay-identical
        if (threadIdx.x < off) part[threadIdx.x] += part[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float l = part[0] / (float)rows;
        loss[0] = l;
        if (loss_sum) *loss_sum += (double)l;         // the episode loop's running loss (meta_template.py:91: avg_loss + loss.item())
    }
}

// dlogits[row, c] = (softmax(row)[c] - [c == y]) * gout / rows; one wave per row
__global__ __launch_bounds__(256) void ce_mean_bwd_kernel(const float* __restrict__ logits, int ld, const void* __restrict__ labels,
                                                          int i64, int C, int rows, const float* __restrict__ gout,
                                                          float* __restrict__ dlogits, int ldd) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* x = logits + row * ld;
    float mx = -3.4e38f;
    for (int c = lane; c < C; c += 64) mx = fmaxf(mx, x[c]);
    mx = wave_max(mx);
    float se = 0.f;
    for (int c = lane; c < C; c += 64) se += __expf(x[c] - mx);
    se = wave_sum(se);
    const float lse = mx + __logf(se);
    const long long y = ce_label(labels, i64, row);
    const float sc = (gout ? gout[0] : 1.f) / (float)rows;
    float* d = dlogits + row * ldd;
    for (int c = lane; c < C; c += 64) d[c] = (__expf(x[c] - lse) - (c == y ? 1.f : 0.f)) * sc;
}
}  // namespace

extern "C" int mft_cross_entropy_mean(const float* logits, int ld, const void* labels, int labels_i64, int C, int rows, float* loss,
                                      double* loss_sum, void* stream) {
    if (rows < 1 || C < 1 || ld < C) return MFT_EINVAL;
    hipLaunchKernelGGL(ce_mean_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, logits, ld, labels, labels_i64, C, rows, loss,
                       loss_sum);
    return mft_launch_status();
}

extern "C" int mft_cross_entropy_mean_backward(const float* logits, int ld, const void* labels, int labels_i64, int C, int rows,
                                               const float* grad_loss, float* dlogits, int ldd, void* stream) {
    if (rows < 1 || C < 1 || ld < C || ldd < C) return MFT_EINVAL;
    hipLaunchKernelGGL(ce_mean_bwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, logits, ld, labels, labels_i64, C,
                       rows, grad_loss, dlogits, ldd);
    return mft_launch_status();
}

extern "C" int mft_softmax_rows(const float* x, int ldx, float* y, int ldy, int C, int rows, void* stream) {
    hipLaunchKernelGGL(softmax_rows_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, ldx, y, ldy, C,
                       rows);
    return mft_launch_status();
}

extern "C" int mft_adam_step(float* p, const float* g, float* m, float* v, long long n, int step, float lr,
                             float beta1, float beta2, float eps, float weight_decay, void* stream) {
    if (step < 1) return MFT_EINVAL;
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    const float step_size = (float)((double)lr / bc1);
    const float inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
    hipLaunchKernelGGL(adam_kernel, dim3(sgrid(n / 4)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n / 4, n,
                       step_size, inv_sqrt_bc2, beta1, beta2, eps, weight_decay, (const float*)nullptr);
    return mft_launch_status();
}

namespace {
// hyper[0] = lr / (1 - b1^t), hyper[1] = 1 / sqrt(1 - b2^t) for t = ++(*step): the step counter lives on the device so that
// a captured graph of one inner step can be replayed for every step (kernel arguments are frozen at capture time)
__global__ void adam_hyper_advance_kernel(int* step, float* hyper, float lr, float b1, float b2) {
    const int t = *step + 1;
    *step = t;
    const double bc1 = 1.0 - pow((double)b1, (double)t);
    const double bc2 = 1.0 - pow((double)b2, (double)t);
    hyper[0] = (float)((double)lr / bc1);
    hyper[1] = (float)(1.0 / sqrt(bc2));
}
}  // namespace

extern "C" int mft_adam_hyper_advance(int* step, float* hyper, float lr, float beta1, float beta2, void* stream) {
    hipLaunchKernelGGL(adam_hyper_advance_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, step, hyper, lr, beta1, beta2);
    return mft_launch_status();
}

extern "C" int mft_adam_step_dev(float* p, const float* g, float* m, float* v, long long n, const float* hyper,
                                 float beta1, float beta2, float eps, float weight_decay, void* stream) {
    if (hyper == nullptr) return MFT_EINVAL;
    hipLaunchKernelGGL(adam_kernel, dim3(sgrid(n / 4)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n / 4, n, 0.f, 0.f,
                       beta1, beta2, eps, weight_decay, hyper);
    return mft_launch_status();
}

extern "C" int mft_sgd_step(float* p, const float* g, float* buf, long long n, int first_step, float lr,
                            float momentum, float dampening, float weight_decay, void* stream) {
    hipLaunchKernelGGL(sgd_kernel, dim3(sgrid(n)), dim3(256), 0, (hipStream_t)stream, p, g, buf, n, first_step, lr,
                       momentum, dampening, weight_decay);
    return mft_launch_status();
}

extern "C" int mft_maml_delta(float* p, const float* p2, const float* p3, long long n, void* stream) {
    hipLaunchKernelGGL(maml_delta_kernel, dim3(sgrid(n)), dim3(256), 0, (hipStream_t)stream, p, p2, p3, n);
    return mft_launch_status();
}

namespace {
__global__ void var_to_rstd_kernel(const float* __restrict__ var, float* __restrict__ rstd, int n, float eps) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) rstd[i] = 1.0f / sqrtf(var[i] + eps);
}
}  // namespace

extern "C" int mft_var_to_rstd(const float* var, float* rstd, int n, float eps, void* stream) {
    hipLaunchKernelGGL(var_to_rstd_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, var, rstd, n, eps);
    return mft_launch_status();
}

// ------------------------------------------------------------------------------------------ linear head
// finetune_linear's classifier (finetune.py:33-42,65,103,147-158): per episode a Linear(D, n_way) on the backbone
// feature; one inner step = logits -> cross entropy -> {d feature (pre-update weights), dW, db} -> Adam with L2
// weight decay on (W, b).  Everything for one episode fits one workgroup, so the whole classifier step is ONE
// launch over the E episodes: no logits / dlogits / gradient tensors in HBM.  rows_per_group <= 16, n_way <= 16.
namespace {

constexpr int LH_MAXR = 16, LH_MAXC = 16;

__global__ __launch_bounds__(256) void linear_head_step_kernel(
    const float* __restrict__ feat, int ldf, const int* __restrict__ labels, int k, int n_way, int D,
    float* __restrict__ W, float* __restrict__ b, float* __restrict__ mW, float* __restrict__ vW,
    float* __restrict__ mb, float* __restrict__ vb, float* __restrict__ dfeat, int lddf, float* __restrict__ loss,
    float step_size, float inv_sqrt_bc2, float b1, float b2, float eps, float wd) {
    const int g = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    __shared__ float s_logit[LH_MAXR][LH_MAXC];
    __shared__ float s_d[LH_MAXR][LH_MAXC];
    __shared__ float s_loss[LH_MAXR];
    const float* F = feat + (long long)g * k * ldf;
    float* Wg = W + (long long)g * n_way * D;
    float* bg = b + (long long)g * n_way;
    // logits: one wave per (row, class) pair
    for (int pr = wave; pr < k * n_way; pr += 4) {
        const int r = pr / n_way, c = pr - r * n_way;
        float s = 0.f;
        for (int d = lane; d < D; d += 64) s += F[(long long)r * ldf + d] * Wg[(long long)c * D + d];
        s = wave_sum(s);
        if (lane == 0) s_logit[r][c] = s + bg[c];
    }
    __syncthreads();
    if (tid < k) {
        const int r = tid;
        float mx = -3.4e38f;
        for (int c = 0; c < n_way; ++c) mx = fmaxf(mx, s_logit[r][c]);
        float se = 0.f;
        for (int c = 0; c < n_way; ++c) se += __expf(s_logit[r][c] - mx);
        const float lse = mx + __logf(se);
        const int y = labels[(long long)g * k + r];
        s_loss[r] = lse - s_logit[r][y];
        const float inv = 1.f / (float)k;
        for (int c = 0; c < n_way; ++c) s_d[r][c] = (__expf(s_logit[r][c] - lse) - (c == y ? 1.f : 0.f)) * inv;
    }
    __syncthreads();
    if (tid == 0 && loss) {
        float s = 0.f;
        for (int r = 0; r < k; ++r) s += s_loss[r];
        loss[g] = s / (float)k;
    }
    // d feature with the pre-update weights
    for (int i = tid; i < k * D; i += 256) {
        const int r = i / D, d = i - r * D;
        float s = 0.f;
        for (int c = 0; c < n_way; ++c) s += s_d[r][c] * Wg[(long long)c * D + d];
        dfeat[((long long)g * k + r) * lddf + d] = s;
    }
    __syncthreads();
    // dW, db -> Adam (weight decay folded into the gradient, torch.optim.Adam semantics)
    float* mWg = mW + (long long)g * n_way * D;
    float* vWg = vW + (long long)g * n_way * D;
    for (int i = tid; i < n_way * D; i += 256) {
        const int c = i / D, d = i - c * D;
        float gr = 0.f;
        for (int r = 0; r < k; ++r) gr += s_d[r][c] * F[(long long)r * ldf + d];
        const float w = Wg[i];
        gr += wd * w;
        const float m = b1 * mWg[i] + (1.f - b1) * gr;
        const float v = b2 * vWg[i] + (1.f - b2) * gr * gr;
        mWg[i] = m;
        vWg[i] = v;
        Wg[i] = w - step_size * (m / (sqrtf(v) * inv_sqrt_bc2 + eps));
    }
    if (tid < n_way) {
        const int c = tid;
        float gr = 0.f;
        for (int r = 0; r < k; ++r) gr += s_d[r][c];
        const float w = bg[c];
        gr += wd * w;
        const long long i = (long long)g * n_way + c;
        const float m = b1 * mb[i] + (1.f - b1) * gr;
        const float v = b2 * vb[i] + (1.f - b2) * gr * gr;
        mb[i] = m;
        vb[i] = v;
        bg[c] = w - step_size * (m / (sqrtf(v) * inv_sqrt_bc2 + eps));
    }
}

// scores = softmax(feat @ W[g]^T + b[g]); one wave per row
__global__ __launch_bounds__(256) void linear_head_scores_kernel(const float* __restrict__ feat, int ldf,
                                                                 int rows_per_group, int n_groups, int n_way, int D,
                                                                 const float* __restrict__ W,
                                                                 const float* __restrict__ b,
                                                                 float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= (long long)rows_per_group * n_groups) return;
    const int g = (int)(row / rows_per_group);
    const float* f = feat + row * ldf;
    float lg[LH_MAXC];
    float mx = -3.4e38f;
    for (int c = 0; c < n_way; ++c) {
        float s = 0.f;
        for (int d = lane; d < D; d += 64) s += f[d] * W[((long long)g * n_way + c) * D + d];
        s = wave_sum(s) + b[(long long)g * n_way + c];
        lg[c] = s;
        mx = fmaxf(mx, s);
    }
    float se = 0.f;
    for (int c = 0; c < n_way; ++c) se += __expf(lg[c] - mx);
    if (lane < n_way) out[row * n_way + lane] = __expf(lg[lane] - mx) / se;
}

}  // namespace

extern "C" int mft_linear_head_step(const float* feat, int ldf, const int* labels, int rows_per_group, int n_groups,
                                    int n_way, int D, float* W, float* b, float* mW, float* vW, float* mb, float* vb,
                                    float* dfeat, int lddf, float* loss, int step, float lr, float beta1, float beta2,
                                    float eps, float weight_decay, void* stream) {
    if (step < 1 || rows_per_group < 1 || rows_per_group > LH_MAXR || n_way < 1 || n_way > LH_MAXC) return MFT_EINVAL;
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    hipLaunchKernelGGL(linear_head_step_kernel, dim3(n_groups), dim3(256), 0, (hipStream_t)stream, feat, ldf, labels,
                       rows_per_group, n_way, D, W, b, mW, vW, mb, vb, dfeat, lddf, loss, (float)((double)lr / bc1),
                       (float)(1.0 / sqrt(bc2)), beta1, beta2, eps, weight_decay);
    return mft_launch_status();
}

extern "C" int mft_linear_head_scores(const float* feat, int ldf, int rows_per_group, int n_groups, int n_way, int D,
                                      const float* W, const float* b, float* out, void* stream) {
    if (n_way < 1 || n_way > LH_MAXC) return MFT_EINVAL;
    const long long rows = (long long)rows_per_group * n_groups;
    hipLaunchKernelGGL(linear_head_scores_kernel, dim3((int)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, feat,
                       ldf, rows_per_group, n_groups, n_way, D, W, b, out);
    return mft_launch_status();
}

// ------------------------------------------------------------------------------------- fused CE + pool backward
// Inner-loop loss on the pooled feature (finetune.py:286-293): per group, cross entropy over the C-wide feature rows,
// dlogits = (softmax - onehot)/rows, and straight on through AvgPool + the block's final ReLU:
// d_out[img*hw + p][c] = out > 0 ? dlogits[img][c] / hw : 0.  One workgroup per group; replaces three launches.
namespace {
__global__ __launch_bounds__(256) void ce_pool_backward_kernel(const float* __restrict__ feat, const int* __restrict__ labels,
                                                               int k, int C, int hw, const float* __restrict__ out,
                                                               float* __restrict__ d_out, float* __restrict__ loss) {
    const int g = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __shared__ float s_lse[16];
    __shared__ float s_loss[16];
    for (int r = wave; r < k; r += 4) {
        const float* x = feat + ((long long)g * k + r) * C;
        float mx = -3.4e38f;
        for (int c = lane; c < C; c += 64) mx = fmaxf(mx, x[c]);
        mx = wave_max(mx);
        float se = 0.f;
        for (int c = lane; c < C; c += 64) se += __expf(x[c] - mx);
        se = wave_sum(se);
        if (lane == 0) {
            const float lse = mx + __logf(se);
            s_lse[r] = lse;
            s_loss[r] = lse - x[labels[(long long)g * k + r]];
        }
    }
    __syncthreads();
    if (threadIdx.x == 0 && loss) {
        float s = 0.f;
        for (int r = 0; r < k; ++r) s += s_loss[r];
        loss[g] = s / (float)k;
    }
    const float inv = 1.f / ((float)k * (float)hw);
    const int cq = C >> 2;
    for (int i = threadIdx.x; i < k * hw * cq; i += 256) {
        const int c = (i % cq) * 4;
        const int pix = i / cq;
        const int r = pix / hw;
        const long long row = ((long long)g * k) * hw + pix;
        const f32x4 o = *(const f32x4*)(out + row * C + c);
        const f32x4 x = *(const f32x4*)(feat + ((long long)g * k + r) * C + c);
        const int y = labels[(long long)g * k + r];
        f32x4 d;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float pr = __expf(x[e] - s_lse[r]) - ((c + e) == y ? 1.f : 0.f);
            d[e] = o[e] > 0.f ? pr * inv : 0.f;
        }
        *(f32x4*)(d_out + row * C + c) = d;
    }
}
}  // namespace

extern "C" int mft_ce_pool_backward(const float* feat, const int* labels, int rows_per_group, int n_groups, int C, int hw,
                                    const float* out, float* d_out, float* loss, void* stream) {
    if (rows_per_group < 1 || rows_per_group > 16 || C % 4 != 0 || hw < 1) return MFT_EINVAL;
    hipLaunchKernelGGL(ce_pool_backward_kernel, dim3(n_groups), dim3(256), 0, (hipStream_t)stream, feat, labels,
                       rows_per_group, C, hw, out, d_out, loss);
    return mft_launch_status();
}

// ------------------------------------------------------------------------------------------ multi-tensor Adam
// torch.optim.Adam over a whole model (train.py:28: 104 parameter tensors, 5.3 M parameters) as ONE launch: the host
// passes a table of (p, g, m, v, n) chunks (<= 65536 elements each); workgroup b updates chunk b.
namespace {
struct AdamChunk { float* p; const float* g; float* m; float* v; long long n; };

__global__ __launch_bounds__(256) void adam_multi_kernel(const AdamChunk* __restrict__ table, float step_size,
                                                         float inv_sqrt_bc2, float b1, float b2, float eps, float wd, float gs) {
    const AdamChunk c = table[blockIdx.x];
    const bool vec = ((((unsigned long long)c.p | (unsigned long long)c.g | (unsigned long long)c.m | (unsigned long long)c.v) & 15ull) == 0);
    const long long n4 = vec ? (c.n >> 2) : 0;
    for (long long i = threadIdx.x; i < n4; i += 256) {
        f32x4 pp = ((const f32x4*)c.p)[i];
        const f32x4 gg = ((const f32x4*)c.g)[i];
        f32x4 mm = ((const f32x4*)c.m)[i], vv = ((const f32x4*)c.v)[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float ge = gg[e] * gs + wd * pp[e];         // gs: 1 / world size when the bucket holds the all-reduced SUM (exact at 1)
            mm[e] = b1 * mm[e] + (1.f - b1) * ge;
            vv[e] = b2 * vv[e] + (1.f - b2) * ge * ge;
            pp[e] -= step_size * (mm[e] / (sqrtf(vv[e]) * inv_sqrt_bc2 + eps));
        }
        ((f32x4*)c.m)[i] = mm;
        ((f32x4*)c.v)[i] = vv;
        ((f32x4*)c.p)[i] = pp;
    }
    for (long long i = 4 * n4 + threadIdx.x; i < c.n; i += 256) {
        const float pp = c.p[i];
        const float ge = c.g[i] * gs + wd * pp;
        const float mm = b1 * c.m[i] + (1.f - b1) * ge;
        const float vv = b2 * c.v[i] + (1.f - b2) * ge * ge;
        c.m[i] = mm;
        c.v[i] = vv;
        c.p[i] = pp - step_size * (mm / (sqrtf(vv) * inv_sqrt_bc2 + eps));
    }
}
}  // namespace

extern "C" int mft_adam_multi(const void* chunk_table, int n_chunks, int step, float lr, float beta1, float beta2, float eps,
                              float weight_decay, float grad_scale, void* stream) {
    if (step < 1 || n_chunks < 1) return MFT_EINVAL;
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    hipLaunchKernelGGL(adam_multi_kernel, dim3(n_chunks), dim3(256), 0, (hipStream_t)stream, (const AdamChunk*)chunk_table,
                       (float)((double)lr / bc1), (float)(1.0 / sqrt(bc2)), beta1, beta2, eps, weight_decay, grad_scale);
    return mft_launch_status();
}

// ------------------------------------------------------------------------------- linear-head SGD adaptation, whole run
// MetaTemplate.set_forward_adaptation / BaselineFinetune.set_forward (meta_template.py:153-186, baselinefinetune.py:17-58):
// a fresh Linear(D, n_way) trained on the frozen support features with SGD(lr .01, momentum .9, dampening .9,
// weight_decay .001) for 100 epochs of mini-batches of 4 -- 700 dependent steps of ~10 kFLOP each.  One workgroup per
// episode keeps the support features, W, b and the momentum buffers in LDS / registers and runs ALL steps in one launch.
// ADAM = true: torch.optim.Adam(lr, betas (b1, b2), eps, L2 weight_decay) instead (finetune.finetune_linear with
// freeze_backbone=True, finetune.py:110,140-160: the classifier on constant eval-mode features); `mom` / `damp` then carry
// beta1 / beta2 and the bias corrections are advanced in double precision inside the loop.
// ZLDS = false: the support features do not fit LDS beside W and its moments (20-shot: 100 rows, 50-shot: 250 rows of 512
// floats = 512 KB); they stay in HBM / L2 -- a step touches only its <= bs rows (8 KB), twice.
namespace {
template <bool ADAM, bool ZLDS>
__global__ __launch_bounds__(256) void linear_head_sgd_kernel(const float* __restrict__ z, const int* __restrict__ y,
                                                              const int* __restrict__ idx, int S, int D, int n_way, int T,
                                                              int bs, float* __restrict__ W, float* __restrict__ b, float lr,
                                                              float mom, float damp, float wd, float eps) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int g = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float* zl = sm;                              // [S][D] (ZLDS only)
    float* Ws = zl + (ZLDS ? S * D : 0);         // [n_way][D]
    float* Bw = Ws + n_way * D;                  // momentum buffer of W
    float* Vw = Bw + n_way * D;                  // ADAM: second moment of W
    float* sl = Vw + (ADAM ? n_way * D : 0);     // logits / dlogits [bs][16]
    __shared__ float bsm[16], bbuf[16], bv[16];
    double p1 = 1.0, p2 = 1.0;                   // beta1^t, beta2^t
    if (ZLDS)
        for (int i = tid; i < S * D; i += 256) zl[i] = z[(long long)g * S * D + i];
    const float* zs = ZLDS ? (const float*)zl : z + (long long)g * S * D;
    for (int i = tid; i < n_way * D; i += 256) {
        Ws[i] = W[(long long)g * n_way * D + i];
        Bw[i] = 0.f;
        if (ADAM) Vw[i] = 0.f;
    }
    if (tid < n_way) { bsm[tid] = b[(long long)g * n_way + tid]; bbuf[tid] = 0.f; bv[tid] = 0.f; }
    __syncthreads();
    const int* yg = y + (long long)g * S;
    const int* ig = idx + (long long)g * T * bs;
    for (int t = 0; t < T; ++t) {
        int k = 0;
        while (k < bs && ig[t * bs + k] >= 0) ++k;             // ragged tail: -1 padded
        for (int pr = wave; pr < k * n_way; pr += 4) {
            const int r = pr / n_way, c = pr - r * n_way;
            const float* zr = zs + ig[t * bs + r] * D;
            float s = 0.f;
            for (int d = lane; d < D; d += 64) s += zr[d] * Ws[c * D + d];
            s = wave_sum(s);
            if (lane == 0) sl[r * 16 + c] = s + bsm[c];
        }
        __syncthreads();
        if (tid < k) {
            const int r = tid;
            float mx = -3.4e38f;
            for (int c = 0; c < n_way; ++c) mx = fmaxf(mx, sl[r * 16 + c]);
            float se = 0.f;
            for (int c = 0; c < n_way; ++c) se += __expf(sl[r * 16 + c] - mx);
            const float lse = mx + __logf(se);
            const int yy = yg[ig[t * bs + r]];
            const float inv = 1.f / (float)k;
            for (int c = 0; c < n_way; ++c) sl[r * 16 + c] = (__expf(sl[r * 16 + c] - lse) - (c == yy ? 1.f : 0.f)) * inv;
        }
        __syncthreads();
        float step_size = lr, inv_sqrt_bc2 = 1.f;
        if (ADAM) {
            p1 *= (double)mom;
            p2 *= (double)damp;
            step_size = (float)((double)lr / (1.0 - p1));
            inv_sqrt_bc2 = (float)(1.0 / sqrt(1.0 - p2));
        }
        for (int i = tid; i < n_way * D; i += 256) {
            const int c = i / D, d = i - c * D;
            float gr = 0.f;
            for (int r = 0; r < k; ++r) gr += sl[r * 16 + c] * zs[ig[t * bs + r] * D + d];
            const float w = Ws[i];
            gr += wd * w;
            if (ADAM) {
                const float m1 = mom * Bw[i] + (1.f - mom) * gr;
                const float v1 = damp * Vw[i] + (1.f - damp) * gr * gr;
                Bw[i] = m1;
                Vw[i] = v1;
                Ws[i] = w - step_size * (m1 / (sqrtf(v1) * inv_sqrt_bc2 + eps));
            } else {
                const float bu = (t == 0) ? gr : mom * Bw[i] + (1.f - damp) * gr;
                Bw[i] = bu;
                Ws[i] = w - lr * bu;
            }
        }
        if (tid < n_way) {
            float gr = 0.f;
            for (int r = 0; r < k; ++r) gr += sl[r * 16 + tid];
            const float w = bsm[tid];
            gr += wd * w;
            if (ADAM) {
                const float m1 = mom * bbuf[tid] + (1.f - mom) * gr;
                const float v1 = damp * bv[tid] + (1.f - damp) * gr * gr;
                bbuf[tid] = m1;
                bv[tid] = v1;
                bsm[tid] = w - step_size * (m1 / (sqrtf(v1) * inv_sqrt_bc2 + eps));
            } else {
                const float bu = (t == 0) ? gr : mom * bbuf[tid] + (1.f - damp) * gr;
                bbuf[tid] = bu;
                bsm[tid] = w - lr * bu;
            }
        }
        __syncthreads();
    }
    for (int i = tid; i < n_way * D; i += 256) W[(long long)g * n_way * D + i] = Ws[i];
    if (tid < n_way) b[(long long)g * n_way + tid] = bsm[tid];
}
}  // namespace

template <bool ADAM>
static int linear_head_run(const float* z_support, const int* y_support, const int* idx_table, int n_groups, int n_support_rows,
                           int D, int n_way, int n_steps, int batch_size, float* W, float* b, float lr, float a, float c, float wd,
                           float eps, void* stream) {
    if (n_way < 1 || n_way > 16 || batch_size < 1 || batch_size > 16 || n_steps < 1) return MFT_EINVAL;
    const size_t head = ((ADAM ? 3 : 2) * (size_t)n_way * D + 16 * 16) * sizeof(float);
    const size_t lds_z = (size_t)n_support_rows * D * sizeof(float) + head;
    if (head > 150 * 1024) return MFT_EINVAL;
    static MftPerDeviceOnce attr_once;
    if (attr_once.need()) {
        hipError_t e = hipFuncSetAttribute((const void*)linear_head_sgd_kernel<ADAM, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           150 * 1024);
        if (e == hipSuccess)
            e = hipFuncSetAttribute((const void*)linear_head_sgd_kernel<ADAM, false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    150 * 1024);
        if (e != hipSuccess) return (int)e;
        attr_once.mark();
    }
    if (lds_z <= 150 * 1024)
        hipLaunchKernelGGL((linear_head_sgd_kernel<ADAM, true>), dim3(n_groups), dim3(256), lds_z, (hipStream_t)stream, z_support,
                           y_support, idx_table, n_support_rows, D, n_way, n_steps, batch_size, W, b, lr, a, c, wd, eps);
    else
        hipLaunchKernelGGL((linear_head_sgd_kernel<ADAM, false>), dim3(n_groups), dim3(256), head, (hipStream_t)stream, z_support,
                           y_support, idx_table, n_support_rows, D, n_way, n_steps, batch_size, W, b, lr, a, c, wd, eps);
    return mft_launch_status();
}

extern "C" int mft_linear_head_sgd_run(const float* z_support, const int* y_support, const int* idx_table, int n_groups,
                                       int n_support_rows, int D, int n_way, int n_steps, int batch_size, float* W, float* b,
                                       float lr, float momentum, float dampening, float weight_decay, void* stream) {
    return linear_head_run<false>(z_support, y_support, idx_table, n_groups, n_support_rows, D, n_way, n_steps, batch_size, W, b, lr,
                                  momentum, dampening, weight_decay, 0.f, stream);
}

extern "C" int mft_linear_head_adam_run(const float* z_support, const int* y_support, const int* idx_table, int n_groups,
                                        int n_support_rows, int D, int n_way, int n_steps, int batch_size, float* W, float* b,
                                        float lr, float beta1, float beta2, float eps, float weight_decay, void* stream) {
    return linear_head_run<true>(z_support, y_support, idx_table, n_groups, n_support_rows, D, n_way, n_steps, batch_size, W, b, lr,
                                 beta1, beta2, weight_decay, eps, stream);
}
