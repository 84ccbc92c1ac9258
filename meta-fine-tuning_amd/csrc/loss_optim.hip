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
                                                   float b2, float eps, float wd) {
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
                       step_size, inv_sqrt_bc2, beta1, beta2, eps, weight_decay);
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
