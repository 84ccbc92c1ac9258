/*
 * mft_hip.h -- C-ABI of libmft_hip.so: hand-written HIP kernels for gfx950 (MI355X)
 * that replace the ATen/cuDNN/cuBLAS dispatches on the hot path of
 * johncai117/Meta-Fine-Tuning (the reference has no native code; every entry
 * point below names the reference call site whose implicit dispatch it replaces).
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer (fp32 unless
 *     stated); `stream` is a hipStream_t passed as void*.
 *   - kernels never allocate and never synchronise; workspaces are passed in.
 *   - return value: 0 on success, otherwise the hipError_t of the failed launch,
 *     or MFT_EINVAL (-22) for an argument the kernel does not support.
 *   - activations are NHWC ("pixel-major"): a tensor [n_img, H, W, C] is a matrix of
 *     n_img*H*W rows with a row stride (`ld*`, in floats) >= C.
 *   - "group": a set of consecutive images (or rows) that forms one BatchNorm
 *     mini-batch / one episode; grouped launches run many independent episodes in
 *     one kernel (per-group statistics, optionally per-group weights).
 */
#ifndef MFT_HIP_H
#define MFT_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define MFT_EINVAL (-22)

enum { MFT_ACT_NONE = 0, MFT_ACT_RELU = 1, MFT_ACT_LRELU = 2 };
/* OR-ed into ``act`` of mft_bn_apply / mft_bn_apply_planes: y = fma(x, rstd*gamma, beta - mean*rstd*gamma) (aten's CPU form, the
 * arithmetic of the frozen trunk's folded BatchNorm kernels) instead of ((x - mean) * rstd) * gamma + beta.  The folded form
 * loses |mean| * rstd ulps to cancellation: meant for convolution outputs (|mean| ~ std), not for degenerate batches. */
enum { MFT_BN_AFFINE_FMA = 0x100 };

/* library / device ------------------------------------------------------------------ */
int mft_version(void);                        /* 100*major + minor                          */
int mft_device_info(int* cu_count, int* gcn_arch_is_gfx950);

/* CU-partitioned streams (no reference counterpart: the reference is one stream on one GPU).  mask bit i set = the
 * queue may use CU i (n_words*32 bits); mft_probe_placement writes {XCC_ID, HW_ID} of each workgroup's CU into
 * out[2*n_blocks] (used to learn the bit -> XCD mapping on the box).                                              */
int mft_stream_create_cumask(const unsigned* mask_words, int n_words, void** stream_out);
/* hipStreamCreateWithPriority with the device's full priority range (range_out = {least, greatest}; stream_out may be NULL to
 * query only).  The engine puts the frozen-trunk stream at the least and the last-block (critical path) stream at the
 * greatest priority.                                                                                                          */
int mft_stream_create_priority(int priority, void** stream_out, int* range_out);
int mft_stream_destroy(void* stream);
/* Measurement aid (no reference counterpart): one Adam-shaped 3-read / 3-write pass over three scratch arrays of n floats
 * (n a multiple of 1024) with no gradient operand -- the pure-stream rate of the memory system bench.py quotes beside the fused
 * weight-gradient + Adam kernel's.                                                                                          */
int mft_stream_probe(float* w, float* m, float* v, long long n, void* stream);
/* Measurement aid (no reference counterpart): hipEvent_t create / record on `stream` / elapsed milliseconds between two recorded
 * events (synchronises on `stop`) / destroy.  _lib.LaunchTimer brackets every launcher call with a pair of these on the stream
 * the launcher enqueues on: live per-kernel durations for bench.py's roofline objects on any stream of the engine.          */
int mft_event_create(void** event_out);
int mft_event_record(void* event, void* stream);
int mft_event_elapsed_ms(void* start, void* stop, float* ms_out);
int mft_event_destroy(void* event);
int mft_probe_placement(unsigned* out, int n_blocks, int spin_cycles, void* stream);

/* Device-side test-time views (datasets/EuroSAT_few_shot.py:145-170,240-276; data/additional_transforms.py:16-31): for every
 * (view, image) resample the crop box params[view][img][0..3] = (y0, x0, h, w) of the uint8 HWC source image to size x size
 * (bilinear), apply ImageEnhance Brightness / Contrast / Color with factors params[..][4..6] when params[..][9] != 0, flip
 * horizontally / vertically when params[..][7] / [8] != 0, scale to [0,1] and normalise with mean3 / std3 (HOST pointers).
 * Output: fp32 NHWC at out + view*view_stride + img*img_stride (floats).  Parameters come from the host sampler.          */
int mft_augment_views(const unsigned char* src, int n_img, int Hs, int Ws, const float* params, int n_views, float* out,
                      long long view_stride, long long img_stride, int size, const float* mean3, const float* std3,
                      void* stream);

/* layout ---------------------------------------------------------------------------- */
/* x.view(-1,3,H,W) NCHW -> NHWC (boundary ingest; gnnnet.py:69-79, finetune.py:210) */
int mft_nchw_to_nhwc(const float* src, float* dst, int n_img, int C, int H, int W, void* stream);

/* loss.backward() + optimizer.step() for a per-episode 3x3 / stride 1 / pad 1 convolution with ONE pass over its weights
 * (finetune.py:293-297 over backbone.py:255-256): mft_conv2d_wgrad_adam_nhwc (weight gradient + Adam on w, m, v) that also
 * multiplies every OLD weight tile with dy before it is overwritten, i.e. the data gradient without re-reading the weights.
 * dx_partials [groups, 9, rows, Cin] (rows = imgs_per_group*H*W <= 64; mft_conv2d_wgrad_adam_dgrad_ws_floats floats) holds one
 * partial per tap; mft_col2im_bn_backward_small sums them into the gradient of the convolution input and applies the
 * BatchNorm + ReLU backward of the layer in front (x_raw, relu_out, mean, rstd, gamma as in mft_bn_backward).  `_dev`: step
 * size / bias correction from a device pointer (hipGraph replay).                                                            */
long long mft_conv2d_wgrad_adam_dgrad_ws_floats(int n_img, int H, int W, int Cin);
int mft_conv2d_wgrad_adam_dgrad_nhwc(const float* x, int ldx, const float* dy, int ldy, float* w, float* m, float* v,
                                     float* dx_partials, int n_img, int H, int W, int Cin, int Cout, int imgs_per_group,
                                     long long group_stride, int step, float lr, float beta1, float beta2, float eps, void* stream);
int mft_conv2d_wgrad_adam_dgrad_nhwc_dev(const float* x, int ldx, const float* dy, int ldy, float* w, float* m, float* v,
                                         float* dx_partials, int n_img, int H, int W, int Cin, int Cout, int imgs_per_group,
                                         long long group_stride, const float* hyper, float beta1, float beta2, float eps,
                                         void* stream);
int mft_col2im_bn_backward_small(const float* dx_partials, const float* x_raw, const float* relu_out, float* dx, int n_img, int H,
                                 int W, int C, int imgs_per_group, const float* mean, const float* rstd, const float* gamma,
                                 long long gb_group_stride, float* dgamma, float* dbeta, void* stream);

/* mft_ce_pool_backward + mft_bn_backward2 in one launch (finetune.py:286-293 over backbone.py:256-261): the gradient of the
 * cross entropy on the pooled feature [n_groups*imgs_per_group, C] through AvgPool(hw) and the block's final ReLU (mask: out > 0)
 * is formed on the fly and pushed through the two BatchNorms of the residual join (main branch xa = c2, shortcut xb = sc):
 * dxa/dxb [rows, C], dgamma/dbeta [n_groups, C] each, loss [n_groups].  Bit-identical to the two launches it replaces.          */
int mft_ce_pool_bn_backward2(const float* feat, const int* labels, int imgs_per_group, int n_groups, int C, int hw,
                             const float* out, const float* xa, const float* xb, float* dxa, float* dxb, const float* mean_a,
                             const float* rstd_a, const float* gamma_a, const float* mean_b, const float* rstd_b,
                             const float* gamma_b, long long gb_group_stride, float* dgamma_a, float* dbeta_a, float* dgamma_b,
                             float* dbeta_b, float* loss, void* stream);

/* Stem cache as per-window extrema (backbone.py:295-297: BatchNorm2d -> ReLU -> MaxPool2d(3, 2, 1) behind the frozen stem
 * convolution).  mft_pool_window_minmax: x [n_img, H, W, C] -> ymax, ymin [n_img, OH, OW, C], the maximum / minimum of every
 * 3x3/stride 2/pad 1 window of the RAW convolution output.  mft_bn_relu_pooled_gather: y[n] = ReLU(BN(pmax or pmin of image
 * src_idx[n])) with the statistics of the group of n (channels with gamma >= 0 take the maximum, the others the minimum):
 * bit-identical to mft_bn_relu_maxpool_gather on the full-resolution cache because the BatchNorm affine is monotone per channel. */
/* The stem-cache fill in ONE launch (round 5, csrc/stem.hip: stem_cache_kernel): trunk.0 (backbone.py:408, 7x7 / stride 2) of n_img
 * NHWC images + per-image per-channel (mean, M2) of its output [n_img, 64] + per 3x3 / stride-2 / pad-1 window the max and min of
 * it [n_img, PH, PW, 64] -- what mft_conv2d_nhwc + mft_bn_image_moments + mft_pool_window_minmax produce, without the
 * full-resolution output ever reaching HBM.  MFT_EINVAL outside the specialised shape (84 x 84): run the three launches.     */
int mft_stem_cache_fill(const float* in, const float* w, int w_ld, int n_img, int H, int W, float* pmax, float* pmin,
                        float* mean_img, float* m2_img, void* stream);
int mft_pool_window_minmax(const float* x, float* ymax, float* ymin, long long n_img, int H, int W, int C, void* stream);
int mft_bn_relu_pooled_gather(const float* pmax, const float* pmin, const int* src_idx, float* y, int n_img, int OH, int OW, int C,
                              int imgs_per_group, const float* mean, const float* rstd, const float* gamma, const float* beta,
                              void* stream);

/* Pre-split activation planes for the bf16x3 convolutions: the frozen trunk convolutions consume every activation 9-36 times
 * (taps x output-channel tiles); the producers below split each fp32 value once into its three bf16 pieces
 * ([3][rows][C] unsigned short, plane_stride elements between planes) so the convolution's operand path is a plain copy.
 *   mft_bn_apply_planes               = mft_bn_apply with an additional planes output (y may be NULL)
 *   mft_bn_relu_maxpool_gather_planes = mft_bn_relu_maxpool_gather writing y AND planes
 *   mft_conv2d_nhwc_x3p_bnstats       = mft_conv2d_nhwc_x3_bnstats reading in_planes ([3][n_img*H*W][ldi]) instead of fp32 */
int mft_bn_apply_planes(const float* x, int ldx, float* y, int ldy, unsigned short* planes, long long plane_stride, int C,
                        int rows_per_group, int n_groups, const float* mean, const float* rstd, const float* gamma,
                        const float* beta, long long gb_group_stride, const float* res, int ldr, const float* res_mean,
                        const float* res_rstd, const float* res_gamma, const float* res_beta, int act, float slope,
                        void* stream);
int mft_bn_relu_maxpool_gather_planes(const float* x, const int* src_idx, float* y, unsigned short* planes,
                                      long long plane_stride, int n_img, int H, int W, int C, int imgs_per_group,
                                      const float* mean, const float* rstd, const float* gamma, const float* beta, void* stream);
int mft_conv2d_nhwc_x3p_bnstats(const unsigned short* in_planes, long long in_plane_elems, int ldi, const unsigned short* w3,
                                long long plane_elems, float* out, int ldo, int n_img, int H, int W, int Cin, int Cout, int KH,
                                int KW, int stride, int pad, int imgs_per_group, float eps, float* stats_ws, float* mean,
                                float* rstd, void* stream);

/* Episode ingest, finetune.py:208-233 (x_a_i = cat(view 0, view 0, view 1, ...) of the support images; x of view 0 for the
 * final pass): views[v] = device pointer of augmentation view v, an fp32 [n_way, per_class, C, H, W] tensor (host array of
 * n_views <= 32 device pointers).  support_store receives [(n_views + double_first) * n_way * n_support, H, W, C] (NHWC,
 * view-major, class-major inside a view), all_store (optional) [n_way * per_class, H, W, C] from view 0.  One launch.       */
int mft_ingest_episode_views(const float* const* views, int n_views, int double_first, int n_way, int per_class, int n_support,
                             int C, int H, int W, float* support_store, float* all_store, void* stream);
/* nn.Conv2d weight OIHW -> packed [Cout][KH][KW][Cin] padded to k_pad floats per row (zeros) */
int mft_pack_oihw(const float* w_oihw, float* w_pk, int Cout, int Cin, int KH, int KW, int k_pad, void* stream);
/* every OIHW -> packed repack of a model in one launch: jobs = n_jobs device records {const float* src; float* dst; int64 Cout,
 * Cin, KH*KW, k_pad, first_element} ordered by first_element over the concatenated packed outputs (total_elements floats).
 * The meta-training step calls it once after optimizer.step() (train.py:28, meta_template.py:87) instead of one launch per tensor. */
int mft_pack_oihw_multi(const void* jobs, int n_jobs, long long total_elements, void* stream);
int mft_unpack_oihw(const float* w_pk, float* w_oihw, int Cout, int Cin, int KH, int KW, int k_pad, void* stream);
/* packed forward weights -> packed dgrad weights: wt[ci][KH-1-kh][KW-1-kw][co] = w[co][kh][kw][ci];
 * `groups` independent weight sets, strides in floats */
int mft_pack_dgrad(const float* w_pk, float* wt_pk, int Cout, int Cin, int KH, int KW,
                   int groups, long long w_stride, long long wt_stride, void* stream);

/* convolution / GEMM on fp32 MFMA ----------------------------------------------------- */
/* nn.Conv2d(bias=False).forward (backbone.py:221,226,239,408) and, with KH=KW=1, nn.Linear /
 * 1x1 nn.Conv2d (gnnnet.py:30; gnn.py:64-76,38).  out[m][co] = sum_k A[m][k] * w[g][co][k] (+ bias[co]),
 * A = implicit im2col of the NHWC input, k = (kh*KW+kw)*Cin + ci.  Cin % 32 == 0 unless the
 * stem path (Cin == 3) is taken; w rows are k_pad = roundup(KH*KW*Cin, 32) floats.
 * imgs_per_group > 0 with w_group_stride != 0 selects per-group weights (M tiles never
 * straddle a group); imgs_per_group == 0 means one group of n_img images.                     */
int mft_conv2d_nhwc(const float* in, int ldi, const float* w, const float* bias, float* out, int ldo,
                    int n_img, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad,
                    int imgs_per_group, long long w_group_stride, void* stream);

/* The same convolution for FROZEN shared weights on the bf16 matrix cores with fp32 accuracy ("bf16x3", csrc/conv_x3.hip):
 * every fp32 operand is split exactly into three bf16 pieces and the six leading piece products are accumulated in fp32
 * (dropped terms <= 2^-24 relative: error at the level of fp32 rounding, 2.67x the fp32-MFMA rate).  mft_split_bf16x3 turns
 * a packed fp32 weight matrix [n] into three bf16 planes [3][n] once; mft_conv2d_nhwc_x3 consumes them (no bias, no groups;
 * Cin % 32 == 0, Cout % 64 == 0).  Replaces the same nn.Conv2d.forward call sites as mft_conv2d_nhwc for trunk.4-6.      */
int mft_split_bf16x3(const float* w, unsigned short* planes, long long n, void* stream);
/* The planes of SEVERAL weight matrices in one launch -- the per-step refresh of the meta-training path, where optimizer.step()
 * (train.py:28, meta_template.py:87) changes every weight: jobs = n_jobs records of 8 x int64 in device memory {packed fp32 source
 * [Cout][taps][Cin], planes [3][n], n = Cout*taps*Cin, Cout, Cin, taps, transposed, first element}, ordered by first element
 * (the running sum of n).  transposed = 1 writes the planes of the stride-1 DATA-GRADIENT operand wt[ci][taps-1-tap][co] =
 * w[co][tap][ci] (autograd of nn.Conv2d, loss.backward() at meta_template.py:86: dx = mft_conv2d_nhwc_x3(dy, wt), same padding). */
int mft_split_bf16x3_multi(const void* jobs, int n_jobs, long long total_elements, void* stream);
int mft_conv2d_nhwc_x3(const float* in, int ldi, const unsigned short* w3, long long plane_elems, float* out, int ldo,
                       int n_img, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad, void* stream);
/* mft_conv2d_nhwc_x3 with the statistics of the train-mode BatchNorm that follows it (backbone.py:224-227) produced in the
 * epilogue: per-tile (sum, sum of squares) of the <= 2 groups a 128-row tile touches go to stats_ws
 * (mft_conv2d_x3_stats_ws_floats floats), then mean/rstd [n_groups, Cout] are merged per group with Chan's formula in tile
 * order.  Groups = imgs_per_group consecutive images; needs imgs_per_group*OH*OW >= 128.  Replaces conv + mft_bn_stats.      */
long long mft_conv2d_x3_stats_ws_floats(int n_img, int H, int W, int Cout, int KH, int KW, int stride, int pad);
int mft_conv2d_nhwc_x3_bnstats(const float* in, int ldi, const unsigned short* w3, long long plane_elems, float* out, int ldo,
                               int n_img, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad,
                               int imgs_per_group, float eps, float* stats_ws, float* mean, float* rstd, void* stream);
/* Frozen-trunk SimpleBlock with the BatchNorms folded into their neighbours (backbone.py:251-261; one launch per convolution,
 * one for the block exit, none for statistics).  The statistics stay per-tile partials: pass mean = rstd = NULL to
 * mft_conv2d_nhwc_x3_bnstats and every CONSUMER merges them itself, in the order the finalize launch used (csrc/bn_fold.h).
 *   mft_conv2d_nhwc_x3_bnin_bnstats: 3x3 / stride 1 / pad 1 convolution of relu(BN(in)), ``in`` = raw output of the previous
 *     convolution, in_ws = its partials, in_gamma / in_beta [Cin] (C1 -> BN1 -> ReLU -> C2 in one launch).  MFT_EINVAL when the
 *     (scale, shift) table does not fit beside the tile at three workgroups per CU (Cin = 256 on 6x6 maps): run
 *     mft_bn_apply_x3ws + mft_conv2d_nhwc_x3_bnstats there.
 *   mft_bn_apply_x3ws: y = act(BN(x) [+ BN_res(res) | + res]) with both BatchNorms' statistics merged from partials
 *     (shared gamma / beta [C]); optional mean / rstd outputs [n_groups, C].
 *   mft_bn_relu_pooled_gather_moments: mft_bn_relu_pooled_gather with the batch statistics combined from the cached
 *     per-image moments in the same launch (replaces mft_bn_combine_moments + gather).
 * All BatchNorm-apply kernels of the trunk compute y = fma(x, rstd*gamma, beta - mean*rstd*gamma) (aten's CPU form).           */
int mft_conv2d_nhwc_x3_bnin_bnstats(const float* in, int ldi, const float* in_ws, const float* in_gamma, const float* in_beta,
                                    const unsigned short* w3, long long plane_elems, float* out, int ldo, int n_img, int H, int W,
                                    int Cin, int Cout, int imgs_per_group, float eps, float* stats_ws, float* mean, float* rstd,
                                    void* stream);
/* "f16x2" forms of the three frozen-trunk convolutions above (csrc/conv_x3.hip, NP = 2): every fp32 operand is split into TWO fp16
 * pieces, x = hi + 2^-11 lo with lo = fp16((x - hi) * 2^11) (|x - hi - 2^-11 lo| <= 2^-22 |x|), and the three leading piece
 * products run on the fp16 matrix cores into two fp32 accumulators (leading / cross terms) combined in the epilogue: half the
 * matrix work and two thirds of the LDS traffic of the bf16x3 forms at an error that stays below an fp32-accumulating GEMM's.
 * Operands must lie inside fp16's range (|x| < 65504): BatchNorm outputs do; the Python side checks the weights and the
 * BatchNorm affine parameters at load time and keeps the bf16x3 forms otherwise.  mft_split_f16x2: fp32 [n] -> planes [2][n].
 * Same arguments, workspaces and partial-statistics layout as the _x3 functions (their consumers do not care which form ran). */
int mft_split_f16x2(const float* w, unsigned short* planes, long long n, void* stream);
int mft_conv2d_nhwc_h2(const float* in, int ldi, const unsigned short* w2, long long plane_elems, float* out, int ldo,
                       int n_img, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad, void* stream);
int mft_conv2d_nhwc_h2_bnstats(const float* in, int ldi, const unsigned short* w2, long long plane_elems, float* out, int ldo,
                               int n_img, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad,
                               int imgs_per_group, float eps, float* stats_ws, float* mean, float* rstd, void* stream);
int mft_conv2d_nhwc_h2_bnin_bnstats(const float* in, int ldi, const float* in_ws, const float* in_gamma, const float* in_beta,
                                    const unsigned short* w2, long long plane_elems, float* out, int ldo, int n_img, int H, int W,
                                    int Cin, int Cout, int imgs_per_group, float eps, float* stats_ws, float* mean, float* rstd,
                                    void* stream);
int mft_bn_apply_x3ws(const float* x, int ldx, float* y, int ldy, int C, int rows_per_group, int n_groups, const float* ws,
                      const float* gamma, const float* beta, const float* res, int ldr, const float* res_ws,
                      const float* res_gamma, const float* res_beta, int act, float slope, float eps, float* mean, float* rstd,
                      float* res_mean, float* res_rstd, void* stream);
/* BatchNorm running_mean / running_var after n_steps train-mode forwards (torch: running = (1 - momentum) * running + momentum *
 * batch value, unbiased variance) whose batch statistics (mean, rstd [groups, C]) came out of grouped launches: step t used group
 * (order[t] & 0xffffff) of set a (order[t] >> 24 == 0) or b.  rows_a / rows_b = rows per group (for the unbiased factor). */
int mft_bn_running_ema(const float* mean_a, const float* rstd_a, int rows_a, const float* mean_b, const float* rstd_b, int rows_b,
                       const int* order, int n_steps, int C, float eps, float momentum, float* running_mean, float* running_var,
                       void* stream);
int mft_bn_apply_x3ws_fits(int C, int rows_per_group, int with_res_bn);  /* 1 when mft_bn_apply_x3ws can stage a group's partials in LDS */
int mft_bn_relu_pooled_gather_moments(const float* pmax, const float* pmin, const int* src_idx, float* y, int n_img, int OH, int OW,
                                      int C, int imgs_per_group, const float* mean_img, const float* m2_img, int rows_per_img,
                                      float eps, const float* gamma, const float* beta, float* mean, float* rstd, void* stream);

/* 1 when the library was built with -DMFT_EXPERIMENTS (MFT_EXPERIMENTS=1 python -m meta_fine_tuning_amd.build): the kernel
 * variants DESIGN.md records as measured slower and the timing / power ablation aids exist only then (tools/).
 * (The form-selection hooks the tests and A/B tools use are NOT part of this header: include/mft_hip_testing.h.)            */
int mft_has_experiments(void);

/* conv data gradient (autograd of nn.Conv2d / nn.Linear inputs in loss.backward(), finetune.py:293; meta_template.py:86):
 * dx[h][w][ci] = sum_{kh,kw,co} dy[(h+pad-kh)/s][(w+pad-kw)/s][co] * w[g][co][kh][kw][ci] (divisible offsets only), reading
 * the FORWARD weight pack directly (the transposition is resolved in the B-tile loader: no per-step weight transpose).
 * H, W: spatial size of dx (the forward input).  Cin/Cout/stride/pad are the forward convolution's; Cout % 32 == 0.   */
int mft_conv2d_dgrad_nhwc(const float* dy, int ldy, const float* w, float* dx, int ldx,
                          int n_img, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad,
                          int imgs_per_group, long long w_group_stride, void* stream);

/* conv weight gradient (autograd of nn.Conv2d in loss.backward(): finetune.py:293, gnnnet.py:174,
 * meta_template.py:86): dw[g][co][kh][kw][ci] = sum_{m in group g} dy[m][co] * im2col(in)[m][(kh,kw,ci)].
 * One weight gradient per group of imgs_per_group images (0: a single group).  Groups with more than 2048 rows are
 * reduced in 1024-row chunks (one workgroup column per chunk) through `ws` (mft_conv2d_wgrad_ws_floats floats; may be
 * NULL = no split) followed by a fixed-order sum.  Cin == 3 selects the 7x7 stem form.  Also nn.Linear / 1x1-conv
 * weight gradients (H=W=1, rows = n_img).                                                                          */
/* K-sliced forms of mft_conv2d_nhwc / mft_conv2d_dgrad_nhwc for single-weight-set launches with few output tiles and a long
 * reduction (the deep layers of ONE 105-image meta-training episode, train.py / meta_template.py:76-92): grid.z slices of the K
 * walk write partial tiles to ws, summed in slice order (deterministic).  mft_conv_ksplit_ws_floats(rows, cols, K) = floats of
 * workspace for rows x cols outputs over a reduction of K (0: the launch is not sliced; the *_ksplit entry points then run the
 * plain launch).  Forward: rows = n_img*OH*OW, cols = Cout, K = roundup(KH*KW*Cin, 32); data gradient: rows = n_img*H*W,
 * cols = Cin, K = KH*KW*Cout. */
long long mft_conv_ksplit_ws_floats(long long rows, int cols, int K);
/* ... and for a few weight sets in lockstep (2-8 episodes; per-episode weights w[g], imgs_per_group images each): grid.y = episode */
long long mft_conv_ksplit_grouped_ws_floats(long long rows_per_group, int cols, int K, int groups);
int mft_conv2d_nhwc_ksplit_grouped(const float* in, int ldi, const float* w, float* out, int ldo, int n_img, int H, int W, int Cin,
                                   int Cout, int KH, int KW, int stride, int pad, int imgs_per_group, long long w_group_stride,
                                   float* ws, void* stream);
int mft_conv2d_dgrad_nhwc_ksplit_grouped(const float* dy, int ldy, const float* w, float* dx, int ldx, int n_img, int H, int W, int Cin,
                                         int Cout, int KH, int KW, int stride, int pad, int imgs_per_group, long long w_group_stride,
                                         float* ws, void* stream);
int mft_conv2d_nhwc_ksplit(const float* in, int ldi, const float* w, const float* bias, float* out, int ldo, int n_img, int H, int W,
                           int Cin, int Cout, int KH, int KW, int stride, int pad, float* ws, void* stream);
int mft_conv2d_dgrad_nhwc_ksplit(const float* dy, int ldy, const float* w, float* dx, int ldx, int n_img, int H, int W, int Cin,
                                 int Cout, int KH, int KW, int stride, int pad, float* ws, void* stream);
long long mft_conv2d_wgrad_ws_floats(int n_img, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad,
                                     int imgs_per_group);
int mft_conv2d_wgrad_nhwc(const float* in, int ldi, const float* dy, int ldy, float* dw,
                          int n_img, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad,
                          int imgs_per_group, long long dw_group_stride, float* ws, void* stream);

/* mft_conv2d_wgrad_nhwc for one weight set, gradient written as torch's [Cout][Cin][KH][KW] (Conv2d.weight.grad): the sum over
 * the split-M partials does the permutation (replaces wgrad + mft_unpack_oihw in the meta-training backward).
 * cin_valid / cout_valid (0 = all): operands whose channel counts are zero-padded to the kernels' multiples (the GNN's 133 / 181 /
 * 229 / 266 / 362 / 458 input features, its 48-, 5- and 1-row layers) hand back the [cout_valid][cin_valid][KH][KW] corner
 * contiguously -- nn.Linear / 1x1 Conv2d .grad exactly, no slice + copy afterwards (gnn.py:38,64-76). */
long long mft_conv2d_wgrad_oihw_ws_floats(int n_img, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad);
int mft_conv2d_wgrad_oihw(const float* in, int ldi, const float* dy, int ldy, float* dw_oihw, int n_img, int H, int W, int Cin,
                          int Cout, int KH, int KW, int stride, int pad, int cin_valid, int cout_valid, float* ws, void* stream);

/* Several independent mft_conv2d_wgrad_oihw problems in ONE pair of launches (up to 16 per pair; a Cin == 3 stem job keeps its own):
 * the meta-training backward (meta_template.py:76-92: loss.backward() over 31 convolution / linear layers) defers every weight
 * gradient to the end of its pass -- nothing downstream reads them -- and runs them together.  jobs: HOST array; each job means
 * what the same-named arguments of mft_conv2d_wgrad_oihw mean (ws: mft_conv2d_wgrad_oihw_ws_floats floats per job); every job's
 * result is bit-identical to its own mft_conv2d_wgrad_oihw launch (same tiles, chunking and summation order). */
typedef struct MftWgradJob {
    const float* in; const float* dy; float* dw_oihw; float* ws;
    int ldi, ldy, n_img, H, W, Cin, Cout, KH, KW, stride, pad, cin_valid, cout_valid, reserved;
} MftWgradJob;
int mft_conv2d_wgrad_oihw_multi(const MftWgradJob* jobs, int n_jobs, void* stream);

/* conv weight gradient with torch.optim.Adam.step fused into the epilogue (finetune.py:293-299): the gradient tile stays
 * in MFMA accumulators; w, m, v (same packed layout / group stride as dw) are updated in place.  dw_or_null, when given,
 * also receives the gradient.  Same Adam form as mft_adam_step (no weight decay).                                    */
int mft_conv2d_wgrad_adam_nhwc(const float* in, int ldi, const float* dy, int ldy, float* w, float* m, float* v,
                               float* dw_or_null, int n_img, int H, int W, int Cin, int Cout, int KH, int KW,
                               int stride, int pad, int imgs_per_group, long long group_stride,
                               int step, float lr, float beta1, float beta2, float eps, void* stream);

/* loss.backward() + delta_opt.step() of inner step t AND the convolution of inner step t+1 in one pass over w, m, v
 * (finetune.py:286-299: the first reader of what Adam wrote is `pretrained_model(z_batch)` of the NEXT iteration;
 * backbone.py:251-261 for the block structure).  Per-episode ("group") weights, <= 48 output pixels per episode.
 * Gradient + Adam exactly as mft_conv2d_wgrad_adam_nhwc (bit-identical w, m, v; dw_or_null also receives the gradient;
 * `hyper` != NULL: step size / bias correction from device memory as in the *_dev launchers, `step`/`lr` ignored).
 * x_next: the NEXT step's input activation (same geometry as x; NULL = no forward, the last inner step).  Every workgroup
 * multiplies each weight tile it has just updated with x_next's im2col rows while the tile is still in LDS, so the updated
 * weights are not read back from HBM by a forward launch.  `mode` selects the epilogue over the episode's pixels:
 *   0  raw [n_img*OH*OW, Cout] only                                   (trunk.7.shortcut; its BatchNorm is applied by mode 2)
 *   1  raw + mean / rstd [groups, Cout] + act = ReLU(BN(raw))          (trunk.7.C1 + BN1 + ReLU, backbone.py:252-254)
 *   2  raw + both BatchNorms' statistics + act = ReLU(BN(raw) + BN_s(sc_raw)) + pooled [n_img, Cout] = global average pool of act
 *                                                                      (trunk.7.C2 .. AvgPool2d, backbone.py:255-261,438)
 * gamma / beta (and gamma_s / beta_s) are per-group with stride gb_group_stride.  MFT_EINVAL outside the domain
 * (Cin % 128, Cout % 32, imgs_per_group * OH * OW <= 48, imgs_per_group <= 8).                                             */
int mft_wgrad_adam_next_forward(const float* x, int ldx, const float* dy, int ldy, float* w, float* m, float* v,
                                float* dw_or_null, int n_img, int H, int W, int Cin, int Cout, int KH, int KW, int stride,
                                int pad, int imgs_per_group, long long group_stride, int step, const float* hyper, float lr,
                                float beta1, float beta2, float eps, const float* x_next, int mode, float* raw, float* act,
                                const float* gamma, const float* beta, long long gb_group_stride, float* mean, float* rstd,
                                const float* sc_raw, const float* gamma_s, const float* beta_s, float* mean_s, float* rstd_s,
                                float* pooled, float bn_eps, void* stream);
/* (its Adam epilogue: v_rcp_f32 / v_sqrt_f32 as mft_conv2d_wgrad_adam_nhwc's default -- DESIGN.md section 6, deviations; an episode's
 * Cout / 32 workgroups are placed on ONE XCD; the correctly rounded form and the natural order are test hooks, mft_hip_testing.h) */


/* BatchNorm (train mode, batch statistics) --------------------------------------------- */
/* F.batch_norm(training=True) statistics (backbone.py:224,227,240,409; gnn.py:65-74; gnnnet.py:30):
 * per (group, channel) mean and 1/sqrt(biased var + eps) over rows_per_group rows.
 * ws: >= mft_bn_stats_ws_floats(...) floats.  running_mean/var (nullable, [C]) get the momentum update
 * with the unbiased variance of GROUP 0 (one group in the reference's loops; with k episodes in lockstep
 * group 0 plays rank 0 of an episode-parallel run); num_batches_tracked (nullable, int64
 * scalar, only with running_mean) is incremented by the same launch (nn.BatchNorm2d's counter).     */
long long mft_bn_stats_ws_floats(int C, int rows_per_group, int n_groups);
int mft_bn_stats(const float* x, int ldx, int C, int rows_per_group, int n_groups, float eps,
                 float* mean, float* rstd, float* ws,
                 float* running_mean, float* running_var, float momentum, long long* num_batches_tracked, void* stream);
/* Several independent mft_bn_stats problems (up to 8) in ONE launch pair -- SimpleBlock's BN2 and BNshortcut normalise two tensors
 * that exist together (backbone.py:256-259).  jobs: HOST array; a job = mft_bn_stats's arguments of the same names; results are
 * bit-identical to its own launches. */
typedef struct MftBnStatsJob {
    const float* x; float* mean; float* rstd; float* ws; float* running_mean; float* running_var; long long* num_batches_tracked;
    int ldx, C, rows_per_group, n_groups; float eps, momentum;
} MftBnStatsJob;
int mft_bn_stats_multi(const MftBnStatsJob* jobs, int n_jobs, void* stream);
/* Train-mode BatchNorm of a SMALL tensor in ONE launch: batch statistics (+ running-statistics update from group 0 + counter, as
 * mft_bn_stats), then y = act( bn(x) [+ res | + bn_r(res)] ) as mft_bn_apply -- for the head's BatchNorm1d layers of a meta-training step
 * (gnnnet.py:44, gnn.py:134-166: 105 / 480 rows per episode), where statistics -> finalize -> apply are three dependent launches of a
 * few microseconds of work.  One workgroup per (group, 4 channels) walks all rows twice; the caller chooses it up to
 * mft_bn_forward_small_max_rows() rows per group (512: at the trunk's 945 / 3,780-row layers the three launches are faster).  res_gamma != NULL: the residual
 * goes through its own BatchNorm (statistics taken here, saved to res_mean / res_rstd, res_running_* updated).  Same formulas as
 * mft_bn_stats + mft_bn_apply; the sums are taken in another fixed order (equal to rounding).  y may be a column window of a wider
 * matrix (ldy = its row length; any 4-byte alignment): Gconv's output lands in the node-feature matrix directly (gnn.py:160-165). */
typedef struct MftBnFwdJob {
    const float* x; float* y; const float* gamma; const float* beta; float* mean; float* rstd;
    float* running_mean; float* running_var; long long* num_batches_tracked;
    const float* res; const float* res_gamma; const float* res_beta; float* res_mean; float* res_rstd;
    float* res_running_mean; float* res_running_var; long long* res_num_batches_tracked;
    int ldx, ldy, ldr, C, rows_per_group, n_groups, act; float eps, momentum, slope;
} MftBnFwdJob;
int mft_bn_forward_small(const MftBnFwdJob* job, void* stream);
int mft_bn_forward_small_max_rows(void);
/* Several independent BatchNorm-apply problems (no residual) in ONE launch: the meta-training backward re-creates the twelve pair-MLP
 * activations h_l = leaky_relu(BatchNorm(z_l)) (gnn.py:87-102; the forward keeps only the raw z_l) in front of its weight-gradient
 * launches -- all twelve depend on the forward's tape only.  jobs: HOST array, up to 16 per launch; each job = mft_bn_apply's
 * arguments of the same names (gamma / beta shared by the groups), bit-identical to its own launch. */
typedef struct MftBnApplyJob {
    const float* x; float* y; const float* mean; const float* rstd; const float* gamma; const float* beta;
    int ldx, ldy, C, rows_per_group, n_groups, act; float slope; int reserved;
} MftBnApplyJob;
int mft_bn_apply_multi(const MftBnApplyJob* jobs, int n_jobs, void* stream);
/* y = act( bn(x) [+ residual | + bn_r(residual)] ); gamma/beta [C] shared (gb_group_stride 0) or per group.
 * SimpleBlock.forward tail (backbone.py:251-261); F.leaky_relu(bn(.)) in Wcompute (gnn.py:84-102). */
int mft_bn_apply(const float* x, int ldx, float* y, int ldy, int C, int rows_per_group, int n_groups,
                 const float* mean, const float* rstd, const float* gamma, const float* beta,
                 long long gb_group_stride,
                 const float* res, int ldr, const float* res_mean, const float* res_rstd,
                 const float* res_gamma, const float* res_beta,
                 int act, float slope, void* stream);
/* eval-mode BatchNorm (pretrained_model.eval() under --freeze_backbone, finetune.py:265-266): rstd = 1/sqrt(running_var+eps) */
int mft_var_to_rstd(const float* var, float* rstd, int n, float eps, void* stream);
/* trunk[1..3]: BatchNorm2d -> ReLU -> MaxPool2d(3,2,1) fused (backbone.py:409-411) */
int mft_bn_relu_maxpool(const float* x, float* y, int n_img, int H, int W, int C, int imgs_per_group,
                        const float* mean, const float* rstd, const float* gamma, const float* beta,
                        void* stream);

/* Stem cache (test-time fine-tune, finetune.py:263-291).  The reference re-runs trunk.0 (conv 7x7/2) on an image every time
 * it is drawn into a mini-batch (once per epoch); the convolution output does not depend on the mini-batch, only the
 * BatchNorm statistics of trunk.1 do.  mft_bn_image_moments reduces a cached conv output [n_img, rows_per_img, C] to per-image
 * (mean, M2 = sum (x-mean)^2); mft_bn_combine_moments turns the moments of the images idx[g*imgs_per_group ..] into the
 * mini-batch statistics of group g (Chan's combination, fixed order); mft_bn_relu_maxpool_gather is mft_bn_relu_maxpool
 * reading image n from slot src_idx[n] of the cache (src_idx NULL: identity).                                              */
int mft_bn_image_moments(const float* x, int ldx, int C, int rows_per_img, long long n_img, float* mean_img, float* m2_img,
                         void* stream);
int mft_bn_combine_moments(const float* mean_img, const float* m2_img, const int* idx, int C, int rows_per_img,
                           int imgs_per_group, int n_groups, float eps, float* mean, float* rstd, void* stream);
int mft_bn_relu_maxpool_gather(const float* x, const int* src_idx, float* y, int n_img, int H, int W, int C,
                               int imgs_per_group, const float* mean, const float* rstd, const float* gamma,
                               const float* beta, void* stream);
/* trunk[8..9]: global average pool + flatten (backbone.py:427-430) */
int mft_global_avgpool(const float* x, float* y, int n_img, int HW, int C, void* stream);

/* BatchNorm backward (train mode).  dy_eff = dy * (y > 0) when relu_out != NULL (ReLU backward fused).
 * dx = gamma*rstd*(dy_eff - mean_g(dy_eff) - xhat*mean_g(dy_eff*xhat)); dgamma = sum dy_eff*xhat; dbeta = sum dy_eff.
  * One workgroup per (group, 64-channel tile): fixed-order reduction, then dx (rows re-read from L2). */
int mft_bn_backward(const float* x, int ldx, const float* dy, int lddy, const float* relu_out, int ldro,
                    float* dx, int lddx, int C, int rows_per_group, int n_groups,
                    const float* mean, const float* rstd, const float* gamma, long long gb_group_stride,
                    float* dgamma, float* dbeta, void* stream);
/* Fused forms for groups of <= 64 rows (the adapted last block: 5 images x 3x3 pixels per episode), one launch instead of
 * 3-6 (same reference call sites as mft_bn_stats/apply/global_avgpool: backbone.py:224-261,427):
 * mft_bn_small_forward: y = act(bn(x1) [+ bn(x2) | + res]) with the statistics of x1 (and x2) computed in the same pass
 *   (returned in mean1, rstd1, mean2, rstd2 for the backward), optional pooled[img][c] = mean over each image's hw rows of y.
 * mft_bn_backward2: BatchNorm backward of two branches (xa, xb) that share the incoming gradient dy.
 * mft_ce_pool_backward: per-group cross entropy on feat (finetune.py:286-293) and its gradient pushed through AvgPool and
 *   the block's final ReLU: d_out = (out > 0) * (softmax(feat) - onehot) / (rows_per_group * hw); loss[g] nullable.      */
int mft_bn_small_forward(const float* x1, int ld1, const float* x2, int ld2, const float* res, int ldr, float* y, int ldy, int C,
                         int rows_per_group, int n_groups, const float* gamma1, const float* beta1, const float* gamma2,
                         const float* beta2, long long gb_group_stride, float* mean1, float* rstd1, float* mean2, float* rstd2,
                         int act, float slope, float eps, float* pooled, int hw, void* stream);
int mft_bn_backward2(const float* xa, const float* xb, int ldx, const float* dy, int lddy, float* dxa, float* dxb, int lddx, int C,
                     int rows_per_group, int n_groups, const float* mean_a, const float* rstd_a, const float* gamma_a,
                     const float* mean_b, const float* rstd_b, const float* gamma_b, long long gb_group_stride, float* dgamma_a,
                     float* dbeta_a, float* dgamma_b, float* dbeta_b, void* stream);
int mft_ce_pool_backward(const float* feat, const int* labels, int rows_per_group, int n_groups, int C, int hw, const float* out,
                         float* d_out, float* loss, void* stream);

/* d(out)[n,hw,c] = (out>0) * dfeat[n,c] / HW : AvgPool + relu2 backward (backbone.py:260,427) */
int mft_avgpool_relu_backward(const float* dfeat, const float* out, float* dout, int n_img, int HW, int C, void* stream);

/* loss --------------------------------------------------------------------------------- */
/* nn.CrossEntropyLoss forward+backward (finetune.py:291-293; gnnnet.py:170-174,221-224): mean over the
 * rows of each group; loss[n_groups]; dlogits = (softmax - onehot)/rows_per_group (nullable).     */
int mft_cross_entropy(const float* logits, int ld, const int* labels, int C, int rows_per_group, int n_groups,
                      float* loss, float* dlogits, void* stream);
/* nn.CrossEntropyLoss(reduction='mean') on [rows, C] logits as the episode / pre-training loss modules call it
 * (gnnnet.py:43,219-231: scores [80,5] vs repeat(range(5),16); baselinetrain.py:20,38-45): ONE launch each way.  labels:
 * int64 (labels_i64 = 1, what torch passes) or int32; loss[0] = mean_r(logsumexp(x_r) - x_r[y_r]) summed in a fixed order;
 * loss_sum (nullable, float64 device scalar) += loss: the running loss the episode loop prints (meta_template.py:91) without a
 * host read-back per step;  backward: dlogits = (softmax - onehot) * grad_loss[0] / rows, grad_loss a DEVICE scalar (NULL = 1). */
int mft_cross_entropy_mean(const float* logits, int ld, const void* labels, int labels_i64, int C, int rows, float* loss,
                           double* loss_sum, void* stream);
int mft_cross_entropy_mean_backward(const float* logits, int ld, const void* labels, int labels_i64, int C, int rows,
                                    const float* grad_loss, float* dlogits, int ldd, void* stream);
int mft_softmax_rows(const float* x, int ldx, float* y, int ldy, int C, int rows, void* stream);

/* optimisers --------------------------------------------------------------------------- */
/* torch.optim.Adam.step (finetune.py:255,299; gnnnet.py:128,177; train.py:28), one flat slab of n floats:
 * g += wd*p; m = b1*m+(1-b1)*g; v = b2*v+(1-b2)*g*g; p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t)+eps) */
int mft_adam_step(float* p, const float* g, float* m, float* v, long long n, int step,
                  float lr, float beta1, float beta2, float eps, float weight_decay, void* stream);
/* torch.optim.SGD(momentum, dampening, weight_decay).step (meta_template.py:166; baselinefinetune.py:35) */
int mft_sgd_step(float* p, const float* g, float* buf, long long n, int first_step,
                 float lr, float momentum, float dampening, float weight_decay, void* stream);
/* GnnNet.MAML_update (gnnnet.py:90-103): p -= (p3 - p2) */
/* Entry half of a down-sampling SimpleBlock for per-episode weights in one launch (backbone.py:251-256 + the shortcut of
 * :258): c1 = C1(x) (3x3, stride, pad 1), r1 = ReLU(BatchNorm_train(c1)) with the statistics of each episode's own
 * imgs_per_group images (mean1/rstd1 [groups, Cout] returned for the backward), sc = shortcut(x) (1x1, stride, pad 0).  Domain:
 * <= 48 output pixels per episode, Cin <= 256 (multiple of 128), Cout multiple of 256; MFT_EINVAL outside it (callers then use
 * mft_conv2d_nhwc + mft_bn_small_forward).                                                                                   */
int mft_block_entry_small_forward(const float* x, int ldx, const float* w_c1, long long w1_group_stride, const float* w_sc,
                                  long long wsc_group_stride, float* c1, float* r1, float* sc, int n_img, int H, int W, int Cin,
                                  int Cout, int stride, int imgs_per_group, const float* gamma1, const float* beta1,
                                  long long gb_group_stride, float* mean1, float* rstd1, float eps, void* stream);

/* Exit half of the same block + ResNet's global average pool in one launch (backbone.py:256-261, :309): c2 = C2(r1) (3x3,
 * stride 1, pad 1, C -> C), out = ReLU(BatchNorm_train(c2) + BatchNorm_train(sc)) with per-episode statistics of both branches
 * (returned in mean2/rstd2/mean_sc/rstd_sc [groups, C]), pooled [n_img, C] = mean of out over each image's H*W pixels.
 * MFT_EINVAL outside the per-episode kernel's domain (callers then use mft_conv2d_nhwc + mft_bn_small_forward).                */
int mft_block_exit_small_forward(const float* r1, const float* w_c2, long long w_group_stride, const float* sc, float* c2,
                                 float* out, float* pooled, int n_img, int H, int W, int C, int imgs_per_group,
                                 const float* gamma2, const float* beta2, const float* gamma_sc, const float* beta_sc,
                                 long long gb_group_stride, float* mean2, float* rstd2, float* mean_sc, float* rstd_sc, float eps,
                                 void* stream);

/* loss.backward() through C2 and the BatchNorm + ReLU in front of it (finetune.py:293 over backbone.py:253-256) in one launch:
 * dx_conv = dgrad(dy, w) as mft_conv2d_dgrad_nhwc, then the BatchNorm/ReLU backward over each episode's rows:
 * dx = gamma*rstd*(g - mean(g) - xhat*mean(g*xhat)), g = dx_conv*(relu_out > 0), xhat = (x_raw - mean)*rstd; dgamma/dbeta
 * [groups, Cin].  Same domain as the per-episode data-gradient kernel (stride 1, "same" padding, <= 48 pixels per episode). */
int mft_conv2d_dgrad_bn_backward_small(const float* dy, int ldy, const float* w, float* dx, int ldx, int n_img, int H, int W,
                                       int Cin, int Cout, int KH, int KW, int pad, int imgs_per_group, long long w_group_stride,
                                       const float* x_raw, const float* relu_out, const float* mean, const float* rstd,
                                       const float* gamma, long long gb_group_stride, float* dgamma, float* dbeta, void* stream);

/* MetaTemplate.set_forward_adaptation / BaselineFinetune.set_forward (meta_template.py:153-186, baselinefinetune.py:17-58) as ONE
 * launch: per group (episode) a Linear(D, n_way) head (W [n_groups,n_way,D], b [n_groups,n_way], updated in place) is trained on
 * the frozen support features z_support [n_groups, n_support_rows, D] with torch.optim.SGD(lr, momentum, dampening, L2
 * weight_decay) semantics for n_steps mini-batches; idx_table [n_groups, n_steps, batch_size] holds the support row of every
 * mini-batch slot (-1 = empty slot of a ragged batch), y_support [n_groups, n_support_rows] the labels.                      */
int mft_linear_head_sgd_run(const float* z_support, const int* y_support, const int* idx_table, int n_groups, int n_support_rows,
                            int D, int n_way, int n_steps, int batch_size, float* W, float* b, float lr, float momentum,
                            float dampening, float weight_decay, void* stream);

/* Same single-launch run with torch.optim.Adam(lr, (beta1, beta2), eps, L2 weight_decay) on the head: finetune.finetune_linear
 * with freeze_backbone=True (finetune.py:45-174, frozen branch of :123-135, :144, :163) trains only Classifier(512, n_way) on
 * constant eval-mode features for 20 epochs of mini-batches of 5.                                                          */
int mft_linear_head_adam_run(const float* z_support, const int* y_support, const int* idx_table, int n_groups, int n_support_rows,
                             int D, int n_way, int n_steps, int batch_size, float* W, float* b, float lr, float beta1, float beta2,
                             float eps, float weight_decay, void* stream);

/* torch.optim.Adam.step over many tensors in one launch (train.py:28, meta_template.py:87): chunk_table is a DEVICE array of
 * n_chunks records {float* p; const float* g; float* m; float* v; long long n;} (40 bytes each, n <= 65536 elements; 16-byte aligned chunks take the float4 path).
 * grad_scale multiplies every gradient as it is read (1 = torch's step; 1 / world size when g is the all-reduced SUM of the ranks'
 * gradients: the division of the episode-parallel step without a launch of its own).        */
int mft_adam_multi(const void* chunk_table, int n_chunks, int step, float lr, float beta1, float beta2, float eps,
                   float weight_decay, float grad_scale, void* stream);

/* hipGraph support: kernel arguments are frozen when a graph is captured, so Adam's bias corrections must come from device
 * memory.  mft_adam_hyper_advance does t = ++(*step) and writes hyper = {lr/(1-beta1^t), 1/sqrt(1-beta2^t)} (double precision,
 * as the host path); the *_dev variants of the Adam launchers read them instead of taking `step`/`lr`.                       */
int mft_adam_hyper_advance(int* step, float* hyper, float lr, float beta1, float beta2, void* stream);
int mft_adam_step_dev(float* p, const float* g, float* m, float* v, long long n, const float* hyper, float beta1, float beta2,
                      float eps, float weight_decay, void* stream);
int mft_conv2d_wgrad_adam_nhwc_dev(const float* in, int ldi, const float* dy, int ldy, float* w, float* m, float* v,
                                   float* dw_or_null, int n_img, int H, int W, int Cin, int Cout, int KH, int KW, int stride,
                                   int pad, int imgs_per_group, long long group_stride, const float* hyper, float beta1,
                                   float beta2, float eps, void* stream);
int mft_maml_delta(float* p, const float* p2, const float* p3, long long n, void* stream);

/* finetune_linear's classifier head (finetune.py:33-42,65,103,147-158,171-174), one Linear(D, n_way) per episode (group).
 * mft_linear_head_step = one inner step for all groups in one launch: logits = feat @ W[g]^T + b[g]; mean cross entropy over
 * the group's rows (loss[g], nullable); dfeat = dlogits @ W[g] (pre-update weights; the gradient handed to the backbone);
 * Adam(lr, betas, eps, L2 weight_decay) on W [n_groups,n_way,D], b [n_groups,n_way] with moments mW,vW,mb,vb.
 * rows_per_group <= 16, n_way <= 16.  mft_linear_head_scores: out[row][:] = softmax(feat[row] @ W[g]^T + b[g]).        */
int mft_linear_head_step(const float* feat, int ldf, const int* labels, int rows_per_group, int n_groups, int n_way, int D,
                         float* W, float* b, float* mW, float* vW, float* mb, float* vb, float* dfeat, int lddf, float* loss,
                         int step, float lr, float beta1, float beta2, float eps, float weight_decay, void* stream);
int mft_linear_head_scores(const float* feat, int ldf, int rows_per_group, int n_groups, int n_way, int D, const float* W,
                           const float* b, float* out, void* stream);

/* GNN head ------------------------------------------------------------------------------ */
/* Wcompute: W_new = abs(x_i - x_j) (gnn.py:79-82): x [n_graphs*N, ldx] -> d [n_graphs*N*N, ldd]; columns
 * F..ldd-1 of d are zero-filled.                                                              */
int mft_pair_absdiff(const float* x, int ldx, float* d, int ldd, int n_graphs, int N, int F, void* stream);
/* Wcompute tail (gnn.py:103-115): s[b,i,j] (ld = lds_, column 0 of the conv2d_last GEMM) -> A[b,i,:] =
 * softmax_j(s - 1e8*[i==j])                                                                    */
int mft_masked_softmax(const float* s, int lds_, float* A, int n_graphs, int N, void* stream);
/* Fused per-pair MLP of Wcompute (gnn.py:78-115; csrc/pair_mlp.hip).  The pair tensor |x_i - x_j| is never materialised and
 * only the N(N+1)/2 pairs i <= j exist ("upper-triangle rows", row p(i,j) = i*N - i(i-1)/2 + (j-i); `ij[p]` = (i << 16) | j).
 * A "group" is one episode: `graphs_per_group` graphs whose pair positions share BatchNorm statistics.
 *   mft_pair_mlp_layer: one 1x1-conv layer over all groups.  mode 0: the A operand is |x_i - x_j| generated from the node
 *     features `in` [n_groups*graphs_per_group*N, ld_in] (columns >= K masked); mode 1: `in` is the previous layer's RAW
 *     output [rows, ld_in = K] and BatchNorm + leaky_relu(slope) is applied while loading, y = in*scale_in + shift_in per
 *     group.  Writes the raw output `out` [n_groups*graphs_per_group*P, Cout] (bias added) and, per (group, 128-row tile),
 *     the weighted per-channel (mean, M2) and weight sum (off-diagonal rows count twice) into ws_mean / ws_m2 [n_groups *
 *     tiles_m, Cout] and ws_n [n_groups * tiles_m]; tiles_m = mft_pair_mlp_tiles_m(graphs_per_group, N).
 *     f16x2 = 0: fp32 MFMA (exact products); 1: both operands split into two fp16 pieces by the loader, three products on the
 *     fp16 matrix cores, fp32 accumulation (fp32-accurate; operands must lie inside fp16's range, |x| < 65504, else the output
 *     is NaN: the caller checks weights / BatchNorm bounds where they are frozen, functional.pair_f16x2_safe).
 *   mft_pair_mlp_stats_finalize: merges a group's tiles (Chan, tile order) into the next layer's affine scale = gamma *
 *     rstd, shift = beta - mean * scale (biased variance over graphs_per_group*N*N positions; mean_out / rstd_out optional).
 *   mft_pair_mlp_score: conv2d_last (C -> 1) on BatchNorm + leaky_relu of the last raw layer -> compact symmetric scores
 *     s_ut [n_groups*graphs_per_group*P].
 *   mft_masked_softmax_ut: A[b,i,:] = softmax_j(s[b,i,j] - 1e8*[i==j]) (gnn.py:105-115) reading s through p(i,j).
 *   mft_pair_mlp_layer_rk / mft_pair_mlp_tiles_m_rk / mft_pair_mlp_stats_finalize_rk: the same layer for SMALL problems (the
 *     meta-training step's single episode, meta_template.py:76-92): 32-row tiles (tiles_m = mft_pair_mlp_tiles_m_rk), the whole
 *     K (Kpad <= 256) held in registers and split over a workgroup's four waves, no LDS staging -- same arguments and outputs;
 *     fp32 MFMA only; its finalize merges the (4x as many) tiles as a wave-wide tree, one wave per channel.  Sums in a different (fixed)
 *     order than mft_pair_mlp_layer / _stats_finalize: equal to rounding, not bit for bit.                                   */
/* Skinny GEMM of the head's linear layers in a meta-training step (gnnnet.py:44 fc; gnn.py:134-166 Gconv.fc; meta_template.py:76-92):
 * out[m][n] = sum_k a[m][k] * w[n][k] + bias[n] for m < M, n < N.  Register-K form: 16-row tiles, the whole K (<= 512, K % 16 == 0)
 * in registers over four waves, no LDS staging (csrc/gnn.hip).  w: packed [w_rows >= N][K]; lda % 4 == 0, lda >= K; columns
 * N..ldo-1 of out are left alone; bias nullable.  Same result as mft_conv2d_nhwc's 1x1 form to rounding (another fixed order over k). */
int mft_gemm_rk(const float* a, int lda, const float* w, int w_rows, int K, const float* bias, float* out, int ldo, int M, int N,
                void* stream);
int mft_pair_mlp_tiles_m(int graphs_per_group, int N);
int mft_pair_mlp_tiles_m_rk(int graphs_per_group, int N);
int mft_pair_mlp_layer_rk(const float* in, int ld_in, int mode, const int* ij, const float* scale_in, const float* shift_in,
                          const float* w, int K, int Kpad, const float* bias, float* out, int Cout, int n_groups,
                          int graphs_per_group, int N, float slope, float* ws_mean, float* ws_m2, float* ws_n, void* stream);
int mft_pair_mlp_stats_finalize_rk(const float* ws_mean, const float* ws_m2, const float* ws_n, int n_groups, int tiles_m, int C,
                                   const float* gamma, const float* beta, float eps, float* scale, float* shift,
                                   float* mean_out, float* rstd_out, void* stream);
int mft_pair_mlp_layer(const float* in, int ld_in, int mode, const int* ij, const float* scale_in, const float* shift_in,
                       const float* w, int K, int Kpad, const float* bias, float* out, int Cout, int n_groups,
                       int graphs_per_group, int N, float slope, float* ws_mean, float* ws_m2, float* ws_n, int f16x2, void* stream);
int mft_pair_mlp_stats_finalize(const float* ws_mean, const float* ws_m2, const float* ws_n, int n_groups, int tiles_m, int C,
                                const float* gamma, const float* beta, float eps, float* scale, float* shift,
                                float* mean_out, float* rstd_out, void* stream);
int mft_pair_mlp_score(const float* h, int C, const float* scale, const float* shift, const float* w5, const float* b5,
                       float slope, float* s_ut, int n_groups, int graphs_per_group, int N, void* stream);
int mft_masked_softmax_ut(const float* s_ut, float* A, int n_graphs, int N, void* stream);

/* loss.backward() through gnn.Wcompute (gnn.py:78-132 under autograd, meta_template.py:76-92) on the forward's upper-triangle pair
 * rows p(i, j), i <= j: a merged row carries the sum of the reference's (i, j) and (j, i) gradients (every operator is linear in
 * the gradient).  Nothing of shape [B*N*N, F] is formed.
 *  mft_pair_softmax_ut_backward: ds[r*ldds] = A_ij (dA_ij - <A_i, dA_i>) + A_ji (dA_ji - <A_j, dA_j>)  (rowdot_ws: n_graphs*N floats;
 *      dbias_zero, nullable: one float <- 0 = the gradient of conv2d_last's bias, identically zero under the row softmax)
 *  mft_pair_bn_act_backward:     BatchNorm2d(train) + leaky_relu backward of one layer over merged rows.  g = dL/d(activation)
 *      [rows, ldg], z = the layer's RAW output [rows, C] (C = 96 | 192), rows = n_groups * rows_per_group (a group = the graphs
 *      of one episode: its own BatchNorm statistics); scale / shift / mean / rstd [n_groups, C] as left by
 *      mft_pair_mlp_stats_finalize; sums[n_groups][2*C] <- per group (sum u, sum u*xhat); dparams[2*C] (nullable) <- their sum over
 *      the groups = (d beta | d gamma); dbias_zero[C] (nullable) <- 0, the gradient of the 1x1 convolution's bias in front of
 *      the BatchNorm (identically zero); dz [rows, C] <- gamma rstd (u - cnt sum_u / n_tot - cnt xhat sum_ux / n_tot) with cnt = 1 on
 *      diagonal rows, 2 elsewhere, n_tot = graphs_per_group*N*N.
 *      ws: n_groups * mft_pair_bwd_stats_ws_floats(rows_per_group, C) floats (fixed-order partial sums: deterministic).
 *  mft_pair_absdiff_ut:          d[r - row0][:] = |x_i - x_j| for rows row0 .. row0 + nrows (layer 1's input, a bounded chunk)
 *  mft_pair_dx_gather:           dX[b, i, :F] += sum_j sign(x_i - x_j) * dd[p(i,j) - row0][:F] over the pairs inside the chunk     */
int mft_pair_softmax_ut_backward(const float* A, const float* dA, const int* ij, float* rowdot_ws, float* ds, int ldds,
                                 int n_graphs, int N, float* dbias_zero, void* stream);
long long mft_pair_bwd_stats_ws_floats(long long rows_per_group, int C);
int mft_pair_bn_act_backward(const float* g, int ldg, const float* z, int C, const float* scale, const float* shift,
                             const float* mean, const float* rstd, const float* gamma, const int* ij, int N, long long rows_per_group,
                             int n_groups, long long n_tot, float slope, float* ws, float* sums, float* dparams, float* dbias_zero,
                             float* dz, void* stream);
int mft_pair_absdiff_ut(const float* x, int ldx, const int* ij, float* d, int Kp, int F, int N, long long row0, long long nrows,
                        void* stream);
int mft_pair_dx_gather(const float* x, int ldx, const float* dd, int lddd, float* dX, int lddx, int n_graphs, int N, int F,
                       long long row0, long long nrows, void* stream);
/* gmul (gnn.py:16-28) with J=2: y[b,i,:] = cat(x[b,i,:F], (A[b] @ x[b])[i,:F]) zero padded to ldy */
int mft_graph_aggregate(const float* A, const float* x, int ldx, float* y, int ldy,
                        int n_graphs, int N, int F, void* stream);
/* y[r, col_off + c] = act(x[r,c]) : strided copy used for GNN_nl's cat (gnn.py:160-161) */
int mft_copy_cols(const float* x, int ldx, float* y, int ldy, int col_off, int C, int rows,
                  int act, float slope, void* stream);

/* z_stack + cat(z, support_label) (gnnnet.py:34-38,82-83,212): z [n_episodes*n_way*(S+n_query), zf] with
 * S = n_support (fold 0) or 2*n_support (fold 1: supports k and k+n_support averaged, gnnnet_copy.py:67-72)
 * -> nodes [n_episodes*n_query*n_way*(n_support+1), ld] = [z | one-hot label (zero for the query slot) | 0..] */
int mft_build_graph_nodes(const float* z, int zf, float* nodes, int ld, int n_episodes, int n_way,
                          int n_support, int n_query, int fold, void* stream);
/* forward_gnn tail (gnnnet.py:215-216): scores[e, c*n_query+q, :] = out[node(e,q,c,last), :n_way] */
int mft_gather_query_scores(const float* out, int ldo, float* scores, int n_episodes, int n_way,
                            int n_support, int n_query, void* stream);
/* x_a_i[selected_id] (finetune.py:282; gnnnet.py:162): dst[r,:] = src[idx[r],:], rows of row_floats (%4==0) floats */
int mft_gather_rows(const float* src, const int* idx, float* dst, int n_rows, long long row_floats, void* stream);

/* backward-only entry points of the meta-training path (loss.backward() in MetaTemplate.train_loop*,
 * meta_template.py:76-109) --------------------------------------------------------------------------------------- */
/* BatchNorm backward for any group size (three launches: chunked partial sums, fixed-order finalize, elementwise dx) with
 * the derivative of the activation that followed the BN fused in: dy_eff = dy * act'(y_act) (y_act = post-activation
 * output, NULL for none).  ws: mft_bn_backward_ws_floats floats.  dx may be NULL (parameter gradients only).
 * dgamma / dbeta [n_groups, C] (nullable); dgamma_sum / dbeta_sum [C] (nullable): the same summed over the groups in group
 * order by the finalize launch (k episodes in lockstep share one BatchNorm's affine parameters);  dbias_zero [C] (nullable):
 * the gradient of a bias added in front of this BatchNorm (nn.Linear / Conv2d bias followed by train-mode BatchNorm: gnn.py:
 * 43-56,64-76, gnnnet.py:30) -- identically zero, written as zeros by the same launch instead of a column-sum of rounding noise. */
long long mft_bn_backward_ws_floats(int C, int rows_per_group, int n_groups);
int mft_bn_backward_act(const float* x, int ldx, const float* dy, int lddy, const float* y_act, int ldya,
                        float* dx, int lddx, int C, int rows_per_group, int n_groups,
                        const float* mean, const float* rstd, const float* gamma, long long gb_group_stride,
                        float* dgamma, float* dbeta, int act, float slope, float* ws, float* dgamma_sum, float* dbeta_sum,
                        float* dbias_zero, void* stream);
/* Several independent mft_bn_backward_act problems (up to 8; each with a dx) in ONE launch triple: the backward of SimpleBlock's BN2
 * and BNshortcut reads the same upstream gradient and ReLU output (backbone.py:256-260).  jobs: HOST array; a job = the same-named
 * arguments of mft_bn_backward_act (gamma shared by the groups); bit-identical to its own launches. */
typedef struct MftBnBwdJob {
    const float* x; const float* dy; const float* y_act; float* dx; const float* mean; const float* rstd; const float* gamma;
    float* dgamma; float* dbeta; float* ws; float* dgamma_sum; float* dbeta_sum; float* dbias_zero;
    int ldx, lddy, ldya, lddx, C, rows_per_group, n_groups, act; float slope; int reserved;
} MftBnBwdJob;
int mft_bn_backward_act_multi(const MftBnBwdJob* jobs, int n_jobs, void* stream);
/* dx (+)= dy * act'(y)  (ReLU / leaky-ReLU backward; GNN_nl's F.leaky_relu before the concat, gnn.py:160) */
int mft_act_backward(const float* dy, int lddy, const float* y, int ldy, float* dx, int lddx, int C, long long rows,
                     int act, float slope, int accumulate, void* stream);
/* out[c] = sum_r x[r][c] (bias gradients of nn.Linear / 1x1 nn.Conv2d); ws >= ceil(rows/256)*C floats */
int mft_colsum(const float* x, int ldx, int C, long long rows, float* out, float* ws, void* stream);
/* trunk[1..3] forward that also records each window's argmax (uint8, first maximum wins) for the backward pass */
int mft_bn_relu_maxpool_arg(const float* x, float* y, unsigned char* argmax, int n_img, int H, int W, int C,
                            int imgs_per_group, const float* mean, const float* rstd, const float* gamma,
                            const float* beta, void* stream);
/* MaxPool2d(3,2,1) + ReLU backward: dx [n,H,W,C] w.r.t. the BatchNorm output, from dy / argmax / y of the pooled map */
int mft_maxpool_relu_backward(const float* dy, const unsigned char* argmax, const float* y, float* dx,
                              int n_img, int H, int W, int C, void* stream);
/* Wcompute softmax backward: ds[(b,i,j)*ldds] = A*(dA - sum_j dA*A) */
int mft_masked_softmax_backward(const float* A, const float* dA, float* ds, int ldds, int n_graphs, int N, void* stream);
/* |x_i - x_j| backward, accumulated into dx */
int mft_pair_absdiff_backward(const float* x, int ldx, const float* dd, int ldd, float* dx, int lddx,
                              int n_graphs, int N, int F, void* stream);
/* gmul backward: dx[:, :F] (+)= dy[:, :F] + A^T dy[:, F:2F];  dA = dy[:, F:2F] x^T.  accumulate = 0: the first writer of a
 * node-gradient buffer overwrites its F columns (no zero fill of the buffer beforehand) */
int mft_graph_aggregate_backward(const float* A, const float* x, int ldx, const float* dy, int lddy, float* dx, int lddx,
                                 float* dA, int n_graphs, int N, int F, int accumulate, void* stream);
int mft_build_graph_nodes_backward(const float* dnodes, int ld, float* dz, int zf, int n_episodes, int n_way,
                                   int n_support, int n_query, int fold, void* stream);
/* (dbias, nullable [n_way]: column sums of dscores = the gradient of the bias of the layer that produced `out`, gnn.layer_last.fc) */
int mft_gather_query_scores_backward(const float* dscores, float* dout, int ldo, int n_episodes, int n_way,
                                     int n_support, int n_query, float* dbias, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MFT_HIP_H */
