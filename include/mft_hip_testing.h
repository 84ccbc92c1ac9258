/* Test and A/B hooks of libmft_hip.so -- NOT part of the drop-in ABI (include/mft_hip.h holds that: launchers only).
 *
 * The launchers of mft_hip.h pick a kernel form by shape (tile sizes, the parity-class walk of strided data gradients, the
 * stream-shaped weight-gradient kernel, ...).  The parity tests must exercise every form AT THE SAME SHAPES (bit-identity of a
 * form against another, or against float64), and the A/B tools under tools/ time one form against another; both need to force a
 * form.  These setters do that.  They are process-global state: tests/conftest.py resets them after every test, the product path
 * (engine, drivers, bench.py's default line) never calls them, and in a product build (no -DMFT_EXPERIMENTS) the codes of the
 * measured-slower families return MFT_EINVAL because those kernels are not compiled. */
#ifndef MFT_HIP_TESTING_H
#define MFT_HIP_TESTING_H
#ifdef __cplusplus
extern "C" {
#endif

/* force the forward tile of the mft_conv2d_nhwc launcher -- 1: 128x128, 2: 128x64, 3: 64x128, 4: 64x64, 5: 128x32; 0: automatic -- and, by the code
 * ranges documented at its definition (csrc/conv_igemm.hip), the forms of the weight-gradient / data-gradient / skinny kernels */
int mft_debug_set_conv_tile(int tile);
/* (code 11000 + r of the same setter: train-mode BatchNorm forward / backward of at most r rows per group run as their one-launch
 * forms (the forward-small launcher and the small path inside the BatchNorm-backward launchers of mft_hip.h); 11000 = never; default
 * r = 512) */
/* (code 2100 / 2101 of the same setter: the stem's weight gradient in torch's layout on the generic gather kernel / on the strip kernel,
 * the default) */
/* the same for the split-precision trunk convolutions (csrc/conv_x3.hip): 0 auto, 1: 128x64, 2: 128x128, code ranges at its definition */
int mft_debug_set_x3_tile(int tile);
/* every hook back to its default */
int mft_debug_reset(void);
/* mft_wgrad_adam_next_forward: 1 = correctly rounded division / square root in the Adam epilogue (default 0: v_rcp_f32 / v_sqrt_f32) */
void mft_wgrad_fwd_set_exact(int on);
/* mft_wgrad_adam_next_forward: 0 = natural workgroup order instead of one XCD per episode (placement only: no result changes) */
void mft_wgrad_fwd_set_xcd(int on);

#ifdef __cplusplus
}
#endif
#endif
