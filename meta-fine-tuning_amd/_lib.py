"""ctypes binding of libmft_hip.so (C-ABI declared in include/mft_hip.h).

The product path has no CPU fallback: ``lib()`` raises if the shared library is
missing, and every wrapper raises ``RuntimeError`` on a non-zero return code.
"""
import ctypes
import os

PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(PKG, "libmft_hip.so")

_P = ctypes.c_void_p
_I = ctypes.c_int
_L = ctypes.c_longlong
_F = ctypes.c_float

# name -> argtypes; every function returns int unless listed in _RESTYPE
SIGNATURES = {
    "mft_version": [],
    "mft_device_info": [_P, _P],
    "mft_stream_create_cumask": [_P, _I, _P],
    "mft_stream_destroy": [_P],
    "mft_stream_probe": [_P, _P, _P, _L, _P],
    "mft_probe_placement": [_P, _I, _I, _P],
    "mft_event_create": [_P],
    "mft_event_record": [_P, _P],
    "mft_event_elapsed_ms": [_P, _P, _P],
    "mft_event_destroy": [_P],
    "mft_augment_views": [_P, _I, _I, _I, _P, _I, _P, _L, _L, _I, _P, _P, _P],
    "mft_nchw_to_nhwc": [_P, _P, _I, _I, _I, _I, _P],
    "mft_pack_oihw": [_P, _P, _I, _I, _I, _I, _I, _P],
    "mft_unpack_oihw": [_P, _P, _I, _I, _I, _I, _I, _P],
    "mft_pack_dgrad": [_P, _P, _I, _I, _I, _I, _I, _L, _L, _P],
    "mft_conv2d_nhwc": [_P, _I, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _L, _P],
    "mft_has_experiments": [],
    "mft_split_bf16x3": [_P, _P, _L, _P],
    "mft_split_bf16x3_multi": [_P, _I, _L, _P],
    "mft_conv2d_nhwc_x3": [_P, _I, _P, _L, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "mft_conv2d_x3_stats_ws_floats": [_I, _I, _I, _I, _I, _I, _I, _I],
    "mft_conv2d_nhwc_x3_bnstats": [_P, _I, _P, _L, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _F, _P, _P, _P, _P],
    "mft_conv2d_dgrad_nhwc": [_P, _I, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _L, _P],
    "mft_conv_ksplit_ws_floats": [_L, _I, _I],
    "mft_conv_ksplit_grouped_ws_floats": [_L, _I, _I, _I],
    "mft_conv2d_nhwc_ksplit_grouped": [_P, _I, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _L, _P, _P],
    "mft_conv2d_dgrad_nhwc_ksplit_grouped": [_P, _I, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _L, _P, _P],
    "mft_conv2d_nhwc_ksplit": [_P, _I, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P],
    "mft_conv2d_dgrad_nhwc_ksplit": [_P, _I, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P],
    "mft_conv2d_wgrad_oihw_ws_floats": [_I, _I, _I, _I, _I, _I, _I, _I, _I],
    "mft_conv2d_wgrad_oihw": [_P, _I, _P, _I, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P],
    "mft_conv2d_wgrad_oihw_multi": [_P, _I, _P],
    "mft_conv2d_wgrad_ws_floats": [_I, _I, _I, _I, _I, _I, _I, _I, _I, _I],
    "mft_conv2d_wgrad_nhwc": [_P, _I, _P, _I, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _L, _P, _P],
    "mft_conv2d_wgrad_adam_nhwc": [_P, _I, _P, _I, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _L, _I, _F, _F, _F,
                                   _F, _P],
    "mft_pair_softmax_ut_backward": [_P, _P, _P, _P, _P, _I, _I, _I, _P, _P],
    "mft_pair_bwd_stats_ws_floats": [_L, _I],
    "mft_pair_bn_act_backward": [_P, _I, _P, _I, _P, _P, _P, _P, _P, _P, _I, _L, _I, _L, _F, _P, _P, _P, _P, _P, _P],
    "mft_pair_absdiff_ut": [_P, _I, _P, _P, _I, _I, _I, _L, _L, _P],
    "mft_pair_dx_gather": [_P, _I, _P, _I, _P, _I, _I, _I, _I, _L, _L, _P],
    "mft_pack_oihw_multi": [_P, _I, _L, _P],
    "mft_wgrad_adam_next_forward": [_P, _I, _P, _I, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _L, _I, _P, _F, _F, _F,
                                    _F, _P, _I, _P, _P, _P, _P, _L, _P, _P, _P, _P, _P, _P, _P, _P, _F, _P],
    "mft_bn_stats_ws_floats": [_I, _I, _I],
    "mft_bn_stats": [_P, _I, _I, _I, _I, _F, _P, _P, _P, _P, _P, _F, _P, _P],
    "mft_bn_apply": [_P, _I, _P, _I, _I, _I, _I, _P, _P, _P, _P, _L, _P, _I, _P, _P, _P, _P, _I, _F, _P],
    "mft_bn_apply_multi": [_P, _I, _P],
    "mft_bn_stats_multi": [_P, _I, _P],
    "mft_bn_relu_maxpool": [_P, _P, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P],
    "mft_bn_relu_maxpool_gather": [_P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P],
    "mft_bn_image_moments": [_P, _I, _I, _I, _L, _P, _P, _P],
    "mft_bn_combine_moments": [_P, _P, _P, _I, _I, _I, _I, _F, _P, _P, _P],
    "mft_global_avgpool": [_P, _P, _I, _I, _I, _P],
    "mft_bn_small_forward": [_P, _I, _P, _I, _P, _I, _P, _I, _I, _I, _I, _P, _P, _P, _P, _L, _P, _P, _P, _P, _I, _F, _F, _P, _I, _P],
    "mft_bn_backward2": [_P, _P, _I, _P, _I, _P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _L, _P, _P, _P, _P, _P],
    "mft_ce_pool_backward": [_P, _P, _I, _I, _I, _I, _P, _P, _P, _P],
    "mft_bn_backward": [_P, _I, _P, _I, _P, _I, _P, _I, _I, _I, _I, _P, _P, _P, _L, _P, _P, _P],
    "mft_avgpool_relu_backward": [_P, _P, _P, _I, _I, _I, _P],
    "mft_cross_entropy": [_P, _I, _P, _I, _I, _I, _P, _P, _P],
    "mft_softmax_rows": [_P, _I, _P, _I, _I, _I, _P],
    "mft_cross_entropy_mean": [_P, _I, _P, _I, _I, _I, _P, _P, _P],
    "mft_cross_entropy_mean_backward": [_P, _I, _P, _I, _I, _I, _P, _P, _I, _P],
    "mft_adam_step": [_P, _P, _P, _P, _L, _I, _F, _F, _F, _F, _F, _P],
    "mft_sgd_step": [_P, _P, _P, _L, _I, _F, _F, _F, _F, _P],
    "mft_maml_delta": [_P, _P, _P, _L, _P],
    "mft_conv2d_wgrad_adam_dgrad_ws_floats": [_I, _I, _I, _I],
    "mft_conv2d_wgrad_adam_dgrad_nhwc": [_P, _I, _P, _I, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _L, _I, _F, _F, _F, _F, _P],
    "mft_conv2d_wgrad_adam_dgrad_nhwc_dev": [_P, _I, _P, _I, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _L, _P, _F, _F, _F, _P],
    "mft_col2im_bn_backward_small": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _P, _L, _P, _P, _P],
    "mft_stream_create_priority": [_I, _P, _P],
    "mft_ce_pool_bn_backward2": [_P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _L, _P, _P, _P, _P, _P, _P],
    "mft_pool_window_minmax": [_P, _P, _P, _L, _I, _I, _I, _P],
    "mft_stem_cache_fill": [_P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _P],
    "mft_conv2d_nhwc_x3_bnin_bnstats": [_P, _I, _P, _P, _P, _P, _L, _P, _I, _I, _I, _I, _I, _I, _I, _F, _P, _P, _P, _P],
    "mft_split_f16x2": [_P, _P, _L, _P],
    "mft_conv2d_nhwc_h2": [_P, _I, _P, _L, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "mft_conv2d_nhwc_h2_bnstats": [_P, _I, _P, _L, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _F, _P, _P, _P, _P],
    "mft_conv2d_nhwc_h2_bnin_bnstats": [_P, _I, _P, _P, _P, _P, _L, _P, _I, _I, _I, _I, _I, _I, _I, _F, _P, _P, _P, _P],
    "mft_bn_apply_x3ws": [_P, _I, _P, _I, _I, _I, _I, _P, _P, _P, _P, _I, _P, _P, _P, _I, _F, _F, _P, _P, _P, _P, _P],
    "mft_bn_apply_x3ws_fits": [_I, _I, _I],
    "mft_bn_running_ema": [_P, _P, _I, _P, _P, _I, _P, _I, _I, _F, _F, _P, _P, _P],
    "mft_bn_relu_pooled_gather_moments": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _I, _F, _P, _P, _P, _P, _P],
    "mft_bn_relu_pooled_gather": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P],
    "mft_bn_apply_planes": [_P, _I, _P, _I, _P, _L, _I, _I, _I, _P, _P, _P, _P, _L, _P, _I, _P, _P, _P, _P, _I, _F, _P],
    "mft_bn_relu_maxpool_gather_planes": [_P, _P, _P, _P, _L, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P],
    "mft_conv2d_nhwc_x3p_bnstats": [_P, _L, _I, _P, _L, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _F, _P, _P, _P, _P],
    "mft_ingest_episode_views": [_P, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P],
    "mft_block_entry_small_forward": [_P, _I, _P, _L, _P, _L, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _P, _L, _P, _P, _F, _P],
    "mft_block_exit_small_forward": [_P, _P, _L, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _P, _P, _L, _P, _P, _P, _P, _F, _P],
    "mft_conv2d_dgrad_bn_backward_small": [_P, _I, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _L, _P, _P, _P, _P, _P, _L,
                                           _P, _P, _P],
    "mft_linear_head_sgd_run": [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _P, _F, _F, _F, _F, _P],
    "mft_linear_head_adam_run": [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _P, _F, _F, _F, _F, _F, _P],
    "mft_adam_multi": [_P, _I, _I, _F, _F, _F, _F, _F, _F, _P],
    "mft_adam_hyper_advance": [_P, _P, _F, _F, _F, _P],
    "mft_adam_step_dev": [_P, _P, _P, _P, _L, _P, _F, _F, _F, _F, _P],
    "mft_conv2d_wgrad_adam_nhwc_dev": [_P, _I, _P, _I, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _L, _P, _F, _F,
                                       _F, _P],
    "mft_linear_head_step": [_P, _I, _P, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _I, _P, _I, _F, _F, _F, _F, _F, _P],
    "mft_linear_head_scores": [_P, _I, _I, _I, _I, _I, _P, _P, _P, _P],
    "mft_pair_absdiff": [_P, _I, _P, _I, _I, _I, _I, _P],
    "mft_masked_softmax": [_P, _I, _P, _I, _I, _P],
    "mft_pair_mlp_tiles_m": [_I, _I],
    "mft_pair_mlp_layer": [_P, _I, _I, _P, _P, _P, _P, _I, _I, _P, _P, _I, _I, _I, _I, _F, _P, _P, _P, _I, _P],
    "mft_bn_forward_small": [_P, _P],
    "mft_bn_forward_small_max_rows": [],
    "mft_gemm_rk": [_P, _I, _P, _I, _I, _P, _P, _I, _I, _I, _P],
    "mft_pair_mlp_tiles_m_rk": [_I, _I],
    "mft_pair_mlp_layer_rk": [_P, _I, _I, _P, _P, _P, _P, _I, _I, _P, _P, _I, _I, _I, _I, _F, _P, _P, _P, _P],
    "mft_pair_mlp_stats_finalize_rk": [_P, _P, _P, _I, _I, _I, _P, _P, _F, _P, _P, _P, _P, _P],
    "mft_pair_mlp_stats_finalize": [_P, _P, _P, _I, _I, _I, _P, _P, _F, _P, _P, _P, _P, _P],
    "mft_pair_mlp_score": [_P, _I, _P, _P, _P, _P, _F, _P, _I, _I, _I, _P],
    "mft_masked_softmax_ut": [_P, _P, _I, _I, _P],
    "mft_graph_aggregate": [_P, _P, _I, _P, _I, _I, _I, _I, _P],
    "mft_copy_cols": [_P, _I, _P, _I, _I, _I, _I, _I, _F, _P],
    "mft_build_graph_nodes": [_P, _I, _P, _I, _I, _I, _I, _I, _I, _P],
    "mft_gather_query_scores": [_P, _I, _P, _I, _I, _I, _I, _P],
    "mft_gather_rows": [_P, _P, _P, _I, _L, _P],
    "mft_var_to_rstd": [_P, _P, _I, _F, _P],
    "mft_bn_backward_ws_floats": [_I, _I, _I],
    "mft_bn_backward_act": [_P, _I, _P, _I, _P, _I, _P, _I, _I, _I, _I, _P, _P, _P, _L, _P, _P, _I, _F, _P, _P, _P, _P, _P],
    "mft_bn_backward_act_multi": [_P, _I, _P],
    "mft_act_backward": [_P, _I, _P, _I, _P, _I, _I, _L, _I, _F, _I, _P],
    "mft_colsum": [_P, _I, _I, _L, _P, _P, _P],
    "mft_bn_relu_maxpool_arg": [_P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P],
    "mft_maxpool_relu_backward": [_P, _P, _P, _P, _I, _I, _I, _I, _P],
    "mft_masked_softmax_backward": [_P, _P, _P, _I, _I, _I, _P],
    "mft_pair_absdiff_backward": [_P, _I, _P, _I, _P, _I, _I, _I, _I, _P],
    "mft_graph_aggregate_backward": [_P, _P, _I, _P, _I, _P, _I, _P, _I, _I, _I, _I, _P],
    "mft_build_graph_nodes_backward": [_P, _I, _P, _I, _I, _I, _I, _I, _I, _P],
    "mft_gather_query_scores_backward": [_P, _P, _I, _I, _I, _I, _I, _P, _P],
}
# test / A-B hooks (include/mft_hip_testing.h): form selection for tests/ and tools/, never called by the product path
TESTING_SIGNATURES = {
    "mft_debug_set_conv_tile": [_I],
    "mft_debug_set_x3_tile": [_I],
    "mft_debug_reset": [],
    "mft_wgrad_fwd_set_exact": [_I],
    "mft_wgrad_fwd_set_xcd": [_I],
}
_RESTYPE = {"mft_conv2d_x3_stats_ws_floats": _L, "mft_bn_stats_ws_floats": _L, "mft_conv2d_wgrad_ws_floats": _L, "mft_bn_backward_ws_floats": _L,
            "mft_conv2d_wgrad_adam_dgrad_ws_floats": _L, "mft_pair_bwd_stats_ws_floats": _L, "mft_conv_ksplit_ws_floats": _L, "mft_conv2d_wgrad_oihw_ws_floats": _L, "mft_conv_ksplit_grouped_ws_floats": _L}

_lib = None


def lib():
    """Load libmft_hip.so; raise loudly if it has not been built (no fallback path exists)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "libmft_hip.so not found at %s -- build it with `python -c 'import __graft_entry__ as g; g.build()'`; "
                "this package has no CPU or PyTorch fallback" % LIB_PATH)
        # Load order matters on this stack: the library's static initialisers register its code objects with the HIP runtime
        # (the one PyTorch already loaded, same SONAME); if that happens before PyTorch has initialised the device, every
        # later launch from this library fails with hipErrorNoDevice.  So bring the device up through PyTorch first.
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
            torch.empty(1, device="cuda")
        h = ctypes.CDLL(LIB_PATH)
        for name, argtypes in list(SIGNATURES.items()) + list(TESTING_SIGNATURES.items()):
            fn = getattr(h, name)          # AttributeError if the .so lacks a declared symbol
            fn.argtypes = argtypes
            fn.restype = _RESTYPE.get(name, _I)
        _lib = h
    return _lib


MFT_EINVAL = -22


def check(rc, what):
    if rc != 0:
        raise RuntimeError("%s failed with code %d (hipError_t / MFT_EINVAL=-22)" % (what, rc))


class LaunchTimer:
    """Measurement aid: ``with LaunchTimer() as t: ...`` brackets EVERY launcher call made through ``lib()`` inside the block with
    a pair of HIP events recorded on the stream the launcher enqueues on (its last argument), whichever stream that is -- the
    engine's raw priority / CU-masked streams included, which torch.cuda.Event (current stream only) cannot see.
    ``t.collect()`` -> {launcher name: [milliseconds per call, ...]} (synchronises).  A launcher that enqueues several kernels
    is timed as one unit.  Not for use under stream capture (an event record would become a graph node); eager phases only."""

    _SKIP = ("mft_event_", "mft_stream_", "mft_debug_", "mft_version", "mft_device_info", "mft_has_experiments",
             "mft_wgrad_fwd_set_", "mft_probe_placement")

    def __init__(self, only=None, keep_args=False):
        self.only = only                       # optional predicate(name) -> bool
        self.keep_args = keep_args             # also keep each call's C-ABI arguments (shapes: callers derive bytes / flops from them)
        self._pairs = []                       # (name, start, stop, args | None)
        self._free = []
        self._real = None

    def _event(self):
        if self._free:
            return self._free.pop()
        e = ctypes.c_void_p()
        check(self._real.mft_event_create(ctypes.byref(e)), "mft_event_create")
        return e

    def __getattr__(self, name):               # stands in for the CDLL while the block runs
        real = self.__dict__["_real"]
        fn = getattr(real, name)
        sig = SIGNATURES.get(name)
        if (sig is None or not sig or sig[-1] is not _P or name.endswith("_ws_floats") or name.startswith(self._SKIP)
                or (self.only is not None and not self.only(name))):
            return fn

        def timed(*args):
            stream = args[-1]
            a, b = self._event(), self._event()
            real.mft_event_record(a, stream)
            rc = fn(*args)
            if rc != 0:                        # refused (MFT_EINVAL: outside the launcher's domain, the host falls back) -- nothing ran
                self._free += [a, b]
                return rc
            real.mft_event_record(b, stream)
            self._pairs.append((name, a, b, args if self.keep_args else None))
            return rc
        return timed

    def __enter__(self):
        global _lib
        self._real = lib()
        assert not isinstance(self._real, LaunchTimer), "LaunchTimer blocks do not nest"
        _lib = self
        return self

    def __exit__(self, *exc):
        global _lib
        _lib = self._real
        if exc and exc[0] is not None:         # the block raised: nobody will collect() -- do not leave the event pairs behind
            for _, a, b, _ in self._pairs:
                self._free += [a, b]
            self._pairs = []
            self.close()
        return False

    def collect(self, calls=False):
        """-> {launcher: [ms per call, ...]}; ``calls=True``: the flat list [(launcher, ms, args | None), ...] in call order."""
        out, flat = {}, []
        ms = ctypes.c_float()
        for name, a, b, args in self._pairs:
            check(self._real.mft_event_elapsed_ms(a, b, ctypes.byref(ms)), "mft_event_elapsed_ms")
            out.setdefault(name, []).append(float(ms.value))
            flat.append((name, float(ms.value), args))
            self._free += [a, b]
        self._pairs = []
        return flat if calls else out

    def close(self):
        for e in self._free:
            self._real.mft_event_destroy(e)
        self._free = []
