"""MI355X-native episodic meta-fine-tuning engine (hot path of johncai117/Meta-Fine-Tuning).

Sub-modules mirror the reference's module names (backbone, io_utils, configs,
finetune, methods.gnn / gnnnet / gnnnet_copy / meta_template / baselinefinetune);
the arithmetic runs in hand-written HIP kernels for gfx950 behind the C-ABI of
``include/mft_hip.h`` (see ``_lib.py``).  There is no CPU fallback.
"""
__version__ = "0.1.0"
