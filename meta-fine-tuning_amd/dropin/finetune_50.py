"""Drop-in alias: put this directory on sys.path ahead of the reference checkout and `import finetune_50`
resolves to the MI355X implementation."""
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))))
import meta_fine_tuning_amd  # noqa: E402,F401
from meta_fine_tuning_amd.finetune_50 import *  # noqa: E402,F401,F403
from meta_fine_tuning_amd import finetune_50 as _impl  # noqa: E402
