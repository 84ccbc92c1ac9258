"""Drop-in alias for the reference module methods.baselinetrain."""
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))))
import meta_fine_tuning_amd  # noqa: E402,F401
from meta_fine_tuning_amd.methods.baselinetrain import *  # noqa: E402,F401,F403
