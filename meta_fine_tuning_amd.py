"""Import shim: the product package lives in the directory ``meta-fine-tuning_amd/``
(name fixed by the build contract; not a valid Python identifier).  Importing
``meta_fine_tuning_amd`` loads that directory as a regular package."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "meta-fine-tuning_amd")
_spec = importlib.util.spec_from_file_location(
    "meta_fine_tuning_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["meta_fine_tuning_amd"] = _mod
_spec.loader.exec_module(_mod)
