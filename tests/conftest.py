import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # a wedged rendezvous or child process must fail its test, not hang the run (pytest-timeout, when installed; the slowest test,
    # an fp64 oracle pass over a 50-shot meta-training episode, takes 2-3 minutes on 8 CPU threads)
    if config.pluginmanager.hasplugin("timeout") and not getattr(config.option, "timeout", None):
        config.option.timeout = float(os.environ.get("MFT_TEST_TIMEOUT", "1200"))


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def pytest_collection_modifyitems(config, items):
    """GPU tests fail loudly (not skip) when selected on a box without a GPU; on a CPU
    box they are deselected by `-m "not gpu"`."""
    return


@pytest.fixture(autouse=True)
def _restore_debug_knobs(request):
    """The library's tuning knobs are process-global: put every one back to its default after each GPU test, whether it passed
    or not (mft_debug_reset), so a failed assert cannot leave later tests on a different kernel variant."""
    yield
    if request.node.get_closest_marker("gpu") is not None:
        import meta_fine_tuning_amd  # noqa: F401
        from meta_fine_tuning_amd import _lib
        _lib.lib().mft_debug_reset()
