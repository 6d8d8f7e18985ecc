import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as orc
    orc.lib()  # builds liboracle.so on first use (gcc only)
    return orc


def have_gpu() -> bool:
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.fixture(autouse=True)
def _fresh_library_state(request):
    """Every GPU test starts from the state of a freshly loaded libtmac_hip.so: kcfg table, tuning table, every
    tmac_hip_set_* / tmac_hip_debug_* knob, the host-pointer layer's caches and workspace (tmac_hip_reset_state).  Round 2's
    suite was order-dependent through exactly this state; knobs a test sets no longer need a `finally` to be undone."""
    if request.node.get_closest_marker("gpu") is not None:
        import torch  # noqa: F401  (first: libtmac_hip.so must bind to the HIP runtime torch brings, not load a second one)
        import tmac_amd
        tmac_amd.binding.check(tmac_amd.lib().tmac_hip_reset_state())
    yield
