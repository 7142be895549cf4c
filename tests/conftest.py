import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(autouse=True)
def _per_device_library_settings_do_not_leak(request):
    """The library's per-device switches (include/mmdit_hip.h conventions block) outlive the objects that set them: a data-parallel model_trainer turns
    tile claiming on, and with it the planner's choice of kernel for the fused QKV launch (8-phase instead of wide-slot: same values to bf16 rounding,
    not bit for bit).  Every GPU test starts from the library's defaults, whatever ran before it."""
    yield
    if request.node.get_closest_marker("gpu") is None:
        return
    mod = sys.modules.get("sd3_amd._lib")
    if mod is not None and getattr(mod, "_lib", None) is not None:      # (only if a test has loaded the library)
        mod.lib().mmdit_gemm_set_claiming(0)
