import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    import gan_class_transfer2_amd as g
    g._lib.load()
    g._lib.call("gct2_device_check")
    return torch.device("cuda", 0)


@pytest.fixture(autouse=True)
def _clean_library_state(request):
    """the library's scratch registrations and tuning hooks are process-wide: every GPU test starts from the defaults (no
    scratch registered - engines register their own on first use - and automatic tile choice)."""
    if "gpu" in request.fixturenames:
        request.getfixturevalue("gpu")
        import gan_class_transfer2_amd as g
        g.engine.reset_workspace_registration()
        g._lib.load().gct2_debug_tapgemm_variant(0)
    yield
