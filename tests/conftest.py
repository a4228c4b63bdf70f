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


@pytest.fixture
def parity_log():
    """record(name, **numbers): measured errors of the step-level parity tests, merged into gpurun_out/parity.json (copied to
    profiles/rNN_parity.json after a GPU run) so that the achieved margins are tracked, not just pass/fail."""
    import json
    path = os.path.join(ROOT, "gpurun_out", "parity.json")

    def record(name, **vals):
        os.makedirs(os.path.dirname(path), exist_ok=True)
        try:
            with open(path) as f:
                data = json.load(f)
        except (OSError, ValueError):
            data = {}
        data[name] = {k: (float("%.4g" % v) if isinstance(v, float) else v) for k, v in vals.items()}
        with open(path, "w") as f:
            json.dump(data, f, indent=1, sort_keys=True)

    return record
