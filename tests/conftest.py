import os
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
for p in (os.path.join(ROOT, "opencl-path-tracer_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


def pytest_sessionfinish(session, exitstatus):
    """What the image gates of the GPU tests measured (gpu_util.record_margin) goes to gpurun_out/parity_margins.json."""
    try:
        import json
        import gpu_util
        if gpu_util.MARGINS:
            out = os.path.join(ROOT, "gpurun_out")
            os.makedirs(out, exist_ok=True)
            with open(os.path.join(out, "parity_margins.json"), "w") as f:
                json.dump(gpu_util.MARGINS, f, indent=1, sort_keys=True)
    except Exception:
        pass


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Build every native piece once per session (no-op when up to date)."""
    import __graft_entry__ as g
    g.build()


@pytest.fixture(scope="session")
def golden():
    import golden_io
    return golden_io.load()


def _gpu_available():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.fixture(scope="session")
def gpu():
    """GPU tests must run the HIP path: no GPU (or no extension) is a failure under -m gpu, not a skip."""
    from ptamd import device as D
    D.lib()
    if not _gpu_available():
        pytest.fail("gpu-marked test selected but no HIP device is visible (there is no CPU fallback)")
    return D
