import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(autouse=True)
def _release_gpu_memory(request):
    """GPU tests: models hold reference cycles (module <-> runtime <-> recorded launch lists), so their workspaces -- up to 250 GB in the
    256-clip ViT-L test -- die only when the cycle collector gets to them.  Collect after every GPU test and hand the cached blocks back:
    the next test (or the bench.py child process some tests start) then finds the memory the previous one used."""
    yield
    if request.node.get_closest_marker("gpu") is not None:
        import gc
        gc.collect()
        try:
            import torch
            if torch.cuda.is_available():
                torch.cuda.empty_cache()
        except Exception:
            pass
