import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# The chunk-invariance / two-stream tests pass small `rayschunk` values on purpose: the library must then cut the call exactly there
# (by default it treats the caller's value as a lower bound, neumesh_amd/renderer.py:_fused_chunk; one test switches that back on).
os.environ.setdefault("NEUMESH_RAYSCHUNK", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # no test may sit on a GPU box for ever: pytest-timeout (installed in this image) ends a test after 10 minutes -- the longest one takes
    # ~10 s on the GPU and ~10 s on the CPU
    # (method "thread": a watchdog thread ends the process -- a signal handler would never run while the interpreter waits inside a HIP call)
    if config.pluginmanager.hasplugin("timeout") and not getattr(config.option, "timeout", None):
        config.option.timeout = 600
        config.option.timeout_method = "thread"


@pytest.fixture(scope="session")
def cuda_device():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    return torch.device("cuda", 0)
