import os
import sys
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


# torch must initialise its HIP context BEFORE libyolo_hip.so makes the first HIP call in this process, otherwise
# torch.cuda reports no devices on the GPU box; do it once at collection time.
_TORCH_SEES_GPU = has_gpu()
if _TORCH_SEES_GPU:
    import torch
    torch.cuda.init()


@pytest.fixture(scope="session")
def hiplib():
    """The production library; GPU tests fail (not skip) if it is missing so a silent fallback cannot pass."""
    from yolo_tensorflow_amd import hip
    hip.load_library()
    return hip
