import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_present() -> bool:
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.fixture(scope="session")
def native_lib():
    """The HIP library must be loadable wherever tests run (it cross-compiles without a GPU)."""
    from cadence_rag_amd import _native
    if not _native.LIB_PATH.exists():
        _native.build_native()
    return _native.load()


@pytest.fixture(scope="session")
def gpu(native_lib):
    """GPU tests go through the C ABI; a missing device or library is a hard failure, never a skip
    to some CPU path."""
    assert _gpu_present(), "test marked gpu but no GPU is visible"
    assert native_lib.crag_device_count() >= 1
    return native_lib
