import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def pytest_addoption(parser):
    # test infrastructure only (the library itself reads no environment variable and has no search path): run the suite
    # against an instrumented build of the same kernels -- tests/test_gpu_sentinel.py passes libevdr_sentinel.so here
    parser.addoption("--evdr-lib", default=None, help="file name (inside the package directory) of the libevdr build to load")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")
    lib = config.getoption("--evdr-lib")
    if lib:
        import evdr_amd  # noqa: F401
        from evdr_amd import _lib
        if os.path.basename(lib) not in ("libevdr.so", "libevdr_sentinel.so"):
            # control / experiment builds (scratch/_variants/) hold deliberately racing kernels: never through the suite
            raise pytest.UsageError(f"--evdr-lib accepts libevdr.so or libevdr_sentinel.so only (got {lib})")
        path = os.path.join(_lib.PKG_DIR, os.path.basename(lib))
        if not os.path.exists(path):
            raise pytest.UsageError(f"--evdr-lib: {path} does not exist (python -m evdr_amd.build --sentinel)")
        _lib.LIB_PATH = path


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN, name + ".npz"))
    return load
