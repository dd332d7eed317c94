"""A/B switches for the scratch scripts: the product library reads no environment; variants are forced through the
debug hooks of include/evdr.h.  `exp_lib()` loads the -DEVDR_EXPERIMENT build (stamped diagnostic instances)."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import evdr_amd  # noqa: E402,F401
from evdr_amd import _lib as L  # noqa: E402


def set_variant(v) -> int:
    return L.load().evdr_debug_set_fwd_variant(int(v))


def set_ppb(v) -> int:
    return L.load().evdr_debug_set_pages_per_block(int(v))


def last_kernel() -> str:
    return L.load().evdr_last_fwd_kernel().decode()


def use_experiment_build():
    """Build libevdr_exp.so and make the package's loader use it (call BEFORE anything loads libevdr.so)."""
    from evdr_amd import build
    L.LIB_PATH = build.build(experiment=True, verbose=False)
    lib = L.load()
    lib.evdr_experiment_set_dbg_buffer.argtypes = [ctypes.c_void_p]
    lib.evdr_experiment_set_dbg_buffer.restype = None
    return lib
