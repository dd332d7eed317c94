"""ctypes binding of the C-ABI library (include/evdr.h).  No fallback of any kind: if libevdr.so is
missing or a call fails, an exception is raised -- the GPU path is the only product path."""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(PKG_DIR, "libevdr.so")

EVDR_OK, EVDR_ERR_ARG, EVDR_ERR_SHAPE, EVDR_ERR_WORKSPACE, EVDR_ERR_HIP = 0, 1, 2, 3, 4
EVDR_F32, EVDR_BF16, EVDR_F16 = 0, 1, 2
EVDR_TOPK_MAX = 128

_i64, _i32, _f32, _f64, _sz, _vp = C.c_int64, C.c_int32, C.c_float, C.c_double, C.c_size_t, C.c_void_p

# name -> (restype, argtypes); mirrors include/evdr.h one to one (tests check the export list against the header)
SIGNATURES = {
    "evdr_version": (C.c_int, []),
    "evdr_last_error": (C.c_char_p, []),
    "evdr_pack_pmask": (C.c_int, [_vp, _i64, _i64, _vp, _vp, _vp]),
    "evdr_split_f32": (C.c_int, [_vp, _i64, _vp, _vp, _vp]),
    "evdr_split_f32_segments": (C.c_int, [_vp, _i64, _i64, _vp, _vp, _vp]),
    "evdr_maxsim_fwd_workspace": (_sz, [_i64, _i64, _i64, _i64, C.c_int]),
    "evdr_maxsim_fwd": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, C.c_int, _vp, _vp, _sz, _vp]),
    "evdr_maxsim_fwd_prepared": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _i64, _i64, _i64, _i64, C.c_int, _i64, _i64,
                                           _vp, _vp, _vp, _vp]),
    "evdr_maxsim_fwd_prepared_subset": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, C.c_int, _i64, _i64,
                                                  _vp, _vp, _vp, _vp, _vp]),
    "evdr_qcache_workspace": (_sz, [_i64]),
    "evdr_maxsim_fwd_prepared_cached": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, C.c_int, _i64, _i64, _vp, _vp,
                                                  _vp, _sz, _vp]),
    "evdr_maxsim_bwd": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _vp]),
    "evdr_maxsim_bwd_q_workspace": (_sz, [_i64, _i64, _i64, _i64]),
    "evdr_maxsim_bwd_q": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _vp, _sz, _vp]),
    "evdr_maxsim_bwd_adamw": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64,
                                        _f64, _f64, _f64, _f64, _f64, _i64, _f32, _vp, _vp]),
    "evdr_maxsim_bwd_adamw_planes": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64,
                                               _f64, _f64, _f64, _f64, _f64, _i64, _f32, _vp, _vp, _vp, _vp, _vp]),
    "evdr_adamw_advance": (C.c_int, [_vp, _f64, _f64, _vp]),
    "evdr_adamw_step": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _f64, _f64, _f64, _f64, _f64, _i64, _vp]),
    "evdr_l2norm_fwd": (C.c_int, [_vp, _vp, _i64, _i64, _f32, _vp, _vp, _vp]),
    "evdr_l2norm_bwd": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _i64, _f32, _vp, _vp]),
    "evdr_l2norm_fwd_split": (C.c_int, [_vp, _vp, _i64, _i64, _f32, _vp, _vp, _vp, _vp, _vp, _i64, _vp]),
    "evdr_flag_nonfinite": (C.c_int, [_vp, C.c_int, _vp, _i64, _i64, _i64, _vp, _vp]),
    "evdr_topk_workspace": (_sz, [_i64, _i64, C.c_int]),
    "evdr_topk": (C.c_int, [_vp, _vp, _i64, _i64, _i64, _i32, C.c_int, _vp, _vp, _vp, _sz, _vp]),
    "evdr_maxsim_topk_workspace": (_sz, [_i64, _i64]),
    "evdr_maxsim_topk": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, C.c_int, _i64, _i64, _vp, _vp, _i32, C.c_int,
                                   _vp, _vp, _vp, _sz, _vp]),
    "evdr_infonce_distill_fwd_bwd": (C.c_int, [_vp, _vp, _i64, _i64, _f32, _vp, _vp, _vp, _vp]),
    "evdr_infonce_distill_fwd_bwd_ws": (C.c_int, [_vp, _vp, _i64, _i64, _f32, _vp, _vp, _vp, _vp]),
    "evdr_debug_set_fwd_variant": (C.c_int, [C.c_int]),
    "evdr_debug_set_pages_per_block": (C.c_int, [C.c_int]),
    "evdr_last_fwd_kernel": (C.c_char_p, []),
}


class EvdrQCache(C.Structure):
    """include/evdr.h `EvdrQCache`: plain host struct of device pointers + geometry (evaluator/retrieval.py builds one per cached tensor)."""
    _fields_ = [("slots", _vp), ("n_slots", _i64), ("ent_hash", _vp), ("ent_k", _vp), ("ent_q", _vp), ("ent_mask", _vp),
                ("ent_scores", _vp), ("n_entries", _vp), ("capacity", _i64), ("row_bytes", _i64), ("lq", _i64), ("np", _i64),
                ("hash_mask", C.c_uint64)]


class EvdrError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"libevdr status {code}: {msg}")
        self.code = code


ABI_VERSION = 303                  # EVDR_VERSION_NUM of csrc/evdr_common.h these signatures belong to

_lib: Optional[C.CDLL] = None


def load() -> C.CDLL:
    """Load libevdr.so (built by `evdr_amd.build.build()`); raises if it is not there."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: the HIP extension has not been built. Run `python -c 'import "
            f"__graft_entry__ as g; g.build()'` (needs hipcc). There is no CPU fallback for this path.")
    lib = C.CDLL(LIB_PATH)
    lib.evdr_version.restype = C.c_int
    if lib.evdr_version() != ABI_VERSION:      # a stale build would be called with the wrong argument lists
        raise RuntimeError(f"{LIB_PATH} has ABI version {lib.evdr_version()}, these bindings are written for {ABI_VERSION}: "
                           f"rebuild it (`python -c 'import __graft_entry__ as g; g.build()'`)")
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError here = header / library out of sync
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int) -> None:
    if rc != EVDR_OK:
        msg = load().evdr_last_error()
        raise EvdrError(rc, msg.decode("utf-8", "replace") if msg else "")


def ptr(t) -> Optional[int]:
    """Device pointer of a torch tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


def current_stream_handle(device) -> int:
    """hipStream_t of torch's current stream on `device`, as an integer.  The raw getter (what torch's own compiled-code
    launchers use) when this torch has it: torch.cuda.current_stream builds a Stream object per call, ~2 us of host time
    in front of every launch."""
    import torch
    raw = getattr(torch._C, "_cuda_getCurrentRawStream", None)
    if raw is not None:
        idx = device.index if isinstance(device, torch.device) else torch.device(device).index
        return raw(torch.cuda.current_device() if idx is None else idx)
    return torch.cuda.current_stream(device).cuda_stream


class _Here:
    """No-op context: the calling thread's current device already is the one the launch targets."""

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


_HERE = _Here()


def on(device):
    """Context that makes `device` current for a launch: torch.cuda.device(device) only when it is not current already
    (that guard costs ~3 us of host time per call, and a training step makes a dozen calls in front of a host sync)."""
    import torch
    idx = device.index
    if idx is None or idx == torch.cuda.current_device():
        return _HERE
    return torch.cuda.device(device)
