"""Helpers around the hot path with the names the reference's drivers import from `utils.utils`: the JSON-line run log
whose `train.log` lines `summary_results.py` parses, seeding, the optimizer factory, the `best_*.npz` writer and the
docid re-alignment of a second payload.  Nothing here touches the GPU; TensorBoard is used only if it is installed."""
import json
import logging
import os
from typing import Any, Dict, Optional, Sequence, Tuple

import numpy as np
import torch

from .preprocess_data import _as_object_array

# line format of train.log: summary_results.py:35,68-87 matches "[<time>][<LEVEL>] <json>"
_LOG_LINE = "[%(asctime)s][%(levelname)s] %(message)s"
_VERBOSITY = (logging.DEBUG, logging.INFO, logging.WARNING)


# ---- run log ----------------------------------------------------------------------------------------------------------
def get_logger(save_dir, name: str = "run", verbosity: int = 1, use_tb: bool = True):
    """(logger, tb): a logger that appends to `<save_dir>/train.log` and echoes to the console, one handler pair per
    (name, directory) however often it is asked for; `tb` is a TensorBoard SummaryWriter, or None when `use_tb` is off or
    the package is missing."""
    directory = os.fspath(save_dir)
    os.makedirs(directory, exist_ok=True)
    level = _VERBOSITY[verbosity] if 0 <= verbosity < len(_VERBOSITY) else logging.INFO
    logger = logging.getLogger(f"{name}@{directory}")
    logger.propagate = False
    logger.setLevel(level)
    if not logger.handlers:
        for sink in (logging.FileHandler(os.path.join(directory, "train.log"), mode="a"), logging.StreamHandler()):
            sink.setLevel(level)
            sink.setFormatter(logging.Formatter(_LOG_LINE))
            logger.addHandler(sink)
    writer = None
    if use_tb:
        try:
            from torch.utils.tensorboard import SummaryWriter
            writer = SummaryWriter(log_dir=directory)
        except Exception:          # tensorboard not installed: the text log is the contract, the event file is a bonus
            writer = None
    return logger, writer


def log_json(logger, obj: Dict[str, Any]):
    """One JSON object per log line (non-ASCII kept as is)."""
    logger.info(json.dumps(obj, ensure_ascii=False))


def log_dict(logger, tb, scalars: Dict[str, Any], step: int):
    """`{"step": step, **scalars}` as one JSON log line; the int / float entries also go to TensorBoard when a writer is
    given (utils/utils.py:62-75 of the reference).  Returns the logger like the reference does."""
    logger.info(json.dumps({"step": step, **scalars}, ensure_ascii=False))
    if tb is not None:
        for key, val in scalars.items():
            if isinstance(val, (int, float)):
                tb.add_scalar(key, val, step)
        tb.flush()
    return logger


# ---- reproducibility / optimizer --------------------------------------------------------------------------------------
def set_seed(seed: int):
    torch.manual_seed(seed)
    np.random.seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


class StreamAdamW(torch.optim.AdamW):
    """torch.optim.AdamW (same constructor, same `state` / `state_dict` layout: step, exp_avg, exp_avg_sq) whose `step()`
    updates every dense fp32 CUDA parameter with ONE kernel pass (evdr_adamw_step: 28 B per element) instead of torch's eight
    foreach passes -- 60 us against 207 for the 13.5 M student parameters of the reference's step.  Same update rule, same
    bias corrections (double precision on the host).  The results agree with torch.optim.AdamW's to ROUNDING, not bit for
    bit: the fused expression rounds once where torch's foreach passes round per pass (a few ulp per step,
    tests/test_gpu_driver.py: rtol 2.4e-7 x steps).
    Eligibility is decided for the WHOLE step before anything is touched: every parameter with a gradient must be a dense
    fp32 CUDA tensor, 16-byte aligned, and so must its gradient and -- once they exist, e.g. after `load_state_dict` -- both
    moments (same device, shape, dense, aligned).  Anything else (CPU tensors, other dtypes, views with an odd storage
    offset, moments restored onto another device, amsgrad / maximize / capturable / differentiable) takes torch's own step
    for the whole optimizer, so a step is never applied half by one rule and half by the other."""

    def _plain(self, group) -> bool:
        return not (group.get("amsgrad") or group.get("maximize") or group.get("capturable") or group.get("differentiable"))

    @staticmethod
    def _dense_f32(t: torch.Tensor, like: torch.Tensor) -> bool:
        return (torch.is_tensor(t) and t.is_cuda and t.device == like.device and t.dtype == torch.float32 and not t.is_sparse
                and t.shape == like.shape and t.is_contiguous() and t.data_ptr() % 16 == 0)

    def _eligible(self, p: torch.Tensor) -> bool:
        if p.grad is None:
            return True
        if not (self._dense_f32(p, p) and self._dense_f32(p.grad, p)):
            return False
        st = self.state.get(p, {})
        if len(st) == 0:
            return True                       # moments are created below, like the parameter
        return all(k in st for k in ("step", "exp_avg", "exp_avg_sq")) and self._dense_f32(st["exp_avg"], p) and self._dense_f32(st["exp_avg_sq"], p)

    @torch.no_grad()
    def step(self, closure=None):
        fast = all(self._plain(g) and all(self._eligible(p) for p in g["params"]) for g in self.param_groups)
        if not fast:
            return super().step(closure)
        from .. import ops
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            lr = float(group["lr"]) if not torch.is_tensor(group["lr"]) else float(group["lr"].item())
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                if len(st) == 0:                                     # torch's own lazy state: a tensor step on the host
                    st["step"] = torch.tensor(0.0, dtype=torch.float32)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                t = int(float(st["step"])) + 1
                ops.adamw_step(p.grad, p, st["exp_avg"], st["exp_avg_sq"], lr, group["betas"], group["eps"],
                               group["weight_decay"], t)
                st["step"] += 1                                      # counted only once the update has been issued
        return loss


def set_optimizer(name, param, lr, wd):
    """The reference trains the student pages with torch's AdamW at default betas / eps (utils/utils.py:78-80); this is the
    same optimizer with a one-pass update kernel for CUDA parameters (`StreamAdamW`): same rule and state layout, results equal
    to torch's to rounding (ulp level per step), not bit for bit; `torch.optim.AdamW([param], lr=lr, weight_decay=wd)` is a
    drop-in replacement where bit-identity with torch's own kernels matters more than the 3.4x shorter update."""
    if name != "adamw":
        raise ValueError(f"unknown optimizer {name!r}")
    return StreamAdamW([param], lr=lr, weight_decay=wd)


# ---- checkpoint payloads ----------------------------------------------------------------------------------------------
def tokens_to_object(P_pad_np: np.ndarray, pmask_np: np.ndarray) -> np.ndarray:
    """Padded (N, L, D) pages + (N, L) validity -> object array of ragged float32 (Li, D) pages (the npz layout)."""
    pages = np.empty(len(P_pad_np), dtype=object)
    for slot, (page, keep) in enumerate(zip(P_pad_np, pmask_np.astype(bool))):      # element-wise: equal-length pages must
        pages[slot] = np.asarray(page[keep], dtype=np.float32)                      # not collapse into one 3-D array
    return pages


def save_compressed_npz(save_path, docid, documents_obj, doc_attnmask_obj, doc_imgmask_obj, meta: Dict[str, Any]):
    """Write a `best_*.npz`: object arrays `docid`, `documents` and, when given, `doc_attnmask` / `doc_imgmask`, plus the
    run's `meta` dict as a 0-d object array."""
    target = os.fspath(save_path)
    _write_npz(target, docid, documents_obj, doc_attnmask_obj, doc_imgmask_obj, meta)
    print(f"[save] {target}")


def _write_npz(target: str, docid, documents_obj, doc_attnmask_obj, doc_imgmask_obj, meta: Dict[str, Any], atomic: bool = False):
    """The file of `save_compressed_npz`; atomic=True writes `<target>.tmp.npz` first and renames it over the target, so that a
    reader (or a crash) never meets a half-written checkpoint (driver.CheckpointWriter)."""
    os.makedirs(os.path.dirname(target) or ".", exist_ok=True)
    fields = [("docid", docid), ("documents", documents_obj), ("doc_attnmask", doc_attnmask_obj),
              ("doc_imgmask", doc_imgmask_obj)]
    arrays = {key: _as_object_array(val) for key, val in fields if val is not None}
    arrays["meta"] = np.array(meta, dtype=object)
    if atomic:
        tmp = target + ".tmp.npz"
        np.savez_compressed(tmp, **arrays)
        os.replace(tmp, target)
    else:
        np.savez_compressed(target, **arrays)


def align_by_docid(docid_ref, docid_other, *arrays_to_perm) -> Tuple[Tuple[Optional[np.ndarray], ...], bool]:
    """Reorder arrays that follow `docid_other` into the order of `docid_ref`.  -> (arrays, True), or the inputs untouched
    and False when the two id lists are not permutations of each other (or there is no second list)."""
    if docid_other is None:
        return arrays_to_perm, False
    wanted: Sequence[str] = [str(d) for d in _as_object_array(docid_ref)]
    position = {str(d): i for i, d in enumerate(_as_object_array(docid_other))}
    if len(_as_object_array(docid_other)) != len(wanted) or not all(d in position for d in wanted):
        return arrays_to_perm, False
    order = np.fromiter((position[d] for d in wanted), dtype=np.int64, count=len(wanted))
    return tuple(None if a is None else _as_object_array(a)[order] for a in arrays_to_perm), True
