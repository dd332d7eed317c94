"""Counterpart of the reference's utils/utils.py for the pieces around the hot path: seeding, the JSON-line
logger whose `train.log` format `summary_results.py` parses, the AdamW factory, the compressed-npz checkpoint
writer and docid alignment.  TensorBoard is optional here (absent in this image)."""
import json
import logging
from pathlib import Path
from typing import Any, Dict, Optional, Tuple

import numpy as np
import torch

from .preprocess_data import _as_object_array


def tokens_to_object(P_pad_np: np.ndarray, pmask_np: np.ndarray) -> np.ndarray:
    """(N, L, D) + (N, L) bool -> object array of the unmasked rows of each page, float32."""
    out = np.empty(P_pad_np.shape[0], dtype=object)
    for i in range(P_pad_np.shape[0]):
        out[i] = P_pad_np[i][pmask_np[i]].astype(np.float32)
    return out


def set_seed(seed: int):
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


def log_json(logger, obj: Dict[str, Any]):
    logger.info(json.dumps(obj, ensure_ascii=False))


def get_logger(save_dir, name: str = "run", verbosity: int = 1, use_tb: bool = True):
    """-> (logger, tb).  `<save_dir>/train.log` is opened in append mode with the line format
    "[%(asctime)s][%(levelname)s] %(message)s" (the contract summary_results.py:35,68-87 regexes).
    tb is a SummaryWriter when tensorboard is importable and use_tb is set, else None."""
    save_dir = Path(save_dir)
    save_dir.mkdir(parents=True, exist_ok=True)
    level = {0: logging.DEBUG, 1: logging.INFO, 2: logging.WARNING}.get(verbosity, logging.INFO)
    logger = logging.getLogger(f"{name}@{save_dir}")
    logger.setLevel(level)
    logger.propagate = False
    if not logger.handlers:
        fmt = logging.Formatter("[%(asctime)s][%(levelname)s] %(message)s")
        for h in (logging.FileHandler(save_dir / "train.log", mode="a"), logging.StreamHandler()):
            h.setFormatter(fmt)
            h.setLevel(level)
            logger.addHandler(h)
    tb = None
    if use_tb:
        try:
            from torch.utils.tensorboard import SummaryWriter
            tb = SummaryWriter(log_dir=str(save_dir))
        except Exception:
            tb = None
    return logger, tb


def set_optimizer(name, param, lr, wd):
    if name == "adamw":
        return torch.optim.AdamW([param], lr=lr, weight_decay=wd)
    raise ValueError(f"unknown optimizer {name!r}")


def save_compressed_npz(save_path, docid, documents_obj, doc_attnmask_obj, doc_imgmask_obj, meta: Dict[str, Any]):
    """best_*.npz writer: docid / documents / doc_attnmask / doc_imgmask object arrays + a 0-d object `meta`."""
    save_path = Path(save_path)
    save_path.parent.mkdir(parents=True, exist_ok=True)
    payload = {"docid": _as_object_array(docid), "documents": _as_object_array(documents_obj)}
    if doc_attnmask_obj is not None:
        payload["doc_attnmask"] = _as_object_array(doc_attnmask_obj)
    if doc_imgmask_obj is not None:
        payload["doc_imgmask"] = _as_object_array(doc_imgmask_obj)
    payload["meta"] = np.array(meta, dtype=object)
    np.savez_compressed(str(save_path), **payload)
    print(f"[save] {save_path}")


def align_by_docid(docid_ref, docid_other, *arrays_to_perm) -> Tuple[Tuple[Optional[np.ndarray], ...], bool]:
    """Permute the `other` arrays into the reference docid order; (arrays unchanged, False) when that is impossible."""
    if docid_other is None:
        return arrays_to_perm, False
    ref = [str(x) for x in _as_object_array(docid_ref)]
    oth = [str(x) for x in _as_object_array(docid_other)]
    if len(ref) != len(oth):
        return arrays_to_perm, False
    where = {d: i for i, d in enumerate(oth)}
    if any(d not in where for d in ref):
        return arrays_to_perm, False
    perm = np.array([where[d] for d in ref], dtype=np.int64)
    return tuple(None if a is None else _as_object_array(a)[perm] for a in arrays_to_perm), True
