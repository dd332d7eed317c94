"""driver.CheckpointWriter: the `best_*.npz` files of a run are compressed and written by one background thread (the
reference writes them inline, mainv2_iter_distill_infonce.py:394-426 -> utils/utils.py:83-103; at 500 x 206 x 128 fp32 pages that is
~2 s of zlib per file, several epochs of the fused training loop).  Same files, submission order, newest snapshot per file wins,
errors surface on the caller's thread."""
import os
import threading
import time
from pathlib import Path
from types import SimpleNamespace

import numpy as np
import pytest
import torch


def test_files_are_written_in_order_and_a_waiting_snapshot_is_superseded(tmp_path):
    import evdr_amd  # noqa: F401
    from evdr_amd import driver
    w = driver.CheckpointWriter()
    done, gate = [], threading.Event()

    def job(tag, wait=False):
        def run():
            if wait:
                gate.wait(5)
            done.append((tag, threading.current_thread().name))
        return run

    w.submit("a", job("a1", wait=True))       # occupies the thread ...
    w.submit("b", job("b1"))                  # ... so these two wait, and b1 is superseded by b2
    w.submit("b", job("b2"))
    w.submit("c", job("c1"))
    gate.set()
    w.drain()
    assert [t for t, _ in done] == ["a1", "b2", "c1"]
    assert all(name == "evdr-checkpoint-writer" for _, name in done)

    def bad():
        raise ValueError("disk full")
    w.submit("d", bad)
    with pytest.raises(RuntimeError, match="disk full"):
        w.drain()
    w.submit("e", job("e1"))                  # the writer keeps working after a failure has been reported
    w.close()
    assert done[-1][0] == "e1"
    w.close()                                 # idempotent


def test_async_checkpoint_equals_the_inline_one(tmp_path):
    """save_best_npz through the writer leaves the same file as the inline call (same arrays, same meta), written atomically, and
    the snapshot is taken at submission: later changes of the parameter do not reach the file."""
    import evdr_amd  # noqa: F401
    from evdr_amd import driver
    from evdr_amd.utils.preprocess_data import _as_object_array
    g = torch.Generator().manual_seed(0)
    P = torch.randn(6, 9, 128, generator=g)
    pm = torch.rand(6, 9, generator=g) > 0.3
    pm[:, 0] = True
    metrics = {"Recall": {"Recall@1": 0.5}, "NDCG": {"NDCG@5": 0.6}, "latency": 0.001}
    best = {"step": 7, "Recall@1": 0.5, "NDCG@5": 0.6}
    args = SimpleNamespace(temp=0.1, lr=1e-3)
    common = dict(dataset="synth", mf=4, step=7, best=best, metrics=metrics, pmask_student=pm,
                  docid_tr=_as_object_array([f"p{i}" for i in range(6)]), doc_attn_in=None, doc_img_in=None, args=args)
    (tmp_path / "a").mkdir()
    (tmp_path / "b").mkdir()
    driver.save_best_npz(out_dir=tmp_path / "a", fname="best_ndcg5.npz", Pbar_param=P, **common)
    w = driver.CheckpointWriter()
    Pw = P.clone()
    driver.save_best_npz(out_dir=tmp_path / "b", fname="best_ndcg5.npz", Pbar_param=Pw, writer=w, **common)
    Pw.add_(100.0)                                           # after the submission: must not reach the file
    best["step"] = 99
    w.close()
    assert sorted(os.listdir(tmp_path / "b")) == ["best_ndcg5.npz"]          # no temporary left behind
    za, zb = (np.load(Path(tmp_path) / d / "best_ndcg5.npz", allow_pickle=True) for d in ("a", "b"))
    assert sorted(za.files) == sorted(zb.files)
    for da, db in zip(za["documents"], zb["documents"]):
        assert da.dtype == np.float32 and np.array_equal(da, db)
    assert list(za["docid"]) == list(zb["docid"])
    assert zb["meta"].item()["step"] == 7 and zb["meta"].item()["best"]["step"] == 7
    assert za["meta"].item() == {**zb["meta"].item()}
