"""Train / eval driver for InfoNCE distillation of page embeddings: the counterpart of the call pattern and the log
contract of the reference's `mainv2_iter_distill_infonce.py` (§5 "Metrics / logging").

Not a port of that script: the step and the evaluation are organised around what is resident on the GPU.
  * the frozen teacher pages are prepared ONCE as a resident corpus (fp16 hi/lo planes + packed masks), so a
    step never re-reads or re-splits the 1030-patch fp32 teacher tensor (the reference re-uploads nothing but
    recomputes the teacher scores from fp32 every step, :283);
  * evaluation ranks on the device (top-100 per query, two D2H copies) instead of `.item()`-ing every score into
    a dict (:311-317); the reported latency is synchronised wall time per query;
  * optional teacher-score cache keyed by training-query index: teacher scores are constant per query over the
    whole run ("a legal, result-identical optimisation").
What is kept exactly: the arithmetic of a step (masked MaxSim of teacher and student, CE against the teacher's
top-1, AdamW on the raw student embeddings through l2_normalize and the mask), the CLI flags, the directory
layout `<out_root>/<name>/mf<k>/<dataset>/`, `config.json`, `best_{recall,ndcg5}.npz`, and the `train.log` JSON
lines including the final `"summary/best_ndcg5"` line that `summary_results.py:35,68-87` parses.
"""
from __future__ import annotations

import argparse
import json
import time
from pathlib import Path
from typing import Any, Dict, Optional, Tuple

import numpy as np
import torch
import torch.nn as nn

from . import ops
from .corpus import PageCorpus, shard_range
from .criterion import infonce_distillation_loss
from .evaluator.metrics import EvalIndex, evaluate_topk, results_from_topk
from .evaluator.retrieval import CustomRetrievalEvaluator, score_multi_vector_masked
from .utils.preprocess_data import (_as_object_array, l2_normalize, load_init_payload, load_payload,
                                    load_query_payload, normalize_masked, preprocess_docs, preprocess_queries)
from .utils.utils import _write_npz, align_by_docid, get_logger, log_json, save_compressed_npz, set_optimizer, set_seed, tokens_to_object


class TeacherScorer:
    """Frozen teacher pages as a resident fp32-accurate corpus; scores(Qb, qmb) -> (B, N) fp32, no autograd."""

    def __init__(self, P_teacher_norm: torch.Tensor, pmask_teacher: torch.Tensor, cache_size: int = 0):
        self.corpus = PageCorpus.from_tensor(P_teacher_norm.detach().float(), pmask_teacher)
        self.cache: Optional[torch.Tensor] = None
        self.have: Optional[np.ndarray] = None          # host-side: which rows of the cache are filled (no device sync to ask)
        if cache_size > 0:
            dev = P_teacher_norm.device
            self.cache = torch.empty((cache_size, self.corpus.n_pages), dtype=torch.float32, device=dev)
            self.have = np.zeros(cache_size, dtype=bool)

    @torch.no_grad()
    def scores(self, Qb: torch.Tensor, qmb: torch.Tensor, qidx: Optional[torch.Tensor] = None, qplanes=None) -> torch.Tensor:
        """Teacher scores of the batch; with a cache and the batch's dataset indices `qidx` (a host tensor, as the
        DataLoader hands them out) a pseudo-query is scored once per run -- the teacher is frozen, so the cached rows are
        the same numbers (mainv2_iter_distill_infonce.py:282-283 recomputes them every epoch)."""
        if self.cache is None or qidx is None:
            return self.corpus.score(Qb.float(), qmb, qplanes=qplanes)
        idx_host = qidx.detach().cpu().numpy().astype(np.int64)
        idx_dev = qidx.to(self.cache.device, non_blocking=True)
        if not self.have[idx_host].all():
            sc = self.corpus.score(Qb.float(), qmb, qplanes=qplanes)
            self.cache[idx_dev] = sc
            self.have[idx_host] = True
            return sc                                       # the rows just written: no gather of them back out of the cache
        return self.cache[idx_dev]


def train_one_step(Qb, qmb, teacher, pmask_teacher, Pbar_param, pmask_student, opt, temp: float, chunk_p: int = 64,
                   qidx: Optional[torch.Tensor] = None) -> float:
    """One update, returns the loss (mainv2_iter_distill_infonce.py:269-292).  `teacher` is either the normalised
    teacher tensor (reference signature) or a TeacherScorer (resident, preferred)."""
    device = Pbar_param.device
    Qb = Qb.to(device, non_blocking=True)
    qmb = qmb.to(device, non_blocking=True)
    Psb = normalize_masked(Pbar_param, pmask_student)          # == l2_normalize(Pbar_param * pmask[..., None]), fused
    if isinstance(teacher, TeacherScorer):
        sc_t = teacher.scores(Qb, qmb, qidx)
    else:
        with torch.no_grad():
            sc_t = score_multi_vector_masked(Qb, teacher, qmb, pmask_teacher, chunk_p)
    sc_s = score_multi_vector_masked(Qb, Psb, qmb, pmask_student, chunk_p)
    loss = infonce_distillation_loss(sc_s, sc_t, temperature=temp)
    opt.zero_grad(set_to_none=True)
    loss.backward()
    opt.step()
    return float(loss.item())


_EVAL_INDEX: list = []          # [(relevant_docs, docidx_2_docid, qsidx_2_query, nq, n, k_values, EvalIndex, content stamp)], newest last


def eval_index(evaluator: CustomRetrievalEvaluator, relevant_docs, docidx_2_docid, qsidx_2_query, nq: int, n: int) -> EvalIndex:
    """The lookup tables of one evaluation set (docid ranks for trec_eval's tie rule, judged pairs, ideal gains), built on
    the first evaluation and reused while the SAME qrels / id-map objects come back (they are constant over a run: the
    reference rebuilds its results dict from them on every evaluation, mainv2_iter_distill_infonce.py:311-317)."""
    # identity of the three containers + a content stamp (sizes, number of judged pairs, first and last id of the map): a
    # dict that is refilled or grows in place is noticed; one whose entries are OVERWRITTEN in place between evaluations with
    # all counts unchanged is not -- treat the qrels / id maps of a run as immutable (the reference's are: loaded once from the npz)
    ks = tuple(evaluator.k_values)
    stamp = (len(relevant_docs), sum(len(v) for v in relevant_docs.values()), len(docidx_2_docid),
             docidx_2_docid.get("0") if n else None, docidx_2_docid.get(str(n - 1)) if n else None,
             len(qsidx_2_query) if qsidx_2_query is not None else -1)
    for ent in _EVAL_INDEX:
        if ent[0] is relevant_docs and ent[1] is docidx_2_docid and ent[2] is qsidx_2_query and ent[3:6] == (nq, n, ks) and ent[7] == stamp:
            return ent[6]
    qkeys = [str(qsidx_2_query[i]) if qsidx_2_query is not None else str(i) for i in range(nq)]
    docids = [docidx_2_docid[str(j)] for j in range(n)]
    index = EvalIndex(relevant_docs, qkeys, docids, ks)
    _EVAL_INDEX.append((relevant_docs, docidx_2_docid, qsidx_2_query, nq, n, ks, index, stamp))
    del _EVAL_INDEX[:-4]
    return index


@torch.no_grad()
def eval_retrieval(evaluator: CustomRetrievalEvaluator, Q_test_norm, qmask_test, Pbar_param, pmask_student,
                   relevant_docs_test, docidx_2_docid_test, qsidx_2_query_test, chunk_p: int = 64, k: int = 100,
                   shard_sizes=None, timing: Optional[Dict[str, float]] = None, keep: Optional[Dict[str, Any]] = None):
    """Retrieval metrics of the current student pages + "latency" (ms per query, synchronised).  With `shard_sizes`
    (page-sharded run) Pbar_param / pmask_student are this rank's pages and the score columns are all-gathered: every rank
    ends up with the same full score matrix and the same metrics.

    Scores, the top-k and the candidate counts of the tie rule stay on the device; ONE device-to-host copy brings
    (nq, 2k + 1) words back and `evaluate_topk` turns them into the metric tables with array operations -- no per-pair
    `.item()` (mainv2_iter_distill_infonce.py:311-317) and no per-query dict.  The numbers equal
    `compute_mteb_metrics(relevant_docs, results_from_topk(...))` bit for bit (tests/test_host_logic.py).
    `keep` (optional dict) receives the full (nq, N) student score matrix of this evaluation ("scores"): `evaluation_loss`
    right after it needs exactly these numbers (the reference scores the test queries three times per evaluation,
    mainv2_iter_distill_infonce.py:308,338,340).
    `timing` (optional dict) receives the split of the call: device_ms (score + top-k, by HIP events), d2h_ms (candidate
    counts of the tie rule + the one copy of the candidates), host_ms (index lookup + metric tables), total_ms (the whole
    call, the normalisation of the pages included)."""
    t_start = time.perf_counter()
    kk = min(k, 128)
    ev0, ev1 = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) if timing is not None else (None, None)
    P_now = l2_normalize(Pbar_param.detach() * pmask_student.unsqueeze(-1))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if ev0 is not None:
        ev0.record()
    scores = score_multi_vector_masked(Q_test_norm, P_now, qmask_test, pmask_student, chunk_p=chunk_p)
    if shard_sizes is not None:
        scores = gather_columns(scores, tuple(shard_sizes))
    if ev1 is None:
        ts, ti, tied = ops.topk_with_ties(scores, kk, to_host=True)   # ties cut at rank k: all tied candidates go to the metric
    else:                                                             # same work, with the device part fenced for the split
        ts_d, ti_d = ops.topk(scores, kk)
        ev1.record()
        ev1.synchronize()
        t_dev = time.perf_counter()
        ts, ti, tied = ops.topk_with_ties(scores, kk, to_host=True, have=(ts_d, ti_d))
    t1 = time.perf_counter()
    if keep is not None:
        keep["scores"] = scores
    latency_ms = (t1 - t0) * 1000 / max(Q_test_norm.shape[0], 1)
    nq, n = scores.shape
    index = eval_index(evaluator, relevant_docs_test, docidx_2_docid_test, qsidx_2_query_test, nq, n)
    if kk >= index.kmax or n <= kk:
        metrics = evaluate_topk(index, ts, ti, tied)
    else:                                                             # cut-offs beyond the device's top-k: the dict entry decides
        metrics = evaluator.compute_mteb_metrics(relevant_docs_test, results_from_topk(ts, ti, index.query_keys, index.docids, extra=tied))
    metrics["latency"] = float(latency_ms)
    if timing is not None:
        t2 = time.perf_counter()
        timing.update(device_ms=float(ev0.elapsed_time(ev1)), d2h_ms=(t1 - t_dev) * 1e3, host_ms=(t2 - t1) * 1e3,
                      total_ms=(t2 - t_start) * 1e3)
    return metrics


@torch.no_grad()
def evaluation_loss(Q_test_norm, qmask_test, teacher, pmask_teacher, Pbar_param, pmask_student, temp: float,
                    chunk_p: int = 64, shard_sizes=None, sc_s: Optional[torch.Tensor] = None,
                    teacher_cache: Optional[Dict[str, Any]] = None) -> float:
    """InfoNCE-distillation loss on the test queries (mainv2_iter_distill_infonce.py:324-344).
    `sc_s`: the full (nq, N) student scores of the current pages when the caller has them already (`eval_retrieval(keep=...)`
    of the same pages: the same kernel on the same inputs).  `teacher_cache` (a dict the caller keeps per dataset): the teacher's
    scores of the test queries are computed on the first evaluation and reused -- teacher and test queries never change
    (the reference recomputes both matrices on every evaluation; same numbers)."""
    # what the cached teacher scores belong to: these test queries (tensor identity + version), this mask, this teacher, this
    # sharding -- a dict reused for another dataset or teacher of the same shape recomputes instead of returning stale scores
    from .evaluator.retrieval import _tensor_key
    owner = (_tensor_key(Q_test_norm), _tensor_key(qmask_test),
             id(teacher) if isinstance(teacher, TeacherScorer) else _tensor_key(teacher), _tensor_key(pmask_teacher),
             tuple(shard_sizes) if shard_sizes is not None else None)
    sc_t = None
    if teacher_cache is not None and teacher_cache.get("owner") == owner:
        sc_t = teacher_cache.get("sc_t")
    if sc_t is None:
        if isinstance(teacher, TeacherScorer):
            sc_t = teacher.scores(Q_test_norm, qmask_test)
        else:
            sc_t = score_multi_vector_masked(Q_test_norm, teacher, qmask_test, pmask_teacher, chunk_p=chunk_p)
        if shard_sizes is not None:
            sc_t = gather_columns(sc_t, tuple(shard_sizes))
        if teacher_cache is not None:
            teacher_cache["sc_t"], teacher_cache["owner"] = sc_t, owner
    if sc_s is None:
        Psb = l2_normalize(Pbar_param * pmask_student.unsqueeze(-1))
        sc_s = score_multi_vector_masked(Q_test_norm, Psb, qmask_test, pmask_student, chunk_p=chunk_p)
        if shard_sizes is not None:
            sc_s = gather_columns(sc_s, tuple(shard_sizes))
    return float(infonce_distillation_loss(sc_s, sc_t, temperature=temp).item())


def update_best(best: Optional[Dict[str, Any]], metrics: Dict[str, Any], step: int, kind: str) -> Tuple[Dict[str, Any], bool]:
    """kind 'r1': Recall@1 then NDCG@5 as tie-break; 'nd5': the other way round."""
    r1, nd5 = float(metrics["Recall"]["Recall@1"]), float(metrics["NDCG"]["NDCG@5"])
    cur = {"step": step, "Recall@1": r1, "NDCG@5": nd5}
    if best is None:
        return cur, True
    a, b = ("Recall@1", "NDCG@5") if kind == "r1" else ("NDCG@5", "Recall@1")
    better = cur[a] > best[a] or (cur[a] == best[a] and cur[b] > best[b])
    return (cur, True) if better else (best, False)


class CheckpointWriter:
    """`best_*.npz` files written by ONE background thread.  The reference writes them inline (mainv2_iter_distill_infonce.py:
    394-426 -> utils/utils.py:83-103, np.savez_compressed); for 500 x 206 x 128 fp32 pages that is ~2.1 s of zlib on the host per
    file -- seven epochs' worth of the fused training loop (782 steps x 0.4 ms), twice per evaluation while both best metrics still
    improve.  Here the training loop takes only the device-to-host snapshot (the values at the moment of the improvement) and goes
    on; the thread builds the object arrays, compresses into `<name>.tmp.npz` and renames it over `<name>` (a reader never sees a
    half-written file).  Files are written in submission order; a snapshot still waiting when a newer one for the same file
    arrives is replaced by it (at most one waits per file: `submit` never blocks and the host memory held is bounded by the number of
    file names); `drain()` waits for everything submitted so far, `close()` also ends the thread.  An exception in the
    thread is raised by the next submit / drain / close."""

    def __init__(self):
        import collections
        import threading
        self._pending: "collections.OrderedDict[str, Any]" = collections.OrderedDict()     # file -> newest snapshot's job, oldest file first
        self._cv = threading.Condition()
        self._busy = False
        self._stop = False
        self._error: Optional[BaseException] = None
        self._thread = threading.Thread(target=self._work, name="evdr-checkpoint-writer", daemon=True)
        self._thread.start()

    def _work(self):
        while True:
            with self._cv:
                while not self._pending and not self._stop:
                    self._cv.wait()
                if not self._pending:
                    return
                _, job = self._pending.popitem(last=False)
                self._busy = True
            try:
                job()
            except BaseException as e:                      # noqa: BLE001  (reported on the caller's thread)
                self._error = e
            finally:
                with self._cv:
                    self._busy = False
                    self._cv.notify_all()

    def _check(self):
        if self._error is not None:
            e, self._error = self._error, None
            raise RuntimeError(f"checkpoint writer failed: {type(e).__name__}: {e}") from e

    def submit(self, path: str, job) -> None:
        """Never blocks: at most one snapshot per file waits (a newer one takes the older one's place in the line)."""
        self._check()
        with self._cv:
            self._pending[path] = job
            self._cv.notify_all()

    def drain(self) -> None:
        with self._cv:
            while self._pending or self._busy:
                self._cv.wait()
        self._check()

    def backpressure(self, max_waiting: int) -> None:
        """Wait until at most `max_waiting` snapshots are waiting (bounds the host memory they hold when files are produced
        faster than one thread compresses them)."""
        with self._cv:
            while len(self._pending) > max_waiting:
                self._cv.wait()
        self._check()

    def close(self) -> None:
        if self._thread.is_alive():
            self.drain_quiet()
            with self._cv:
                self._stop = True
                self._cv.notify_all()
            self._thread.join()
        self._check()

    def drain_quiet(self) -> None:
        with self._cv:
            while self._pending or self._busy:
                self._cv.wait()


def save_best_npz(*, out_dir: Path, fname: str, dataset: str, mf: int, step: int, best, metrics, Pbar_param,
                  pmask_student, docid_tr, doc_attn_in, doc_img_in, args, writer: Optional[CheckpointWriter] = None):
    """One `best_*.npz` (schema: utils/utils.py:83-103 of the reference).  The snapshot of the parameters is taken HERE; with a
    `writer` the object arrays, the compression and the file are the background thread's work."""
    import os
    P_np = (Pbar_param.detach() * pmask_student.unsqueeze(-1)).cpu().numpy().astype(np.float32)
    pm_np = pmask_student.detach().cpu().numpy().astype(bool)
    meta = {"dataset": dataset, "mf": mf, "step": int(step),
            "best_type": "Recall@1" if fname == "best_recall.npz" else "NDCG@5", "best": dict(best) if isinstance(best, dict) else best,
            "eval": {"Recall@1": float(metrics["Recall"]["Recall@1"]), "NDCG@5": float(metrics["NDCG"]["NDCG@5"])},
            "latency": float(metrics["latency"]), "loss": "infonce_distillation_loss", "temp": args.temp, "lr": args.lr}
    target = os.fspath(out_dir / fname)
    docid = _as_object_array(docid_tr)

    def job():
        _write_npz(target, docid, tokens_to_object(P_np, pm_np), doc_attn_in, doc_img_in, meta, atomic=True)
        print(f"[save] {target}")

    if writer is None:
        job()
    else:
        print(f"[save] queued {target} (step {int(step)})")      # "[save] <path>" follows when the file is on disk
        writer.submit(target, job)


def summary_record(last_metrics, best_r1, best_nd5) -> Dict[str, Any]:
    """The last line of a run's train.log: what the reference's summary_results.py looks for (`"summary/best_ndcg5"` with
    `NDCG@5` / `Recall@1`, summary_results.py:35,68-87; emitted by the reference at mainv2_iter_distill_infonce.py:254-259)."""
    return {"summary/latency": float(last_metrics.get("latency", 0.0)), "summary/best_recall": best_r1,
            "summary/best_ndcg5": best_nd5, "note": "training finished"}


def log_eval(logger, tb, *, dataset: str, mf: int, step: int, metrics, loss: float):
    if tb is not None:
        tb.add_scalar("eval/Recall@1", float(metrics["Recall"]["Recall@1"]), step)
        tb.add_scalar("eval/NDCG@5", float(metrics["NDCG"]["NDCG@5"]), step)
        tb.add_scalar("eval/loss", float(loss), step)
    log_json(logger, {"dataset": dataset, "mf": mf, "step": int(step), "eval/loss": float(loss),
                      "eval/Recall@1": float(metrics["Recall"]["Recall@1"]),
                      "eval/NDCG@5": float(metrics["NDCG"]["NDCG@5"]), "eval/latency": float(metrics["latency"])})


def build_argparser():
    p = argparse.ArgumentParser(description="InfoNCE distillation of compressed page embeddings (MI355X)")
    p.add_argument("--datasets", type=str, nargs="+", required=True)
    p.add_argument("--mapping_json", type=str, required=True,
                   help="JSON {dataset: {pseudoQ, split_before, mf5, mf10, ...: npz filename}} (the reference's DATASETMAP)")
    p.add_argument("--query_root", type=str, default=".")
    p.add_argument("--teacher_root", type=str, default=".")
    p.add_argument("--init_root", type=str, default=".")
    p.add_argument("--mfs", type=int, nargs="+", default=[5, 10, 25, 50])
    p.add_argument("--out_root", type=str, default="results")
    p.add_argument("--name", type=str, default="infonce_distill train")
    p.add_argument("--max_steps", type=int, default=23460)
    p.add_argument("--eval_every", type=int, default=500)
    p.add_argument("--q_batch", type=int, default=32)
    p.add_argument("--opt", type=str, default="adamw")
    p.add_argument("--lr", type=float, default=1e-3)
    p.add_argument("--weight_decay", type=float, default=1e-2)
    p.add_argument("--temp", type=float, default=0.1)
    p.add_argument("--print_every", type=int, default=20)
    p.add_argument("--device", type=str, default="auto")
    p.add_argument("--seed", type=int, default=42)
    p.add_argument("--sync_checkpoints", action="store_true",
                   help="write best_*.npz inline like the reference (default: one background thread compresses and writes them; "
                        "the files are the same)")
    # The two result-identical fast paths are the DEFAULT: `python -m evdr_amd.driver` with the reference's flags alone runs the
    # fused 0.15-ms step (the teacher is frozen, caching its scores is "a legal, result-identical optimisation";
    # tests/test_gpu_driver.py pins both forms to the reference's own train_one_step).  --no_* restores the reference's call pattern.
    p.add_argument("--fused_step", dest="fused_step", action="store_true", default=None,
                   help="student update through evdr_maxsim_bwd_adamw (backward + normalise backward + AdamW in one kernel; "
                        "result-identical to autograd + torch.optim.AdamW).  Default: on when --opt is adamw")
    p.add_argument("--no_fused_step", dest="fused_step", action="store_false",
                   help="the reference's step: autograd through the drop-in scorer + the optimizer of --opt")
    p.add_argument("--cache_teacher_scores", dest="cache_teacher_scores", action="store_true", default=None,
                   help="keep the (n_train_queries, N) teacher score matrix on the device (result-identical: the teacher is frozen). "
                        "Default: on while it fits --teacher_cache_gb")
    p.add_argument("--no_cache_teacher_scores", dest="cache_teacher_scores", action="store_false",
                   help="recompute the teacher's scores every step like the reference (mainv2_iter_distill_infonce.py:282-283)")
    p.add_argument("--teacher_cache_gb", type=float, default=8.0,
                   help="byte budget of the teacher score cache per dataset (n_train_queries x pages-on-this-rank x 4 B); a cache that "
                        "does not fit is not made, which is logged, and the teacher is scored per step")
    return p


def resolve_fast_paths(args, n_train: int, n_pages_local: int, log=print) -> Tuple[bool, bool]:
    """(fused step?, teacher score cache?) for one dataset from the flags -- None = not given = the default: both on.  The fused
    kernel implements AdamW only: with another --opt the default falls back to the autograd step (an EXPLICIT --fused_step is an
    error there).  The cache is bounded by --teacher_cache_gb; a fallback is logged, never silent."""
    fused = getattr(args, "fused_step", None)
    if fused is None:
        fused = args.opt == "adamw"
        if not fused:
            log(f"[fast paths] --opt {args.opt}: the fused step implements AdamW only -> autograd step + torch optimizer")
    elif fused and args.opt != "adamw":
        raise ValueError("--fused_step implements AdamW only")
    cache = getattr(args, "cache_teacher_scores", None)
    explicit = cache is not None
    if cache is None:
        cache = True
    if cache:
        need = int(n_train) * int(n_pages_local) * 4
        budget = float(getattr(args, "teacher_cache_gb", 8.0)) * (1 << 30)
        if need > budget:
            log(f"[fast paths] teacher score cache {'(asked for) ' if explicit else ''}needs {need / (1 << 30):.2f} GiB "
                f"({n_train} queries x {n_pages_local} pages x 4 B) > --teacher_cache_gb {budget / (1 << 30):.2f}: not made, "
                f"the teacher is scored every step")
            cache = False
    return bool(fused), bool(cache)


def _dist_context():
    """(rank, world) of an initialised torch.distributed job, (0, 1) otherwise."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def gather_rows(block: torch.Tensor, sizes) -> torch.Tensor:
    """This rank's pages (n_local, ...) -> all pages (N, ...) on every rank (checkpoints of a page-sharded run)."""
    flat = block.reshape(block.shape[0], -1).t().contiguous()                       # (features, n_local): pages as columns
    return gather_columns(flat.float(), tuple(sizes)).t().reshape((int(sum(sizes)),) + tuple(block.shape[1:])).to(block.dtype)


LOAD_STATS: list = []           # one dict per dataset of the last run(): which pages this rank put on its GPU (tests read it)


def _shard_docs(docs_obj, attn_obj, img_obj, lo: int, hi: int, world: int, device):
    """(P_raw, pmask) of the pages [lo, hi) only: with world > 1 the object arrays are sliced on the host and padded to the
    longest page of the WHOLE dump (a pass over lengths, no copy), so no rank ever holds the (N, Lmax, 128) fp32 tensor of all
    pages."""
    if world == 1:
        P, pm, _ = preprocess_docs(docs_obj, attn_obj, img_obj, device)
        return P, pm
    docs = _as_object_array(docs_obj)
    lmax = max(int(t.shape[0]) for t in docs)
    d = int(docs[0].shape[1])
    if hi <= lo:                                                    # more ranks than pages: an empty shard of the right shape
        return (torch.zeros((0, lmax, d), dtype=torch.float32, device=device), torch.zeros((0, lmax), dtype=torch.bool, device=device))
    cut = lambda a: None if a is None else _as_object_array(a)[lo:hi]
    P, pm, _ = preprocess_docs(docs[lo:hi], cut(attn_obj), cut(img_obj), device, pad_to=lmax)
    return P, pm


def run(args) -> None:
    """`_run` with the run's checkpoint writer (one background thread for the `best_*.npz` files unless --sync_checkpoints);
    returns when every file is on disk."""
    writer = None if getattr(args, "sync_checkpoints", False) else CheckpointWriter()
    try:
        _run(args, writer)
    except BaseException:
        if writer is not None:                        # the run's own exception stays the primary error: the writer's, if any, is logged
            try:
                writer.close()
            except Exception as e:                    # noqa: BLE001
                print(f"[save] while handling the run's error: {e}")
        raise
    else:
        if writer is not None:
            writer.close()


def _run(args, writer: Optional[CheckpointWriter]) -> None:
    """Single process: the reference's loop.  Under torch.distributed (WORLD_SIZE > 1, see main): PAGE-SHARDED -- every rank
    holds the teacher and student pages [lo, hi) of its shard, the query batch is replicated, score columns are
    all-gathered (training: `sharded_*_train_one_step`; evaluation: `shard_sizes`), rank 0 logs and checkpoints."""
    set_seed(args.seed)
    LOAD_STATS.clear()
    rank, world = _dist_context()
    device = torch.device("cuda" if args.device == "auto" else args.device)
    mapping = json.loads(Path(args.mapping_json).read_text())
    # The feature dumps are zlib-compressed npz files (a 500-page teacher dump: ~1.8 s to inflate): while one dataset trains,
    # the NEXT dataset's two files are read by a background thread (zlib releases the GIL), one dataset ahead at most.
    from concurrent.futures import ThreadPoolExecutor

    def read_dataset(name: str):
        pt = mapping[name]
        inits = {f"mf{m}": load_init_payload(f"{args.init_root}/{pt[f'mf{m}']}") for m in args.mfs if f"mf{m}" in pt}   # a few tens of MB each
        qp, tp = load_query_payload(f"{args.query_root}/{pt['pseudoQ']}"), load_payload(f"{args.teacher_root}/{pt['split_before']}")
        # this rank's teacher pages padded to one (n, Lmax, 128) host tensor + mask here as well (pure copies: numpy releases the GIL)
        lo_, hi_ = shard_range(len(tp["documents"]), rank, world)
        host_pages = _shard_docs(tp["documents"], tp["doc_attnmask"], tp["doc_imgmask"], lo_, hi_, world, "cpu")
        return qp, tp, inits, host_pages

    loader = ThreadPoolExecutor(max_workers=1, thread_name_prefix="evdr-dataset-loader")
    try:
        _run_datasets(args, writer, loader, read_dataset, mapping, device, rank, world)
    finally:
        loader.shutdown(wait=False, cancel_futures=True)       # also on an exception: no read-ahead left running behind it


def _run_datasets(args, writer, loader, read_dataset, mapping, device, rank, world) -> None:
    ahead = {}
    for di, dataset in enumerate(args.datasets):
        paths = mapping[dataset]
        q_payload, t_payload, init_payloads, host_pages = ahead.pop(dataset).result() if dataset in ahead else read_dataset(dataset)
        if di + 1 < len(args.datasets) and args.datasets[di + 1] not in ahead:
            ahead[args.datasets[di + 1]] = loader.submit(read_dataset, args.datasets[di + 1])
        docid_tr = t_payload["docid"]
        Q_train, qmask_train = preprocess_queries(q_payload["query"], q_payload["query_attnmask"], device="cpu")
        if Q_train.numel() * 4 <= (8 << 30):        # the pseudo-queries (25 k x 32 x 128 fp32 = 0.4 GB) live in HBM: no H2D per step
            Q_train, qmask_train = Q_train.to(device), qmask_train.to(device)
        else:
            Q_train, qmask_train = Q_train.pin_memory(), qmask_train.pin_memory()
        Q_test, qmask_test = preprocess_queries(t_payload["query"], t_payload["query_attnmask"], device=device)
        # page-sharded run: this rank pads, uploads and normalises ONLY its pages [lo, hi) -- the object arrays are cut on the
        # host before anything reaches the GPU (padded to the dump's longest page, so that all shards have one shape)
        n_pages = len(t_payload["documents"])
        lo, hi = shard_range(n_pages, rank, world)
        shard_sizes = [shard_range(n_pages, r, world)[1] - shard_range(n_pages, r, world)[0] for r in range(world)] if world > 1 else None
        torch.cuda.reset_peak_memory_stats(device)
        P_t_raw, pmask_t = host_pages[0].to(device), host_pages[1].to(device)       # padded on the host by read_dataset
        del host_pages
        P_t_norm = l2_normalize(P_t_raw * pmask_t.unsqueeze(-1)).detach()
        n_train = Q_train.shape[0]
        stats = {"dataset": dataset, "rank": rank, "world": world, "n_pages": n_pages, "lo": lo, "hi": hi,
                 "teacher_rows_on_device": int(P_t_raw.shape[0]),
                 "peak_bytes_after_teacher_load": int(torch.cuda.max_memory_allocated(device))}
        LOAD_STATS.append(stats)
        use_fused, use_cache = resolve_fast_paths(args, n_train, int(P_t_norm.shape[0]), log=(print if rank == 0 else (lambda *_: None)))
        stats["fused_step"], stats["teacher_score_cache"] = use_fused, use_cache
        if rank == 0:
            print(f"[fast paths] {dataset}: fused step {'on' if use_fused else 'off'}, teacher score cache {'on' if use_cache else 'off'}"
                  + (f" ({n_train * int(P_t_norm.shape[0]) * 4 / (1 << 20):.1f} MiB)" if use_cache else ""))
        teacher = TeacherScorer(P_t_norm, pmask_t, cache_size=n_train if use_cache else 0)
        test_teacher_scores: Dict[str, Any] = {}              # the teacher's scores of the test queries: constant over the run
        del P_t_raw
        steps_per_epoch = (n_train + args.q_batch - 1) // args.q_batch
        eval_every = max(int(args.eval_every if args.eval_every and args.eval_every > 0 else steps_per_epoch), 1)

        for mf in args.mfs:
            key = f"mf{mf}"
            if key not in paths:
                raise ValueError(f"Missing mapping for {dataset}:{key}")
            init = init_payloads.pop(key)                            # read with the dataset's other files (read_dataset)
            Pbar_obj, attn_in, img_in = init["documents"], init["doc_attnmask"], init["doc_imgmask"]
            if init.get("docid") is not None:
                (Pbar_obj, attn_in, img_in), ok = align_by_docid(_as_object_array(docid_tr), _as_object_array(init["docid"]),
                                                                 Pbar_obj, attn_in, img_in)
                if ok:
                    print(f"[align] {dataset} mf{mf}: init matched by docid")
            if len(Pbar_obj) != n_pages:
                raise ValueError(f"init doc count mismatch: got {len(Pbar_obj)} vs teacher {n_pages}")
            Pbar_raw, pmask_s = _shard_docs(Pbar_obj, attn_in, img_in, lo, hi, world, device)
            stats[f"student_rows_on_device_mf{mf}"] = int(Pbar_raw.shape[0])
            if use_fused:
                student = FusedStudent(Pbar_raw, pmask_s, lr=args.lr, weight_decay=args.weight_decay)
                Pbar_param, opt = student.x, None                 # evaluated / checkpointed through the same tensor
            else:
                student = None
                Pbar_param = nn.Parameter(Pbar_raw * pmask_s.unsqueeze(-1))
                opt = set_optimizer(args.opt, Pbar_param, args.lr, args.weight_decay)
            out_dir = Path(args.out_root) / args.name / f"mf{mf}" / dataset
            if rank == 0:
                out_dir.mkdir(parents=True, exist_ok=True)
                logger, tb = get_logger(out_dir)
                cfg = out_dir / "config.json"
                if not cfg.exists():
                    cfg.write_text(json.dumps({"dataset": dataset, "mf": mf, **vars(args),
                                               "fast_paths": {"fused_step": use_fused, "teacher_score_cache": use_cache}},
                                              ensure_ascii=False, indent=2))
            else:                                                   # the other ranks compute the same numbers and stay silent
                import logging
                logger, tb = logging.getLogger(f"evdr.rank{rank}"), None
                logger.addHandler(logging.NullHandler())
                logger.propagate = False
            evaluator = CustomRetrievalEvaluator()
            ev_args = dict(evaluator=evaluator, Q_test_norm=Q_test, qmask_test=qmask_test, Pbar_param=Pbar_param,
                           pmask_student=pmask_s, relevant_docs_test=t_payload["relevant_docs"],
                           docidx_2_docid_test=t_payload["docidx_2_docid"], qsidx_2_query_test=t_payload["qsidx_2_query"],
                           shard_sizes=shard_sizes)
            el_args = dict(Q_test_norm=Q_test, qmask_test=qmask_test, teacher=teacher, pmask_teacher=pmask_t,
                           Pbar_param=Pbar_param, pmask_student=pmask_s, temp=args.temp, shard_sizes=shard_sizes)
            # one student pass per evaluation (shared by the ranking and the loss) and one teacher pass per dataset and mf
            kept: Dict[str, Any] = {}
            el_args["teacher_cache"] = test_teacher_scores
            metrics = eval_retrieval(**ev_args, keep=kept)
            log_eval(logger, tb, dataset=dataset, mf=mf, step=0, metrics=metrics, loss=evaluation_loss(**el_args, sc_s=kept.pop("scores")))
            log_json(logger, {"dataset": dataset, "mf": mf, "step": 0, "note": "init Pbar before training"})
            best_r1, _ = update_best(None, metrics, 0, "r1")
            best_nd5, _ = update_best(None, metrics, 0, "nd5")
            last = metrics

            gen = torch.Generator().manual_seed(args.seed)
            perm, cursor = torch.randperm(n_train, generator=gen), 0
            # resident pseudo-queries: the epoch's permutation goes to the device once, a batch's indices are a view of it
            # (no index upload per step) and the rows come out with index_select (half the host cost of advanced indexing)
            perm_dev = perm.to(Q_train.device) if Q_train.is_cuda else None
            # single-process runs: the epoch's batches are gathered once per epoch and -- fused steps -- split into planes in one launch (EpochBatches)
            # (the autograd path takes the gathered rows only: its scorer calls split the batch themselves)
            use_epoch = (world == 1 or student is not None) and perm_dev is not None and (student is None or args.q_batch * Q_train.shape[1] <= 2048)
            epoch = EpochBatches(Q_train, qmask_train, perm_dev, args.q_batch, planes=student is not None, teacher=teacher) if use_epoch else None
            t0, loss_sum, loss_cnt = time.time(), 0.0, 0
            # fused single-process steps leave their loss on the device; the host reads the pending ones when a line is due
            # (same numbers, same double-precision running sum in the same order: one sync per log line, not per step)
            pending: list = []

            def settle():
                nonlocal loss_sum, loss_val
                if pending:
                    vals = torch.stack([l for _, l in pending]).tolist()
                    for (s_no, _), v in zip(pending, vals):
                        loss_sum += v
                        if tb is not None:                           # the TensorBoard scalars are buffered with the losses
                            tb.add_scalar("train/loss", float(v), s_no)
                    pending.clear()
                    loss_val = vals[-1]

            defer = student is not None                             # fused steps (single process or page-sharded: the loss is the same number on every rank): no host wait per step
            loss_val = 0.0
            for step in range(1, args.max_steps + 1):
                if cursor >= n_train:                               # epoch boundary: reshuffle (DataLoader(shuffle=True))
                    perm, cursor = torch.randperm(n_train, generator=gen), 0
                    perm_dev = perm.to(Q_train.device) if Q_train.is_cuda else None
                    if use_epoch:
                        epoch = EpochBatches(Q_train, qmask_train, perm_dev, args.q_batch, planes=student is not None, teacher=teacher)
                idx = perm[cursor:cursor + args.q_batch]
                qidx = idx if use_cache else None
                if perm_dev is not None:
                    idx = perm_dev[cursor:cursor + args.q_batch]
                qpl_step = sct_step = None
                if use_epoch:
                    Qb_step, qmb_step, qpl_step = epoch.get(cursor // args.q_batch)
                    sct_step = epoch.teacher_scores(cursor // args.q_batch)
                else:
                    Qb_step, qmb_step = Q_train.index_select(0, idx), qmask_train.index_select(0, idx)
                cursor += args.q_batch
                if world > 1 and student is not None:
                    loss_val = sharded_fused_train_one_step(Qb_step, qmb_step, teacher, student, args.temp,
                                                           shard_sizes, qidx=qidx, qplanes=qpl_step, sync=not defer,
                                                           sc_t_local=sct_step)
                elif world > 1:
                    loss_val = sharded_train_one_step(Qb_step, qmb_step, teacher, Pbar_param, pmask_s, opt,
                                                     args.temp, shard_sizes, qidx=qidx)
                elif student is not None:
                    loss_val = fused_train_one_step(Qb_step, qmb_step, teacher, student, args.temp, qidx=qidx,
                                                    sync=not defer, qplanes=qpl_step, sc_t=sct_step)
                else:
                    loss_val = train_one_step(Qb_step, qmb_step, teacher, pmask_t, Pbar_param, pmask_s, opt,
                                              temp=args.temp, qidx=qidx)
                loss_cnt += 1
                if defer:
                    pending.append((step, loss_val))
                    if (args.print_every and step % args.print_every == 0) or step % eval_every == 0 or step == args.max_steps \
                            or len(pending) >= 256:
                        settle()
                else:
                    loss_sum += loss_val
                if tb is not None and not defer:
                    tb.add_scalar("train/loss", float(loss_val), step)
                if args.print_every and step % args.print_every == 0:
                    log_json(logger, {"dataset": dataset, "mf": mf, "step": step, "train/loss": float(loss_val),
                                      "train/avg_loss": float(loss_sum / max(loss_cnt, 1)), "time_sec": float(time.time() - t0)})
                if step % eval_every == 0 or step == args.max_steps:
                    metrics = eval_retrieval(**ev_args, keep=kept)
                    log_eval(logger, tb, dataset=dataset, mf=mf, step=step, metrics=metrics,
                             loss=evaluation_loss(**el_args, sc_s=kept.pop("scores")))
                    last = metrics
                    best_r1, upd_r1 = update_best(best_r1, metrics, step, "r1")
                    best_nd5, upd_nd5 = update_best(best_nd5, metrics, step, "nd5")
                    for upd, best, fname, tag in ((upd_r1, best_r1, "best_recall.npz", "best recall"),
                                                  (upd_nd5, best_nd5, "best_ndcg5.npz", "best nDCG@5")):
                        if upd:
                            logger.info(f"{tag} step| {step} | nDCG@5={best['NDCG@5']:.5f} | Recall@1={best['Recall@1']:.5f} "
                                        f"| Latency {metrics['latency']:.5f}")
                            P_all, pm_all = Pbar_param, pmask_s
                            if world > 1:                           # every rank takes part in the gather, rank 0 writes
                                P_all = gather_rows(Pbar_param.detach(), shard_sizes)
                                pm_all = gather_rows(pmask_s.float(), shard_sizes) > 0.5
                            if rank == 0:
                                save_best_npz(out_dir=out_dir, fname=fname, dataset=dataset, mf=mf, step=step, best=best,
                                              metrics=metrics, Pbar_param=P_all, pmask_student=pm_all, docid_tr=docid_tr,
                                              doc_attn_in=attn_in, doc_img_in=img_in, args=args, writer=writer)
            log_json(logger, summary_record(last, best_r1, best_nd5))
            if writer is not None:
                writer.backpressure(8)                              # the last files of this run may still be in the writer's hands while
            if rank == 0:                                           # the next (dataset, mf) trains; run() returns when all are on disk
                print(f"[done] {dataset} mf{mf} -> {out_dir}")
            if tb is not None:
                tb.flush()
                tb.close()


class EpochBatches:
    """The pseudo-queries of ONE epoch in batch order, prepared when the epoch starts instead of step by step.  The reference
    draws its batches from DataLoader(shuffle=True) (mainv2_iter_distill_infonce.py:81,168-178): the epoch's permutation fixes
    every batch up front, so the rows are gathered with ONE index_select per epoch (not two per step) and -- for the fused
    step -- split into the scorer's fp16 hi/lo planes in ONE launch (ops.split_f32_segments: every batch keeps its own absmax
    word, i.e. its planes are bit for bit what ops.split_f32 of that batch gives).  A step then launches nothing in front of
    its teacher forward: 3 launches (~18 us of mostly launch latency on the step's critical path) become views.
    Costs one gathered fp32 copy of the query set plus its planes (2 x 16 KiB per 32-token query)."""

    def __init__(self, Q: torch.Tensor, qmask: torch.Tensor, perm: torch.Tensor, batch: int, planes: bool = True,
                 teacher: Optional["TeacherScorer"] = None):
        self.batch = int(batch)
        self.Q = Q.index_select(0, perm)
        self.qmask = qmask.index_select(0, perm)
        self.n = int(self.Q.shape[0])
        self.planes = ops.split_f32_segments(self.Q, self.batch) if (planes and self.Q.is_cuda and self.n) else None
        # a teacher whose score cache is complete (every pseudo-query scored once: from the second epoch on): the epoch's rows
        # of it in batch order, one gather per epoch instead of one per step
        self.sc_t = None
        if teacher is not None and teacher.cache is not None and teacher.have is not None and bool(teacher.have.all()):
            self.sc_t = teacher.cache.index_select(0, perm.to(teacher.cache.device))

    def teacher_scores(self, i: int) -> Optional[torch.Tensor]:
        """Cached teacher scores (B, N) of batch i, or None (no complete cache: score the batch)."""
        return None if self.sc_t is None else self.sc_t[i * self.batch:(i + 1) * self.batch]

    def __len__(self) -> int:
        return (self.n + self.batch - 1) // self.batch

    def get(self, i: int):
        """(Qb, qmb, qplanes or None) of batch i: views."""
        lo = i * self.batch
        Qb, qmb = self.Q[lo:lo + self.batch], self.qmask[lo:lo + self.batch]
        qpl = ops.segment_planes(*self.planes, i, Qb.shape) if self.planes is not None else None
        return Qb, qmb, qpl


# ----------------------------------------------------------------------------------------------------
# Fused student update: no autograd graph, four launches per step on the student side
# ----------------------------------------------------------------------------------------------------
class FusedStudent:
    """Student page embeddings with their AdamW state, updated by ONE kernel per step: MaxSim backward gather ->
    l2-normalise(+mask) backward -> AdamW (evdr_maxsim_bwd_adamw).  Same arithmetic as
    `Psb = l2_normalize(Pbar * pmask); ...; loss.backward(); torch.optim.AdamW.step()` of the reference's step
    (mainv2_iter_distill_infonce.py:279-291, utils/utils.py:78-80: torch defaults betas (0.9, 0.999), eps 1e-8)."""

    def __init__(self, Pbar_init: torch.Tensor, pmask_student: torch.Tensor, lr: float, weight_decay: float,
                 betas=(0.9, 0.999), eps: float = 1e-8, l2_eps: float = 1e-12):
        self.pmask = pmask_student.bool().contiguous()
        self.x = (Pbar_init.detach().float() * self.pmask.unsqueeze(-1)).contiguous()
        self.exp_avg = torch.zeros_like(self.x)
        self.exp_avg_sq = torch.zeros_like(self.x)
        self.lr, self.weight_decay, self.betas, self.eps, self.l2_eps = lr, weight_decay, betas, eps, l2_eps
        self.steps = 0
        npg, ls, _ = self.x.shape
        self.tilemask, self.pageflags = ops.pack_pmask(self.pmask, npg, ls, self.x.device)     # the mask never changes
        # l2_normalize(Pbar * pmask) as the scorer's planes: written by the update kernel for the NEXT step's forward; valid
        # for the tensor object and version of x recorded in _planes_of (any torch in-place write to x invalidates them)
        self._planes = None
        self._planes_of = None
        self.direct_loss_store = True   # update(..., loss_to_host=True): the loss kernel writes the pinned host word itself (False: a copy launch, for the A/B)
        self._loss_host = None          # pinned scalar + event of update(..., loss_to_host=True)
        self._loss_event = None
        self._loss_ws: Dict[int, torch.Tensor] = {}     # batch size -> workspace of the one-launch loss, owned by this student

    def loss_workspace(self, b: int) -> torch.Tensor:
        """Per-row losses + ticket word of the one-launch InfoNCE kernel for batches of `b` queries, zeroed when first asked
        for (never inside a stream capture: GraphedStep asks before it captures).  Owned by this student and kept for its
        lifetime, so a captured graph's baked pointer stays valid; steps of one student are ordered by their updates of x, so
        the workspace is never used by two launches at once."""
        ws = self._loss_ws.get(int(b))
        if ws is None or ws.device != self.x.device:
            ws = self._loss_ws[int(b)] = ops.infonce_workspace(b, self.x.device)
        return ws

    def normalized(self) -> torch.Tensor:
        return ops.l2norm_forward(self.x, self.pmask, self.l2_eps)[0]

    def planes(self) -> Tuple[torch.Tensor, torch.Tensor]:
        """(planes, absmax word) of l2_normalize(Pbar * pmask) for the current x: left behind by the last update, or made
        here (first step, or x was written by something else since: a checkpoint restore, a manual edit)."""
        if self._planes is not None and self._planes_of is not None and self._planes_of[0] is self.x \
                and self._planes_of[1] == self.x._version:
            return self._planes
        keep = self._planes
        if keep is not None and (keep[0].shape[1:] != self.x.shape or keep[0].device != self.x.device):
            keep = None                                                  # x was replaced by a tensor of another shape / device
        self._planes = ops.l2norm_split(self.x, self.pmask, self.l2_eps, pageflags=self.pageflags, out=keep)  # a diverged page scores NaN
        self._planes_of = (self.x, self.x._version)
        return self._planes

    def scores(self, Qb, qmb, qplanes=None) -> Tuple[torch.Tensor, torch.Tensor]:
        """Student scores (B, n_pages) and the argmax the update needs.  l2_normalize(Pbar * pmask) lands directly in the
        scorer's fp16 hi/lo planes (no fp32 copy, no absmax/split pass).  `qplanes`: the batch's planes if already split."""
        pplanes, pamax = self.planes()
        qplanes, qamax = qplanes if qplanes is not None else ops.split_f32(Qb)
        return ops.maxsim_forward_prepared(qplanes, qamax, pplanes, pamax, qmb, self.tilemask, self.pageflags,
                                           want_argmax=True)

    def apply(self, dscore, Qb, qmb, arg, state: Optional[torch.Tensor] = None) -> None:
        """Backward gather -> normalise backward -> AdamW, one kernel, given d(loss)/d(scores) of THIS student's pages."""
        self.steps += 1
        if state is not None:
            ops.adamw_advance(state, self.betas)
        if self._planes is None:
            self.planes()
        ops.maxsim_backward_adamw(dscore, Qb, qmb, self.pmask, arg, self.x, self.exp_avg, self.exp_avg_sq, self.lr,
                                  self.betas, self.eps, self.weight_decay, self.steps, self.l2_eps, state=state,
                                  next_planes=self._planes, pageflags=self.pageflags)
        self._planes_of = (self.x, self.x._version)           # the kernel left the planes of the UPDATED x

    def update(self, Qb, qmb, sc_t, temp: float, state: Optional[torch.Tensor] = None, qplanes=None,
               loss_to_host: bool = False, scored=None) -> torch.Tensor:
        """One step given the teacher scores; returns the loss as a device scalar (no host sync).  With `state` (a
        device-side step counter, ops.adamw_state) nothing in the step depends on a host scalar: graph-capturable.
        `loss_to_host`: the loss is also copied to pinned host memory BEFORE the update kernel is launched
        (`wait_loss()` returns it as soon as that copy has landed, while the update still runs).  `scored` = (scores,
        argmax) of `self.scores(Qb, qmb)` when the caller has issued the student forward already (on another stream)."""
        sc_s, arg = scored if scored is not None else self.scores(Qb, qmb, qplanes)
        if loss_to_host and self._loss_host is None:
            self._loss_host = torch.empty((), dtype=torch.float32).pin_memory()
            self._loss_event = torch.cuda.Event()
        # loss_to_host: the loss kernel stores its scalar straight into the pinned host word (device-accessible memory): no copy
        # launch between the loss and the update (4.4 us of kernel + a launch gap per step), only the event
        direct = loss_to_host and self.direct_loss_store
        loss, dscore = ops.infonce_distill(sc_s, sc_t, temp, want_grad=True, ws=self.loss_workspace(sc_s.shape[0]),
                                           loss_out=self._loss_host if direct else None)
        if loss_to_host:
            if not direct:
                self._loss_host.copy_(loss, non_blocking=True)
            self._loss_event.record()
        self.apply(dscore, Qb, qmb, arg, state)
        return loss

    def wait_loss(self) -> float:
        """The loss of the last update(..., loss_to_host=True): waits for its copy only, not for the update kernel behind it."""
        self._loss_event.synchronize()
        return float(self._loss_host.item())

    def graphed(self, batch: int, lq: int, temp: float, teacher: Optional["TeacherScorer"] = None) -> "GraphedStep":
        """The whole step captured ONCE in a HIP graph (torch.cuda.CUDAGraph): ~15 launches replayed with one call.
        With `teacher`, its forward is part of the graph; without, the caller supplies the teacher scores per step
        (e.g. from a TeacherScorer cache)."""
        return GraphedStep(self, batch, lq, temp, teacher)


class GraphedStep:
    """HIP-graph replay of FusedStudent.update for a fixed batch shape.  Inputs are copied into static device buffers,
    the AdamW step counter lives on the device (evdr_adamw_advance), the loss comes back as a device scalar."""

    def __init__(self, student: FusedStudent, batch: int, lq: int, temp: float, teacher: Optional[TeacherScorer]):
        dev = student.x.device
        self.student, self.teacher = student, teacher
        n = student.x.shape[0]
        self.Qb = torch.zeros((batch, lq, ops.D), dtype=torch.float32, device=dev)
        self.Qb[..., 0] = 1.0                                        # unit-norm rows for the warm-up passes
        self.qmb = torch.ones((batch, lq), dtype=torch.bool, device=dev)
        self.sc_t = torch.zeros((batch, n), dtype=torch.float32, device=dev)
        self.state = ops.adamw_state(dev)
        student.loss_workspace(batch)                                # zeroed NOW, outside the capture below

        def body():
            qplanes = ops.split_f32(self.Qb)                         # once per step, shared by teacher and student
            sc_t = self.sc_t if teacher is None else teacher.corpus.score(self.Qb, self.qmb, qplanes=qplanes)
            return student.update(self.Qb, self.qmb, sc_t, temp, state=self.state, qplanes=qplanes)

        # warm-up on a side stream (lazy kernel attributes, allocator pools), then restore the parameters it touched
        keep = [t.clone() for t in (student.x, student.exp_avg, student.exp_avg_sq)]
        steps0 = student.steps
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(2):
                body()
        torch.cuda.current_stream(dev).wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.loss = body()
        for t, k in zip((student.x, student.exp_avg, student.exp_avg_sq), keep):
            t.copy_(k)
        # the captured step reads the planes its predecessor's update left behind: rebuild them for the restored x
        student._planes_of = None
        student.planes()
        student.steps = steps0
        self.state.zero_()
        if steps0:                                                   # resume: the device counter continues from steps0
            self.state[0] = steps0

    def __call__(self, Qb: torch.Tensor, qmb: torch.Tensor, sc_t: Optional[torch.Tensor] = None) -> torch.Tensor:
        self.Qb.copy_(Qb, non_blocking=True)
        self.qmb.copy_(qmb, non_blocking=True)
        if self.teacher is None:
            if sc_t is None:
                raise RuntimeError("this graph was captured without a teacher: pass the teacher scores")
            self.sc_t.copy_(sc_t, non_blocking=True)
        self.graph.replay()
        self.student.steps += 1
        return self.loss


def fused_train_one_step(Qb, qmb, teacher: "TeacherScorer", student: FusedStudent, temp: float,
                         qidx: Optional[torch.Tensor] = None, sync: bool = True, overlap: bool = False, qplanes=None,
                         sc_t: Optional[torch.Tensor] = None):
    """One fused update.  sync=True returns float(loss) like the reference's train_one_step (one host wait per step, for the
    loss only: the parameter update may still be running when it returns -- later work on the stream is ordered behind it);
    sync=False returns the loss as a device scalar and leaves the stream running, so that the host queues the next step
    while this one executes (read the loss when it is logged).
    overlap=True issues the student forward on a second stream beside the teacher forward (the two are independent; the
    loss waits for both).  Measured on MI355X (profiles/r03_experiments.txt): 0.4625 ms per step against 0.438 -- SLOWER.
    Both kernels fill the chip with one 256-register workgroup per CU that takes most of the LDS, so workgroups of the two
    launches cannot share a CU and only trade places, and the events cost host time; kept as an option for the A/B only.
    `qplanes`: the batch's (planes, absmax word) when they exist already (EpochBatches: split once per epoch).
    `sc_t`: the batch's teacher scores when they exist already (EpochBatches.teacher_scores: rows of a complete cache)."""
    device = student.x.device
    Qb = Qb.to(device, non_blocking=True).float()
    qmb = qmb.to(device, non_blocking=True)
    if qplanes is None:
        qplanes = ops.split_f32(Qb)                                  # once per step, shared by teacher and student
    scored = None
    if overlap:
        main = torch.cuda.current_stream(device)
        if getattr(student, "_side", None) is None:
            student._side = torch.cuda.Stream(device=device)
            student._ev_in, student._ev_out = torch.cuda.Event(), torch.cuda.Event()
        student._ev_in.record(main)
        with torch.cuda.stream(student._side):
            student._side.wait_event(student._ev_in)
            scored = student.scores(Qb, qmb, qplanes)
            student._ev_out.record(student._side)
        for t in (Qb, qmb, qplanes[0], qplanes[1]):                  # made on the main stream, read on the side stream
            t.record_stream(student._side)
    if sc_t is None:
        sc_t = teacher.scores(Qb, qmb, qidx, qplanes=qplanes)
    if overlap:
        main.wait_event(student._ev_out)
        for t in scored:                                             # made on the side stream, read on the main stream
            t.record_stream(main)
    loss = student.update(Qb, qmb, sc_t, temp, qplanes=qplanes, loss_to_host=sync, scored=scored)
    # sync: the loss left for the host before the update kernel was launched, so float(loss) is back while that kernel (a
    # seventh of the step) still runs and the caller's next launches queue up behind it instead of behind an idle GPU
    return student.wait_loss() if sync else loss


# ----------------------------------------------------------------------------------------------------
# Page-sharded training step (SURVEY §8(e) "Training partitioning"; no counterpart in the reference)
# ----------------------------------------------------------------------------------------------------
def gather_columns(block: torch.Tensor, sizes, group=None) -> torch.Tensor:
    """(B, n_local) score block of this rank -> full (B, N) rows on every rank: ONE all-gather of equal-sized messages
    (B * max(sizes) floats per rank; 8 KB at B = 32, N = 500, 8 ranks).  nccl = RCCL over xGMI with device tensors;
    gloo (CPU rehearsal of the exchange) goes through host memory."""
    import torch.distributed as dist
    b = block.shape[0]
    nmax = max(sizes)
    msg = torch.zeros((b, nmax), dtype=torch.float32, device=block.device)
    msg[:, : block.shape[1]] = block
    if dist.get_backend(group) == "gloo" and msg.is_cuda:
        buf = torch.empty((len(sizes) * b, nmax), dtype=torch.float32)
        dist.all_gather_into_tensor(buf, msg.cpu(), group=group)
        buf = buf.to(block.device)
    else:
        buf = torch.empty((len(sizes) * b, nmax), dtype=torch.float32, device=block.device)
        dist.all_gather_into_tensor(buf, msg, group=group)
    buf = buf.view(len(sizes), b, nmax)
    return torch.cat([buf[r, :, : sizes[r]] for r in range(len(sizes))], dim=1)


def gather_column_blocks(blocks, sizes, group=None):
    """`gather_columns` of SEVERAL (B, n_local) blocks of this rank (teacher and student scores of one step) in ONE all-gather:
    the blocks ride stacked in one (k, B, max(sizes)) message per rank.  -> a list of full (B, N) tensors, the same on every rank."""
    import torch.distributed as dist
    k, b, nmax = len(blocks), blocks[0].shape[0], max(sizes)
    dev = blocks[0].device
    msg = torch.zeros((k, b, nmax), dtype=torch.float32, device=dev)
    for i, blk in enumerate(blocks):
        msg[i, :, : blk.shape[1]] = blk
    if dist.get_backend(group) == "gloo" and msg.is_cuda:
        buf = torch.empty((len(sizes) * k, b, nmax), dtype=torch.float32)
        dist.all_gather_into_tensor(buf, msg.cpu(), group=group)
        buf = buf.to(dev)
    else:
        buf = torch.empty((len(sizes) * k, b, nmax), dtype=torch.float32, device=dev)
        dist.all_gather_into_tensor(buf, msg, group=group)
    buf = buf.view(len(sizes), k, b, nmax)
    return [torch.cat([buf[r, i, :, : sizes[r]] for r in range(len(sizes))], dim=1) for i in range(k)]


class _GatherColumns(torch.autograd.Function):
    """gather_columns with autograd: the backward keeps this rank's own columns of the upstream gradient -- the
    parameters are sharded, not replicated, so there is NO gradient all-reduce."""

    @staticmethod
    def forward(ctx, block, sizes, group):
        import torch.distributed as dist
        rank = dist.get_rank(group)
        ctx.lo = int(sum(sizes[:rank]))
        ctx.n_local = int(sizes[rank])
        return gather_columns(block, sizes, group)

    @staticmethod
    def backward(ctx, g):
        return g[:, ctx.lo: ctx.lo + ctx.n_local].contiguous(), None, None


def sharded_fused_train_one_step(Qb, qmb, teacher_shard: "TeacherScorer", student_shard: FusedStudent, temp: float,
                                 shard_sizes, group=None, qidx: Optional[torch.Tensor] = None, qplanes=None, sync: bool = True,
                                 sc_t_local: Optional[torch.Tensor] = None):
    """`sharded_train_one_step` on the fused kernels: every rank scores its page shard for teacher and student, the two
    (B, n_local) blocks ride in ONE all-gather (`gather_column_blocks`), the loss kernel runs redundantly on the full rows, and
    each rank's slice of d(loss)/d(scores) drives its own backward + AdamW kernel.  No autograd graph, no gradient all-reduce.
    `qplanes`: the batch's planes when they exist already (EpochBatches); `sc_t_local`: this rank's (B, n_local) teacher block when
    it exists already (EpochBatches.teacher_scores over the shard's complete score cache); sync=False returns the loss as a device
    scalar (the same number on every rank) instead of waiting for it."""
    import torch.distributed as dist
    device = student_shard.x.device
    Qb = Qb.to(device, non_blocking=True).float()
    qmb = qmb.to(device, non_blocking=True)
    rank = dist.get_rank(group)
    lo = int(sum(shard_sizes[:rank]))
    if qplanes is None:
        qplanes = ops.split_f32(Qb)
    sc_s_local, arg = student_shard.scores(Qb, qmb, qplanes)
    if sc_t_local is None:
        sc_t_local = teacher_shard.scores(Qb, qmb, qidx, qplanes=qplanes)
    sc_t, sc_s = gather_column_blocks([sc_t_local, sc_s_local], tuple(shard_sizes), group)
    loss, dscore = ops.infonce_distill(sc_s, sc_t, temp, want_grad=True, ws=student_shard.loss_workspace(sc_s.shape[0]))
    student_shard.apply(dscore[:, lo: lo + int(shard_sizes[rank])].contiguous(), Qb, qmb, arg)
    return float(loss.item()) if sync else loss


def sharded_train_one_step(Qb, qmb, teacher_shard, Pbar_shard, pmask_student_shard, opt, temp: float,
                           shard_sizes, group=None, qidx: Optional[torch.Tensor] = None) -> float:
    """One InfoNCE-distillation update with the pages sharded over the ranks of `group`.

    Every rank holds the SAME query batch, its own slice of the teacher pages (`teacher_shard`: a TeacherScorer over
    the shard) and of the student parameter (`Pbar_shard`, with its own AdamW state).  Per step: local teacher and
    student score blocks (B, n_local) -> ONE all-gather each (B*N*4 bytes in total: 64 KB at B=32, N=500) -> the
    softmax / teacher arg-max are computed redundantly on the full rows -> each rank back-propagates only into its
    own columns.  Identical arithmetic to the single-device step (the loss is the same number on every rank)."""
    device = Pbar_shard.device
    Qb = Qb.to(device, non_blocking=True)
    qmb = qmb.to(device, non_blocking=True)
    Psb = normalize_masked(Pbar_shard, pmask_student_shard)
    sc_t = _GatherColumns.apply(teacher_shard.scores(Qb, qmb, qidx), tuple(shard_sizes), group)
    sc_s = _GatherColumns.apply(score_multi_vector_masked(Qb, Psb, qmb, pmask_student_shard), tuple(shard_sizes), group)
    loss = infonce_distillation_loss(sc_s, sc_t, temperature=temp)
    opt.zero_grad(set_to_none=True)
    loss.backward()
    opt.step()
    return float(loss.item())


def main(argv=None):
    """`python -m torch.distributed.run --nproc-per-node N ... driver.py ...` shards the pages over N GPUs (one process per
    GPU, RCCL); without a launcher it is the single-process loop.  EVDR_DIST_BACKEND=gloo rehearses the exchange on one GPU."""
    import os
    import torch.distributed as dist
    args = build_argparser().parse_args(argv)
    started = False
    if int(os.environ.get("WORLD_SIZE", "1")) > 1 and not dist.is_initialized():
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        local = int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1)
        torch.cuda.set_device(local)
        backend = os.environ.get("EVDR_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
        if args.device == "auto":
            args.device = f"cuda:{local}"
        started = True
    try:
        run(args)
    finally:
        if started:
            dist.destroy_process_group()


if __name__ == "__main__":
    main()
