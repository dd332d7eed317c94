"""Drop-in counterpart of the reference's `evaluator/retrieval.py` for the MaxSim hot path.

Same public names, argument meaning, return types and error behaviour as the reference module, so that
`from evaluator.retrieval import score_multi_vector_masked, CustomRetrievalEvaluator`
(mainv2_iter_distill_infonce.py:20 and the 20 sibling scripts) keeps working when this module is put in its
place.  The arithmetic runs in hand-written gfx950 kernels behind the C ABI of
include/evdr.h; there is no torch/CPU fallback for the multi-vector scorers.

Reference lines mirrored:
  get_torch_device                      evaluator/retrieval.py:10-28
  BaseVisualRetrieverProcessor          :47-164  (score_single_vector :78-99, score_multi_vector :101-150)
  score_multi_vector_masked             :166-213
  CustomRetrievalEvaluator              :220-255
"""
from __future__ import annotations

import collections
import weakref
from abc import ABC, abstractmethod
from typing import Dict, List, Optional, Tuple, Union

import torch

from .. import ops
from . import metrics as _metrics

__all__ = ["get_torch_device", "left_padding", "BaseVisualRetrieverProcessor", "score_multi_vector_masked",
           "CustomRetrievalEvaluator", "forget_prepared", "enable_score_cache", "disable_score_cache", "score_cache_stats"]


def get_torch_device(device: str = "auto") -> str:
    """"auto" -> "cuda:0" when a GPU is visible (ROCm exposes MI355X as cuda), else "cpu"."""
    if device == "auto":
        return "cuda:0" if torch.cuda.is_available() else "cpu"
    return device


# Frozen fp32 pages (the teacher of a training run, the corpus of repeated evaluations) arrive here unchanged call after
# call (mainv2_iter_distill_infonce.py:282-283 re-scores P_teacher_norm every step).  Their device-side preparation --
# absmax + fp16 hi/lo split + mask packing, three passes over the pages -- is kept per tensor and reused while the tensor
# is alive and its autograd version counter has not moved (in-place torch ops bump it; so do this package's own in-place
# kernels).  Writes through raw pointers by other libraries and writes through `tensor.data` (which do not bump the
# counter) are not seen: call `forget_prepared()` after such a write.
_PREPARED: "collections.OrderedDict" = collections.OrderedDict()
_PREPARED_MAX = 4                      # tensors
_PREPARED_MAX_BYTES = 32 << 30         # of prepared planes in total (they are as large as the fp32 pages themselves)


def forget_prepared() -> None:
    _PREPARED.clear()
    _QPLANES.clear()
    _PACKED.clear()
    ops._DERIVED.clear()
    _SCORE_CACHES.clear()


# ---- opt-in score-row cache for FROZEN pages (the reference's per-step teacher re-scoring, mainv2_iter_distill_infonce.py:282-283) ----
# `enable_score_cache()` once -- in the re-export shim of INTEGRATION.md §1 -- and every `score_multi_vector_masked(Q, P_frozen, ...)`
# that needs no gradient keeps, per frozen page tensor, the score row of every query row it has seen (ops.ScoreRowCache: keyed on the
# row's bits + mask row + plane shift, verified by a full-row comparison, entirely on the device, no host synchronisation).  The
# unmodified reference scripts then pay the frozen teacher's forward once per query instead of once per step.  The caches hang on the
# SAME key as the prepared pages (tensor identity, layout, autograd version of P and pmask): an in-place write to P makes a new key --
# a miss --, a dead tensor frees its cache, inference tensors (no version counter) are never cached.  Memory: `max_bytes` bounds the
# sum over all live caches (least recently used caches are dropped first); a full cache stops storing, it never evicts rows.
_SCORE_CACHE_BUDGET = 0                # 0 = off (the default): bytes over all live caches
_SCORE_CACHES: "collections.OrderedDict" = collections.OrderedDict()      # prepared-pages key -> ops.ScoreRowCache | None (None: geometry refused)
_SCORE_CACHE_HASH_MASK = 0xFFFFFFFFFFFFFFFF                                # tests narrow it to force hash collisions


def enable_score_cache(max_bytes: int = 1 << 30) -> None:
    """Switch the frozen-page score-row cache on (see above); `max_bytes` = device memory all caches together may hold."""
    global _SCORE_CACHE_BUDGET
    if max_bytes <= 0:
        raise ValueError("enable_score_cache: max_bytes must be positive")
    _SCORE_CACHE_BUDGET = int(max_bytes)


def disable_score_cache() -> None:
    global _SCORE_CACHE_BUDGET
    _SCORE_CACHE_BUDGET = 0
    _SCORE_CACHES.clear()


def score_cache_stats(sync: bool = True) -> dict:
    """{"caches", "bytes", "entries", "capacity"} over the live caches (`entries` reads device counters: one synchronising copy)."""
    live = [c for c in _SCORE_CACHES.values() if c is not None]
    return {"caches": len(live), "bytes": sum(c.nbytes for c in live), "capacity": sum(c.capacity for c in live),
            "entries": sum(int(c.n_entries.item()) for c in live) if sync else None, "budget": _SCORE_CACHE_BUDGET}


def _score_cache_for(key, P: torch.Tensor, Q: torch.Tensor):
    """The cache of the prepared-pages entry `key` for batches shaped like Q, or None (off, not cacheable, geometry refused)."""
    if _SCORE_CACHE_BUDGET <= 0 or key not in _PREPARED:
        return None
    lq = int(Q.shape[1])
    if not (2 <= lq <= 32):                              # single-token packs and 32-token slices of long queries take the plain path
        return None
    hit = _SCORE_CACHES.get(key)
    if hit is not None or key in _SCORE_CACHES:
        if hit is not None and not hit.accepts(Q, int(P.shape[0])):
            return None                                  # another query geometry than the one this cache was made for: scored plainly
        if hit is not None:
            _SCORE_CACHES.move_to_end(key)
        return hit
    for k in [k for k in _SCORE_CACHES if k not in _PREPARED]:          # caches of tensors that died / changed
        del _SCORE_CACHES[k]
    # the budget is shared: a new frozen tensor takes what the live caches leave, dropping least recently used caches if that is nothing
    used = lambda: sum(c.nbytes for c in _SCORE_CACHES.values() if c is not None)
    while _SCORE_CACHES and _SCORE_CACHE_BUDGET - used() < (_SCORE_CACHE_BUDGET >> 2):
        _SCORE_CACHES.popitem(last=False)
    try:
        cache = ops.ScoreRowCache(lq, Q.dtype, int(P.shape[0]), _SCORE_CACHE_BUDGET - used(), Q.device, _SCORE_CACHE_HASH_MASK)
    except ValueError:
        cache = None                                     # the budget holds not even one row of this geometry
    _SCORE_CACHES[key] = cache
    return cache


_tensor_key = ops.tensor_key           # None for None AND for inference tensors (no version counter: never cached)


_LAST_KEY = None


def _prepared_key(P: torch.Tensor, pmask: Optional[torch.Tensor]):
    return (_tensor_key(P), _tensor_key(pmask))


def _prepared_pages(P: torch.Tensor, pmask: Optional[torch.Tensor]):
    """(planes, amax, tilemask, pageflags) of frozen fp32-like pages, from the cache when P and pmask are unchanged.
    P is the CALLER's tensor (the cache is keyed on it and dies with it); embeddings narrower than 128 are padded here, on
    a miss only -- keyed on a padded temporary, every call would redo pad + split + mask packing + the non-finite scan and push a
    live entry out of the small LRU."""
    global _LAST_KEY
    key = _LAST_KEY = _prepared_key(P, pmask)           # (the score-row cache of the same call is keyed on it too: computed once)
    cacheable = key[0] is not None and (pmask is None or key[1] is not None)       # not under torch.inference_mode()
    hit = _PREPARED.get(key) if cacheable else None
    if hit is not None and hit[0]() is not None and (pmask is None or hit[1]() is not None):
        _PREPARED.move_to_end(key)
        return hit[2]
    Pw = ops.pad_width(P)
    wide = Pw.shape[-1] == ops.D_WIDE                  # 129..256 columns: four planes (fp16 hi/lo x two column blocks), any input dtype
    if wide:
        planes, amax = ops.split_wide(Pw)
    elif Pw.dtype == torch.bfloat16:                   # scored as they are: only the packed masks and the non-finite scan are kept
        planes, amax = Pw.contiguous()[None], None
    else:
        # pages that came out of this package's l2_normalize bring their planes along (ops.planes_of); others are split here
        made = ops.planes_of(P) if (Pw is P and P.dtype == torch.float32) else None
        planes, amax = made if made is not None else ops.split_f32(Pw)
    tilemask, pageflags = ops.pack_pmask(pmask, P.shape[0], P.shape[1], P.device)
    ops.flag_nonfinite(planes[0], pmask, pageflags)
    if wide:
        ops.flag_nonfinite(planes[1], pmask, pageflags)               # the hi plane of the second column block
    prep = (planes, amax, tilemask, pageflags)
    nbytes = 0 if (Pw is P and P.dtype == torch.bfloat16 and P.is_contiguous()) else planes.numel() * planes.element_size()
    if cacheable and nbytes <= _PREPARED_MAX_BYTES:
        drop = lambda _ref, key=key: (_PREPARED.pop(key, None), _SCORE_CACHES.pop(key, None))   # the tensor died: free its planes (and score rows) right away
        _PREPARED[key] = (weakref.ref(P, drop), weakref.ref(pmask, drop) if pmask is not None else None, prep, nbytes)
        while len(_PREPARED) > _PREPARED_MAX or sum(e[3] for e in _PREPARED.values()) > _PREPARED_MAX_BYTES:
            _PREPARED.popitem(last=False)
    return prep


# The student's page mask is the same tensor step after step (mainv2_iter_distill_infonce.py:279,286 pass pmask_s every step): its
# packed form (tile words + page flag words) is kept per mask tensor, keyed like everything else here (address, layout, version), and
# every call gets its own copy of the small flag array, because the non-finite scan of the step's fresh planes ORs its findings into it.
_PACKED: list = []                     # [(weakref(pmask), key, tilemask, pageflags)]: the last page mask that was packed


def _packed_mask(pmask, npages: int, lp: int, dev):
    key = _tensor_key(pmask)
    if key is None:
        return ops.pack_pmask(pmask, npages, lp, dev)
    if _PACKED and _PACKED[0][1] == key and _PACKED[0][0]() is not None and _PACKED[0][2].device == dev:
        return _PACKED[0][2], _PACKED[0][3].clone()
    tilemask, pageflags = ops.pack_pmask(pmask, npages, lp, dev)
    _PACKED[:] = [(weakref.ref(pmask, lambda _r: _PACKED.clear() if (_PACKED and _PACKED[0][1] == key) else None), key, tilemask, pageflags)]
    return tilemask, pageflags.clone()


_QPLANES: list = []                    # [(weakref(Q), key, (planes, absmax word))]: the last fp32 query batch that was split


def _query_planes(Q: torch.Tensor):
    """fp16 hi/lo planes of an fp32 query batch.  The reference's step scores ONE batch twice in a row -- against the teacher
    and against the student (mainv2_iter_distill_infonce.py:283,286) -- so the last batch's planes are kept while that tensor is
    alive and unwritten (key as for the prepared pages)."""
    split = ops.split_wide if Q.shape[-1] == ops.D_WIDE else ops.split_f32
    key = _tensor_key(Q)
    if key is None:                                     # inference tensor: split per call
        return split(Q)
    if _QPLANES and _QPLANES[0][1] == key and _QPLANES[0][0]() is not None:
        return _QPLANES[0][2]
    made = split(Q)
    _QPLANES[:] = [(weakref.ref(Q, lambda _r: _QPLANES.clear() if (_QPLANES and _QPLANES[0][1] == key) else None), key, made)]
    return made


# ------------------------------------------------------------------------------------------------
# A1 + A6: masked MaxSim, differentiable w.r.t. the page embeddings
# ------------------------------------------------------------------------------------------------
def _maxsim_forward(Q, P, qmask, pmask, need_dq: bool, need_dp: bool):
    """The forward of A1 on whatever operands the call brings (frozen pages: prepared once per tensor and, with the score-row cache
    on, scored once per query row; trainable pages that came out of this package's l2_normalize: their planes ride along).
    -> (out, argmax or None, Q and P as the kernels took them: what a backward needs)."""
    P_caller = P                                      # what the prepared-pages cache is keyed on
    if (Q.dim() == 3 and P.dim() == 3 and Q.shape[-1] == P.shape[-1] and 0 < Q.shape[-1] <= ops.D_WIDE
            and Q.shape[-1] not in (ops.D, ops.D_WIDE)):
        Q, P = ops.pad_width(Q), ops.pad_width(P)     # widths between the kernels' 128 / 256 ride on zero columns (exact); gradients are cut back
    wide = Q.dim() == 3 and Q.shape[-1] == ops.D_WIDE   # 129..256 columns: fp16 hi/lo planes x two column blocks, whatever the dtype
    both_bf16 = P.dtype == torch.bfloat16 and Q.dtype == torch.bfloat16 and not wide
    frozen = (not need_dp and P.is_cuda and P.dim() == 3 and Q.dim() == 3 and P.shape[-1] == Q.shape[-1] and Q.shape[-1] in (ops.D, ops.D_WIDE)
              and (both_bf16 or wide or P.dtype != torch.bfloat16) and not (both_bf16 and need_dq)
              and P.shape[0] > 0 and P.shape[1] > 0
              and Q.shape[0] > 0 and Q.shape[1] > 0 and P.shape[1] <= 65535 and Q.shape[1] <= 65535)
    if frozen:
        # frozen pages: mask packing, the non-finite scan and (fp32) the plane split are done once per tensor
        planes, amax, tilemask, pageflags = _prepared_pages(P_caller, pmask)
        qplanes, qamax = (Q.contiguous()[None], None) if both_bf16 else _query_planes(Q)
        cache = None
        if _SCORE_CACHE_BUDGET > 0 and not need_dq and not wide and Q.is_contiguous() and Q.dtype in (torch.float32, torch.bfloat16):
            cache = _score_cache_for(_LAST_KEY, P, Q)        # the key _prepared_pages has just computed for (P_caller, pmask)
        if cache is not None:
            out, arg = ops.maxsim_forward_cached(cache, Q, qplanes, qamax, planes, amax, qmask, tilemask, pageflags), None
        else:
            out, arg = ops.maxsim_forward_prepared(qplanes, qamax, planes, amax, qmask, tilemask, pageflags,
                                                   want_argmax=need_dq)
    else:
        # trainable pages that came out of this package's l2_normalize (the reference's Psb, :279) bring their planes along
        derived = None
        if (P_caller is P and P.is_cuda and P.dtype == torch.float32 and Q.dtype == torch.float32 and P.dim() == 3 and Q.dim() == 3
                and P.shape[-1] == ops.D and Q.shape[-1] == ops.D and P.shape[0] > 0 and 0 < P.shape[1] <= 65535
                and Q.shape[0] > 0 and 0 < Q.shape[1] <= 65535):
            derived = ops.planes_of(P)
        if derived is not None:
            planes, amax = derived
            tilemask, pageflags = _packed_mask(pmask, P.shape[0], P.shape[1], P.device)
            ops.flag_nonfinite(planes[0], pmask, pageflags)
            qplanes, qamax = _query_planes(Q)
            out, arg = ops.maxsim_forward_prepared(qplanes, qamax, planes, amax, qmask, tilemask, pageflags, want_argmax=True)
        else:
            out, arg = ops.maxsim_forward(Q, P, qmask, pmask, want_argmax=need_dp or need_dq)
    return out, arg, Q, P


class _MaxSimMasked(torch.autograd.Function):
    @staticmethod
    def forward(ctx, Q, P, qmask, pmask):
        need_dq, need_dp = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        ctx.width = int(Q.shape[-1]) if Q.dim() == 3 else ops.D
        out, arg, Q, P = _maxsim_forward(Q, P, qmask, pmask, need_dq, need_dp)
        if need_dp or need_dq:
            ctx.save_for_backward(Q.detach(), P.detach() if need_dq else None, qmask, pmask, arg)
            ctx.p_shape, ctx.p_dtype, ctx.q_dtype = tuple(P.shape), P.dtype, Q.dtype
        return out

    @staticmethod
    def backward(ctx, g):
        Q, P, qmask, pmask, arg = ctx.saved_tensors
        npg, lp, _ = ctx.p_shape
        dQ = dP = None
        if ctx.needs_input_grad[1]:
            dP = ops.maxsim_backward(g, Q, qmask, pmask, arg, npg, lp)[..., :ctx.width].to(ctx.p_dtype)
        if ctx.needs_input_grad[0]:
            dQ = ops.maxsim_backward_q(g, P, qmask, pmask, arg, Q.shape[0], Q.shape[1])[..., :ctx.width].to(ctx.q_dtype)
        return dQ, dP, None, None


def score_multi_vector_masked(
    Q: torch.Tensor,        # (Nq, Lq, D)
    P: torch.Tensor,        # (Np, Lp, D)
    qmask: torch.Tensor,    # (Nq, Lq) bool-like
    pmask: torch.Tensor,    # (Np, Lp) bool-like
    chunk_p: int = 128,
) -> torch.Tensor:
    """out[q,p] = sum_n qmask[q,n] * any(pmask[p]) * max_m(Q[q,n]·P[p,m] if pmask[p,m] else -1e4), fp32,
    on the inputs' device, autograd-capable w.r.t. P and Q.  `chunk_p` only bounded the reference's 4-D
    intermediate (evaluator/retrieval.py:187); the fused kernel has none, so it is accepted and ignored."""
    del chunk_p
    if not (torch.is_grad_enabled() and (Q.requires_grad or P.requires_grad)):
        # nothing to differentiate (the reference's teacher call under torch.no_grad(), :282-283; evaluation): the same forward without
        # the autograd.Function round trip (~10 us of host time per call, and the step is host-bound once the teacher's rows are cached)
        return _maxsim_forward(Q, P, qmask, pmask, False, False)[0]
    return _MaxSimMasked.apply(Q, P, qmask, pmask)


# ------------------------------------------------------------------------------------------------
# A2 / A3: the ColPali-style list scorers
# ------------------------------------------------------------------------------------------------
def _left_pad_stack(seqs: List[torch.Tensor], device) -> torch.Tensor:
    """Zero LEFT padding to the batch's own max length (evaluator/retrieval.py:30-45), built in one tensor instead of
    per-sequence torch.cat; host sequences are padded on the host and cross PCIe as ONE copy."""
    seqs = [s.unsqueeze(0) if s.ndim == 1 else s for s in seqs]
    lmax = max(int(s.shape[0]) for s in seqs)
    d = int(seqs[0].shape[-1])
    on_host = all(not s.is_cuda for s in seqs)
    out = torch.zeros((len(seqs), lmax, d), dtype=seqs[0].dtype, device="cpu" if on_host else device)
    for i, s in enumerate(seqs):
        if s.shape[0]:
            out[i, lmax - s.shape[0]:] = s if on_host else s.to(device)
    return out.to(device) if on_host else out


def left_padding(sequences, batch_first=True, padding_value=0):
    """Pad ragged (Li, D) sequences on the LEFT with `padding_value` to the longest one and stack them on the GPU:
    (B, Lmax, D), or (Lmax, B, D) with batch_first=False; 1-D inputs count as one-token sequences
    (evaluator/retrieval.py:30-45 -- the reference hard-codes device='cuda' there, so this helper needs a GPU too)."""
    if not torch.cuda.is_available():
        raise RuntimeError("left_padding places its result on the GPU like the reference (device='cuda'); no GPU is visible")
    seqs = [s.unsqueeze(0) if s.ndim == 1 else s for s in sequences]
    lmax = max(int(s.size(0)) for s in seqs)
    d = int(seqs[0].size(-1))
    out = torch.full((len(seqs), lmax, d), padding_value, dtype=seqs[0].dtype, device="cuda")
    for i, s in enumerate(seqs):
        if s.size(0):
            out[i, lmax - s.size(0):] = s.to(device="cuda")
    return out if batch_first else out.transpose(0, 1)


class BaseVisualRetrieverProcessor(ABC):
    """Base class for visual retriever processors (same abstract surface as the reference)."""

    @abstractmethod
    def process_images(self, images):
        pass

    @abstractmethod
    def process_queries(self, queries: List[str], max_length: int = 50, suffix: Optional[str] = None):
        pass

    @abstractmethod
    def score(self, qs: List[torch.Tensor], ps: List[torch.Tensor],
              device: Optional[Union[str, torch.device]] = None, **kwargs) -> torch.Tensor:
        pass

    @staticmethod
    def score_single_vector(qs: List[torch.Tensor], ps: List[torch.Tensor],
                            device: Optional[Union[str, torch.device]] = None) -> torch.Tensor:
        """Dense dot product of pooled vectors -> (Nq, Np) fp32 on `device` (plain torch
        plumbing, 64 MFLOP at 500x500; runs on CPU or GPU exactly like the reference)."""
        device = device or get_torch_device("auto")
        if len(qs) == 0:
            raise ValueError("No queries provided")
        if len(ps) == 0:
            raise ValueError("No passages provided")
        qv = torch.stack(list(qs)).to(device)
        pv = torch.stack(list(ps)).to(device)
        scores = qv @ pv.transpose(0, 1)
        assert scores.shape[0] == len(qs), f"Expected {len(qs)} scores, got {scores.shape[0]}"
        return scores.to(torch.float32)

    @staticmethod
    def score_multi_vector(qs: Union[torch.Tensor, List[torch.Tensor]], ps: Union[torch.Tensor, List[torch.Tensor]],
                           batch_size: int = 128, device: Optional[Union[str, torch.device]] = None) -> torch.Tensor:
        """Unmasked late-interaction scores of ragged lists -> (Nq, Np) fp32 on the CPU.

        Reproduces the reference's batch-composition quirk: each (query batch, page batch) block is
        zero-left-padded to its own max lengths and the zero page rows TAKE PART in the max
        (evaluator/retrieval.py:122-136).  With the fused kernel this is simply "no page mask"."""
        device = device or get_torch_device("auto")
        if len(qs) == 0:
            raise ValueError("No queries provided")
        if len(ps) == 0:
            raise ValueError("No passages provided")
        if torch.device(device).type != "cuda":
            raise RuntimeError("score_multi_vector runs on the GPU (HIP kernels, no CPU fallback); "
                               f"got device={device!r}")
        rows = []
        p_batches = [_left_pad_stack(list(ps[j:j + batch_size]), device) for j in range(0, len(ps), batch_size)]
        for i in range(0, len(qs), batch_size):
            qb = _left_pad_stack(list(qs[i:i + batch_size]), device)
            blocks = []
            for pb in p_batches:
                s, _ = ops.maxsim_forward(qb, pb, None, None, want_argmax=False)
                blocks.append(s)
            rows.append(torch.cat(blocks, dim=1).cpu())
        scores = torch.cat(rows, dim=0)
        assert scores.shape[0] == len(qs), f"Expected {len(qs)} scores, got {scores.shape[0]}"
        return scores.to(torch.float32)

    @abstractmethod
    def get_n_patches(self, image_size: Tuple[int, int], patch_size: int = 14, *args, **kwargs) -> Tuple[int, int]:
        pass


# ------------------------------------------------------------------------------------------------
# A9: metric wrapper
# ------------------------------------------------------------------------------------------------
class CustomRetrievalEvaluator:
    def __init__(self, k_values: List[int] = [1, 3, 5, 10, 50, 70, 100], score_function: str = "cos_sim"):
        self.k_values = list(k_values)
        self.score_function = score_function

    def compute_mteb_metrics(self, relevant_docs: Dict[str, Dict[str, int]],
                             results: Dict[str, Dict[str, float]], **kwargs) -> Dict[str, Dict[str, float]]:
        """-> {"NDCG": {"NDCG@k"}, "mAP": {"MAP@k"}, "Recall": {"Recall@k"}, "Precision": {"P@k"},
        "mRR": {"MRR@k"}} like the reference's wrapper over mteb (evaluator/retrieval.py:230-255).
        `ignore_identical_ids=True` drops result entries whose docid equals the query id, as mteb does."""
        if kwargs.get("ignore_identical_ids", False):
            results = {q: {d: s for d, s in ds.items() if d != q} for q, ds in results.items()}
        return _metrics.evaluate(relevant_docs, results, self.k_values)
