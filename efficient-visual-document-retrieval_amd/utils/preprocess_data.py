"""Counterpart of the reference's utils/preprocess_data.py: the npz feature-dump schema <-> the tensors the MaxSim
path consumes.  Same function names, arguments and return values; written fresh.

npz schema (preprocess/split_data.py:29-36, utils/utils.py:83-103 of the reference): `documents` / `query` are
object arrays of (Li, D) float arrays; `doc_attnmask`, `doc_imgmask`, `query_attnmask` object arrays of (Li,)
bool-likes (optional); `docid`, `qid`; `relevant_docs` a 0-d object dict {qid: {docid: rel}}; `docidx_2_docid`
{str(i): docid}; `qsidx_2_query`.  pmask = valid & attn & img (preprocess_data.py:101).

`corpus_from_payload` is the MI355X-side addition: object arrays -> a resident PageCorpus (bf16 planes + packed
masks) without first materialising the zero-padded (N, Lmax, D) fp32 tensor on the device.
"""
from typing import Dict, Optional, Tuple

import numpy as np
import torch


def load_npz(path: str):
    return np.load(path, allow_pickle=True)


class _L2NormMasked(torch.autograd.Function):
    """y = m * x / (||m * x|| + eps) on the HIP kernels (evdr_l2norm_fwd / _bwd); m = optional per-row mask."""
    last_planes = None          # (address of y, planes, absmax word) of the forward that just ran: handed to _normalized()

    @staticmethod
    def forward(ctx, x, rowmask, eps, want_planes=False):
        from .. import ops
        if want_planes and x.dim() == 3 and x.numel() <= (1 << 30):
            # page-shaped TRAINABLE input (N, L, 128) -- the reference's Psb = l2_normalize(Pbar_param * pmask), scored by the very
            # next call of its step (mainv2_iter_distill_infonce.py:279,286): the launch also leaves y as the scorer's fp16 hi/lo
            # planes and _normalized() below hangs them on the returned tensor, so that score_multi_vector_masked(Q, y, ...) runs no
            # absmax + split passes over y.  Only there: a tensor normalised without a graph (the teacher at load time,
            # preprocess_queries, evaluation under no_grad) may never be scored, and its planes would hold a second copy of it in
            # HBM for as long as it lives (up to 4 GiB); such pages are split by the scorer when -- and if -- they are scored, and
            # frozen ones are cached there (evaluator/retrieval._prepared_pages).
            y, norm, planes, amax = ops.l2norm_forward(x, rowmask, eps, want_planes=True)
            _L2NormMasked.last_planes = (y.data_ptr(), planes, amax)
        else:
            y, norm = ops.l2norm_forward(x, rowmask, eps)
        if ctx.needs_input_grad[0]:
            ctx.save_for_backward(x, rowmask, norm)
            ctx.eps = eps
        return y

    @staticmethod
    def backward(ctx, gy):
        from .. import ops
        x, rowmask, norm = ctx.saved_tensors
        return ops.l2norm_backward(gy, x, rowmask, norm, ctx.eps), None, None, None


def _kernel_ok(x: torch.Tensor) -> bool:
    return x.is_cuda and x.dtype == torch.float32 and x.dim() >= 2 and x.shape[-1] == 128 and x.numel() > 0


def _normalized(x: torch.Tensor, rowmask, eps: float) -> torch.Tensor:
    from .. import ops
    _L2NormMasked.last_planes = None
    # (decided HERE: inside Function.forward the grad mode is always off)
    trainable = x.requires_grad and torch.is_grad_enabled() and not torch.is_inference_mode_enabled()
    y = _L2NormMasked.apply(x, rowmask, float(eps), trainable)
    made = _L2NormMasked.last_planes
    _L2NormMasked.last_planes = None
    if made is not None and made[0] == y.data_ptr():
        ops.remember_planes(y, made[1], made[2])          # alive as long as y is, dropped by any in-place write to y
    return y


def l2_normalize(x: torch.Tensor, eps: float = 1e-12) -> torch.Tensor:
    """x / (||x||_2 + eps) over the last dim -- eps is ADDED to the norm, zero rows stay exactly zero and get
    the subgradient 0 through the norm (utils/preprocess_data.py:8-9; applied to Pbar*pmask every step,
    mainv2_iter_distill_infonce.py:279).  fp32 CUDA tensors of width 128 (the page / query embeddings of the
    training step) run on one fused HIP kernel each way; anything else (host-side preprocessing on the CPU, other
    widths) is the same formula in torch."""
    if _kernel_ok(x):
        return _normalized(x, None, eps)
    return x / (torch.linalg.vector_norm(x, ord=2, dim=-1, keepdim=True) + eps)


def normalize_masked(x: torch.Tensor, rowmask: torch.Tensor, eps: float = 1e-12) -> torch.Tensor:
    """l2_normalize(x * rowmask[..., None]) in ONE kernel each way (the `Pbar_param * pmask` multiply of
    mainv2_iter_distill_infonce.py:279 fused into the normalisation); same values and gradients."""
    if _kernel_ok(x):
        return _normalized(x, rowmask, eps)
    return l2_normalize(x * rowmask.unsqueeze(-1).to(x.dtype), eps)


def parse_relevant_docs(z) -> Dict[str, dict]:
    v = z["relevant_docs"]
    return v if isinstance(v, dict) else v.item()


def _as_object_array(x):
    if isinstance(x, np.ndarray):
        return x.astype(object)
    out = np.empty(len(x), dtype=object)
    for i, v in enumerate(x):
        out[i] = v
    return out


def _to_bool_1d(arr) -> Optional[np.ndarray]:
    if arr is None:
        return None
    a = np.asarray(arr)
    if a.dtype == object:
        a = np.asarray(a.tolist())
    a = a.astype(bool)
    if a.ndim == 2 and a.shape[-1] == 1:
        a = a[:, 0]
    return a


def pad_tokens_object(tok_list, pad_to: Optional[int] = None) -> Tuple[np.ndarray, np.ndarray]:
    """object array (N,) of (Li, D) -> zero-padded (N, Lmax, D) float32 and the (N, Lmax) validity mask.  `pad_to`: pad to
    at least this length (a rank that loads only its page shard pads to the whole dump's longest page, so that all shards
    have one shape)."""
    toks = _as_object_array(tok_list)
    lens = np.fromiter((int(t.shape[0]) for t in toks), dtype=np.int64, count=len(toks))
    d = int(toks[0].shape[1])
    lmax = max(int(lens.max()), int(pad_to or 0))
    pad = np.zeros((len(toks), lmax, d), dtype=np.float32)
    for i, t in enumerate(toks):
        pad[i, : lens[i]] = t
    valid = np.arange(lmax)[None, :] < lens[:, None]
    return pad, valid


def pad_mask_object(mask_list, L: int, N: int, valid: np.ndarray) -> np.ndarray:
    """object array (N,) of (Li,) bool-likes (or None = 'all valid positions') -> (N, L) bool, padded False."""
    if mask_list is None:
        return valid.copy()
    masks = _as_object_array(mask_list)
    out = np.zeros((N, L), dtype=bool)
    for i in range(N):
        m = _to_bool_1d(masks[i])
        if m is None:
            out[i] = valid[i]
        else:
            n = min(L, m.shape[0])
            out[i, :n] = m[:n]
    return out


def preprocess_docs(documents_obj, doc_attnmask_obj, doc_imgmask_obj, device, pad_to: Optional[int] = None):
    """-> (P_raw (N,L,D) fp32 NOT normalised, pmask (N,L) bool = valid & attn & img, valid (N,L) numpy bool).  `pad_to`: see
    `pad_tokens_object` (not in the reference's signature; only the page-sharded driver passes it)."""
    pad, valid = pad_tokens_object(documents_obj, pad_to)
    n, l, _ = pad.shape
    pm = valid & pad_mask_object(doc_attnmask_obj, l, n, valid) & pad_mask_object(doc_imgmask_obj, l, n, valid)
    return (torch.from_numpy(pad).to(device=device, dtype=torch.float32),
            torch.from_numpy(pm).to(device=device, dtype=torch.bool), valid)


def preprocess_queries(query_obj, query_attnmask_obj, device):
    """-> (Q (Nq,Lq,D) fp32 L2-normalised, qmask (Nq,Lq) bool = valid & attn)."""
    pad, valid = pad_tokens_object(query_obj)
    n, l, _ = pad.shape
    qm = valid & pad_mask_object(query_attnmask_obj, l, n, valid)
    q = l2_normalize(torch.from_numpy(pad).to(device=device, dtype=torch.float32))
    return q, torch.from_numpy(qm).to(device=device)


def _get(z, key, item=False):
    if key not in z.files:
        return None
    return z[key].item() if item else z[key]


def load_payload(npz_path: str):
    """All keys of a feature dump as a dict (missing optional keys -> None)."""
    z = load_npz(npz_path)
    return {
        "docid": z["docid"],
        "documents": _get(z, "documents"), "doc_attnmask": _get(z, "doc_attnmask"), "doc_imgmask": _get(z, "doc_imgmask"),
        "query": _get(z, "query"), "qid": _get(z, "qid"), "query_attnmask": _get(z, "query_attnmask"),
        "relevant_docs": _get(z, "relevant_docs", item=True), "docidx_2_docid": _get(z, "docidx_2_docid", item=True),
        "qsidx_2_query": _get(z, "qsidx_2_query"),
    }


def load_train_payload(train_npz: str):
    z = load_npz(train_npz)
    return {
        "docid": z["docid"], "documents": z["documents"],
        "doc_attnmask": _get(z, "doc_attnmask"), "doc_imgmask": _get(z, "doc_imgmask"),
        "query": z["query"], "query_attnmask": _get(z, "query_attnmask"),
        "relevant_docs": _get(z, "relevant_docs", item=True), "docidx_2_docid": _get(z, "docidx_2_docid", item=True),
        "qsidx_2_query": _get(z, "qsidx_2_query"),
    }


def load_test_payload(test_npz: str):
    """Same keys as `load_train_payload` (the reference keeps two identical loaders, utils/preprocess_data.py:143-164);
    the parameter carries the reference's name so that keyword calls keep working."""
    return load_train_payload(test_npz)


def load_init_payload(init_npz: str):
    z = load_npz(init_npz)
    return {"docid": _get(z, "docid"), "documents": z["documents"],
            "doc_attnmask": _get(z, "doc_attnmask"), "doc_imgmask": _get(z, "doc_imgmask")}


def load_query_payload(npz_path: str):
    z = load_npz(npz_path)
    return {"query": z["query"], "qid": z["qid"], "query_attnmask": _get(z, "query_attnmask"),
            "qsidx_2_query": _get(z, "qsidx_2_query"), "relevant_docs": _get(z, "relevant_docs", item=True)}


def corpus_from_payload(documents_obj, doc_attnmask_obj, doc_imgmask_obj, device, dtype=torch.bfloat16,
                        normalize: bool = True, idx_base: int = 0):
    """Object arrays of a feature dump -> resident PageCorpus, page by page: each page is masked, L2-normalised
    (what the scripts do before scoring, mainv2_iter_distill_infonce.py:94) and stored in the corpus dtype; the
    only full-size device buffer is the final (N, Lmax, 128) one in `dtype` (bf16: half of the reference's fp32
    padded tensor; torch.float32 keeps fp32 accuracy through the fp16 hi/lo split)."""
    from ..corpus import PageCorpus
    docs = _as_object_array(documents_obj)
    n = len(docs)
    lens = np.fromiter((int(t.shape[0]) for t in docs), dtype=np.int64, count=n)
    lmax, d = int(lens.max()), int(docs[0].shape[1])
    valid = np.arange(lmax)[None, :] < lens[:, None]
    pm = valid & pad_mask_object(doc_attnmask_obj, lmax, n, valid) & pad_mask_object(doc_imgmask_obj, lmax, n, valid)
    P = torch.zeros((n, lmax, d), dtype=dtype, device=device)
    for i in range(n):
        x = torch.from_numpy(np.ascontiguousarray(docs[i], dtype=np.float32)).to(device)
        x = x * torch.from_numpy(pm[i, : lens[i]]).to(device).unsqueeze(-1)
        if normalize:
            x = l2_normalize(x)
        P[i, : lens[i]] = x.to(dtype)
    pmask = torch.from_numpy(pm).to(device)
    return PageCorpus.from_tensor(P, pmask, idx_base=idx_base), pmask
