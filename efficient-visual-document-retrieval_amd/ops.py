"""Thin torch-tensor wrappers over the C ABI (include/evdr.h).  PyTorch is plumbing here: it owns device
memory and the stream; every computation below runs in libevdr.so's HIP kernels on the caller's current
stream, with no synchronisation.  CPU tensors are rejected (there is no CPU path in the product)."""
from __future__ import annotations

from typing import Optional, Tuple

import collections
import weakref

import torch

from . import _lib as L

D = 128                  # width of one column block (the ColPali / ColQwen projection width: every tuned instance)
D_WIDE = 256             # two column blocks: the widest embedding the kernels score (fp16 hi/lo planes x 2 blocks, round 6)


def _require_cuda(*tensors) -> torch.device:
    dev = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError(
                "evdr_amd scores on the GPU only (HIP kernels, no CPU fallback): got a tensor on "
                f"{t.device}; move inputs to a cuda device")
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise RuntimeError(f"tensors on different devices: {dev} vs {t.device}")
    if dev is None:
        raise RuntimeError("no tensors given")
    return dev


def _mask_u8(m: Optional[torch.Tensor], shape, dev) -> Optional[torch.Tensor]:
    """mask.bool() as the reference does (evaluator/retrieval.py:179-180); one byte per token."""
    if m is None:
        return None
    if tuple(m.shape) != tuple(shape):
        raise RuntimeError(f"mask shape {tuple(m.shape)} does not match {tuple(shape)}")
    if m.dtype is torch.bool and m.device == dev and m.is_contiguous():
        return m                                        # the usual case, a dozen times per training step: no dispatcher round trips
    return m.to(device=dev).bool().contiguous()


def kernel_width(d: int) -> int:
    """Width the kernels score a d-wide embedding at: 128 (one column block) or 256 (two)."""
    if d <= 0 or d > D_WIDE:
        raise NotImplementedError(f"embedding width {d} unsupported (kernels are built for widths up to {D_WIDE})")
    return D if d <= D else D_WIDE


def pad_width(x: torch.Tensor) -> torch.Tensor:
    """(..., d) -> (..., 128) for d <= 128, (..., 256) for 128 < d <= 256, with zero columns appended (a zero column contributes
    an exact 0 to every dot product, so scores, arg-max and the first d columns of every gradient are those of the narrow
    tensors).  The reference takes any width (evaluator/retrieval.py:173); wider than 256 is not supported here."""
    d = x.shape[-1]
    w = kernel_width(d)
    return x if d == w else torch.nn.functional.pad(x, (0, w - d))


def split_wide(x: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """(..., 256) -> ((4, ..., 128) fp16 planes, absmax word): `split_f32` of the two 128-column blocks under ONE power of two for
    the whole tensor; plane index = 2 * (0 hi | 1 lo) + column block -- the layout evdr_maxsim_fwd_prepared takes as nplanes = 4."""
    if x.shape[-1] != D_WIDE:
        raise RuntimeError(f"split_wide needs rows of width {D_WIDE}")
    lead = tuple(x.shape[:-1])
    blocks = x.float().reshape(-1, 2, D).transpose(0, 1).contiguous()           # (column block, row, 128)
    planes, amax = split_f32(blocks)                                           # (hi | lo, column block, row, 128)
    return planes.view((4,) + lead + (D,)), amax


def workspace(nbytes: int, dev) -> torch.Tensor:
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=dev)


def pack_pmask(pmask: Optional[torch.Tensor], npages: int, lp: int, dev) -> Tuple[torch.Tensor, torch.Tensor]:
    """(np, lp) byte mask -> (tilemask (np, ceil(lp/32)) int32 bits, pageflags (np,) int32)."""
    lib = L.load()
    ntiles = (lp + 31) // 32
    tilemask = torch.empty((npages, ntiles), dtype=torch.int32, device=dev)
    pageflags = torch.empty((npages,), dtype=torch.int32, device=dev)
    pm = _mask_u8(pmask, (npages, lp), dev)
    with L.on(dev):
        L.check(lib.evdr_pack_pmask(L.ptr(pm), npages, lp, L.ptr(tilemask), L.ptr(pageflags),
                                    L.current_stream_handle(dev)))
    return tilemask, pageflags


def flag_nonfinite(P: torch.Tensor, pmask: Optional[torch.Tensor], pageflags: torch.Tensor) -> None:
    """Set bit 3 of pageflags[p] for every page with a NaN / Inf element in a valid patch (evdr_flag_nonfinite): the
    forward kernels then return NaN for that page, where torch.max would have propagated it (evaluator/retrieval.py:201).
    P (np, lp, 128) fp32 / bf16 / fp16 (the hi plane of fp16 hi/lo planes), dense rows, any page stride."""
    dev = _require_cuda(P, pageflags)
    npg, lp, d = P.shape
    if npg == 0 or lp == 0:
        return
    kind = {torch.float32: L.EVDR_F32, torch.bfloat16: L.EVDR_BF16, torch.float16: L.EVDR_F16}.get(P.dtype)
    if kind is None or d != D or P.stride(2) != 1 or P.stride(1) != D:
        raise RuntimeError("flag_nonfinite needs (np, lp, 128) fp32 / bf16 / fp16 pages with dense rows")
    pm = _mask_u8(pmask, (npg, lp), dev)
    lib = L.load()
    with L.on(dev):
        L.check(lib.evdr_flag_nonfinite(L.ptr(P), kind, L.ptr(pm), npg, lp, int(P.stride(0)), L.ptr(pageflags),
                                        L.current_stream_handle(dev)))


def split_f32(x: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """(..., 128) fp32 -> ((2, ..., 128) fp16 planes hi/lo of x * 2^k, absmax word): hi + lo == x * 2^k to ~2^-22, with
    one power of two k per tensor that the kernels derive from the absmax word (int32 tensor of 1 element holding the
    bits of max|x|; keep it with the planes)."""
    dev = _require_cuda(x)
    if x.shape[-1] != D:
        raise NotImplementedError(f"rows of width {x.shape[-1]}: this entry takes 128-wide rows (129..256 columns: `split_wide`; the scorers pad and split by themselves)")
    lib = L.load()
    xc = x.float().contiguous()
    rows = xc.numel() // D
    planes = torch.empty((2,) + tuple(xc.shape), dtype=torch.float16, device=dev)
    amax = torch.empty((1,), dtype=torch.int32, device=dev)
    with L.on(dev):
        L.check(lib.evdr_split_f32(L.ptr(xc), rows, L.ptr(planes), L.ptr(amax), L.current_stream_handle(dev)))
    return planes, amax


def split_f32_segments(x: torch.Tensor, seg: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """`split_f32` of consecutive groups of `seg` leading entries of x (n, ..., 128) in ONE launch -- the batches of a training
    epoch, known when the epoch starts.  Returns (planes, absmax words): planes (nseg, 2 * seg * rows_per_entry * 128) fp16,
    one row per segment, absmax (nseg,) int32; `segment_planes` cuts out segment s in `split_f32`'s format.  Every segment
    keeps its own absmax word: bit for bit the planes `split_f32(x[s * seg:(s + 1) * seg])` makes."""
    dev = _require_cuda(x)
    if x.shape[-1] != D:
        raise NotImplementedError(f"rows of width {x.shape[-1]}: this entry takes 128-wide rows (129..256 columns: `split_wide`; the scorers pad and split by themselves)")
    xc = x.float().contiguous()
    n = xc.shape[0]
    per = (xc.numel() // D) // max(n, 1)                       # rows per leading entry
    if seg < 1 or seg * per > 2048:
        raise ValueError(f"segments of {seg} x {per} rows: one segment is at most 2048 rows")
    nseg = (n + seg - 1) // seg
    planes = torch.empty((nseg, 2 * seg * per * D), dtype=torch.float16, device=dev)
    amax = torch.empty((max(nseg, 1),), dtype=torch.int32, device=dev)
    if n:
        lib = L.load()
        with L.on(dev):
            for s0 in range(0, nseg, 65535):                       # one launch covers at most 65535 segments (grid limit)
                s1 = min(nseg, s0 + 65535)
                rows = (min(n, s1 * seg) - s0 * seg) * per
                L.check(lib.evdr_split_f32_segments(xc.data_ptr() + s0 * seg * per * D * 4, rows, seg * per, planes[s0:].data_ptr(),
                                                    amax[s0:].data_ptr(), L.current_stream_handle(dev)))
    return planes, amax


def segment_planes(planes: torch.Tensor, amax: torch.Tensor, s: int, shape) -> Tuple[torch.Tensor, torch.Tensor]:
    """Segment s of `split_f32_segments` as ((2,) + shape fp16 planes, its absmax word): views, nothing is copied.  `shape` =
    shape of the fp32 batch (a short last batch has fewer leading entries)."""
    n = 1
    for d in shape:
        n *= int(d)
    return planes[s, : 2 * n].view((2,) + tuple(int(d) for d in shape)), amax[s:s + 1]


def maxsim_forward(Q: torch.Tensor, P: torch.Tensor, qmask: Optional[torch.Tensor], pmask: Optional[torch.Tensor],
                   want_argmax: bool = False) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
    """A1 (evaluator/retrieval.py:166-213) through evdr_maxsim_fwd.  bf16 x bf16 inputs are scored as
    they are; anything else is upcast to fp32 like the reference (:176-177) and scored to fp32 accuracy."""
    dev = _require_cuda(Q, P)
    if Q.dim() != 3 or P.dim() != 3 or Q.shape[-1] != P.shape[-1]:
        raise RuntimeError(f"expected Q (Nq,Lq,D) and P (Np,Lp,D), got {tuple(Q.shape)} and {tuple(P.shape)}")
    Q, P = pad_width(Q), pad_width(P)                 # narrower embeddings: zero columns add exact zeros to every dot product
    nq, lq, d = Q.shape
    npg, lp, _ = P.shape
    out = torch.empty((nq, npg), dtype=torch.float32, device=dev)
    arg = torch.empty((nq, npg, lq), dtype=torch.int16, device=dev) if want_argmax else None
    if nq == 0 or npg == 0:
        return out, arg
    if lq == 0 or lp == 0:
        raise RuntimeError(f"zero-length token axis (Lq={lq}, Lp={lp})")
    if lp > 65535 or lq > 65535:
        raise NotImplementedError("token axes longer than 65535 are not supported")
    if d == D_WIDE:
        # 129..256 columns: every dtype is upcast like the reference does (evaluator/retrieval.py:176-177) and scored to fp32
        # accuracy on four planes (fp16 hi/lo x two column blocks), through the prepared entry point
        qp, qa = split_wide(Q)
        pp, pa = split_wide(P)
        tilemask, pageflags = pack_pmask(pmask, npg, lp, dev)
        for cb in (0, 1):
            flag_nonfinite(pp[cb], pmask, pageflags)                          # the hi planes of both column blocks
        return maxsim_forward_prepared(qp, qa, pp, pa, qmask, tilemask, pageflags, want_argmax=want_argmax)
    lib = L.load()
    if Q.dtype == torch.bfloat16 and P.dtype == torch.bfloat16:
        dtype = L.EVDR_BF16
        Qc = Q.contiguous()
        Pc = P if (P.stride(2) == 1 and P.stride(1) == D and P.stride(0) >= lp * D) else P.contiguous()
    else:
        dtype = L.EVDR_F32
        Qc = Q.float().contiguous()
        Pc = P.float().contiguous()
    strides = (L.C.c_int64 * 2)(Qc.stride(0), Pc.stride(0))
    qm = _mask_u8(qmask, (nq, lq), dev)
    pm = _mask_u8(pmask, (npg, lp), dev)
    nbytes = lib.evdr_maxsim_fwd_workspace(nq, lq, npg, lp, dtype)
    ws = workspace(nbytes, dev)
    with L.on(dev):
        L.check(lib.evdr_maxsim_fwd(L.ptr(Qc), L.ptr(Pc), L.ptr(qm), L.ptr(pm), L.ptr(out), L.ptr(arg),
                                    nq, lq, npg, lp, d, dtype, strides, L.ptr(ws), ws.numel(),
                                    L.current_stream_handle(dev)))
    return out, arg


def maxsim_forward_prepared(qplanes: torch.Tensor, qamax: Optional[torch.Tensor], pplanes: torch.Tensor,
                            pamax: Optional[torch.Tensor], qmask: Optional[torch.Tensor], tilemask: torch.Tensor,
                            pageflags: torch.Tensor, want_argmax: bool = False, out: Optional[torch.Tensor] = None,
                            out_col: int = 0) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
    """A1 on PREPARED operands (evdr_maxsim_fwd_prepared): planes from `split_f32` / `l2norm_split` (2 fp16 planes + absmax
    word) or bf16 tensors with a leading plane axis of 1, masks from `pack_pmask`.  Nothing is converted or packed here.
    With `out` (nq, >= out_col + np) given, the scores land in out[:, out_col:out_col+np] (a shard's column block)."""
    dev = _require_cuda(qplanes, pplanes)
    nplanes, nq, lq, _ = qplanes.shape
    _, npg, lp, _ = pplanes.shape
    if (nplanes not in (1, 2, 4) or pplanes.shape[0] != nplanes or (nplanes >= 2) != (qplanes.dtype == torch.float16)
            or pplanes.dtype != qplanes.dtype):
        raise RuntimeError("query and page planes must both be (1, ., ., 128) bf16, both (2, ., ., 128) fp16, or both (4, ., ., 128) "
                           "fp16 (width 256: `split_wide`)")
    if pplanes.stride(3) != 1 or pplanes.stride(2) != D or not qplanes.is_contiguous():
        raise RuntimeError("planes must be dense in their last two dims (queries fully contiguous)")
    if out is None:
        out = torch.empty((nq, npg), dtype=torch.float32, device=dev)
        out_col = 0
    view = out[:, out_col:out_col + npg]
    arg = torch.empty((nq, npg, lq), dtype=torch.int16, device=dev) if want_argmax else None
    if nq == 0 or npg == 0:
        return view, arg
    qm = _mask_u8(qmask, (nq, lq), dev)
    qlist = torch.empty((nq + 1,), dtype=torch.int32, device=dev) if (lq > 32 and qm is not None and not want_argmax) else None
    lib = L.load()
    with L.on(dev):
        L.check(lib.evdr_maxsim_fwd_prepared(
            L.ptr(qplanes), L.ptr(pplanes), L.ptr(qm), L.ptr(tilemask), L.ptr(pageflags), view.data_ptr(), out.stride(0),
            L.ptr(arg), nq, lq, npg, lp, nplanes, int(pplanes.stride(1)), int(pplanes.stride(0)), L.ptr(qamax), L.ptr(pamax),
            L.ptr(qlist), L.current_stream_handle(dev)))
    return view, arg


class ScoreRowCache:
    """Device-side (query row -> score row) cache of ONE prepared, frozen page tensor (include/evdr.h "score-row cache";
    evaluator/retrieval.py builds it for the reference's per-step re-scoring of the frozen teacher,
    mainv2_iter_distill_infonce.py:282-283).  Fixed geometry -- `lq` tokens per query, `row_dtype` rows, `npg` pages -- and a fixed
    `capacity` chosen from a byte budget; nothing is evicted, a full cache simply stops storing.  `nbytes` = device memory held."""

    def __init__(self, lq: int, row_dtype: torch.dtype, npg: int, max_bytes: int, dev, hash_mask: int = 0xFFFFFFFFFFFFFFFF):
        self.lq, self.row_dtype, self.npg, self.dev = int(lq), row_dtype, int(npg), dev
        self.row_bytes = self.lq * D * torch.empty((), dtype=row_dtype).element_size()
        per_entry = self.row_bytes + self.lq + 4 * self.npg + 8 + 4 + 4 * 4        # row, mask, scores, hash, k, four table slots
        self.capacity = int(min(max(int(max_bytes) // per_entry, 0), (1 << 30)))
        if self.capacity < 1:
            raise ValueError(f"score-row cache: a budget of {max_bytes} bytes holds no entry of {per_entry} bytes")
        n_slots = 4
        while n_slots < 2 * self.capacity:                # [2, 4) slots per entry: load factor <= 0.5, <= 16 bytes of table per entry
            n_slots *= 2
        self.slots = torch.zeros((n_slots,), dtype=torch.int32, device=dev)
        self.ent_hash = torch.empty((self.capacity,), dtype=torch.int64, device=dev)
        self.ent_k = torch.empty((self.capacity,), dtype=torch.int32, device=dev)
        self.ent_q = torch.empty((self.capacity, self.row_bytes), dtype=torch.uint8, device=dev)
        self.ent_mask = torch.empty((self.capacity, self.lq), dtype=torch.uint8, device=dev)
        self.ent_scores = torch.empty((self.capacity, self.npg), dtype=torch.float32, device=dev)
        self.n_entries = torch.zeros((1,), dtype=torch.int32, device=dev)
        self.count = torch.zeros((1,), dtype=torch.int32, device=dev)              # view of the last call's miss count (its workspace word)
        self._scratch = {}
        self.c = L.EvdrQCache(self.slots.data_ptr(), n_slots, self.ent_hash.data_ptr(), self.ent_k.data_ptr(), self.ent_q.data_ptr(),
                              self.ent_mask.data_ptr(), self.ent_scores.data_ptr(), self.n_entries.data_ptr(), self.capacity,
                              self.row_bytes, self.lq, self.npg, hash_mask)
        self.nbytes = sum(t.numel() * t.element_size() for t in (self.slots, self.ent_hash, self.ent_k, self.ent_q, self.ent_mask, self.ent_scores))

    def scratch(self, nq: int):
        """(workspace, miss-count view) for batches of nq queries: zeroed once, then owned by the library's ticket protocol."""
        s = self._scratch.get(nq)
        if s is None:
            if len(self._scratch) > 8:
                self._scratch.clear()
            nbytes = int(L.load().evdr_qcache_workspace(nq))
            ws = torch.zeros((nbytes,), dtype=torch.uint8, device=self.dev)
            s = self._scratch[nq] = (ws, ws[nbytes - 256:nbytes - 252].view(torch.int32))
        return s

    def accepts(self, Q: torch.Tensor, npg: int) -> bool:
        return (Q.dim() == 3 and Q.shape[1] == self.lq and Q.shape[2] == D and Q.dtype == self.row_dtype and npg == self.npg
                and Q.device == self.dev and 2 <= self.lq <= 32)


def maxsim_forward_cached(cache: ScoreRowCache, Q: torch.Tensor, qplanes: torch.Tensor, qamax: Optional[torch.Tensor], pplanes: torch.Tensor,
                          pamax: Optional[torch.Tensor], qmask: Optional[torch.Tensor], tilemask: torch.Tensor,
                          pageflags: torch.Tensor) -> torch.Tensor:
    """A1 on prepared operands THROUGH the score-row cache (evdr_maxsim_fwd_prepared_cached): lookup + plan -> forward over the
    missing queries only -> exchange, three launches on the current stream behind ONE C call and no host synchronisation (which queries were missing is decided and consumed on the
    device).  Q is the caller's dense (nq, lq, 128) batch -- its bits are the key --, qplanes / qamax its planes as the forward
    takes them.  The result is, bit for bit, what `maxsim_forward_prepared` returns for the same call."""
    dev = _require_cuda(Q, qplanes, pplanes)
    nq, lq, _ = Q.shape
    nplanes, npg, lp, _ = pplanes.shape
    if not cache.accepts(Q, npg) or not Q.is_contiguous():
        raise RuntimeError("maxsim_forward_cached: the batch does not match the cache's geometry")
    out = torch.empty((nq, npg), dtype=torch.float32, device=dev)
    if nq == 0:
        return out
    qm = _mask_u8(qmask, (nq, lq), dev)
    ws, cache.count = cache.scratch(nq)
    lib = L.load()
    with L.on(dev):
        L.check(lib.evdr_maxsim_fwd_prepared_cached(
            L.C.byref(cache.c), L.ptr(Q), L.ptr(qplanes), L.ptr(pplanes), L.ptr(qm), L.ptr(tilemask), L.ptr(pageflags), L.ptr(out),
            out.stride(0), nq, lp, nplanes, int(pplanes.stride(1)), int(pplanes.stride(0)), L.ptr(qamax), L.ptr(pamax), L.ptr(ws),
            ws.numel(), L.current_stream_handle(dev)))
    return out


def maxsim_backward(g: torch.Tensor, Q: torch.Tensor, qmask: Optional[torch.Tensor], pmask: Optional[torch.Tensor],
                    argmax: torch.Tensor, npg: int, lp: int) -> torch.Tensor:
    """A6: dP (np, lp, 128) fp32 from upstream g (nq, np) and the forward's argmax."""
    dev = _require_cuda(g, Q, argmax)
    nq, lq, d = Q.shape
    if d == D_WIDE:
        # the gather is linear in Q's columns: one launch per 128-column block, same arg-max and weights
        return torch.cat([maxsim_backward(g, Q[..., cb * D:(cb + 1) * D], qmask, pmask, argmax, npg, lp) for cb in (0, 1)], dim=-1)
    lib = L.load()
    gc = g.float().contiguous()
    Qc = Q.float().contiguous()
    qm = _mask_u8(qmask, (nq, lq), dev)
    pm = _mask_u8(pmask, (npg, lp), dev)
    dP = torch.empty((npg, lp, d), dtype=torch.float32, device=dev)
    with L.on(dev):
        L.check(lib.evdr_maxsim_bwd(L.ptr(gc), L.ptr(Qc), L.ptr(qm), L.ptr(pm), L.ptr(argmax), L.ptr(dP),
                                    nq, lq, npg, lp, d, L.current_stream_handle(dev)))
    return dP


def topk(scores: torch.Tensor, k: int, idx_base: int = 0,
         idx_map: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """A8: per-row top-k (score desc, index asc), on device.  scores (nq, n) fp32, row-strided allowed."""
    dev = _require_cuda(scores)
    if scores.dim() != 2 or scores.dtype != torch.float32 or scores.stride(1) != 1:
        raise RuntimeError("scores must be a 2-D fp32 tensor with unit column stride")
    if not (1 <= k <= L.EVDR_TOPK_MAX):
        raise ValueError(f"k={k} outside 1..{L.EVDR_TOPK_MAX}")
    nq, n = scores.shape
    lib = L.load()
    ts = torch.empty((nq, k), dtype=torch.float32, device=dev)
    ti = torch.empty((nq, k), dtype=torch.int32, device=dev)
    im = None
    if idx_map is not None:
        if tuple(idx_map.shape) != (nq, n):
            raise RuntimeError("idx_map must have the shape of scores")
        im = idx_map.to(device=dev, dtype=torch.int32).contiguous()
    row_stride = scores.stride(0) if nq > 1 else max(n, 1)
    wsb = lib.evdr_topk_workspace(nq, n, k)          # > 0: few long rows, ranked by many workgroups in two levels
    ws = workspace(wsb, dev) if wsb else None
    with L.on(dev):
        L.check(lib.evdr_topk(L.ptr(scores), L.ptr(im), nq, n, max(row_stride, n), idx_base, k, L.ptr(ts), L.ptr(ti),
                              L.ptr(ws), wsb, L.current_stream_handle(dev)))
    return ts, ti


def _order_key(x: torch.Tensor) -> torch.Tensor:
    """fp32 -> int32 keys in evdr_topk's order (csrc/topk.hip): a larger key ranks earlier, -0.0 == +0.0, every NaN is the
    greatest key (NaN scores rank first, like torch.topk)."""
    u = x.contiguous().view(torch.int32)
    key = torch.where(u < 0, u ^ 0x7FFFFFFF, u)
    key = torch.where(u == -2147483648, torch.zeros_like(key), key)
    return torch.where(torch.isnan(x), torch.full_like(key, 2147483647), key)


def topk_with_ties(scores: torch.Tensor, k: int, to_host: bool = False, have=None):
    """`topk` plus what a ranking by another tie rule needs: for every row whose k-th score is shared by columns that did
    NOT make the cut (score desc, index asc keeps the lowest indices), ALL columns that rank at or above the k-th one.

    The reference hands every score to trec_eval (mainv2_iter_distill_infonce.py:311-317), which breaks ties by docid
    DESCENDING; a device cut by index ascending would hand it a different candidate set whenever equal scores straddle
    rank k.  With the extra columns the metric layer sees every candidate that can appear in ANY top-k under ANY tie
    rule, so its result equals the all-pairs evaluation for every cut-off <= k.  Candidates are counted on the kernel's own
    order keys (NaN = greatest, -0 = +0), so a row whose top-k holds NaNs is cut where the kernel cut it.
    Returns (top_scores, top_idx, extra) with extra = {row: (column indices int64, their scores fp32)} as host numpy
    arrays -- empty when no tie is cut.  to_host=True: top_scores / top_idx come back as numpy arrays too, through ONE
    device-to-host copy that also carries the per-row candidate counts (a second copy only when a tie was cut).
    `have` = (top_scores, top_idx) of `topk(scores, k)` when the caller has run it already."""
    ts, ti = have if have is not None else topk(scores, k)
    extra = {}
    nq, n = scores.shape
    count = None
    if nq and n > k:
        key = _order_key(scores)
        kth = _order_key(ts[:, k - 1:k])
        count = (key >= kth).sum(dim=1, dtype=torch.int32)
    if to_host:
        cols = [ts.view(torch.int32), ti] + ([count[:, None]] if count is not None else [])
        host = torch.cat(cols, dim=1).cpu().numpy()                                # the one copy (and the one sync)
        ts_h, ti_h = host[:, :k].view("float32"), host[:, k:2 * k]
        cut = (host[:, 2 * k] > k).nonzero()[0] if count is not None else ()
    else:
        cut = (count > k).nonzero().flatten().cpu().numpy() if count is not None else ()
    if len(cut):
        rows = torch.as_tensor(cut, device=scores.device, dtype=torch.int64)
        sel = key[rows] >= kth[rows]                                               # (m, n)
        rc = sel.nonzero()
        packed = torch.stack([rows[rc[:, 0]].to(torch.int32), rc[:, 1].to(torch.int32),
                              scores[rows][sel].contiguous().view(torch.int32)]).cpu().numpy()
        r_all, c_all, s_all = packed[0], packed[1].astype("int64"), packed[2].view("float32")
        starts = (r_all[1:] != r_all[:-1]).nonzero()[0] + 1                        # nonzero() lists row-major: rows are contiguous runs
        for a, b in zip([0, *starts.tolist()], [*starts.tolist(), len(r_all)]):
            extra[int(r_all[a])] = (c_all[a:b], s_all[a:b])
    return (ts_h, ti_h, extra) if to_host else (ts, ti, extra)


def infonce_workspace(b: int, dev) -> torch.Tensor:
    """Workspace of the one-launch loss (`infonce_distill(..., ws=...)`): b per-row losses + the ticket word, ZEROED here.
    The caller owns it: one per object that issues the loss (a FusedStudent), never shared between streams that may run
    concurrently, alive as long as any captured graph holds its address.  Must not be made inside a stream capture (the
    zero fill would only be recorded, and the first replay would find an uninitialised ticket)."""
    if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
        raise RuntimeError("infonce_workspace() inside a stream capture: allocate the loss workspace before capturing")
    return torch.zeros((int(b) + 1,), dtype=torch.float32, device=dev)


def infonce_distill(score_s: torch.Tensor, score_t: torch.Tensor, temperature: float,
                    want_grad: bool, ws: Optional[torch.Tensor] = None,
                    loss_out: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
    """A5 (criterion.py:56-68) + its closed-form gradient in one pass.  Without `ws`: the stateless two-launch form (row
    kernel + mean kernel, scratch allocated per call).  With `ws` = `infonce_workspace(b, dev)` owned by the caller: ONE
    launch (the last workgroup reduces the row losses; same bits) -- the workspace carries a ticket word between calls, so
    it belongs to exactly one issuer.
    `loss_out`: where the scalar goes -- an fp32 scalar on the device, or a PINNED host scalar (hipHostMalloc memory is
    device-accessible under the same address: the kernel's one 4-byte store lands in host memory and a caller that wants
    float(loss) needs no device-to-host copy launch, only an event behind this kernel)."""
    dev = _require_cuda(score_s, score_t)
    if score_s.shape != score_t.shape or score_s.dim() != 2:
        raise RuntimeError("score_s and score_t must be (B, N) and equal-shaped")
    b, n = score_s.shape
    lib = L.load()
    ss = score_s.float().contiguous()
    st = score_t.float().contiguous()
    if loss_out is None:
        loss = torch.empty((), dtype=torch.float32, device=dev)
    else:
        ok = loss_out.dtype == torch.float32 and loss_out.numel() == 1 and (loss_out.device == dev or (not loss_out.is_cuda and loss_out.is_pinned()))
        if not ok:
            raise RuntimeError("infonce_distill: `loss_out` must be one fp32 element on the scores' device or in pinned host memory")
        loss = loss_out
    grad = torch.empty_like(ss) if want_grad else None
    stream = L.current_stream_handle(dev)
    if ws is None:
        row_loss = torch.empty((max(b, 1),), dtype=torch.float32, device=dev)
        with L.on(dev):
            L.check(lib.evdr_infonce_distill_fwd_bwd(L.ptr(ss), L.ptr(st), b, n, float(temperature), L.ptr(loss),
                                                     L.ptr(grad), L.ptr(row_loss), stream))
        return loss, grad
    if ws.dtype != torch.float32 or ws.device != dev or ws.numel() != b + 1 or not ws.is_contiguous():
        raise RuntimeError(f"infonce_distill: `ws` must be infonce_workspace({b}, {dev}) (got {tuple(ws.shape)} {ws.dtype} on {ws.device})")
    with L.on(dev):
        L.check(lib.evdr_infonce_distill_fwd_bwd_ws(L.ptr(ss), L.ptr(st), b, n, float(temperature), L.ptr(loss),
                                                    L.ptr(grad), L.ptr(ws), stream))
    return loss, grad


def l2norm_forward(x: torch.Tensor, rowmask: Optional[torch.Tensor], eps: float, want_planes: bool = False):
    """A4 fused with the row mask: y = m*x / (||m*x|| + eps) over the last (128-wide) dim; returns (y, norms), with
    `want_planes` (y, norms, planes, absmax word): the same launch also leaves y as the scorer's fp16 hi/lo planes
    (`l2norm_split`'s format, constant scale 2^14 since |y| <= 1), so that scoring y needs no absmax + split passes."""
    dev = _require_cuda(x)
    if x.shape[-1] != D or x.dtype != torch.float32:
        raise RuntimeError("l2norm kernel needs fp32 rows of width 128")
    lib = L.load()
    xc = x.contiguous()
    rows = xc.numel() // D
    y = torch.empty_like(xc)
    norm = torch.empty(xc.shape[:-1], dtype=torch.float32, device=dev)
    m = _mask_u8(rowmask, xc.shape[:-1], dev)
    if want_planes:
        planes = torch.empty((2,) + tuple(xc.shape), dtype=torch.float16, device=dev)
        amax = torch.empty((1,), dtype=torch.int32, device=dev)
        with L.on(dev):
            L.check(lib.evdr_l2norm_fwd_split(L.ptr(xc), L.ptr(m), rows, D, float(eps), L.ptr(y), L.ptr(norm), L.ptr(planes), L.ptr(amax),
                                              None, 1, L.current_stream_handle(dev)))
        return y, norm, planes, amax
    with L.on(dev):
        L.check(lib.evdr_l2norm_fwd(L.ptr(xc), L.ptr(m), rows, D, float(eps), L.ptr(y), L.ptr(norm),
                                    L.current_stream_handle(dev)))
    return y, norm


# ---- planes that were made together with an fp32 tensor (l2_normalize's output: utils/preprocess_data.py) -----------------
# The reference's step scores Psb = l2_normalize(Pbar * pmask) right after making it (mainv2_iter_distill_infonce.py:279,286):
# the normalise kernel leaves Psb's planes behind here, keyed on the tensor (address, layout, autograd version) and alive
# only as long as the tensor is; the scorer picks them up instead of running absmax + split over Psb again.
_DERIVED: "collections.OrderedDict" = collections.OrderedDict()
_DERIVED_MAX = 8


def tensor_key(t: Optional[torch.Tensor]):
    """Identity of a tensor's CONTENT for the per-tensor caches (derived planes here, prepared pages and query planes in
    evaluator/retrieval.py): storage address, layout and autograd version counter -- any in-place torch write changes it.
    None = not cacheable: tensors made under torch.inference_mode() track no version counter (`t._version` raises for them), so
    nothing is ever remembered for or looked up by such a tensor; the scorer then prepares its operands per call."""
    if t is None or t.is_inference():
        return None
    return (t.data_ptr(), tuple(t.shape), tuple(t.stride()), t.dtype, t.device.index, t._version)


def remember_planes(y: torch.Tensor, planes: torch.Tensor, amax: torch.Tensor) -> None:
    key = tensor_key(y)
    if key is None:
        return
    _DERIVED[key] = (weakref.ref(y, lambda _r, key=key: _DERIVED.pop(key, None)), planes, amax)
    while len(_DERIVED) > _DERIVED_MAX:
        _DERIVED.popitem(last=False)


def planes_of(t: torch.Tensor):
    """(planes, absmax word) made together with `t` (same storage, layout and autograd version: any in-place torch write
    since then changes the key), or None."""
    key = tensor_key(t)
    hit = _DERIVED.get(key) if key is not None else None
    if hit is None or hit[0]() is None:
        return None
    return hit[1], hit[2]


def l2norm_split(x: torch.Tensor, rowmask: Optional[torch.Tensor], eps: float,
                 pageflags: Optional[torch.Tensor] = None, out=None) -> Tuple[torch.Tensor, torch.Tensor]:
    """y = m*x / (||m*x|| + eps) emitted directly as the scorer's fp16 hi/lo planes (see `split_f32`): ((2, ..., 128) fp16,
    absmax word).  One kernel instead of normalise -> absmax -> split, and no fp32 copy of y in HBM.  With `pageflags`
    (x is (np, lp, 128)) non-finite unmasked rows are reported in their page's flag word on the way (`flag_nonfinite`)."""
    dev = _require_cuda(x)
    if x.shape[-1] != D or x.dtype != torch.float32:
        raise RuntimeError("l2norm kernel needs fp32 rows of width 128")
    lib = L.load()
    xc = x.contiguous()
    rows = xc.numel() // D
    if out is not None:                           # (planes, amax) of an earlier call, overwritten
        planes, amax = out
        if planes.shape != (2,) + tuple(xc.shape) or planes.dtype != torch.float16 or not planes.is_contiguous():
            raise RuntimeError("l2norm_split: `out` planes do not match x")
    else:
        planes = torch.empty((2,) + tuple(xc.shape), dtype=torch.float16, device=dev)
        amax = torch.empty((1,), dtype=torch.int32, device=dev)
    m = _mask_u8(rowmask, xc.shape[:-1], dev)
    with L.on(dev):
        L.check(lib.evdr_l2norm_fwd_split(L.ptr(xc), L.ptr(m), rows, D, float(eps), None, None, L.ptr(planes), L.ptr(amax),
                                          L.ptr(pageflags), int(xc.shape[-2]) if pageflags is not None else 1,
                                          L.current_stream_handle(dev)))
    return planes, amax


def l2norm_backward(gy: torch.Tensor, x: torch.Tensor, rowmask: Optional[torch.Tensor], norm: torch.Tensor,
                    eps: float) -> torch.Tensor:
    dev = _require_cuda(gy, x, norm)
    lib = L.load()
    gc, xc = gy.float().contiguous(), x.contiguous()
    rows = xc.numel() // D
    dx = torch.empty_like(xc)
    m = _mask_u8(rowmask, xc.shape[:-1], dev)
    with L.on(dev):
        L.check(lib.evdr_l2norm_bwd(L.ptr(gc), L.ptr(xc), L.ptr(m), L.ptr(norm), rows, D, float(eps), L.ptr(dx),
                                    L.current_stream_handle(dev)))
    return dx


def maxsim_backward_adamw(g: torch.Tensor, Q: torch.Tensor, qmask: Optional[torch.Tensor], pmask: Optional[torch.Tensor],
                          argmax: torch.Tensor, x: torch.Tensor, exp_avg: torch.Tensor, exp_avg_sq: torch.Tensor,
                          lr: float, betas: Tuple[float, float], eps: float, weight_decay: float, step: int,
                          l2_eps: float = 1e-12, state: Optional[torch.Tensor] = None, next_planes=None,
                          pageflags: Optional[torch.Tensor] = None) -> None:
    """A6 + normalise/mask backward + AdamW in one launch, in place on x / exp_avg / exp_avg_sq (fp32, contiguous).
    `state` (see `adamw_state` / `adamw_advance`): bias corrections come from the device-side step counter instead of
    `step` -- what a HIP-graph replay of the step needs.  `next_planes` = (planes, amax) as `l2norm_split` returns them:
    the launch also leaves l2_normalize(pmask * x_new) there for the next forward (and reports non-finite updated rows in
    `pageflags`), so that the next step needs no normalise pass."""
    dev = _require_cuda(g, Q, argmax, x, exp_avg, exp_avg_sq)
    for t in (x, exp_avg, exp_avg_sq):
        if t.dtype != torch.float32 or not t.is_contiguous() or t.shape != x.shape:
            raise RuntimeError("x / exp_avg / exp_avg_sq must be contiguous fp32 tensors of one shape")
    npg, lp, d = x.shape
    nq, lq, _ = Q.shape
    lib = L.load()
    gc, Qc = g.float().contiguous(), Q.float().contiguous()
    qm = _mask_u8(qmask, (nq, lq), dev)
    pm = _mask_u8(pmask, (npg, lp), dev)
    with L.on(dev):
        planes, amax = next_planes if next_planes is not None else (None, None)
        if planes is not None and (planes.shape != (2,) + tuple(x.shape) or planes.dtype != torch.float16 or not planes.is_contiguous()):
            raise RuntimeError("maxsim_backward_adamw: `next_planes` do not match x")
        L.check(lib.evdr_maxsim_bwd_adamw_planes(L.ptr(gc), L.ptr(Qc), L.ptr(qm), L.ptr(pm), L.ptr(argmax), L.ptr(x),
                                                 L.ptr(exp_avg), L.ptr(exp_avg_sq), nq, lq, npg, lp, d, float(lr),
                                                 float(betas[0]), float(betas[1]), float(eps), float(weight_decay), int(step),
                                                 float(l2_eps), L.ptr(state), L.ptr(planes), L.ptr(amax),
                                                 L.ptr(pageflags) if planes is not None else None,
                                                 L.current_stream_handle(dev)))
    for t in (x, exp_avg, exp_avg_sq):            # written through raw pointers: tell autograd (and version-keyed caches)
        torch.autograd.graph.increment_version(t)


def adamw_step(grad: torch.Tensor, x: torch.Tensor, exp_avg: torch.Tensor, exp_avg_sq: torch.Tensor, lr: float,
               betas: Tuple[float, float], eps: float, weight_decay: float, step: int) -> None:
    """torch.optim.AdamW's update of ONE fp32 tensor in one pass (evdr_adamw_step), in place on x / exp_avg / exp_avg_sq;
    `step` counts this update.  All four tensors dense fp32 of one shape on one device."""
    dev = _require_cuda(grad, x, exp_avg, exp_avg_sq)
    for t in (grad, x, exp_avg, exp_avg_sq):
        if t.dtype != torch.float32 or not t.is_contiguous() or t.shape != x.shape:
            raise RuntimeError("adamw_step: grad / x / exp_avg / exp_avg_sq must be contiguous fp32 tensors of one shape")
    lib = L.load()
    with L.on(dev):
        L.check(lib.evdr_adamw_step(L.ptr(grad), L.ptr(x), L.ptr(exp_avg), L.ptr(exp_avg_sq), x.numel(), float(lr),
                                    float(betas[0]), float(betas[1]), float(eps), float(weight_decay), int(step),
                                    L.current_stream_handle(dev)))
    for t in (x, exp_avg, exp_avg_sq):
        torch.autograd.graph.increment_version(t)


def adamw_state(dev) -> torch.Tensor:
    """Zeroed device-side AdamW step counter {int64 step; float bc1; float bc2_sqrt} (16 bytes)."""
    return torch.zeros(2, dtype=torch.int64, device=dev)


def adamw_advance(state: torch.Tensor, betas: Tuple[float, float]) -> None:
    """step += 1 and refresh the bias corrections, on the device, on the current stream."""
    dev = _require_cuda(state)
    lib = L.load()
    with L.on(dev):
        L.check(lib.evdr_adamw_advance(L.ptr(state), float(betas[0]), float(betas[1]), L.current_stream_handle(dev)))


def maxsim_backward_q(g: torch.Tensor, P: torch.Tensor, qmask: Optional[torch.Tensor], pmask: Optional[torch.Tensor],
                      argmax: torch.Tensor, nq: int, lq: int) -> torch.Tensor:
    """A6, query side: dQ (nq, lq, 128) fp32 from upstream g (nq, np), the fp32 pages and the forward's argmax.
    Deterministic (per-segment partial sums added in a fixed order, no float atomics)."""
    dev = _require_cuda(g, P, argmax)
    npg, lp, d = P.shape
    if d == D_WIDE:
        return torch.cat([maxsim_backward_q(g, P[..., cb * D:(cb + 1) * D], qmask, pmask, argmax, nq, lq) for cb in (0, 1)], dim=-1)
    lib = L.load()
    gc, Pc = g.float().contiguous(), P.float().contiguous()
    qm = _mask_u8(qmask, (nq, lq), dev)
    pm = _mask_u8(pmask, (npg, lp), dev)
    dQ = torch.empty((nq, lq, d), dtype=torch.float32, device=dev)
    ws = workspace(lib.evdr_maxsim_bwd_q_workspace(nq, lq, npg, lp), dev)
    with L.on(dev):
        L.check(lib.evdr_maxsim_bwd_q(L.ptr(gc), L.ptr(Pc), L.ptr(qm), L.ptr(pm), L.ptr(argmax), L.ptr(dQ), nq, lq, npg, lp, d,
                                      L.ptr(ws), ws.numel(), L.current_stream_handle(dev)))
    return dQ
