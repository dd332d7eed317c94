"""Resident page corpus, per-shard top-k and the multi-GPU merge (SURVEY §8(d),(e); BASELINE.json config 4).

None of this exists in the reference (single process, whole corpus on one device, ranking on the host from a
dict of every score).  MI355X layout: the corpus is stored page-major in HBM as 16-bit planes (1 plane = a bf16
corpus, 2 planes = an fp32 corpus split into fp16 hi/lo), 263 680 B per 1030-patch page and plane, plus 4 B per
32-patch tile of packed mask; it is prepared once and stays resident (100 k pages = 26.4 GB of 288 GB).
Pages shard contiguously over ranks; queries are replicated; each rank scores its shard and keeps k
candidates per query; ONE all-gather of (nq, 2k) int32 words per rank (scores bit-cast next to global page
indices: 0.8 MB per rank at nq = 1024, k = 100) moves them; every rank merges with the same top-k kernel.
"""
from __future__ import annotations

from typing import Optional, Sequence, Tuple

import torch

from . import _lib as L
from . import ops


def shard_range(n_pages: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous shard [lo, hi) of rank `rank`: ceil-sized shards, the last ones may be short or empty."""
    per = (n_pages + world - 1) // world
    lo = min(rank * per, n_pages)
    return lo, min(lo + per, n_pages)


class PageCorpus:
    """One rank's prepared, HBM-resident slice of the page corpus."""

    def __init__(self, planes: torch.Tensor, tilemask: torch.Tensor, pageflags: torch.Tensor, idx_base: int = 0,
                 amax: Optional[torch.Tensor] = None):
        ok = planes.dim() == 4 and planes.shape[-1] == ops.D and (
            (planes.dtype == torch.bfloat16 and planes.shape[0] == 1) or
            (planes.dtype == torch.float16 and planes.shape[0] in (2, 4) and amax is not None))
        if not ok:
            raise RuntimeError("planes must be (1, np, lp, 128) bf16, (2, np, lp, 128) fp16 hi/lo or (4, np, lp, 128) fp16 hi/lo x two "
                               "column blocks (width 256: ops.split_wide), the fp16 forms with their absmax word")
        self.planes = planes
        self.amax = amax                     # absmax word of the fp32 tensor the fp16 planes were split from
        self.tilemask = tilemask
        self.pageflags = pageflags
        self.idx_base = int(idx_base)
        self.nplanes, self.n_pages, self.lp, _ = planes.shape
        if planes.stride(3) != 1 or planes.stride(2) != ops.D:
            raise RuntimeError("planes must be dense in their last two dims")
        self.p_stride, self.p_plane_stride = int(planes.stride(1)), int(planes.stride(0))
        self.device = planes.device
        self._ws: Optional[torch.Tensor] = None       # (nq, n_pages) score block of topk(), kept between calls (410 MB at 1024 x 100k)
        # bench.py: a list here makes topk() record a HIP-event pair around its MaxSim launch (on the launch stream) and append
        # it -- the scorer's own share of every TIMED step; the launches themselves are the ones every search issues
        self.score_events: Optional[list] = None

    @classmethod
    def from_tensor(cls, P: torch.Tensor, pmask: Optional[torch.Tensor] = None, idx_base: int = 0) -> "PageCorpus":
        """P (np, lp, d): at d = 128 bf16 is kept as one plane and anything else is treated as fp32 and split into fp16 hi/lo; narrower
        embeddings ride on zero columns; 129 <= d <= 256 is kept as four planes (fp16 hi/lo x two column blocks, any input dtype)."""
        dev = ops._require_cuda(P)
        npg, lp, _ = P.shape
        P = ops.pad_width(P)
        if P.shape[-1] == ops.D_WIDE:
            planes, amax = ops.split_wide(P)
        else:
            planes, amax = (P.contiguous()[None], None) if P.dtype == torch.bfloat16 else ops.split_f32(P)
        tilemask, pageflags = ops.pack_pmask(pmask, npg, lp, dev)
        for hi in range(planes.shape[0] // 2 if planes.shape[0] > 1 else 1):       # the hi plane of every column block
            ops.flag_nonfinite(planes[hi], pmask, pageflags)                       # once per corpus: NaN / Inf pages score NaN (evdr.h)
        return cls(planes, tilemask, pageflags, idx_base, amax)

    def shard(self, lo: int, hi: int) -> "PageCorpus":
        """Pages [lo, hi) of this corpus as a corpus of their own (views, nothing copied)."""
        return PageCorpus(self.planes[:, lo:hi], self.tilemask[lo:hi], self.pageflags[lo:hi], self.idx_base + lo, self.amax)

    def _query_planes(self, Q: torch.Tensor) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
        if self.nplanes == 4:                             # a 256-wide corpus: queries of 129..256 columns, any dtype
            if ops.kernel_width(Q.shape[-1]) != ops.D_WIDE:
                raise RuntimeError(f"this corpus holds embeddings of width 129..256; got queries of width {Q.shape[-1]}")
            return ops.split_wide(ops.pad_width(Q))
        if Q.shape[-1] != ops.D:
            Q = ops.pad_width(Q)
            if Q.shape[-1] != ops.D:
                raise RuntimeError(f"this corpus holds embeddings of width <= 128; got queries of width {Q.shape[-1]}")
        if self.nplanes == 1:
            if Q.dtype != torch.bfloat16:
                raise RuntimeError("bf16 corpus needs bf16 queries (round them explicitly with .bfloat16(), or "
                                   "build the corpus from fp32 to score at fp32 accuracy)")
            return Q.contiguous()[None], None
        return ops.split_f32(Q)

    def score(self, Q: torch.Tensor, qmask: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None,
              out_col: int = 0, qplanes=None) -> torch.Tensor:
        """(nq, n_pages) fp32 scores of this shard; with `out` given, written into out[:, out_col:out_col+n_pages].
        `qplanes` = (planes, absmax word) of Q from `ops.split_f32`, when the caller has split the batch already (a training
        step scores the same queries against teacher and student)."""
        dev = ops._require_cuda(Q)
        nq, lq, _ = Q.shape
        if out is None:
            out = torch.empty((nq, self.n_pages), dtype=torch.float32, device=dev)
            out_col = 0
        view = out[:, out_col:out_col + self.n_pages]
        if nq == 0 or self.n_pages == 0:
            return view
        qp, qamax = qplanes if (qplanes is not None and self.nplanes == 2) else self._query_planes(Q)
        ops.maxsim_forward_prepared(qp, qamax, self.planes, self.amax, qmask, self.tilemask, self.pageflags, out=out,
                                    out_col=out_col)
        return view

    def topk(self, Q: torch.Tensor, qmask: Optional[torch.Tensor], k: int) -> Tuple[torch.Tensor, torch.Tensor]:
        """Per-query top-k of this shard with GLOBAL page indices (idx_base added): (nq,k) fp32, (nq,k) int32.
        Two C-ABI calls on the current stream, no sync: evdr_maxsim_fwd_prepared into this corpus's score block (kept between
        calls: 410 MB at 1024 x 100k), then evdr_topk over it (evdr_maxsim_topk is the same two launches behind one call, for C
        callers; tests pin the two forms to the same bits).  With `score_events` set (bench.py) a HIP-event pair is recorded around
        the MaxSim launch -- the ONLY difference between the timed step of the bench and what `ShardedRetriever.search` runs."""
        dev = ops._require_cuda(Q)
        nq, lq, _ = Q.shape
        if not (1 <= k <= L.EVDR_TOPK_MAX):               # before the early returns: an invalid k never passes silently
            raise ValueError(f"k={k} outside 1..{L.EVDR_TOPK_MAX}")
        if nq == 0:
            return (torch.empty((0, k), dtype=torch.float32, device=dev), torch.empty((0, k), dtype=torch.int32, device=dev))
        if self.n_pages == 0:
            return (torch.full((nq, k), float("-inf"), dtype=torch.float32, device=dev), torch.full((nq, k), -1, dtype=torch.int32, device=dev))
        qp, qamax = self._query_planes(Q)
        # the score block is a FLAT buffer that only grows (410 MB at 1024 x 100k): a short last batch, or alternating batch sizes,
        # view its head instead of reallocating it
        need = nq * self.n_pages
        if self._ws is None or self._ws.numel() < need or self._ws.device != dev:
            self._ws = torch.empty((need,), dtype=torch.float32, device=dev)
        ws = self._ws[:need].view(nq, self.n_pages)
        ev = None
        if self.score_events is not None:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        ops.maxsim_forward_prepared(qp, qamax, self.planes, self.amax, qmask, self.tilemask, self.pageflags, out=ws)
        if ev is not None:
            ev[1].record()
            self.score_events.append(ev)
        return ops.topk(ws, k, idx_base=self.idx_base)


# ---- candidate exchange ---------------------------------------------------------------------------
def pack_candidates(scores: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    """(nq,k) fp32 + (nq,k) int32 -> one (nq, 2k) int32 message (scores bit-cast), so one collective moves both."""
    return torch.cat([scores.contiguous().view(torch.int32), idx.to(torch.int32)], dim=1).contiguous()


def unpack_candidates(buf: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """(world, nq, 2k) int32 -> rank-major candidate lists (nq, world*k) scores fp32 and global indices int32."""
    world, nq, two_k = buf.shape
    k = two_k // 2
    sc = buf[:, :, :k].contiguous().view(torch.float32).permute(1, 0, 2).reshape(nq, world * k)
    ix = buf[:, :, k:].permute(1, 0, 2).reshape(nq, world * k)
    return sc.contiguous(), ix.contiguous()


def gather_candidates(scores: torch.Tensor, idx: torch.Tensor, group=None) -> Tuple[torch.Tensor, torch.Tensor]:
    """All-gather every rank's (nq,k) candidates.  backend nccl (= RCCL over xGMI): device tensors go straight
    into ncclAllGather; backend gloo (CPU rehearsal): staged through host memory."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    msg = pack_candidates(scores, idx)
    backend = dist.get_backend(group)
    if backend == "gloo" and msg.is_cuda:
        host = msg.cpu()
        buf = torch.empty((world * host.shape[0], host.shape[1]), dtype=torch.int32)
        dist.all_gather_into_tensor(buf, host, group=group)
        buf = buf.to(msg.device)
    else:
        # output = the ranks' messages concatenated along dim 0 (the layout both RCCL and gloo accept)
        buf = torch.empty((world * msg.shape[0], msg.shape[1]), dtype=torch.int32, device=msg.device)
        dist.all_gather_into_tensor(buf, msg, group=group)
    return unpack_candidates(buf.view(world, msg.shape[0], msg.shape[1]))


def merge_candidates(scores: torch.Tensor, idx: torch.Tensor, k: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """Rank-major candidate lists -> final top-k (score desc, global index asc) with the HIP top-k kernel."""
    return ops.topk(scores, k, idx_map=idx)


def tie_candidates(local_scores: torch.Tensor, idx_base: int, kth_scores: torch.Tensor, k: int, group=None):
    """Tie completion across shards (the page-sharded form of `ops.topk_with_ties`): for every query row whose GLOBAL k-th
    score is shared by pages that did not make the merged cut, all pages of ALL shards that rank at or above it --
    {row: (global page indices int64 ascending, scores fp32)} as host numpy arrays, the same on every rank.
    local_scores (nq, n_local): this rank's score block; kth_scores (nq,): the merged top-k's last column.
    One all-reduce of nq int32 counts (always); an object all-gather of the tied triples only when some row is cut.  The
    metric layer then ranks the tied candidates by trec_eval's rule (docid descending) exactly as the reference's all-pairs
    evaluation would (mainv2_iter_distill_infonce.py:311-319)."""
    import numpy as np
    import torch.distributed as dist
    multi = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
    nq = local_scores.shape[0]
    key = ops._order_key(local_scores) if local_scores.shape[1] else local_scores.new_zeros((nq, 0), dtype=torch.int32)
    kth = ops._order_key(kth_scores.reshape(nq, 1).to(local_scores.device))
    total = (key >= kth).sum(dim=1, dtype=torch.int32)
    if multi:
        if dist.get_backend(group) == "gloo" and total.is_cuda:
            host = total.cpu()
            dist.all_reduce(host, group=group)
            total = host
        else:
            dist.all_reduce(total, group=group)
    cut = (total > k).nonzero().flatten().to(local_scores.device)
    if cut.numel() == 0:
        return {}
    sel = key[cut] >= kth[cut]
    rc = sel.nonzero()
    mine = (cut[rc[:, 0]].cpu().numpy().astype(np.int64), (rc[:, 1] + int(idx_base)).cpu().numpy().astype(np.int64),
            local_scores[cut][sel].float().cpu().numpy())
    parts = [mine]
    if multi:
        parts = [None] * dist.get_world_size(group)
        dist.all_gather_object(parts, mine, group=group)
    rows = np.concatenate([p[0] for p in parts])
    cols = np.concatenate([p[1] for p in parts])
    sc = np.concatenate([p[2] for p in parts])
    order = np.lexsort((cols, rows))                                   # by row, then global page index (shards are contiguous)
    rows, cols, sc = rows[order], cols[order], sc[order]
    starts = np.flatnonzero(rows[1:] != rows[:-1]) + 1
    return {int(rows[a]): (cols[a:b], sc[a:b]) for a, b in zip([0, *starts.tolist()], [*starts.tolist(), len(rows)])}


class ShardedRetriever:
    """Late-interaction retrieval over a page-sharded corpus: local MaxSim + top-k, one all-gather, merge."""

    def __init__(self, shard: PageCorpus, group=None):
        self.shard = shard
        self.group = group

    def search(self, Q: torch.Tensor, qmask: Optional[torch.Tensor], k: int = 100, with_ties: bool = False):
        """(top_scores, top_idx) of the whole corpus, (score desc, global index asc), identical on every rank.  with_ties=True
        also returns `extra` (see `tie_candidates`): the candidates a metric with another tie rule needs when equal scores
        straddle rank k -- (top_scores, top_idx, extra); it keeps the shard's score block (nq x n_local fp32) for one more pass."""
        import torch.distributed as dist
        multi = dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1
        if not with_ties:
            ls, li = self.shard.topk(Q, qmask, k)
            if not multi:
                return ls, li
            sc, ix = gather_candidates(ls, li, self.group)
            return merge_candidates(sc, ix, k)
        scores = self.shard.score(Q, qmask)
        if self.shard.n_pages:
            ls, li = ops.topk(scores, min(k, self.shard.n_pages), idx_base=self.shard.idx_base)
            if ls.shape[1] < k:                                        # a shard shorter than k: holes at the end, like evdr_maxsim_topk
                pad = k - ls.shape[1]
                ls = torch.cat([ls, ls.new_full((ls.shape[0], pad), float("-inf"))], dim=1)
                li = torch.cat([li, li.new_full((li.shape[0], pad), -1)], dim=1)
        else:
            ls, li = self.shard.topk(Q, qmask, k)
        ts, ti = merge_candidates(*gather_candidates(ls, li, self.group), k) if multi else (ls, li)
        return ts, ti, tie_candidates(scores, self.shard.idx_base, ts[:, k - 1], k, self.group)
