"""The N>1 exchange step rehearsed on CPU: 2 ranks, gloo.  Each rank owns a contiguous page shard, builds
per-query candidate lists, all-gathers them with the product's gather_candidates(), and every rank must end
up with the same rank-major candidate matrix, whose merge equals the single-shard ranking."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import evdr_amd  # noqa: F401
        from evdr_amd.corpus import gather_candidates, shard_range
        from oracle import maxsim_oracle as O
        g = torch.Generator().manual_seed(42)                  # same "global" scores on every rank
        nq, npg, k = 6, 101, 10
        scores = torch.randn(nq, npg, generator=g)
        scores[:, 7] = scores[:, 77]                           # cross-shard tie
        lo, hi = shard_range(npg, rank, world)
        ls, li = O.topk_rows(scores[:, lo:hi], k)              # stand-in for the local HIP top-k (no GPU here)
        sc, ix = gather_candidates(ls, li + lo)
        assert sc.shape == (nq, world * k) and ix.dtype == torch.int32
        # merge contract: score desc, global index asc
        order = torch.sort(-sc, dim=1, stable=True).indices    # rank-major lists keep index order among ties
        ms, mi = sc.gather(1, order)[:, :k], ix.gather(1, order)[:, :k]
        fs, fi = O.topk_rows(scores, k)
        ok = torch.equal(ms, fs) and torch.equal(mi, fi)
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


def test_two_rank_candidate_exchange():
    world = 2
    port = 29500 + (os.getpid() % 2000)
    with mp.Manager() as m:
        ret = m.dict()
        mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
        assert dict(ret) == {0: True, 1: True}


def _worker_cols(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import evdr_amd  # noqa: F401
        from evdr_amd.driver import _GatherColumns
        g = torch.Generator().manual_seed(7)
        full = torch.randn(5, 11, generator=g)                      # the "global" (B, N) score matrix
        up = torch.randn(5, 11, generator=g)                        # upstream gradient on the full rows
        sizes = (6, 5)                                              # ragged page shards
        lo = sum(sizes[:rank])
        block = full[:, lo:lo + sizes[rank]].clone().requires_grad_(True)
        out = _GatherColumns.apply(block, sizes, None)
        (out * up).sum().backward()
        ret[rank] = bool(torch.equal(out.detach(), full) and torch.equal(block.grad, up[:, lo:lo + sizes[rank]]))
    finally:
        dist.destroy_process_group()


def test_two_rank_score_column_gather():
    """Training partitioning (§8(e)): forward = all-gather of ragged column blocks, backward = own columns only."""
    world = 2
    port = 29900 + (os.getpid() % 1000)
    with mp.Manager() as m:
        ret = m.dict()
        mp.spawn(_worker_cols, args=(world, port, ret), nprocs=world, join=True)
        assert dict(ret) == {0: True, 1: True}


def _worker_ties(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import numpy as np
        import evdr_amd  # noqa: F401
        from evdr_amd.corpus import gather_candidates, shard_range, tie_candidates
        from evdr_amd.evaluator.metrics import EvalIndex, evaluate, evaluate_topk
        from oracle import maxsim_oracle as O
        g = torch.Generator().manual_seed(43)
        nq, npg, k = 9, 157, 20
        scores = (torch.randn(nq, npg, generator=g) * 1.5).round() / 2      # quantised: tie runs across the cut AND across shards
        scores[0] = torch.randn(npg, generator=g)                           # one row without ties
        scores[1, [3, 90]] = float("nan")                                   # NaNs rank first on the device
        lo, hi = shard_range(npg, rank, world)
        local = scores[:, lo:hi].contiguous()
        ls, li = O.topk_rows(torch.nan_to_num(local, nan=float("inf")), k)  # stand-in for the local HIP top-k (NaN first)
        ls = local.gather(1, li.long())
        sc, ix = gather_candidates(ls, li + lo)
        key = torch.nan_to_num(sc, nan=float("inf"))
        order = torch.sort(-key, dim=1, stable=True).indices
        ts, ti = sc.gather(1, order)[:, :k], ix.gather(1, order)[:, :k]
        extra = tie_candidates(local, lo, ts[:, k - 1], k)
        # reference: the single-process completion on the full matrix
        want = {}
        full_key = torch.nan_to_num(scores, nan=float("inf"))
        for r in range(nq):
            kth = torch.nan_to_num(ts[r, k - 1], nan=float("inf"))
            cols = (full_key[r] >= kth).nonzero().flatten()
            if len(cols) > k:
                want[r] = cols.numpy()
        ok = set(extra) == set(want) and all(np.array_equal(extra[r][0], want[r]) for r in want) and 0 not in extra and len(extra) >= 6
        ok = ok and all(np.array_equal(extra[r][1], scores[r, extra[r][0]].numpy(), equal_nan=True) for r in extra)
        # and the metric from the sharded candidates equals the all-pairs evaluation (rows without NaN)
        docids = [f"d{(i * 7919) % npg:04d}" for i in range(npg)]
        qkeys = [f"q{i}" for i in range(nq)]
        rel = {qkeys[i]: {docids[int(c)]: 1 for c in (want[i][-2:] if i in want else [int(scores[i].argmax())])} for i in range(nq) if i != 1}
        allpairs = {qkeys[i]: {docids[j]: float(scores[i, j]) for j in range(npg)} for i in range(nq) if i != 1}
        got = evaluate_topk(EvalIndex(rel, qkeys, docids, [1, 5, 10, 20]), ts.numpy(), ti.numpy(), extra)
        ok = ok and got == evaluate(rel, allpairs, [1, 5, 10, 20])
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


def test_two_rank_tie_completion():
    """corpus.tie_candidates: equal scores straddling the merged top-k cut, in both shards; every rank ends up with every
    candidate a docid-descending tie rule could rank inside the top k, and the metric equals the all-pairs one."""
    world = 2
    port = 28500 + (os.getpid() % 1000)
    with mp.Manager() as m:
        ret = m.dict()
        mp.spawn(_worker_ties, args=(world, port, ret), nprocs=world, join=True)
        assert dict(ret) == {0: True, 1: True}
