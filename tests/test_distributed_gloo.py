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
