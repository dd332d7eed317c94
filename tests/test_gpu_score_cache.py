"""The opt-in score-row cache of frozen pages on the drop-in call pattern (the reference re-scores its frozen
teacher every step, /root/reference/mainv2_iter_distill_infonce.py:282-283; SURVEY §8 A7: caching is result-identical).

Every cached call is compared with `torch.equal` against the SAME call with the cache off -- the row a hit returns must be, bit for
bit, what the forward produces for this very batch (the plane shift of the batch is part of the key for that reason).  Covered: hits
/ misses / mixed batches, in-batch duplicates, a hash collision (forced through the test mask) being a miss, the plane shift and the
mask row being part of the key, an in-place write to P being a miss, inference tensors never being cached, the byte budget, the
device-side subset forward on its own."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu


@pytest.fixture()
def R():
    import evdr_amd  # noqa: F401
    from evdr_amd.evaluator import retrieval as R
    R.forget_prepared()
    R.disable_score_cache()
    yield R
    R.disable_score_cache()
    R._SCORE_CACHE_HASH_MASK = 0xFFFFFFFFFFFFFFFF
    R.forget_prepared()


def _unit(gen, *shape):
    return torch.nn.functional.normalize(torch.randn(*shape, generator=gen), dim=-1)


def _plain(R, Q, P, qm, pm):
    """the same call with the cache off (budget 0 keeps the caches that exist, it only bypasses them)"""
    keep, R._SCORE_CACHE_BUDGET = R._SCORE_CACHE_BUDGET, 0
    try:
        with torch.no_grad():
            return R.score_multi_vector_masked(Q, P, qm, pm)
    finally:
        R._SCORE_CACHE_BUDGET = keep


def _cached(R, Q, P, qm, pm):
    with torch.no_grad():
        return R.score_multi_vector_masked(Q, P, qm, pm)


def _the_cache(R):
    live = [c for c in R._SCORE_CACHES.values() if c is not None]
    assert len(live) == 1
    return live[0]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("lp", [70, 206])
def test_cached_scores_equal_uncached_bit_for_bit(R, dtype, lp):
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(11 + lp)
    P = _unit(gen, 37, lp, 128).to(dtype).to(dev)
    pm = (torch.rand(37, lp, generator=gen) > 0.1).to(dev)
    pool = _unit(gen, 40, 32, 128).to(dtype).to(dev)
    qmp = (torch.rand(40, 32, generator=gen) > 0.15).to(dev)
    R.enable_score_cache(64 << 20)
    seen = set()
    for step, idx in enumerate([list(range(0, 12)), list(range(0, 12)), [3, 20, 5, 21, 22, 7], list(range(8, 40)), [39, 39, 1, 1, 30],
                                list(range(40))]):
        ii = torch.tensor(idx, device=dev)
        Q, qm = pool.index_select(0, ii), qmp.index_select(0, ii)
        want = _plain(R, Q, P, qm, pm)
        got = _cached(R, Q, P, qm, pm)
        assert torch.equal(got, want), f"step {step}"
        c = _the_cache(R)
        # fp32 batches carry their plane shift in the key: a row seen in a batch with another absmax exponent is a miss again,
        # so only an upper bound on the misses holds there; bf16 has no shift and the count is exact
        new = [i for i in idx if i not in seen]
        misses = int(c.count.item())
        if dtype == torch.bfloat16:
            assert misses == len(new), (step, misses, len(new))      # duplicates inside a batch both miss, both are scored
        else:
            assert misses >= len(new) and (step != 1 or misses == 0)
        seen.update(idx)
    st = R.score_cache_stats()
    assert st["caches"] == 1 and st["entries"] >= 40 and st["bytes"] <= 64 << 20


def test_hash_collision_is_a_miss(R):
    """hash mask 0: every row has the same 64-bit hash -- only the full-row comparison tells them apart."""
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(5)
    P = _unit(gen, 21, 96, 128).bfloat16().to(dev)
    pm = torch.ones(21, 96, dtype=torch.bool, device=dev)
    R._SCORE_CACHE_HASH_MASK = 0
    R.enable_score_cache(32 << 20)
    batches = [_unit(gen, 8, 32, 128).bfloat16().to(dev) for _ in range(5)]
    qm = torch.ones(8, 32, dtype=torch.bool, device=dev)
    for b in batches:                                        # 40 colliding rows: the probe window holds 32 of them
        assert torch.equal(_cached(R, b, P, qm, pm), _plain(R, b, P, qm, pm))
    c = _the_cache(R)
    assert int(c.n_entries.item()) == 40                     # (entry indices are handed out; the last eight found no slot in the window)
    for i, b in enumerate(batches):
        assert torch.equal(_cached(R, b, P, qm, pm), _plain(R, b, P, qm, pm))
        assert int(c.count.item()) == (0 if i < 4 else 8), i  # the first 32 rows hit (after walking up to 32 colliding entries), the rest are scored


def test_plane_shift_and_mask_row_are_part_of_the_key(R):
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(6)
    P = _unit(gen, 19, 130, 128).to(dev)
    pm = torch.ones(19, 130, dtype=torch.bool, device=dev)
    A = _unit(gen, 8, 32, 128).to(dev)
    A[0, 0, 0] = 0.9                                          # every batch that holds row 0 has its absmax in [0.5, 1): one plane shift
    qm = torch.ones(8, 32, dtype=torch.bool, device=dev)
    R.enable_score_cache(32 << 20)
    assert torch.equal(_cached(R, A, P, qm, pm), _plain(R, A, P, qm, pm))
    c = _the_cache(R)
    # the same four rows in a batch whose absmax is 8x larger: another power of two for the planes -> scored again, and equal to
    # what the forward gives for THIS batch
    big = A[4:5].clone()
    big[0, 0, 0] = 6.0
    B = torch.cat([A[:4], big])
    assert torch.equal(_cached(R, B, P, qm[:5], pm), _plain(R, B, P, qm[:5], pm))
    assert int(c.count.item()) == 5
    # back in a batch with the first exponent: hits
    assert torch.equal(_cached(R, A[:4].clone(), P, qm[:4], pm), _plain(R, A[:4].clone(), P, qm[:4], pm))
    assert int(c.count.item()) == 0
    # another mask row for a known query row: a miss
    qm2 = qm.clone()
    qm2[2, 20:] = False
    assert torch.equal(_cached(R, A, P, qm2, pm), _plain(R, A, P, qm2, pm))
    assert int(c.count.item()) == 1


def test_inplace_write_to_the_pages_is_a_miss_and_dead_tensors_free_their_cache(R):
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(7)
    P = _unit(gen, 16, 64, 128).to(dev)
    pm = torch.ones(16, 64, dtype=torch.bool, device=dev)
    Q = _unit(gen, 6, 32, 128).to(dev)
    qm = torch.ones(6, 32, dtype=torch.bool, device=dev)
    R.enable_score_cache(32 << 20)
    s0 = _cached(R, Q, P, qm, pm)
    assert torch.equal(_cached(R, Q, P, qm, pm), s0) and int(_the_cache(R).count.item()) == 0
    P[3].mul_(-1.0)                                           # in place: the autograd version counter moves, the key with it
    s1 = _cached(R, Q, P, qm, pm)
    assert not torch.equal(s1, s0) and torch.equal(s1, _plain(R, Q, P, qm, pm))
    live = [c for c in R._SCORE_CACHES.values() if c is not None]
    assert len(live) == 1 and int(live[0].n_entries.item()) == 6              # a fresh cache under the new key (the stale one gave way to the budget)
    del P, s0, s1
    import gc
    gc.collect()
    assert len(R._SCORE_CACHES) == 0 and len(R._PREPARED) == 0


def test_inference_tensors_are_never_cached(R):
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(8)
    R.enable_score_cache(32 << 20)
    with torch.inference_mode():
        P = _unit(gen, 9, 64, 128).to(dev)
        pm = torch.ones(9, 64, dtype=torch.bool, device=dev)
        Q = _unit(gen, 4, 32, 128).to(dev)
        qm = torch.ones(4, 32, dtype=torch.bool, device=dev)
        a = R.score_multi_vector_masked(Q, P, qm, pm)
        b = R.score_multi_vector_masked(Q, P, qm, pm)
    assert torch.equal(a, b) and len(R._SCORE_CACHES) == 0 and R.score_cache_stats()["caches"] == 0


def test_memory_is_bounded_by_the_byte_budget(R):
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(9)
    P = _unit(gen, 50, 40, 128).bfloat16().to(dev)
    pm = torch.ones(50, 40, dtype=torch.bool, device=dev)
    budget = 200_000                                          # a row costs 8 192 + 32 + 200 + 28 bytes: 23 entries
    R.enable_score_cache(budget)
    qm = torch.ones(16, 32, dtype=torch.bool, device=dev)
    batches = [_unit(gen, 16, 32, 128).bfloat16().to(dev) for _ in range(3)]
    for b in batches:
        assert torch.equal(_cached(R, b, P, qm, pm), _plain(R, b, P, qm, pm))
    st = R.score_cache_stats()
    assert st["capacity"] == 23 and st["entries"] == 23 and st["bytes"] <= budget
    c = _the_cache(R)
    assert torch.equal(_cached(R, batches[0], P, qm, pm), _plain(R, batches[0], P, qm, pm)) and int(c.count.item()) == 0
    assert torch.equal(_cached(R, batches[2], P, qm, pm), _plain(R, batches[2], P, qm, pm)) and int(c.count.item()) == 16   # never stored: scored
    # a second frozen tensor shares the budget: the older cache is dropped to make room
    P2 = _unit(gen, 50, 40, 128).bfloat16().to(dev)
    assert torch.equal(_cached(R, batches[0], P2, qm, pm), _plain(R, batches[0], P2, qm, pm))
    assert R.score_cache_stats()["bytes"] <= budget
    # a budget that holds no row at all: the call is simply scored
    R.enable_score_cache(1000)
    R.forget_prepared()
    assert torch.equal(_cached(R, batches[1], P, qm, pm), _plain(R, batches[1], P, qm, pm)) and R.score_cache_stats()["caches"] == 0


def test_queries_that_need_a_gradient_and_other_geometries_bypass_the_cache(R):
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(10)
    P = _unit(gen, 12, 64, 128).to(dev)
    pm = torch.ones(12, 64, dtype=torch.bool, device=dev)
    R.enable_score_cache(32 << 20)
    Q = _unit(gen, 5, 32, 128).to(dev).requires_grad_(True)
    qm = torch.ones(5, 32, dtype=torch.bool, device=dev)
    s = R.score_multi_vector_masked(Q, P, qm, pm)
    s.sum().backward()
    assert Q.grad is not None and R.score_cache_stats()["caches"] == 0
    # queries longer than 32 tokens and single-token packs: scored plainly, equal to the cache-off call
    for lq in (1, 40):
        Ql = _unit(gen, 6, lq, 128).to(dev)
        qml = torch.ones(6, lq, dtype=torch.bool, device=dev)
        assert torch.equal(_cached(R, Ql, P, qml, pm), _plain(R, Ql, P, qml, pm))
    assert R.score_cache_stats()["caches"] == 0
    # a cache made for 32-token batches is not used for 20-token ones (and the call is still right)
    Q32, Q20 = _unit(gen, 4, 32, 128).to(dev), _unit(gen, 4, 20, 128).to(dev)
    m32, m20 = torch.ones(4, 32, dtype=torch.bool, device=dev), torch.ones(4, 20, dtype=torch.bool, device=dev)
    assert torch.equal(_cached(R, Q32, P, m32, pm), _plain(R, Q32, P, m32, pm))
    assert torch.equal(_cached(R, Q20, P, m20, pm), _plain(R, Q20, P, m20, pm))
    assert R.score_cache_stats()["entries"] == 4


def test_subset_forward_scores_exactly_the_listed_queries(R):
    from evdr_amd import _lib as L, ops
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(12)
    for dtype, nq in ((torch.float32, 32), (torch.bfloat16, 70), (torch.bfloat16, 7)):
        P = _unit(gen, 45, 206, 128).to(dtype).to(dev)
        Q = _unit(gen, nq, 32, 128).to(dtype).to(dev)
        if dtype == torch.float32:
            (qp, qa), (pp, pa) = ops.split_f32(Q), ops.split_f32(P)
        else:
            qp, qa, pp, pa = Q[None], None, P[None], None
        tm, pf = ops.pack_pmask(None, 45, 206, dev)
        full, _ = ops.maxsim_forward_prepared(qp, qa, pp, pa, None, tm, pf)
        lib = L.load()
        for sel in ([], [0], [nq - 1], list(range(0, nq, 3)), list(range(nq))):
            out = torch.full((nq, 45), -7.0, device=dev)
            qsel = torch.tensor(sel + [0] * (nq - len(sel)), dtype=torch.int32, device=dev)
            cnt = torch.tensor([len(sel)], dtype=torch.int32, device=dev)
            L.check(lib.evdr_maxsim_fwd_prepared_subset(L.ptr(qp), L.ptr(pp), None, L.ptr(tm), L.ptr(pf), L.ptr(out), 45, nq, 32, 45, 206,
                                                        qp.shape[0], 206 * 128, 45 * 206 * 128, L.ptr(qa), L.ptr(pa), L.ptr(qsel), L.ptr(cnt),
                                                        L.current_stream_handle(dev)))
            want = torch.full((nq, 45), -7.0, device=dev)
            if sel:
                want[sel] = full[sel]
            assert torch.equal(out, want), (dtype, nq, sel[:4])
    rc = L.load().evdr_maxsim_fwd_prepared_subset(None, None, None, None, None, None, 45, 8, 40, 45, 206, 1, 206 * 128, 0, None, None, None, None, None)
    assert rc == L.EVDR_ERR_SHAPE
