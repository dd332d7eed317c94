"""BASELINE.json's full sizes on the GPU.  Every score of configs[2] (256 x 6 847, bf16 and fp32 inputs) and configs[3]
(64 x 100 000 and the bench's own 1024 x 100 000) is compared with the oracle's own torch code run on device tensors
(`oracle_scores_on_device`; the host-CPU oracle sees samples it finishes in seconds, and the two runs of the oracle are tied
together on those samples), a failure names the (query, page) pairs with their per-token maxima; plus the size-independent
properties of the domain: shard/merge == single shard, patch-permutation invariance, token additivity, idempotence, sortedness
and score/index consistency of the top-k, planted-target recall."""
import os

import numpy as np
import pytest
import torch

from oracle import maxsim_oracle as O

import sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))      # golden_recipes

pytestmark = pytest.mark.gpu

LP, D, LQ = 1030, 128, 32


@pytest.fixture(scope="module")
def dev():
    import evdr_amd  # noqa: F401
    return torch.device("cuda:0")


def synth(n_pages, n_q, dev, seed):
    """bf16 unit-norm pages; queries planted on page (i*7919) % n_pages with unit noise of weight 0.5."""
    g = torch.Generator(device=dev).manual_seed(seed)
    P = torch.empty((n_pages, LP, D), dtype=torch.bfloat16, device=dev)
    for lo in range(0, n_pages, 2000):
        hi = min(lo + 2000, n_pages)
        P[lo:hi] = torch.nn.functional.normalize(torch.randn((hi - lo, LP, D), generator=g, device=dev), dim=-1).bfloat16()
    tgt = (torch.arange(n_q, device=dev) * 7919) % n_pages
    rows = torch.stack([torch.randperm(LP, generator=g, device=dev)[:LQ] for _ in range(n_q)])
    eps = torch.nn.functional.normalize(torch.randn((n_q, LQ, D), generator=g, device=dev), dim=-1)
    Q = torch.nn.functional.normalize(P[tgt[:, None], rows].float() + 0.5 * eps, dim=-1).bfloat16()
    return P, Q, tgt


def oracle_scores_on_device(Q, P, qm, pm, page_block=512, chunk_p=64):
    """EVERY score of (Q, P) by the oracle's own code (oracle/maxsim_oracle.py `maxsim_masked`, the restatement of
    /root/reference/evaluator/retrieval.py:166-213) run on DEVICE tensors: the same torch ops, fp32 rocBLAS matmul, in page blocks
    so that the (nq, chunk, lq, lp) intermediate stays a few GB.  1024 x 100 000 pairs take about half a minute."""
    nq, n = Q.shape[0], P.shape[0]
    out = torch.empty((nq, n), dtype=torch.float32, device=Q.device)
    for lo in range(0, n, page_block):
        hi = min(lo + page_block, n)
        out[:, lo:hi] = O.maxsim_masked(Q, P[lo:hi], qm, pm[lo:hi], chunk_p=chunk_p)
    return out


def topk_rows_on_device(scores, k):
    """oracle.topk_rows (score descending, index ascending on ties) on the device: a stable sort of -score."""
    order = torch.sort(-scores, dim=1, stable=True).indices[:, :k]
    return scores.gather(1, order), order.to(torch.int32)


def assert_full_matrix(got, want, Q, P, qm, pm, tol=1e-4, what=""):
    """All |got - want| <= tol; on failure name the pairs: (q, p), both scores, and the pair's per-token maxima by the oracle."""
    err = (got - want).abs()
    err = torch.where(torch.isfinite(err), err, torch.full_like(err, float("inf")))
    worst = err.max().item()
    if worst <= tol:
        return worst
    bad = (err > tol).nonzero()
    lines = []
    for q, p_ in bad[:6].tolist():
        sc, arg = O.maxsim_masked_argmax(Q[q:q + 1], P[p_:p_ + 1], qm[q:q + 1], pm[p_:p_ + 1])
        sim = torch.einsum("nd,md->nm", Q[q].float(), P[p_].float())
        tokmax = sim.masked_fill(~pm[p_].bool()[None, :], -1e4).amax(dim=1)
        lines.append(f"(q={q}, p={p_}) kernel {got[q, p_].item():.7f} oracle {want[q, p_].item():.7f} delta {got[q, p_].item() - want[q, p_].item():+.3e}; "
                     f"oracle per-token maxima {[round(v, 6) for v in tokmax.tolist()]} at patches {arg[0, 0].tolist()}")
    raise AssertionError(f"{what}: {len(bad)} of {got.numel()} scores off by more than {tol} (worst {worst:.3e}):\n" + "\n".join(lines))


def assert_topk_where_the_gap_allows(ti, want, k=100, gap=2e-4):
    """top-k indices identical wherever the ORACLE's own ranking gap to both neighbours exceeds `gap` (SURVEY 8(d) parity gate)."""
    ws, wi = topk_rows_on_device(want, k + 1)
    gap_ok = (ws[:, :-1] - ws[:, 1:]) > gap                                 # (nq, k): gap below rank j
    safe = gap_ok & torch.cat([gap_ok[:, :1], gap_ok[:, :-1]], 1)           # ... and above it (rank 0 has none above)
    assert torch.equal(ti[safe], wi[:, :k][safe]), f"{int((ti[safe] != wi[:, :k][safe]).sum())} top-{k} indices differ where the gap exceeds {gap}"
    return int(safe.sum()), safe.numel()


@pytest.fixture(scope="module")
def corpus100k(dev):
    """configs[3]: 100 000 pages x 1030 patches bf16 (26.4 GB resident) + 1024 planted queries, shared by the two tests below."""
    from evdr_amd.corpus import PageCorpus
    P, Q, tgt = synth(100000, 1024, dev, seed=13)
    corpus = PageCorpus.from_tensor(P)
    yield P, Q, tgt, corpus
    del corpus, P
    torch.cuda.empty_cache()


def test_config1_docvqa_shape_vs_oracle(dev):
    """configs[1]: ~500 pages x 1030 patches, bf16.  64 queries x 500 pages against the oracle: scores, top-100
    indices (where the ranking gap exceeds twice the score tolerance) and nDCG@5."""
    import evdr_amd.ops as ops
    from evdr_amd.evaluator.retrieval import score_multi_vector_masked, CustomRetrievalEvaluator
    from evdr_amd.evaluator.metrics import results_from_topk
    P, Q, tgt = synth(500, 64, dev, seed=11)
    gen = torch.Generator().manual_seed(1)
    pm = torch.ones(500, LP, dtype=torch.bool)
    lens = torch.randint(700, LP + 1, (500,), generator=gen)             # ragged valid lengths
    pm[torch.arange(LP)[None, :] >= lens[:, None]] = False
    qm = torch.ones(64, LQ, dtype=torch.bool)
    qm[:, 20:] = torch.rand(64, 12, generator=gen) > 0.5
    got = score_multi_vector_masked(Q, P, qm.to(dev), pm.to(dev))
    torch.set_num_threads(16)
    want = O.maxsim_masked(Q.float().cpu(), P.float().cpu(), qm, pm, chunk_p=64)
    assert (got.cpu() - want).abs().max().item() < 1e-4
    ts, ti = ops.topk(got, 100)
    ws, wi = O.topk_rows(want, 100)
    gap_ok = (ws[:, :-1] - ws[:, 1:]) > 2e-4
    safe = torch.cat([gap_ok, gap_ok[:, -1:]], 1) & torch.cat([gap_ok[:, :1], gap_ok], 1)
    assert torch.equal(ti.cpu()[safe], wi[safe])
    docids = [f"doc{j}" for j in range(500)]
    qrels = {str(i): {docids[int(t)]: 1} for i, t in enumerate(tgt.tolist())}
    ev = CustomRetrievalEvaluator()
    m_gpu = ev.compute_mteb_metrics(qrels, results_from_topk(ts.cpu().numpy(), ti.cpu().numpy(), range(64), docids))
    m_cpu = ev.compute_mteb_metrics(qrels, {str(i): {docids[j]: float(want[i, j]) for j in range(500)} for i in range(64)})
    assert abs(m_gpu["NDCG"]["NDCG@5"] - m_cpu["NDCG"]["NDCG@5"]) <= 1e-4
    assert abs(m_gpu["Recall"]["Recall@1"] - m_cpu["Recall"]["Recall@1"]) <= 1e-4


def test_config2_full_vidore_size_properties(dev):
    """configs[2]: the 10-subset corpus size (6 847 pages), 256 queries."""
    import evdr_amd.ops as ops
    from evdr_amd.corpus import PageCorpus, shard_range, pack_candidates, unpack_candidates, merge_candidates
    n, nq, k = 6847, 256, 100
    P, Q, tgt = synth(n, nq, dev, seed=12)
    corpus = PageCorpus.from_tensor(P)
    s1 = corpus.score(Q).clone()
    s2 = corpus.score(Q)
    assert torch.equal(s1, s2)                                           # idempotent, deterministic
    assert torch.equal(s1.argmax(dim=1), tgt)                            # planted page is rank 1 for every query
    # patches inside a page are a set: permuting them leaves every score bit-exact
    perm = torch.randperm(LP, device=dev)
    sp = PageCorpus.from_tensor(P[:512][:, perm].contiguous()).score(Q)
    assert torch.equal(sp, s1[:, :512])
    # token additivity: disjoint query-token masks add up
    ma = torch.zeros(nq, LQ, dtype=torch.bool, device=dev)
    ma[:, ::2] = True
    sa = corpus.score(Q, ma)
    sb = corpus.score(Q, ~ma)
    add = (sa + sb - s1).abs()
    if add.max().item() >= 2e-5:       # seen ONCE (round 3, one box, 8.7e-3) and never reproduced in 450 k stress launches: say where
        again = [(corpus.score(Q) != s1).sum().item(), (corpus.score(Q, ma) != sa).sum().item(), (corpus.score(Q, ~ma) != sb).sum().item()]
        bad = (add >= 2e-5).nonzero()
        raise AssertionError(f"token additivity: {len(bad)} entries off, max {add.max().item():.3e}, first (q, p) {bad[:8].tolist()}; "
                             f"entries that differ when s1 / sa / sb are recomputed: {again}")
    # EVERY one of the 1 752 832 scores against the oracle's own code run on the device (round 5; rounds 1-4 sampled 768 of them)
    qm1 = torch.ones(nq, LQ, dtype=torch.bool, device=dev)
    pm1 = torch.ones(n, LP, dtype=torch.bool, device=dev)
    want_dev = oracle_scores_on_device(Q, P, qm1, pm1)
    assert_full_matrix(s1, want_dev, Q, P, qm1, pm1, what="configs[2] bf16, 256 x 6847")
    assert_full_matrix(sa, oracle_scores_on_device(Q, P, ma, pm1), Q, P, ma, pm1, what="configs[2] bf16, even query tokens")
    # and a random sample of pages against the oracle on the host CPU (ties the device run of the oracle to the host run)
    cols = torch.randperm(n)[:48]
    want = O.maxsim_masked(Q[:16].float().cpu(), P[cols.to(dev)].float().cpu(), torch.ones(16, LQ, dtype=torch.bool),
                           torch.ones(48, LP, dtype=torch.bool))
    assert (s1[:16][:, cols.to(dev)].cpu() - want).abs().max().item() < 1e-4
    assert (want_dev[:16][:, cols.to(dev)].cpu() - want).abs().max().item() < 1e-5
    # top-k: sorted, consistent with the score matrix, and shard+merge == single shard (bit-exact)
    ts, ti = corpus.topk(Q, None, k)
    assert torch.all(ts[:, :-1] >= ts[:, 1:])
    assert torch.equal(s1.gather(1, ti.long()), ts)
    ws, wi = O.topk_rows(s1.cpu(), k)
    assert torch.equal(ti.cpu(), wi)
    assert_topk_where_the_gap_allows(ti, want_dev, k)                    # against the ORACLE's ranking of the oracle's scores
    corpus.score_events = []                                             # bench.py's bracketed form of the same step: same bits
    ts_b, ti_b = corpus.topk(Q, None, k)
    assert len(corpus.score_events) == 1 and torch.equal(ts_b, ts) and torch.equal(ti_b, ti)
    torch.cuda.synchronize()
    assert corpus.score_events[0][0].elapsed_time(corpus.score_events[0][1]) > 0
    corpus.score_events = None
    # the one-call C entry (evdr_maxsim_topk: the same two launches behind one call, what a C host binds): same bits
    from evdr_amd import _lib as L
    lib = L.load()
    ws = torch.empty(lib.evdr_maxsim_topk_workspace(nq, n), dtype=torch.uint8, device=dev)
    ts_1 = torch.empty((nq, k), dtype=torch.float32, device=dev)
    ti_1 = torch.empty((nq, k), dtype=torch.int32, device=dev)
    qp = Q.contiguous()[None]
    L.check(lib.evdr_maxsim_topk(qp.data_ptr(), corpus.planes.data_ptr(), None, corpus.tilemask.data_ptr(), corpus.pageflags.data_ptr(),
                                 nq, LQ, n, LP, 1, corpus.p_stride, corpus.p_plane_stride, None, None, corpus.idx_base, k,
                                 ts_1.data_ptr(), ti_1.data_ptr(), ws.data_ptr(), ws.numel(), L.current_stream_handle(dev)))
    assert torch.equal(ts_1, ts) and torch.equal(ti_1, ti)
    msgs = []
    for r in range(3):
        lo, hi = shard_range(n, r, 3)
        msgs.append(pack_candidates(*PageCorpus.from_tensor(P[lo:hi], None, idx_base=lo).topk(Q, None, k)))
    ms, mi = merge_candidates(*unpack_candidates(torch.stack(msgs)), k)
    assert torch.equal(mi, ti) and torch.equal(ms, ts)


def test_config3_100k_pages_properties(dev, corpus100k):
    """configs[3] at its full 100 000-page size on one GPU (26.4 GB resident), 64 queries: properties, the 8-way shard / merge
    replayed bit-exactly, and ALL 6 400 000 scores against the oracle's own code run on the device."""
    from evdr_amd.corpus import shard_range, pack_candidates, unpack_candidates, merge_candidates
    n, nq, k = 100000, 64, 100
    P, Q, tgt, corpus = corpus100k
    Q, tgt = Q[:nq].contiguous(), tgt[:nq]
    ts, ti = corpus.topk(Q, None, k)
    assert torch.equal(ti[:, 0].long(), tgt)                             # planted page retrieved at rank 1
    assert torch.all(ts[:, :-1] >= ts[:, 1:])
    s = corpus.score(Q)
    assert torch.equal(s.gather(1, ti.long()), ts)
    cols = torch.randperm(n)[:32]
    want = O.maxsim_masked(Q[:8].float().cpu(), P[cols.to(dev)].float().cpu(), torch.ones(8, LQ, dtype=torch.bool),
                           torch.ones(32, LP, dtype=torch.bool))
    assert (s[:8][:, cols.to(dev)].cpu() - want).abs().max().item() < 1e-4
    qm1 = torch.ones(nq, LQ, dtype=torch.bool, device=dev)
    pm1 = torch.ones(n, LP, dtype=torch.bool, device=dev)
    want_dev = oracle_scores_on_device(Q, P, qm1, pm1, page_block=2048)
    assert_full_matrix(s, want_dev, Q, P, qm1, pm1, what="configs[3] 64 x 100 000")
    assert_topk_where_the_gap_allows(ti, want_dev, k)
    msgs = []
    for r in range(8):                                                  # the 8-GPU sharding, replayed on one GPU
        lo, hi = shard_range(n, r, 8)
        shard = corpus.shard(lo, hi)
        msgs.append(pack_candidates(*shard.topk(Q, None, k)))
    ms, mi = merge_candidates(*unpack_candidates(torch.stack(msgs)), k)
    assert torch.equal(mi, ti) and torch.equal(ms, ts)


def test_config3_bench_size_every_score_against_the_oracle(dev, corpus100k):
    """configs[3] at bench.py's OWN size -- 1024 queries x 100 000 pages, the headline kernel instance and launch shape -- with every
    one of the 102 400 000 scores compared with the oracle's torch code run on the device (fp32 rocBLAS; about half a minute), the
    top-100 of the one-call retrieval path identical wherever the oracle's gap exceeds 2e-4, and the planted page at rank 1."""
    from evdr_amd import _lib as L
    n, nq, k = 100000, 1024, 100
    P, Q, tgt, corpus = corpus100k
    s = corpus.score(Q)
    assert L.load().evdr_last_fwd_kernel().decode().startswith("maxsim_fwd16s_kernel<4,1,false,8,2,")      # the bench's instance
    qm1 = torch.ones(nq, LQ, dtype=torch.bool, device=dev)
    pm1 = torch.ones(n, LP, dtype=torch.bool, device=dev)
    want_dev = oracle_scores_on_device(Q, P, qm1, pm1, page_block=256)
    worst = assert_full_matrix(s, want_dev, Q, P, qm1, pm1, what="configs[3] 1024 x 100 000 (bench size)")
    ts, ti = corpus.topk(Q, None, k)
    assert torch.equal(ti[:, 0].long(), tgt)
    assert torch.equal(s.gather(1, ti.long()), ts)
    safe, total = assert_topk_where_the_gap_allows(ti, want_dev, k)
    print(f"[fullsize] 1024 x 100000: max |kernel - oracle| = {worst:.3e}; top-{k} indices checked at {safe} of {total} ranks")


def test_rccl_exchange_single_rank(dev):
    """The candidate all-gather through the nccl (= RCCL) backend with device tensors; one rank is all a 1-GPU box has."""
    import torch.distributed as dist
    from evdr_amd.corpus import gather_candidates, merge_candidates
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(29600 + os.getpid() % 300))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        g = torch.Generator().manual_seed(2)
        sc = torch.randn(20, 100, generator=g).sort(dim=1, descending=True).values.to(dev)
        ix = torch.randint(0, 10000, (20, 100), generator=g, dtype=torch.int32).to(dev)
        s2, i2 = gather_candidates(sc, ix)
        assert torch.equal(s2, sc) and torch.equal(i2, ix)
        ms, mi = merge_candidates(s2, i2, 100)
        assert torch.equal(ms, sc)
        dist.barrier(device_ids=[dev.index or 0])                        # the barrier form bench.py uses under nccl
        # the training-side exchange (score columns) and one page-sharded fused step through RCCL
        import golden_recipes as R
        from evdr_amd import driver
        from evdr_amd.utils.preprocess_data import l2_normalize
        blk = torch.randn(8, 37, generator=g).to(dev)
        assert torch.equal(driver.gather_columns(blk, (37,)), blk)
        Qb, qmb, Pt, pmt, Pbar0, pms, hp = R.train_case("b4n8")
        teacher = driver.TeacherScorer(l2_normalize(Pt * pmt.unsqueeze(-1)).to(dev), pmt.to(dev))
        a = driver.FusedStudent(Pbar0.to(dev), pms.to(dev), lr=hp["lr"], weight_decay=hp["wd"])
        b = driver.FusedStudent(Pbar0.to(dev), pms.to(dev), lr=hp["lr"], weight_decay=hp["wd"])
        la = driver.sharded_fused_train_one_step(Qb, qmb, teacher, a, hp["temp"], (Pt.shape[0],))
        lb = driver.fused_train_one_step(Qb, qmb, teacher, b, hp["temp"])
        assert abs(la - lb) <= 1e-6 * abs(lb) and torch.allclose(a.x, b.x, atol=1e-6)
        # tie completion of the sharded retriever through RCCL's all-reduce / object gather at world = 1: duplicate pages
        from evdr_amd import ops
        from evdr_amd.corpus import PageCorpus, ShardedRetriever
        base = torch.nn.functional.normalize(torch.randn(90, 40, 128, generator=g), dim=-1).bfloat16()
        P = torch.cat([base, base[:1].repeat(60, 1, 1)]).to(dev)           # page 0 sixty-one times: ties across rank k = 50
        Qd = base[:7, :16].contiguous().to(dev)
        corpus = PageCorpus.from_tensor(P, None, idx_base=1000)
        ts, ti, extra = ShardedRetriever(corpus).search(Qd, None, 50, with_ties=True)
        ws, wi, wextra = ops.topk_with_ties(corpus.score(Qd, None), 50)
        assert torch.equal(ts, ws) and torch.equal(ti, wi + 1000) and extra.keys() == wextra.keys() and len(extra) >= 1
        for r in extra:
            assert np.array_equal(extra[r][0], wextra[r][0] + 1000) and np.array_equal(extra[r][1], wextra[r][1])
        ps, pi = ShardedRetriever(corpus).search(Qd, None, 50)
        assert torch.equal(ps, ts) and torch.equal(pi, ti)
    finally:
        dist.destroy_process_group()


def test_corpus_from_npz_payload(dev, tmp_path):
    """Feature-dump object arrays -> resident corpus (bf16 and fp32) == the reference's preprocess_docs + l2_normalize path."""
    import golden_recipes as R
    from evdr_amd.utils import preprocess_data as PD
    docs, attn, img, queries, qattn, docid = R.npz_payload_case()
    P_raw, pmask, _ = PD.preprocess_docs(docs, attn, img, device="cpu")
    Pn = O.l2_normalize(P_raw * pmask.unsqueeze(-1))
    Q, qm = PD.preprocess_queries(queries, qattn, device="cpu")
    want = O.maxsim_masked(Q, Pn, qm, pmask)
    c32, pm32 = PD.corpus_from_payload(docs, attn, img, dev, dtype=torch.float32)
    assert torch.equal(pm32.cpu(), pmask)
    got = c32.score(Q.to(dev), qm.to(dev))
    assert (got.cpu() - want).abs().max().item() < 1e-5
    c16, _ = PD.corpus_from_payload(docs, attn, img, dev, dtype=torch.bfloat16)
    got16 = c16.score(Q.bfloat16().to(dev), qm.to(dev))
    want16 = O.maxsim_masked(Q.bfloat16().float(), Pn.bfloat16().float(), qm, pmask)
    assert (got16.cpu() - want16).abs().max().item() < 1e-4


def test_config2_fp32_inputs_properties(dev):
    """configs[2] at the reference's own dtype: 6847 pages x 1030 patches of genuinely fp32 (not bf16-representable)
    embeddings through the drop-in scorer (fp16 hi/lo planes).  Sampled columns against an fp64 computation, the cached
    prepared planes against a fresh preparation, row/column permutation equivariance, and bf16 rounding as a sanity
    bound on how far a wrong plane product would be."""
    from evdr_amd.evaluator import retrieval as ER
    g = torch.Generator(device=dev).manual_seed(23)
    n, nq = 6847, 256
    P = torch.nn.functional.normalize(torch.randn((n, LP, D), generator=g, device=dev), dim=-1)
    tgt = (torch.arange(nq, device=dev) * 7919) % n
    eps = torch.nn.functional.normalize(torch.randn((nq, LQ, D), generator=g, device=dev), dim=-1)        # unit-norm noise
    Q = torch.nn.functional.normalize(P[tgt, :LQ] + 0.3 * eps, dim=-1)
    qm = torch.ones(nq, LQ, dtype=torch.bool, device=dev)
    pm = torch.ones(n, LP, dtype=torch.bool, device=dev)
    pm[::7, 900:] = False
    ER.forget_prepared()
    s = ER.score_multi_vector_masked(Q, P, qm, pm)
    assert torch.equal(s.argmax(dim=1), tgt)                                   # planted page on top
    cols = torch.randperm(n, generator=torch.Generator().manual_seed(3))[:24].to(dev)
    sim = torch.einsum("qnd,pmd->qpnm", Q.double(), P[cols].double()).masked_fill(~pm[cols][None, :, None, :], -1e4)
    want = sim.amax(-1).sum(-1)
    assert (s[:, cols].double() - want).abs().max().item() < 5e-6              # fp32-level accuracy, 20x inside the 1e-4 gate
    # EVERY score against the oracle's own fp32 code on the device (fp32 rocBLAS), and the top-100 where its gap allows
    from evdr_amd import ops
    want_dev = oracle_scores_on_device(Q, P, qm, pm)
    assert (want_dev[:, cols].double() - want).abs().max().item() < 2e-5       # the device run of the oracle itself, against fp64
    assert_full_matrix(s, want_dev, Q, P, qm, pm, what="configs[2] fp32 inputs, 256 x 6847")
    assert_topk_where_the_gap_allows(ops.topk(s, 100)[1], want_dev, 100)
    s2 = ER.score_multi_vector_masked(Q, P, qm, pm)                            # second call: cached planes
    assert torch.equal(s, s2)
    perm_q = torch.randperm(nq, device=dev)
    perm_p = torch.randperm(n, device=dev)
    s3 = ER.score_multi_vector_masked(Q[perm_q], P[perm_p], qm[perm_q], pm[perm_p])
    assert torch.equal(s3, s[perm_q][:, perm_p])                               # same per-tensor scale, same arithmetic per pair
    sb = ER.score_multi_vector_masked(Q.bfloat16(), P.bfloat16(), qm, pm)
    assert 1e-4 < (sb - s).abs().max().item() < 0.1                            # bf16 rounding is visible, fp32 path is not that
    ER.forget_prepared()


def test_kernel_rate_floors(dev):
    """Guard rails, not measurements: floors far below what every box of the pool delivered in round 1 (bf16 1735-1850
    TFLOP/s, fp32 1570-1620, single-query streaming 5.6-6.3 TB/s), so that a change which throws a hot kernel off its
    schedule (register spills in the straight-line block, a drained ring) fails loudly instead of costing 30 % silently."""
    import evdr_amd.ops as ops
    from evdr_amd.corpus import PageCorpus

    def best_ms(fn, reps=3):
        fn()
        torch.cuda.synchronize()
        out = []
        for _ in range(reps):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            fn()
            b.record()
            torch.cuda.synchronize()
            out.append(a.elapsed_time(b))
        return min(out)

    P, Q, _ = synth(12000, 1024, dev, seed=21)
    corpus = PageCorpus.from_tensor(P)
    out = torch.empty((1024, 12000), dtype=torch.float32, device=dev)
    ms = best_ms(lambda: corpus.score(Q, None, out=out))
    tf = 1024 * 12000 * 2 * LQ * LP * D / ms / 1e9
    assert tf > 1300, f"bf16 MaxSim kernel at {tf:.0f} TFLOP/s"
    ms1 = best_ms(lambda: corpus.score(Q[:1], None, out=out[:1]))
    tbs = 12000 * LP * D * 2 / ms1 / 1e9
    assert tbs > 3.5, f"single-query corpus streaming at {tbs:.2f} TB/s"
    # the few-queries regime (two 4-wave workgroups per CU, 1 / 2 / 3 queries per wave): round 2 measured 1.57 / 2.27 / 2.78 ms
    # at 40 k pages for 4 / 8 / 12 queries, i.e. 0.47 / 0.68 / 0.83 ms at 12 k pages
    for nq_small, limit_ms in ((4, 0.80), (8, 1.10), (12, 1.35)):
        msn = best_ms(lambda: corpus.score(Q[:nq_small], None, out=out[:nq_small]), reps=5)
        assert msn < limit_ms, f"{nq_small} queries x 12000 pages took {msn:.3f} ms"
    # ragged corpus (valid lengths 600-1030 + 4 masked in front): within ~5 % of the all-valid rate on valid patches
    gm = torch.Generator(device=dev).manual_seed(23)
    lens = torch.randint(600, LP + 1, (12000,), generator=gm, device=dev)
    pm = torch.arange(LP, device=dev)[None, :] < lens[:, None]
    pm[:, :4] = False
    cr = PageCorpus.from_tensor(P, pm)
    msr = best_ms(lambda: cr.score(Q, None, out=out))
    tfr = 1024 * float(pm.sum()) * 2 * LQ * D / msr / 1e9
    assert tfr > 1250, f"ragged + masked-prefix corpus at {tfr:.0f} TFLOP/s on valid patches"
    g = torch.Generator(device=dev).manual_seed(22)
    P32 = torch.nn.functional.normalize(torch.randn((2000, LP, D), generator=g, device=dev), dim=-1)
    Q32 = torch.nn.functional.normalize(torch.randn((512, LQ, D), generator=g, device=dev), dim=-1)
    c32 = PageCorpus.from_tensor(P32)
    o32 = torch.empty((512, 2000), dtype=torch.float32, device=dev)
    qp, qa = ops.split_f32(Q32)
    ms32 = best_ms(lambda: ops.maxsim_forward_prepared(qp, qa, c32.planes, c32.amax, None, c32.tilemask, c32.pageflags, out=o32))
    tf32 = 512 * 2000 * 2 * LQ * LP * D * 3 / ms32 / 1e9
    assert tf32 > 1000, f"fp32 (fp16 hi/lo) MaxSim kernel at {tf32:.0f} TFLOP/s of plane products"


def test_bench_forms_its_rccl_data_group_with_one_rank():
    """`bench.py --dist-at-one`: the N > 1 code path of the bench (gloo control group, RCCL data group formed beside it and agreed
    over the control plane, barrier on the data group, step-clock all-reduce, candidate all-gather + merge in the phase breakdown)
    with the one rank a 1-GPU box has -- the success path of the group formation, which the 2-rank rehearsals (gloo; RCCL refusing
    a shared GPU) do not reach."""
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--dist-at-one", "--pages", "3000", "--queries", "64", "--steps", "2",
                        "--warmup", "1", "--no-cpu-baseline", "--no-extras", "--no-other-regimes"], capture_output=True, text=True,
                       timeout=600, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    d = line["dist"]
    assert d["backend"] == "nccl" and d["control_backend"] == "gloo" and d["backend_fallback_reason"] is None and d["ranks_seen"] == 1
    assert set(line["phases"]["rank0"]) == {"score_ms", "topk_ms", "exchange_ms", "merge_ms"} and line["ndcg_at_5"] == 1.0
    assert line["device_errors_after_timed_region"] is None


def test_config1_full_size_against_the_reference_itself(dev, golden):
    """BASELINE.json configs[1] at its FULL size (500 queries x 500 pages x 1030 patches, ragged pages, masked query tails) against
    scores the REFERENCE's own `score_multi_vector_masked` produced for the same seeded inputs (tests/golden/config1_full.npz,
    made by tests/golden/make_golden_config1.py): every one of the 250 000 scores within 1e-4 on the bf16 kernel and on the fp32
    (fp16 hi/lo) kernel, top-100 indices identical wherever the reference's own ranking gap exceeds 2e-4, and nDCG@5 / Recall@1 of
    the device pipeline (`driver.eval_retrieval`) equal to the metric of the reference's all-pairs scores within 1e-4."""
    import golden_recipes as R
    import evdr_amd.ops as ops
    from evdr_amd import driver
    from evdr_amd.evaluator.retrieval import score_multi_vector_masked, CustomRetrievalEvaluator
    z = golden("config1_full")
    want = torch.from_numpy(z["scores"])
    Q, P, qm, pm, targets = R.config1_case()
    docmap = {str(j): f"doc{j}" for j in range(500)}
    qrels = {str(i): {docmap[str(int(t))]: 1} for i, t in enumerate(targets.tolist())}
    ev = CustomRetrievalEvaluator()
    m_ref = ev.compute_mteb_metrics(qrels, {str(i): {docmap[str(j)]: float(want[i, j]) for j in range(500)} for i in range(500)})
    ws, wi = O.topk_rows(want, 100)
    gap_ok = (ws[:, :-1] - ws[:, 1:]) > 2e-4
    safe = torch.cat([gap_ok, gap_ok[:, -1:]], 1) & torch.cat([gap_ok[:, :1], gap_ok], 1)
    for Qx, Px in ((Q.bfloat16(), P.bfloat16()), (Q, P)):                     # bf16 kernel; fp32 inputs -> fp16 hi/lo planes
        got = score_multi_vector_masked(Qx.to(dev), Px.to(dev), qm.to(dev), pm.to(dev))
        assert (got.cpu() - want).abs().max().item() < 1e-4
        ts, ti = ops.topk(got, 100)
        assert torch.equal(ti.cpu()[safe], wi[safe])
        m = driver.eval_retrieval(ev, Qx.to(dev), qm.to(dev), Px.to(dev), pm.to(dev), qrels, docmap, None, k=100)
        for fam, key in (("NDCG", "NDCG@5"), ("Recall", "Recall@1"), ("NDCG", "NDCG@10"), ("mRR", "MRR@10"), ("mAP", "MAP@100")):
            assert abs(m[fam][key] - m_ref[fam][key]) <= 1e-4, (key, m[fam][key], m_ref[fam][key])
