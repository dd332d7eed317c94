"""Host-side logic that needs no GPU: metrics, sharding arithmetic, candidate packing, input checks."""
import math

import numpy as np
import pytest
import torch

from oracle import maxsim_oracle as O


@pytest.fixture(scope="module")
def pkg():
    import evdr_amd  # noqa: F401
    import evdr_amd.evaluator.retrieval as er
    return er


def random_run(seed, nq=40, nd=120, multi_rel=False, ties=False):
    rng = np.random.default_rng(seed)
    docids = [f"doc{j}" for j in range(nd)]
    results, qrels = {}, {}
    for i in range(nq):
        sc = rng.normal(size=nd)
        if ties:
            sc = np.round(sc, 1)
        results[f"q{i}"] = {d: float(s) for d, s in zip(docids, sc)}
        rel = {docids[int(rng.integers(nd))]: 1}
        if multi_rel:
            for j in rng.integers(nd, size=3):
                rel[docids[int(j)]] = int(rng.integers(1, 4))
        qrels[f"q{i}"] = rel
    qrels["q_unscored"] = {"doc0": 1}
    return qrels, results


@pytest.mark.parametrize("multi_rel,ties", [(False, False), (True, False), (True, True)])
def test_metrics_match_oracle(pkg, multi_rel, ties):
    qrels, results = random_run(3, multi_rel=multi_rel, ties=ties)
    ev = pkg.CustomRetrievalEvaluator()
    got = ev.compute_mteb_metrics(qrels, results)
    want = O.trec_metrics(qrels, results, ev.k_values)
    assert set(got) == {"NDCG", "mAP", "Recall", "Precision", "mRR"}
    for fam in want:
        assert got[fam].keys() == want[fam].keys()
        for k in want[fam]:
            assert got[fam][k] == pytest.approx(want[fam][k], abs=1.1e-5), (fam, k)
    assert "NDCG@5" in got["NDCG"] and "Recall@1" in got["Recall"]          # keys read by the train scripts


def test_metrics_single_relevant_closed_form(pkg):
    """nDCG@5 = 1/log2(1+rank) if rank <= 5 else 0 -- independent of any library (parity unpinned vs mteb)."""
    rng = np.random.default_rng(0)
    nd = 50
    qrels, results, want = {}, {}, []
    for i in range(200):
        sc = rng.permutation(nd).astype(float)
        tgt = int(rng.integers(nd))
        rank = int((sc > sc[tgt]).sum()) + 1
        want.append(1 / math.log2(1 + rank) if rank <= 5 else 0.0)
        results[str(i)] = {f"d{j}": float(sc[j]) for j in range(nd)}
        qrels[str(i)] = {f"d{tgt}": 1}
    got = pkg.CustomRetrievalEvaluator().compute_mteb_metrics(qrels, results)
    assert got["NDCG"]["NDCG@5"] == round(float(np.mean(want)), 5)


def test_metrics_tie_rule_docid_descending(pkg):
    qrels = {"q": {"a": 1}}
    results = {"q": {"a": 1.0, "b": 1.0, "c": 0.5}}
    m = pkg.CustomRetrievalEvaluator(k_values=[1, 3]).compute_mteb_metrics(qrels, results)
    assert m["Recall"]["Recall@1"] == 0.0      # trec_eval ranks "b" above "a" on equal scores
    assert m["Recall"]["Recall@3"] == 1.0


def test_results_from_topk(pkg):
    from evdr_amd.evaluator.metrics import results_from_topk
    ts = np.array([[3.0, 2.0, -np.inf]], dtype=np.float32)
    ti = np.array([[2, 0, -1]], dtype=np.int32)
    r = results_from_topk(ts, ti, ["q7"], ["d0", "d1", "d2"])
    assert r == {"q7": {"d2": 3.0, "d0": 2.0}}


def test_shard_range_partitions():
    from evdr_amd.corpus import shard_range
    for n in (0, 1, 7, 100000, 100003):
        for w in (1, 2, 3, 8):
            spans = [shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert all(0 <= lo <= hi for lo, hi in spans)


def test_candidate_pack_roundtrip():
    from evdr_amd.corpus import pack_candidates, unpack_candidates
    g = torch.Generator().manual_seed(0)
    world, nq, k = 3, 5, 4
    sc = [torch.randn(nq, k, generator=g) for _ in range(world)]
    ix = [torch.randint(0, 1000, (nq, k), generator=g, dtype=torch.int32) for _ in range(world)]
    buf = torch.stack([pack_candidates(s, i) for s, i in zip(sc, ix)])
    s2, i2 = unpack_candidates(buf)
    assert torch.equal(s2, torch.cat(sc, dim=1)) and torch.equal(i2, torch.cat(ix, dim=1))


def test_list_scorers_errors_and_cpu_single_vector(pkg):
    import golden_recipes as R
    qs, ps = R.single_vector_case()
    s = pkg.BaseVisualRetrieverProcessor.score_single_vector(qs, ps, device="cpu")      # config 0: CPU plumbing
    assert s.dtype == torch.float32 and tuple(s.shape) == (7, 11)
    np.testing.assert_allclose(s.numpy(), O.dot_single_vector(qs, ps).numpy(), atol=1e-5)
    with pytest.raises(ValueError, match="No queries provided"):
        pkg.BaseVisualRetrieverProcessor.score_single_vector([], ps, device="cpu")
    with pytest.raises(ValueError, match="No passages provided"):
        pkg.BaseVisualRetrieverProcessor.score_single_vector(qs, [], device="cpu")
    with pytest.raises(ValueError, match="No queries provided"):
        pkg.BaseVisualRetrieverProcessor.score_multi_vector([], ps)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        pkg.BaseVisualRetrieverProcessor.score_multi_vector([torch.zeros(3, 128)], [torch.zeros(5, 128)], device="cpu")
    with pytest.raises(TypeError):
        pkg.BaseVisualRetrieverProcessor()          # abstract, like the reference


def test_masked_scorer_rejects_cpu_and_wrong_width(pkg):
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        pkg.score_multi_vector_masked(torch.zeros(2, 3, 128), torch.zeros(2, 5, 128), torch.ones(2, 3), torch.ones(2, 5))


def test_l2_normalize_matches_golden(golden):
    from evdr_amd.utils.preprocess_data import l2_normalize
    z = golden("a4_l2norm")
    x = torch.from_numpy(z["x"]).requires_grad_(True)
    y = l2_normalize(x)
    np.testing.assert_allclose(y.detach().numpy(), z["y"], atol=1e-7, rtol=1e-6)
    y.sum().backward()
    assert torch.isfinite(x.grad).all()             # zero rows: subgradient 0, not NaN


def test_secondary_losses_match_reference(golden):
    """criterion.py's other six losses (plain torch on the kernel's (B,N) outputs) vs the reference's values+grads."""
    import golden_recipes as R
    import evdr_amd.criterion as crit
    z = golden("losses")
    ss, st, labels = R.losses_case()
    cases = {
        "infonce_supervised_loss": lambda s: crit.infonce_supervised_loss(s, labels, temperature=0.07),
        "score_preserving_loss": lambda s: crit.score_preserving_loss(s, st),
        "pairwise_distillation_loss": lambda s: crit.pairwise_distillation_loss(s, st),
        "listwise_distillation_loss": lambda s: crit.listwise_distillation_loss(s, st, k=10, temperature=2.0),
        "lambda_loss": lambda s: crit.lambda_loss(s, st),
        "ranknce_loss": lambda s: crit.ranknce_loss(s, st, temperature=0.5, lambda_weight=0.7),
    }
    for name, fn in cases.items():
        sg = ss.clone().requires_grad_(True)
        val = fn(sg)
        val.backward()
        np.testing.assert_allclose(val.item(), float(z[name]), rtol=1e-5, err_msg=name)
        np.testing.assert_allclose(sg.grad.numpy(), z[name + "_grad"], atol=1e-6, rtol=1e-4, err_msg=name)


def test_npz_helpers_match_reference(golden, tmp_path):
    """utils/preprocess_data.py + utils/utils.py counterparts vs outputs of the reference's own functions."""
    import golden_recipes as R
    from evdr_amd.utils import preprocess_data as PD, utils as U
    z = golden("npz_helpers")
    docs, attn, img, queries, qattn, docid = R.npz_payload_case()
    P_raw, pmask, valid = PD.preprocess_docs(docs, attn, img, device="cpu")
    assert P_raw.dtype == torch.float32 and pmask.dtype == torch.bool
    assert np.array_equal(P_raw.numpy(), z["P_raw"]) and np.array_equal(pmask.numpy(), z["pmask"])
    assert np.array_equal(valid, z["valid"])
    _, pm2, _ = PD.preprocess_docs(docs, None, None, device="cpu")
    assert np.array_equal(pm2.numpy(), z["pmask_nomask"])
    Q, qm = PD.preprocess_queries(queries, qattn, device="cpu")
    np.testing.assert_allclose(Q.numpy(), z["Q"], atol=1e-7)
    assert np.array_equal(qm.numpy(), z["qmask"])
    objs = U.tokens_to_object(P_raw.numpy(), pmask.numpy())
    assert np.array_equal(np.array([o.shape[0] for o in objs]), z["obj_lens"])
    assert np.array_equal(np.concatenate(list(objs), axis=0), z["obj_concat"])
    perm = np.array([3, 0, 6, 1, 5, 2, 4])
    (docs_al,), ok = U.align_by_docid(docid, docid[perm], docs[perm])
    assert ok == bool(z["align_ok"]) and np.array_equal(np.stack([d[0] for d in docs_al]), z["align_first_rows"])
    assert U.align_by_docid(docid, docid[:3], docs[:3])[1] is False
    # writer -> loader round trip in the reference's schema
    path = tmp_path / "best_ndcg5.npz"
    U.save_compressed_npz(path, docid, objs, attn, img, meta={"dataset": "synthetic", "mf": 5, "step": 1})
    back = PD.load_init_payload(str(path))
    assert [str(x) for x in back["docid"]] == [str(x) for x in docid]
    assert all(np.array_equal(a, b) for a, b in zip(back["documents"], objs))
    assert PD.load_npz(str(path))["meta"].item()["mf"] == 5


def test_logger_line_contract(tmp_path):
    """train.log lines must stay parseable by the reference's summary_results.py regex (JSON object at line end)."""
    import json, re
    from evdr_amd.utils.utils import get_logger, log_json
    logger, tb = get_logger(tmp_path, use_tb=False)
    log_json(logger, {"summary/best_ndcg5": {"step": 500, "Recall@1": 0.71, "NDCG@5": 0.83}, "note": "training finished"})
    for h in logger.handlers:
        h.flush()
    line = (tmp_path / "train.log").read_text().strip().splitlines()[-1]
    assert re.match(r"^\[[^\]]+\]\[INFO\] \{", line)
    obj = json.loads(line[line.index("{"):])
    assert obj["summary/best_ndcg5"]["NDCG@5"] == 0.83


def test_metrics_vs_sklearn(pkg):
    """Independent third-party cross-check of the metric the reference takes from mteb/pytrec_eval (absent here, parity
    unpinned): scikit-learn's ndcg_score (linear gain, log2 discount -- trec_eval's ndcg_cut) and average_precision_score
    (trec_eval's map when the cut-off covers the whole ranking), on distinct scores with graded relevance."""
    sk = pytest.importorskip("sklearn.metrics")
    rng = np.random.default_rng(5)
    nd, nq = 40, 60
    docids = [f"doc{j}" for j in range(nd)]
    qrels, results = {}, {}
    want_ndcg = {k: [] for k in (1, 3, 5, 10)}
    want_map = []
    for i in range(nq):
        sc = rng.permutation(nd).astype(np.float64) + rng.random(nd) * 0.1        # distinct scores: no tie rule involved
        rel = np.zeros(nd, dtype=np.int64)
        rel[rng.choice(nd, size=int(rng.integers(1, 6)), replace=False)] = rng.integers(1, 4, size=1)[0]
        rel[int(rng.integers(nd))] = int(rng.integers(1, 4))
        qrels[f"q{i}"] = {docids[j]: int(rel[j]) for j in range(nd) if rel[j] > 0}
        results[f"q{i}"] = {docids[j]: float(sc[j]) for j in range(nd)}
        for k in want_ndcg:
            want_ndcg[k].append(sk.ndcg_score(rel[None, :], sc[None, :], k=k))
        want_map.append(sk.average_precision_score((rel > 0).astype(int), sc))
    got = pkg.CustomRetrievalEvaluator(k_values=[1, 3, 5, 10, 50]).compute_mteb_metrics(qrels, results)
    for k, vals in want_ndcg.items():
        assert got["NDCG"][f"NDCG@{k}"] == pytest.approx(float(np.mean(vals)), abs=1.1e-5), k
    assert got["mAP"]["MAP@50"] == pytest.approx(float(np.mean(want_map)), abs=1.1e-5)


def test_results_from_topk_uses_the_tie_completion():
    """Rows whose k-th score is tied beyond the device cut carry ALL candidates with a score >= the k-th one
    (ops.topk_with_ties); the metric then ranks them by trec_eval's rule (docid descending) like the reference's all-pairs
    dict does -- here: 6 pages, k = 2, three pages tied at the cut."""
    from evdr_amd.evaluator.metrics import evaluate, results_from_topk
    scores = np.array([[3.0, 1.0, 2.0, 2.0, 2.0, 0.5]], dtype=np.float32)
    docids = ["a", "b", "c", "d", "e", "f"]
    ts, ti = np.array([[3.0, 2.0]], dtype=np.float32), np.array([[0, 2]], dtype=np.int32)       # device order: index ascending
    extra = {0: (np.array([0, 2, 3, 4]), np.array([3.0, 2.0, 2.0, 2.0], dtype=np.float32))}
    qrels = {"q": {"e": 1}}                                                                     # trec_eval puts "e" at rank 2
    allpairs = {"q": {d: float(s) for d, s in zip(docids, scores[0])}}
    want = evaluate(qrels, allpairs, [1, 2])
    assert evaluate(qrels, results_from_topk(ts, ti, ["q"], docids, extra=extra), [1, 2]) == want
    assert want["Recall"]["Recall@2"] == 1.0
    assert evaluate(qrels, results_from_topk(ts, ti, ["q"], docids), [1, 2])["Recall"]["Recall@2"] == 0.0   # the bare cut misses it


# ---- the array metric path (evaluate_topk) == the dict entry (evaluate), bit for bit ------------------------------------
def _device_like_topk(scores: np.ndarray, k: int):
    """What evdr_topk + ops.topk_with_ties hand to the host, computed with numpy: top-k by (score desc with NaN first,
    index asc) and, for rows whose k-th score is tied beyond the cut, every column ranking at or above it."""
    nq, n = scores.shape
    key = np.where(np.isnan(scores), np.inf, scores.astype(np.float64))
    nanfirst = np.isnan(scores)
    order = np.lexsort((np.broadcast_to(np.arange(n), (nq, n)), -key, ~nanfirst), axis=-1)
    ti = order[:, :k].astype(np.int32)
    ts = np.take_along_axis(scores, order[:, :k], 1)
    extra = {}
    for r in range(nq):
        kth = ts[r, k - 1]
        cand = np.nonzero(np.isnan(scores[r]) | (scores[r] >= kth))[0] if not np.isnan(kth) else np.nonzero(np.isnan(scores[r]))[0]
        if len(cand) > k:
            extra[r] = (cand.astype(np.int64), scores[r, cand])
    return ts, ti, extra


@pytest.mark.parametrize("nq,n,k,quant,multi,holes,nan", [
    (50, 300, 100, 0, False, False, False),       # ViDoRe-like: one relevant page, no ties
    (50, 300, 100, 4, False, False, False),       # quantised scores: tie runs across the cut in almost every row
    (64, 400, 100, 3, True, False, False),        # multi-relevant qrels with graded and non-positive relevance
    (40, 90, 100, 2, True, False, False),         # fewer pages than the largest cut-off
    (30, 200, 100, 0, True, True, False),         # holes (idx -1) at the end of a short shard's list
    (30, 200, 100, 5, True, False, True),         # NaN scores (ranked first by the device, last by the metric's sort)
    (30, 200, 20, 5, True, False, False),         # device k below the largest cut-off
])
def test_evaluate_topk_equals_the_dict_entry(nq, n, k, quant, multi, holes, nan):
    from evdr_amd.evaluator.metrics import EvalIndex, evaluate, evaluate_topk, results_from_topk
    ks = [1, 3, 5, 10, 50, 70, 100]
    for seed in range(4):
        rng = np.random.default_rng(1000 * seed + nq + n)
        scores = rng.standard_normal((nq, n)).astype(np.float32)
        if quant:
            scores = (np.round(scores * quant) / quant).astype(np.float32)
            scores[scores == 0] = np.where(rng.random((scores == 0).sum()) < 0.5, -0.0, 0.0)      # -0.0 ties +0.0
        if nan:
            scores[rng.integers(0, nq, 6), rng.integers(0, n, 6)] = np.nan
        docids = [f"doc{rng.integers(0, 10 ** 6)}_{j}" for j in range(n)]                        # string order != page order
        qkeys = [f"q{i}" for i in range(nq)]
        qrels = {"unanswered": {"doc0": 1}}                                                       # counts in MRR's denominator
        for i in range(nq):
            if i % 7 == 3:
                continue                                                                          # a query without judgements
            m = int(rng.integers(1, 6)) if multi else 1
            qrels[qkeys[i]] = {docids[j]: (int(rng.integers(-1, 4)) if multi else 1) for j in rng.choice(n, m, replace=False)}
            if multi and i % 5 == 0:
                qrels[qkeys[i]]["not_in_corpus"] = 2
        ts, ti, extra = _device_like_topk(scores, min(k, n))
        if holes:
            ti[:, -3:], ts[:, -3:], extra = -1, -np.inf, {}
        want = evaluate(qrels, results_from_topk(ts, ti, qkeys, docids, extra=extra), ks)
        got = evaluate_topk(EvalIndex(qrels, qkeys, docids, ks), ts, ti, extra)
        assert got == want, (seed, got, want)
        if quant and not holes and n > k:
            assert extra                                                                          # the tie path really ran


def test_evaluate_topk_with_repeated_docids_goes_through_the_dict_entry():
    from evdr_amd.evaluator.metrics import EvalIndex, evaluate, evaluate_topk, results_from_topk
    ts = np.array([[3.0, 2.0, 1.0]], dtype=np.float32)
    ti = np.array([[0, 1, 2]], dtype=np.int32)
    docids, qkeys, qrels = ["a", "b", "a"], ["q"], {"q": {"a": 1}}
    idx = EvalIndex(qrels, qkeys, docids, [1, 3])
    assert not idx.usable
    assert evaluate_topk(idx, ts, ti) == evaluate(qrels, results_from_topk(ts, ti, qkeys, docids), [1, 3])


def test_fast_path_defaults_and_their_fallbacks():
    """`resolve_fast_paths`: no flag = fused step + teacher score cache; --no_* opts out; a cache beyond --teacher_cache_gb is not made
    and says so; another optimizer than AdamW takes the autograd step by default and refuses an explicit --fused_step."""
    import evdr_amd  # noqa: F401
    from evdr_amd import driver
    ap = driver.build_argparser()
    base = ["--datasets", "d", "--mapping_json", "m.json"]
    said = []
    assert driver.resolve_fast_paths(ap.parse_args(base), 25000, 500, said.append) == (True, True) and not said
    assert driver.resolve_fast_paths(ap.parse_args(base + ["--no_fused_step", "--no_cache_teacher_scores"]), 25000, 500, said.append) == (False, False)
    assert driver.resolve_fast_paths(ap.parse_args(base + ["--fused_step", "--cache_teacher_scores"]), 25000, 500, said.append) == (True, True)
    assert not said
    # 25 000 queries x 100 000 pages x 4 B = 9.3 GiB > the default 8 GiB budget: logged fallback, never silent
    assert driver.resolve_fast_paths(ap.parse_args(base), 25000, 100000, said.append) == (True, False)
    assert len(said) == 1 and "teacher score cache" in said[0] and "not made" in said[0]
    assert driver.resolve_fast_paths(ap.parse_args(base + ["--teacher_cache_gb", "16"]), 25000, 100000, said.append) == (True, True)
    said.clear()
    assert driver.resolve_fast_paths(ap.parse_args(base + ["--opt", "adam"]), 100, 10, said.append) == (False, True)
    assert len(said) == 1 and "AdamW only" in said[0]
    with pytest.raises(ValueError, match="AdamW only"):
        driver.resolve_fast_paths(ap.parse_args(base + ["--opt", "adam", "--fused_step"]), 100, 10, said.append)
