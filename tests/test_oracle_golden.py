"""Pin the CPU oracle against fixtures produced by the reference's own functions
(tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

import golden_recipes as R
from oracle import maxsim_oracle as O


def T(x):
    return torch.from_numpy(np.asarray(x))


@pytest.mark.parametrize("case", ["small_ragged", "lq1", "chunk_tail"])
def test_a1_small_forward_backward(golden, case):
    z = golden(f"a1_{case}")
    Q, P, qm, pm, g = R.small_case(case)
    # the recipe re-creates exactly what the reference was fed
    assert torch.equal(Q, T(z["Q"])) and torch.equal(P, T(z["P"]))
    assert torch.equal(qm, T(z["qmask"])) and torch.equal(pm, T(z["pmask"]))
    s = O.maxsim_masked(Q, P, qm, pm, chunk_p=R.SMALL_CHUNK[case])
    np.testing.assert_allclose(s.numpy(), z["scores"], atol=2e-6, rtol=0)
    s2, arg = O.maxsim_masked_argmax(Q, P, qm, pm)
    np.testing.assert_allclose(s2.numpy(), z["scores"], atol=2e-6, rtol=0)
    assert np.array_equal(arg.numpy().astype(np.int32), z["argmax"])
    dP = O.maxsim_backward(g, Q, P, qm, pm)
    np.testing.assert_allclose(dP.numpy(), z["dP"], atol=1e-6, rtol=0)
    # autograd through the oracle agrees too
    Pg = P.clone().requires_grad_(True)
    (O.maxsim_masked(Q, Pg, qm, pm) * g).sum().backward()
    np.testing.assert_allclose(Pg.grad.numpy(), z["dP"], atol=1e-6, rtol=0)


@pytest.mark.parametrize("d", [64, 200, 256])
def test_a1_other_embedding_widths(golden, d):
    """The oracle against the reference's own output at widths other than 128 (tests/golden/make_golden_width.py; the reference takes
    the width from its tensors, evaluator/retrieval.py:173): scores, arg-max, dP and dQ."""
    z = golden(f"a1_width{d}")
    Q, P, qm, pm, g = R.width_case(d)
    s, arg = O.maxsim_masked_argmax(Q, P, qm, pm)
    np.testing.assert_allclose(s.numpy(), z["scores"], atol=2e-6, rtol=0)
    assert np.array_equal(arg.numpy().astype(np.int32), z["argmax"])
    Qg, Pg = Q.clone().requires_grad_(True), P.clone().requires_grad_(True)
    (O.maxsim_masked(Qg, Pg, qm, pm) * g).sum().backward()
    np.testing.assert_allclose(Pg.grad.numpy(), z["dP"], atol=1e-6, rtol=0)
    np.testing.assert_allclose(Qg.grad.numpy(), z["dQ"], atol=1e-6, rtol=0)
    np.testing.assert_allclose(O.maxsim_backward(g, Q, P, qm, pm).numpy(), z["dP"], atol=1e-6, rtol=0)


def test_a1_known_answers(golden):
    z = golden("a1_small_ragged")
    Q, P, qm, pm, g = R.small_case("small_ragged")
    s = z["scores"]
    assert np.all(s[:, 2] == 0.0)                      # all-masked page -> exactly 0
    assert np.all(z["dP"][2] == 0.0)                   # and exactly zero gradient
    assert np.all(z["dP"][~pm.numpy()] == 0.0)         # masked positions never get gradient
    # duplicate patches 3 / 7 / 30 of page 1: only the first may be selected
    arg = z["argmax"][:, 1, :]
    assert not np.any(arg == 7) and not np.any(arg == 30)


@pytest.mark.parametrize("tag,bf16", [("f32", False), ("bf16", True)])
def test_a1_seeded_1030(golden, tag, bf16):
    z = golden("a1_seeded1030_" + tag)
    Q, P, qm, pm = R.seeded_1030(bf16_inputs=bf16)
    s = O.maxsim_masked(Q, P, qm, pm, chunk_p=64)
    np.testing.assert_allclose(s.numpy(), z["scores"], atol=5e-6, rtol=0)
    if not bf16:  # SURVEY §8(c) pin (1)
        np.testing.assert_allclose(z["scores"][0, :6], [5.7109, 5.7141, 5.6196, 0.0, 5.6686, 5.1231], atol=1e-4)


def test_a4_l2_normalize(golden):
    z = golden("a4_l2norm")
    y = O.l2_normalize(T(z["x"]))
    np.testing.assert_allclose(y.numpy(), z["y"], atol=1e-7, rtol=1e-6)
    assert np.all(z["y"][0, 0] == 0.0)


def test_a5_infonce(golden):
    z = golden("a5_infonce")
    ss, st = T(z["score_s"]), T(z["score_t"])
    loss = O.infonce_distill(ss, st, float(z["temp"]))
    np.testing.assert_allclose(loss.item(), float(z["loss"]), rtol=1e-6)
    np.testing.assert_allclose(O.infonce_distill_grad(ss, st, float(z["temp"])).numpy(), z["dscore"], atol=1e-7)


def test_a7_train_step_small(golden):
    z = golden("a7_step_b4n8")
    Qb, qmb, Pt, pmt, Pbar0, pms, hp = R.train_case("b4n8")
    Ptn = O.l2_normalize(Pt * pmt.unsqueeze(-1))
    loss, grad, after, sc_t, sc_s = O.distill_train_step(
        Qb, qmb, Ptn, pmt, Pbar0 * pms.unsqueeze(-1), pms, hp["temp"], hp["lr"], hp["wd"])
    np.testing.assert_allclose(sc_t.numpy(), z["sc_t"], atol=5e-6)
    np.testing.assert_allclose(sc_s.numpy(), z["sc_s"], atol=5e-6)
    np.testing.assert_allclose(loss, float(z["loss"]), rtol=1e-5)
    np.testing.assert_allclose(grad.numpy(), z["grad"], atol=1e-6)
    np.testing.assert_allclose(after.numpy(), z["param_after"], atol=1e-6)


def test_a3_single_vector(golden):
    z = golden("a3_single")
    qs, ps = R.single_vector_case()
    np.testing.assert_allclose(O.dot_single_vector(qs, ps).numpy(), z["scores"], atol=1e-5)
    with pytest.raises(ValueError):
        O.dot_single_vector([], ps)
    with pytest.raises(ValueError):
        O.dot_single_vector(qs, [])


def test_a2_unmasked_lists(golden):
    z = golden("a2_unmasked_lists")
    qs, ps = R.ragged_lists_case()
    np.testing.assert_allclose(O.maxsim_unmasked_lists(qs, ps, batch_size=4).numpy(), z["scores_bs4"], atol=2e-6)
    np.testing.assert_allclose(O.maxsim_unmasked_lists(qs, ps, batch_size=128).numpy(), z["scores_bs128"], atol=2e-6)
    # the zero-pad quirk: batch composition changes scores (padding rows join the max)
    assert not np.allclose(z["scores_bs4"], z["scores_bs128"])


def test_metrics_closed_form():
    """PARITY UNPINNED vs mteb; single-relevant closed form: nDCG@k = 1/log2(1+rank) if rank<=k."""
    import math
    qrels = {"q0": {"d3": 1}, "q1": {"d0": 1}, "q2": {"d9": 1}}
    results = {
        "q0": {f"d{i}": float(10 - i) for i in range(10)},            # d3 at rank 4
        "q1": {f"d{i}": float(10 - i) for i in range(10)},            # d0 at rank 1
        "q2": {f"d{i}": float(10 - i) for i in range(10)},            # d9 at rank 10
    }
    m = O.trec_metrics(qrels, results, [1, 3, 5, 10])
    assert m["NDCG"]["NDCG@5"] == round((1 / math.log2(5) + 1.0 + 0.0) / 3, 5)
    assert m["Recall"]["Recall@1"] == round(1 / 3, 5)
    assert m["NDCG"]["NDCG@10"] == round((1 / math.log2(5) + 1.0 + 1 / math.log2(11)) / 3, 5)
    assert m["mRR"]["MRR@10"] == round((1 / 4 + 1 + 1 / 10) / 3, 5)
    assert m["Precision"]["P@5"] == round((1 / 5 + 1 / 5 + 0) / 3, 5)
    assert m["mAP"]["MAP@5"] == round((1 / 4 + 1 + 0) / 3, 5)


def test_c_oracle_matches_reference_fixtures_and_torch_oracle(golden):
    """oracle/maxsim_oracle.c (plain C, double accumulation) against the fixtures the reference's own function produced,
    and against the torch restatement on a masked random case incl. an all-masked page and the argmax tie rule."""
    import ctypes
    import os
    import shutil
    import sys
    if shutil.which("gcc") is None:
        pytest.skip("gcc not available")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import __graft_entry__ as ge
    lib = ctypes.CDLL(ge.build_oracle_c())
    fn = lib.evdr_oracle_maxsim
    fn.restype = None
    fn.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_long] * 5 + [ctypes.c_void_p, ctypes.c_void_p]

    def run(Q, P, qm, pm, want_arg=False):
        Qc = np.ascontiguousarray(Q, dtype=np.float32)
        Pc = np.ascontiguousarray(P, dtype=np.float32)
        qmc = np.ascontiguousarray(qm, dtype=np.uint8)
        pmc = np.ascontiguousarray(pm, dtype=np.uint8)
        nq, lq, d = Qc.shape
        npg, lp, _ = Pc.shape
        out = np.empty((nq, npg), dtype=np.float64)
        arg = np.empty((nq, npg, lq), dtype=np.int32) if want_arg else None
        fn(Qc.ctypes.data, Pc.ctypes.data, qmc.ctypes.data, pmc.ctypes.data, nq, lq, npg, lp, d, out.ctypes.data,
           arg.ctypes.data if want_arg else None)
        return out, arg

    for case in ("small_ragged", "lq1", "chunk_tail"):
        z = golden(f"a1_{case}")
        out, arg = run(z["Q"], z["P"], z["qmask"], z["pmask"], want_arg=True)
        np.testing.assert_allclose(out, z["scores"], atol=2e-5)
        live = z["qmask"].astype(bool)[:, None, :] & z["pmask"].astype(bool).any(-1)[None, :, None]
        assert np.array_equal(arg[live], z["argmax"][live])
    gen = torch.Generator().manual_seed(4)
    Q = torch.nn.functional.normalize(torch.randn(5, 9, 128, generator=gen), dim=-1)
    P = torch.nn.functional.normalize(torch.randn(7, 33, 128, generator=gen), dim=-1)
    P[2, 5] = P[2, 1]                                              # exact tie: the first index wins
    qm = torch.rand(5, 9, generator=gen) > 0.2
    pm = torch.rand(7, 33, generator=gen) > 0.3
    pm[4] = False
    want, warg = O.maxsim_masked_argmax(Q, P, qm, pm)
    out, arg = run(Q.numpy(), P.numpy(), qm.numpy(), pm.numpy(), want_arg=True)
    np.testing.assert_allclose(out, want.numpy(), atol=2e-5)
    assert np.array_equal(arg, warg.numpy())
    assert np.all(out[:, 4] == 0.0)


def test_config1_full_size_slice_vs_reference(golden):
    """BASELINE.json configs[1] at full size, scored by the reference itself (tests/golden/make_golden_config1.py): the oracle on a
    slice of the same inputs (48 queries x 40 pages; the whole matrix is the GPU test's job) and the fixture's own sanity."""
    z = golden("config1_full")
    Q, P, qm, pm, targets = R.config1_case()
    assert z["scores"].shape == (500, 500) and np.array_equal(z["targets"], targets.numpy()) and float(z["rank1"]) == 1.0
    qs, ps = slice(100, 148), slice(230, 270)
    got = O.maxsim_masked(Q[qs], P[ps], qm[qs], pm[ps], chunk_p=16)
    np.testing.assert_allclose(got.numpy(), z["scores"][qs, ps], atol=1e-5, rtol=0)


def test_a7_eight_step_trajectory(golden):
    """Eight consecutive steps of the reference's own train_one_step (tests/golden/make_golden_trajectory.py): the oracle's
    restatement, driven the same way (one optimizer over all steps), reproduces every loss, the parameters after steps 1 / 4 / 8
    and AdamW's moments."""
    z = golden("a7_trajectory")
    batches, Pt, pmt, Pbar0, pms, hp = R.trajectory_case()
    Ptn = O.l2_normalize(Pt * pmt.unsqueeze(-1)).detach()
    param = torch.nn.Parameter(Pbar0 * pms.unsqueeze(-1))
    opt = torch.optim.AdamW([param], lr=hp["lr"], weight_decay=hp["wd"])
    big = torch.from_numpy(z["big"])
    for i, (Qb, qmb) in enumerate(batches, 1):
        Ps = O.l2_normalize(param * pms.unsqueeze(-1))
        with torch.no_grad():
            sc_t = O.maxsim_masked(Qb, Ptn, qmb, pmt, 64)
        loss = O.infonce_distill(O.maxsim_masked(Qb, Ps, qmb, pms, 64), sc_t, hp["temp"])
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        np.testing.assert_allclose(loss.item(), z["losses"][i - 1], rtol=1e-5)
        if i in (1, 4, 8):
            d = (param.detach() - torch.from_numpy(z[f"param_after_{i}"])).abs()
            assert d[big].max().item() < 1e-6 and d.max().item() < 2 * i * hp["lr"]
    st = opt.state[param]
    assert float(st["step"]) == float(z["step"]) == 8.0
    np.testing.assert_allclose(st["exp_avg"].numpy(), z["exp_avg"], atol=1e-7)
    np.testing.assert_allclose(st["exp_avg_sq"].numpy(), z["exp_avg_sq"], atol=1e-10, rtol=1e-4)
