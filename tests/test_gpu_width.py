"""Embedding widths other than 128.  The reference takes the width from its tensors
(/root/reference/evaluator/retrieval.py:173); the kernels score d <= 128 on one 128-column block (zero columns appended: exact) and
128 < d <= 256 on two blocks of fp16 hi/lo planes (`maxsim_fwd16s_kernel<...>x2cols`, four planes, six plane products per k-step
into one accumulator chain).  Gates as for width 128: scores |d| <= 1e-4 against the reference's own output (tests/golden/
a1_width*.npz, made by tests/golden/make_golden_width.py), arg-max identical wherever the reference's own top-2 gap exceeds 1e-6,
dP / dQ atol 1e-6; beyond the fixtures the oracle is the checker."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import golden_recipes as R  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ER():
    import evdr_amd  # noqa: F401
    from evdr_amd.evaluator import retrieval
    return retrieval


@pytest.fixture(scope="module")
def O():
    from oracle import maxsim_oracle
    return maxsim_oracle


def _sim(Q, P, pm):
    return torch.einsum("qnd,cmd->qcnm", Q, P).masked_fill(~pm[None, :, None, :], -1e4)


@pytest.mark.parametrize("d", [64, 128, 200, 256])
def test_a1_scores_gradients_and_argmax_at_width(golden, ER, O, d):
    from evdr_amd import ops
    dev = torch.device("cuda:0")
    Q, P, qm, pm, g = R.width_case(d)
    if d == 128:                                            # no fixture of the reference needed: the oracle (pinned at 128) is the checker
        Qo, Po = Q.clone().requires_grad_(True), P.clone().requires_grad_(True)
        so = O.maxsim_masked(Qo, Po, qm, pm)
        (so * g).sum().backward()
        want = {"scores": so.detach().numpy(), "dP": Po.grad.numpy(), "dQ": Qo.grad.numpy(),
                "argmax": _sim(Q, P, pm).max(dim=-1).indices.to(torch.int32).numpy()}
    else:
        want = golden(f"a1_width{d}")
    Qd, Pd = Q.to(dev).requires_grad_(True), P.to(dev).requires_grad_(True)
    s = ER.score_multi_vector_masked(Qd, Pd, qm.to(dev), pm.to(dev))
    (s * g.to(dev)).sum().backward()
    np.testing.assert_allclose(s.detach().cpu().numpy(), want["scores"], atol=1e-4, rtol=0)
    assert Pd.grad.shape == P.shape and Qd.grad.shape == Q.shape                     # gradients come back in the caller's width
    np.testing.assert_allclose(Pd.grad.cpu().numpy(), want["dP"], atol=1e-6, rtol=0)
    np.testing.assert_allclose(Qd.grad.cpu().numpy(), want["dQ"], atol=1e-6, rtol=0)
    _, arg = ops.maxsim_forward(Q.to(dev), P.to(dev), qm.to(dev), pm.to(dev), want_argmax=True)
    arg = arg.cpu().long() & 0xFFFF
    sim = _sim(Q, P, pm)
    top2 = sim.topk(2, dim=-1).values
    clear = (top2[..., 0] - top2[..., 1]) > 1e-6
    clear[:, ~pm.any(dim=1)] = False                        # an all-masked page: every patch carries -1e4 (index 0 by the first-max rule, below)
    ref_arg = torch.as_tensor(want["argmax"]).long()
    assert torch.equal(arg[clear], ref_arg[clear])
    assert bool((arg[:, 4] == 0).all())                     # all-masked page -> index 0 like torch.max
    # the frozen-page path (prepared once per tensor) gives the same scores as the trainable-page path
    with torch.no_grad():
        s2 = ER.score_multi_vector_masked(Q.to(dev), P.to(dev), qm.to(dev), pm.to(dev))
    assert torch.equal(s2, s.detach())


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_wide_half_precision_inputs_are_upcast_like_the_reference(ER, O, dtype):
    dev = torch.device("cuda:0")
    Q, P, qm, pm, _ = R.width_case(256)
    Qh, Ph = Q.to(dtype), P.to(dtype)
    want = O.maxsim_masked(Qh.float(), Ph.float(), qm, pm)                          # evaluator/retrieval.py:176-177
    with torch.no_grad():
        got = ER.score_multi_vector_masked(Qh.to(dev), Ph.to(dev), qm.to(dev), pm.to(dev))
    np.testing.assert_allclose(got.cpu().numpy(), want.numpy(), atol=1e-4, rtol=0)
    assert got.dtype == torch.float32


@pytest.mark.parametrize("nq,lq,npg,lp", [(3, 32, 40, 1030), (9, 1, 25, 206), (4, 40, 18, 130), (33, 7, 9, 31), (1, 5, 300, 64)])
def test_wide_shapes_against_the_oracle(ER, O, nq, lq, npg, lp):
    """page lengths around the tile and stage boundaries, single-token packs, queries longer than one 32-token slice, more queries
    than one workgroup takes, many short pages -- at width 256, with ragged masks."""
    from evdr_amd import _lib as L
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(77 * nq + lp)
    Q = torch.nn.functional.normalize(torch.randn(nq, lq, 256, generator=gen), dim=-1)
    P = torch.nn.functional.normalize(torch.randn(npg, lp, 256, generator=gen), dim=-1)
    qm = torch.rand(nq, lq, generator=gen) > 0.2
    qm[:, 0] = True
    pm = torch.rand(npg, lp, generator=gen) > 0.15
    pm[0] = True
    pm[min(2, npg - 1)] = False
    pm[min(3, npg - 1), lp // 2:] = False
    want = O.maxsim_masked(Q, P, qm, pm)
    Pd = P.to(dev).requires_grad_(True)
    got = ER.score_multi_vector_masked(Q.to(dev), Pd, qm.to(dev), pm.to(dev))
    np.testing.assert_allclose(got.detach().cpu().numpy(), want.numpy(), atol=1e-4, rtol=0)
    assert L.load().evdr_last_fwd_kernel().decode().endswith("x2cols")
    g = torch.randn(nq, npg, generator=gen)
    (got * g.to(dev)).sum().backward()
    Po = P.clone().requires_grad_(True)
    (O.maxsim_masked(Q, Po, qm, pm) * g).sum().backward()
    np.testing.assert_allclose(Pd.grad.cpu().numpy(), Po.grad.numpy(), atol=2e-6, rtol=0)


def test_wide_non_finite_inputs_land_where_the_reference_puts_them(ER):
    dev = torch.device("cuda:0")
    Q, P, qm, pm, _ = R.width_case(256)
    P = P.clone()
    Q = Q.clone()
    P[2, 10, 200] = float("nan")                            # second column block of a valid patch -> the page's column
    P[6, 50, 3] = float("nan")                              # a MASKED patch (pm[6, 40:] is False): replaced by -1e4, does not count
    Q[3, 4, 130] = float("inf")                             # second column block of a query token -> the query's row (Inf treated like NaN)
    with torch.no_grad():
        s = ER.score_multi_vector_masked(Q.to(dev), P.to(dev), qm.to(dev), pm.to(dev)).cpu()
    has = pm.any(dim=1)
    assert bool(torch.isnan(s[:, 2]).all()) and bool(torch.isnan(s[3, has]).all())
    clean = torch.ones_like(s, dtype=torch.bool)
    clean[:, 2] = False
    clean[3, :] = False
    assert bool(torch.isfinite(s[clean]).all()) and float(s[3, 4]) == 0.0           # the all-masked page keeps its exact zero


def test_a2_list_scorer_and_width_limits(ER, O):
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(5)
    qs = [torch.randn(n, 256, generator=gen) for n in (5, 9, 3)]
    ps = [torch.randn(n, 256, generator=gen) for n in (40, 17, 66, 33)]
    got = ER.BaseVisualRetrieverProcessor.score_multi_vector(qs, ps, batch_size=2, device="cuda:0")
    want = O.maxsim_unmasked_lists(qs, ps, batch_size=2)
    np.testing.assert_allclose(got.numpy(), want.numpy(), atol=2e-4, rtol=1e-5)
    with pytest.raises(NotImplementedError):
        ER.score_multi_vector_masked(torch.randn(2, 4, 300, device=dev), torch.randn(3, 8, 300, device=dev),
                                     torch.ones(2, 4, dtype=torch.bool, device=dev), torch.ones(3, 8, dtype=torch.bool, device=dev))


def test_resident_corpus_of_wide_pages(O):
    """PageCorpus / ShardedRetriever on 200-wide pages (kept as four planes): scores against the oracle, the per-shard top-k + merge
    equal to the single-shard ranking bit for bit, a width mismatch refused."""
    import evdr_amd  # noqa: F401
    from evdr_amd.corpus import PageCorpus, merge_candidates
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(44)
    P = torch.nn.functional.normalize(torch.randn(90, 130, 200, generator=gen), dim=-1)
    Q = torch.nn.functional.normalize(torch.randn(12, 32, 200, generator=gen), dim=-1)
    pm = torch.rand(90, 130, generator=gen) > 0.1
    qm = torch.rand(12, 32, generator=gen) > 0.1
    qm[:, 0] = True
    corpus = PageCorpus.from_tensor(P.to(dev), pm.to(dev))
    assert corpus.nplanes == 4
    s = corpus.score(Q.to(dev), qm.to(dev))
    np.testing.assert_allclose(s.cpu().numpy(), O.maxsim_masked(Q, P, qm, pm).numpy(), atol=1e-4, rtol=0)
    ts, ti = corpus.topk(Q.to(dev), qm.to(dev), 10)
    parts = [corpus.shard(lo, hi).topk(Q.to(dev), qm.to(dev), 10) for lo, hi in ((0, 31), (31, 64), (64, 90))]
    ms, mi = merge_candidates(torch.cat([p[0] for p in parts], dim=1), torch.cat([p[1] for p in parts], dim=1), 10)
    assert torch.equal(ms, ts) and torch.equal(mi, ti)
    with pytest.raises(RuntimeError):
        corpus.score(torch.randn(2, 32, 128, device=dev))
