"""Parity of the HIP path (through the C ABI) against the golden fixtures and the CPU oracle.
Tolerances (SURVEY §8(d) parity gates): scores |d| <= 1e-4 abs on identical inputs (observed ~2e-6);
argmax / top-k indices bit-exact where ties are absent or resolved by the stated rule; train step:
loss rtol 1e-5, grad atol 1e-6, AdamW-updated parameters atol 1e-6."""
import numpy as np
import pytest
import torch

import golden_recipes as R
from oracle import maxsim_oracle as O

pytestmark = pytest.mark.gpu

SCORE_ATOL = 1e-4


@pytest.fixture(scope="module")
def dev():
    import evdr_amd  # noqa: F401
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def ER(dev):
    import evdr_amd.evaluator.retrieval as er
    return er


def T(x):
    return torch.from_numpy(np.asarray(x))


@pytest.mark.parametrize("case", ["small_ragged", "lq1", "chunk_tail"])
@pytest.mark.parametrize("as_bf16", [False, True])
def test_a1_small_forward(golden, dev, ER, case, as_bf16):
    z = golden(f"a1_{case}")
    Q, P, qm, pm, g = R.small_case(case)
    cast = (lambda t: t.bfloat16()) if as_bf16 else (lambda t: t)       # values are bf16-representable
    s = ER.score_multi_vector_masked(cast(Q).to(dev), cast(P).to(dev), qm.to(dev), pm.to(dev), chunk_p=R.SMALL_CHUNK[case])
    assert s.dtype == torch.float32 and s.device.type == "cuda" and tuple(s.shape) == z["scores"].shape
    np.testing.assert_allclose(s.cpu().numpy(), z["scores"], atol=SCORE_ATOL, rtol=0)


@pytest.mark.parametrize("case", ["small_ragged", "lq1", "chunk_tail"])
def test_a6_small_backward_and_argmax(golden, dev, ER, case):
    import evdr_amd.ops as ops
    z = golden(f"a1_{case}")
    Q, P, qm, pm, g = R.small_case(case)
    Pd = P.to(dev).requires_grad_(True)
    s = ER.score_multi_vector_masked(Q.to(dev), Pd, qm.to(dev), pm.to(dev))
    (s * g.to(dev)).sum().backward()
    np.testing.assert_allclose(s.detach().cpu().numpy(), z["scores"], atol=SCORE_ATOL, rtol=0)
    np.testing.assert_allclose(Pd.grad.cpu().numpy(), z["dP"], atol=1e-6, rtol=0)
    # known answers: masked positions / all-masked pages get EXACT zeros
    assert torch.all(Pd.grad.cpu()[~pm] == 0)
    # argmax: bit-exact vs torch.max of the reference, first index on ties (duplicates in page 1)
    for as_bf16 in (False, True):
        Qx, Px = (Q.bfloat16(), P.bfloat16()) if as_bf16 else (Q, P)
        _, arg = ops.maxsim_forward(Qx.to(dev), Px.to(dev), qm.to(dev), pm.to(dev), want_argmax=True)
        arg = arg.cpu().numpy().astype(np.int32) & 0xFFFF
        live = pm.any(dim=1).numpy()                     # argmax of an all-masked page is irrelevant (weight 0)
        assert np.array_equal(arg[:, live, :], z["argmax"][:, live, :])
        assert np.array_equal(arg[:, ~live, :], z["argmax"][:, ~live, :])   # but it matches anyway: index 0


def test_a1_all_masked_page_is_exact_zero(dev, ER):
    Q, P, qm, pm, g = R.small_case("small_ragged")
    s = ER.score_multi_vector_masked(Q.to(dev), P.to(dev), qm.to(dev), pm.to(dev)).cpu()
    assert torch.all(s[:, 2] == 0.0)


@pytest.mark.parametrize("tag,bf16", [("f32", False), ("bf16", True)])
def test_a1_seeded_1030(golden, dev, ER, tag, bf16):
    z = golden("a1_seeded1030_" + tag)
    Q, P, qm, pm = R.seeded_1030(bf16_inputs=bf16)
    if bf16:
        Q, P = Q.bfloat16(), P.bfloat16()
    s = ER.score_multi_vector_masked(Q.to(dev), P.to(dev), qm.to(dev), pm.to(dev), chunk_p=64)
    np.testing.assert_allclose(s.cpu().numpy(), z["scores"], atol=SCORE_ATOL, rtol=0)
    # observed accuracy is far inside the gate
    assert np.abs(s.cpu().numpy() - z["scores"]).max() < 2e-5


@pytest.mark.parametrize("nq", [1, 5, 12, 33, 70])          # exercises the 1/2/4 queries-per-wave variants + ragged groups
@pytest.mark.parametrize("lq,lp", [(32, 1030), (7, 45), (50, 100), (1, 64)])
def test_a1_vs_oracle_shapes(dev, ER, nq, lq, lp):
    gen = torch.Generator().manual_seed(nq * 1000 + lq * 10 + lp)
    npg = 37
    Q = torch.nn.functional.normalize(torch.randn(nq, lq, 128, generator=gen), dim=-1).bfloat16()
    P = torch.nn.functional.normalize(torch.randn(npg, lp, 128, generator=gen), dim=-1).bfloat16()
    qm = torch.rand(nq, lq, generator=gen) > 0.2
    pm = torch.rand(npg, lp, generator=gen) > 0.3
    pm[5] = False
    pm[6] = True
    want = O.maxsim_masked(Q.float(), P.float(), qm, pm)
    got = ER.score_multi_vector_masked(Q.to(dev), P.to(dev), qm.to(dev), pm.to(dev)).cpu()
    np.testing.assert_allclose(got.numpy(), want.numpy(), atol=SCORE_ATOL, rtol=0)
    got32 = ER.score_multi_vector_masked(Q.float().to(dev), P.float().to(dev), qm.to(dev), pm.to(dev)).cpu()
    np.testing.assert_allclose(got32.numpy(), want.numpy(), atol=SCORE_ATOL, rtol=0)


def test_a1_fp32_path_accuracy(dev, ER):
    """fp32 inputs NOT representable in 16 bits: the fp16 hi/lo path must track an fp64 computation like fp32 does."""
    gen = torch.Generator().manual_seed(77)
    Q = torch.nn.functional.normalize(torch.randn(16, 32, 128, generator=gen), dim=-1)
    P = torch.nn.functional.normalize(torch.randn(64, 300, 128, generator=gen), dim=-1)
    qm = torch.ones(16, 32, dtype=torch.bool)
    pm = torch.ones(64, 300, dtype=torch.bool)
    sim = torch.einsum("qnd,pmd->qpnm", Q.double(), P.double())
    want = sim.amax(-1).sum(-1)
    got = ER.score_multi_vector_masked(Q.to(dev), P.to(dev), qm.to(dev), pm.to(dev)).cpu().double()
    ref32 = O.maxsim_masked(Q, P, qm, pm).double()
    err_got, err_ref = (got - want).abs().max().item(), (ref32 - want).abs().max().item()
    assert err_got < 1e-5, err_got
    assert err_got < 10 * max(err_ref, 1e-6), (err_got, err_ref)


@pytest.mark.parametrize("nq", [5, 8, 9, 13, 16])
@pytest.mark.parametrize("lp", [256, 300, 1030])
def test_a1_few_queries_long_pages(dev, ER, nq, lp):
    """5..16 bf16 queries against long pages (partly idle workgroups of the staged kernel): prefix masks, holes, an
    all-masked page, ragged query masks, several pages per workgroup."""
    gen = torch.Generator().manual_seed(1000 + 17 * nq + lp)
    npg = 37
    Q = torch.nn.functional.normalize(torch.randn(nq, 32, 128, generator=gen), dim=-1).bfloat16()
    P = torch.nn.functional.normalize(torch.randn(npg, lp, 128, generator=gen), dim=-1).bfloat16()
    qm = torch.rand(nq, 32, generator=gen) > 0.15
    pm = torch.ones(npg, lp, dtype=torch.bool)
    pm[3] = False
    pm[5, lp // 2:] = False
    pm[6, 17:] = False
    pm[9] = torch.rand(lp, generator=gen) > 0.4
    pm[20, : lp - 3] = False
    want = O.maxsim_masked(Q.float(), P.float(), qm, pm)
    got = ER.score_multi_vector_masked(Q.to(dev), P.to(dev), qm.to(dev), pm.to(dev))
    np.testing.assert_allclose(got.cpu().numpy(), want.numpy(), atol=SCORE_ATOL)
    assert got[:, 3].abs().max().item() == 0.0


@pytest.mark.parametrize("nq", [1, 2, 31, 32, 33, 70])
@pytest.mark.parametrize("bf16", [False, True])
def test_a1_single_token_queries(dev, ER, nq, bf16):
    """Lq = 1 "virtual queries" (mainv3_iter_liscore_QA_hardtoken.py:428-434): scored 32 to an MFMA tile with one output
    row per token; forward, argmax and both gradients against the oracle."""
    gen = torch.Generator().manual_seed(500 + nq)
    npg, lp = 19, 77
    Q = torch.nn.functional.normalize(torch.randn(nq, 1, 128, generator=gen), dim=-1)
    P = torch.nn.functional.normalize(torch.randn(npg, lp, 128, generator=gen), dim=-1)
    if bf16:
        Q, P = Q.bfloat16().float(), P.bfloat16().float()
    qm = torch.rand(nq, 1, generator=gen) > 0.2
    pm = torch.rand(npg, lp, generator=gen) > 0.3
    pm[2] = False
    pm[4, 30:] = False
    want, warg = O.maxsim_masked_argmax(Q, P, qm, pm)
    dt = torch.bfloat16 if bf16 else torch.float32
    from evdr_amd import ops
    got, arg = ops.maxsim_forward(Q.to(dev, dt), P.to(dev, dt), qm.to(dev), pm.to(dev), want_argmax=True)
    np.testing.assert_allclose(got.cpu().numpy(), want.numpy(), atol=SCORE_ATOL)
    live = qm[:, None, :] & pm.any(-1)[None, :, None]
    assert torch.equal(arg.cpu().long()[live], warg.long()[live])
    Pd = P.to(dev).requires_grad_(True)
    Qd = Q.to(dev).requires_grad_(True)
    gsc = torch.randn(nq, npg, generator=gen)
    ER.score_multi_vector_masked(Qd, Pd, qm.to(dev), pm.to(dev)).backward(gsc.to(dev))
    Pc, Qc = P.clone().requires_grad_(True), Q.clone().requires_grad_(True)
    O.maxsim_masked(Qc, Pc, qm, pm).backward(gsc)
    np.testing.assert_allclose(Pd.grad.cpu().numpy(), Pc.grad.numpy(), atol=2e-5, rtol=1e-5)
    np.testing.assert_allclose(Qd.grad.cpu().numpy(), Qc.grad.numpy(), atol=2e-5, rtol=1e-5)


def test_prepared_pages_cache(dev, ER):
    """Frozen fp32 pages are prepared once per tensor; any in-place change (version counter) or a new tensor re-prepares."""
    gen = torch.Generator().manual_seed(81)
    Q = torch.nn.functional.normalize(torch.randn(6, 16, 128, generator=gen), dim=-1).to(dev)
    P = torch.nn.functional.normalize(torch.randn(12, 60, 128, generator=gen), dim=-1).to(dev)
    pm = (torch.rand(12, 60, generator=gen) > 0.2).to(dev)
    qm = torch.ones(6, 16, dtype=torch.bool, device=dev)
    ER.forget_prepared()
    a = ER.score_multi_vector_masked(Q, P, qm, pm)
    assert len(ER._PREPARED) == 1
    b = ER.score_multi_vector_masked(Q, P, qm, pm)                       # hit
    assert len(ER._PREPARED) == 1 and torch.equal(a, b)
    np.testing.assert_allclose(a.cpu().numpy(), O.maxsim_masked(Q.cpu(), P.cpu(), qm.cpu(), pm.cpu()).numpy(), atol=SCORE_ATOL)
    P.mul_(-1.0)                                                          # in-place change: must not hit the stale planes
    c = ER.score_multi_vector_masked(Q, P, qm, pm)
    np.testing.assert_allclose(c.cpu().numpy(), O.maxsim_masked(Q.cpu(), P.cpu(), qm.cpu(), pm.cpu()).numpy(), atol=SCORE_ATOL)
    pm[0] = False                                                         # so does a change of the mask
    d = ER.score_multi_vector_masked(Q, P, qm, pm)
    assert d[:, 0].abs().max().item() == 0.0
    Pg = P.clone().requires_grad_(True)                                   # trainable pages never go through the cache
    n0 = len(ER._PREPARED)
    ER.score_multi_vector_masked(Q, Pg, qm, pm).sum().backward()
    assert len(ER._PREPARED) == n0 and Pg.grad is not None
    Qg = Q.clone().requires_grad_(True)                                   # frozen pages, trainable queries: dQ through the cache path
    ER.score_multi_vector_masked(Qg, P, qm, pm).sum().backward()
    Qc = Q.cpu().clone().requires_grad_(True)
    O.maxsim_masked(Qc, P.cpu(), qm.cpu(), pm.cpu()).sum().backward()
    np.testing.assert_allclose(Qg.grad.cpu().numpy(), Qc.grad.numpy(), atol=2e-5, rtol=1e-5)
    # embeddings narrower than 128 are padded INSIDE the cache (keyed on the caller's tensor): a second call hits
    ER.forget_prepared()
    Pn, Qn = P[..., :48].contiguous(), Q[..., :48].contiguous()
    e1 = ER.score_multi_vector_masked(Qn, Pn, qm, pm)
    assert len(ER._PREPARED) == 1 and next(iter(ER._PREPARED.values()))[0]().data_ptr() == Pn.data_ptr()      # the 48-wide tensor itself
    e2 = ER.score_multi_vector_masked(Qn, Pn, qm, pm)
    assert len(ER._PREPARED) == 1 and torch.equal(e1, e2)
    np.testing.assert_allclose(e1.cpu().numpy(), O.maxsim_masked(Qn.cpu(), Pn.cpu(), qm.cpu(), pm.cpu()).numpy(), atol=SCORE_ATOL)
    del P, a, b, c, d, Pn, e1, e2
    import gc
    gc.collect()
    assert all(e[0]() is not None for e in ER._PREPARED.values())         # entries of dead tensors are dropped at once
    ER.forget_prepared()


@pytest.mark.parametrize("frac", [1.0, 0.4])
def test_a6_backward_hot_patch_row(dev, ER, frac):
    """A salient patch that wins most (query, token) pairs of a page: the balanced backward splits that row's bucket over
    several lane groups (partial sums through LDS atomics); dP and the fused AdamW update must not care."""
    gen = torch.Generator().manual_seed(90)
    B, Lq, N, Ls = 32, 32, 6, 100
    u = torch.nn.functional.normalize(torch.randn(128, generator=gen), dim=0)
    Q = torch.nn.functional.normalize(u + 0.15 * torch.randn(B, Lq, 128, generator=gen), dim=-1)
    hot = torch.rand(B, Lq, generator=gen) < frac
    Q = torch.where(hot[..., None], Q, torch.nn.functional.normalize(torch.randn(B, Lq, 128, generator=gen), dim=-1))
    P = torch.nn.functional.normalize(torch.randn(N, Ls, 128, generator=gen), dim=-1)
    P[:, 7] = u                                                           # every page: patch 7 is the salient one
    qm = torch.rand(B, Lq, generator=gen) > 0.1
    pm = torch.ones(N, Ls, dtype=torch.bool)
    pm[2, 50:] = False
    _, warg = O.maxsim_masked_argmax(Q, P, qm, pm)
    assert (warg == 7).float().mean().item() > 0.8 * frac                 # the construction does concentrate the argmax
    gsc = torch.randn(B, N, generator=gen)
    Pd = P.to(dev).requires_grad_(True)
    ER.score_multi_vector_masked(Q.to(dev), Pd, qm.to(dev), pm.to(dev)).backward(gsc.to(dev))
    Pc = P.clone().requires_grad_(True)
    O.maxsim_masked(Q, Pc, qm, pm).backward(gsc)
    np.testing.assert_allclose(Pd.grad.cpu().numpy(), Pc.grad.numpy(), atol=5e-5, rtol=1e-5)
    assert float(Pd.grad.cpu()[2, 50:].abs().max()) == 0.0                # masked rows: exact zeros


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("lp,lq", [(40, 50), (300, 50), (300, 70)])
def test_a1_queries_padded_beyond_32_tokens(dev, ER, dt, lp, lq):
    """Queries padded to the longest of a set (50 tokens), most of them shorter than 32: the second 32-token slice runs on
    the compacted list of long queries only (and not at all for workgroups without one); scores of every query, long or
    short, empty or full, against the oracle."""
    gen = torch.Generator().manual_seed(300 + lp + lq)
    nq, npg = 77, 21
    lens = torch.randint(1, 33, (nq,), generator=gen)
    lens[[3, 40, 41, 76]] = torch.tensor([lq, 33, lq - 3, 34])      # a few long ones, scattered over the wave slots
    lens[5] = 0                                                     # a query with no valid token at all
    qm = torch.arange(lq)[None, :] < lens[:, None]
    Q = torch.nn.functional.normalize(torch.randn(nq, lq, 128, generator=gen), dim=-1)
    P = torch.nn.functional.normalize(torch.randn(npg, lp, 128, generator=gen), dim=-1)
    pm = torch.rand(npg, lp, generator=gen) > 0.2
    if dt == torch.bfloat16:
        Q, P = Q.bfloat16().float(), P.bfloat16().float()
    want = O.maxsim_masked(Q, P, qm, pm)
    got = ER.score_multi_vector_masked(Q.to(dev, dt), P.to(dev, dt), qm.to(dev), pm.to(dev))
    np.testing.assert_allclose(got.cpu().numpy(), want.numpy(), atol=SCORE_ATOL)
    assert got[5].abs().max().item() == 0.0
    got_all = ER.score_multi_vector_masked(Q.to(dev, dt), P.to(dev, dt), torch.ones(nq, lq, dtype=torch.bool, device=dev), pm.to(dev))
    np.testing.assert_allclose(got_all.cpu().numpy(), O.maxsim_masked(Q, P, torch.ones(nq, lq, dtype=torch.bool), pm).numpy(), atol=SCORE_ATOL)


def test_a1_many_queries_few_pages(dev, ER):
    """The teacher-score precompute shape of the training scripts (tens of thousands of pseudo-queries against a small
    page set): thousands of query groups per page chunk, a single page, and the padded-slice path at that scale."""
    gen = torch.Generator().manual_seed(31)
    nq, npg, lq, lp = 20011, 3, 40, 45
    Q = torch.nn.functional.normalize(torch.randn(nq, lq, 128, generator=gen), dim=-1).bfloat16()
    P = torch.nn.functional.normalize(torch.randn(npg, lp, 128, generator=gen), dim=-1).bfloat16()
    lens = torch.randint(1, 41, (nq,), generator=gen)
    qm = torch.arange(lq)[None, :] < lens[:, None]
    pm = torch.rand(npg, lp, generator=gen) > 0.3
    got = ER.score_multi_vector_masked(Q.to(dev), P.to(dev), qm.to(dev), pm.to(dev)).cpu()
    sel = torch.randperm(nq, generator=gen)[:400]
    want = O.maxsim_masked(Q[sel].float(), P.float(), qm[sel], pm)
    np.testing.assert_allclose(got[sel].numpy(), want.numpy(), atol=SCORE_ATOL)
    one = ER.score_multi_vector_masked(Q[:700].to(dev), P[:1].to(dev), qm[:700].to(dev), pm[:1].to(dev)).cpu()
    assert torch.equal(one[:, 0], got[:700, 0])                    # a single page: the same numbers as inside the larger call


def test_split_f32_planes(dev):
    """evdr_split_f32: hi + lo == x * 2^k to 2^-21 relative, k from the absmax word, which holds the bits of max|x|."""
    from evdr_amd import ops
    gen = torch.Generator().manual_seed(78)
    # 37 x 19 rows: the one-launch form for tensors up to 1 MB (a query batch); 300 x 32 rows: absmax pass + split pass
    for scale, shape in ((1.0, (37, 19, 128)), (3e4, (37, 19, 128)), (2e-7, (37, 19, 128)), (1.0, (300, 32, 128)), (5e-3, (300, 32, 128))):
        x = (torch.randn(*shape, generator=gen) * scale).to(dev)
        x[0, 0, :5] = 0.0
        planes, amax = ops.split_f32(x)
        assert planes.shape == (2,) + shape and planes.dtype == torch.float16
        bits = int(amax.item())
        assert bits == int(x.abs().max().view(torch.int32).item())
        k = 141 - ((bits >> 23) & 0xFF)
        back = (planes[0].double() + planes[1].double()) * 2.0 ** (-k)
        rel = ((back - x.double()).abs() / x.abs().max().double()).max().item()
        assert torch.isfinite(planes.float()).all() and rel < 2.0 ** -21, (scale, rel)
        assert planes[0].float().abs().max().item() < 2.0 ** 15 + 16
    z, za = ops.split_f32(torch.zeros(4, 128, device=dev))
    assert int(za.item()) == 0 and float(z.float().abs().max()) == 0.0


def test_l2norm_split_planes(dev):
    """evdr_l2norm_fwd_split: the planes it writes decode to l2_normalize(x * mask), and score like the fp32 tensor."""
    from evdr_amd import ops
    gen = torch.Generator().manual_seed(80)
    x = (torch.randn(9, 50, 128, generator=gen) * 3.0).to(dev)
    m = (torch.rand(9, 50, generator=gen) > 0.25).to(dev)
    x[2, 7] = 0.0                                                   # zero row stays exactly zero
    y, _ = ops.l2norm_forward(x, m, 1e-12)
    planes, amax = ops.l2norm_split(x, m, 1e-12)
    assert int(amax.item()) == 0x3F800000
    back = (planes[0].double() + planes[1].double()) * 2.0 ** -14
    assert (back - y.double()).abs().max().item() < 2.0 ** -22
    assert float(back[2, 7].abs().max()) == 0.0 and float(back[~m].abs().max()) == 0.0
    Q = torch.nn.functional.normalize(torch.randn(5, 12, 128, generator=gen), dim=-1).to(dev)
    qp, qa = ops.split_f32(Q)
    tm, pf = ops.pack_pmask(m, 9, 50, dev)
    got, arg = ops.maxsim_forward_prepared(qp, qa, planes, amax, None, tm, pf, want_argmax=True)
    want, warg = ops.maxsim_forward(Q, y, None, m, want_argmax=True)
    np.testing.assert_allclose(got.cpu().numpy(), want.cpu().numpy(), atol=2e-6)
    assert torch.equal(arg, warg)


@pytest.mark.parametrize("npg,lp,nq,lq", [(3, 1, 2, 5), (5, 37, 4, 32), (7, 128, 3, 17), (4, 206, 32, 32), (2, 300, 6, 32)])
def test_update_kernel_emits_next_planes(dev, npg, lp, nq, lq):
    """evdr_maxsim_bwd_adamw_planes over page lengths on both sides of the 128-row slab: the update equals the plain
    evdr_maxsim_bwd_adamw one, and the planes it leaves are bit-for-bit evdr_l2norm_fwd_split of the updated parameter
    (masked rows, an empty page, a zero row)."""
    from evdr_amd import ops
    gen = torch.Generator().manual_seed(1000 * lp + npg)
    x0 = (torch.randn(npg, lp, 128, generator=gen) * 0.7).to(dev)
    pm = (torch.rand(npg, lp, generator=gen) > 0.3)
    pm[0] = True
    if npg > 2:
        pm[2] = False                                               # a page without a valid patch
    pm = pm.to(dev)
    x0 = x0 * pm.unsqueeze(-1)
    Q = torch.nn.functional.normalize(torch.randn(nq, lq, 128, generator=gen), dim=-1).to(dev)
    g = (torch.randn(nq, npg, generator=gen) * 1e-2).to(dev)
    arg = torch.randint(0, lp, (nq, npg, lq), generator=gen).to(torch.int16).to(dev)
    st = {}
    for tag in ("plain", "planes"):
        x, ea, es = x0.clone(), torch.zeros_like(x0), torch.zeros_like(x0)
        tm, pf = ops.pack_pmask(pm, npg, lp, dev)
        planes = ops.l2norm_split(x, pm, 1e-12, pageflags=pf) if tag == "planes" else None
        for step in (1, 2):
            ops.maxsim_backward_adamw(g, Q, None, pm, arg, x, ea, es, 1e-3, (0.9, 0.999), 1e-8, 1e-2, step, next_planes=planes,
                                      pageflags=pf if planes is not None else None)
        st[tag] = (x, ea, es, planes, pf)
    for a, b in zip(st["plain"][:3], st["planes"][:3]):
        np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), atol=1e-7, rtol=1e-6)
    x, _, _, (planes, amax), pf = st["planes"]
    fresh, fresh_amax = ops.l2norm_split(x, pm, 1e-12)
    assert torch.equal(planes.view(torch.int16), fresh.view(torch.int16)) and int(amax.item()) == int(fresh_amax.item()) == 0x3F800000
    assert not (pf & 8).any()
    assert float(x[~pm].abs().max() if (~pm).any() else 0.0) == 0.0   # masked rows never move


@pytest.mark.parametrize("case", ["no_pairs", "empty_page"])
@pytest.mark.parametrize("lp", [37, 206])
def test_update_kernel_rows_without_gradient_follow_torch_adamw(dev, case, lp):
    """The fused update on rows it has NO gradient for -- a zero-gradient step (nq = 0: g / Q / argmax may be NULL,
    include/evdr.h) and a page without a valid patch whose masked rows hold non-zero parameters AND non-zero moments --
    against torch.optim.AdamW fed the same gradient (zero there) from the same seeded state: decay + moment update, never
    a parameter rebuilt from zeros (the early parameter-row request of the kernel sits inside the gather loop, which such
    workgroups never enter).  Moments are seeded directly (step 2 of an optimiser), so that no update is sign-like."""
    from evdr_amd import ops
    gen = torch.Generator().manual_seed(4100 + lp)
    npg, nq, lq = 5, 6, 32
    x0 = (torch.randn(npg, lp, 128, generator=gen) * 0.7).to(dev)          # NOT pre-multiplied by the mask
    ea0 = (torch.randn(npg, lp, 128, generator=gen) * 1e-3).to(dev)
    es0 = (torch.rand(npg, lp, 128, generator=gen) * 1e-5 + 1e-7).to(dev)
    pm = (torch.rand(npg, lp, generator=gen) > 0.3)
    pm[0] = True
    pm[2] = False                                                           # a page without a valid patch
    pm = pm.to(dev)
    n = 0 if case == "no_pairs" else nq
    Q = torch.nn.functional.normalize(torch.randn(nq, lq, 128, generator=gen), dim=-1).to(dev)[:n]
    g = (torch.randn(nq, npg, generator=gen) * 1e-2).to(dev)[:n]
    arg = torch.randint(0, lp, (nq, npg, lq), generator=gen).to(torch.int16).to(dev)[:n]
    hyper = dict(lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2)

    x, ea, es = x0.clone(), ea0.clone(), es0.clone()
    ops.maxsim_backward_adamw(g, Q, None, pm, arg, x, ea, es, hyper["lr"], hyper["betas"], hyper["eps"], hyper["weight_decay"], 2)

    xr = x0.clone().requires_grad_(True)
    opt = torch.optim.AdamW([xr], **hyper)
    opt.state[xr] = {"step": torch.tensor(1.0), "exp_avg": ea0.clone(), "exp_avg_sq": es0.clone()}
    xm = xr * pm.unsqueeze(-1)
    y = xm / (xm.norm(dim=-1, keepdim=True) + 1e-12)
    y.backward(ops.maxsim_backward(g, Q, None, pm, arg, npg, lp) if n else torch.zeros_like(x0))
    opt.step()
    want = (xr.detach(), opt.state[xr]["exp_avg"], opt.state[xr]["exp_avg_sq"])

    quiet = torch.ones(npg, lp, dtype=torch.bool, device=dev) if n == 0 else ~pm      # rows without a gradient
    assert bool(quiet[2].all())
    for a, b, tol in zip((x, ea, es), want, (2e-7, 1e-9, 1e-11)):
        np.testing.assert_allclose(a[quiet].cpu().numpy(), b[quiet].cpu().numpy(), atol=tol, rtol=2e-6)
        np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), atol=50 * tol, rtol=1e-4)
    moved = (x - x0).abs()[quiet]
    assert float(moved.max()) < 0.02 and float((x[quiet] - x0[quiet] * (1 - 1e-5)).abs().max()) < 0.02   # decayed, not rebuilt from zero


@pytest.mark.parametrize("qs,ps", [(1.0, 1.0), (4096.0, 1.0 / 8192.0), (1e-3, 37.5), (250.0, 250.0)])
def test_a1_fp32_path_any_magnitude(dev, ER, qs, ps):
    """fp32 inputs far from unit norm (the per-tensor power-of-two scaling of the fp16 planes): scores, argmax and the
    -1e4 of masked patches in real units."""
    gen = torch.Generator().manual_seed(79)
    Q = torch.randn(11, 20, 128, generator=gen) * qs
    P = torch.randn(23, 70, 128, generator=gen) * ps
    qm = torch.rand(11, 20, generator=gen) > 0.2
    pm = torch.rand(23, 70, generator=gen) > 0.3
    pm[4] = False
    pm[6, 40:] = False
    sim = torch.einsum("qnd,pmd->qpnm", Q.double(), P.double()).masked_fill(~pm[None, :, None, :], -1e4)
    mx, ix = sim.max(-1)
    want = (mx * pm.any(-1)[None, :, None] * qm[:, None, :]).sum(-1)
    from evdr_amd import ops
    got, arg = ops.maxsim_forward(Q.to(dev), P.to(dev), qm.to(dev), pm.to(dev), want_argmax=True)
    scale = float(want.abs().max())
    assert (got.cpu().double() - want).abs().max().item() <= 2e-6 * scale
    live = qm[:, None, :] & pm.any(-1)[None, :, None]
    assert torch.equal(arg.cpu().long()[live], ix[live])


def test_non_contiguous_and_nonbool_inputs(dev, ER):
    gen = torch.Generator().manual_seed(8)
    Pbig = torch.nn.functional.normalize(torch.randn(20, 40, 128, generator=gen), dim=-1).bfloat16()
    Q = torch.nn.functional.normalize(torch.randn(9, 8, 128, generator=gen), dim=-1).bfloat16()
    pm = (torch.rand(20, 40, generator=gen) > 0.3)
    qm = torch.ones(9, 8)
    Pd, pmd = Pbig.to(dev), pm.to(dev)
    got = ER.score_multi_vector_masked(Q.to(dev), Pd[3:17], qm.to(dev), pmd[3:17].to(torch.int64))   # slice + int mask
    want = O.maxsim_masked(Q.float(), Pbig[3:17].float(), qm, pm[3:17])
    np.testing.assert_allclose(got.cpu().numpy(), want.numpy(), atol=SCORE_ATOL)
    got2 = ER.score_multi_vector_masked(Q.to(dev), Pd[::2], qm.to(dev), pmd[::2])                     # strided pages
    want2 = O.maxsim_masked(Q.float(), Pbig[::2].float(), qm, pm[::2])
    np.testing.assert_allclose(got2.cpu().numpy(), want2.numpy(), atol=SCORE_ATOL)


def test_cpu_tensors_are_rejected(ER):
    Q, P, qm, pm, g = R.small_case("chunk_tail")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ER.score_multi_vector_masked(Q, P, qm, pm)


def test_a2_unmasked_lists(golden, dev, ER):
    z = golden("a2_unmasked_lists")
    qs, ps = R.ragged_lists_case()
    for dtype in (torch.float32, torch.bfloat16, torch.float16):
        q2, p2 = [q.to(dtype) for q in qs], [p.to(dtype) for p in ps]
        s4 = ER.BaseVisualRetrieverProcessor.score_multi_vector(q2, p2, batch_size=4, device="cuda:0")
        s128 = ER.BaseVisualRetrieverProcessor.score_multi_vector(q2, p2, batch_size=128, device="cuda:0")
        assert s4.device.type == "cpu" and s4.dtype == torch.float32
        tol = SCORE_ATOL if dtype != torch.float16 else 2e-2      # fp16 rounds the bf16-representable inputs
        np.testing.assert_allclose(s4.numpy(), z["scores_bs4"], atol=tol)
        np.testing.assert_allclose(s128.numpy(), z["scores_bs128"], atol=tol)
    with pytest.raises(ValueError, match="No queries provided"):
        ER.BaseVisualRetrieverProcessor.score_multi_vector([], ps, device="cuda:0")
    with pytest.raises(ValueError, match="No passages provided"):
        ER.BaseVisualRetrieverProcessor.score_multi_vector(qs, [], device="cuda:0")


def test_a3_single_vector_on_gpu(golden, ER):
    z = golden("a3_single")
    qs, ps = R.single_vector_case()
    s = ER.BaseVisualRetrieverProcessor.score_single_vector(qs, ps, device="cuda:0")
    assert s.device.type == "cuda"
    np.testing.assert_allclose(s.cpu().numpy(), z["scores"], atol=1e-4)


def test_a5_infonce_kernel(golden, dev):
    from evdr_amd.criterion import infonce_distillation_loss
    z = golden("a5_infonce")
    ss = T(z["score_s"]).to(dev).requires_grad_(True)
    st = T(z["score_t"]).to(dev)
    loss = infonce_distillation_loss(ss, st, temperature=float(z["temp"]))
    loss.backward()
    np.testing.assert_allclose(loss.item(), float(z["loss"]), rtol=1e-5)
    np.testing.assert_allclose(ss.grad.cpu().numpy(), z["dscore"], atol=1e-6, rtol=1e-5)


@pytest.mark.parametrize("b,n", [(1, 7), (32, 500), (300, 129)])
def test_a5_one_launch_form_equals_two_launch_form(dev, b, n):
    """evdr_infonce_distill_fwd_bwd_ws (row kernel whose last workgroup also takes the mean; ops.infonce_distill with a
    caller-owned workspace) == evdr_infonce_distill_fwd_bwd (row kernel + mean kernel; ops.infonce_distill without one), bit
    for bit, call after call on one workspace (the ticket word returns to zero)."""
    from evdr_amd import _lib as L, ops
    gen = torch.Generator().manual_seed(b * 1000 + n)
    lib = L.load()
    st = L.current_stream_handle(dev)
    for rep in range(3):
        ss = (torch.randn(b, n, generator=gen) * 3).to(dev)
        tt = torch.randn(b, n, generator=gen).to(dev)
        loss2, row2, d2 = torch.empty((), device=dev), torch.empty(b, device=dev), torch.empty(b, n, device=dev)
        L.check(lib.evdr_infonce_distill_fwd_bwd(L.ptr(ss), L.ptr(tt), b, n, 0.1, L.ptr(loss2), L.ptr(d2), L.ptr(row2), st))
        if rep == 0:
            ws = ops.infonce_workspace(b, dev)
        loss1, d1 = ops.infonce_distill(ss, tt, 0.1, want_grad=True, ws=ws)
        assert loss1.item() == loss2.item() and torch.equal(d1, d2)
        assert ws.view(torch.int32)[b].item() == 0 and torch.equal(ws[:b], row2)
        loss3, d3 = ops.infonce_distill(ss, tt, 0.1, want_grad=True)                   # stateless form
        assert loss3.item() == loss2.item() and torch.equal(d3, d2)
    with pytest.raises(RuntimeError):
        ops.infonce_distill(ss, tt, 0.1, want_grad=True, ws=torch.zeros(b + 2, device=dev))


@pytest.mark.parametrize("tag", ["b4n8", "b32n128"])
def test_a7_train_step(golden, dev, ER, tag):
    """mainv2_iter_distill_infonce.py:269-292 call pattern with the drop-in functions, incl. the optimizer the scripts get from
    utils.set_optimizer (utils/utils.py:78-80: AdamW -- here on the one-pass update kernel)."""
    from evdr_amd.criterion import infonce_distillation_loss
    from evdr_amd.utils.preprocess_data import l2_normalize
    from evdr_amd.utils.utils import StreamAdamW, set_optimizer
    z = golden("a7_step_" + tag)
    Qb, qmb, Pt, pmt, Pbar0, pms, hp = [x.to(dev) if torch.is_tensor(x) else x for x in R.train_case(tag)]
    Pt_norm = l2_normalize(Pt * pmt.unsqueeze(-1)).detach()
    param = torch.nn.Parameter(Pbar0 * pms.unsqueeze(-1))
    opt = set_optimizer("adamw", param, hp["lr"], hp["wd"])
    assert isinstance(opt, StreamAdamW) and isinstance(opt, torch.optim.AdamW)
    Psb = l2_normalize(param * pms.unsqueeze(-1))
    with torch.no_grad():
        sc_t = ER.score_multi_vector_masked(Qb, Pt_norm, qmb, pmt, 64)
    sc_s = ER.score_multi_vector_masked(Qb, Psb, qmb, pms, 64)
    loss = infonce_distillation_loss(sc_s, sc_t, temperature=hp["temp"])
    opt.zero_grad(set_to_none=True)
    loss.backward()
    grad = param.grad.detach().clone()
    opt.step()
    np.testing.assert_allclose(sc_t.cpu().numpy(), z["sc_t"], atol=SCORE_ATOL)
    np.testing.assert_allclose(sc_s.detach().cpu().numpy(), z["sc_s"], atol=SCORE_ATOL)
    np.testing.assert_allclose(loss.item(), float(z["loss"]), rtol=1e-5)
    if tag == "b4n8":
        np.testing.assert_allclose(grad.cpu().numpy(), z["grad"], atol=1e-6)
        np.testing.assert_allclose(param.detach().cpu().numpy(), z["param_after"], atol=1e-6)
    else:
        np.testing.assert_allclose(grad[::8, ::8, ::4].cpu().numpy(), z["grad_sample"], atol=1e-6)
        # the normalisation runs on the GPU here (1-ulp input differences vs the CPU-made fixture), which may flip
        # a near-tied argmax and move one query token's contribution to a neighbouring row: the norm is compared
        # loosely, the exact full-tensor comparison on identical inputs is test_a6_full_gradient_identical_inputs
        np.testing.assert_allclose(grad.double().norm().item(), float(z["grad_norm"]), rtol=2e-3)
        # AdamW's first step is lr*g/(|g|+1e-8): where |g| ~ 1e-8 the 1e-8-level summation noise of the gradient is
        # amplified into the parameter (bounded by lr).  Require atol 1e-6 on all but a vanishing fraction.
        dpar = np.abs(param.detach()[::8, ::8, ::4].cpu().numpy() - z["param_sample"])
        assert (dpar > 1e-6).mean() < 1e-3 and dpar.max() < 2 * hp["lr"], ((dpar > 1e-6).mean(), dpar.max())
        np.testing.assert_allclose(param.detach().double().norm().item(), float(z["param_norm"]), rtol=1e-6)


@pytest.mark.parametrize("shape", [(500, 206, 128), (7, 13, 128), (3, 5), (1,), (4099,)])
def test_stream_adamw_equals_torch_adamw(dev, shape):
    """StreamAdamW (evdr_adamw_step: one pass) against torch.optim.AdamW step after step: weight decay, bias corrections, zero
    and tiny gradients, sizes that are not multiples of 4; the state keeps torch's layout, so state_dict round-trips into a plain
    torch.optim.AdamW that continues identically."""
    from evdr_amd.utils.utils import StreamAdamW
    gen = torch.Generator().manual_seed(sum(shape))
    x0 = torch.randn(shape, generator=gen)
    a = torch.nn.Parameter(x0.clone().to(dev))
    b = torch.nn.Parameter(x0.clone().to(dev))
    oa = StreamAdamW([a], lr=1e-3, weight_decay=1e-2)
    ob = torch.optim.AdamW([b], lr=1e-3, weight_decay=1e-2)
    for i in range(12):
        g = torch.randn(shape, generator=gen) * (10.0 ** -(i % 5))
        if i == 3:
            g.zero_()
        if i == 4:
            g.flatten()[:: 3] = 1e-9
        a.grad, b.grad = g.clone().to(dev), g.clone().to(dev)
        oa.step()
        ob.step()
        # the two forms round `x * decay - step_size * (m / denom)` in a different order (fused multiply-add or not): a parameter
        # may differ by an ulp per step (1.4 % of the elements do on the first step), nothing more
        np.testing.assert_allclose(a.detach().cpu().numpy(), b.detach().cpu().numpy(), rtol=2.4e-7 * (i + 1), atol=2e-7, err_msg=f"step {i}")
    sa, sb = oa.state[a], ob.state[b]
    assert float(sa["step"]) == float(sb["step"]) == 12.0
    # moments: the same formulas (lerp; mul + addcmul) with possibly different fused-multiply-add contraction: ulp-level
    np.testing.assert_allclose(sa["exp_avg"].cpu().numpy(), sb["exp_avg"].cpu().numpy(), rtol=1e-6, atol=5e-7)
    np.testing.assert_allclose(sa["exp_avg_sq"].cpu().numpy(), sb["exp_avg_sq"].cpu().numpy(), rtol=2e-6, atol=1e-12)
    c = torch.nn.Parameter(a.detach().clone())
    oc = torch.optim.AdamW([c], lr=1e-3, weight_decay=1e-2)
    import copy
    oc.load_state_dict(copy.deepcopy(oa.state_dict()))              # load_state_dict keeps same-device tensors by reference
    g = torch.randn(shape, generator=gen).to(dev)
    a.grad, c.grad = g.clone(), g.clone()
    oa.step()
    oc.step()
    np.testing.assert_allclose(a.detach().cpu().numpy(), c.detach().cpu().numpy(), rtol=2.4e-7, atol=2e-7)
    # a CPU parameter falls through to torch's own step (host-side plumbing; nothing of the scoring path is involved)
    d = torch.nn.Parameter(x0.clone())
    od = StreamAdamW([d], lr=1e-3, weight_decay=1e-2)
    d.grad = torch.ones_like(d)
    od.step()
    assert torch.isfinite(d).all() and not torch.equal(d.detach(), x0)


@pytest.mark.parametrize("mode", ["call_pattern", "call_pattern_cached", "fused"])
def test_a7_eight_step_trajectory(golden, dev, ER, mode, request):
    """Eight consecutive steps against the trajectory the reference's OWN train_one_step produced
    (tests/golden/make_golden_trajectory.py): every loss, the parameters after steps 1 / 4 / 8 and AdamW's moments --
    through the drop-in modules used like the script (autograd + utils.set_optimizer) and through the fused step.  Pins the
    optimizer's step count / bias corrections, the moments and the feedback of updated pages into the next forward.
    `call_pattern_cached` (round 6): the same script steps with `enable_score_cache()` after one pass over the query batches (an
    epoch that has filled the cache): all eight teacher calls are served from the device-side table -- same trajectory."""
    from evdr_amd import driver
    from evdr_amd.utils.preprocess_data import l2_normalize
    from evdr_amd.utils.utils import set_optimizer
    z = golden("a7_trajectory")
    batches, Pt, pmt, Pbar0, pms, hp = R.trajectory_case()
    Pt, pmt, Pbar0, pms = Pt.to(dev), pmt.to(dev), Pbar0.to(dev), pms.to(dev)
    Ptn = l2_normalize(Pt * pmt.unsqueeze(-1)).detach()
    big = torch.from_numpy(z["big"])
    if mode == "call_pattern_cached":
        ER.forget_prepared()
        ER.enable_score_cache(32 << 20)
        request.addfinalizer(lambda: (ER.disable_score_cache(), ER.forget_prepared()))      # also when an assertion below fails
        with torch.no_grad():
            for Qb, qmb in batches:                                 # the first epoch: every query row of the run is scored once
                ER.score_multi_vector_masked(Qb.to(dev), Ptn, qmb.to(dev), pmt)
        cache = next(c for c in ER._SCORE_CACHES.values() if c is not None)
        assert int(cache.n_entries.item()) == 4 * len(batches)
    if mode in ("call_pattern", "call_pattern_cached"):
        param = torch.nn.Parameter(Pbar0 * pms.unsqueeze(-1))
        opt = set_optimizer("adamw", param, hp["lr"], hp["wd"])
        step = lambda Qb, qmb: driver.train_one_step(Qb.to(dev), qmb.to(dev), Ptn, pmt, param, pms, opt, hp["temp"])
        cur = lambda: (param.detach(), opt.state[param]["exp_avg"], opt.state[param]["exp_avg_sq"])
    else:
        teacher = driver.TeacherScorer(Ptn, pmt)
        student = driver.FusedStudent(Pbar0, pms, lr=hp["lr"], weight_decay=hp["wd"])
        step = lambda Qb, qmb: driver.fused_train_one_step(Qb.to(dev), qmb.to(dev), teacher, student, hp["temp"])
        cur = lambda: (student.x, student.exp_avg, student.exp_avg_sq)
    for i, (Qb, qmb) in enumerate(batches, 1):
        loss = step(Qb, qmb)
        np.testing.assert_allclose(loss, z["losses"][i - 1], rtol=2e-5, err_msg=f"step {i}")
        if mode == "call_pattern_cached":
            assert int(cache.count.item()) == 0, i                 # nothing was scored: all four rows came out of the cache
        if i in (1, 4, 8):
            d = (cur()[0].cpu() - torch.from_numpy(z[f"param_after_{i}"])).abs()
            # tight wherever the reference's own gradient was well above its summation noise (or exactly zero) in every step;
            # elsewhere AdamW's m / (sqrt(v) + 1e-8) amplifies 1e-8-level noise, bounded by lr per step
            assert d[big].max().item() < 3e-6 * i ** 0.5 + 1e-6, (i, d[big].max().item())
            assert d.max().item() < 2 * i * hp["lr"]
    _, ea, es = cur()
    np.testing.assert_allclose(ea.cpu().numpy()[z["big"]], z["exp_avg"][z["big"]], atol=2e-6)
    np.testing.assert_allclose(es.cpu().numpy()[z["big"]], z["exp_avg_sq"][z["big"]], atol=1e-9, rtol=2e-3)
    if mode == "call_pattern_cached":
        assert int(cache.n_entries.item()) == 4 * len(batches)


def test_stream_adamw_takes_torchs_step_whole_when_anything_is_ineligible(dev):
    """eligibility is decided before anything is touched.  A parameter that is a view at an odd storage offset
    (not 16-byte aligned), a gradient that is not dense, or moments restored onto another device make the WHOLE step torch's
    own -- bit-equal to torch.optim.AdamW, step counters in step -- instead of failing after some parameters were updated."""
    from evdr_amd.utils.utils import StreamAdamW
    gen = torch.Generator().manual_seed(77)
    base = torch.randn(4 * 33 + 1, generator=gen).to(dev)
    good0, odd0 = torch.randn(64, 128, generator=gen).to(dev), base[1:].view(4, 33)        # odd0: data_ptr % 16 == 4
    assert odd0.data_ptr() % 16 != 0

    def pair(cls):
        a, b = torch.nn.Parameter(good0.clone()), torch.nn.Parameter(base.clone()[1:].view(4, 33))
        return a, b, cls([a, b], lr=1e-3, weight_decay=1e-2)
    a1, b1, o1 = pair(StreamAdamW)
    a2, b2, o2 = pair(torch.optim.AdamW)
    assert b1.data_ptr() % 16 != 0
    for i in range(3):
        ga, gb = torch.randn(64, 128, generator=gen).to(dev), torch.randn(4, 33, generator=gen).to(dev)
        a1.grad, b1.grad, a2.grad, b2.grad = ga.clone(), gb.clone(), ga.clone(), gb.clone()
        o1.step()
        o2.step()
        assert torch.equal(a1, a2) and torch.equal(b1, b2)                                  # torch's rule for BOTH parameters
        assert float(o1.state[a1]["step"]) == float(o1.state[b1]["step"]) == i + 1
    # a non-dense gradient: torch's step again
    a3 = torch.nn.Parameter(good0.clone())
    a4 = torch.nn.Parameter(good0.clone())
    o3, o4 = StreamAdamW([a3], lr=1e-3), torch.optim.AdamW([a4], lr=1e-3)
    g = torch.randn(128, 64, generator=gen).to(dev).t()
    a3.grad, a4.grad = g, g.clone(memory_format=torch.preserve_format)
    o3.step()
    o4.step()
    assert torch.equal(a3, a4)
    # moments restored as CPU tensors (a state dict loaded by hand): the kernel path must not be entered
    a5 = torch.nn.Parameter(good0.clone())
    o5 = StreamAdamW([a5], lr=1e-3)
    a5.grad = torch.ones_like(a5)
    o5.step()
    o5.state[a5]["exp_avg"] = o5.state[a5]["exp_avg"].cpu()
    assert not o5._eligible(a5)


def test_a6_full_gradient_identical_inputs(dev, ER):
    """Student pages normalised on the CPU and handed to both sides unchanged: scores, argmax and the FULL
    gradient w.r.t. the normalised pages must agree with the oracle (B=32, N=128, Ls=206)."""
    import evdr_amd.ops as ops
    from evdr_amd.criterion import infonce_distillation_loss
    Qb, qmb, Pt, pmt, Pbar0, pms, hp = R.train_case("b32n128")
    Ps = O.l2_normalize(Pbar0 * pms.unsqueeze(-1))
    T_ = torch.randn(32, 128, generator=torch.Generator().manual_seed(9)) * 3
    Po = Ps.clone().requires_grad_(True)
    so = O.maxsim_masked(Qb, Po, qmb, pms)
    lo = O.infonce_distill(so, T_, hp["temp"])
    lo.backward()
    Pd = Ps.to(dev).requires_grad_(True)
    sd = ER.score_multi_vector_masked(Qb.to(dev), Pd, qmb.to(dev), pms.to(dev))
    ld = infonce_distillation_loss(sd, T_.to(dev), hp["temp"])
    ld.backward()
    np.testing.assert_allclose(sd.detach().cpu().numpy(), so.detach().numpy(), atol=2e-5)
    np.testing.assert_allclose(ld.item(), lo.item(), rtol=1e-5)
    np.testing.assert_allclose(Pd.grad.cpu().numpy(), Po.grad.numpy(), atol=1e-6)
    _, arg_o = O.maxsim_masked_argmax(Qb, Ps, qmb, pms)
    _, arg_d = ops.maxsim_forward(Qb.to(dev), Ps.to(dev), qmb.to(dev), pms.to(dev), want_argmax=True)
    assert torch.equal(arg_d.cpu().to(torch.int32) & 0xFFFF, arg_o.to(torch.int32))


def test_a4_l2norm_kernels(golden, dev):
    """Fused normalise(+mask) forward/backward vs the reference's output (fixture) and vs torch autograd of the formula."""
    from evdr_amd.utils.preprocess_data import l2_normalize, normalize_masked
    z = golden("a4_l2norm")
    x = T(z["x"]).to(dev)
    np.testing.assert_allclose(l2_normalize(x).cpu().numpy(), z["y"], atol=1e-7, rtol=1e-6)
    assert torch.all(l2_normalize(x)[0, 0] == 0)
    gen = torch.Generator().manual_seed(3)
    xr = torch.randn(7, 33, 128, generator=gen)
    xr[2, 5] = 0.0
    m = torch.rand(7, 33, generator=gen) > 0.3
    gy = torch.randn(7, 33, 128, generator=gen)
    xo = xr.clone().requires_grad_(True)
    yo = O.l2_normalize(xo * m.unsqueeze(-1))
    (yo * gy).sum().backward()
    xd = xr.to(dev).requires_grad_(True)
    yd = normalize_masked(xd, m.to(dev))
    (yd * gy.to(dev)).sum().backward()
    np.testing.assert_allclose(yd.detach().cpu().numpy(), yo.detach().numpy(), atol=1e-7, rtol=1e-6)
    np.testing.assert_allclose(xd.grad.cpu().numpy(), xo.grad.numpy(), atol=1e-6, rtol=1e-5)
    assert torch.all(xd.grad.cpu()[~m] == 0) and torch.isfinite(xd.grad).all()


@pytest.mark.parametrize("case", ["small_ragged", "lq1", "chunk_tail"])
def test_a6_query_gradient(dev, ER, case):
    """d/dQ of the masked scorer (the reference's function is differentiable in Q too) vs autograd through the oracle."""
    Q, P, qm, pm, g = R.small_case(case)
    Qo, Po = Q.clone().requires_grad_(True), P.clone().requires_grad_(True)
    (O.maxsim_masked(Qo, Po, qm, pm) * g).sum().backward()
    Qd, Pd = Q.to(dev).requires_grad_(True), P.to(dev).requires_grad_(True)
    (ER.score_multi_vector_masked(Qd, Pd, qm.to(dev), pm.to(dev)) * g.to(dev)).sum().backward()
    np.testing.assert_allclose(Qd.grad.cpu().numpy(), Qo.grad.numpy(), atol=2e-6)
    np.testing.assert_allclose(Pd.grad.cpu().numpy(), Po.grad.numpy(), atol=2e-6)
    assert torch.all(Qd.grad.cpu()[~qm] == 0)                      # masked query tokens get exact zeros


# ---- top-k --------------------------------------------------------------------------------------
@pytest.mark.parametrize("n,k", [(500, 100), (37, 100), (100, 100), (5000, 10), (100001, 100), (1, 1), (20000, 128), (8193, 7)])
def test_topk_vs_oracle(dev, n, k):
    import evdr_amd.ops as ops
    gen = torch.Generator().manual_seed(n + k)
    s = torch.randn(9, n, generator=gen)
    s[0, : n // 2] = 1.5                       # heavy ties at the top
    s[1] = 0.0                                 # everything ties
    s[2, ::3] = float("-inf")
    s[3, : min(n, 7)] = -0.0                   # -0.0 ties with +0.0
    s[3, min(n, 7):] = -1.0
    if n > 3:
        s[3, 3] = 0.0
    ws, wi = O.topk_rows(s, k)
    gs, gi = ops.topk(s.to(dev), k)
    keff = min(k, n)
    assert torch.equal(gi.cpu()[:, :keff], wi)
    assert torch.equal(gs.cpu()[:, :keff] + 0.0, ws + 0.0)
    if keff < k:
        assert torch.all(gi.cpu()[:, keff:] == -1) and torch.all(torch.isinf(gs.cpu()[:, keff:]))


def test_topk_two_level_strided_and_mapped(dev):
    """Few long rows take the two-level form (per-segment candidates, then a merge): row-strided input, idx_map and
    idx_base must come out exactly as from one workgroup per row."""
    import evdr_amd.ops as ops
    gen = torch.Generator().manual_seed(12)
    big = torch.randn(3, 40000, generator=gen).to(dev)
    big[0, 100:30000:2] = 2.5                                            # ties that straddle segments
    s = big[:, 5:30006]                                                  # row stride 40000, rows not 16-B aligned
    ws, wi = O.topk_rows(s.cpu(), 100)
    gs, gi = ops.topk(s, 100, idx_base=7)
    assert torch.equal(gi.cpu(), wi + 7) and torch.equal(gs.cpu(), ws)
    sc = s.contiguous()
    idx_map = torch.randperm(sc.shape[1], generator=gen).to(torch.int32)[None, :].repeat(3, 1).to(dev)
    ms, mi = ops.topk(sc, 50, idx_map=idx_map)
    # reference: rank by (score desc, mapped index asc)
    for r in range(3):
        order = sorted(range(sc.shape[1]), key=lambda j: (-float(sc[r, j] + 0.0), int(idx_map[r, j])))[:50]
        assert mi[r].tolist() == [int(idx_map[r, j]) for j in order]


def test_topk_idx_map_and_base(dev):
    import evdr_amd.ops as ops
    gen = torch.Generator().manual_seed(1)
    s = torch.randn(4, 300, generator=gen)
    s[:, 100:200] = s[:, 0:100]                # duplicates -> ties resolved by the REPORTED index
    m = torch.arange(300).repeat(4, 1).to(torch.int32) + 7000
    gs, gi = ops.topk(s.to(dev), 50, idx_map=m.to(dev))
    ws, wi = O.topk_rows(s, 50)
    assert torch.equal(gi.cpu(), wi + 7000)
    gs2, gi2 = ops.topk(s.to(dev), 50, idx_base=123)
    assert torch.equal(gi2.cpu(), wi + 123)
    # strided rows: a column block of a wider matrix
    wide = torch.randn(4, 1000, generator=gen).to(dev)
    a, b = ops.topk(wide[:, 200:700], 20, idx_base=200)
    c, d = O.topk_rows(wide.cpu()[:, 200:700], 20)
    assert torch.equal(b.cpu(), d + 200)


# ---- resident corpus / retrieval --------------------------------------------------------------------
def test_corpus_score_and_topk(dev):
    from evdr_amd.corpus import PageCorpus
    gen = torch.Generator().manual_seed(5)
    P = torch.nn.functional.normalize(torch.randn(300, 70, 128, generator=gen), dim=-1).bfloat16()
    Q = torch.nn.functional.normalize(torch.randn(40, 32, 128, generator=gen), dim=-1).bfloat16()
    pm = torch.rand(300, 70, generator=gen) > 0.1
    qm = torch.rand(40, 32, generator=gen) > 0.1
    want = O.maxsim_masked(Q.float(), P.float(), qm, pm)
    corpus = PageCorpus.from_tensor(P.to(dev), pm.to(dev), idx_base=1000)
    got = corpus.score(Q.to(dev), qm.to(dev))
    np.testing.assert_allclose(got.cpu().numpy(), want.numpy(), atol=SCORE_ATOL)
    ts, ti = corpus.topk(Q.to(dev), qm.to(dev), 100)
    ws, wi = O.topk_rows(got.cpu(), 100)               # ranking of the device scores themselves: bit-exact
    assert torch.equal(ti.cpu(), wi + 1000) and torch.equal(ts.cpu(), ws)
    # fp32 corpus (fp16 hi/lo planes) with fp32 queries
    c32 = PageCorpus.from_tensor(P.float().to(dev), pm.to(dev))
    got32 = c32.score(Q.float().to(dev), qm.to(dev))
    np.testing.assert_allclose(got32.cpu().numpy(), want.numpy(), atol=SCORE_ATOL)
    with pytest.raises(RuntimeError, match="bf16 corpus needs bf16 queries"):
        corpus.score(Q.float().to(dev), qm.to(dev))


def test_sharded_merge_equals_single_shard(dev):
    """Shards scored separately + candidate merge == one big shard (the N>1 data path, minus the wire)."""
    from evdr_amd.corpus import PageCorpus, shard_range, pack_candidates, unpack_candidates, merge_candidates
    gen = torch.Generator().manual_seed(6)
    npg, world, k = 203, 4, 50
    P = torch.nn.functional.normalize(torch.randn(npg, 40, 128, generator=gen), dim=-1).bfloat16().to(dev)
    P[50:60] = P[150:160]                                  # cross-shard exact ties
    Q = torch.nn.functional.normalize(torch.randn(17, 32, 128, generator=gen), dim=-1).bfloat16().to(dev)
    full = PageCorpus.from_tensor(P)
    fs, fi = full.topk(Q, None, k)
    msgs = []
    for r in range(world):
        lo, hi = shard_range(npg, r, world)
        sh = PageCorpus.from_tensor(P[lo:hi], None, idx_base=lo)
        msgs.append(pack_candidates(*sh.topk(Q, None, k)))
    sc, ix = unpack_candidates(torch.stack(msgs))
    ms, mi = merge_candidates(sc, ix, k)
    assert torch.equal(mi, fi) and torch.equal(ms, fs)


# ---- stream and graph behaviour promised by include/evdr.h ---------------------------------------------------------
def test_runs_on_callers_stream_without_sync(dev, ER):
    """Work is enqueued on torch's CURRENT stream (SURVEY §8(b) threading): results computed on a side stream are
    correct once that stream is waited on, and inputs produced on that stream are consumed in order."""
    gen = torch.Generator().manual_seed(21)
    Q = torch.nn.functional.normalize(torch.randn(24, 32, 128, generator=gen), dim=-1).bfloat16().to(dev)
    P = torch.nn.functional.normalize(torch.randn(96, 300, 128, generator=gen), dim=-1).bfloat16().to(dev)
    qm = torch.ones(24, 32, dtype=torch.bool, device=dev)
    pm = torch.ones(96, 300, dtype=torch.bool, device=dev)
    want = ER.score_multi_vector_masked(Q, P, qm, pm).clone()
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        P2 = P * 2                                        # producer on the side stream
        got = ER.score_multi_vector_masked(Q, P2, qm, pm) # must run after it, on the same stream
        back = got / 2
    torch.cuda.current_stream(dev).wait_stream(side)
    np.testing.assert_allclose(back.cpu().numpy(), want.cpu().numpy(), atol=2e-5)


def test_hip_graph_capture_and_replay(dev):
    """The prepared-corpus scorer and the top-k launch nothing but kernels (no allocation, no sync inside the C ABI):
    a captured graph replays them and tracks updated inputs."""
    import evdr_amd.ops as ops
    from evdr_amd.corpus import PageCorpus
    gen = torch.Generator().manual_seed(22)
    P = torch.nn.functional.normalize(torch.randn(200, 260, 128, generator=gen), dim=-1).bfloat16().to(dev)
    corpus = PageCorpus.from_tensor(P)
    Qs = torch.nn.functional.normalize(torch.randn(40, 32, 128, generator=gen), dim=-1).bfloat16().to(dev)   # static input
    out = torch.empty((40, 200), dtype=torch.float32, device=dev)
    corpus.score(Qs, None, out=out)                        # warm-up outside capture (kernel attributes, lib load)
    ops.topk(out, 10)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        corpus.score(Qs, None, out=out)
        ts, ti = ops.topk(out, 10)
    Qn = torch.nn.functional.normalize(torch.randn(40, 32, 128, generator=gen), dim=-1).bfloat16().to(dev)
    Qs.copy_(Qn)
    g.replay()
    torch.cuda.synchronize()
    want = O.maxsim_masked(Qn.float().cpu(), P.float().cpu(), torch.ones(40, 32, dtype=torch.bool), torch.ones(200, 260, dtype=torch.bool))
    np.testing.assert_allclose(out.cpu().numpy(), want.numpy(), atol=SCORE_ATOL)
    ws, wi = O.topk_rows(out.cpu(), 10)
    assert torch.equal(ti.cpu(), wi)
