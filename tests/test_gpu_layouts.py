"""The page-mask layouts that utils/preprocess_data.py:101 produces (pmask = valid AND attention AND image mask) against the
oracle, for every query-count regime of the forward dispatch: 1-4 queries (one per wave, top-of-stage refill, non-temporal
stream), 5-8 (two 4-wave workgroups per CU), 9-16, 17-32 and several query groups -- bf16 (with and without argmax) and
fp32.  1030-patch pages, so that the page-range stage walk, the partial-first-tile block and the rolled boundary loop all
run (csrc/maxsim_fwd16.hip)."""
import numpy as np
import pytest
import torch

from oracle import maxsim_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
LP = 1030


def layouts(npg, gen):
    ar = torch.arange(LP)[None, :]
    full = torch.ones(npg, LP, dtype=torch.bool)
    lens = torch.randint(1, LP + 1, (npg,), generator=gen)
    yield "ragged prefix 1..1030", ar < lens[:, None]
    m = full.clone(); m[:, :4] = False
    yield "4 masked in front", m
    m = full.clone(); m[:, :5] = False; m[:, -1] = False
    yield "image in the middle (5 in front, 1 behind)", m
    front = torch.randint(0, 300, (npg,), generator=gen)
    yield "ragged start 0..299 and ragged end", (ar >= front[:, None]) & (ar < torch.maximum(lens, front + 1)[:, None])
    m = (ar >= 2) & (ar < lens[:, None]); m[:, 500] = False
    yield "range with ONE hole (mask words decide) and a partial head", m
    m = full.clone(); m[::3] = False; m[1::3, 40:] = False
    yield "all-masked pages, 40-patch pages, full pages interleaved", m


@pytest.mark.parametrize("nq", [1, 3, 6, 8, 12, 20, 40])
def test_layouts_bf16(nq):
    import evdr_amd  # noqa: F401
    from evdr_amd import ops
    from evdr_amd.corpus import PageCorpus
    gen = torch.Generator().manual_seed(100 + nq)
    npg = 48
    Q = torch.nn.functional.normalize(torch.randn(nq, 32, 128, generator=gen), dim=-1).bfloat16()
    P = torch.nn.functional.normalize(torch.randn(npg, LP, 128, generator=gen), dim=-1).bfloat16()
    qm = torch.rand(nq, 32, generator=gen) > 0.2
    for name, pm in layouts(npg, gen):
        want, warg = O.maxsim_masked_argmax(Q.float(), P.float(), qm, pm)
        got = PageCorpus.from_tensor(P.to(DEV), pm.to(DEV)).score(Q.to(DEV), qm.to(DEV)).cpu()      # prepared entry, no argmax
        np.testing.assert_allclose(got.numpy(), want.numpy(), atol=1e-4, rtol=0, err_msg=f"{name} nq={nq}")
        s2, arg = ops.maxsim_forward(Q.to(DEV), P.to(DEV), qm.to(DEV), pm.to(DEV), want_argmax=True)
        np.testing.assert_allclose(s2.cpu().numpy(), want.numpy(), atol=1e-4, rtol=0, err_msg=f"{name} nq={nq} (argmax kernel)")
        assert torch.equal(arg.cpu().to(torch.int32) & 0xFFFF, warg.to(torch.int32)), f"{name} nq={nq}: argmax"


@pytest.mark.parametrize("nq", [2, 7, 24])
def test_layouts_fp32(nq):
    import evdr_amd  # noqa: F401
    from evdr_amd import ops
    gen = torch.Generator().manual_seed(200 + nq)
    npg = 24
    Q = torch.nn.functional.normalize(torch.randn(nq, 32, 128, generator=gen), dim=-1)
    P = torch.nn.functional.normalize(torch.randn(npg, LP, 128, generator=gen), dim=-1)
    qm = torch.rand(nq, 32, generator=gen) > 0.2
    for name, pm in layouts(npg, gen):
        want, warg = O.maxsim_masked_argmax(Q, P, qm, pm)
        s, arg = ops.maxsim_forward(Q.to(DEV), P.to(DEV), qm.to(DEV), pm.to(DEV), want_argmax=True)
        np.testing.assert_allclose(s.cpu().numpy(), want.numpy(), atol=1e-4, rtol=0, err_msg=f"{name} nq={nq}")
        assert torch.equal(arg.cpu().to(torch.int32) & 0xFFFF, warg.to(torch.int32)), f"{name} nq={nq}: argmax"
        s0, _ = ops.maxsim_forward(Q.to(DEV), P.to(DEV), qm.to(DEV), pm.to(DEV))
        np.testing.assert_allclose(s0.cpu().numpy(), want.numpy(), atol=1e-4, rtol=0, err_msg=f"{name} nq={nq} (no argmax)")


@pytest.mark.parametrize("d", [32, 64, 96, 100])
def test_embedding_widths_below_128(d):
    """The reference scores any embedding width (evaluator/retrieval.py:166-213); the kernels are built for 128, and narrower
    embeddings ride on zero columns, which add exact zeros to every dot product: scores, arg-max and gradients (w.r.t. pages
    and queries, cut back to d columns) against the oracle on the narrow tensors; the list scorer too.  Widths of 129..256 run on
    two column blocks (tests/test_gpu_width.py); beyond 256 raises."""
    import evdr_amd  # noqa: F401
    from evdr_amd.evaluator.retrieval import BaseVisualRetrieverProcessor, score_multi_vector_masked
    gen = torch.Generator().manual_seed(300 + d)
    nq, lq, npg, lp = 9, 20, 30, 77
    Q = torch.nn.functional.normalize(torch.randn(nq, lq, d, generator=gen), dim=-1)
    P = torch.nn.functional.normalize(torch.randn(npg, lp, d, generator=gen), dim=-1)
    qm = torch.rand(nq, lq, generator=gen) > 0.2
    pm = torch.rand(npg, lp, generator=gen) > 0.2
    pm[4] = False
    up = torch.randn(nq, npg, generator=gen)
    Qo, Po = Q.clone().requires_grad_(True), P.clone().requires_grad_(True)
    so = O.maxsim_masked(Qo, Po, qm, pm)
    so.backward(up)
    Qd, Pd = Q.to(DEV).requires_grad_(True), P.to(DEV).requires_grad_(True)
    s = score_multi_vector_masked(Qd, Pd, qm.to(DEV), pm.to(DEV))
    s.backward(up.to(DEV))
    np.testing.assert_allclose(s.detach().cpu().numpy(), so.detach().numpy(), atol=1e-4, rtol=0)
    assert Pd.grad.shape == P.shape and Qd.grad.shape == Q.shape
    np.testing.assert_allclose(Pd.grad.cpu().numpy(), Po.grad.numpy(), atol=2e-5, rtol=1e-5)
    np.testing.assert_allclose(Qd.grad.cpu().numpy(), Qo.grad.numpy(), atol=2e-5, rtol=1e-5)
    with torch.no_grad():                                            # frozen pages (prepared once per tensor) and bf16
        s2 = score_multi_vector_masked(Q.to(DEV), P.to(DEV), qm.to(DEV), pm.to(DEV))
        np.testing.assert_allclose(s2.cpu().numpy(), so.detach().numpy(), atol=1e-4, rtol=0)
        sb = score_multi_vector_masked(Q.bfloat16().to(DEV), P.bfloat16().to(DEV), qm.to(DEV), pm.to(DEV))
        want_b = O.maxsim_masked(Q.bfloat16().float(), P.bfloat16().float(), qm, pm)
        np.testing.assert_allclose(sb.cpu().numpy(), want_b.numpy(), atol=1e-4, rtol=0)
    qs = [Q[i, : 3 + i] for i in range(5)]
    ps = [P[j, : 10 + 3 * j] for j in range(7)]
    got = BaseVisualRetrieverProcessor.score_multi_vector(qs, ps, batch_size=4, device=DEV)
    np.testing.assert_allclose(got.numpy(), O.maxsim_unmasked_lists(qs, ps, batch_size=4).numpy(), atol=1e-4, rtol=0)
    with pytest.raises(NotImplementedError):
        score_multi_vector_masked(torch.zeros(2, 3, 257, device=DEV), torch.zeros(2, 5, 257, device=DEV),
                                  torch.ones(2, 3, device=DEV), torch.ones(2, 5, device=DEV))
