"""Queries longer than one 32-token MFMA tile: the C ABI scores them in 32-token slices (later slices add into `out`, only for
the queries that have a valid token there) and admits lq <= 65535 like lp.  Against the oracle: lq in {33 ... 4097} for the bf16
and fp32 paths with random and padded-tail query masks, arg-max + both backward kernels at 100 tokens, and lq = 65535.
Scores of long queries are sums of hundreds to thousands of maxima (640 at 4097 tokens): the tolerance is the 1e-4 gate plus 2e-6
relative (fp32 sums in a different order: the slices are accumulated with atomics)."""
import numpy as np
import pytest
import torch

from oracle import maxsim_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _unit(shape, gen):
    return torch.nn.functional.normalize(torch.randn(*shape, generator=gen), dim=-1)


@pytest.mark.parametrize("lq", [33, 64, 100, 513, 4097])
@pytest.mark.parametrize("nq", [1, 7, 40])
def test_long_queries_forward(lq, nq):
    import evdr_amd  # noqa: F401
    from evdr_amd import ops
    from evdr_amd.corpus import PageCorpus
    gen = torch.Generator().manual_seed(lq * 100 + nq)
    npg, lp = 11, 150
    Q, P = _unit((nq, lq, 128), gen), _unit((npg, lp, 128), gen)
    pm = torch.rand(npg, lp, generator=gen) > 0.2
    pm[2] = False
    lens = torch.randint(1, lq + 1, (nq,), generator=gen)
    lens[0] = lq
    masks = {"random": torch.rand(nq, lq, generator=gen) > 0.3,
             "padded tails (a query set padded to its longest member)": torch.arange(lq)[None, :] < lens[:, None],
             "none": None}
    for name, qm in masks.items():
        qmo = qm if qm is not None else torch.ones(nq, lq, dtype=torch.bool)
        qmd = qm.to(DEV) if qm is not None else None
        want = O.maxsim_masked(Q, P, qmo, pm)
        s32, _ = ops.maxsim_forward(Q.to(DEV), P.to(DEV), qmd, pm.to(DEV))
        np.testing.assert_allclose(s32.cpu().numpy(), want.numpy(), atol=1e-4, rtol=2e-6, err_msg=f"fp32 {name}")
        Qb, Pb = Q.bfloat16(), P.bfloat16()
        wb = O.maxsim_masked(Qb.float(), Pb.float(), qmo, pm)
        sb, _ = ops.maxsim_forward(Qb.to(DEV), Pb.to(DEV), qmd, pm.to(DEV))
        np.testing.assert_allclose(sb.cpu().numpy(), wb.numpy(), atol=1e-4, rtol=2e-6, err_msg=f"bf16 {name}")
        sc = PageCorpus.from_tensor(Pb.to(DEV), pm.to(DEV)).score(Qb.to(DEV), qmd)            # prepared entry: compacted query list
        np.testing.assert_allclose(sc.cpu().numpy(), wb.numpy(), atol=1e-4, rtol=2e-6, err_msg=f"resident {name}")


def test_long_queries_argmax_and_backward():
    import evdr_amd  # noqa: F401
    from evdr_amd import ops
    gen = torch.Generator().manual_seed(4242)
    nq, lq, npg, lp = 9, 100, 13, 77
    Q, P = _unit((nq, lq, 128), gen), _unit((npg, lp, 128), gen)
    qm = torch.rand(nq, lq, generator=gen) > 0.25
    pm = torch.rand(npg, lp, generator=gen) > 0.2
    g = torch.randn(nq, npg, generator=gen)
    s_o, arg_o = O.maxsim_masked_argmax(Q, P, qm, pm)
    s, arg = ops.maxsim_forward(Q.to(DEV), P.to(DEV), qm.to(DEV), pm.to(DEV), want_argmax=True)
    np.testing.assert_allclose(s.cpu().numpy(), s_o.numpy(), atol=1e-4, rtol=0)
    assert torch.equal(arg.cpu().to(torch.int32) & 0xFFFF, arg_o.to(torch.int32))
    dP = ops.maxsim_backward(g.to(DEV), Q.to(DEV), qm.to(DEV), pm.to(DEV), arg, npg, lp)
    np.testing.assert_allclose(dP.cpu().numpy(), O.maxsim_backward(g, Q, P, qm, pm).numpy(), atol=2e-5, rtol=1e-5)
    Qg = Q.clone().requires_grad_(True)
    (O.maxsim_masked(Qg, P, qm, pm) * g).sum().backward()
    dQ = ops.maxsim_backward_q(g.to(DEV), P.to(DEV), qm.to(DEV), pm.to(DEV), arg, nq, lq)
    np.testing.assert_allclose(dQ.cpu().numpy(), Qg.grad.numpy(), atol=2e-5, rtol=1e-5)


def test_query_length_at_the_abi_bound():
    import evdr_amd  # noqa: F401
    from evdr_amd import ops
    gen = torch.Generator().manual_seed(65535)
    nq, lq, npg, lp = 2, 65535, 3, 40
    Q, P = _unit((nq, lq, 128), gen), _unit((npg, lp, 128), gen)
    qm = torch.rand(nq, lq, generator=gen) > 0.5
    qm[1, 1000:] = False                                           # a short query inside a set padded to 65535 tokens
    pm = torch.rand(npg, lp, generator=gen) > 0.2
    want = O.maxsim_masked(Q, P, qm, pm)
    got, _ = ops.maxsim_forward(Q.to(DEV), P.to(DEV), qm.to(DEV), pm.to(DEV))
    np.testing.assert_allclose(got.cpu().numpy(), want.numpy(), rtol=2e-6, atol=1e-3)      # sums of ~30 000 maxima: scores ~ 1e4
    with pytest.raises((NotImplementedError, RuntimeError)):
        ops.maxsim_forward(torch.zeros(1, 65536, 128, device=DEV), P.to(DEV), None, pm.to(DEV))
