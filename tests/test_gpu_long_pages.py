"""Pages longer than the 1030-patch benchmark shape: the reference accepts any Lp
(evaluator/retrieval.py:166-213) and the C ABI admits lp <= 65535 (csrc/evdr_capi.hip check_common), so every regime the
ABI admits is checked against the oracle here:

  * lp in {1057, 2048, 4100, 8200}: more than 33 tiles through the stage cursors, bf16 and fp32 (fp16 hi/lo planes), every
    queries-per-wave regime of the dispatch;
  * a page whose first valid patch sits at index >= 4096 (the one-range flag's `va < 4096` fallback, csrc/prep.hip: the
    mask words decide), a ragged prefix that ends inside the last tile, holes beyond patch 4096, all-masked pages;
  * lp = 65535 (the ABI's bound) with few pages: first-masked index and argmax indices above 2^15 (argmax is uint16),
    forward + argmax + backward gather.
"""
import numpy as np
import pytest
import torch

from oracle import maxsim_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _unit(shape, gen):
    return torch.nn.functional.normalize(torch.randn(*shape, generator=gen), dim=-1)


def long_layouts(npg, lp, gen):
    """Mask layouts for npg >= 6 pages of lp patches (utils/preprocess_data.py:101 produces the range-style ones)."""
    ar = torch.arange(lp)[None, :]
    yield "all valid", torch.ones(npg, lp, dtype=torch.bool)
    last_tile = (lp - 1) // 32 * 32
    lens = torch.randint(last_tile + 1, lp + 1, (npg,), generator=gen)
    lens[0] = last_tile + 1                                      # exactly one patch in the last tile
    lens[1] = lp
    yield "ragged prefix ending in the last tile", ar < lens[:, None]
    lens = torch.randint(1, lp + 1, (npg,), generator=gen)
    yield "ragged prefix anywhere", ar < lens[:, None]
    front = torch.randint(0, lp - 1, (npg,), generator=gen)
    if lp > 4100:
        front[0], front[1], front[2] = 4096, 4095, lp - 1        # va on both sides of the 12-bit field, and a 1-patch page
    else:
        front[0], front[1] = lp - 1, lp // 2
    end = torch.maximum(torch.randint(1, lp + 1, (npg,), generator=gen), front + 1)
    end[0] = lp
    yield "valid range [va, vb) with va up to lp - 1", (ar >= front[:, None]) & (ar < end[:, None])
    m = torch.rand(npg, lp, generator=gen) > 0.3
    m[0, : lp // 2] = False                                      # holes only beyond the middle / beyond 4096 on long pages
    m[1] = True
    m[1, lp - 3] = False                                         # ONE hole, in the last tile
    m[2] = False                                                 # all-masked page
    m[3] = False
    m[3, lp - 1] = True                                          # only the very last patch valid
    yield "holes", m


@pytest.mark.parametrize("lp", [1057, 2048, 4100, 8200])
@pytest.mark.parametrize("nq", [1, 6, 12, 20, 40])
def test_long_pages_bf16(lp, nq):
    import evdr_amd  # noqa: F401
    from evdr_amd import ops, _lib as L
    from evdr_amd.corpus import PageCorpus
    gen = torch.Generator().manual_seed(7000 + lp + nq)
    npg = 9
    Q, P = _unit((nq, 32, 128), gen).bfloat16(), _unit((npg, lp, 128), gen).bfloat16()
    qm = torch.rand(nq, 32, generator=gen) > 0.2
    lib = L.load()
    for name, pm in long_layouts(npg, lp, gen):
        want = O.maxsim_masked(Q.float(), P.float(), qm, pm, chunk_p=3)
        got = PageCorpus.from_tensor(P.to(DEV), pm.to(DEV)).score(Q.to(DEV), qm.to(DEV)).cpu()
        np.testing.assert_allclose(got.numpy(), want.numpy(), atol=1e-4, rtol=0, err_msg=f"{name} lp={lp} nq={nq}")
        for variant in (1, 2, 30, 31):           # flat ring forced, no priority schedule, 8-wave workgroups only, no nt stream
            lib.evdr_debug_set_fwd_variant(variant)
            try:
                g2, _ = ops.maxsim_forward(Q.to(DEV), P.to(DEV), qm.to(DEV), pm.to(DEV))
            finally:
                lib.evdr_debug_set_fwd_variant(0)
            np.testing.assert_allclose(g2.cpu().numpy(), want.numpy(), atol=1e-4, rtol=0,
                                       err_msg=f"{name} lp={lp} nq={nq} variant {variant}")


@pytest.mark.parametrize("lp", [1057, 2048, 4100, 8200])
@pytest.mark.parametrize("nq", [2, 7, 24])
def test_long_pages_fp32_with_argmax(lp, nq):
    import evdr_amd  # noqa: F401
    from evdr_amd import ops
    gen = torch.Generator().manual_seed(8000 + lp + nq)
    npg = 7
    Q, P = _unit((nq, 32, 128), gen), _unit((npg, lp, 128), gen)
    qm = torch.rand(nq, 32, generator=gen) > 0.2
    for name, pm in long_layouts(npg, lp, gen):
        want, warg = O.maxsim_masked_argmax(Q, P, qm, pm)
        s, arg = ops.maxsim_forward(Q.to(DEV), P.to(DEV), qm.to(DEV), pm.to(DEV), want_argmax=True)
        np.testing.assert_allclose(s.cpu().numpy(), want.numpy(), atol=1e-4, rtol=0, err_msg=f"{name} lp={lp} nq={nq}")
        assert torch.equal(arg.cpu().to(torch.int32) & 0xFFFF, warg.to(torch.int32)), f"{name} lp={lp} nq={nq}: argmax"
        s0, _ = ops.maxsim_forward(Q.to(DEV), P.to(DEV), qm.to(DEV), pm.to(DEV))
        np.testing.assert_allclose(s0.cpu().numpy(), want.numpy(), atol=1e-4, rtol=0, err_msg=f"{name} lp={lp} nq={nq} (no argmax)")


@pytest.mark.parametrize("lp", [4100, 8200])
def test_long_pages_argmax_and_backward(lp):
    """bf16 and fp32 forward with argmax, then the backward gather (dP) and the query-side gradient on long pages: the
    backward owns a page in 128-row slabs (65 slabs at 8200 patches)."""
    import evdr_amd  # noqa: F401
    from evdr_amd import ops
    gen = torch.Generator().manual_seed(9000 + lp)
    nq, npg = 9, 6
    Q, P = _unit((nq, 32, 128), gen).bfloat16(), _unit((npg, lp, 128), gen).bfloat16()
    qm = torch.rand(nq, 32, generator=gen) > 0.2
    g = torch.randn(nq, npg, generator=gen)
    for name, pm in long_layouts(npg, lp, gen):
        Qf, Pf = Q.float(), P.float()
        s_o, arg_o = O.maxsim_masked_argmax(Qf, Pf, qm, pm)
        dP_o = O.maxsim_backward(g, Qf, Pf, qm, pm)
        for Qx, Px in ((Q, P), (Qf, Pf)):
            s, arg = ops.maxsim_forward(Qx.to(DEV), Px.to(DEV), qm.to(DEV), pm.to(DEV), want_argmax=True)
            np.testing.assert_allclose(s.cpu().numpy(), s_o.numpy(), atol=1e-4, rtol=0, err_msg=name)
            assert torch.equal(arg.cpu().to(torch.int32) & 0xFFFF, arg_o.to(torch.int32)), f"{name}: argmax"
            dP = ops.maxsim_backward(g.to(DEV), Qf.to(DEV), qm.to(DEV), pm.to(DEV), arg, npg, lp)
            np.testing.assert_allclose(dP.cpu().numpy(), dP_o.numpy(), atol=2e-5, rtol=1e-5, err_msg=name)
            assert torch.all(dP.cpu()[~pm] == 0)
        # query side: dQ through autograd of the oracle
        Qg = Qf.clone().requires_grad_(True)
        (O.maxsim_masked(Qg, Pf, qm, pm) * g).sum().backward()
        dQ = ops.maxsim_backward_q(g.to(DEV), Pf.to(DEV), qm.to(DEV), pm.to(DEV), arg, nq, 32)
        np.testing.assert_allclose(dQ.cpu().numpy(), Qg.grad.numpy(), atol=2e-5, rtol=1e-5, err_msg=f"{name}: dQ")


@pytest.mark.parametrize("dtype", ["bf16", "f32"])
def test_page_length_at_the_abi_bound(dtype):
    """lp = 65535, the largest page the C ABI admits: valid ranges that start and end above 2^15, a first masked patch at
    65534, holes in the last tiles; forward, argmax (uint16 indices up to 65534) and the backward gather."""
    import evdr_amd  # noqa: F401
    from evdr_amd import ops
    lp, npg, nq, lq = 65535, 5, 5, 17
    gen = torch.Generator().manual_seed(65535)
    Q, P = _unit((nq, lq, 128), gen), _unit((npg, lp, 128), gen)
    if dtype == "bf16":
        Q, P = Q.bfloat16(), P.bfloat16()
    qm = torch.rand(nq, lq, generator=gen) > 0.2
    ar = torch.arange(lp)
    pm = torch.ones(npg, lp, dtype=torch.bool)
    pm[0, lp - 1] = False                                        # first masked patch = 65534
    pm[1] = ar >= 40000                                          # valid range far above the 12-bit va field
    pm[2] = (ar >= 33000) & (ar < 65000) & (ar % 7 != 0)         # holes, all above 2^15
    pm[3] = False                                                # all-masked
    pm[4] = ar < 3                                               # a 3-patch page inside a 65535-patch corpus
    Qf, Pf = Q.float(), P.float()
    want, warg = O.maxsim_masked_argmax(Qf, Pf, qm, pm)
    s, arg = ops.maxsim_forward(Q.to(DEV), P.to(DEV), qm.to(DEV), pm.to(DEV), want_argmax=True)
    np.testing.assert_allclose(s.cpu().numpy(), want.numpy(), atol=1e-4, rtol=0)
    got_arg = arg.cpu().to(torch.int32) & 0xFFFF
    assert torch.equal(got_arg, warg.to(torch.int32))
    assert int(got_arg.max()) > 32768                            # indices beyond int16's positive range were really produced
    s0, _ = ops.maxsim_forward(Q.to(DEV), P.to(DEV), qm.to(DEV), pm.to(DEV))
    np.testing.assert_allclose(s0.cpu().numpy(), want.numpy(), atol=1e-4, rtol=0)
    g = torch.randn(nq, npg, generator=gen)
    dP_o = O.maxsim_backward(g, Qf, Pf, qm, pm)
    dP = ops.maxsim_backward(g.to(DEV), Qf.to(DEV), qm.to(DEV), pm.to(DEV), arg, npg, lp).cpu()
    np.testing.assert_allclose(dP.numpy(), dP_o.numpy(), atol=2e-5, rtol=1e-5)
    assert torch.all(dP[~pm] == 0)


def test_page_length_above_the_abi_bound_is_rejected():
    """lp = 65536 does not fit the uint16 argmax / the 16-bit fields of the page flag word: the ABI says so (status, no
    launch) and the Python layer raises."""
    import evdr_amd  # noqa: F401
    from evdr_amd import ops
    Q = torch.zeros(1, 4, 128, dtype=torch.bfloat16, device=DEV)
    P = torch.zeros(1, 65536, 128, dtype=torch.bfloat16, device=DEV)
    with pytest.raises((NotImplementedError, RuntimeError)):
        ops.maxsim_forward(Q, P, None, None)
