"""The InfoNCE-distillation loss folded into the training step's two big launches: the student forward's
last workgroup computes the row statistics + the mean loss, the fused update forms its gradient column from them
(/root/reference/criterion.py:56-68, mainv2_iter_distill_infonce.py:286-291).  Gate: the SAME BITS as the three-launch form
(forward -> evdr_infonce_distill_fwd_bwd_ws -> update): scores, arg-max, loss, and after every step the parameter, both moments and
the planes left for the next forward."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu


def _unit(gen, *shape):
    return torch.nn.functional.normalize(torch.randn(*shape, generator=gen), dim=-1)


@pytest.mark.parametrize("b,n,ls,lq", [(32, 500, 206, 32), (5, 37, 70, 20), (256, 40, 64, 32), (1, 9, 33, 2), (32, 1000, 40, 32), (33, 1024, 33, 7)])
def test_folded_step_equals_three_launch_step_bit_for_bit(b, n, ls, lq):
    import evdr_amd  # noqa: F401
    from evdr_amd import driver, ops
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(1000 * b + n)
    P0 = (torch.randn(n, ls, 128, generator=gen) * 0.5).to(dev)
    pm = (torch.rand(n, ls, generator=gen) > 0.2)
    pm[0] = True
    if n > 3:
        pm[3] = False                                            # a page without a valid patch
    pm = pm.to(dev)
    students = {}
    for fold in (False, True):
        st = driver.FusedStudent(P0.clone(), pm, lr=1e-3, weight_decay=1e-2)
        st.fold_loss = fold
        students[fold] = st
    for step in range(3):
        Q = _unit(gen, b, lq, 128).to(dev)
        qm = (torch.rand(b, lq, generator=gen) > 0.15)
        qm[:, 0] = True
        qm = qm.to(dev)
        sc_t = (torch.randn(b, n, generator=gen) * 3.0).to(dev)
        if step == 1 and n > 5:
            sc_t[0, 2] = sc_t[0, 5] = sc_t[0].max() + 1.0         # a tie at the teacher's top: the first index wins
        losses = {}
        for fold, st in students.items():
            losses[fold] = st.update(Q, qm, sc_t, 0.1).clone()
        assert torch.equal(losses[True], losses[False]), (step, float(losses[True]), float(losses[False]))
        a, f = students[False], students[True]
        for name in ("x", "exp_avg", "exp_avg_sq"):
            assert torch.equal(getattr(a, name), getattr(f, name)), (step, name)
        assert torch.equal(a._planes[0].view(torch.int16), f._planes[0].view(torch.int16)) and torch.equal(a.pageflags, f.pageflags)
    # the folded form really ran: its workspace exists and its ticket is back at zero
    ws = students[True]._fold_ws.get(b)
    assert ws is not None and int(ws[1].item()) == 0 and not students[False]._fold_ws
    # the statistics are the loss kernel's: row loss column and the loss itself against evdr_infonce_distill_fwd_bwd
    sc_s, arg = students[False].scores(Q, qm)
    sc_f, arg_f, loss_f = ops.maxsim_forward_nce(*ops.split_f32(Q), *students[True].planes(), qm, students[True].tilemask,
                                                 students[True].pageflags, sc_t, 0.1, students[True].fold_workspace(b))
    loss, dscore = ops.infonce_distill(sc_s, sc_t, 0.1, want_grad=True)
    assert torch.equal(sc_s, sc_f) and torch.equal(arg, arg_f) and torch.equal(loss, loss_f)
    stats = ws[0]
    tidx = stats[:, 2].contiguous().view(torch.int32).long()
    assert torch.equal(tidx, sc_t.argmax(dim=1))


def test_shapes_beyond_the_folded_form_take_the_three_launch_step():
    import evdr_amd  # noqa: F401
    from evdr_amd import driver
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(3)
    n, ls, b = 1100, 33, 8                                       # more pages than the tail job plays in registers
    P0 = (torch.randn(n, ls, 128, generator=gen) * 0.5).to(dev)
    pm = torch.ones(n, ls, dtype=torch.bool, device=dev)
    st = driver.FusedStudent(P0.clone(), pm, lr=1e-3, weight_decay=1e-2)
    ref = driver.FusedStudent(P0.clone(), pm, lr=1e-3, weight_decay=1e-2)
    ref.fold_loss = False
    Q = _unit(gen, b, 32, 128).to(dev)
    qm = torch.ones(b, 32, dtype=torch.bool, device=dev)
    sc_t = torch.randn(b, n, generator=gen).to(dev)
    assert torch.equal(st.update(Q, qm, sc_t, 0.1), ref.update(Q, qm, sc_t, 0.1)) and torch.equal(st.x, ref.x)
    assert not st._fold_ws


def test_c_abi_refuses_what_the_folded_form_cannot_hold():
    import evdr_amd  # noqa: F401
    from evdr_amd import _lib as L
    lib = L.load()
    rc = lib.evdr_maxsim_fwd_prepared_nce(1, 1, None, 1, 1, 1, 2000, 1, 8, 32, 2000, 64, 64 * 128, 0, None, None, 1, 0.1, 1, 1, 1, None)
    assert rc == L.EVDR_ERR_SHAPE and b"np <= 1024" in lib.evdr_last_error()
    rc = lib.evdr_maxsim_bwd_adamw_planes_nce(1, 1, 0.1, 1, None, None, 1, 1, 1, 1, 300, 32, 8, 64, 128, 1e-3, 0.9, 0.999, 1e-8, 1e-2, 1, 1e-12,
                                              None, None, None, None, None)
    assert rc == L.EVDR_ERR_SHAPE and b"nq <= 256" in lib.evdr_last_error()
    rc = lib.evdr_maxsim_fwd_prepared_nce(1, 1, None, 1, 1, 1, 64, 1, 8, 32, 64, 64, 64 * 128, 0, None, None, 1, 0.0, 1, 1, 1, None)
    assert rc == L.EVDR_ERR_ARG and b"temperature" in lib.evdr_last_error()
