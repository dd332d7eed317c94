"""The reference's secondary training call patterns, re-enacted against a BACKEND -- the functions an
unmodified reference script would import: `score_multi_vector_masked`, `l2_normalize`, `listwise_distillation_loss`,
`score_preserving_loss`.  The same code runs the oracle on the CPU (-m "not gpu") and the drop-in modules on the GPU
(-m gpu); fixtures come from the reference's own train_one_step functions (tests/golden/make_golden_v3.py).

Call sequences follow  mainv2_iter_liscore.py:282-311,  mainv3_iter_liscore_noisev1.py:284-316,
mainv3_iter_liscore_mixup.py:291-343,  mainv3_iter_liscore_QA_hardtoken.py:340-445."""
from types import SimpleNamespace

import torch


def oracle_backend():
    import evdr_amd  # noqa: F401
    from evdr_amd import criterion as C          # the (B, N) losses are plain torch and run on either device
    from oracle import maxsim_oracle as O
    return SimpleNamespace(score=lambda Q, P, qm, pm, chunk=64: O.maxsim_masked(Q, P, qm, pm, chunk_p=chunk), l2=O.l2_normalize,
                           listwise=C.listwise_distillation_loss, preserve=C.score_preserving_loss,
                           infonce=lambda s, t, temperature: O.infonce_distill(s, t, temperature), device="cpu")


def hip_backend():
    import evdr_amd  # noqa: F401
    from evdr_amd import criterion as C
    from evdr_amd.evaluator.retrieval import score_multi_vector_masked
    from evdr_amd.utils.preprocess_data import l2_normalize
    return SimpleNamespace(score=score_multi_vector_masked, l2=l2_normalize, listwise=C.listwise_distillation_loss,
                           preserve=C.score_preserving_loss, infonce=C.infonce_distillation_loss, device="cuda:0")


def setup(be, case):
    Qb, qmb, Pt, pmt, Pbar0, pms, hp = case
    d = be.device
    Ptn = be.l2((Pt * pmt.unsqueeze(-1)).to(d)).detach()
    param = torch.nn.Parameter((Pbar0 * pms.unsqueeze(-1)).to(d))
    opt = torch.optim.AdamW([param], lr=hp["lr"], weight_decay=hp["wd"])
    return Qb.to(d), qmb.to(d), Ptn, pmt.to(d), param, pms.to(d), opt, hp


def _main_losses(be, Qb, qmb, Ptn, pmt, Psb, pms, hp):
    with torch.no_grad():
        sc_t = be.score(Qb, Ptn, qmb, pmt, 64)
    sc_s = be.score(Qb, Psb, qmb, pms, 64)
    l_list = be.listwise(sc_s, sc_t, k=hp["k"], temperature=hp["temp"])
    l_score = be.preserve(sc_s, sc_t)
    return sc_t, sc_s, l_list, l_score


def _finish(total, opt, param, **out):
    opt.zero_grad(set_to_none=True)
    total.backward()
    grad = param.grad.detach().clone()
    opt.step()
    return dict(total_loss=float(total.item()), grad=grad, param_after=param.detach(), **out)


def step_liscore(be, case, Qb_override=None):
    Qb, qmb, Ptn, pmt, param, pms, opt, hp = setup(be, case)
    if Qb_override is not None:
        Qb = Qb_override
    Psb = be.l2(param * pms.unsqueeze(-1))
    sc_t, sc_s, l_list, l_score = _main_losses(be, Qb, qmb, Ptn, pmt, Psb, pms, hp)
    total = hp["lambda_list"] * l_list + hp["lambda_score"] * l_score
    return _finish(total, opt, param, loss_list=float(l_list.item()), loss_score=float(l_score.item()), sc_t=sc_t, sc_s=sc_s.detach())


def step_noise(be, case):
    Qb, qmb, *_rest, hp = setup(be, case)
    torch.manual_seed(hp["noise_seed"])
    noise = torch.randn(Qb.shape).to(be.device) * hp["q_noise_std"]          # the CPU generator's draw, as in the fixture run
    Qn = Qb + noise * qmb.unsqueeze(-1)
    Qn = be.l2(Qn * qmb.unsqueeze(-1))
    out = step_liscore(be, case, Qb_override=Qn)
    out["Qb_used"] = Qn
    return out


def step_mixup(be, case, lam, perm):
    Qb, qmb, Ptn, pmt, param, pms, opt, hp = setup(be, case)
    perm = perm.to(be.device)
    P_masked = param * pms.unsqueeze(-1)
    Psb = be.l2(P_masked)
    sc_t, sc_s, l_list, l_score = _main_losses(be, Qb, qmb, Ptn, pmt, Psb, pms, hp)
    total = hp["lambda_list"] * l_list + hp["lambda_score"] * l_score
    pm_mix = pms & pms[perm]
    P_mix = lam * P_masked + (1.0 - lam) * P_masked[perm]
    Psb_mix = be.l2(P_mix * pm_mix.unsqueeze(-1))
    sc_s_mix = be.score(Qb, Psb_mix, qmb, pm_mix, 64)          # second student forward (and backward) of the step
    with torch.no_grad():
        sc_t_mix = lam * sc_t + (1.0 - lam) * sc_t[:, perm]
    l_mix = be.preserve(sc_s_mix, sc_t_mix.detach())
    total = total + hp["lambda_mixed"] * (hp["lambda_score"] * l_mix)
    return _finish(total, opt, param, loss_list=float(l_list.item()), loss_score=float(l_score.item()),
                   loss_score_mix=float(l_mix.item()), sc_s_mix=sc_s_mix.detach())


def select_hard_tokens(sc_t, sc_s, Qb, qmb, Ptn, pmt, k, aux_docs):
    """Host-side selection of the hard-token step on CPU tensors (rank gaps -> hard pages -> their most query-like patch);
    runs on the CPU like the fixture run did, so integer-gap ties resolve the same way."""
    rank_t = torch.argsort(torch.argsort(sc_t, dim=-1, descending=True), dim=-1)
    rank_s = torch.argsort(torch.argsort(sc_s, dim=-1, descending=True), dim=-1)
    gap = rank_t.float() - rank_s.float()
    kk = min(int(k), sc_t.shape[1])
    top_i = torch.topk(sc_t, k=kk, dim=-1, largest=True).indices
    a = min(int(aux_docs), kk)
    aux = top_i.gather(1, torch.topk(gap.gather(1, top_i).abs(), k=a, dim=1, largest=True).indices)
    qv = []
    for qi in range(Qb.shape[0]):
        toks = Qb[qi][qmb[qi].bool()]
        for di in aux[qi].tolist():
            sim = toks @ Ptn[di].T
            sim[:, ~pmt[di].bool()] = float("-inf")
            qv.append(Ptn[di][torch.argmax(sim.max(dim=0).values)])
    qv = torch.stack(qv)
    return (qv / (qv.norm(dim=-1, keepdim=True) + 1e-12)).view(-1, 1, qv.shape[-1])


def step_hardtoken(be, case):
    Qb, qmb, Ptn, pmt, param, pms, opt, hp = setup(be, case)
    Psb = be.l2(param * pms.unsqueeze(-1))
    sc_t, sc_s, l_list, l_score = _main_losses(be, Qb, qmb, Ptn, pmt, Psb, pms, hp)
    main = hp["lambda_list"] * l_list + hp["lambda_score"] * l_score
    qv = select_hard_tokens(sc_t.cpu(), sc_s.detach().cpu(), Qb.cpu(), qmb.cpu(), Ptn.cpu(), pmt.cpu(), hp["k"], hp["aux_docs"]).to(be.device)
    qmv = torch.ones(qv.shape[0], 1, dtype=torch.bool, device=be.device)
    with torch.no_grad():
        sc_t_v = be.score(qv, Ptn, qmv, pmt, 64)               # Lq = 1 through the teacher scorer ...
    sc_s_v = be.score(qv, Psb, qmv, pms, 64)                   # ... and, with grad, through the student scorer
    l_list_v = be.listwise(sc_s_v, sc_t_v, k=hp["k"], temperature=hp["temp"])
    l_score_v = be.preserve(sc_s_v, sc_t_v)
    aux = hp["lambda_list"] * l_list_v + hp["lambda_score"] * l_score_v
    total = main + hp["lambda_aux"] * aux
    return _finish(total, opt, param, loss_main=float(main.item()), loss_aux=float(aux.item()), loss_list_aux=float(l_list_v.item()),
                   loss_score_aux=float(l_score_v.item()), q_virtual=qv, sc_t_v=sc_t_v, sc_s_v=sc_s_v.detach())


# ---- the nine remaining single-step scripts: same step, different loss (tests/golden/make_golden_losses_steps.py) ------------
def loss_steps(be):
    """script tag -> (loss(sc_s, sc_t) -> (total, {name: value}), needs_teacher): the loss lines of
    mainv2_iter_{lambda,linfo_distill,lipairwise,listwise,pairscore,ranknce,ranknet,score_preserve,super_infonce}.py"""
    import evdr_amd  # noqa: F401
    from evdr_amd import criterion as C

    def linfo(s, t):
        a, b = C.listwise_distillation_loss(s, t, k=8, temperature=2.0), be.infonce(s, t, temperature=0.1)   # the HIP loss kernel on the GPU
        return 1.0 * a + 0.5 * b, {"loss_list": a, "loss_info": b}

    def lipair(s, t):
        a, b = C.listwise_distillation_loss(s, t, k=8, temperature=2.0), C.pairwise_distillation_loss(s, t)
        return 1.0 * a + 0.5 * b, {"loss_list": a, "loss_pair": b}

    def pairscore(s, t):
        a, b = C.pairwise_distillation_loss(s, t), C.score_preserving_loss(s, t)
        return 1.0 * a + 0.5 * b, {"loss_pair": a, "loss_score": b}

    pos = torch.tensor([3, 0, 15, 7, 7, 9])
    single = lambda f: (lambda s, t: (f(s, t), {}))
    return {
        "lambda": single(lambda s, t: C.lambda_loss(s, t, alpha=1.0, eps=1e-6)),
        "linfo_distill": linfo,
        "lipairwise": lipair,
        "listwise": single(lambda s, t: C.listwise_distillation_loss(s, t, k=8, temperature=2.0)),
        "pairscore": pairscore,
        "ranknce": single(lambda s, t: C.ranknce_loss(s, t, temperature=0.5, lambda_weight=0.7)),
        "ranknet": single(lambda s, t: C.pairwise_distillation_loss(s, t)),
        "score_preserve": single(lambda s, t: C.score_preserving_loss(s, t)),
        "super_infonce": single(lambda s, t: C.infonce_supervised_loss(s, pos.to(s.device), temperature=0.07)),
    }


def step_with_loss(be, case, loss_fn):
    Qb, qmb, Ptn, pmt, param, pms, opt, hp = setup(be, case)
    Psb = be.l2(param * pms.unsqueeze(-1))
    with torch.no_grad():
        sc_t = be.score(Qb, Ptn, qmb, pmt, 64)
    sc_s = be.score(Qb, Psb, qmb, pms, 64)
    total, parts = loss_fn(sc_s, sc_t)
    return _finish(total, opt, param, **{k: float(v.item()) for k, v in parts.items()})
