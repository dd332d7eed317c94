"""CPU oracle for the late-interaction (MaxSim) hot path.  TEST INFRASTRUCTURE ONLY.

This file is a from-scratch CPU restatement (torch fp32 on the host) of the
reference's scoring path.  It is *not* product code: only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import it, and
only as the checker / the timed CPU baseline.  The product path
(`efficient-visual-document-retrieval_amd/`) never imports anything from `oracle/`.

Pinning: every function here is checked in `tests/test_oracle_golden.py` against
fixtures under `tests/golden/` that were produced by importing the reference's own
functions in the build container (`tests/golden/make_golden.py`, committed).
The retrieval *metric* (`trec_metrics`) is the exception: the reference delegates
it to `mteb`/`pytrec_eval`, which are absent and unpinned -> "parity unpinned"
for that one function (closed-form single-relevant checks only).

Each function cites the reference file:line it follows (paths relative to
/root/reference).
"""
from __future__ import annotations

import math
from typing import Dict, List, Sequence, Tuple

import torch

NEG_FILL = -1e4  # evaluator/retrieval.py:185


def l2_normalize(x: torch.Tensor, eps: float = 1e-12) -> torch.Tensor:
    """utils/preprocess_data.py:8-9 -- eps is ADDED to the norm (not clamped).
    torch's vector norm is used on purpose: its backward takes the subgradient 0 at an
    all-zero row (masked patches), where sqrt(sum(x*x)) would produce NaN."""
    n = torch.linalg.vector_norm(x, ord=2, dim=-1, keepdim=True)
    return x / (n + eps)


def maxsim_masked(
    Q: torch.Tensor, P: torch.Tensor, qmask: torch.Tensor, pmask: torch.Tensor, chunk_p: int = 128
) -> torch.Tensor:
    """evaluator/retrieval.py:166-213 (score_multi_vector_masked).

    out[q,p] = sum_n qmask[q,n] * has(p) * max_m( Q[q,n]·P[p,m] if pmask[p,m] else -1e4 )
    fp32 throughout; differentiable w.r.t. P and Q through torch autograd.
    """
    Qf = Q.float()
    Pf = P.float()
    qm = qmask.bool()
    pm = pmask.bool()
    nq, lq, d = Qf.shape
    npg = Pf.shape[0]
    qw = qm.float()
    cols = []
    for s in range(0, npg, chunk_p):
        Pc = Pf[s : s + chunk_p]
        mc = pm[s : s + chunk_p]
        c, lp, _ = Pc.shape
        # (nq*lq, d) @ (d, c*lp) -> (nq, lq, c, lp) -> (nq, c, lq, lp)
        sim = (Qf.reshape(nq * lq, d) @ Pc.reshape(c * lp, d).t()).reshape(nq, lq, c, lp).permute(0, 2, 1, 3)
        sim = torch.where(mc[None, :, None, :], sim, torch.full_like(sim, NEG_FILL))
        best = sim.amax(dim=-1)                                   # (nq, c, lq)
        alive = mc.any(dim=1).to(best.dtype)                      # (c,)
        best = best * alive[None, :, None] * qw[:, None, :]
        cols.append(best.sum(dim=-1))
    return torch.cat(cols, dim=1)


def maxsim_masked_argmax(
    Q: torch.Tensor, P: torch.Tensor, qmask: torch.Tensor, pmask: torch.Tensor
) -> Tuple[torch.Tensor, torch.Tensor]:
    """Scores plus the first-maximal patch index per (q, p, n), as torch's max picks it
    (evaluator/retrieval.py:201; SURVEY §4: ties go to the FIRST maximal index)."""
    Qf, Pf = Q.float(), P.float()
    sim = torch.einsum("qnd,pmd->qpnm", Qf, Pf)
    sim = torch.where(pmask.bool()[None, :, None, :], sim, torch.full_like(sim, NEG_FILL))
    best, arg = sim.max(dim=-1)
    alive = pmask.bool().any(dim=1).float()
    sc = (best * alive[None, :, None] * qmask.float()[:, None, :]).sum(-1)
    return sc, arg


def maxsim_backward(
    g: torch.Tensor, Q: torch.Tensor, P: torch.Tensor, qmask: torch.Tensor, pmask: torch.Tensor
) -> torch.Tensor:
    """Analytic dP of maxsim_masked for upstream gradient g (nq, np): scatter of
    g[q,p]*qmask[q,n]*has(p)*Q[q,n,:] into row argmax."""
    _, arg = maxsim_masked_argmax(Q, P, qmask, pmask)
    nq, npg, lq = arg.shape
    alive = pmask.bool().any(dim=1).float()
    w = g[:, :, None] * qmask.float()[:, None, :] * alive[None, :, None]   # (nq,np,lq)
    dP = torch.zeros_like(P, dtype=torch.float32)
    Qf = Q.float()
    for p in range(npg):
        contrib = w[:, p, :, None] * Qf                                     # (nq,lq,d)
        dP[p].index_add_(0, arg[:, p, :].reshape(-1), contrib.reshape(nq * lq, -1))
    return dP


def left_pad_stack(seqs: Sequence[torch.Tensor]) -> torch.Tensor:
    """evaluator/retrieval.py:30-45 -- zero LEFT padding to the batch max length."""
    seqs = [s[None, :] if s.ndim == 1 else s for s in seqs]
    lmax = max(s.shape[0] for s in seqs)
    d = seqs[0].shape[-1]
    out = []
    for s in seqs:
        pad = torch.zeros(lmax - s.shape[0], d, dtype=s.dtype)
        out.append(torch.cat([pad, s], dim=0))
    return torch.stack(out)


def maxsim_unmasked_lists(qs: Sequence[torch.Tensor], ps: Sequence[torch.Tensor], batch_size: int = 128) -> torch.Tensor:
    """evaluator/retrieval.py:101-150 (score_multi_vector): per (query-batch, page-batch)
    block, zero-left-pad each side to its own block max, fp32 upcast for half types or
    mixed dtypes, einsum -> max over page tokens (padding rows take part) -> sum over
    query tokens.  Result fp32 on CPU."""
    if len(qs) == 0:
        raise ValueError("No queries provided")
    if len(ps) == 0:
        raise ValueError("No passages provided")
    rows = []
    for i in range(0, len(qs), batch_size):
        qb = left_pad_stack(qs[i : i + batch_size])
        blocks = []
        for j in range(0, len(ps), batch_size):
            pb = left_pad_stack(ps[j : j + batch_size])
            a, b = qb, pb
            if a.dtype != b.dtype or a.dtype in (torch.float16, torch.bfloat16):
                a, b = a.float(), b.float()
            sim = torch.einsum("bnd,csd->bcns", a, b)
            blocks.append(sim.amax(dim=3).sum(dim=2))
        rows.append(torch.cat(blocks, dim=1))
    return torch.cat(rows, dim=0).float()


def dot_single_vector(qs: Sequence[torch.Tensor], ps: Sequence[torch.Tensor]) -> torch.Tensor:
    """evaluator/retrieval.py:78-99 (score_single_vector)."""
    if len(qs) == 0:
        raise ValueError("No queries provided")
    if len(ps) == 0:
        raise ValueError("No passages provided")
    return (torch.stack(list(qs)) @ torch.stack(list(ps)).t()).float()


def infonce_distill(score_s: torch.Tensor, score_t: torch.Tensor, temperature: float = 0.07) -> torch.Tensor:
    """criterion.py:56-68 -- CE(student/τ, argmax teacher), mean over the batch."""
    tgt = score_t.detach().argmax(dim=1)
    z = score_s / temperature
    lse = torch.logsumexp(z, dim=1)
    return (lse - z.gather(1, tgt[:, None]).squeeze(1)).mean()


def infonce_distill_grad(score_s: torch.Tensor, score_t: torch.Tensor, temperature: float) -> torch.Tensor:
    """d loss / d score_s in closed form: (softmax(s/τ) - onehot(argmax t)) / (τ B)."""
    b = score_s.shape[0]
    p = torch.softmax(score_s / temperature, dim=1)
    p[torch.arange(b), score_t.argmax(dim=1)] -= 1.0
    return p / (temperature * b)


def distill_train_step(
    Qb, qmb, P_teacher_norm, pmask_t, Pbar, pmask_s, temp: float, lr: float, wd: float, chunk_p: int = 64
):
    """mainv2_iter_distill_infonce.py:269-292 (train_one_step) with AdamW from
    utils/utils.py:78-80 (torch defaults).  Returns (loss, grad, updated Pbar)."""
    param = torch.nn.Parameter(Pbar.clone())
    opt = torch.optim.AdamW([param], lr=lr, weight_decay=wd)
    Ps = l2_normalize(param * pmask_s.unsqueeze(-1))
    with torch.no_grad():
        sc_t = maxsim_masked(Qb, P_teacher_norm, qmb, pmask_t, chunk_p)
    sc_s = maxsim_masked(Qb, Ps, qmb, pmask_s, chunk_p)
    loss = infonce_distill(sc_s, sc_t, temp)
    opt.zero_grad(set_to_none=True)
    loss.backward()
    grad = param.grad.detach().clone()
    opt.step()
    return float(loss.item()), grad, param.detach().clone(), sc_t, sc_s.detach()


# --------------------------------------------------------------------------------------
# retrieval metrics: trec_eval semantics restated from first principles.
# PARITY UNPINNED: the reference calls mteb -> pytrec_eval (evaluator/retrieval.py:239-246),
# neither is installed nor pinned anywhere in the reference.
# --------------------------------------------------------------------------------------
def _trec_rank(doc_scores: Dict[str, float]) -> List[str]:
    # trec_eval orders by score descending, ties by docno descending.
    return [d for d, _ in sorted(doc_scores.items(), key=lambda kv: (kv[1], kv[0]), reverse=True)]


def trec_metrics(qrels: Dict[str, Dict[str, int]], results: Dict[str, Dict[str, float]], k_values: Sequence[int]):
    """Slow, obviously-correct nDCG/MAP/Recall/P/MRR @k (means over judged∩scored queries)."""
    ndcg = {f"NDCG@{k}": 0.0 for k in k_values}
    amap = {f"MAP@{k}": 0.0 for k in k_values}
    rec = {f"Recall@{k}": 0.0 for k in k_values}
    prec = {f"P@{k}": 0.0 for k in k_values}
    mrr = {f"MRR@{k}": 0.0 for k in k_values}
    qids = [q for q in results if q in qrels]
    for q in qids:
        rels = {d: r for d, r in qrels[q].items()}
        order = _trec_rank(results[q])
        gains = [rels.get(d, 0) for d in order]
        ideal = sorted([r for r in rels.values() if r > 0], reverse=True)
        nrel = len(ideal)
        for k in k_values:
            top = gains[:k]
            dcg = sum(g / math.log2(i + 2) for i, g in enumerate(top) if g > 0)
            idcg = sum(g / math.log2(i + 2) for i, g in enumerate(ideal[:k]))
            ndcg[f"NDCG@{k}"] += dcg / idcg if idcg > 0 else 0.0
            hits = [1 if g > 0 else 0 for g in top]
            rec[f"Recall@{k}"] += (sum(hits) / nrel) if nrel else 0.0
            prec[f"P@{k}"] += sum(hits) / k
            ap, seen = 0.0, 0
            for i, h in enumerate(hits):
                if h:
                    seen += 1
                    ap += seen / (i + 1)
            amap[f"MAP@{k}"] += (ap / nrel) if nrel else 0.0
        # MRR follows mteb's evaluate_custom("mrr"): plain score sort (stable), first relevant hit.
        plain = [d for d, _ in sorted(results[q].items(), key=lambda kv: kv[1], reverse=True)]
        for k in k_values:
            rr = 0.0
            for i, d in enumerate(plain[:k]):
                if rels.get(d, 0) > 0:
                    rr = 1.0 / (i + 1)
                    break
            mrr[f"MRR@{k}"] += rr
    n = max(len(qids), 1)
    rnd = lambda dct, den: {k: round(v / den, 5) for k, v in dct.items()}
    return {"NDCG": rnd(ndcg, n), "mAP": rnd(amap, n), "Recall": rnd(rec, n), "Precision": rnd(prec, n),
            "mRR": rnd(mrr, max(len(qrels), 1))}


def topk_rows(scores: torch.Tensor, k: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """Deterministic top-k per row: score descending, index ascending on ties."""
    n = scores.shape[1]
    k = min(k, n)
    idx = torch.arange(n).expand_as(scores)
    # stable sort on -score keeps ascending index order among equals
    order = torch.sort(-scores, dim=1, stable=True).indices[:, :k]
    return scores.gather(1, order), idx.gather(1, order).to(torch.int32)
