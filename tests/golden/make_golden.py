#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by RUNNING THE REFERENCE'S OWN FUNCTIONS.

Runs only in the build container (needs /root/reference); the fixtures (numbers only) are
committed, this script is committed, the reference's sources are not.

How the reference is made importable:
  * `mteb` is absent, and evaluator/retrieval.py:218 imports it at module level -> empty stub
    modules are registered in sys.modules (the metric wrapper is never called here).
  * evaluator/retrieval.py:39,41 hard-code device='cuda' inside left_padding; there is no GPU in
    the container, so for the ONE fixture that exercises score_multi_vector the script redirects
    'cuda' -> 'cpu' in Tensor.to / torch.full while that call runs.  The arithmetic executed is
    the reference's own.

Input recipes are seeded (torch CPU generator) and re-created by tests from the same recipe in
tests/golden_recipes.py; large inputs are therefore not stored, only their outputs.
"""
import contextlib
import io
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))          # tests/
import golden_recipes as R  # noqa: E402

REF = "/root/reference"


def import_reference():
    for name in ["mteb", "mteb.evaluation", "mteb.evaluation.evaluators",
                 "mteb.evaluation.evaluators.RetrievalEvaluator"]:
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["mteb.evaluation.evaluators.RetrievalEvaluator"].RetrievalEvaluator = object
    tb = types.ModuleType("torch.utils.tensorboard")        # utils/utils.py:9 imports it at module level; absent here
    tb.SummaryWriter = object
    sys.modules.setdefault("torch.utils.tensorboard", tb)
    sys.path.insert(0, REF)
    import criterion as ref_criterion
    import evaluator.retrieval as ref_retrieval
    import utils.preprocess_data as ref_prep
    return ref_retrieval, ref_criterion, ref_prep


@contextlib.contextmanager
def cuda_means_cpu():
    orig_to, orig_full = torch.Tensor.to, torch.full

    def to(self, *a, **kw):
        a = tuple("cpu" if (isinstance(x, str) and x.startswith("cuda")) else x for x in a)
        if isinstance(kw.get("device"), str) and kw["device"].startswith("cuda"):
            kw["device"] = "cpu"
        return orig_to(self, *a, **kw)

    def full(*a, **kw):
        if isinstance(kw.get("device"), str) and kw["device"].startswith("cuda"):
            kw["device"] = "cpu"
        return orig_full(*a, **kw)

    torch.Tensor.to, torch.full = to, full
    try:
        yield
    finally:
        torch.Tensor.to, torch.full = orig_to, orig_full


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **{k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v))
                                 for k, v in arrays.items()})
    print(f"[golden] {name}.npz  {os.path.getsize(path) / 1024:.1f} KiB")


def main():
    torch.set_num_threads(8)
    ref_retrieval, ref_criterion, ref_prep = import_reference()
    score = ref_retrieval.score_multi_vector_masked

    # ---- A1 forward + autograd, small, inputs stored ------------------------------------
    for case in ("small_ragged", "lq1", "chunk_tail"):
        Q, P, qm, pm, g = R.small_case(case)
        Pg = P.clone().requires_grad_(True)
        s = score(Q, Pg, qm, pm, chunk_p=R.SMALL_CHUNK[case])
        (s * g).sum().backward()
        sim = torch.einsum("qnd,cmd->qcnm", Q, P).masked_fill(~pm[None, :, None, :], -1e4)
        save(f"a1_{case}", Q=Q, P=P, qmask=qm, pmask=pm, g=g, scores=s, dP=Pg.grad,
             argmax=sim.max(dim=-1).indices.to(torch.int32))

    # ---- A1 forward, seeded 1030-patch pages, only outputs stored -------------------------
    for bf16 in (False, True):
        Q, P, qm, pm = R.seeded_1030(bf16_inputs=bf16)
        s = score(Q, P, qm, pm, chunk_p=64)
        save("a1_seeded1030_" + ("bf16" if bf16 else "f32"), scores=s)

    # ---- A4 l2_normalize ------------------------------------------------------------------
    x = R.l2_case()
    save("a4_l2norm", x=x, y=ref_prep.l2_normalize(x))

    # ---- A5 loss + closed-form gradient ------------------------------------------------------
    ss, st = R.infonce_case()
    ssg = ss.clone().requires_grad_(True)
    loss = ref_criterion.infonce_distillation_loss(ssg, st, temperature=0.1)
    loss.backward()
    save("a5_infonce", score_s=ss, score_t=st, loss=loss.detach(), dscore=ssg.grad, temp=np.float32(0.1))

    # ---- A7 one train step (mainv2_iter_distill_infonce.py:269-292 call pattern) ---------------
    for tag in ("b4n8", "b32n128"):
        Qb, qmb, Pt, pmt, Pbar0, pms, hp = R.train_case(tag)
        Pt_norm = ref_prep.l2_normalize(Pt * pmt.unsqueeze(-1)).detach()
        param = torch.nn.Parameter(Pbar0 * pms.unsqueeze(-1))
        opt = torch.optim.AdamW([param], lr=hp["lr"], weight_decay=hp["wd"])
        Psb = ref_prep.l2_normalize(param * pms.unsqueeze(-1))
        with torch.no_grad():
            sc_t = score(Qb, Pt_norm, qmb, pmt, 64)
        sc_s = score(Qb, Psb, qmb, pms, 64)
        loss = ref_criterion.infonce_distillation_loss(sc_s, sc_t, temperature=hp["temp"])
        opt.zero_grad(set_to_none=True)
        loss.backward()
        grad = param.grad.detach().clone()
        opt.step()
        if tag == "b4n8":
            save("a7_step_" + tag, loss=loss.detach(), sc_t=sc_t, sc_s=sc_s.detach(), grad=grad,
                 param_after=param.detach())
        else:  # big tensors: keep scores, loss and a strided sample + norms of grad/param
            save("a7_step_" + tag, loss=loss.detach(), sc_t=sc_t, sc_s=sc_s.detach(),
                 grad_sample=grad[::8, ::8, ::4].contiguous(), grad_norm=grad.double().norm(),
                 grad_abs_sum=grad.abs().sum(),
                 param_sample=param.detach()[::8, ::8, ::4].contiguous(),
                 param_norm=param.detach().double().norm())   # float64: an fp32 norm over 3.4M values is only good to ~1e-4

    # ---- the six secondary losses of criterion.py (value + gradient w.r.t. the student scores) -------------
    ss, st, labels = R.losses_case()
    cases = {
        "infonce_supervised_loss": lambda s: ref_criterion.infonce_supervised_loss(s, labels, temperature=0.07),
        "score_preserving_loss": lambda s: ref_criterion.score_preserving_loss(s, st),
        "pairwise_distillation_loss": lambda s: ref_criterion.pairwise_distillation_loss(s, st),
        "listwise_distillation_loss": lambda s: ref_criterion.listwise_distillation_loss(s, st, k=10, temperature=2.0),
        "lambda_loss": lambda s: ref_criterion.lambda_loss(s, st),
        "ranknce_loss": lambda s: ref_criterion.ranknce_loss(s, st, temperature=0.5, lambda_weight=0.7),
    }
    arrays = {}
    for name, fn in cases.items():
        sg = ss.clone().requires_grad_(True)
        val = fn(sg)
        val.backward()
        arrays[name] = val.detach()
        arrays[name + "_grad"] = sg.grad
    save("losses", **arrays)

    # ---- npz schema helpers (utils/preprocess_data.py, utils/utils.py) -------------------------------------
    import utils.utils as ref_utils
    docs, attn, img, queries, qattn, docid = R.npz_payload_case()
    P_raw, pmask, valid = ref_prep.preprocess_docs(docs, attn, img, device="cpu")
    P_raw2, pmask2, _ = ref_prep.preprocess_docs(docs, None, None, device="cpu")
    Qn, qmask = ref_prep.preprocess_queries(queries, qattn, device="cpu")
    objs = ref_utils.tokens_to_object(P_raw.numpy(), pmask.numpy())
    perm = np.array([3, 0, 6, 1, 5, 2, 4])
    (docs_al,), ok = ref_utils.align_by_docid(docid, docid[perm], docs[perm])
    save("npz_helpers", P_raw=P_raw, pmask=pmask, valid=valid, pmask_nomask=pmask2, Q=Qn, qmask=qmask,
         obj_lens=np.array([o.shape[0] for o in objs]), obj_concat=np.concatenate(list(objs), axis=0),
         align_ok=np.array(ok), align_first_rows=np.stack([d[0] for d in docs_al]))

    # ---- A3 single vector ---------------------------------------------------------------------
    qs, ps = R.single_vector_case()
    save("a3_single", scores=ref_retrieval.BaseVisualRetrieverProcessor.score_single_vector(qs, ps, device="cpu"))

    # ---- A2 unmasked, zero-left-padded ColPali scorer ---------------------------------------------
    qs, ps = R.ragged_lists_case()
    with cuda_means_cpu(), contextlib.redirect_stdout(io.StringIO()):
        s2 = ref_retrieval.BaseVisualRetrieverProcessor.score_multi_vector(qs, ps, batch_size=4, device="cpu")
        s2_one = ref_retrieval.BaseVisualRetrieverProcessor.score_multi_vector(qs, ps, batch_size=128, device="cpu")
    save("a2_unmasked_lists", scores_bs4=s2, scores_bs128=s2_one)


if __name__ == "__main__":
    main()
