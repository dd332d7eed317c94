#!/usr/bin/env python3
"""Golden fixtures for SURVEY §8(f) row 4 -- the other training call patterns on the same boundary -- produced by RUNNING
THE REFERENCE'S OWN `train_one_step` of four of its scripts on the CPU (build container only; needs /root/reference):

  liscore    mainv2_iter_liscore.py:282-311            listwise + score-preserving loss on (B, N) MaxSim scores
  noise      mainv3_iter_liscore_noisev1.py:284-316    query noise + re-normalisation in front of the same step
  mixup      mainv3_iter_liscore_mixup.py:291-343      a SECOND student forward/backward on mixed pages in the same step
  hardtoken  mainv3_iter_liscore_QA_hardtoken.py:340-445   single-token "virtual queries" (Lq = 1) through both scorers

The scripts are imported as modules (their `train_one_step` is a module-level function); `score_multi_vector_masked` inside
them is wrapped by a recorder so that the arguments/results of every scorer call of the step are captured (the reference
only returns the losses).  Only numbers are stored; inputs come from tests/golden_recipes.py:v3_case()."""
import importlib
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
import golden_recipes as R  # noqa: E402
from make_golden import import_reference, save  # noqa: E402


class Recorder:
    def __init__(self, fn):
        self.fn, self.calls = fn, []

    def __call__(self, Q, P, qmask, pmask, chunk_p=128):
        out = self.fn(Q, P, qmask, pmask, chunk_p)
        self.calls.append({"Q": Q.detach().clone(), "qmask": qmask.detach().clone(), "pmask": pmask.detach().clone(),
                           "out": out.detach().clone()})
        return out


def fresh(ref_prep):
    Qb, qmb, Pt, pmt, Pbar0, pms, hp = R.v3_case()
    Ptn = ref_prep.l2_normalize(Pt * pmt.unsqueeze(-1)).detach()
    param = torch.nn.Parameter(Pbar0 * pms.unsqueeze(-1))
    opt = torch.optim.AdamW([param], lr=hp["lr"], weight_decay=hp["wd"])
    return Qb, qmb, Ptn, pmt, param, pms, opt, hp


def check_margins(sc, what, tol=1e-4):
    """Rank-based selections of the hard-token step must not hinge on fp32 noise: adjacent sorted scores differ by > tol."""
    s = torch.sort(sc, dim=1).values
    gap = (s[:, 1:] - s[:, :-1]).min().item()
    assert gap > tol, f"{what}: adjacent scores only {gap:.2e} apart -- pick another seed"
    return gap


def main():
    torch.set_num_threads(8)
    _, _, ref_prep = import_reference()
    common = lambda hp: dict(k=hp["k"], temp=hp["temp"], chunk_p=64, lambda_list=hp["lambda_list"], lambda_score=hp["lambda_score"])

    # ---- liscore ----------------------------------------------------------------------------------------------------
    mod = importlib.import_module("mainv2_iter_liscore")
    rec = mod.score_multi_vector_masked = Recorder(mod.score_multi_vector_masked)
    Qb, qmb, Ptn, pmt, param, pms, opt, hp = fresh(ref_prep)
    out = mod.train_one_step(Qb, qmb, Ptn, pmt, param, pms, opt, **common(hp))
    save("v3_liscore", total_loss=np.float64(out["total_loss"]), loss_list=np.float64(out["loss_list"]),
         loss_score=np.float64(out["loss_score"]), sc_t=rec.calls[0]["out"], sc_s=rec.calls[1]["out"], grad=param.grad,
         param_after=param.detach())

    # ---- noise: torch.randn_like(Qb) is the first draw after the seed ---------------------------------------------
    mod = importlib.import_module("mainv3_iter_liscore_noisev1")
    rec = mod.score_multi_vector_masked = Recorder(mod.score_multi_vector_masked)
    Qb, qmb, Ptn, pmt, param, pms, opt, hp = fresh(ref_prep)
    torch.manual_seed(hp["noise_seed"])
    out = mod.train_one_step(Qb, qmb, Ptn, pmt, param, pms, opt, q_noise_std=hp["q_noise_std"], **common(hp))
    save("v3_noise", total_loss=np.float64(out["total_loss"]), loss_list=np.float64(out["loss_list"]),
         loss_score=np.float64(out["loss_score"]), Qb_used=rec.calls[0]["Q"], sc_t=rec.calls[0]["out"], sc_s=rec.calls[1]["out"],
         grad=param.grad, param_after=param.detach())

    # ---- mixup: lam = np.random.beta(a, a), perm = torch.randperm(N) are the first draws of their generators -------
    mod = importlib.import_module("mainv3_iter_liscore_mixup")
    rec = mod.score_multi_vector_masked = Recorder(mod.score_multi_vector_masked)
    Qb, qmb, Ptn, pmt, param, pms, opt, hp = fresh(ref_prep)
    np.random.seed(hp["mixup_seed"])
    lam = float(np.random.beta(hp["mixup_alpha"], hp["mixup_alpha"]))
    torch.manual_seed(hp["mixup_seed"])
    perm = torch.randperm(param.shape[0])
    np.random.seed(hp["mixup_seed"])
    torch.manual_seed(hp["mixup_seed"])
    out = mod.train_one_step(Qb, qmb, Ptn, pmt, param, pms, opt, lambda_mixed=hp["lambda_mixed"], mixup_alpha=hp["mixup_alpha"],
                             **common(hp))
    assert torch.equal(rec.calls[2]["pmask"], pms & pms[perm]), "captured perm is not the one the step drew"
    save("v3_mixup", lam=np.float64(lam), perm=perm, total_loss=np.float64(out["total_loss"]), loss_list=np.float64(out["loss_list"]),
         loss_score=np.float64(out["loss_score"]), loss_score_mix=np.float64(out["loss_score_mix"]),
         sc_s_mix=rec.calls[2]["out"], grad=param.grad, param_after=param.detach())

    # ---- hard-token virtual queries (Lq = 1) ------------------------------------------------------------------------
    mod = importlib.import_module("mainv3_iter_liscore_QA_hardtoken")
    rec = mod.score_multi_vector_masked = Recorder(mod.score_multi_vector_masked)
    Qb, qmb, Ptn, pmt, param, pms, opt, hp = fresh(ref_prep)
    out = mod.train_one_step(Qb, qmb, Ptn, pmt, param, pms, opt, lambda_aux=hp["lambda_aux"], virt_noise_std=0.0,
                             aux_docs=hp["aux_docs"], **common(hp))
    assert len(rec.calls) == 4 and rec.calls[2]["Q"].shape[1] == 1
    gaps = (check_margins(rec.calls[0]["out"], "teacher scores"), check_margins(rec.calls[1]["out"], "student scores"))
    print(f"[golden] hardtoken: min adjacent score gaps teacher/student = {gaps[0]:.2e} / {gaps[1]:.2e}; "
          f"{rec.calls[2]['Q'].shape[0]} virtual queries")
    save("v3_hardtoken", q_virtual=rec.calls[2]["Q"], sc_t_v=rec.calls[2]["out"], sc_s_v=rec.calls[3]["out"],
         **{k: np.float64(v) for k, v in out.items() if isinstance(v, float)}, grad=param.grad, param_after=param.detach())
    print("[golden] keys of the hard-token step's return dict:", sorted(out))


if __name__ == "__main__":
    main()
