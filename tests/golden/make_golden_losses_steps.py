#!/usr/bin/env python3
"""Golden fixture for the nine remaining single-step training scripts of the reference -- one `train_one_step` each, same
structure (teacher scores, student scores, a loss from criterion.py, backward, AdamW), different losses:

  mainv2_iter_lambda / _linfo_distill / _lipairwise / _listwise / _pairscore / _ranknce / _ranknet / _score_preserve /
  _super_infonce .py

Produced by RUNNING THE REFERENCE'S OWN FUNCTIONS on the CPU (build container only; needs /root/reference); inputs from
tests/golden_recipes.py:v3_case().  Stored per script: the returned loss value(s), a strided sample of Pbar.grad with its
norm, and the parameter after the AdamW step at the same sample points (numbers only)."""
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
from make_golden_v3 import fresh  # noqa: E402

# script -> extra keyword arguments of its train_one_step (tests/v3_patterns.py: LOSS_STEPS restates the same table)
STEPS = {
    "mainv2_iter_lambda": dict(alpha=1.0, eps=1e-6),
    "mainv2_iter_linfo_distill": dict(k=8, list_temp=2.0, info_temp=0.1, lambda_list=1.0, lambda_info=0.5),
    "mainv2_iter_lipairwise": dict(k=8, temp=2.0, lambda_list=1.0, lambda_pair=0.5),
    "mainv2_iter_listwise": dict(k=8, temp=2.0),
    "mainv2_iter_pairscore": dict(lambda_pair=1.0, lambda_score=0.5),
    "mainv2_iter_ranknce": dict(temperature=0.5, lambda_weight=0.7),
    "mainv2_iter_ranknet": dict(),
    "mainv2_iter_score_preserve": dict(),
    "mainv2_iter_super_infonce": dict(pos_idx=torch.tensor([3, 0, 15, 7, 7, 9]), temp=0.07),
}


def main():
    torch.set_num_threads(8)
    _, _, ref_prep = import_reference()
    arrays = {}
    for script, kw in STEPS.items():
        mod = importlib.import_module(script)
        Qb, qmb, Ptn, pmt, param, pms, opt, hp = fresh(ref_prep)
        if script == "mainv2_iter_super_infonce":          # no teacher: (Qb, qmb, pos_idx, Pbar_param, pmask_student, opt, temp)
            out = mod.train_one_step(Qb, qmb, kw["pos_idx"], param, pms, opt, kw["temp"], chunk_p=64)
        else:
            out = mod.train_one_step(Qb, qmb, Ptn, pmt, param, pms, opt, chunk_p=64, **kw)
        tag = script.replace("mainv2_iter_", "")
        if isinstance(out, dict):
            for k, v in out.items():
                arrays[f"{tag}__{k}"] = np.float64(v)
        else:
            arrays[f"{tag}__loss"] = np.float64(out)
        g = param.grad.detach()
        arrays[f"{tag}__grad_sample"] = g[::2, ::2, ::2].contiguous().numpy()
        arrays[f"{tag}__grad_norm"] = np.float64(g.double().norm())
        arrays[f"{tag}__param_sample"] = param.detach()[::2, ::2, ::2].contiguous().numpy()
        print(f"[golden] {script}: {out if not isinstance(out, dict) else {k: round(v, 6) for k, v in out.items()}}")
    save("v3_loss_steps", **arrays)


if __name__ == "__main__":
    main()
