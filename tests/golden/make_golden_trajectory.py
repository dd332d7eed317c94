#!/usr/bin/env python3
"""EIGHT consecutive steps of the north-star script's own `train_one_step` (mainv2_iter_distill_infonce.py:269-292: the reference's
scorer, loss, autograd and torch.optim.AdamW from utils/utils.py:78-80) on the CPU -> tests/golden/a7_trajectory.npz: the loss of
every step, the parameter after steps 1, 4 and 8 and the optimizer's moments after step 8 (numbers only).  Pins what a one-step
fixture cannot: AdamW's bias-correction step count, the moments' evolution and the feedback of the updated pages into the next
forward.  Inputs: tests/golden_recipes.trajectory_case().  Runs only where /root/reference exists."""
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


def main():
    torch.set_num_threads(8)
    _, _, ref_prep = import_reference()
    import utils.utils as ref_utils
    mod = importlib.import_module("mainv2_iter_distill_infonce")
    batches, Pt, pmt, Pbar0, pms, hp = R.trajectory_case()
    Ptn = ref_prep.l2_normalize(Pt * pmt.unsqueeze(-1)).detach()
    param = torch.nn.Parameter(Pbar0 * pms.unsqueeze(-1))
    opt = ref_utils.set_optimizer("adamw", param, hp["lr"], hp["wd"])           # the script's own optimizer factory (:127)
    losses, snaps, gmin = [], {}, []
    for i, (Qb, qmb) in enumerate(batches, 1):
        losses.append(mod.train_one_step(Qb, qmb, Ptn, pmt, param, pms, opt, hp["temp"], chunk_p=64))
        gmin.append(param.grad.abs().clone())
        if i in (1, 4, 8):
            snaps[f"param_after_{i}"] = param.detach().clone()
    st = opt.state[param]
    # elements whose gradient stayed above 1e-6 of the step's largest in EVERY step: AdamW turns a gradient g into
    # lr * m / (sqrt(v) + 1e-8); where |g| ~ 1e-8 summation noise decides the sign of the update, so only these are compared tightly
    # (an exactly-zero gradient -- a patch row no token chose -- is as well-defined as a large one)
    big = torch.stack([(g > 1e-6 * g.max()) | (g == 0) for g in gmin]).all(dim=0)
    print("[golden] losses", [round(x, 6) for x in losses], "elements compared tightly:", float(big.float().mean()))
    save("a7_trajectory", losses=np.array(losses, dtype=np.float64), exp_avg=st["exp_avg"], exp_avg_sq=st["exp_avg_sq"],
         step=np.float64(float(st["step"])), big=big, **snaps)


if __name__ == "__main__":
    main()
