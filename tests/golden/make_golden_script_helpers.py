#!/usr/bin/env python3
"""Fixtures from the north-star script's OTHER functions (mainv2_iter_distill_infonce.py), run as they are on the CPU in the build
container: `evaluation_loss` (:324-344) on the v3 case, and `update_best` (:373-392) over a sequence of evaluations with ties in
either metric.  Numbers only (tests/golden/script_helpers.json); tests/test_script_helpers.py checks driver.update_best on the CPU
and driver.evaluation_loss on the GPU against them."""
import importlib
import json
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
import golden_recipes as R  # noqa: E402
from make_golden import import_reference  # noqa: E402
from make_golden_v3 import fresh  # noqa: E402
import script_helper_recipe as H  # noqa: E402


def main():
    torch.set_num_threads(8)
    _, _, ref_prep = import_reference()
    mod = importlib.import_module("mainv2_iter_distill_infonce")
    Qb, qmb, Ptn, pmt, param, pms, opt, hp = fresh(ref_prep)
    out = {"evaluation_loss": {}, "update_best": []}
    for temp in (0.1, 0.05, 1.0):
        out["evaluation_loss"][str(temp)] = mod.evaluation_loss(Qb, qmb, Ptn, pmt, param, pms, temp=temp, chunk_p=64)
    for kind in ("r1", "nd5"):
        best, trace = None, []
        for step, r1, nd5 in H.EVALS:
            best, upd = mod.update_best(best, {"Recall": {"Recall@1": r1}, "NDCG": {"NDCG@5": nd5}}, step, kind)
            trace.append({"best": dict(best), "updated": bool(upd)})
        out["update_best"].append({"kind": kind, "trace": trace})
    with open(os.path.join(HERE, "script_helpers.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print(json.dumps(out["evaluation_loss"]), len(out["update_best"][0]["trace"]))


if __name__ == "__main__":
    main()
