"""The run logs used to pin the train.log contract (tests/golden/make_golden_summary.py, tests/test_summary_contract.py): written
with this repo's own logger and the driver's own record builders, on the CPU."""
import logging
import os
import re

import evdr_amd  # noqa: F401
from evdr_amd import driver
from evdr_amd.utils.utils import get_logger, log_json

CASES = [
    {"name": "fractions", "setting": "infonce_distill train", "mf": 5, "dataset": "docvqa",
     "evals": [(0, 0.412, 0.5731, 7.9), (500, 0.61, 0.70211, 3.2), (1000, 0.588, 0.71034, 2.9)]},
    {"name": "perfect", "setting": "run", "mf": 25, "dataset": "ArxivQA", "evals": [(0, 1.0, 1.0, 0.01)]},
    {"name": "zero", "setting": "x", "mf": 10, "dataset": "shift", "evals": [(0, 0.0, 0.0, 11.0), (20, 0.0, 0.0, 10.5)]},
]


def write_log(root, case):
    """<root>/<setting>/mf<k>/<dataset>/train.log like driver.run writes it; returns (run_dir, lines without the time stamp)."""
    run_dir = os.path.join(root, case["setting"], f"mf{case['mf']}", case["dataset"])
    logger, _ = get_logger(run_dir, name=f"contract-{case['name']}-{os.path.basename(root)}", use_tb=False)
    best_r1 = best_nd5 = None
    last = {}
    for step, r1, nd5, loss in case["evals"]:
        metrics = {"Recall": {"Recall@1": r1}, "NDCG": {"NDCG@5": nd5}, "latency": 0.0123}
        driver.log_eval(logger, None, dataset=case["dataset"], mf=case["mf"], step=step, metrics=metrics, loss=loss)
        if step == 0:
            log_json(logger, {"dataset": case["dataset"], "mf": case["mf"], "step": 0, "note": "init Pbar before training"})
        else:
            log_json(logger, {"dataset": case["dataset"], "mf": case["mf"], "step": step, "train/loss": loss, "train/avg_loss": loss + 1,
                              "time_sec": 1.5})
        best_r1, _ = driver.update_best(best_r1, metrics, step, "r1")
        best_nd5, _ = driver.update_best(best_nd5, metrics, step, "nd5")
        last = metrics
    log_json(logger, driver.summary_record(last, best_r1, best_nd5))
    for h in list(logger.handlers):
        h.flush()
        h.close()
        logger.removeHandler(h)
    logging.getLogger(logger.name).handlers.clear()
    text = open(os.path.join(run_dir, "train.log")).read().splitlines()
    return run_dir, [re.sub(r"^\[[^\]]+\]", "[T]", ln) for ln in text]
