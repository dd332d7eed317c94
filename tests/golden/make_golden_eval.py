#!/usr/bin/env python3
"""Fixture from the north-star script's `eval_retrieval` (mainv2_iter_distill_infonce.py:298-321), run as it is on the CPU in the
build container: its own normalisation, scorer call, all-pairs `.item()` loop and results-dict construction (query keys from
`qsidx_2_query`, docids from `docidx_2_docid`).  The one thing it cannot bring is the metric: mteb is absent, so the `evaluator`
argument is THIS repo's CustomRetrievalEvaluator (the function only calls `.compute_mteb_metrics(qrels, results)` on it) -- the
fixture therefore pins the reference's path from tensors to the (qrels, results) dicts, under this repo's metric on both sides.
Also stored: the smallest score gap around a relevant page, so that the GPU test knows rank order cannot hinge on fp32 noise."""
import importlib
import json
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
from make_golden import import_reference  # noqa: E402
import eval_recipe as E  # noqa: E402


def main():
    torch.set_num_threads(8)
    ref_retrieval, _, ref_prep = import_reference()
    mod = importlib.import_module("mainv2_iter_distill_infonce")
    sys.path.insert(0, ROOT)
    import evdr_amd  # noqa: F401
    from evdr_amd.evaluator.retrieval import CustomRetrievalEvaluator
    Qb, qmb, Pbar0, pms, rel, docmap, names = E.eval_case()
    metrics = mod.eval_retrieval(CustomRetrievalEvaluator(), Qb, qmb, torch.nn.Parameter(Pbar0), pms, rel, docmap, names, chunk_p=64)
    metrics.pop("latency")
    # margins: for every query, the gap between each relevant page's score and its nearest neighbour in the ranking
    P_now = ref_prep.l2_normalize(Pbar0 * pms.unsqueeze(-1))
    sc = ref_retrieval.score_multi_vector_masked(Qb, P_now, qmb, pms, chunk_p=64)
    inv = {v: int(k) for k, v in docmap.items()}
    gap = 1e9
    for i in range(sc.shape[0]):
        for d in rel[str(names[i])]:
            j = inv[d]
            others = torch.cat([sc[i, :j], sc[i, j + 1:]])
            gap = min(gap, float((others - sc[i, j]).abs().min()))
    assert gap > 2e-5, f"a relevant page is only {gap:.2e} from a neighbour: pick another case"
    with open(os.path.join(HERE, "eval_retrieval.json"), "w") as f:
        json.dump({"metrics": metrics, "min_gap_around_relevant_pages": gap}, f, indent=1, sort_keys=True)
    print({k: v for k, v in metrics["NDCG"].items()}, gap)


if __name__ == "__main__":
    main()
