"""Inputs of the eval_retrieval fixture (tests/golden/make_golden_eval.py, tests/test_script_helpers.py): the b32n128 training
case re-used as an evaluation set -- 32 test queries with string names, 128 pages with docids in an order unrelated to the page
index, two relevant pages per query (graded)."""
import numpy as np

import golden_recipes as R


def eval_case():
    Qb, qmb, Pt, pmt, Pbar0, pms, hp = R.train_case("b32n128")
    n = Pbar0.shape[0]
    docmap = {str(i): f"page-{(i * 37) % n:03d}" for i in range(n)}
    names = np.empty(Qb.shape[0], dtype=object)
    for i in range(Qb.shape[0]):
        names[i] = f"what is shown on sheet {i}?"
    rel = {str(names[i]): {docmap[str((i * 5) % n)]: 2, docmap[str((i * 11 + 7) % n)]: 1} for i in range(Qb.shape[0])}
    return Qb, qmb, Pbar0, pms, rel, docmap, names
