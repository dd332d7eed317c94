#!/usr/bin/env python3
"""BASELINE.json configs[1] at FULL size scored by THE REFERENCE ITSELF (evaluator/retrieval.py:166-213, fp32, chunk_p = 64) on the
build container's CPU: 500 x 500 scores -> tests/golden/config1_full.npz (numbers only; the inputs are re-created from the seeded
recipe tests/golden_recipes.config1_case, 263 MB of pages that are not stored).  About a minute and ~9 GB of RAM (two
(500,64,32,1030) fp32 intermediates per chunk).  Runs only where /root/reference exists."""
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
import golden_recipes as R  # noqa: E402
from make_golden import import_reference, save  # noqa: E402


def main():
    torch.set_num_threads(8)
    ref_retrieval, _, _ = import_reference()
    Q, P, qm, pm, targets = R.config1_case()
    t0 = time.time()
    with torch.no_grad():
        s = ref_retrieval.score_multi_vector_masked(Q, P, qm, pm, chunk_p=64)
    print(f"[golden] reference scored {s.shape[0]} x {s.shape[1]} pairs in {time.time() - t0:.1f} s")
    assert s.dtype == torch.float32 and tuple(s.shape) == (500, 500)
    rank1 = (s.argmax(dim=1) == targets).float().mean().item()
    save("config1_full", scores=s, targets=targets.to(torch.int32), rank1=np.float32(rank1))


if __name__ == "__main__":
    main()
