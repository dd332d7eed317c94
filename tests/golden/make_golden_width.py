#!/usr/bin/env python3
"""Fixtures for embedding widths other than 128, made by RUNNING THE REFERENCE'S OWN
score_multi_vector_masked (evaluator/retrieval.py:166-213, which takes the width from its tensors, :173) and its autograd on the CPU
of the build container: scores, dP, dQ and the arg-max of the same similarities, at d = 64, 200 and 256 (tests/golden_recipes.py
`width_case`).  Only numbers are stored; needs /root/reference, like tests/golden/make_golden.py (same import shims)."""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import golden_recipes as R  # noqa: E402
from make_golden import import_reference, save  # noqa: E402


def main():
    torch.set_num_threads(8)
    ref_retrieval, _, _ = import_reference()
    score = ref_retrieval.score_multi_vector_masked
    for d in (64, 200, 256):
        Q, P, qm, pm, g = R.width_case(d)
        Qg, Pg = Q.clone().requires_grad_(True), P.clone().requires_grad_(True)
        s = score(Qg, Pg, qm, pm, chunk_p=4)
        (s * g).sum().backward()
        sim = torch.einsum("qnd,cmd->qcnm", Q, P).masked_fill(~pm[None, :, None, :], -1e4)
        save(f"a1_width{d}", scores=s.detach(), dP=Pg.grad, dQ=Qg.grad, argmax=sim.max(dim=-1).indices.to(torch.int32))


if __name__ == "__main__":
    main()
