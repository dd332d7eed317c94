"""Rate of the 256-wide forward (two column blocks of fp16 hi/lo planes, one query per wave) beside the 128-wide fp32 forward on the same
shape: 500 queries x 500 pages x 1030 patches through score_multi_vector_masked (frozen pages, prepared once), plus the training shape."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); import evdr_amd
from evdr_amd import _lib as L
from evdr_amd.evaluator.retrieval import score_multi_vector_masked
dev = torch.device("cuda:0"); g = torch.Generator(device=dev).manual_seed(1)
for nq, npg, lp in ((500, 500, 1030), (32, 500, 206)):
    for d in (128, 256):
        Q = torch.nn.functional.normalize(torch.randn((nq, 32, d), generator=g, device=dev), dim=-1)
        P = torch.nn.functional.normalize(torch.randn((npg, lp, d), generator=g, device=dev), dim=-1)
        qm = torch.ones(nq, 32, dtype=torch.bool, device=dev); pm = torch.ones(npg, lp, dtype=torch.bool, device=dev)
        with torch.no_grad():
            for _ in range(3): s = score_multi_vector_masked(Q, P, qm, pm)
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(5)]
            for a, b in ev:
                a.record(); s = score_multi_vector_masked(Q, P, qm, pm); b.record()
        torch.cuda.synchronize()
        ms = sum(a.elapsed_time(b) for a, b in ev) / len(ev)
        flop = 2.0 * nq * npg * 32 * lp * d
        print(f"{nq} x {npg} x {lp}, d = {d}: {ms:8.3f} ms  {nq * npg / ms / 1e3:7.2f} M pairs/s  algorithmic {flop / ms / 1e9:7.1f} TFLOP/s  executed x3 {3 * flop / ms / 1e9:7.1f}  {L.load().evdr_last_fwd_kernel().decode()}", flush=True)
