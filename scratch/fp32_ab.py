"""fp32-input forward (fp16 hi/lo planes): accuracy vs fp64 and time, per EVDR_FWD_VARIANT."""
import _hooks as H
import os, sys, torch
sys.path.insert(0, "."); import evdr_amd
from evdr_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
def unit(*s): return torch.nn.functional.normalize(torch.randn(*s, device=dev), dim=-1)
variants = [int(v) for v in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["0"])]
for nq, np_, lp, am in [(32, 500, 1030, False), (32, 500, 206, True), (500, 6847, 1030, False), (32, 64, 1030, False), (8, 500, 1030, True)]:
    Q, P = unit(nq, 32, 128), unit(np_, lp, 128)
    pm = torch.ones(np_, lp, dtype=torch.bool, device=dev); pm[3] = False; pm[5, lp // 2:] = False; pm[7, ::3] = False
    qm = torch.ones(nq, 32, dtype=torch.bool, device=dev); qm[:, 28:] = False
    sub = slice(0, min(np_, 48))
    sim = torch.einsum("qnd,pmd->qpnm", Q.double(), P[sub].double()).masked_fill(~pm[sub][None, :, None, :], -1e4)
    mx, ix = sim.max(-1)
    want = (mx * pm[sub].any(-1)[None, :, None] * qm[:, None, :]).sum(-1)
    for v in variants:
        H.set_variant(v)
        out, arg = ops.maxsim_forward(Q, P, qm, pm, want_argmax=am)
        torch.cuda.synchronize()
        err = (out[:, sub].double() - want).abs().max().item()
        aerr = -1
        if am:
            valid = qm[:, None, :] & pm[sub].any(-1)[None, :, None]
            aerr = int(((arg[:, sub].long() != ix) & valid).sum())
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 10
        a.record()
        for _ in range(reps): ops.maxsim_forward(Q, P, qm, pm, want_argmax=am)
        b.record(); torch.cuda.synchronize(); ms = a.elapsed_time(b) / reps
        print(f"nq={nq:4d} np={np_:5d} lp={lp:5d} argmax={int(am)} v{v}: {ms*1e3:9.1f} us  {nq*np_*2*32*lp*128*3/ms/1e9:7.1f} TF(x3)  max|err| {err:.2e}  argmax mismatches {aerr}", flush=True)
