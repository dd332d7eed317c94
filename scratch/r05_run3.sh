#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python - <<'PY' > gpurun_out/r05_train_parity_rec.txt 2>&1
import sys, json, torch
sys.path.insert(0, ".")
import evdr_amd, bench_train as BT
torch.set_num_threads(16)
inp = BT.make_inputs(500, 32, torch.device("cuda:0"))
rec = BT.parity_vs_gpu(inp, 500)
print(json.dumps(rec, indent=1))
# where do the large parameter differences sit?
from oracle import maxsim_oracle as O
from evdr_amd import driver
B = 32
Qc, qmc = inp["Qall"][:B].cpu(), inp["qmall"][:B].cpu()
loss_c, grad_c, param_c, sc_t_c, sc_s_c = O.distill_train_step(Qc, qmc, inp["Pt"].cpu(), inp["pmt"].cpu(), inp["Pbar0"].cpu(), inp["pms"].cpu(), 0.1, 1e-3, 1e-2)
teacher = driver.TeacherScorer(inp["Pt"], inp["pmt"]); student = driver.FusedStudent(inp["Pbar0"].clone(), inp["pms"], lr=1e-3, weight_decay=1e-2)
driver.fused_train_one_step(inp["Qall"][:B].contiguous(), inp["qmall"][:B].contiguous(), teacher, student, 0.1)
torch.cuda.synchronize()
pd = (student.x.cpu() - param_c).abs()
g = grad_c.abs()
for lo, hi in ((0, 0), (1e-30, 1e-12), (1e-12, 1e-10), (1e-10, 1e-9), (1e-9, 1e-8), (1e-8, 1e-7), (1e-7, 1e-6), (1e-6, 1e-5), (1e-5, 1)):
    m = (g == 0) if hi == 0 else ((g >= lo) & (g < hi))
    if m.any():
        print(f"|oracle grad| in [{lo:g}, {hi:g}): {int(m.sum()):9d} entries, max |param diff| {pd[m].max().item():.3e}, mean {pd[m].mean().item():.3e}")
# the gradient itself: reconstruct the GPU's from m1 = (1 - beta1) * g after step 1
g_gpu = student.exp_avg.cpu() / 0.1
print("max |grad_gpu - grad_oracle|", (g_gpu - grad_c).abs().max().item(), " max |grad_oracle|", g.max().item())
rel = ((g_gpu - grad_c).abs() / g.clamp_min(1e-30))
for lo, hi in ((1e-12, 1e-10), (1e-10, 1e-9), (1e-9, 1e-8), (1e-8, 1e-7), (1e-7, 1e-6), (1e-6, 1e-5), (1e-5, 1)):
    m = (g >= lo) & (g < hi)
    if m.any():
        print(f"|oracle grad| in [{lo:g}, {hi:g}): max rel grad diff {rel[m].max().item():.3e}, max abs {((g_gpu - grad_c).abs()[m]).max().item():.3e}")
PY
echo "parity rc=$?"; cat gpurun_out/r05_train_parity_rec.txt | tail -40
