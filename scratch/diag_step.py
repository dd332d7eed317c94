import sys, torch, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import evdr_amd, golden_recipes as R
from oracle import maxsim_oracle as O
import evdr_amd.evaluator.retrieval as ER, evdr_amd.ops as ops
from evdr_amd.criterion import infonce_distillation_loss
from evdr_amd.utils.preprocess_data import l2_normalize
dev = torch.device("cuda:0")
torch.set_num_threads(16)
Qb, qmb, Pt, pmt, Pbar0, pms, hp = R.train_case("b32n128")
Ptn = O.l2_normalize(Pt * pmt.unsqueeze(-1))
loss, grad, after, sc_t, sc_s = O.distill_train_step(Qb, qmb, Ptn, pmt, Pbar0 * pms.unsqueeze(-1), pms, hp["temp"], hp["lr"], hp["wd"])
param = torch.nn.Parameter((Pbar0 * pms.unsqueeze(-1)).to(dev))
Psb = l2_normalize(param * pms.to(dev).unsqueeze(-1))
s_t = ER.score_multi_vector_masked(Qb.to(dev), Ptn.to(dev), qmb.to(dev), pmt.to(dev))
s_s = ER.score_multi_vector_masked(Qb.to(dev), Psb, qmb.to(dev), pms.to(dev))
l = infonce_distillation_loss(s_s, s_t, hp["temp"]); l.backward()
g = param.grad.cpu()
d = (g - grad).abs()
print("loss", l.item(), loss, "max|dgrad|", d.max().item(), "rows>1e-6:", (d.amax(-1) > 1e-6).sum().item(), "of", d.shape[0]*d.shape[1])
print("norms", g.norm().item(), grad.norm().item())
# argmax comparison
Ps_cpu = O.l2_normalize((Pbar0 * pms.unsqueeze(-1)))
_, arg_o = O.maxsim_masked_argmax(Qb, Ps_cpu, qmb, pms)
_, arg_g = ops.maxsim_forward(Qb.to(dev), Ps_cpu.to(dev), qmb.to(dev), pms.to(dev), want_argmax=True)
arg_g = (arg_g.cpu().to(torch.int32) & 0xFFFF)
neq = (arg_g != arg_o.to(torch.int32))
print("argmax mismatches", neq.sum().item(), "of", neq.numel())
idx = neq.nonzero()[:5]
sim = torch.einsum("qnd,pmd->qpnm", Qb.double(), Ps_cpu.double())
for q,p,n in idx.tolist():
    a, b = arg_o[q,p,n].item(), arg_g[q,p,n].item()
    print(q,p,n, a, b, sim[q,p,n,a].item(), sim[q,p,n,b].item(), "qmask", qmb[q,n].item(), "pm", pms[p,a].item(), pms[p,b].item())
bad = (d.amax(-1) > 1e-6).nonzero()[:8]
print(bad.tolist())
