"""Host simulation behind DESIGN.md §9 "fp32 inputs with the cross terms in a cheaper format": error of scoring fp32 inputs as
hi*hi (fp16) + cross terms rounded to OCP e4m3 / block-scaled e2m3, against exact arithmetic, the three-product scheme and an fp32 einsum."""
import torch, math
torch.manual_seed(0)
def unit(*s): return torch.nn.functional.normalize(torch.randn(*s, dtype=torch.float64), dim=-1)
nq, lq, npg, lp, D = 16, 32, 48, 1030, 128
Q = unit(nq, lq, D).float(); P = unit(npg, lp, D).float()
def split(x):
    amax = x.abs().max().item()
    k = 14 - math.floor(math.log2(amax))          # scaled absmax in [2^14, 2^15)
    xs = x.double() * 2.0**k
    hi = xs.half(); lo = (xs - hi.double()).half()
    return hi, lo, k
qh, ql, kq = split(Q); ph, pl, kp = split(P)
def sims(a, b):   # (nq,lq,D) x (npg,lp,D) -> (nq,npg,lq,lp) in float64 from exactly-representable operands, then rounded like an fp32 accumulation? keep f64 to isolate representation error
    return torch.einsum("qnd,pmd->qpnm", a.double(), b.double())
exact = sims(Q, P)
s3 = (sims(qh, ph) + sims(qh, pl) + sims(ql, ph)) * 2.0**(-kq - kp)
def f8(x, scale):   # OCP e4m3fn rounding of x*scale, returned unscaled (float64)
    return (x.double() * scale).float().to(torch.float8_e4m3fn).double() / scale
def score(sim): return sim.max(-1).values.sum(-1)
for (sh, sl) in [(2.0**-7, 2.0**5), (2.0**-7, 2.0**4), (2.0**-6, 2.0**6)]:
    qh8, ph8 = f8(qh, sh), f8(ph, sh)
    ql8, pl8 = f8(ql, sl), f8(pl, sl)
    s8 = (sims(qh, ph) + sims(qh8, pl8) + sims(ql8, ph8)) * 2.0**(-kq - kp)
    print(f"scales hi {sh} lo {sl}: per-sim max err 3prod {((s3-exact).abs().max()):.2e}  fp8cross {((s8-exact).abs().max()):.2e}  std {((s8-exact).std()):.2e};"
          f" score err max 3prod {((score(s3)-score(exact)).abs().max()):.2e} fp8cross {((score(s8)-score(exact)).abs().max()):.2e}"
          f"  argmax flips {(s8.argmax(-1) != exact.argmax(-1)).float().mean():.2e} (3prod {(s3.argmax(-1) != exact.argmax(-1)).float().mean():.2e})")
# hi only, and hi + one-sided
s1 = sims(qh, ph) * 2.0**(-kq - kp)
print(f"hi*hi only: per-sim max {((s1-exact).abs().max()):.2e} score max {((score(s1)-score(exact)).abs().max()):.2e}")
# the fp32 reference's own noise: fp32 einsum vs exact
s32 = torch.einsum("qnd,pmd->qpnm", Q, P).double()
print(f"fp32 einsum: per-sim max {((s32-exact).abs().max()):.2e} score max {((score(s32)-score(exact)).abs().max()):.2e} argmax flips {(s32.argmax(-1) != exact.argmax(-1)).float().mean():.2e}")
# fp6 e2m3 simulation: 1 sign, 2 exp (bias 1), 3 mantissa: values +-{0, 0.125..0.875 (subnormal step .125), 1..1.875, 2..3.75, 4..7.5}; with a power-of-two block scale per 32 elements along d
def f6_block(x, block=32):
    xd = x.double()
    shp = xd.shape
    xb = xd.reshape(*shp[:-1], shp[-1] // block, block)
    amax = xb.abs().amax(-1, keepdim=True).clamp_min(1e-300)
    e = torch.floor(torch.log2(amax)) - 2            # scaled amax in [4, 8): top binade of e2m3
    y = xb / 2.0**e
    mag = y.abs()
    ex = torch.clamp(torch.floor(torch.log2(mag.clamp_min(1e-300))), min=0.0, max=2.0)
    step = 2.0**(ex - 3)
    r = torch.round(mag / step) * step
    r = torch.clamp(r, max=7.5)
    return (torch.sign(y) * r * 2.0**e).reshape(shp)
qh6, ph6, ql6, pl6 = f6_block(qh), f6_block(ph), f6_block(ql), f6_block(pl)
s6 = (sims(qh, ph) + sims(qh6, pl6) + sims(ql6, ph6)) * 2.0**(-kq - kp)
print(f"mx-fp6 cross: per-sim max {((s6-exact).abs().max()):.2e} std {((s6-exact).std()):.2e} score max {((score(s6)-score(exact)).abs().max()):.2e} argmax flips {(s6.argmax(-1) != exact.argmax(-1)).float().mean():.2e}")
