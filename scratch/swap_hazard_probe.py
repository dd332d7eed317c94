"""the hazard-table entries of v_permlane16/32_swap in BOTH directions, read off hipcc's own output for the
builtin form (cross-compile, no GPU).  Prints every swap of scratch/probe/swap_hazard.hip with the instructions around it."""
import os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(tempfile.gettempdir(), "swap_hazard.s")
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-S", "--cuda-device-only", os.path.join(ROOT, "scratch", "probe", "swap_hazard.hip"), "-o", out],
               check=True, capture_output=True)
L = [l.strip() for l in open(out) if l.startswith("\t") and not l.strip().startswith((".", ";"))]
s, e = next(i for i, l in enumerate(L) if "global_load" in l), next(i for i, l in enumerate(L) if "s_endpgm" in l)
body = L[s:e + 1]
print("\n".join(body))
for i, l in enumerate(body):
    if l.startswith("v_permlane"):
        print(f"\n{l.split()[0]}:  in front: {body[i-1]!r}   behind: {body[i+1]!r}")
print("\n=> VALU write -> swap read: s_nop 1 (2 wait states) from hipcc; swap write -> VALU read: nothing; swap-derived VALU result -> DPP read: "
      "the ordinary VALU->DPP s_nop 1.  swap16 / swap32 in csrc/maxsim_device.h carry two v_nop in front of the swap inside the asm string and "
      "feed plain VALU (v_max / v_cndmask / v_add) behind it; the DPP row sums read compiler-visible VALU results.")
