"""ISA statistics of one kernel from a `hipcc -S --cuda-device-only` listing: instruction counts and, per straight-line run of
MFMAs (a "block"), the number of MFMAs and of scratch / readlane / writelane ops inside.
usage: python scratch/isa_stats.py <file.s> <mangled-name substring>"""
import re, sys
s = open(sys.argv[1]).read().splitlines()
sub = sys.argv[2]
start = next(i for i, l in enumerate(s) if sub in l and not l.startswith(("\t", ".")) and l.split(";")[0].strip().endswith(":"))
end = next(i for i in range(start, len(s)) if "s_endpgm" in s[i])
body = [l.strip() for l in s[start + 1:end]]
ins = [l for l in body if l and not l.startswith((";", ".")) and not l.split(";")[0].strip().endswith(":")]
ops = {}
for l in ins:
    ops[l.split()[0]] = ops.get(l.split()[0], 0) + 1
show = [k for k in ops if k.startswith(("v_mfma", "scratch_", "v_readlane", "v_writelane", "s_load", "s_cbranch", "s_barrier", "ds_read", "global_load_lds", "s_waitcnt", "s_nop", "v_accvgpr"))]
print(f"{s[start][:-1][:90]}: {len(ins)} instructions")
print("  " + ", ".join(f"{k} {ops[k]}" for k in sorted(show)))
# blocks: maximal label-free, branch-free runs
blocks, cur = [], []
for l in body:
    if not l or l.startswith((";", ".")):
        continue
    if l.split(";")[0].strip().endswith(":") or l.startswith(("s_cbranch", "s_branch", "s_endpgm")):
        if cur: blocks.append(cur)
        cur = []
        continue
    cur.append(l)
if cur: blocks.append(cur)
for b in blocks:
    n = sum(1 for l in b if l.startswith("v_mfma"))
    if n >= 32:
        sc = sum(1 for l in b if l.startswith("scratch_"))
        rl = sum(1 for l in b if l.startswith(("v_readlane", "v_writelane")))
        va = sum(1 for l in b if l.startswith("v_") and not l.startswith("v_mfma"))
        sa = sum(1 for l in b if l.startswith("s_") and not l.startswith(("s_waitcnt", "s_nop")))
        print(f"  block: {len(b):5d} instr, {n:4d} mfma, {va:4d} other VALU, {sa:4d} SALU, scratch {sc}, lane-spill {rl}, ds_read {sum(1 for l in b if l.startswith('ds_read'))}, s_nop {sum(1 for l in b if l.startswith('s_nop'))}, waitcnt {sum(1 for l in b if l.startswith('s_waitcnt'))}")
