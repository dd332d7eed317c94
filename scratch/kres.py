"""Per-kernel register / spill / scratch / LDS figures of one HIP source, from hipcc's -Rpass-analysis=kernel-resource-usage
(cross-compile, no GPU).  usage: python scratch/kres.py [file.hip] [filter substring]"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else "maxsim_fwd16.hip"
flt = sys.argv[2] if len(sys.argv) > 2 else ""
extra = sys.argv[3:]
path = src if os.path.exists(src) else os.path.join(ROOT, "efficient-visual-document-retrieval_amd", "csrc", src)
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-fno-honor-nans", "-std=c++17",
       "-Rpass-analysis=kernel-resource-usage", "-c", path, "-o", "/tmp/_kres.o"] + extra
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = {}
for ln in out.splitlines():
    m = re.search(r"Function Name: (\S+)", ln)
    if m:
        cur = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        cur = re.sub(r"\(anonymous namespace\)::", "", cur).replace("(EvdrFwdParams)", "")
        rows[cur] = {}
        continue
    m = re.search(r"remark:\s+(VGPRs|AGPRs|TotalSGPRs|ScratchSize \[bytes/lane\]|VGPRs Spill|SGPRs Spill|LDS Size \[bytes/block\]|Occupancy \[waves/SIMD\]): (\d+)", ln)
    if m and cur:
        rows[cur][m.group(1).split(" [")[0]] = int(m.group(2))
if not rows:
    print(out[-3000:])
for k, v in rows.items():
    if flt in k:
        print(f"{k:70s} VGPR {v.get('VGPRs',0):3d} AGPR {v.get('AGPRs',0):3d} spill {v.get('VGPRs Spill',0):3d} sspill {v.get('SGPRs Spill',0):3d} scratch {v.get('ScratchSize',0):4d} occ {v.get('Occupancy',0)}")
