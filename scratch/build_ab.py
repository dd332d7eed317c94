"""Build a variant of libevdr.so from a sed-patched COPY of csrc/ (A/B measurements of a source change without keeping a
switch in the product sources).  usage: python scratch/build_ab.py <name> <sed-expr> [<sed-expr> ...]
-> scratch/ab/libevdr_<name>.so ; use it with EVDR_LIB_AB=scratch/ab/libevdr_<name>.so python scratch/headline_ab.py
`--rev <git rev>` as the first expression: the sources of that commit instead of the working tree's."""
import os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
name, exprs = sys.argv[1], sys.argv[2:]
src = os.path.join(ROOT, "efficient-visual-document-retrieval_amd", "csrc")
work = f"/tmp/evdr_ab_{name}"
shutil.rmtree(work, ignore_errors=True)
os.makedirs(work + "/pkg")
if exprs[:1] == ["--rev"]:
    rev, exprs = exprs[1], exprs[2:]
    tar = subprocess.run(["git", "-C", ROOT, "archive", rev, "efficient-visual-document-retrieval_amd/csrc", "include"], check=True, capture_output=True).stdout
    subprocess.run(["tar", "-x", "-C", work], input=tar, check=True)
    os.rename(work + "/efficient-visual-document-retrieval_amd/csrc", work + "/pkg/csrc")
else:
    shutil.copytree(src, work + "/pkg/csrc")
    shutil.copytree(os.path.join(ROOT, "include"), work + "/include")
for f in os.listdir(work + "/pkg/csrc"):
    for e in exprs:
        subprocess.run(["sed", "-i", "-E", e, os.path.join(work, "pkg/csrc", f)], check=True)
flags = ["--offload-arch=gfx950", "-O3", "-fPIC", "-fno-honor-nans", "-std=c++17"]
objs = []
procs = []
for f in sorted(os.listdir(work + "/pkg/csrc")):
    if f.endswith(".hip"):
        o = os.path.join(work, f[:-4] + ".o")
        objs.append(o)
        procs.append(subprocess.Popen(["/opt/rocm/bin/hipcc"] + flags + ["-c", os.path.join(work, "pkg/csrc", f), "-o", o]))
assert all(p.wait() == 0 for p in procs)
out = os.path.join(ROOT, "scratch", "ab", f"libevdr_{name}.so")
os.makedirs(os.path.dirname(out), exist_ok=True)
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs, check=True)
print(out)
