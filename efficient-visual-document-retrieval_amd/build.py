"""Build recipe for the gfx950 C-ABI library (libevdr.so) -- explicit hipcc, in-tree output.

`python -m evdr_amd.build` (or `__graft_entry__.build()`) cross-compiles without a GPU.
The .so is git-ignored but travels with the tree to the GPU box.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
LIB_PATH = os.path.join(PKG_DIR, "libevdr.so")
OBJ_DIR = os.path.join(PKG_DIR, "build")
VARIANT_DIR = os.path.join(os.path.dirname(PKG_DIR), "scratch", "_variants")     # control / experiment builds (never loaded by the package)

SOURCES = ["maxsim_fwd.hip", "maxsim_fwd16.hip", "maxsim_bwd.hip", "topk.hip", "prep.hip", "qcache.hip", "evdr_capi.hip"]
HEADERS = [os.path.join(CSRC, "evdr_common.h"), os.path.join(CSRC, "maxsim_device.h"), os.path.join(os.path.dirname(PKG_DIR), "include", "evdr.h")]
# -fno-honor-nans: lets fmaxf chains fold to v_max3_f32 without canonicalising moves (infinities are kept)
# -fvisibility=hidden: the dynamic symbol table holds the EVDR_API entry points of include/evdr.h and nothing else
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-fno-honor-nans", "-fvisibility=hidden", "-std=c++17", "-Wall", "-Wno-unused-function"]
# per-source extras.  maxsim_fwd16.hip: MFMA destinations stay in VGPRs also in a kernel that uses AGPRs (the eight-queries-per-wave
# instance keeps its query fragments there): LLVM otherwise switches such a kernel to the AGPR-destination form and reads every
# accumulator back with v_accvgpr_read; every other instance compiles to the same instructions with or without the option
EXTRA_FLAGS = {"maxsim_fwd16.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"]}


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.sep not in cand or os.path.exists(cand)):
            return cand
    return "hipcc"


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True, experiment: bool = False, sentinel: bool = False,
          ring_fault: int = 0) -> str:
    """Compile every HIP source for gfx950 and link libevdr.so; returns its path.  experiment=True builds
    scratch/_variants/libevdr_exp.so with -DEVDR_EXPERIMENT instead (the stamped diagnostic kernel instances used by scratch/; never
    loaded by the package).  sentinel=True builds libevdr_sentinel.so with -DEVDR_SENTINEL: the same kernels with every
    LDS-DMA piece poisoning its destination first (csrc/maxsim_device.h), loaded only by tests/test_gpu_sentinel.py.
    ring_fault (scratch/sentinel_control.py only; output under scratch/_variants/, never in the package directory): 1 removes the ring hand-over's vmcnt wait (a RAW race, libevdr[_sentinel]_fault.so),
    2 issues the flat kernel's refill in front of the hand-over (a WAR race, libevdr[_sentinel]_faultwar.so), 3 does the same in the
    STAGED two-slot ring -- the headline kernel's -- (libevdr[_sentinel]_faultwar2.so), 4 is build 3 with one wave of each workgroup held
    back in front of its last reads of every stage, i.e. with the race window open by construction (libevdr[_sentinel]_faultwar2held.so)."""
    suffix = "_exp" if experiment else ("_sentinel" if sentinel else "")
    suffix += {0: "", 1: "_fault", 2: "_faultwar", 3: "_faultwar2", 4: "_faultwar2held"}[int(ring_fault)]   # 1: hand-over without its vmcnt wait (RAW); 2 / 3: refill in front of it (WAR: flat / staged ring)
    if ring_fault or experiment:
        # deliberately broken / instrumented-for-experiments builds never sit beside the product library: objects and library go to
        # scratch/_variants/ (git-ignored; control scripts load them by absolute path), so that no option of the test suite can
        # pick one up from the package directory
        # (ring_fault builds one level deeper, in faults/, which .gpurunignore lists: they reach a GPU box only when a control run
        # takes that line out on purpose)
        vdir = os.path.join(VARIANT_DIR, "faults") if ring_fault else VARIANT_DIR
        os.makedirs(vdir, exist_ok=True)
        obj_dir = os.path.join(vdir, "build" + suffix)
        lib_path = os.path.join(vdir, f"libevdr{suffix}.so")
    else:
        obj_dir = OBJ_DIR + suffix
        lib_path = LIB_PATH.replace("libevdr.so", f"libevdr{suffix}.so")
    flags = FLAGS + (["-DEVDR_EXPERIMENT"] if experiment else []) + (["-DEVDR_SENTINEL"] if sentinel else []) + (
        [f"-DEVDR_RING_FAULT={int(ring_fault)}"] if ring_fault else [])
    os.makedirs(obj_dir, exist_ok=True)
    hipcc = _hipcc()
    jobs = []
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        obj = os.path.join(obj_dir, src.replace(".hip", ".o"))
        if force or _stale(obj, [sp] + HEADERS):
            jobs.append((sp, obj))

    def compile_one(job):
        sp, obj = job
        cmd = [hipcc] + flags + EXTRA_FLAGS.get(os.path.basename(sp), []) + ["-c", sp, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {sp}:\n{r.stdout}\n{r.stderr}")
        if verbose:
            print(f"[evdr build] {os.path.basename(sp)} -> {os.path.basename(obj)}", file=sys.stderr)
        return obj

    if jobs:
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            list(ex.map(compile_one, jobs))
    objs = [os.path.join(obj_dir, s.replace(".hip", ".o")) for s in SOURCES]
    if force or jobs or _stale(lib_path, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib_path] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
        if verbose:
            print(f"[evdr build] linked {lib_path}", file=sys.stderr)
    return lib_path


if __name__ == "__main__":
    build(force="--force" in sys.argv, experiment="--experiment" in sys.argv, sentinel="--sentinel" in sys.argv,
          ring_fault=4 if "--ring-fault-war2-held" in sys.argv else 3 if "--ring-fault-war2" in sys.argv else (2 if "--ring-fault-war" in sys.argv else int("--ring-fault" in sys.argv)))
