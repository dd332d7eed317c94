"""Positive control of the sentinel instrument (tests/test_gpu_sentinel.py): the same scoring jobs through builds of the same kernels --
  libevdr.so                    product
  libevdr_sentinel.so           product + poison in front of every LDS-DMA piece                 -> must agree with the reference
  libevdr_fault.so              ring hand-over WITHOUT its vmcnt wait (a deliberate RAW race)    -> wrong now and then, by little
  libevdr_sentinel_fault.so     the same race under the sentinel                                 -> wrong by ~1e36 wherever it bites
  libevdr_faultwar.so           flat kernel: refill issued IN FRONT of the hand-over (WAR race)  -> overwrites tiles slower waves still read
  libevdr_sentinel_faultwar.so  the same under the sentinel
  libevdr_faultwar2.so          STAGED two-slot ring (the headline kernel): the whole refill of slot s^1 issued IN FRONT of the hand-over
                                that retires the slower waves' last ds_reads of that slot (WAR race; round 5)
  libevdr_sentinel_faultwar2.so the same under the sentinel
  libevdr_faultwar2held.so      build faultwar2 with the race window held OPEN: one wave of every workgroup sleeps ~14 us in front of its reads of
                                each stage's last tile, so the other waves' early refill is certain to overtake it -> what a WAR race that
                                really happens looks like, plain ...
  libevdr_sentinel_faultwar2held.so ... and under the sentinel
Job A: 256 queries x 2048 pages x 1030 patches (staged kernel; the RAW fault lives in every kernel's hand-over).
Job B: 256 queries x 4096 pages x 200 patches (7 tiles: the flat 3-slot ring, where the WAR fault of build 2 lives; build 3's lives in job A's ring).
Each build runs in a child process (one library handle per process).  What the control shows: how often a real ring race is
visible WITHOUT the poison (the blind spot of bit-compare stress loops) and WITH it.
Where the builds live (round 6): the product and its sentinel twin in the package directory; every deliberately racing build under
scratch/_variants/faults/ (`python -m evdr_amd.build --ring-fault[-war|-war2|-war2-held] [--sentinel]`), which .gitignore AND
.gpurunignore list -- a control run on a GPU box takes the `scratch/_variants/faults/` line out of .gpurunignore for that one call.
usage: python scratch/sentinel_control.py            (parent: runs the children)
       python scratch/sentinel_control.py <libname>  (child)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIBS = ["libevdr.so", "libevdr_sentinel.so", "libevdr_fault.so", "libevdr_sentinel_fault.so", "libevdr_faultwar.so", "libevdr_sentinel_faultwar.so",
        "libevdr_faultwar2.so", "libevdr_sentinel_faultwar2.so", "libevdr_faultwar2held.so", "libevdr_sentinel_faultwar2held.so"]


def child(libname):
    import torch
    import evdr_amd  # noqa: F401
    from evdr_amd import _lib
    # product + sentinel twin live in the package directory; every control build under scratch/_variants/ (absolute path:
    # tests/conftest.py --evdr-lib refuses them, nothing but this script loads one)
    _lib.LIB_PATH = os.path.join(_lib.PKG_DIR if libname in ("libevdr.so", "libevdr_sentinel.so") else os.path.join(ROOT, "scratch", "_variants", "faults"), libname)
    from evdr_amd.corpus import PageCorpus
    dev = torch.device("cuda:0")
    for job, (n, lp, nq, dt) in (("A 2048 x 1030", (2048, 1030, 256, torch.bfloat16)), ("B 4096 x 200 ", (4096, 200, 256, torch.bfloat16)),
                                 ("C 512x1030 f32", (512, 1030, 32, torch.float32))):          # C: the fp16-plane staged instance of the teacher forward
        g = torch.Generator(device=dev).manual_seed(5)
        P = torch.nn.functional.normalize(torch.randn((n, lp, 128), generator=g, device=dev), dim=-1).to(dt)
        Q = torch.nn.functional.normalize(torch.randn((nq, 32, 128), generator=g, device=dev), dim=-1).to(dt)
        ref = torch.empty((nq, n), device=dev)
        for lo in range(0, n, 64):                 # plain torch fp32 reference of the same op (bf16 products are exact in fp32)
            sim = torch.einsum("qnd,pmd->qpnm", Q.float(), P[lo:lo + 64].float())
            ref[:, lo:lo + 64] = sim.max(dim=3).values.sum(dim=2)
        c = PageCorpus.from_tensor(P)
        launches, wrong_launches, wrong_entries, huge_entries, worst = 0, 0, 0, 0, 0.0
        for rep in range(40):
            s = c.score(Q)
            err = (s - ref).abs()
            err = torch.where(torch.isfinite(err), err, torch.full_like(err, 1e38))
            bad = err > 1e-4
            launches += 1
            wrong_launches += int(bad.any().item())
            wrong_entries += int(bad.sum().item())
            huge_entries += int((err > 1e3).sum().item())
            worst = max(worst, err.max().item())
        print(f"{libname:30s} job {job} {_lib.load().evdr_last_fwd_kernel().decode()[:44]:44s} wrong launches {wrong_launches:2d} of {launches}  wrong entries "
              f"{wrong_entries:6d} of {launches * nq * n}  of them > 1e3: {huge_entries:6d}  worst |err| {worst:.3e}", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        child(sys.argv[1])
    else:
        for lib in LIBS:
            if not os.path.exists(os.path.join(ROOT, "efficient-visual-document-retrieval_amd", lib)):
                print(f"{lib}: not built, skipped")
                continue
            r = subprocess.run([sys.executable, os.path.abspath(__file__), lib], cwd=ROOT)
            if r.returncode != 0:
                print(f"{lib}: child exited with {r.returncode}")
