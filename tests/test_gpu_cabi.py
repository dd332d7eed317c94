"""The C ABI used from a torch-free C++ program (tests/cabi/cabi_smoke.cpp): hipcc-built, linked against libevdr.so only."""
import os
import shutil
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cabi_from_plain_cpp(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available on this box")
    libdir = os.path.join(ROOT, "efficient-visual-document-retrieval_amd")
    assert os.path.exists(os.path.join(libdir, "libevdr.so")), "libevdr.so missing: run __graft_entry__.build()"
    exe = str(tmp_path / "cabi_smoke")
    subprocess.run([hipcc, "-O1", "-std=c++17", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cabi", "cabi_smoke.cpp"),
                    "-L", libdir, "-levdr", f"-Wl,-rpath,{libdir}", "-o", exe], check=True, timeout=300)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "cabi_smoke OK" in r.stdout, (r.returncode, r.stdout, r.stderr)


def test_cabi_from_several_host_threads(tmp_path):
    """Eight host threads, each with its own stream, buffers and problem shape (one per kernel family), start together so that
    kernel instances see their first launch from racing threads; every thread's scores are bit-equal to the same call made
    alone afterwards, and each thread reads its own error text (include/evdr.h, THREADING)."""
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available on this box")
    libdir = os.path.join(ROOT, "efficient-visual-document-retrieval_amd")
    exe = str(tmp_path / "cabi_threads")
    subprocess.run([hipcc, "-O1", "-std=c++17", "-pthread", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cabi", "cabi_threads.cpp"),
                    "-L", libdir, "-levdr", f"-Wl,-rpath,{libdir}", "-o", exe], check=True, timeout=300)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=180)
    assert r.returncode == 0 and "cabi_threads OK" in r.stdout, (r.returncode, r.stdout, r.stderr)


def test_score_row_cache_and_subset_forward_from_plain_cpp(tmp_path):
    """The round-6 entry points from a torch-free C++ program (tests/cabi/cabi_qcache.cpp): a resident bf16 corpus, the score-row
    cache over three batches (all misses / all hits / half known: bit-equal to the plain forward, device-side miss counts 8 / 0 / 4),
    the subset forward on a device-side list, a short workspace refused with a status code."""
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available on this box")
    libdir = os.path.join(ROOT, "efficient-visual-document-retrieval_amd")
    exe = str(tmp_path / "cabi_qcache")
    subprocess.run([hipcc, "-O1", "-std=c++17", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cabi", "cabi_qcache.cpp"),
                    "-L", libdir, "-levdr", f"-Wl,-rpath,{libdir}", "-o", exe], check=True, timeout=300)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "cabi_qcache OK" in r.stdout, (r.returncode, r.stdout, r.stderr)


def test_the_ctypes_stub_printed_in_integration_md_runs():
    """INTEGRATION.md §2 shows the reference-side binding a maintainer would add (a ctypes stub around evdr_maxsim_fwd).  The block
    is taken out of the document, pointed at the built library and executed: it must score like the oracle (no doc rot)."""
    import re
    import numpy as np
    import torch
    from oracle import maxsim_oracle as O
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "INTEGRATION.md")).read()
    blocks = re.findall(r"```python\n(.*?)```", text, flags=re.S)
    stub = next(b for b in blocks if "def maxsim_fwd(Q, P, qmask, pmask)" in b)
    lib = os.path.join(root, "efficient-visual-document-retrieval_amd", "libevdr.so")
    ns = {}
    exec(compile(stub.replace("/path/to/libevdr.so", lib), "INTEGRATION.md:stub", "exec"), ns)
    g = torch.Generator().manual_seed(21)
    Q = torch.nn.functional.normalize(torch.randn(5, 12, 128, generator=g), dim=-1)
    P = torch.nn.functional.normalize(torch.randn(9, 70, 128, generator=g), dim=-1)
    qm = torch.rand(5, 12, generator=g) > 0.2
    pm = torch.rand(9, 70, generator=g) > 0.2
    pm[2] = False
    got = ns["maxsim_fwd"](Q.cuda(), P.cuda(), qm.cuda(), pm.cuda())
    np.testing.assert_allclose(got.cpu().numpy(), O.maxsim_masked(Q, P, qm, pm).numpy(), atol=1e-4, rtol=0)
