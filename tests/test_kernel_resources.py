"""Compile-time resources of the occupancy-critical kernels, read from hipcc's own metadata (cross-compile, no GPU).

* the fused update kernel (csrc/maxsim_bwd.hip) must fit FOUR waves per SIMD -- 128 VGPRs, no scratch: two 512-thread workgroups
  per CU is what overlaps one workgroup's bucketing with the other's HBM stream.  Round 5 put it at exactly 128 (the
  epilogue's first parameter row is prefetched in front of the gather): one register more and the second workgroup is gone, silently.
* every forward instance of csrc/maxsim_fwd16.hip keeps the forms the LDS-DMA ring's ordering argument rests on and uses no scratch
  (scratch/audit_ring_isa.py: hand-over as one asm statement, DMA behind its m0 write, no compiler vmcnt in the loop, no spills)."""
import os
import re
import shutil
import subprocess
import sys
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "efficient-visual-document-retrieval_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"

pytestmark = pytest.mark.skipif(not os.path.exists(HIPCC) and shutil.which("hipcc") is None, reason="hipcc not available")


def _kernel_metadata(listing: str):
    """{demangled-ish symbol: {vgpr_count, private_segment_fixed_size, vgpr_spill_count}} from the .amdgpu_metadata of a listing."""
    out, cur = {}, {}
    for ln in listing.splitlines():
        m = re.match(r"\s+\.(name|vgpr_count|private_segment_fixed_size|vgpr_spill_count):\s+(\S+)", ln)
        if not m:
            continue
        cur[m.group(1)] = m.group(2)
        if len(cur) == 4:
            out[cur["name"]] = {k: int(v) for k, v in cur.items() if k != "name"}
            cur = {}
    return out


def test_fused_update_kernel_fits_four_waves_per_simd():
    from evdr_amd import build as B
    with tempfile.TemporaryDirectory() as tmp:
        lst = os.path.join(tmp, "bwd.s")
        cmd = [HIPCC if os.path.exists(HIPCC) else "hipcc"] + B.FLAGS + ["--cuda-device-only", "-S", os.path.join(CSRC, "maxsim_bwd.hip"), "-o", lst]
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        meta = _kernel_metadata(open(lst).read())
    bwd = {k: v for k, v in meta.items() if "maxsim_bwd_kernel" in k}
    assert len(bwd) == 4, sorted(meta)                                   # <64 | 128 rows> x <dP to HBM | fused update>
    for name, m in bwd.items():
        assert m["vgpr_count"] <= 128, (name, m)                          # 512 threads x 2 workgroups per CU = 4 waves per SIMD
        assert m["private_segment_fixed_size"] == 0 and m["vgpr_spill_count"] == 0, (name, m)


def test_forward_instances_pass_the_ring_isa_audit():
    audit = os.path.join(ROOT, "scratch", "audit_ring_isa.py")
    if not os.path.exists(audit):
        pytest.skip("scratch/audit_ring_isa.py not in this checkout")
    r = subprocess.run([sys.executable, audit], capture_output=True, text=True, timeout=1500, cwd=ROOT)
    tail = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else ""
    assert r.returncode == 0 and re.match(r"^\d+ instances audited, 0 failing$", tail), (r.stdout[-1500:], r.stderr[-500:])
    assert int(tail.split()[0]) >= 25
