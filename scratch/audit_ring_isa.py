"""ISA audit of the LDS-DMA ring of every forward-kernel instance (cross-compile, no GPU).

The ring's ordering is meant to hold BY CONSTRUCTION (csrc/maxsim_device.h: ring_barrier, lds_dma_16B*, swap16/32); this
script checks the emitted gfx950 listing of every instance for exactly the forms that construction promises, and fails
(exit 1) on any instance that deviates:

  A  every s_barrier of the ring sits INSIDE an inline-asm statement directly behind `s_waitcnt vmcnt(N) lgkmcnt(0)`:
     no LDS access can be scheduled between the waits and the barrier, every ds_read issued before the hand-over is
     retired before the wave arrives (WAR against the refill), every wave's own pieces have landed (RAW).  The only
     barriers outside asm are the three __syncthreads() of the later-token-slice prologue, in front of the first LDS-DMA.
  B  every global_load_lds_dwordx4 sits inside an asm statement behind `s_mov_b32 m0, s*` and an `s_nop` (>= 4 wait
     states when the load takes an SGPR base: the base may come straight from a VALU->SGPR move, which hipcc's hazard
     recognizer cannot see from outside the string).
  C  no scratch traffic (a spill reload waits on vmcnt, i.e. on the ring's DMA in flight; a spill store is a vector store the
     hand count does not know).
  D  every v_permlane16/32_swap sits behind two v_nop inside its asm statement (VALU write -> permlane read, 2 wait states;
     hipcc emits `s_nop 1` for the builtin and nothing on the result side: the only hazard the gfx950 table has for it).
  E  the query windows of the staged kernels (prologue, one wave reads back what it fetched itself, no barrier): walking the
     prologue in layout order, no ds_read while a DMA issued since the last asm `vmcnt(0)` is pending, and no DMA while a
     ds_read issued since the last `lgkmcnt(0)` is outstanding (the window's previous tenant).
  F  no compiler-generated vmcnt wait inside the ring loop (it would mean hipcc is waiting for a load of its own there and
     drains the ring with it).

usage: python scratch/audit_ring_isa.py [-DEVDR_SENTINEL ...]      (extra flags are passed to hipcc)
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "efficient-visual-document-retrieval_amd", "csrc", "maxsim_fwd16.hip")
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-fno-honor-nans", "-std=c++17", "-S", "--cuda-device-only", "-mllvm", "-amdgpu-mfma-vgpr-form"]     # as build.py compiles this source


def listing(extra):
    out = os.path.join(tempfile.gettempdir(), "evdr_audit_fwd16.s")
    r = subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + extra + [SRC, "-o", out], capture_output=True, text=True)
    if r.returncode != 0:
        sys.exit(r.stderr[-3000:])
    return open(out).read().splitlines()


def kernels(lines):
    i = 0
    while i < len(lines):
        m = re.match(r"^(_Z\w+):", lines[i])
        if m:
            j = i + 1
            while not lines[j].startswith(".Lfunc_end"):     # (not the first s_endpgm: an early-exit path may carry its own)
                j += 1
            name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
            name = name.replace("(anonymous namespace)::", "").replace("(EvdrFwdParams)", "").replace("void ", "")
            yield name, lines[i + 1:j + 1]
            i = j
        i += 1


def instructions(body):
    """[(text, in_asm, asm_id)] of real instructions, in layout order."""
    out, in_asm, aid = [], False, 0
    for ln in body:
        t = ln.split(";")[0].strip() if not ln.strip().startswith(";;#") else ln.strip()
        if t.startswith(";;#ASMSTART"):
            in_asm, aid = True, aid + 1
            continue
        if t.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if not t or t.startswith(".") or t.endswith(":"):
            continue
        out.append((t, in_asm, aid if in_asm else 0))
    return out


def audit(name, body):
    ins = instructions(body)
    errs = []
    first_dma = next((k for k, (t, _, _) in enumerate(ins) if t.startswith("global_load_lds")), None)
    if first_dma is None:
        return ["no LDS-DMA found"], {}
    ring_barriers = []
    for k, (t, in_asm, aid) in enumerate(ins):
        op = t.split()[0]
        if op == "s_barrier":
            if in_asm:
                prev, pa, paid = ins[k - 1]
                if not (pa and paid == aid and re.fullmatch(r"s_waitcnt vmcnt\(\d+\) lgkmcnt\(0\)", prev)):
                    errs.append(f"A: asm s_barrier not directly behind 's_waitcnt vmcnt(N) lgkmcnt(0)' (found '{prev}')")
                ring_barriers.append(k)
            elif k > first_dma:
                errs.append("A: compiler-placed s_barrier behind the first LDS-DMA")
        elif op.startswith("global_load_lds"):
            if not in_asm:
                errs.append("B: LDS-DMA outside inline asm")
                continue
            blk = [ins[q][0] for q in range(max(0, k - 4), k) if ins[q][2] == aid]
            nop = next((int(b.split()[1]) for b in reversed(blk) if b.startswith("s_nop")), None)
            has_m0 = any(re.match(r"s_mov_b32 m0, (s\d+|vcc_lo|vcc_hi)$", b) for b in blk)       # (vcc halves are SGPRs: the allocator may hand them out)
            sgpr_base = re.search(r", s\[\d+:\d+\]", t) is not None
            if not has_m0 or nop is None or (sgpr_base and nop < 4):
                errs.append(f"B: LDS-DMA '{t}' without m0 write / wait states in its statement ({blk})")
        elif op.startswith("scratch_"):
            errs.append("C: scratch access (" + op + ")")
        elif op.startswith("v_permlane16_swap") or op.startswith("v_permlane32_swap"):
            ok = in_asm and k >= 2 and ins[k - 1][0] == "v_nop" and ins[k - 2][0] == "v_nop" and ins[k - 1][2] == aid and ins[k - 2][2] == aid
            if not ok:
                errs.append(f"D: '{t}' without two v_nop in its statement")
    if not ring_barriers:
        errs.append("A: no ring barrier found")
        return errs, {}
    # E: prologue walk (first DMA .. first ring barrier)
    pending_dma = outstanding_read = False
    for k in range(first_dma, ring_barriers[0]):
        t, in_asm, _ = ins[k]
        op = t.split()[0]
        if op.startswith("global_load_lds"):
            if outstanding_read:
                errs.append("E: LDS-DMA issued while a ds_read of the window's previous tenant may be outstanding")
                outstanding_read = False
            pending_dma = True
        elif op == "s_waitcnt":
            if in_asm and "vmcnt(0)" in t:
                pending_dma = False
            if "lgkmcnt(0)" in t:
                outstanding_read = False
        elif op.startswith("ds_read"):
            if pending_dma:
                errs.append("E: ds_read in the prologue while an LDS-DMA issued since the last asm vmcnt(0) is pending")
                pending_dma = False
            outstanding_read = True
    # F: compiler vmcnt waits inside the ring loop (between the first and the last ring barrier in layout order and beyond:
    # everything after the first ring barrier is loop body or epilogue)
    comp_vm = [t for (t, in_asm, _) in ins[ring_barriers[0]:] if t.startswith("s_waitcnt") and "vmcnt" in t and not in_asm]
    if comp_vm:
        errs.append(f"F: compiler-generated vmcnt wait(s) behind the first ring barrier: {sorted(set(comp_vm))}")
    stats = {
        "ring_barriers": len(ring_barriers),
        "lds_dma": sum(1 for t, _, _ in ins if t.startswith("global_load_lds")),
        "ds_read": sum(1 for t, _, _ in ins if t.startswith("ds_read")),
        "mfma": sum(1 for t, _, _ in ins if t.startswith("v_mfma")),
        "permlane_swap": sum(1 for t, _, _ in ins if t.startswith("v_permlane")),
    }
    return errs, stats


def main():
    extra = sys.argv[1:]
    bad = 0
    n = 0
    for name, body in kernels(listing(extra)):
        errs, st = audit(name, body)
        n += 1
        tag = "FAIL" if errs else "ok  "
        print(f"{tag} {name:68s} " + " ".join(f"{k}={v}" for k, v in st.items()))
        for e in sorted(set(errs)):
            print(f"       {e}  (x{errs.count(e)})")
        bad += bool(errs)
    print(f"{n} instances audited, {bad} failing")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
