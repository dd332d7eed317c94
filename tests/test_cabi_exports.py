"""The C-ABI library loads on a CPU-only host and exports exactly what include/evdr.h declares."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import evdr_amd  # noqa: F401
    from evdr_amd import build, _lib
    build.build(verbose=False)
    return _lib.load()


def declared_functions():
    text = open(os.path.join(ROOT, "include", "evdr.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(evdr_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported(lib):
    from evdr_amd import _lib
    names = declared_functions()
    assert len(names) >= 12
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/evdr.h but not exported by libevdr.so"
        assert n in _lib.SIGNATURES, f"{n} has no ctypes signature in _lib.py"
    assert sorted(_lib.SIGNATURES) == names


def test_exports_equal_the_header(lib):
    """`nm -D --defined-only` of libevdr.so == the functions include/evdr.h declares, both ways: the library is built with
    -fvisibility=hidden and only the EVDR_API entry points are dynamic symbols (no C++-mangled internals, no launch helpers).
    The only other dynamic symbols allowed are the HIP compiler's own per-translation-unit id words (`__hip_cuid_*`, data)."""
    import subprocess
    from evdr_amd import _lib
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    syms = [ln.split() for ln in out.splitlines() if ln.strip()]
    funcs = sorted(name for _addr, kind, name in syms if kind in "TtWw")
    other = sorted(name for _addr, kind, name in syms if kind not in "TtWw" and not name.startswith("__hip_cuid_"))
    assert funcs == declared_functions(), (set(funcs) ^ set(declared_functions()))
    assert other == [], other
    text = open(os.path.join(ROOT, "include", "evdr.h")).read()
    for n in declared_functions():                  # every declaration carries the visibility attribute
        assert re.search(r"EVDR_API\s+[\w \*]+?\b" + n + r"\s*\(", text), n


def test_version_and_error_string(lib):
    from evdr_amd import _lib
    assert lib.evdr_version() == _lib.ABI_VERSION == 303
    assert isinstance(lib.evdr_last_error(), bytes)


def test_argument_errors_need_no_gpu(lib):
    """Status codes come back (never an abort) before anything touches the device."""
    from evdr_amd import _lib as L
    rc = lib.evdr_maxsim_fwd(None, None, None, None, None, None, 4, 8, 4, 8, 64, L.EVDR_BF16, None, None, 0, None)
    assert rc == L.EVDR_ERR_SHAPE and b"128" in lib.evdr_last_error()
    rc = lib.evdr_maxsim_fwd(None, None, None, None, None, None, 4, 8, 4, 8, 128, 7, None, None, 0, None)
    assert rc == L.EVDR_ERR_ARG
    rc = lib.evdr_maxsim_fwd(None, None, None, None, None, None, 4, 8, 4, 8, 128, L.EVDR_BF16, None, None, 0, None)
    assert rc == L.EVDR_ERR_ARG          # null Q/P/out
    rc = lib.evdr_maxsim_fwd(None, None, None, None, None, None, -1, 8, 4, 8, 128, L.EVDR_BF16, None, None, 0, None)
    assert rc == L.EVDR_ERR_ARG
    rc = lib.evdr_maxsim_fwd(None, None, None, None, None, None, 0, 8, 4, 8, 128, L.EVDR_BF16, None, None, 0, None)
    assert rc == L.EVDR_OK               # empty score matrix is fine
    rc = lib.evdr_topk(None, None, 3, 10, 10, 0, 500, None, None, None, 0, None)
    assert rc == L.EVDR_ERR_ARG and b"k=500" in lib.evdr_last_error()
    rc = lib.evdr_infonce_distill_fwd_bwd(None, None, 4, 4, ctypes.c_float(0.0), None, None, None, None)
    assert rc == L.EVDR_ERR_ARG
    rc = lib.evdr_split_f32_segments(None, 64, 4096, None, None, None)
    assert rc == L.EVDR_ERR_ARG and b"seg_rows" in lib.evdr_last_error()
    assert lib.evdr_split_f32_segments(None, 0, 1024, None, None, None) == L.EVDR_OK      # nothing to split
    assert lib.evdr_split_f32_segments(None, 64, 32, None, None, None) == L.EVDR_ERR_ARG   # null pointers
    assert lib.evdr_maxsim_fwd_workspace(8, 32, 64, 1030, L.EVDR_F32) > 2 * 64 * 1030 * 128 * 2
    assert lib.evdr_maxsim_fwd_workspace(8, 32, 64, 1030, L.EVDR_BF16) < 64 * 1030 * 4
    assert lib.evdr_maxsim_topk_workspace(10, 10) >= 400
    # round 6: the score-row cache entries and the subset forward refuse bad arguments before touching the device
    assert lib.evdr_maxsim_fwd_prepared_cached(None, None, None, None, None, None, None, None, 8, 4, 64, 1, 64 * 128, 0, None, None, None, 0, None) == L.EVDR_ERR_ARG
    assert b"null cache" in lib.evdr_last_error()
    bad = L.EvdrQCache(1, 6, 1, 1, 1, 1, 1, 1, 3, 16, 2, 8, 0xFFFFFFFFFFFFFFFF)      # pointers are fake: nothing is dereferenced on the host
    assert lib.evdr_maxsim_fwd_prepared_cached(ctypes.byref(bad), None, None, None, None, None, None, None, 8, 4, 64, 1, 64 * 128, 0, None, None, None, 0, None) == L.EVDR_ERR_ARG
    assert b"power of two" in lib.evdr_last_error()
    bad.n_slots = 8
    assert lib.evdr_maxsim_fwd_prepared_cached(ctypes.byref(bad), 1, 1, 1, None, 1, 1, 1, 8, 4, 64, 1, 64 * 128, 0, None, None, None, 0, None) == L.EVDR_ERR_WORKSPACE
    assert lib.evdr_qcache_workspace(32) >= 32 * 16 + 8 and lib.evdr_qcache_workspace(-1) == 0
    rc = lib.evdr_maxsim_fwd_prepared_subset(None, None, None, None, None, None, 8, 4, 40, 8, 64, 1, 64 * 128, 0, None, None, None, None, None)
    assert rc == L.EVDR_ERR_SHAPE and b"lq" in lib.evdr_last_error()


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from evdr_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.load()


def test_header_is_plain_c():
    """include/evdr.h is the boundary a non-Python host binds: it must compile as C99 and as C++11 on its own."""
    import shutil
    import subprocess
    header = os.path.join(ROOT, "include", "evdr.h")
    for cc, std, lang in (("gcc", "-std=c99", "c"), ("g++", "-std=c++11", "c++")):
        if shutil.which(cc) is None:
            pytest.skip(f"{cc} not available")
        r = subprocess.run([cc, std, "-Wall", "-Wextra", "-Werror", "-fsyntax-only", "-x", lang, header],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr


def test_product_library_reads_no_environment_and_has_no_experiment_code(lib):
    """Experiment scaffolding lives in the -DEVDR_EXPERIMENT build only: the shipped library neither imports getenv nor
    exports the diagnostic hooks, and no DIAG (stamped) kernel instance is linked into it."""
    import subprocess
    from evdr_amd import _lib
    und = subprocess.run(["nm", "-D", "--undefined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    assert "getenv" not in und
    exp = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    assert "experiment" not in exp
    # kernel symbols (host stubs) of the staged forward: template argument 6 is DIAG
    stubs = [ln for ln in subprocess.run(["nm", "-C", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout.splitlines()
             if "maxsim_fwd16s_kernel<" in ln]
    assert stubs, "no forward kernel instances found"
    for ln in stubs:
        targs = ln[ln.index("maxsim_fwd16s_kernel<") + len("maxsim_fwd16s_kernel<"):].split(">")[0].split(",")
        assert targs[5].strip() == "false", ln


def test_debug_hooks_round_trip(lib):
    assert lib.evdr_debug_set_fwd_variant(2) == 0 and lib.evdr_debug_set_fwd_variant(0) == 2
    assert lib.evdr_debug_set_pages_per_block(7) == 0 and lib.evdr_debug_set_pages_per_block(0) == 7
    assert lib.evdr_last_fwd_kernel() == b""            # nothing dispatched on this thread yet


def test_debug_hooks_are_thread_local(lib):
    """The overrides are per calling thread (include/evdr.h): what one thread forces is invisible to every other thread, so the
    library keeps no shared mutable state between concurrent callers."""
    import threading
    seen = {}

    def other():
        seen["variant"] = lib.evdr_debug_set_fwd_variant(11)       # previous value IN THIS THREAD: 0, whatever main set
        seen["pages"] = lib.evdr_debug_set_pages_per_block(5)

    assert lib.evdr_debug_set_fwd_variant(2) == 0 and lib.evdr_debug_set_pages_per_block(9) == 0
    try:
        t = threading.Thread(target=other)
        t.start()
        t.join()
        assert seen == {"variant": 0, "pages": 0}
    finally:
        assert lib.evdr_debug_set_fwd_variant(0) == 2               # and the other thread's 11 / 5 never reached this one
        assert lib.evdr_debug_set_pages_per_block(0) == 9


def test_pair_count_overflow_is_rejected(lib):
    """(query, token) pairs are indexed in 32 bits inside the kernels: nq * lq >= 2^31 must come back as a status."""
    from evdr_amd import _lib as L
    rc = lib.evdr_maxsim_bwd(None, None, None, None, None, None, 70_000_000, 32, 4, 8, 128, None)
    assert rc == L.EVDR_ERR_SHAPE and b"2^31" in lib.evdr_last_error()
