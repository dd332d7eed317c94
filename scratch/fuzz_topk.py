"""Fuzz of the device ranking against a numpy reference: evdr_topk's order (NaN first, score desc with -0 == +0, index asc), the
two-level path for few long rows, idx_map merges, and ops.topk_with_ties (device and host forms) -- random shapes, k, quantised
scores with long tie runs, NaN / +-inf / -0.0.   usage: python scratch/fuzz_topk.py <first_seed> <count>"""
import os, sys, numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import evdr_amd  # noqa: F401
from evdr_amd import ops
dev = "cuda:0"
s0, cnt = int(sys.argv[1]), int(sys.argv[2])
bad = 0


def ref_order(row):
    nan = np.isnan(row)
    key = np.where(nan, np.inf, row.astype(np.float64))
    key = np.where(key == 0, 0.0, key)                     # -0 == +0
    return np.lexsort((np.arange(len(row)), -key, ~nan))   # NaN first, then score desc, then index asc


for seed in range(s0, s0 + cnt):
    rng = np.random.default_rng(seed)
    nq = int(rng.choice([1, 2, 3, 17, 64, 300]))
    n = int(rng.choice([1, 5, 99, 100, 101, 128, 129, 1000, 4095, 4096, 4097, 20000, 100000 if nq <= 3 else 5000]))
    k = int(rng.choice([1, 2, 10, 50, 100, 128]))
    sc = rng.standard_normal((nq, n)).astype(np.float32) * float(rng.choice([1e-3, 1.0, 1e3]))
    style = int(rng.integers(0, 5))
    if style >= 1:
        q = float(rng.choice([0.5, 2.0, 8.0]))
        sc = (np.round(sc * q / max(np.abs(sc).max(), 1e-9) * 4) / 4).astype(np.float32)      # heavy quantisation: long tie runs
    if style >= 2:
        z = sc == 0
        sc[z] = np.where(rng.random(z.sum()) < 0.5, -0.0, 0.0)
    if style >= 3:
        m = rng.random(sc.shape) < 0.01
        sc[m] = np.nan
    if style >= 4:
        m = rng.random(sc.shape) < 0.01
        sc[m] = np.where(rng.random(m.sum()) < 0.5, np.inf, -np.inf)
    t = torch.from_numpy(sc).to(dev)
    kk = k
    ts, ti = ops.topk(t, kk, idx_base=7)
    ts_h, ti_h = ts.cpu().numpy(), ti.cpu().numpy()
    ok = True
    for r in range(nq):
        o = ref_order(sc[r])[:kk]
        want_i = np.full(kk, -1, dtype=np.int64); want_i[: len(o)] = o + 7
        if not np.array_equal(ti_h[r], want_i):
            ok = False; break
        got_s = ts_h[r, : len(o)]
        if not np.array_equal(got_s, sc[r, o], equal_nan=True):
            ok = False; break
    if ok and n > kk:
        for form in (False, True):
            a, b, extra = ops.topk_with_ties(t, kk, to_host=form)
            for r in range(nq):
                o = ref_order(sc[r])
                kth = sc[r, o[kk - 1]]
                if np.isnan(kth):
                    cand = np.nonzero(np.isnan(sc[r]))[0]
                else:
                    kv = 0.0 if kth == 0 else kth
                    cand = np.nonzero(np.isnan(sc[r]) | (np.where(sc[r] == 0, 0.0, sc[r]) >= kv))[0]
                if len(cand) > kk:
                    if r not in extra or not np.array_equal(extra[r][0], cand) or not np.array_equal(extra[r][1], sc[r, cand], equal_nan=True):
                        ok = False; break
                elif r in extra:
                    ok = False; break
            if not ok:
                break
    if not ok:
        bad += 1
        print(f"FAIL seed={seed} nq={nq} n={n} k={k} style={style}", flush=True)
    if (seed - s0) % 50 == 49:
        print(f"... {seed - s0 + 1} seeds, {bad} failures", flush=True)
print(f"done: {cnt} seeds, {bad} failures")
