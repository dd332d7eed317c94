"""Round 3: the one token-additivity failure happened inside a full pytest process, after test_gpu_cabi / test_gpu_driver /
test_gpu_edge_semantics and config1 had run in it.  This script rebuilds that context -- the same test files run first IN THIS
PROCESS -- and then repeats config2's sequence (fresh corpus, two equal launches, permuted corpus, masked launches) many times,
reporting any additivity or determinism violation with positions.  usage: python scratch/flake_context.py [reps]"""
import os, sys, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.chdir(R)
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))


def main():
    import pytest
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    rc = pytest.main(["tests/test_gpu_cabi.py", "tests/test_gpu_driver.py", "tests/test_gpu_edge_semantics.py",
                      "tests/test_gpu_fullsize.py::test_config1_docvqa_shape_vs_oracle", "-q", "-m", "gpu", "-p", "no:cacheprovider"])
    print("preamble rc", rc, flush=True)
    import evdr_amd  # noqa: F401
    from evdr_amd.corpus import PageCorpus
    import test_gpu_fullsize as T
    dev = torch.device("cuda:0")
    n, nq = 6847, 256
    bad = 0
    for r in range(reps):
        P, Q, tgt = T.synth(n, nq, dev, seed=12)
        corpus = PageCorpus.from_tensor(P)
        s1 = corpus.score(Q).clone()
        s2 = corpus.score(Q)
        d12 = int((s1 != s2).sum().item())
        perm = torch.randperm(T.LP, device=dev)
        sp = PageCorpus.from_tensor(P[:512][:, perm].contiguous()).score(Q)
        dp = int((sp != s1[:, :512]).sum().item())
        ma = torch.zeros(nq, T.LQ, dtype=torch.bool, device=dev)
        ma[:, ::2] = True
        sa = corpus.score(Q, ma)
        sb = corpus.score(Q, ~ma)
        add = (sa + sb - s1).abs()
        mx = add.max().item()
        if d12 or dp or mx >= 2e-5:
            bad += 1
            where = (add >= 2e-5).nonzero()[:6].tolist()
            again = [(corpus.score(Q) != s1).sum().item(), (corpus.score(Q, ma) != sa).sum().item(), (corpus.score(Q, ~ma) != sb).sum().item()]
            print(f"rep {r}: s1!=s2 in {d12}, permuted != in {dp}, additivity {mx:.3e} at {where}; recomputed differ: {again}", flush=True)
        del P, Q, corpus, s1, s2, sp, sa, sb, add
    print(f"done: {reps} reps of config2's sequence after the suite preamble, {bad} violations")


if __name__ == "__main__":          # children of the suite's mp.spawn re-import __main__: nothing may run at import time
    main()
