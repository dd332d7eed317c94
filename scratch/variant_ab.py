"""A/B of forward dispatch variants (evdr_debug_set_fwd_variant) in one process: python scratch/variant_ab.py <nq> <pages> <rounds> <v1,v2,...>"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); import evdr_amd, bench as B
from evdr_amd import _lib as L
from evdr_amd.corpus import PageCorpus
nq, pages, rounds = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]); variants = [int(v) for v in sys.argv[4].split(",")]
dev = torch.device("cuda:0"); lib = L.load()
P = B.gen_pages(0, pages, dev); corpus = PageCorpus.from_tensor(P, None)
Q, _ = B.make_queries(max(nq, 32), pages, P, 0, pages, dev, 1); Q = Q[:nq].contiguous()
out = torch.empty((nq, pages), dtype=torch.float32, device=dev); res = {v: [] for v in variants}; ref = None
for rnd in range(rounds + 1):
    for v in variants:
        lib.evdr_debug_set_fwd_variant(v)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); corpus.score(Q, None, out=out); b.record(); torch.cuda.synchronize()
        if rnd == 0:
            if ref is None: ref = out.clone()
            else:
                d = (out - ref).abs().max().item(); print(f"  variant {v}: max |diff| vs variant {variants[0]} = {d:.3e} ({'bit-identical' if torch.equal(out, ref) else 'NOT bit-identical'}), kernel {lib.evdr_last_fwd_kernel().decode()}", flush=True)
                assert d < 1e-4
        else: res[v].append(a.elapsed_time(b))
lib.evdr_debug_set_fwd_variant(0)
for v in variants:
    ts = res[v]; print(f"nq={nq} variant {v:3d}: mean {sum(ts)/len(ts):8.4f} ms  min {min(ts):8.4f}", flush=True)
