"""A few forward launches of one shape (for rocprofv3): python scratch/one_launch.py <nq> <pages> [reps] [mask]
EVDR_ONE_LAUNCH_WIDTH=256 (scratch-only switch): the same launch at embedding width 256 in fp32 (the `x2cols` instance, round 6), 128 = fp32 at
the tuned width for comparison; default = the bf16 corpus of the bench."""
import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); import evdr_amd, bench as B
from evdr_amd.corpus import PageCorpus
nq, pages = int(sys.argv[1]), int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
dev = torch.device("cuda:0")
width = int(os.environ.get("EVDR_ONE_LAUNCH_WIDTH", "0"))
if width:
    from evdr_amd.evaluator.retrieval import score_multi_vector_masked
    g = torch.Generator(device=dev).manual_seed(1)
    Q = torch.nn.functional.normalize(torch.randn((nq, 32, width), generator=g, device=dev), dim=-1)
    P = torch.nn.functional.normalize(torch.randn((pages, 1030, width), generator=g, device=dev), dim=-1)
    qm = torch.ones(nq, 32, dtype=torch.bool, device=dev); pm = torch.ones(pages, 1030, dtype=torch.bool, device=dev)
    with torch.no_grad():
        for _ in range(reps): score_multi_vector_masked(Q, P, qm, pm)
    torch.cuda.synchronize()
    sys.exit(0)
P = B.gen_pages(0, pages, dev); corpus = PageCorpus.from_tensor(P, None)
Q, _ = B.make_queries(max(nq, 32), pages, P, 0, pages, dev, 1); Q = Q[:nq].contiguous()
out = torch.empty((nq, pages), dtype=torch.float32, device=dev)
for _ in range(reps): corpus.score(Q, None, out=out)
torch.cuda.synchronize()
