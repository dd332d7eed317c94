"""BASELINE.json configs[1] and configs[2] through the drop-in API: score_multi_vector_masked + device top-k, bf16 and fp32."""
import sys, time, torch
sys.path.insert(0, "."); import evdr_amd, bench as B
from evdr_amd import ops
from evdr_amd.evaluator.retrieval import score_multi_vector_masked
dev = torch.device("cuda:0")
P = B.gen_pages(0, 6847, dev)
Qb, _ = B.make_queries(500, 6847, P, 0, 6847, dev, 1)
for name, npg in (("configs[1] docvqa_test_subsampled 500 x 500", 500), ("configs[2] 10-subset corpus 500 x 6847", 6847)):
    for dt in (torch.bfloat16, torch.float32):
        Q, Pp = Qb.to(dt), P[:npg].to(dt)
        if dt == torch.float32:      # genuinely fp32 data: bf16-representable values would leave the lo planes zero (less power, higher clock)
            g = torch.Generator(device=dev).manual_seed(5)
            Pp = torch.nn.functional.normalize(torch.randn(Pp.shape, generator=g, device=dev), dim=-1)
            Q = torch.nn.functional.normalize(Pp[torch.arange(500, device=dev) % npg, :32] + 0.05 * torch.randn((500, 32, 128), generator=g, device=dev), dim=-1)
        qm = torch.ones(500, 32, dtype=torch.bool, device=dev); pm = torch.ones(npg, 1030, dtype=torch.bool, device=dev)
        for _ in range(2):
            s = score_multi_vector_masked(Q, Pp, qm, pm); ts, ti = ops.topk(s, 100)
        torch.cuda.synchronize()
        reps = 5
        t0 = time.perf_counter()
        for _ in range(reps):
            s = score_multi_vector_masked(Q, Pp, qm, pm); ts, ti = ops.topk(s, 100)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / reps * 1e3
        print(f"{name:44s} {str(dt):15s} {ms:8.2f} ms  {500*npg/ms/1e3:7.1f} M pairs/s  {500/ms*1e3:9.0f} queries/s", flush=True)
