#!/usr/bin/env python3
"""Headline benchmark: late-interaction retrieval over a synthetic 100k-page corpus (BASELINE.json configs[3]).

One "step" = one pass of the hot path over one query batch: 1024 queries (32 tokens x 128 dims, bf16) are
scored against EVERY page of the corpus (1030 patches x 128 dims bf16 per page, resident in HBM), each rank
keeps its shard's top-100 per query, one all-gather (RCCL) exchanges the candidates and every rank merges.
The corpus is sharded by pages over the N ranks (strong scaling: 100k pages in total whatever N is).

  python bench.py --gpus 1 --steps 5 --warmup 1
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (contract in the task statement): value = query-page pairs scored per second
over the whole job, plus `roofline` (dominant kernel = the fused MaxSim MFMA kernel, timed with HIP events on
the launch stream) and `cpu_baseline` (the oracle's torch-CPU restatement of the reference scorer, timed on
this box's host cores on a bounded slice; rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SEED = 20261004
LP, D, LQ = 1030, 128, 32
FLOP_PER_PAIR = 2 * LQ * LP * D            # 8 437 760
MFMA_BF16_PEAK_TFLOPS = 2500.0             # dense bf16, MI355X_MICROARCH.md "Chip-level parameters"
GEN_CHUNK = 100                            # pages per deterministic generation chunk


def gen_pages(lo: int, hi: int, dev) -> torch.Tensor:
    """Pages [lo, hi) of the synthetic corpus: unit-norm Gaussian patches rounded to bf16.  Chunk-seeded, so a
    page's content does not depend on how the corpus is sharded."""
    out = torch.empty((hi - lo, LP, D), dtype=torch.bfloat16, device=dev)
    g = torch.Generator(device=dev)
    c0 = lo // GEN_CHUNK
    c1 = (hi + GEN_CHUNK - 1) // GEN_CHUNK
    for c in range(c0, c1):
        g.manual_seed(SEED + c)
        x = torch.randn((GEN_CHUNK, LP, D), generator=g, device=dev)
        x = torch.nn.functional.normalize(x, dim=-1).bfloat16()
        a, b = max(lo, c * GEN_CHUNK), min(hi, (c + 1) * GEN_CHUNK)
        out[a - lo:b - lo] = x[a - c * GEN_CHUNK:b - c * GEN_CHUNK]
    return out


def make_queries(nq: int, n_pages: int, shard, lo: int, hi: int, dev, world: int, group=None, multi=None):
    """Planted queries: query i targets page t_i = (i*7919) mod N; token n = normalise(P[t_i, pi_i(n)]
    + 0.5 eps).  Each rank fills the queries whose target lives in its shard; an all-reduce (setup, untimed) sums."""
    import torch.distributed as dist
    g = torch.Generator(device="cpu").manual_seed(SEED - 1)
    targets = (torch.arange(nq) * 7919) % n_pages
    rows = torch.stack([torch.randperm(LP, generator=g)[:LQ] for _ in range(nq)])          # (nq, LQ)
    eps = torch.nn.functional.normalize(torch.randn((nq, LQ, D), generator=g), dim=-1)      # unit-norm noise
    Q = torch.zeros((nq, LQ, D), dtype=torch.float32, device=dev)
    mine = ((targets >= lo) & (targets < hi)).nonzero().flatten()
    if len(mine):
        t_local = (targets[mine] - lo).to(dev)
        base = shard[t_local[:, None], rows[mine].to(dev)].float()                          # (m, LQ, D)
        Q[mine.to(dev)] = torch.nn.functional.normalize(base + 0.5 * eps[mine].to(dev), dim=-1)
    if world > 1 or multi:
        if dist.get_backend(group) == "gloo":
            host = Q.cpu()
            dist.all_reduce(host, group=group)
            Q = host.to(dev)
        else:
            dist.all_reduce(Q, group=group)
    return Q.bfloat16(), targets


def host_cores() -> int:
    """Cores this process may really use: affinity mask, capped by the cgroup CPU quota (a GPU box hands one
    GPU's job a 16-core share of a 256-thread host) -- oversubscribing MKL beyond it is several times slower."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, int(os.environ.get("EVDR_CPU_THREADS", "16"))))


def cpu_model() -> str:
    """The host CPU the baseline legs ran on ("core count and CPU model printed in the report")."""
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.lower().startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    import platform
    return platform.processor() or platform.machine()


def cpu_baseline_leg(dev_corpus_slice: torch.Tensor, Qdev: torch.Tensor, reps: int = 3):
    """The reference's scorer as restated in oracle/ (torch fp32 on the host, chunk_p=64), on a bounded slice of the
    SAME workload: 32 queries x 2048 pages, median of `reps` passes (about 15 s of host work in all).  A
    reported baseline, not a target."""
    from oracle import maxsim_oracle as O
    cores = host_cores()
    torch.set_num_threads(cores)
    P = dev_corpus_slice.float().cpu()
    Q = Qdev[:32].float().cpu()
    qm = torch.ones(Q.shape[:2], dtype=torch.bool)
    pm = torch.ones(P.shape[:2], dtype=torch.bool)
    O.maxsim_masked(Q[:4], P[:64], qm[:4], pm[:64], chunk_p=64)            # warm the thread pool
    times = []
    for _ in range(reps):
        t0 = time.perf_counter()
        s = O.maxsim_masked(Q, P, qm, pm, chunk_p=64)
        times.append(time.perf_counter() - t0)
    dt = sorted(times)[len(times) // 2]
    pairs = Q.shape[0] * P.shape[0]
    return {"value": pairs / dt, "unit": "pairs/s", "cores": cores, "cpu_model": cpu_model(), "kind": "port",
            "sample": f"{Q.shape[0]} queries x {P.shape[0]} pages of the same corpus, chunk_p=64, torch fp32 CPU, median of "
                      f"{reps} passes ({', '.join(f'{t:.2f}' for t in times)} s), {cores} threads"}, s


def visible_gpu_count():
    """GPUs on this host WITHOUT any HIP / HSA call (the launcher parent must stay GPU-free): KFD topology nodes that have
    SIMDs, narrowed by the *_VISIBLE_DEVICES variables.  None when the KFD sysfs tree is not visible (then only the workers
    can tell, and they do: see main)."""
    import glob
    import re
    if not os.path.isdir("/sys/class/kfd/kfd/topology/nodes"):
        return None
    n = 0
    for props in glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties"):
        try:
            m = re.search(r"^simd_count\s+(\d+)", open(props).read(), re.M)
        except OSError:
            continue
        if m and int(m.group(1)) > 0:
            n += 1
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(",") if x.strip()]))
    return n


def self_launch(n: int, backend: str, need_gpus: bool = True) -> int:
    """`python bench.py --gpus N` without a launcher: start N fresh worker processes of this script, one per GPU, with the
    rendezvous environment torch.distributed.run would give them, and wait.  This parent never touches the GPU: devices are
    counted from the KFD sysfs tree (`visible_gpu_count`), not through torch.cuda (which falls through to hipGetDeviceCount
    when amdsmi is missing), and nothing is exec'ed from a GPU-initialised process; a worker that finds fewer GPUs than
    ranks stops with the same message.  Rank 0's JSON line goes to the inherited stdout.  Returns the exit code."""
    import socket
    import subprocess
    ndev = visible_gpu_count()
    if need_gpus and backend == "nccl" and ndev is not None and ndev < n and not os.environ.get("EVDR_BENCH_ALLOW_SHARED_GPU"):
        print(f"bench.py: --gpus {n} with backend nccl needs {n} visible GPUs, found {ndev} "
              f"(--backend gloo rehearses N>1 on fewer GPUs)", file=sys.stderr)
        return 2
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0",
                   EVDR_BENCH_LAUNCHER="self")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    try:
        pending = set(range(n))
        while pending:
            for r in sorted(pending):
                code = procs[r].poll()
                if code is None:
                    continue
                pending.discard(r)
                if code != 0 and rc == 0:
                    rc = code
                    for o in pending:                    # one rank failed: the others would wait in a collective forever
                        procs[o].terminate()
            time.sleep(0.05)
    finally:
        for pr in procs:
            if pr.poll() is None:
                pr.kill()
    return rc


def extras(dev, corpus_pages: torch.Tensor, args):
    """The two other halves of the path, measured AFTER the timed retrieval region (N=1, rank 0; about 3 s in all) so that
    the driver-run line carries them too:
      train_step -- BASELINE.json configs[4] (mainv2_iter_distill_infonce.py:269-292) at B=32, N=500, Lt=1030, Ls=206: ms per step
                    of the reference's call pattern, of the fused step and of the fused step with cached teacher scores, the
                    kernel rooflines (executed AND algorithmic FLOP, named basis) and the oracle step on a bounded sample;
      eval       -- BASELINE.json configs[1] through driver.eval_retrieval (mainv2_iter_distill_infonce.py:298-321): 500 planted
                    queries x the first 500 pages of the corpus, end to end, split into device / copy / host-metric time, in
                    bf16 (as configs[1] names it) and in fp32 (what the reference's scripts hand over)."""
    import bench_train
    from evdr_amd import driver
    from evdr_amd.evaluator.retrieval import CustomRetrievalEvaluator
    # cpu_pages = 500: the oracle step at the bench's OWN size (about 2.5 s of host work), so that its `cpu_baseline` also carries
    # the step-level parity of configs[4] at N = 500 (loss / parameters / arg-max against the fused GPU step on the same inputs)
    train = bench_train.measure(pages=500, batch=32, steps=30, warmup=15, kinds=["call_pattern", "call_pattern_cached", "fused", "fused_cached"],
                                cpu_pages=500, cpu_reps=1, dev=dev)
    n = min(500, corpus_pages.shape[0])
    pages = corpus_pages[:n]
    Qe, targets = make_queries(500, n, pages, 0, n, dev, 1)
    docmap = {str(j): f"doc{j}" for j in range(n)}
    qrels = {str(i): {docmap[str(int(t))]: 1} for i, t in enumerate(targets.tolist())}
    pm = torch.ones(pages.shape[:2], dtype=torch.bool, device=dev)
    qm = torch.ones(Qe.shape[:2], dtype=torch.bool, device=dev)
    ev = CustomRetrievalEvaluator()
    rec = {"config": {"workload": "driver.eval_retrieval, BASELINE.json configs[1] shape: 500 queries x 500 pages x 1030 patches, top-100, "
                                  "metric tables at k = 1,3,5,10,50,70,100", "queries": 500, "pages": n},
           "note": "median of 5 calls after 2 warm-ups; total_ms = the whole call incl. the page normalisation; device_ms = score + "
                   "top-k by HIP events; d2h_ms = tie-rule candidate counts + the ONE device-to-host copy; host_ms = metric tables "
                   "(numpy); nDCG under this repo's trec_eval-semantics metric (parity with mteb unpinned)"}
    for name, (Qx, Px) in {"bf16": (Qe, pages), "fp32": (Qe.float(), pages.float())}.items():
        runs = []
        for i in range(7):
            t = {}
            m = driver.eval_retrieval(ev, Qx, qm, Px, pm, qrels, docmap, None, k=100, timing=t)
            if i >= 2:
                runs.append(t)
        med = {k: sorted(r[k] for r in runs)[len(runs) // 2] for k in runs[0]}
        rec[name] = {**med, "ndcg_at_5": m["NDCG"]["NDCG@5"], "recall_at_1": m["Recall"]["Recall@1"],
                     "latency_ms_per_query": m["latency"]}
    return train, rec


def configs2_record(dev, corpus_pages_fn, n_pages: int = 6847, nq: int = 500, reps: int = 5):
    """BASELINE.json configs[2] -- the 10-subset ProxyQ/ViDoRe corpus shape, 500 queries x 6 847 pages x 1030 patches -- through the
    drop-in API `score_multi_vector_masked` (evaluator/retrieval.py:166-213), in bf16 (as configs[2] names the patch embeddings) and in
    fp32 (what the reference's scripts hand over, :176-177: genuinely fp32 data, not bf16-representable values, so that the lo planes
    are not zero).  After the timed region, N = 1.  Per dtype: the whole call (mean of `reps` by HIP events: query preparation + MaxSim
    kernel; the pages' preparation is cached per tensor after the first call, as in an evaluation loop) and the MaxSim kernel alone on
    the prepared operands, with its symbol and a roofline entry -- `frac` on the EXECUTED matrix FLOP (three fp16-plane products per
    fp32 product) and `frac_algorithmic` on 2*Lq*Lp*D per pair."""
    from evdr_amd import _lib as L
    from evdr_amd import ops
    from evdr_amd.evaluator import retrieval as R
    rec = {"config": {"workload": "BASELINE.json configs[2] shape through score_multi_vector_masked: 500 queries x 6847 pages x 1030 patches, "
                                  "all-valid masks, synthetic pages of the headline generator", "queries": nq, "pages": n_pages},
           "note": f"call_ms / kernel_ms = MEAN of {reps} launches after 2 warm-ups, HIP events on the launch stream; kernel_ms_min beside it; "
                   "call = query preparation + MaxSim kernel on per-tensor cached page operands (an evaluation loop's steady state)"}
    Pb = corpus_pages_fn(n_pages)
    Qb, _ = make_queries(nq, n_pages, Pb, 0, n_pages, dev, 1)
    qm = torch.ones(nq, LQ, dtype=torch.bool, device=dev)
    pm = torch.ones(n_pages, LP, dtype=torch.bool, device=dev)

    def timed(fn):
        for _ in range(2):
            fn()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
        for a, b in ev:
            a.record()
            fn()
            b.record()
        torch.cuda.synchronize()
        each = [a.elapsed_time(b) for a, b in ev]
        return sum(each) / len(each), min(each)

    for name in ("bf16", "fp32"):
        if name == "bf16":
            Q, P = Qb, Pb
        else:
            g = torch.Generator(device=dev).manual_seed(SEED + 5)
            P = torch.nn.functional.normalize(torch.randn((n_pages, LP, D), generator=g, device=dev), dim=-1)
            Q = torch.nn.functional.normalize(P[torch.arange(nq, device=dev) % n_pages, :LQ]
                                              + 0.05 * torch.randn((nq, LQ, D), generator=g, device=dev), dim=-1)
        with torch.no_grad():
            call_ms, call_min = timed(lambda: R.score_multi_vector_masked(Q, P, qm, pm))
            planes, amax, tilemask, pageflags = R._prepared_pages(P, pm)
            qplanes, qamax = (Q.contiguous()[None], None) if name == "bf16" else ops.split_f32(Q)
            out = torch.empty((nq, n_pages), dtype=torch.float32, device=dev)
            k_ms, k_min = timed(lambda: ops.maxsim_forward_prepared(qplanes, qamax, planes, amax, qm, tilemask, pageflags, out=out))
        sym = L.load().evdr_last_fwd_kernel().decode()
        algo = nq * n_pages * FLOP_PER_PAIR
        executed = algo * (1 if name == "bf16" else 3)
        tf = executed / (k_ms * 1e-3) / 1e12
        top1 = out.argmax(dim=1)
        want = (torch.arange(nq, device=dev) * 7919) % n_pages if name == "bf16" else torch.arange(nq, device=dev) % n_pages
        rec[name] = {"call_ms": call_ms, "call_ms_min": call_min, "pairs_per_s": nq * n_pages / (call_ms * 1e-3),
                     "queries_per_s": nq / (call_ms * 1e-3), "planted_top1": float((top1 == want).float().mean().item()),
                     "roofline": {"bound": "mfma", "kernel": sym, "kernel_ms": k_ms, "kernel_ms_min": k_min,
                                  "kernel_ms_basis": f"mean of {reps} launches of the prepared-operand forward alone, HIP events",
                                  "algorithmic_flop_per_launch": algo, "executed_flop_per_launch": executed,
                                  "achieved": tf, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / MFMA_BF16_PEAK_TFLOPS,
                                  "frac_basis": "EXECUTED matrix FLOP (bf16: one product per k-step; fp32: three fp16-plane products) over the "
                                                "dense bf16 / fp16 MFMA peak",
                                  "frac_algorithmic": algo / (k_ms * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS}}
        del P, Q, planes, qplanes, out
        R.forget_prepared()
        torch.cuda.empty_cache()
    return rec


class _quiet_stdout:
    """File descriptor 1 points at stderr while a process group is formed: Gloo's C++ side prints "[Gloo] Rank r is connected to ..."
    lines to stdout, and stdout is where rank 0's ONE JSON line goes (the driver parses it)."""

    def __enter__(self):
        sys.stdout.flush()
        self._keep = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self._keep, 1)
        os.close(self._keep)
        return False


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--pages", type=int, default=100000, help="total corpus pages (headline config: 100000)")
    ap.add_argument("--queries", type=int, default=1024, help="queries per step")
    ap.add_argument("--topk", type=int, default=100)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-regimes", action="store_true",
                    help="skip the 1- and 8-query streaming launches (PMC passes aggregate over the launches they see)")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the `train_step` (configs[4]) and `eval` (configs[1]) records that follow the timed region at N=1")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="collective backend for N>1 (nccl = RCCL over xGMI; gloo only to rehearse N>1 on a 1-GPU box)")
    ap.add_argument("--dist-at-one", action="store_true",
                    help="with --gpus 1: still form the process groups (gloo control + RCCL data group of ONE rank) and run the barrier, "
                         "the step-clock all-reduce and the candidate exchange of the phase breakdown -- a rehearsal of the N > 1 code path "
                         "on a 1-GPU box; the timed step itself has nothing to exchange")
    ap.add_argument("--rendezvous-only", action="store_true",
                    help="start the ranks, form the process group, print its size and exit (launch check; no GPU work)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args.gpus, args.backend, not args.rendezvous_only))   # plain `python bench.py --gpus N`: N fresh workers
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher's world size must equal --gpus")
    import torch.distributed as dist
    multi = world > 1 or args.dist_at_one             # process groups exist (world 1 only as the --dist-at-one rehearsal)
    if multi and world == 1 and "MASTER_ADDR" not in os.environ:
        import socket
        with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
            sk.bind(("127.0.0.1", 0))
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(sk.getsockname()[1]), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC for RCCL; before the first HIP call
    launcher = os.environ.get("EVDR_BENCH_LAUNCHER", "torchrun" if "WORLD_SIZE" in os.environ else "none")
    if args.rendezvous_only:
        if world > 1:
            with _quiet_stdout():
                dist.init_process_group("gloo")
                seen = torch.ones(1)
                dist.all_reduce(seen)
            if rank == 0:
                print(json.dumps({"rendezvous": "ok", "world_size": dist.get_world_size(), "ranks_seen": int(seen.item()),
                                  "launcher": launcher}), flush=True)
            dist.destroy_process_group()
        else:
            print(json.dumps({"rendezvous": "ok", "world_size": 1, "ranks_seen": 1, "launcher": launcher}), flush=True)
        return
    # EVDR_BENCH_ALLOW_SHARED_GPU=1: rehearsal hook -- lets several nccl ranks land on one GPU, which RCCL refuses; that refusal
    # is how the gloo fallback below is exercised on a 1-GPU box
    if world > 1 and args.backend == "nccl" and torch.cuda.device_count() < world and not os.environ.get("EVDR_BENCH_ALLOW_SHARED_GPU"):
        print(f"bench.py: --gpus {world} with backend nccl needs {world} visible GPUs, found {torch.cuda.device_count()} "
              f"(--backend gloo rehearses N>1 on fewer GPUs)", file=sys.stderr)
        raise SystemExit(2)
    dev_index = local_rank % max(torch.cuda.device_count(), 1)     # gloo rehearsal: several ranks may share a GPU
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    backend, fallback_reason, ranks_seen = args.backend, None, 1
    group = None                                      # the DATA group: candidates, score columns, barriers, the step clock
    if multi:
        import datetime
        # The default group is ALWAYS gloo: a control plane that needs no GPU and cannot fail the way a first RCCL run can.  The
        # data group (nccl = RCCL over xGMI) is formed next to it, and whether it is usable is AGREED over the control plane: every
        # rank min-reduces its own verdict, so either all ranks use RCCL or all of them fall back to gloo through host memory
        # (0.8 MB per rank and step) -- a one-sided failure can no longer split the job between two backends -- and the line says
        # which it was.  The control group's timeout is longer than the data group's: a rank whose RCCL probe failed at once
        # waits there for the ranks whose probe is still running into its timeout.
        with _quiet_stdout():
            dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=900))
            dist.all_reduce(torch.ones(1))                # (Gloo connects its pairs lazily: the chatter belongs in here)
        ok_mine, why = 1, None
        if backend == "nccl":
            # a ONE-SIDED RCCL failure (group creation or the probe raising on one rank only) leaves the peers blocked in the probe's
            # all-reduce until the data group's 240-s timeout; torch's DEFAULT async error handling then lets the NCCL watchdog tear the
            # process down with a non-zero exit, `self_launch` / torchrun end the other ranks, and the job fails fast and loudly.  That
            # default is kept on purpose (it also guards every timed step's collective): TORCH_NCCL_ASYNC_ERROR_HANDLING=0 would mean
            # "no handling" -- a silent hang, not an exception.  The agreed gloo fallback below therefore covers the failures that RAISE
            # (every rank refusing, group creation failing everywhere); a one-sided hang ends the job.  Unrehearsed on a 1-GPU pool.
            try:
                with _quiet_stdout():                   # (RCCL prints its version banner to stdout when the communicator is made)
                    group = dist.new_group(backend="nccl", timeout=datetime.timedelta(seconds=240), device_id=dev)
                    probe = torch.ones(1, device=dev)
                    dist.all_reduce(probe, group=group)
                    torch.cuda.synchronize()
                ranks_seen = int(probe.item())
                ok_mine = int(ranks_seen == world)
                why = None if ok_mine else f"RCCL all-reduce saw {ranks_seen} of {world} ranks"
            except Exception as e:                       # noqa: BLE001
                ok_mine, why = 0, f"{type(e).__name__}: {str(e)[:300]}"
            verdict = torch.tensor([ok_mine], dtype=torch.int32)
            dist.all_reduce(verdict, op=dist.ReduceOp.MIN)            # control plane (gloo)
            if int(verdict.item()) == 0:
                reasons = [None] * world
                dist.all_gather_object(reasons, why)
                fallback_reason = "; ".join(f"rank {r}: {w}" for r, w in enumerate(reasons) if w) or "a peer rank could not use RCCL"
                backend, group = "gloo", None
        if backend == "gloo":
            group = dist.group.WORLD
            probe = torch.ones(1)
            dist.all_reduce(probe, group=group)
            ranks_seen = int(probe.item())

    import evdr_amd  # noqa: F401
    from evdr_amd.corpus import PageCorpus, ShardedRetriever, gather_candidates, shard_range
    from evdr_amd.evaluator.retrieval import CustomRetrievalEvaluator
    from evdr_amd.evaluator.metrics import results_from_topk

    # ---- resident corpus shard + replicated queries (setup, untimed)
    lo, hi = shard_range(args.pages, rank, world)
    shard_pages = gen_pages(lo, hi, dev)
    corpus = PageCorpus.from_tensor(shard_pages, None, idx_base=lo)
    Q, targets = make_queries(args.queries, args.pages, shard_pages, lo, hi, dev, world, group, multi)
    retriever = ShardedRetriever(corpus, group)

    def step():
        return retriever.search(Q, None, args.topk)

    def barrier():
        if backend == "nccl":
            dist.barrier(group=group, device_ids=[dev_index])       # this rank's GPU, stated explicitly
        else:
            dist.barrier(group=group)

    def fence():
        if multi:
            barrier()
        torch.cuda.synchronize()

    if multi:
        # setup, like the corpus: the data group's FIRST all-gather (RCCL sets its rings / channels up lazily, per collective kind)
        # happens here, on a dummy message of the step's own shape, so that it can never land in the timed region -- also with --warmup 0
        gather_candidates(torch.zeros((args.queries, args.topk), dtype=torch.float32, device=dev),
                          torch.zeros((args.queries, args.topk), dtype=torch.int32, device=dev), group)
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    # the dominant kernel of every TIMED step bracketed by a HIP-event pair on its launch stream.  The timed step is the production
    # path itself (ShardedRetriever.search -> PageCorpus.topk: evdr_maxsim_fwd_prepared + evdr_topk); the two event records are all
    # that `score_events` adds to it
    corpus.score_events = []
    for _ in range(min(2, args.warmup)):
        step()                      # (the first event pair allocates)
    corpus.score_events.clear()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ts, ti = step()
    fence()
    elapsed = time.perf_counter() - t0
    step_events, corpus.score_events = corpus.score_events, None
    if multi:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX, group=group)
        elapsed = float(tmax.item())
    ms_per_step = 1e3 * elapsed / max(args.steps, 1)
    pairs_per_step = args.queries * args.pages
    value = pairs_per_step / (ms_per_step * 1e-3)

    # ---- where a step's time goes (instrumented passes AFTER the timed region; the timed steps above run unfenced): device
    # phases by HIP events on the launch stream, the exchange by a host timer between a barrier and the arrival of the
    # gathered candidates; per phase the mean over `reps` on this rank, then the max over ranks
    from evdr_amd import ops as _ops
    from evdr_amd.corpus import merge_candidates

    def all_ranks_ok(ok: bool) -> bool:
        """Control-plane agreement (gloo default group): True only if EVERY rank says ok -- taken before a section that holds
        collectives, so that a rank which failed locally makes all ranks skip it together instead of leaving its peers waiting
        in the collective for their timeout."""
        if not multi:
            return ok
        v = torch.tensor([1 if ok else 0], dtype=torch.int32)
        dist.all_reduce(v, op=dist.ReduceOp.MIN)
        return bool(v.item())

    def phase_breakdown(reps=3):
        acc = [0.0, 0.0, 0.0, 0.0]
        sbuf = None
        for _ in range(reps):
            err = None
            try:                                           # device phases: local to this rank
                if sbuf is None:
                    sbuf = torch.empty((args.queries, corpus.n_pages), dtype=torch.float32, device=dev)
                e = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
                e[0].record()
                corpus.score(Q, None, out=sbuf)
                e[1].record()
                if corpus.n_pages:
                    ls, li = _ops.topk(sbuf, args.topk, idx_base=corpus.idx_base)
                else:
                    ls, li = corpus.topk(Q, None, args.topk)
                e[2].record()
                torch.cuda.synchronize()
                acc[0] += e[0].elapsed_time(e[1])
                acc[1] += e[1].elapsed_time(e[2])
            except Exception as ex:                        # noqa: BLE001
                err = ex
            if not all_ranks_ok(err is None):
                raise err if err is not None else RuntimeError("a peer rank failed in the device phases")
            if multi:
                barrier()
                torch.cuda.synchronize()
                th = time.perf_counter()
                sc, ix = gather_candidates(ls, li, group)
                torch.cuda.synchronize()
                acc[2] += (time.perf_counter() - th) * 1e3
                e[3].record()
                merge_candidates(sc, ix, args.topk)
                e[4].record()
                torch.cuda.synchronize()
                acc[3] += e[3].elapsed_time(e[4])
        mine = torch.tensor([a / reps for a in acc], dtype=torch.float64)
        worst = mine.clone()
        if multi:
            dist.all_reduce(worst, op=dist.ReduceOp.MAX)          # control plane (host tensor)
        names = ("score_ms", "topk_ms", "exchange_ms", "merge_ms")
        keep = 4 if multi else 2
        return {"rank0": {n: float(v) for n, v in zip(names[:keep], mine[:keep])},
                "max_over_ranks": {n: float(v) for n, v in zip(names[:keep], worst[:keep])},
                "note": "separate instrumented passes after the timed region (score and top-k as two calls), mean of %d" % reps}
    # everything after the timed region is evidence around the headline number: a failure there is recorded in the line, it
    # must not cost the line itself (the first N > 1 run happens where nobody can re-run it)
    device_errors = []

    def guarded(what, fn):
        try:
            return fn()
        except Exception as e:                              # noqa: BLE001
            msg = f"{type(e).__name__}: {e}"
            print(f"bench.py: {what} failed: {msg}", file=sys.stderr)
            # a HIP / accelerator error is sticky: the line is still printed (the timed region was clean), but the process must
            # not report success -- it exits 3 after the line
            if any(k in msg for k in ("HIP", "hip", "CUDA", "accelerator", "libevdr status 4", "device-side")):
                device_errors.append(f"{what}: {msg[:200]}")
            return {"error": msg[:300]}
    phases = guarded("phase breakdown", phase_breakdown)

    # ---- roofline of the dominant kernel: HIP events around the MaxSim launch on its launch stream, INSIDE the timed steps
    # (one pair per step, recorded by PageCorpus.topk; every bracket lies inside its step, so their mean cannot exceed ms_per_step)
    out = torch.empty((args.queries, corpus.n_pages), dtype=torch.float32, device=dev)
    corpus.score(Q, None, out=out)                      # (the score block the CPU leg below is checked against)
    torch.cuda.synchronize()
    k_each = [a.elapsed_time(b) for a, b in step_events]
    k_ms = sum(k_each) / max(len(k_each), 1)
    flop_per_launch = args.queries * corpus.n_pages * FLOP_PER_PAIR
    achieved = flop_per_launch / (k_ms * 1e-3) / 1e12
    from evdr_amd import _lib as L
    kernel_symbol = L.load().evdr_last_fwd_kernel().decode()        # the instance the launches above really dispatched
    # HBM traffic: PMC counters cannot be read inside this process; `traffic` REPLAYS the figure of the committed
    # rocprofv3 --pmc passes (profiles/hbm_traffic.json, made by scratch/pmc.sh) -- only when they were taken for this
    # kernel instance and launch shape; the source file and its hash go into the line, null otherwise.
    traffic, traffic_src = None, None
    tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    if os.path.exists(tpath):
        try:
            import hashlib
            raw = open(tpath, "rb").read()
            rec = json.loads(raw)
            if (rec.get("queries") == args.queries and rec.get("pages_per_gpu") == corpus.n_pages
                    and rec.get("kernel") == kernel_symbol):
                traffic = rec.get("hbm_bytes_per_launch")
                traffic_src = {"kind": "replayed rocprofv3 --pmc counters (not measured in this run)",
                               "file": "profiles/hbm_traffic.json", "sha256": hashlib.sha256(raw).hexdigest()[:16],
                               "round": rec.get("round")}
        except Exception:
            traffic, traffic_src = None, None
    roofline = {"bound": "mfma", "achieved": achieved, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": achieved / MFMA_BF16_PEAK_TFLOPS, "traffic": traffic, "traffic_source": traffic_src,
                "kernel": kernel_symbol, "kernel_ms": k_ms,
                "kernel_ms_basis": f"mean of {len(k_each)} launches, one per TIMED step, HIP events on the launch stream inside the timed region",
                "algorithmic_flop_per_launch": flop_per_launch,
                "algorithmic_bytes_per_launch": corpus.n_pages * LP * D * 2}

    # ---- the other end of the roofline: few queries per pass stream the corpus from HBM (263 680 B per page, read once)
    def stream_regime(nq_small):
        Qs = Q[:nq_small].contiguous()
        o = torch.empty((nq_small, corpus.n_pages), dtype=torch.float32, device=dev)
        for _ in range(3):
            corpus.score(Qs, None, out=o)
        torch.cuda.synchronize()
        e = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(8)]
        for a, b in e:
            a.record()
            corpus.score(Qs, None, out=o)
            b.record()
        torch.cuda.synchronize()
        each = [a.elapsed_time(b) for a, b in e]
        ms = sum(each) / len(each)                       # the MEAN, like the headline's (what rocprofv3's average reports); the minimum rides beside it
        gbps = corpus.n_pages * LP * D * 2 / (ms * 1e-3) / 1e9
        return {"queries_per_pass": nq_small, "kernel": L.load().evdr_last_fwd_kernel().decode(), "kernel_ms": ms, "kernel_ms_min": min(each),
                "kernel_ms_basis": f"mean of {len(each)} back-to-back launches after 3 warm-ups, HIP events on the launch stream (after the timed region); "
                                   "kernel_ms_min = the fastest of them, never the basis of `frac`",
                "bound": "hbm", "achieved": gbps, "peak": 8000.0, "unit": "GB/s", "frac": gbps / 8000.0,
                "algorithmic_bytes_per_launch": corpus.n_pages * LP * D * 2,
                "mfma_tflops": nq_small * corpus.n_pages * FLOP_PER_PAIR / (ms * 1e-3) / 1e12}
    roofline["other_regimes"] = guarded("other regimes", lambda: [stream_regime(1), stream_regime(8)]) \
        if (args.queries >= 8 and not args.no_other_regimes) else []

    # ---- quality on the planted queries (rank 0; uses the merged top-k of the last step)
    ndcg5 = None
    cpu_base = None
    if rank == 0:
        docids = [f"doc{j}" for j in range(args.pages)]
        qrels = {str(i): {docids[int(t)]: 1} for i, t in enumerate(targets.tolist())}
        res = results_from_topk(ts.cpu().numpy(), ti.cpu().numpy(), [str(i) for i in range(args.queries)], docids)
        ndcg5 = CustomRetrievalEvaluator().compute_mteb_metrics(qrels, res)["NDCG"]["NDCG@5"]
        if world == 1 and not args.no_cpu_baseline:
            n_cpu = min(2048, corpus.n_pages)
            def cpu_leg():
                base, s_cpu = cpu_baseline_leg(shard_pages[:n_cpu], Q)
                base["max_abs_diff_vs_gpu"] = (out[:32, :n_cpu].cpu() - s_cpu).abs().max().item()   # same inputs: the oracle as checker
                return base
            cpu_base = guarded("cpu baseline", cpu_leg)
        train_step = eval_rec = configs2 = None
        if world == 1 and not args.no_extras:
            both = guarded("train_step / eval records", lambda: extras(dev, shard_pages, args))
            train_step, eval_rec = both if isinstance(both, tuple) else (both, both)
            # configs[2] on the first 6 847 pages of the resident corpus (a smaller --pages run generates them)
            configs2 = guarded("configs[2] record", lambda: configs2_record(
                dev, lambda n: shard_pages[:n] if shard_pages.shape[0] >= n else gen_pages(0, n, dev)))
        line = {
            "metric": "query-page pairs scored/sec", "value": value, "unit": "pairs/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "synthetic 100k-page late-interaction retrieval (BASELINE.json configs[3]): "
                                   "MaxSim of every query against every page + per-shard top-k + all-gather merge",
                       "pages": args.pages, "patches_per_page": LP, "dim": D, "queries_per_step": args.queries,
                       "query_tokens": LQ, "topk": args.topk, "parallelism": f"page-shard x{world}"},
            "queries_per_sec": args.queries / (ms_per_step * 1e-3), "ndcg_at_5": ndcg5,
            "timed_path": "ShardedRetriever.search -> PageCorpus.topk = evdr_maxsim_fwd_prepared + evdr_topk on the current stream "
                          "(+ all-gather + merge at N > 1): the calls every search issues, plus one HIP-event pair around the MaxSim launch",
            "dist": {"world_size": dist.get_world_size() if multi else 1, "ranks_seen": ranks_seen,
                     "backend": dist.get_backend(group) if multi else None, "control_backend": "gloo" if multi else None,
                     "backend_requested": args.backend if multi else None,
                     "backend_fallback_reason": fallback_reason, "launcher": launcher,
                     "exchange": "all_gather_into_tensor of (nq, 2k) int32 per rank" if multi else None,
                     "pages_per_rank": corpus.n_pages,
                     "rehearsal": "--dist-at-one: groups, barrier, clock all-reduce and the phase breakdown's exchange with ONE rank; "
                                  "the timed step had nothing to exchange" if (multi and world == 1) else None},
            "phases": phases,
            "roofline": roofline, "cpu_baseline": cpu_base,
            "train_step": train_step, "eval": eval_rec, "configs2": configs2,
            "device_errors_after_timed_region": device_errors or None,
        }
        print(json.dumps(line), flush=True)
    if multi:
        dist.barrier()                                 # control plane: leaves together whatever happened to the data group
        dist.destroy_process_group()
    if device_errors:
        raise SystemExit(3)


if __name__ == "__main__":
    main()
