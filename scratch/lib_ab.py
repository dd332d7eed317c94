"""Interleaved A/B of several BUILDS of the library in one process (same box, same clocks): forward kernel on a prepared bf16
corpus.  usage: python scratch/lib_ab.py <nq> <pages> <rounds> <lib.so> [<lib.so> ...]   ("default" = the package's)
optional env AB_MASK=ragged|front|none  AB_DTYPE=bf16|f32"""
import ctypes as C, os, sys, torch
sys.path.insert(0, "."); import evdr_amd
from evdr_amd import _lib as L, ops
import bench as B
nq, pages, rounds = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
paths = [L.LIB_PATH if p == "default" else os.path.abspath(p) for p in sys.argv[4:]]
dev = torch.device("cuda:0")
P = B.gen_pages(0, pages, dev)
Q, _ = B.make_queries(max(nq, 32), pages, P, 0, pages, dev, 1); Q = Q[:nq].contiguous()
mask = os.environ.get("AB_MASK", "none")
pm = None
if mask != "none":
    g = torch.Generator(device=dev).manual_seed(1)
    pm = torch.ones((pages, B.LP), dtype=torch.bool, device=dev)
    if mask in ("ragged", "raggedfront"):
        lens = torch.randint(600, B.LP + 1, (pages,), generator=g, device=dev)
        pm &= torch.arange(B.LP, device=dev)[None, :] < lens[:, None]
    if mask in ("front", "raggedfront"):
        pm[:, :4] = False
fp32 = os.environ.get("AB_DTYPE", "bf16") == "f32"
if fp32:
    planes, pamax = ops.split_f32(P.float()); qpl, qamax = ops.split_f32(Q.float()); npl = 2
else:
    planes, pamax, qpl, qamax, npl = P[None], None, Q[None], None, 1
tilemask, pageflags = ops.pack_pmask(pm, pages, B.LP, dev)
out = torch.empty((nq, pages), dtype=torch.float32, device=dev)
libs = []
for p in paths:
    lib = C.CDLL(p)
    lib.evdr_maxsim_fwd_prepared.restype = C.c_int
    lib.evdr_maxsim_fwd_prepared.argtypes = L.SIGNATURES["evdr_maxsim_fwd_prepared"][1]
    libs.append(lib)
stream = torch.cuda.current_stream(dev).cuda_stream
def run(lib):
    rc = lib.evdr_maxsim_fwd_prepared(qpl.data_ptr(), planes.data_ptr(), None, tilemask.data_ptr(), pageflags.data_ptr(), out.data_ptr(),
                                      pages, None, nq, 32, pages, B.LP, npl, planes.stride(1), planes.stride(0),
                                      qamax.data_ptr() if fp32 else None, pamax.data_ptr() if fp32 else None, None, stream)
    assert rc == 0, rc
res = {p: [] for p in paths}; ref = None
valid = float(pm.sum().item()) if pm is not None else pages * B.LP
for rnd in range(rounds + 1):
    for p, lib in zip(paths, libs):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); run(lib); b.record(); torch.cuda.synchronize()
        if rnd == 0:
            if ref is None: ref = out.clone()
            else: print(f"  {os.path.basename(p)}: max |diff| vs first = {(out - ref).abs().max().item():.3e}", flush=True)
        else: res[p].append(a.elapsed_time(b))
for p in paths:
    ts = res[p]; ms = sum(ts) / len(ts)
    print(f"{os.path.basename(p):32s} nq={nq} pages={pages} mask={mask} {'f32' if fp32 else 'bf16'}: mean {ms:9.4f} ms  min {min(ts):9.4f}  "
          f"{nq * valid * 2 * 32 * 128 * (3 if fp32 else 1) / ms / 1e9:8.1f} TF(valid)  {pages * B.LP * 256 * npl / ms / 1e6:7.1f} GB/s", flush=True)
