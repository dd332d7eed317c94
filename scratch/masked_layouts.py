"""Headline-shape throughput under different page-mask layouts (1024 queries x 20000 pages x 1030 patches, bf16)."""
import sys, torch
sys.path.insert(0, "."); import evdr_amd, bench as B
from evdr_amd.corpus import PageCorpus
dev = torch.device("cuda:0"); pages = 20000
P = B.gen_pages(0, pages, dev)
Q, _ = B.make_queries(1024, pages, P, 0, pages, dev, 1)
def layouts():
    pm = torch.ones(pages, 1030, dtype=torch.bool, device=dev); yield "all valid", None
    m = pm.clone(); m[:, 1024:] = False; yield "valid prefix of 1024 (PaliGemma: text tokens after the image)", m
    m = pm.clone(); m[:, :5] = False; m[:, -1] = False; yield "5 masked in front, 1 at the end (image in the middle)", m
    g = torch.Generator(device=dev).manual_seed(1)
    lens = torch.randint(600, 1031, (pages,), generator=g, device=dev)
    m = torch.arange(1030, device=dev)[None, :] < lens[:, None]; yield "ragged valid prefix 600..1030", m
    m = m.clone(); m[:, :4] = False; yield "ragged + 4 masked in front", m
out = torch.empty((1024, pages), dtype=torch.float32, device=dev)
for name, m in layouts():
    corpus = PageCorpus.from_tensor(P, m)
    corpus.score(Q, None, out=out); torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); corpus.score(Q, None, out=out); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    valid = 1030 if m is None else float(m.sum()) / pages
    ms = min(ts)
    print(f"{name:62s} {ms:8.2f} ms   {1024*pages*2*32*valid*128/ms/1e9:7.1f} TFLOP/s on valid patches", flush=True)
