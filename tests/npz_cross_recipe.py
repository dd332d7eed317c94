"""Recipe shared by tests/golden/make_golden_npz_cross.py and tests/test_npz_cross.py: a best-checkpoint written by this repo's
driver (CPU tensors; the writer is host code)."""
import argparse
from pathlib import Path

import numpy as np
import torch

import golden_recipes as R

import evdr_amd  # noqa: F401
from evdr_amd import driver
from evdr_amd.utils import preprocess_data as PD


def write_with_this_repo(out_dir: Path) -> Path:
    docs, attn, img, _, _, docid = R.npz_payload_case()
    P_raw, pmask, _ = PD.preprocess_docs(docs, attn, img, device="cpu")
    g = torch.Generator().manual_seed(77)
    Pbar = P_raw + 0.25 * torch.randn(P_raw.shape, generator=g)               # a "trained" parameter: masked rows hold junk too
    best = {"step": 40, "Recall@1": 0.5, "NDCG@5": 0.625}
    metrics = {"Recall": {"Recall@1": 0.5}, "NDCG": {"NDCG@5": 0.625}, "latency": 0.004}
    driver.save_best_npz(out_dir=Path(out_dir), fname="best_ndcg5.npz", dataset="synthetic", mf=5, step=40, best=best, metrics=metrics,
                         Pbar_param=Pbar, pmask_student=pmask, docid_tr=docid, doc_attn_in=attn, doc_img_in=img,
                         args=argparse.Namespace(temp=0.1, lr=1e-3))
    return Path(out_dir) / "best_ndcg5.npz"
