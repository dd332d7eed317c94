"""The npz checkpoint format, pinned in BOTH directions by the reference's own code (fixtures made in the build container by
tests/golden/make_golden_npz_cross.py):
  * a best-checkpoint written by this repo's driver was read by the reference's load_init_payload + preprocess_docs -- here this
    repo's loaders must get the same tensors out of the same (re-written) file, i.e. the reference can take this repo's output as
    an `--init_root` file;
  * a checkpoint written by the reference's save_compressed_npz is read by this repo's loaders."""
import os

import numpy as np
import torch

import golden_recipes as R
import npz_cross_recipe as X
from evdr_amd.utils import preprocess_data as PD

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_checkpoint_written_here_reads_the_same_in_the_reference(tmp_path):
    z = np.load(os.path.join(GOLDEN, "npz_ours_read_by_reference.npz"))
    path = X.write_with_this_repo(tmp_path)
    back = PD.load_init_payload(str(path))
    P_raw, pmask, valid = PD.preprocess_docs(back["documents"], back["doc_attnmask"], back["doc_imgmask"], device="cpu")
    assert np.array_equal(P_raw.numpy(), z["P_raw"]) and np.array_equal(pmask.numpy(), z["pmask"]) and np.array_equal(valid, z["valid"])
    assert [str(d) for d in back["docid"]] == [str(d) for d in z["docid"]]
    meta = PD.load_npz(str(path))["meta"].item()
    assert sorted(meta.keys()) == [str(k) for k in z["meta_keys"]]
    assert int(meta["step"]) == int(z["meta_step"]) == 40 and float(meta["best"]["NDCG@5"]) == float(z["meta_best_ndcg5"]) == 0.625
    # what was stored: only the VALID rows of each page (the mask of the teacher's dump applied), as float32 object arrays
    docs, attn, img, _, _, _ = R.npz_payload_case()
    _, pm0, _ = PD.preprocess_docs(docs, attn, img, device="cpu")
    assert [int(d.shape[0]) for d in back["documents"]] == pm0.sum(dim=1).tolist()


def test_checkpoint_written_by_the_reference_reads_here():
    path = os.path.join(GOLDEN, "npz_written_by_reference.npz")
    docs, attn, img, _, _, docid = R.npz_payload_case()
    for loader in (PD.load_init_payload, PD.load_payload):
        back = loader(path)
        assert [str(d) for d in back["docid"]] == [str(d) for d in docid]
        assert all(np.array_equal(a, b) for a, b in zip(back["documents"], docs))
        P_raw, pmask, _ = PD.preprocess_docs(back["documents"], back["doc_attnmask"], back["doc_imgmask"], device="cpu")
        P0, pm0, _ = PD.preprocess_docs(docs, attn, img, device="cpu")
        assert torch.equal(P_raw, P0) and torch.equal(pmask, pm0)
    meta = PD.load_npz(path)["meta"].item()
    assert meta == {"dataset": "synthetic", "mf": 5, "step": 7, "best_type": "Recall@1"}
