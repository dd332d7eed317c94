#!/usr/bin/env python3
"""Cross-pin the npz checkpoint format in both directions with the reference's own code:
  (i)  a `best_ndcg5.npz` written by THIS repo's driver.save_best_npz is read by the reference's load_init_payload + preprocess_docs
       (the reference accepts such files as `--init_root` inputs): what the reference got out of it is the fixture
       `npz_ours_read_by_reference.npz`;
  (ii) a `best_recall.npz` written by the REFERENCE's utils.save_compressed_npz (from recipe arrays) is the fixture
       `npz_written_by_reference.npz`; the test reads it with this repo's loaders.
Runs only in the build container (needs /root/reference).  Both fixtures are data (arrays), no source."""
import argparse
import os
import sys
import tempfile
from pathlib import Path

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from make_golden import import_reference  # noqa: E402
import golden_recipes as R  # noqa: E402
import npz_cross_recipe as X  # noqa: E402


def main():
    _, _, ref_prep = import_reference()
    import utils.utils as ref_utils
    # ---- (i) ours -> reference
    with tempfile.TemporaryDirectory() as tmp:
        path = X.write_with_this_repo(Path(tmp))
        back = ref_prep.load_init_payload(str(path))
        P_raw, pmask, valid = ref_prep.preprocess_docs(back["documents"], back["doc_attnmask"], back["doc_imgmask"], device="cpu")
        meta = ref_prep.load_npz(str(path))["meta"].item()
        np.savez_compressed(os.path.join(HERE, "npz_ours_read_by_reference.npz"), P_raw=P_raw.numpy(), pmask=pmask.numpy(), valid=valid,
                            docid=np.array([str(d) for d in back["docid"]]), meta_keys=np.array(sorted(meta.keys())),
                            meta_step=np.array(int(meta["step"])), meta_best_ndcg5=np.array(float(meta["best"]["NDCG@5"])))
    # ---- (ii) reference -> ours
    docs, attn, img, _, _, docid = R.npz_payload_case()
    ref_utils.save_compressed_npz(Path(HERE) / "npz_written_by_reference.npz", docid, docs, attn, img,
                                  meta={"dataset": "synthetic", "mf": 5, "step": 7, "best_type": "Recall@1"})
    print("ok:", os.path.getsize(os.path.join(HERE, "npz_written_by_reference.npz")), os.path.getsize(os.path.join(HERE, "npz_ours_read_by_reference.npz")))


if __name__ == "__main__":
    main()
