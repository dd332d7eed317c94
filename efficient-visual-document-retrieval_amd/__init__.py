"""MI355X-native late-interaction (MaxSim) scorer: the one hot path of
kimjy-st/Efficient-Visual-Document-Retrieval, behind the reference's own Python signatures.

Import as `evdr_amd` (the repo-root `evdr_amd.py` registers this directory, whose on-disk name is not a
Python identifier, under that name).  Layout:
  csrc/                 hand-written gfx950 HIP kernels + the C ABI (include/evdr.h) -> libevdr.so
  _lib.py, ops.py       ctypes binding, torch-tensor wrappers (no fallback: GPU or exception)
  evaluator/retrieval.py  mirror of the reference's evaluator/retrieval.py
  criterion.py          infonce_distillation_loss (fused loss + gradient kernel)
  utils/preprocess_data.py  l2_normalize (on the autograd path of the training step)
  corpus.py             resident page corpus, per-shard top-k, RCCL all-gather merge
"""
__version__ = "0.1.0"
