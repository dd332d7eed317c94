"""Counterpart of the part of the reference's utils/preprocess_data.py that sits on the differentiable
path of the training step."""
import torch


def l2_normalize(x: torch.Tensor, eps: float = 1e-12) -> torch.Tensor:
    """x / (||x||_2 + eps) over the last dim -- eps is ADDED to the norm, zero rows stay exactly zero and get
    the subgradient 0 through the norm (utils/preprocess_data.py:8-9; applied to Pbar*pmask every step,
    mainv2_iter_distill_infonce.py:279)."""
    return x / (torch.linalg.vector_norm(x, ord=2, dim=-1, keepdim=True) + eps)
