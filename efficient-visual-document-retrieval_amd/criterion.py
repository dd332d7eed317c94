"""Counterpart of the reference's criterion.py for the north-star loss.

`infonce_distillation_loss` keeps the reference signature (criterion.py:56-68) and runs loss + gradient in
one HIP kernel (evdr_infonce_distill_fwd_bwd): d loss/d score_s = (softmax(s/τ) - onehot(argmax t)) / (τ B),
which feeds the MaxSim backward directly."""
import torch

from . import ops


class _InfoNCEDistill(torch.autograd.Function):
    @staticmethod
    def forward(ctx, score_s, score_t, temperature):
        loss, grad = ops.infonce_distill(score_s, score_t, temperature, want_grad=ctx.needs_input_grad[0])
        if grad is not None:
            ctx.save_for_backward(grad)
            ctx.in_dtype = score_s.dtype
        return loss

    @staticmethod
    def backward(ctx, gout):
        (grad,) = ctx.saved_tensors
        return (grad * gout).to(ctx.in_dtype), None, None


def infonce_distillation_loss(score_s: torch.Tensor, score_t: torch.Tensor, temperature: float = 0.07) -> torch.Tensor:
    """CE(score_s / temperature, argmax_p score_t), mean over the batch; the teacher is not differentiated."""
    return _InfoNCEDistill.apply(score_s, score_t.detach(), float(temperature))
