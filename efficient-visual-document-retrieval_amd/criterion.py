"""Counterpart of the reference's criterion.py for the north-star loss.

`infonce_distillation_loss` keeps the reference signature (criterion.py:56-68) and runs loss + gradient in
one HIP kernel (evdr_infonce_distill_fwd_bwd): d loss/d score_s = (softmax(s/τ) - onehot(argmax t)) / (τ B),
which feeds the MaxSim backward directly."""
import torch

from . import ops


# Workspaces of the one-launch loss kernel (row losses + the ticket word the last workgroup resets), one per (device, stream, batch
# size): launches on one stream are ordered, so a workspace never serves two launches at once; the reference drives this from a
# single Python thread on one stream (SURVEY §8(b) "Threading / streams").  Inside a stream capture the stateless two-launch form is
# used instead (same bits: a captured graph must not share a ticket word with eager launches).
_LOSS_WS = {}


def _loss_workspace(score_s):
    if not score_s.is_cuda or torch.cuda.is_current_stream_capturing():
        return None
    dev = score_s.device
    key = (dev.index, ops.L.current_stream_handle(dev), int(score_s.shape[0]))
    ws = _LOSS_WS.get(key)
    if ws is None:
        if len(_LOSS_WS) > 16:
            _LOSS_WS.clear()
        ws = _LOSS_WS[key] = ops.infonce_workspace(score_s.shape[0], dev)
    return ws


class _InfoNCEDistill(torch.autograd.Function):
    @staticmethod
    def forward(ctx, score_s, score_t, temperature):
        loss, grad = ops.infonce_distill(score_s, score_t, temperature, want_grad=ctx.needs_input_grad[0],
                                         ws=_loss_workspace(score_s) if score_s.dim() == 2 and score_s.shape[0] > 0 else None)
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


# ----------------------------------------------------------------------------------------------------
# The other losses of the reference's criterion.py: small (B, N) elementwise / pairwise math that consumes the
# MaxSim kernel's score matrices unchanged.  Plain torch on the scores' device;
# the gradient w.r.t. score_s then flows into the HIP MaxSim backward.  Signatures and defaults as in the
# reference; the teacher scores are never differentiated.
# ----------------------------------------------------------------------------------------------------
import torch.nn.functional as _F


def infonce_supervised_loss(score_s: torch.Tensor, labels: torch.Tensor, temperature: float = 0.07) -> torch.Tensor:
    """CE(score_s / τ, labels) with explicit top-1 labels (criterion.py:43-53)."""
    return _F.cross_entropy(score_s / temperature, labels)


def score_preserving_loss(score_s: torch.Tensor, score_t: torch.Tensor) -> torch.Tensor:
    """Mean squared error between student and (detached) teacher scores (criterion.py:74-83)."""
    return ((score_s - score_t.detach()) ** 2).mean()


def pairwise_distillation_loss(score_s: torch.Tensor, score_t: torch.Tensor) -> torch.Tensor:
    """RankNet-style: for every ordered page pair (i, j) the student's margin s_i - s_j is a logit whose soft
    target is sigmoid(t_i - t_j); mean BCE over all B*N*N pairs (criterion.py:89-108)."""
    t = score_t.detach()
    margin_s = score_s[:, :, None] - score_s[:, None, :]
    target = torch.sigmoid(t[:, :, None] - t[:, None, :])
    return _F.binary_cross_entropy_with_logits(margin_s, target)


def listwise_distillation_loss(score_s: torch.Tensor, score_t: torch.Tensor, k: int = 10,
                               temperature: float = 1.0) -> torch.Tensor:
    """Partial cross-entropy over the teacher's top-k pages, scaled by τ² (criterion.py:114-142)."""
    logp_s = torch.log_softmax(score_s / temperature, dim=1)
    p_t = torch.softmax(score_t.detach() / temperature, dim=1)
    top_p, top_i = p_t.topk(k, dim=1)
    return -(top_p * logp_s.gather(1, top_i)).sum(dim=1).mean() * (temperature ** 2)


def lambda_loss(score_s: torch.Tensor, score_t: torch.Tensor, alpha: float = 1.0, eps: float = 1e-6) -> torch.Tensor:
    """LambdaLoss-style pairwise logistic loss in the teacher's ranking order, weighted by
    10 * |Δgain| * |Δdiscount| over the upper-triangular pairs (criterion.py:148-189)."""
    t = score_t.detach()
    n = score_s.shape[1]
    t_sorted, order = t.sort(dim=1, descending=True)
    s_sorted = score_s.gather(1, order)
    disc = 1.0 / torch.log2(torch.arange(1, n + 1, device=score_s.device, dtype=torch.float) + 1.0)
    d_disc = (disc[None, :, None] - disc[None, None, :]).abs()
    gain = torch.sigmoid(t_sorted)
    d_gain = (gain[:, :, None] - gain[:, None, :]).abs()
    weight = d_gain * d_disc * 10.0
    pair = -_F.logsigmoid(alpha * (s_sorted[:, :, None] - s_sorted[:, None, :]))
    upper = torch.triu(torch.ones(n, n, device=score_s.device), diagonal=1)
    return (weight * pair * upper).sum() / (upper.sum() + eps)


def ranknce_loss(score_s: torch.Tensor, score_t: torch.Tensor, temperature: float = 1.0,
                 lambda_weight: float = 1.0) -> torch.Tensor:
    """InfoNCE against the teacher's top-1 (position 0 after sorting by the teacher) plus a softplus penalty on
    adjacent student margins weighted by sigmoid of the teacher's adjacent margins (criterion.py:192-226)."""
    t = score_t.detach()
    t_sorted, order = t.sort(dim=1, descending=True)
    s_sorted = score_s.gather(1, order)
    labels = torch.zeros(score_s.shape[0], dtype=torch.long, device=score_s.device)
    nce = _F.cross_entropy(s_sorted / temperature, labels)
    ds = s_sorted[:, :-1] - s_sorted[:, 1:]
    dt = t_sorted[:, :-1] - t_sorted[:, 1:]
    rank = (torch.sigmoid(dt) * _F.softplus(-ds)).mean()
    return nce + lambda_weight * rank
