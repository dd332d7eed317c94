"""Seeded input recipes shared by tests/golden/make_golden.py (which feeds them to the
reference) and by the tests (which feed the same inputs to the oracle and to the HIP path).

All randomness comes from torch's CPU generator with explicit seeds, so the container and the
GPU box (same image, same torch build) re-create identical inputs.
"""
import torch
import torch.nn.functional as F


def _unit(gen, *shape):
    return F.normalize(torch.randn(*shape, generator=gen), dim=-1)


def _bf16r(x):
    return x.bfloat16().float()


SMALL_CHUNK = {"small_ragged": 128, "lq1": 2, "chunk_tail": 3}


def small_case(name):
    """Small A1 cases with bf16-representable values (usable by the bf16 and the fp32 path)."""
    if name == "small_ragged":
        gen = torch.Generator().manual_seed(1234)
        Q = _bf16r(_unit(gen, 4, 8, 128))
        P = _bf16r(_unit(gen, 8, 40, 128))
        qm = torch.ones(4, 8, dtype=torch.bool)
        qm[1, 5:] = False
        qm[3, :2] = False
        pm = torch.ones(8, 40, dtype=torch.bool)
        pm[2] = False                       # all-masked page -> score exactly 0, grad 0
        pm[5, 25:] = False                  # suffix padding
        pm[6, 3:30:4] = False               # holes in the middle
        pm[7, :10] = False                  # masked prefix
        P[1, 7] = P[1, 3]                   # duplicate patch -> tie, first index must win
        P[1, 30] = P[1, 3]
        g = torch.randn(4, 8, generator=gen)
    elif name == "lq1":
        gen = torch.Generator().manual_seed(4321)
        Q = _bf16r(_unit(gen, 6, 1, 128))
        P = _bf16r(_unit(gen, 5, 33, 128))
        qm = torch.ones(6, 1, dtype=torch.bool)
        qm[4, 0] = False
        pm = torch.ones(5, 33, dtype=torch.bool)
        pm[0, 32] = False
        g = torch.randn(6, 5, generator=gen)
    elif name == "chunk_tail":
        gen = torch.Generator().manual_seed(99)
        Q = _bf16r(_unit(gen, 3, 5, 128))
        P = _bf16r(_unit(gen, 7, 20, 128))
        qm = torch.ones(3, 5, dtype=torch.bool)
        pm = torch.ones(7, 20, dtype=torch.bool)
        pm[6, 10:] = False
        g = torch.randn(3, 7, generator=gen)
    else:
        raise KeyError(name)
    return Q, P, qm, pm, g


def seeded_1030(bf16_inputs: bool):
    """SURVEY §8(c) pin (1): torch.manual_seed(0); (8,32,128) x (64,1030,128)."""
    gen = torch.Generator().manual_seed(0)
    Q = _unit(gen, 8, 32, 128)
    P = _unit(gen, 64, 1030, 128)
    if bf16_inputs:
        Q, P = _bf16r(Q), _bf16r(P)
    qm = torch.ones(8, 32, dtype=torch.bool)
    qm[:, 20:] = False
    pm = torch.ones(64, 1030, dtype=torch.bool)
    pm[3] = False
    pm[5, 500:] = False
    return Q, P, qm, pm


def l2_case():
    gen = torch.Generator().manual_seed(7)
    x = torch.randn(5, 9, 128, generator=gen) * 3.0
    x[0, 0] = 0.0            # zero row stays exactly zero
    x[2, 4] = 1e-20          # tiny row: eps is added to the norm, not clamped
    return x


def infonce_case():
    gen = torch.Generator().manual_seed(11)
    ss = torch.randn(32, 500, generator=gen) * 2.0 + 9.0
    st = torch.randn(32, 500, generator=gen) * 2.0 + 9.0
    return ss, st


def train_case(tag):
    """A7 capture: (Qb, qmb, P_teacher_raw, pmask_t, Pbar0_raw, pmask_s, hyper-params)."""
    if tag == "b4n8":
        B, N, Ls, Lt, seed = 4, 8, 40, 1030, 2024
    elif tag == "b32n128":
        B, N, Ls, Lt, seed = 32, 128, 206, 1030, 2025
    else:
        raise KeyError(tag)
    gen = torch.Generator().manual_seed(seed)
    Qb = _unit(gen, B, 32, 128)
    qmb = torch.ones(B, 32, dtype=torch.bool)
    qmb[:, 26:] = False
    qmb[0, 10:] = False
    Pt = torch.randn(N, Lt, 128, generator=gen)
    pmt = torch.ones(N, Lt, dtype=torch.bool)
    pmt[:, :4] = False                      # text-prefix tokens masked by the image mask
    pmt[1, 700:] = False
    # student init: block means of the teacher + noise (compressed pages), ragged lengths
    blk = Lt // Ls
    Pbar0 = Pt[:, : blk * Ls].reshape(N, Ls, blk, 128).mean(2) + 0.05 * torch.randn(N, Ls, 128, generator=gen)
    pms = torch.ones(N, Ls, dtype=torch.bool)
    pms[2, Ls - 7:] = False
    pms[N - 1, Ls // 2:] = False
    hp = {"temp": 0.1, "lr": 1e-3, "wd": 1e-2}
    return Qb, qmb, Pt, pmt, Pbar0, pms, hp


def single_vector_case():
    gen = torch.Generator().manual_seed(5)
    qs = [torch.randn(128, generator=gen) for _ in range(7)]
    ps = [torch.randn(128, generator=gen) for _ in range(11)]
    return qs, ps


def ragged_lists_case():
    """A2: ragged lists, bf16-representable, lengths chosen so every 4-batch has padding."""
    gen = torch.Generator().manual_seed(17)
    qlens = [5, 9, 12, 7, 3, 12]
    plens = [33, 1, 17, 40, 25, 64, 8, 31, 2]
    qs = [_bf16r(_unit(gen, l, 128)) for l in qlens]
    ps = [_bf16r(_unit(gen, l, 128)) for l in plens]
    return qs, ps


def losses_case():
    """(B, N) student / teacher score matrices + labels for the six secondary losses."""
    gen = torch.Generator().manual_seed(23)
    ss = torch.randn(6, 40, generator=gen) * 1.5 + 8.0
    st = torch.randn(6, 40, generator=gen) * 1.5 + 8.0
    labels = torch.randint(0, 40, (6,), generator=gen)
    return ss, st, labels


def npz_payload_case():
    """A small feature dump in the reference's object-array schema (ragged pages/queries, optional masks)."""
    import numpy as np
    rng = np.random.default_rng(31)
    dlens, qlens = [5, 12, 9, 1, 12, 3, 8], [3, 7, 7, 2, 5]
    def obj(items):
        a = np.empty(len(items), dtype=object)
        for i, v in enumerate(items):
            a[i] = v
        return a
    docs = obj([rng.normal(size=(l, 128)).astype(np.float32) for l in dlens])
    attn = obj([rng.random(l) > 0.15 for l in dlens])
    img = obj([np.concatenate([np.zeros(min(2, l), bool), np.ones(max(l - 2, 0), bool)]) for l in dlens])
    img[3] = np.ones((1, 1), dtype=np.int64)              # (Li,1) integer mask variant
    queries = obj([rng.normal(size=(l, 128)).astype(np.float32) for l in qlens])
    qattn = obj([np.ones(l, dtype=np.int64) for l in qlens])
    qattn[1] = np.array([1, 1, 1, 0, 0, 1, 1])
    docid = obj([f"doc_{i}" for i in range(len(dlens))])
    return docs, attn, img, queries, qattn, docid


def v3_case():
    """Inputs of the secondary-loss / v3-augmentation step fixtures (tests/golden/make_golden_v3.py): a small
    distillation problem -- (Qb, qmb, teacher raw, pmask_t, student init raw, pmask_s, hyper-parameters)."""
    B, N, Ls, Lt, seed = 6, 16, 24, 300, 2026
    gen = torch.Generator().manual_seed(seed)
    Qb = _unit(gen, B, 32, 128)
    qmb = torch.ones(B, 32, dtype=torch.bool)
    qmb[:, 27:] = False
    qmb[2, 9:] = False
    Pt = torch.randn(N, Lt, 128, generator=gen)
    pmt = torch.ones(N, Lt, dtype=torch.bool)
    pmt[:, :3] = False
    pmt[5, 200:] = False
    blk = Lt // Ls
    Pbar0 = Pt[:, : blk * Ls].reshape(N, Ls, blk, 128).mean(2) + 0.05 * torch.randn(N, Ls, 128, generator=gen)
    pms = torch.ones(N, Ls, dtype=torch.bool)
    pms[3, Ls - 5:] = False
    pms[N - 2, Ls // 2:] = False
    hp = {"k": 8, "temp": 2.0, "lr": 1e-3, "wd": 1e-2, "lambda_list": 1.0, "lambda_score": 0.5,
          "q_noise_std": 0.05, "noise_seed": 77,
          "lambda_mixed": 0.5, "mixup_alpha": 0.4, "mixup_seed": 78,
          "lambda_aux": 0.3, "aux_docs": 3}
    return Qb, qmb, Pt, pmt, Pbar0, pms, hp


def config1_case():
    """BASELINE.json configs[1] at its FULL size (docvqa_test_subsampled shape: 500 queries x 500 pages x 1030 patches, D = 128),
    synthetic as SURVEY §8(d) defines it: query i is planted on page t_i = (i * 7919) mod 500 (token n = normalise(P[t_i, pi_i(n)]
    + 0.5 eps)), ragged valid page lengths 700..1030, the last 0..12 tokens of a query masked.  Values are bf16-representable, so
    the ONE fixture made from them by the reference's fp32 scorer serves the bf16 kernel and the fp32 (fp16 hi/lo) kernel alike.
    -> Q (500,32,128), P (500,1030,128), qmask, pmask, targets (500,)"""
    gen = torch.Generator().manual_seed(20261004)
    n, nq, lp, lq = 500, 500, 1030, 32
    P = _bf16r(_unit(gen, n, lp, 128))
    lens = torch.randint(700, lp + 1, (n,), generator=gen)
    pm = torch.arange(lp)[None, :] < lens[:, None]
    targets = (torch.arange(nq) * 7919) % n
    rows = torch.stack([torch.randperm(700, generator=gen)[:lq] for _ in range(nq)])       # valid rows of every page
    eps = _unit(gen, nq, lq, 128)
    Q = _bf16r(F.normalize(P[targets[:, None], rows] + 0.5 * eps, dim=-1))
    qm = torch.ones(nq, lq, dtype=torch.bool)
    cut = lq - torch.randint(0, 13, (nq,), generator=gen)
    qm[torch.arange(lq)[None, :] >= cut[:, None]] = False
    return Q, P, qm, pm, targets


def trajectory_case(steps: int = 8):
    """A7 over several steps: the b4n8 capture's pages and masks with `steps` different seeded query batches (B = 4).
    -> ([(Qb, qmb), ...], P_teacher_raw, pmask_t, Pbar0_raw, pmask_s, hyper-params)"""
    _, _, Pt, pmt, Pbar0, pms, hp = train_case("b4n8")
    gen = torch.Generator().manual_seed(31337)
    batches = []
    for s in range(steps):
        Qb = _unit(gen, 4, 32, 128)
        qmb = torch.ones(4, 32, dtype=torch.bool)
        qmb[:, 28 - (s % 5):] = False
        batches.append((Qb, qmb))
    return batches, Pt, pmt, Pbar0, pms, hp


def width_case(d: int):
    """A1 + autograd at embedding width d (round 6: the reference takes any width, evaluator/retrieval.py:173; the kernels score
    d <= 128 on one 128-column block and 128 < d <= 256 on two): ragged masks, an all-masked page, pages crossing several
    tiles, queries with masked heads and tails."""
    gen = torch.Generator().manual_seed(6000 + d)
    Q = _unit(gen, 7, 20, d)
    P = _unit(gen, 11, 75, d)
    qm = torch.ones(7, 20, dtype=torch.bool)
    qm[2, 13:] = False
    qm[5, :3] = False
    pm = torch.ones(11, 75, dtype=torch.bool)
    pm[4] = False                           # all-masked page
    pm[6, 40:] = False                      # ragged tail
    pm[8, 5:60:7] = False                   # holes
    pm[9, :33] = False                      # masked prefix (a whole tile and a bit)
    # (no planted tie here: in fp32 two identical patch rows do not give bit-equal similarities in the reference's CPU einsum --
    # the blocked GEMM sums them in different orders --, so which of them its max picks depends on chunk_p; exact ties are pinned
    # on bf16-representable values by `small_case`)
    g = torch.randn(7, 11, generator=gen)
    return Q, P, qm, pm, g
