"""§8(e) training partitioning: 2 ranks (gloo exchange, both computing on the one GPU of the box) run the page-sharded
step; the concatenated shard parameters and the loss must equal the single-process step on the whole page set."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _inputs():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import golden_recipes as R
    return R.train_case("b4n8")


def _worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import evdr_amd  # noqa: F401
        from evdr_amd import driver
        from evdr_amd.corpus import shard_range
        from evdr_amd.utils.preprocess_data import l2_normalize
        dev = torch.device("cuda:0")
        Qb, qmb, Pt, pmt, Pbar0, pms, hp = _inputs()
        n = Pt.shape[0]
        sizes = [shard_range(n, r, world)[1] - shard_range(n, r, world)[0] for r in range(world)]
        lo, hi = shard_range(n, rank, world)
        Ptn = l2_normalize(Pt * pmt.unsqueeze(-1))
        teacher = driver.TeacherScorer(Ptn[lo:hi].to(dev), pmt[lo:hi].to(dev))
        param = torch.nn.Parameter((Pbar0 * pms.unsqueeze(-1))[lo:hi].to(dev))
        opt = torch.optim.AdamW([param], lr=hp["lr"], weight_decay=hp["wd"])
        losses = [driver.sharded_train_one_step(Qb, qmb, teacher, param, pms[lo:hi].to(dev), opt, hp["temp"], sizes)
                  for _ in range(2)]
        # the same two steps on the fused kernels (no autograd graph): FusedStudent over this rank's pages
        student = driver.FusedStudent(Pbar0[lo:hi].to(dev), pms[lo:hi].to(dev), lr=hp["lr"], weight_decay=hp["wd"])
        flosses = [driver.sharded_fused_train_one_step(Qb, qmb, teacher, student, hp["temp"], sizes) for _ in range(2)]
        rows = driver.gather_rows(student.x, sizes)                   # checkpoint path of the sharded driver: all pages on every rank
        ret[rank] = (losses, param.detach().cpu().numpy(), flosses, student.x.cpu().numpy(), rows.cpu().numpy())
    finally:
        dist.destroy_process_group()


def test_sharded_step_equals_single_device():
    import evdr_amd  # noqa: F401
    from evdr_amd import driver
    from evdr_amd.utils.preprocess_data import l2_normalize
    world = 2
    port = 29700 + (os.getpid() % 200)
    ctx = mp.get_context("spawn")
    with ctx.Manager() as m:
        ret = m.dict()
        mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
        got = dict(ret)
    dev = torch.device("cuda:0")
    Qb, qmb, Pt, pmt, Pbar0, pms, hp = _inputs()
    Ptn = l2_normalize(Pt * pmt.unsqueeze(-1))
    teacher = driver.TeacherScorer(Ptn.to(dev), pmt.to(dev))
    param = torch.nn.Parameter((Pbar0 * pms.unsqueeze(-1)).to(dev))
    opt = torch.optim.AdamW([param], lr=hp["lr"], weight_decay=hp["wd"])
    ref_losses = [driver.train_one_step(Qb, qmb, teacher, pmt.to(dev), param, pms.to(dev), opt, temp=hp["temp"])
                  for _ in range(2)]
    for r in range(world):
        np.testing.assert_allclose(got[r][0], ref_losses, rtol=1e-6)           # same loss on every rank, both steps
    merged = np.concatenate([got[r][1] for r in range(world)], axis=0)
    np.testing.assert_allclose(merged, param.detach().cpu().numpy(), atol=1e-6)
    for r in range(world):
        np.testing.assert_allclose(got[r][2], ref_losses, rtol=2e-6)           # fused sharded step: same losses ...
    fused = np.concatenate([got[r][3] for r in range(world)], axis=0)
    for r in range(world):
        assert np.array_equal(got[r][4], fused)                            # gather_rows: bit-exact reassembly on every rank
    np.testing.assert_allclose(fused, param.detach().cpu().numpy(), atol=2e-6)  # ... and the same parameters
