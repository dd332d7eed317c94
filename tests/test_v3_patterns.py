"""the listwise + score-preserving step and the v3 augmentations (query noise, page mixup, single-token
hard-token virtual queries), against fixtures produced by the reference's own train_one_step functions
(tests/golden/make_golden_v3.py).  CPU leg: the oracle through the same call sequences (pins the oracle and the
re-enactment); GPU leg: the drop-in modules -- loss values, the gradient that reaches Pbar through the HIP backward, and the
parameters after the AdamW step."""
import numpy as np
import pytest
import torch

import golden_recipes as R
import v3_patterns as V


def _check(out, z, keys, grad_atol, sc_atol=1e-4):
    for k in keys:
        np.testing.assert_allclose(out[k], float(z[k]), rtol=1e-5, err_msg=k)
    np.testing.assert_allclose(out["total_loss"], float(z["total_loss"]), rtol=1e-5)
    np.testing.assert_allclose(out["grad"].cpu().numpy(), z["grad"], atol=grad_atol, rtol=1e-5)
    # AdamW's first step is lr * g / (|g| + 1e-8): d(update)/dg = lr * 1e-8 / (|g| + 1e-8)^2, i.e. 1e-8-level summation noise
    # of the gradient moves the parameter by < 1e-7 where |g| > 1e-6 and by up to ~lr where |g| ~ 1e-8.  So: atol 1e-6
    # wherever the reference's own gradient exceeds 1e-6, bounded by 2 lr (and rare) elsewhere.
    d = np.abs(out["param_after"].cpu().numpy() - z["param_after"])
    big = np.abs(z["grad"]) > 1e-6
    assert d[big].max() <= 1e-6, d[big].max()
    assert d.max() < 2e-3 and (d > 1e-6).mean() < 1e-3, (d.max(), (d > 1e-6).mean())
    for k in ("sc_t", "sc_s", "sc_s_mix", "sc_t_v", "sc_s_v"):
        if k in out and k in z.files:
            np.testing.assert_allclose(out[k].cpu().numpy(), z[k], atol=sc_atol, rtol=0, err_msg=k)


def _run_all(be, golden, grad_atol):
    case = R.v3_case()
    _check(V.step_liscore(be, case), golden("v3_liscore"), ["loss_list", "loss_score"], grad_atol)
    z = golden("v3_noise")
    out = V.step_noise(be, case)
    np.testing.assert_allclose(out["Qb_used"].cpu().numpy(), z["Qb_used"], atol=1e-6)     # noise + re-normalisation of the queries
    _check(out, z, ["loss_list", "loss_score"], grad_atol)
    z = golden("v3_mixup")
    _check(V.step_mixup(be, case, float(z["lam"]), torch.from_numpy(z["perm"])), z, ["loss_list", "loss_score", "loss_score_mix"], grad_atol)
    z = golden("v3_hardtoken")
    out = V.step_hardtoken(be, case)
    assert out["q_virtual"].shape == z["q_virtual"].shape == (18, 1, 128)                 # 6 queries x 3 hard pages
    np.testing.assert_allclose(out["q_virtual"].cpu().numpy(), z["q_virtual"], atol=1e-6)  # same hard tokens chosen
    _check(out, z, ["loss_main", "loss_aux", "loss_list_aux", "loss_score_aux"], grad_atol)


def test_oracle_reproduces_the_reference_steps(golden):
    _run_all(V.oracle_backend(), golden, grad_atol=1e-7)


@pytest.mark.gpu
def test_hip_path_reproduces_the_reference_steps(golden):
    _run_all(V.hip_backend(), golden, grad_atol=1e-6)


@pytest.mark.gpu
def test_hip_path_with_the_score_row_cache_on_reproduces_the_reference_steps(golden, monkeypatch):
    """The same four script steps with `evaluator.retrieval.enable_score_cache()` (the one-line switch of INTEGRATION.md §1): every
    no-grad scoring of a frozen tensor with 2..32-token queries goes through the device-side cache (noise / mixup queries are new rows
    every call, the single-token virtual queries bypass it) -- the fixtures of the reference's own train_one_step functions hold with
    the switch on, and the cached forward really was the path taken."""
    import evdr_amd  # noqa: F401
    from evdr_amd import ops
    from evdr_amd.evaluator import retrieval as ER
    calls = []
    real = ops.maxsim_forward_cached
    monkeypatch.setattr(ops, "maxsim_forward_cached", lambda cache, *a, **k: (calls.append(int(a[0].shape[0])), real(cache, *a, **k))[1])
    ER.forget_prepared()
    ER.enable_score_cache(64 << 20)
    try:
        _run_all(V.hip_backend(), golden, grad_atol=1e-6)
    finally:
        ER.disable_score_cache()
        ER.forget_prepared()
    assert len(calls) >= 4 and all(n > 0 for n in calls), calls          # at least the teacher call of each of the four steps


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["infonce_supervised_loss", "score_preserving_loss", "pairwise_distillation_loss",
                                  "listwise_distillation_loss", "lambda_loss", "ranknce_loss"])
def test_each_secondary_loss_drives_the_hip_backward(name):
    """criterion.py:43-226 consumed like mainv2_iter_{super_infonce,score_preserve,lipairwise,listwise,lambda,ranknce}.py:
    loss(score_multi_vector_masked(Q, l2_normalize(Pbar * m)), teacher scores) -> backward: d loss / d Pbar through the HIP
    MaxSim backward and the fused normalise backward == autograd of the oracle through the same loss."""
    import evdr_amd  # noqa: F401
    from evdr_amd import criterion as C
    Qb, qmb, Pt, pmt, Pbar0, pms, hp = R.v3_case()
    labels = torch.tensor([3, 0, 15, 7, 7, 9])
    kw = {"infonce_supervised_loss": dict(temperature=0.07), "listwise_distillation_loss": dict(k=8, temperature=2.0),
          "ranknce_loss": dict(temperature=0.5, lambda_weight=0.7)}.get(name, {})
    res = {}
    for tag, be in (("oracle", V.oracle_backend()), ("hip", V.hip_backend())):
        d = be.device
        Ptn = be.l2((Pt * pmt.unsqueeze(-1)).to(d))
        param = torch.nn.Parameter((Pbar0 * pms.unsqueeze(-1)).to(d))
        sc_s = be.score(Qb.to(d), be.l2(param * pms.to(d).unsqueeze(-1)), qmb.to(d), pms.to(d), 64)
        with torch.no_grad():
            sc_t = be.score(Qb.to(d), Ptn, qmb.to(d), pmt.to(d), 64)
        loss = getattr(C, name)(sc_s, labels.to(d) if name == "infonce_supervised_loss" else sc_t, **kw)
        loss.backward()
        res[tag] = (float(loss.item()), param.grad.cpu())
    np.testing.assert_allclose(res["hip"][0], res["oracle"][0], rtol=1e-5)
    # scores of the two sides differ by fp32 summation noise (~2e-6); a loss with temperature tau turns that into a relative
    # gradient error of ~2e-6 / tau (3e-5 at tau = 0.07): atol scales with the gradient's magnitude, floor 1e-6
    want = res["oracle"][1].numpy()
    np.testing.assert_allclose(res["hip"][1].numpy(), want, atol=max(1e-6, 5e-5 * float(np.abs(want).max())), rtol=1e-5)


def _run_loss_steps(be, golden, grad_atol):
    """Each of the nine remaining single-loss scripts: loss value(s), Pbar.grad (strided sample + norm) and the parameters after
    AdamW against what the reference's own train_one_step produced."""
    z = golden("v3_loss_steps")
    case = R.v3_case()
    for tag, loss_fn in V.loss_steps(be).items():
        out = V.step_with_loss(be, case, loss_fn)
        main_key = f"{tag}__total_loss" if f"{tag}__total_loss" in z.files else f"{tag}__loss"
        np.testing.assert_allclose(out["total_loss"], float(z[main_key]), rtol=1e-5, err_msg=tag)
        for k, v in out.items():
            if f"{tag}__{k}" in z.files and k not in ("total_loss",):
                np.testing.assert_allclose(v, float(z[f"{tag}__{k}"]), rtol=1e-5, err_msg=f"{tag}:{k}")
        g = out["grad"].cpu().numpy()
        want = z[f"{tag}__grad_sample"]
        # losses with a temperature tau turn the ~2e-6 score noise of the GPU sums into ~2e-6 / tau relative gradient noise
        np.testing.assert_allclose(g[::2, ::2, ::2], want, atol=max(grad_atol, 5e-5 * float(np.abs(want).max())), rtol=1e-5, err_msg=tag)
        np.testing.assert_allclose(np.linalg.norm(g.astype(np.float64)), float(z[f"{tag}__grad_norm"]), rtol=1e-4, err_msg=tag)
        d = np.abs(out["param_after"].cpu().numpy()[::2, ::2, ::2] - z[f"{tag}__param_sample"])
        big = np.abs(want) > 1e-6 * max(1.0, float(np.abs(want).max()) / 1e-2)
        assert d[big].max() <= 2e-6 and d.max() < 2e-3, (tag, d[big].max(), d.max())


def test_oracle_reproduces_the_nine_loss_steps(golden):
    _run_loss_steps(V.oracle_backend(), golden, grad_atol=1e-7)


@pytest.mark.gpu
def test_hip_path_reproduces_the_nine_loss_steps(golden):
    _run_loss_steps(V.hip_backend(), golden, grad_atol=1e-6)
