"""BASELINE.json configs[0]: the reference's own test data `tests/data/pyp_off.mtz` (copied to tests/golden/pyp_off.mtz), Wilson prior,
Normal likelihood, 2x32 MLP scaler, mc-samples 1 -- the configuration of the reference's CPU plumbing tests
(reference tests/conftest.py:98-219, tests/models/merging/test_variational_mono.py:22-77).  Model wiring = `DataManager.build_model`
defaults (identity-initialised scaler, q from the prior's moments, image scales on)."""
import numpy as np
import pytest
import torch

from oracle import elbo_oracle as O
from tests import mtz_fixture, util


def _problem(S=1, seed=3):
    data = mtz_fixture.build_inputs()
    cfg = O.ElboConfig(mc_samples=S)
    params = O.init_params(data, cfg, 2, 32)                 # reference initialisation, no perturbation
    x = O.inputs_from_numpy(data)
    rng = np.random.default_rng(seed)
    return data, cfg, params, x, rng


def test_fixture_matches_reference_description():
    data = mtz_fixture.build_inputs()
    assert len(data["refl_id"]) == 166 and len(np.unique(data["refl_id"])) == 111 and data["n_images"] == 5   # conftest.py:169 ff.
    assert data["n_refl"] == 485 and int(data["centric"].sum()) == 83
    assert np.all(data["hkl_asu"][data["centric"], 2] == 0)                      # P6_3: the hk0 zone is centric
    assert sorted(np.unique(data["multiplicity"])) == [1.0, 6.0]                 # 00l reflections sit on the 6-fold axis
    assert not np.any((data["hkl_asu"][:, 0] == 0) & (data["hkl_asu"][:, 1] == 0) & (data["hkl_asu"][:, 2] % 2 == 1))   # 6_3 absences
    assert data["metadata"].shape == (166, 2) and abs(float(data["metadata"].mean())) < 1e-6


def test_oracle_trains_on_pyp():
    data, cfg, params, x, rng = _problem()
    st = O.AdamState.zeros_like(params.tensors())
    hist = [O.train_step(params, x, cfg, st, torch.as_tensor(rng.random((1, 485))), torch.as_tensor(rng.normal(size=(1, 166))))
            for _ in range(10)]
    assert all(np.isfinite(h["loss"]) and np.isfinite(h["Grad Norm"]) for h in hist)


@pytest.mark.gpu
def test_hip_engine_matches_oracle_on_pyp():
    data, cfg, params, x, rng = _problem()
    steps = 10
    noises = [(rng.random((1, 485)).astype(np.float32), rng.normal(size=(1, 166)).astype(np.float32)) for _ in range(steps)]
    model = util.build_model(data, cfg, params, 2, 32)
    inputs = util.reference_inputs(data)
    ipred = model(inputs, u_f=noises[0][0], eta=noises[0][1]).cpu().numpy()
    eng = model._engine
    out, grads = O.elbo_value_and_grads(params, x, cfg, torch.as_tensor(noises[0][0], dtype=torch.float64),
                                        torch.as_tensor(noises[0][1], dtype=torch.float64))
    t = eng.loss_terms()
    assert abs(t["loss"] - float(out["loss"])) <= 1e-4 * abs(float(out["loss"]))
    assert np.all(np.isfinite(ipred)) and util.rel_err(ipred, out["ipred"].numpy()) < 1e-4      # the reference's own assertion: finite
    errs = [util.rel_err(a.cpu().numpy(), b.numpy()) for a, b in zip(eng.grad_tensors(), grads)]
    assert max(errs) < 2e-4, errs
    model2 = util.build_model(data, cfg, params, 2, 32)
    hist = model2.train_model(inputs, steps, progress=False, noise=lambda i: noises[i])
    p = params.clone()
    st = O.AdamState.zeros_like(p.tensors())
    ref = [O.train_step(p, x, cfg, st, torch.as_tensor(u, dtype=torch.float64), torch.as_tensor(e, dtype=torch.float64)) for u, e in noises]
    assert np.allclose(hist["loss"], [r["loss"] for r in ref], rtol=1e-4)
    from careless_amd.results import get_results
    res = get_results(model2.surrogate_posterior, inputs)
    loc, scale = O.tn_loc_scale(p.q_loc_raw, p.q_scale_raw, cfg.epsilon)
    F = O.tn_mean(loc, scale, x.low, torch.tensor(cfg.high, dtype=torch.float64)).numpy()
    assert util.rel_err(res["F"], F) < 1e-4 and int(res["observed"].sum()) == 111                # merged |F| within 1e-4
