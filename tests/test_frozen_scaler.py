"""A frozen scaling model (`--freeze-scales`; the half-dataset trainings of `--merge-half-datasets`: reference careless/careless.py:48-50,
102-128): the engine takes (loc, sigma) of every observation once and runs only the sampling / likelihood part per step
(`ElboEngine._data_term_frozen`, round 5).  Held here against the fused step of the same engine with the short cut switched off -- same
in-kernel noise, same loss, same gradients of everything that is trainable -- for every data kind and scaler family, against the oracle on
injected noise, and over a short Adam trajectory."""
import numpy as np
import pytest
import torch

from oracle import elbo_oracle as O
from tests import util

pytestmark = pytest.mark.gpu

CASES = {
    "cli_default_20x10_S3": dict(N=2000, R=60, d0=5, L=20, w=10, S=3, perturb=0.02),
    "mono_5x64_studentt_posenc_S2": dict(N=1500, R=50, d0=5, posenc=True, L=5, w=64, S=2, likelihood="studentt", dof=8.0),
    "laue_2x32_S2": dict(N=900, R=60, L=2, w=32, S=2, laue=True),
    "laue_20x10_ev11": dict(N=900, R=60, L=20, w=10, S=1, laue=True, ev11=True, perturb=0.02),
    "double_wilson_2x32": dict(N=800, R=60, d0=5, L=2, w=32, S=2, double_wilson=True),
    "image_layers2_20x10": dict(N=1200, R=50, d0=5, L=20, w=10, S=2, n_images=7, image_layers=2, perturb=0.03),
    "image_layers1_2x32": dict(N=700, R=40, d0=5, L=2, w=32, S=3, n_images=5, image_layers=1),
    "wide_2x96": dict(N=600, R=40, d0=5, L=2, w=96, S=2),
    "chained_12x32": dict(N=600, R=40, d0=5, L=12, w=32, S=2),
    "peeled_20x10_d37_ev11": dict(N=900, R=40, d0=37, L=20, w=10, S=2, ev11=True, perturb=0.02),
}


def _engine(kw, fast, shard=None, seed=31):
    from careless_amd.engine import ElboEngine
    data, cfg, params, x, _, _ = util.make_problem(**kw)
    model = util.build_model(data, cfg, params, kw["L"], kw["w"])
    model.scaling_model.trainable = False
    model.frozen_scaler_fast_path = fast
    eng = ElboEngine(model, util.reference_inputs(data), seed=seed, shard=shard)
    if shard is not None:
        eng.local_only = True
    return eng, model


@pytest.mark.parametrize("name", list(CASES))
def test_frozen_step_equals_the_fused_step(name):
    kw = CASES[name]
    fast, _ = _engine(kw, True)
    full, _ = _engine(kw, False)
    assert fast.scaler_frozen and fast._frozen_layout and not full._frozen_layout
    for eng in (fast, full):
        eng.forward_backward(4)
    torch.cuda.synchronize()
    tf, tu = fast.loss_terms(), full.loss_terms()
    assert abs(tf["nll"] - tu["nll"]) <= 2e-5 * abs(tu["nll"]) and abs(tf["kl"] - tu["kl"]) <= 1e-6 * max(abs(tu["kl"]), 1.0)
    lay, R = fast.layout, fast.R
    gq_f, gq_u = fast.grads[: 2 * R].cpu().numpy(), full.grads[: 2 * R].cpu().numpy()
    assert util.rel_err(gq_f, gq_u) < 2e-5                                  # d a, d b of every reflection
    if lay.n_ev11 > 0:                                                     # the Evans-2011 parameters stay trainable
        e_f, e_u = fast.grads[lay.off_ev11: lay.off_ev11 + 3].cpu().numpy(), full.grads[lay.off_ev11: lay.off_ev11 + 3].cpu().numpy()
        assert util.rel_err(e_f, e_u) < 5e-5
    # the scaler's own gradient is not computed (the reference does not take it either: trainable_variables only)
    # (the image scales' gradient falls out of the likelihood kernel; it is neither applied nor part of the norm: `frozen`)
    assert float(fast.grads[lay.off_mlp: lay.off_img].abs().max()) == 0.0


@pytest.mark.parametrize("name", ["cli_default_20x10_S3", "laue_2x32_S2", "image_layers2_20x10", "double_wilson_2x32"])
def test_frozen_step_matches_the_oracle_on_injected_noise(name):
    kw = CASES[name]
    data, cfg, params, x, u_f, eta = util.make_problem(**kw)
    out, grads = O.elbo_value_and_grads(params, x, cfg, torch.as_tensor(u_f, dtype=torch.float64), torch.as_tensor(eta, dtype=torch.float64))
    model = util.build_model(data, cfg, params, kw["L"], kw["w"])
    model.scaling_model.trainable = False
    ipred = model(util.reference_inputs(data), u_f=u_f, eta=eta)
    eng = model._engine
    torch.cuda.synchronize()
    assert eng.scaler_frozen and eng._frozen_layout
    t = eng.loss_terms()
    assert abs(t["loss"] - float(out["loss"])) <= 1e-4 * abs(float(out["loss"]))
    assert util.rel_err(ipred.cpu().numpy(), out["ipred"].numpy()) < 1e-4
    g = eng.grad_tensors()
    for k in (0, 1):                                                       # q's two tensors lead both lists
        assert util.rel_err(g[k].cpu().numpy(), grads[k].numpy()) < 2e-4


def test_frozen_rank_shards_sum_to_the_full_batch():
    from careless_amd.engine import make_shard
    kw = dict(N=1500, R=60, d0=5, L=20, w=10, S=2, perturb=0.02)
    full, _ = _engine(kw, True)
    full.forward_backward(3)
    torch.cuda.synchronize()
    g, nll = torch.zeros_like(full.grads), 0.0
    for r in range(2):
        eng, _ = _engine(kw, True, shard=make_shard(kw["N"], kw["R"], r, 2))
        eng.forward_backward(3)
        torch.cuda.synchronize()
        g += eng.grads
        nll += eng.loss_terms()["nll"]
    assert abs(nll - full.loss_terms()["nll"]) <= 1e-5 * abs(nll)
    assert util.rel_err(g.cpu().numpy(), full.grads.cpu().numpy()) < 2e-5


def test_frozen_reflection_owner_shards_sum_to_the_full_batch():
    """The reflection-owner split behind a frozen scaler: a rank's rows are not a contiguous range (the noise key is the global row number
    the sorted-rows kernel carries per row), its reflections are its own."""
    from careless_amd.engine import ElboEngine, make_shard
    kw = dict(N=3000, R=60, d0=5, L=20, w=10, S=3, perturb=0.02)
    full, _ = _engine(kw, True)
    full.forward_backward(3)
    torch.cuda.synchronize()
    data, cfg, params, x, _, _ = util.make_problem(**kw)
    g, nll = torch.zeros_like(full.grads), 0.0
    for r in range(3):
        model = util.build_model(data, cfg, params, kw["L"], kw["w"])
        model.scaling_model.trainable = False
        model.owner_shard = True
        eng = ElboEngine(model, util.reference_inputs(data), seed=31, shard=make_shard(kw["N"], kw["R"], r, 3))
        eng.local_only = True
        assert eng.owner and eng.scaler_frozen
        eng.forward_backward(3)
        torch.cuda.synchronize()
        assert getattr(eng.obs, "frozen_sorted", None) is not None
        g += eng.grads
        nll += eng.loss_terms()["nll"]
    assert abs(nll - full.loss_terms()["nll"]) <= 1e-5 * abs(nll)
    assert util.rel_err(g[: 2 * full.R].cpu().numpy(), full.grads[: 2 * full.R].cpu().numpy()) < 2e-5


def test_frozen_trajectory_and_unfreezing():
    """Ten Adam steps with the scaler frozen: the same q(F) as with the short cut off; the "Grad Norm" of the history is the norm over
    the trainable tensors alone in both (reference variational.py:201-205); unfreezing afterwards lays the observations out for the
    fused kernels again and trains the scaler."""
    kw = dict(N=1500, R=60, d0=5, L=20, w=10, S=2, n_images=6, image_layers=1, perturb=0.03)
    hist, q = {}, {}
    for fast in (True, False):
        data, cfg, params, x, _, _ = util.make_problem(**kw)
        model = util.build_model(data, cfg, params, kw["L"], kw["w"])
        model.scaling_model.trainable = False
        model.frozen_scaler_fast_path = fast
        model.seed = 5
        inputs = util.reference_inputs(data)
        w0 = model.scaling_model.mlp_scaler.flat.clone() if hasattr(model.scaling_model, "mlp_scaler") else None
        hist[fast] = model.train_model(inputs, 10, progress=False)
        q[fast] = model.surrogate_posterior.loc_raw.cpu().numpy().copy()
        if fast:
            model.scaling_model.trainable = True
            before = model._engine.params.clone()
            model.train_model(inputs, 2, progress=False)
            assert not model._engine._frozen_layout and model._engine.obs.row_map is not None       # packed by image again
            lay = model._engine.layout
            assert not torch.equal(before[lay.off_mlp:], model._engine.params[lay.off_mlp:])        # the scaler moved
    assert util.rel_err(q[True], q[False]) < 1e-4
    for k in ("loss", "NLL", "Grad Norm"):
        assert np.allclose(hist[True][k], hist[False][k], rtol=2e-4), k


# ---- round 6: rows sorted by reflection, `cl_frozen_rows` (no float atomics into dz_f) -----------------------------------------------------
SORTED_CASES = {
    "cli_default_S1": dict(N=5000, R=60, d0=5, L=20, w=10, S=1, perturb=0.02),
    "studentt_S8": dict(N=4000, R=50, d0=5, posenc=True, L=5, w=64, S=8, likelihood="studentt", dof=8.0),
    "S11_few_reflections": dict(N=3000, R=7, d0=5, L=2, w=32, S=11),                 # runs of ~430 rows: chains over several waves; two sample batches
    "many_reflections": dict(N=3000, R=2500, d0=5, L=2, w=32, S=3),                  # most runs a single row
    "ev11_no_image_scales": dict(N=2500, R=40, d0=5, L=20, w=10, S=2, ev11=True, use_image_scales=False, perturb=0.02),
    "double_wilson": dict(N=2000, R=60, d0=5, L=2, w=32, S=2, double_wilson=True),
}


@pytest.mark.parametrize("name", list(SORTED_CASES))
def test_sorted_rows_kernel_equals_the_slot_kernel_and_repeats_bit_for_bit(name):
    """The frozen step on `cl_frozen_rows` against round 5's `cl_slot_rows` on the same engine configuration (same in-kernel noise: the
    noise key is the row's global number in both), and twice on fresh engines: no float atomic touches dz_f, so the amplitude gradients
    are bit-identical from run to run."""
    from careless_amd.engine import ElboEngine
    kw = SORTED_CASES[name]
    runs = []
    for sorted_rows in (True, True, False):
        eng, _ = _engine(kw, True)
        eng.FROZEN_SORTED_ROWS = sorted_rows
        eng.forward_backward(4)
        torch.cuda.synchronize()
        runs.append((eng.grads.clone(), eng.dz_f.clone(), eng.loss_terms()))
        assert (getattr(eng.obs, "frozen_sorted", None) is not None) == sorted_rows
    (g1, z1, t1), (g2, z2, t2), (g0, z0, t0) = runs
    R, lay = eng.R, eng.layout
    assert torch.equal(z1, z2) and torch.equal(g1[: 2 * R], g2[: 2 * R])     # (the NLL is a sum of fp64 workgroup atomics, the three Evans-2011 terms
    #                                                                          of per-wave float atomics: equal to rounding)
    assert float((g1 - g2).abs().max()) <= 2e-6 * float(g1.abs().max())
    assert abs(t1["nll"] - t2["nll"]) <= 1e-12 * abs(t1["nll"])
    assert abs(t1["nll"] - t0["nll"]) <= 1e-6 * abs(t0["nll"])
    assert util.rel_err(z1.cpu().numpy(), z0.cpu().numpy()) < 2e-6
    assert util.rel_err(g1[: 2 * R].cpu().numpy(), g0[: 2 * R].cpu().numpy()) < 1e-5
    if lay.n_ev11 > 0:
        assert util.rel_err(g1[lay.off_ev11: lay.off_ev11 + 3].cpu().numpy(), g0[lay.off_ev11: lay.off_ev11 + 3].cpu().numpy()) < 2e-5


def test_sorted_rows_kernel_on_a_chunked_shard_adds_its_pieces():
    """A shard cut into several launches (`ObsChunks`) shares dz_f between its pieces: `accumulate` (atomics) instead of stores."""
    from careless_amd import engine as E
    kw = dict(N=6000, R=40, d0=5, L=20, w=10, S=2, perturb=0.02)
    one, _ = _engine(kw, True)
    one.forward_backward(2)
    old = E.launch_row_limit
    try:
        E.launch_row_limit = lambda d, S=0: 2560
        cut, _ = _engine(kw, True)
    finally:
        E.launch_row_limit = old
    assert isinstance(cut.obs, E.ObsChunks) and len(cut.obs.children) == 3
    cut.forward_backward(2)
    torch.cuda.synchronize()
    assert util.rel_err(cut.dz_f.cpu().numpy(), one.dz_f.cpu().numpy()) < 2e-6
    assert abs(cut.loss_terms()["nll"] - one.loss_terms()["nll"]) <= 1e-9 * abs(one.loss_terms()["nll"])


LAUE_CASES = {
    "laue_S1": dict(N=3000, R=60, L=2, w=32, S=1, laue=True),
    "laue_S3_studentt": dict(N=2500, R=40, L=20, w=10, S=3, laue=True, likelihood="studentt", dof=8.0, perturb=0.02),
    "laue_S9_ev11_few_reflections": dict(N=2000, R=9, L=2, w=32, S=9, laue=True, ev11=True),
    "laue_double_wilson_like_noimg": dict(N=1500, R=50, L=5, w=64, S=2, laue=True, use_image_scales=False),
}


@pytest.mark.parametrize("name", list(LAUE_CASES))
def test_frozen_harmonic_groups_two_call_form_equals_the_slot_launches_and_repeats(name):
    """Laue data behind a frozen scaler (round 6): `cl_frozen_rows` twice (group sums in packed order, per-reflection sums in reflection
    order) against round 5's `cl_laue_predict / _likelihood / _backward` on plain rows -- same in-kernel noise -- and twice on fresh engines:
    dz_f bit-identical (no float atomics)."""
    kw = LAUE_CASES[name]
    runs = []
    for packed in (True, True, False):
        from careless_amd.engine import ElboEngine
        old = ElboEngine.FROZEN_LAUE_PACKED
        ElboEngine.FROZEN_LAUE_PACKED = packed
        try:
            eng, _ = _engine(kw, True)
            eng.forward_backward(4)
            torch.cuda.synchronize()
        finally:
            ElboEngine.FROZEN_LAUE_PACKED = old
        assert bool(eng.obs.fused_laue) == packed and (getattr(eng.obs, "frozen_sorted", None) is not None) == packed
        runs.append((eng.grads.clone(), eng.dz_f.clone(), eng.loss_terms(), eng))
    (g1, z1, t1, e1), (g2, z2, t2, _), (g0, z0, t0, _) = runs
    R, lay = e1.R, e1.layout
    assert torch.equal(z1, z2) and torch.equal(g1[: 2 * R], g2[: 2 * R])
    assert abs(t1["nll"] - t0["nll"]) <= 2e-6 * abs(t0["nll"])
    assert util.rel_err(z1.cpu().numpy(), z0.cpu().numpy()) < 5e-6
    assert util.rel_err(g1[: 2 * R].cpu().numpy(), g0[: 2 * R].cpu().numpy()) < 1e-5
    if lay.n_ev11 > 0:
        assert util.rel_err(g1[lay.off_ev11: lay.off_ev11 + 3].cpu().numpy(), g0[lay.off_ev11: lay.off_ev11 + 3].cpu().numpy()) < 2e-5


def test_frozen_laue_rank_shards_sum_to_the_full_batch():
    """Shards of whole harmonic groups (the rows of a rank are not a contiguous range: the noise key is the global row number)."""
    from careless_amd.engine import make_shard
    kw = dict(N=2400, R=60, L=20, w=10, S=2, laue=True, perturb=0.02)
    full, _ = _engine(kw, True)
    full.forward_backward(3)
    torch.cuda.synchronize()
    assert full.obs.fused_laue and full.obs.frozen_sorted is not None
    g, nll = torch.zeros_like(full.grads), 0.0
    for r in range(3):
        eng, _ = _engine(kw, True, shard=make_shard(kw["N"], kw["R"], r, 3))
        eng.forward_backward(3)
        torch.cuda.synchronize()
        g += eng.grads
        nll += eng.loss_terms()["nll"]
    assert abs(nll - full.loss_terms()["nll"]) <= 1e-5 * abs(nll)
    assert util.rel_err(g.cpu().numpy(), full.grads.cpu().numpy()) < 2e-5
