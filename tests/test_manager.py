"""Model assembly and data splits (careless_amd/manager.py) against the reference's `DataManager` behaviour
(careless/io/manager.py:273-507; reference tests/io/test_data_manager.py, tests/test_cli.py).  CPU except where marked."""
import numpy as np
import pytest
import torch

from careless_amd.manager import DataManager, default_args, merge_half_datasets
from careless_amd.models.base import BaseModel
from tests import util


def _dm(laue=False, **flags):
    data = util.make_problem(N=240, R=30, laue=laue)[0]
    inputs = util.reference_inputs(data)
    dw = None
    return data, inputs, DataManager(inputs, data["centric"], data["multiplicity"], default_args(**flags), double_wilson=dw)


def test_default_flags_match_reference_cli():
    a = default_args()
    assert (a.mc_samples, a.mlp_layers, a.mlp_width, a.image_layers, a.use_image_scales, a.scale_bijector) == (1, 20, 10, 0, True, "exp")
    assert (a.learning_rate, a.beta_1, a.beta_2, a.iterations, a.epsilon, a.seed) == (1e-3, 0.9, 0.99, 10000, 1e-7, 1234)
    assert a.studentt_likelihood_dof is None and a.kl_weight is None and a.clipnorm is None
    with pytest.raises(ValueError):
        default_args(not_a_flag=1)


def test_build_model_default_wiring():
    """manager.py:432-506: q from the prior's moments, low = 1e-32 * ~centric, identity MLP of width d, hybrid image scaler"""
    from careless_amd.models.likelihoods.mono import NormalLikelihood, StudentTEv11Likelihood
    from careless_amd.models.scaling.image import HybridImageScaler
    data, inputs, dm = _dm()
    model = dm.build_model()
    q, prior = model.surrogate_posterior, model.prior
    assert np.allclose(q.loc.numpy(), prior.mean(), rtol=1e-6) and np.allclose(q.scale.numpy(), prior.stddev(), rtol=1e-6)
    assert np.array_equal(q.low.numpy(), (1e-32 * ~np.asarray(data["centric"])).astype(np.float32))
    assert isinstance(model.likelihood, NormalLikelihood) and isinstance(model.scaling_model, HybridImageScaler)
    mlp = model.scaling_model.mlp_scaler
    assert (mlp.n_layers, mlp.width, mlp.scale_bijector, mlp.scale_multiplier) == (20, 10, "exp", None)   # the CLI defaults (args/scaling.py)
    assert model.scaling_model.image_scaler.max_images == int(data["image_id"].max()) + 1
    assert (model.optimizer.learning_rate, model.optimizer.beta_2, model.mc_sample_size) == (1e-3, 0.99, 1)
    m2 = dm.build_model(default_args(studentt_likelihood_dof=4.0, refine_uncertainties=True, scale_bijector="softplus",
                                     use_image_scales=False, mlp_layers=3, mlp_width=16, mc_samples=4,
                                     structure_factor_init_scale=0.5, kl_weight=0.1, clipvalue=1.0))
    assert isinstance(m2.likelihood, StudentTEv11Likelihood) and m2.likelihood.dof == 4.0
    assert m2.scaling_model.scale_bijector == "softplus"
    assert np.isclose(m2.scaling_model.scale_multiplier, float(np.asarray(data["iobs"]).std()), rtol=1e-6)    # tfb.Shift(std(Iobs))
    assert np.allclose(m2.surrogate_posterior.scale.numpy(), 0.5 * prior.stddev(), rtol=1e-5)
    assert m2.mc_sample_size == 4 and m2.kl_weight == 0.1 and m2.optimizer.clipvalue == 1.0
    with pytest.raises(ValueError):
        dm.build_model(default_args(scale_bijector="tanh"))
    with pytest.raises(ValueError):                                  # reference tests/test_cli.py:92-110
        DataManager(inputs, data["centric"], data["multiplicity"], default_args(parents="None,0", dwr="0.,1.0"),
                    double_wilson=dict(reflids=np.zeros(30, int), root=np.ones(30, bool), asu_ids=np.zeros(30, int))).build_model()
    m3 = dm.build_model(default_args(image_layers=2, mlp_layers=3, mlp_width=8))          # manager.py:467-478
    from careless_amd.models.scaling.image import NeuralImageScaler
    assert isinstance(m3.scaling_model, NeuralImageScaler)
    assert (m3.scaling_model.n_image_layers, m3.scaling_model.max_images) == (2, int(data["image_id"].max()) + 1)
    m3.scaling_model.build(5)
    kern, bias = m3.scaling_model.image_weights[:2]                                        # identity per image, zero bias
    assert tuple(kern.shape) == (m3.scaling_model.max_images, 8, 8) and bool((kern == np.eye(8, dtype=np.float32)).all()) and not bias.any()


def test_wilson_prior_b_uses_resolution():
    data, inputs, _ = _dm()
    d = np.linspace(2.0, 10.0, 30)
    dm = DataManager(inputs, data["centric"], data["multiplicity"], default_args(wilson_prior_b=20.0), dHKL=d)
    prior = dm.build_model().prior
    assert np.allclose(prior.sigma, np.exp(-0.25 * 20.0 / d ** 2), rtol=1e-6)     # manager.py:43-46
    with pytest.raises(ValueError):
        DataManager(inputs, data["centric"], data["multiplicity"], default_args(wilson_prior_b=20.0)).build_model()


def test_mono_splits_partition_the_data():
    np.random.seed(0)
    data, inputs, dm = _dm()
    train, test = dm.split_data_by_refl(0.3)
    assert len(train[0]) + len(test[0]) == 240 and 30 < len(test[0]) < 120
    for a, b, c in zip(train, test, inputs):
        assert a.shape[1:] == c.shape[1:] and a.dtype == c.dtype
    tr, te = dm.split_data_by_image(0.5)
    assert not set(np.unique(tr[1])) & set(np.unique(te[1])) and len(tr[0]) + len(te[0]) == 240


def test_laue_split_repacks_harmonics():
    """manager.py:299-343: no harmonic group is split, harmonic ids are re-packed to 0..G'-1, intensity slots re-padded with 1"""
    np.random.seed(1)
    data, inputs, dm = _dm(laue=True)
    train, test = dm.split_data_by_refl(0.4)
    for part in (train, test):
        hid = BaseModel.get_harmonic_id(part).reshape(-1)
        n = len(hid)
        G = hid.max() + 1
        assert sorted(np.unique(hid)) == list(range(G)) and all(len(a) == n for a in part)
        iobs = BaseModel.get_intensities(part).reshape(-1)
        assert np.all(iobs[G:] == 1.0)
    assert len(train[0]) + len(test[0]) == 240
    bad = np.zeros(240, bool)
    hid = BaseModel.get_harmonic_id(inputs).reshape(-1)
    k = np.flatnonzero(np.bincount(hid) > 1)[0]
    bad[np.flatnonzero(hid == k)[0]] = True
    with pytest.raises(ValueError):
        dm.split_laue_data_by_mask(bad)


@pytest.mark.gpu
def test_build_model_trains_and_merges_half_datasets():
    np.random.seed(2)
    for laue in (False, True):
        data, inputs, dm = _dm(laue=laue, mlp_layers=3, mlp_width=16, mc_samples=2, type="poly" if laue else "mono")
        train, test = dm.split_data_by_refl(0.2)
        model = dm.build_model()
        hist = model.train_model(train, 6, progress=False, validation_data=test, validation_frequency=2)
        assert len(hist["loss"]) == 6 and all(np.isfinite(hist["loss"])) and len(hist["NLL_val"]) == 6
        res = dm.get_results(model.surrogate_posterior, inputs=train)
        assert np.all(np.isfinite(res["F"])) and np.all(res["SigF"] > 0)
        pred = dm.get_predictions(model, inputs=train)
        assert np.all(np.isfinite(pred["Ipred"])) and np.all(pred["SigIpred"] >= 0) and len(pred["Scale"]) == len(train[0])
        if laue:                                   # per harmonic slot (variational.py:70-76, 113-119): slots without rows hold 0
            G = int(BaseModel.get_harmonic_id(train).max()) + 1
            assert np.all(pred["Ipred"][:G] > 0) and not pred["Ipred"][G:].any() and not pred["SigScale"][G:].any()
        flat0 = model.scaling_model.mlp_scaler.flat.clone()
        halves = merge_half_datasets(dm, dm.parser, model.scaling_model, iterations=4, repeats=1)
        assert len(halves) == 2 and all(np.all(np.isfinite(r["F"])) for _, _, r in halves)
        assert torch.equal(model.scaling_model.mlp_scaler.flat.cpu(), flat0.cpu())          # the scaler stayed frozen
