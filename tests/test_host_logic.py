"""Host-side logic of careless_amd that needs no GPU: the input-tuple contract, the flat parameter layout, sharding,
plugin classes, and that libcareless_hip.so loads and exports every symbol include/careless_hip.h declares.  CPU only."""
import os
import re

import numpy as np
import pytest
import torch

from careless_amd import _lib
from careless_amd.engine import make_layout, make_shard
from careless_amd.models.base import BaseModel
from careless_amd.models.likelihoods.mono import NormalLikelihood, StudentTLikelihood
from careless_amd.models.merging.surrogate_posteriors import TruncatedNormal
from careless_amd.models.merging.variational import VariationalMergingModel
from careless_amd.models.priors.wilson import WilsonPrior
from careless_amd.models.scaling.image import HybridImageScaler, ImageScaler
from careless_amd.models.scaling.nn import MLPScaler
from careless_amd.workloads import bytes_per_obs, flops_per_obs, make_workload, reference_inputs
from tests import util

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# --- input contract (reference tests/models/test_base_model.py:6-28) -----------------------------------------
def test_base_model_contract():
    assert BaseModel.input_index == {"refl_id": 0, "image_id": 1, "file_id": 2, "metadata": 3, "intensities": 4,
                                     "uncertainties": 5, "wavelength": 6, "harmonic_id": 7}
    data = util.make_problem(N=50, R=8)[0]
    mono = reference_inputs(data)
    assert not BaseModel.is_laue(mono)
    laue = mono + (np.ones((50, 1), np.float32), np.zeros((50, 1), np.int64))
    assert BaseModel.is_laue(laue)
    for name, idx in BaseModel.input_index.items():
        assert BaseModel.get_name_by_index(idx) == name
        assert BaseModel.get_index_by_name(name) == idx
        assert BaseModel.get_input_by_name(laue, name) is laue[idx]
    assert BaseModel.get_metadata(mono).shape == (50, 5) and BaseModel.get_metadata(mono).dtype == np.float32
    assert BaseModel.get_refl_id(mono).dtype == np.int64 and BaseModel.get_refl_id(mono).shape == (50, 1)
    with pytest.raises(ValueError):
        BaseModel.get_index_by_name("nope")
    with pytest.raises(ValueError):
        BaseModel.get_name_by_index(99)
    with pytest.raises(ValueError):
        BaseModel.get_harmonic_id(mono)
    batched = tuple(a[None] for a in mono)          # a leading batch axis of 1 is squeezed (base.py:79-80)
    assert BaseModel.get_metadata(batched).shape == (50, 5)


# --- flat layout / sharding ---------------------------------------------------------------------------------------
def test_flat_layout_matches_scaler_and_library():
    lay = make_layout(R=100, d=21, w=64, L=5, n_img=9)
    mlp = MLPScaler(5, 64)
    assert lay.P == mlp.param_count(21) == 21 * 64 + 64 + 4 * (64 * 64 + 64) + 2 * 64 + 2
    assert lay.n == 200 + lay.P + 9
    assert lay.seg_off[0] == 0 and lay.seg_off[-1] == lay.n and len(lay.seg_owner) == len(lay.seg_off) - 1
    assert lay.seg_owner[:2] == ["q", "q"] and set(lay.seg_owner[2:]) == {"scaler"}
    assert np.all(np.diff(lay.seg_off) > 0)
    lib = _lib.get_lib()
    assert int(lib.cl_mlp_param_count(21, 64, 5)) == lay.P
    assert [int(lib.cl_mlp_max_layers(w)) for w in (10, 15, 16, 17, 32, 33, 64, 65)] == [20, 20, 20, 10, 10, 5, 5, 0]      # (16: its own instance since round 5)
    assert [int(lib.cl_mlp_meta_rows(d)) for d in (1, 4, 5, 21, 64)] == [4, 4, 8, 24, 64]


def test_shards_partition_observations_and_reflections():
    for N, R, W in [(1000, 31, 1), (1001, 31, 2), (999, 7, 4), (12345, 400, 8)]:
        sh = [make_shard(N, R, r, W) for r in range(W)]
        assert sh[0].start == 0 and sh[-1].stop == N and sh[0].kl_begin == 0 and sh[-1].kl_end == R
        for a, b in zip(sh[:-1], sh[1:]):
            assert a.stop == b.start and a.kl_end == b.kl_begin
        assert max(s.stop - s.start for s in sh) - min(s.stop - s.start for s in sh) <= W
    with pytest.raises(ValueError):
        make_shard(10, 3, 2, 2)


# --- plugin classes -----------------------------------------------------------------------------------------------
def test_scaler_parameter_views_and_identity_init():
    s = MLPScaler(3, 8)
    s.build(5)
    ws = s.weights
    assert [tuple(w.shape) for w in ws] == [(5, 8), (8,), (8, 8), (8,), (8, 8), (8,), (8, 2), (2,)]
    assert torch.equal(ws[0], torch.eye(5, 8)) and torch.equal(ws[6], torch.eye(8, 2)) and float(ws[1].abs().sum()) == 0.0
    ws[2][1, 3] = 7.0                                # views: Keras (in,out) element lands at W^T[out][in] of the flat buffer
    off = s.layer_slices()[1][0]
    assert float(s.flat[off + 3 * 8 + 1]) == 7.0
    with pytest.raises(ValueError):
        MLPScaler(2, 4, scale_bijector="tanh")
    with pytest.raises(ValueError):
        s.build(6)


def test_image_scaler_pins_first_image():
    im = ImageScaler(4)
    assert im.scales.tolist() == [1.0, 1.0, 1.0, 1.0] and im._scales.numel() == 3
    im._scales[:] = torch.tensor([2.0, 3.0, 4.0])
    ids = np.array([[0], [1], [3], [3]])
    inputs = (ids, ids, ids, np.zeros((4, 2), np.float32), np.zeros((4, 1), np.float32), np.ones((4, 1), np.float32))
    assert im(inputs).tolist() == [1.0, 2.0, 4.0, 4.0]


def test_wilson_prior_and_truncated_normal_host_protocol():
    from scipy import stats
    c = np.array([True, False, False])
    eps = np.array([1.0, 2.0, 1.0])
    p = WilsonPrior(c, eps, 1.0)
    E = np.array([0.5, 1.0, 2.0])
    ref = np.where(c, stats.halfnorm.logpdf(E, scale=np.sqrt(eps)), stats.weibull_min.logpdf(E, 2.0, scale=np.sqrt(eps)))
    assert np.allclose(p.log_prob(E), ref, rtol=1e-6)
    q = TruncatedNormal.from_loc_and_scale(p.mean(), p.stddev(), low=(1e-32 * ~c).astype(np.float32))
    assert len(q.trainable_variables) == 2 and q.trainable_variables[0].shape == (3,)      # reference test_truncated_normal.py:15-17
    assert np.allclose(q.loc.numpy(), p.mean(), rtol=1e-6) and np.allclose(q.scale.numpy(), p.stddev(), rtol=1e-6)
    loc, scale = q.loc.numpy().astype(float), q.scale.numpy().astype(float)
    a = (q.low.numpy() - loc) / scale
    m4 = stats.truncnorm.moment(4, a, np.inf, loc, scale)
    assert np.allclose(q.moment_4(method="scipy"), m4, rtol=1e-5)                          # reference test_truncated_normal.py:29-42
    if not torch.cuda.is_available():           # mean / stddev / moment_4('tf') are `cl_tn_moments`: no CPU path (tests/test_gpu_parity.py has them)
        from careless_amd._lib import CarelessHipError
        for f in (q.mean, q.stddev, lambda: q.moment_4(method="tf")):
            with pytest.raises(CarelessHipError):
                f()
    with pytest.raises(ValueError):
        q.moment_4(method="nope")
    q.trainable = False
    assert q.trainable_variables == []


def test_likelihood_objects_match_scipy():
    from scipy import stats
    data = util.make_problem(N=40, R=8)[0]
    inputs = reference_inputs(data)
    x = np.asarray(data["iobs"]) + 3.0
    assert np.allclose(NormalLikelihood()(inputs).log_prob(x), stats.norm.logpdf(x, data["iobs"], data["sigiobs"]), rtol=1e-5)
    assert np.allclose(StudentTLikelihood(4.0)(inputs).log_prob(x), stats.t.logpdf(x, 4.0, data["iobs"], data["sigiobs"]), rtol=1e-5)


def test_workload_bookkeeping():
    assert flops_per_obs(5, 64, 5) == 100992 and flops_per_obs(21, 64, 5) == 107136
    assert bytes_per_obs(5, 1) == 44 and bytes_per_obs(21, 8) == 164
    model, inputs, data, spec = make_workload("mono_1M_normal_5x64_S1", N=2000)
    assert spec["d"] == 5 and spec["R"] == 62 and inputs[3].shape == (2000, 5) and inputs[0].dtype == np.int64
    assert isinstance(model.scaling_model, HybridImageScaler) and model.mc_sample_size == 1
    assert np.all(np.diff(inputs[1][:, 0]) >= 0)                                          # image ids sorted
    assert set(np.unique(inputs[0][:, 0])) == set(range(62))                              # every reflection observed


# --- the C-ABI library -----------------------------------------------------------------------------------------------
def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "careless_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(cl_[a-z0-9_]+)\s*\(", hdr))
    assert {"cl_tn_forward", "cl_tn_backward", "cl_elbo_mono_fwd_bwd", "cl_mlp_forward", "cl_adam_step"} <= declared
    lib = _lib.get_lib()
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/careless_hip.h but not exported"
        assert name in _lib.EXPORTS, f"{name} not bound in careless_amd/_lib.py"
    assert lib.cl_version().startswith(b"careless_hip")


def test_compute_entry_points_fail_loudly_without_gpu():
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    data, cfg, params, x, u_f, eta = util.make_problem(N=64, R=8, L=2, w=8, S=1)
    model = util.build_model(data, cfg, params, 2, 8)
    with pytest.raises(_lib.CarelessHipError):
        model.train_model(reference_inputs(data), 1, progress=False)
    with pytest.raises(_lib.CarelessHipError):
        model.surrogate_posterior.sample(2)
    with pytest.raises(_lib.CarelessHipError):
        model.scaling_model(reference_inputs(data))


def test_unsupported_plugins_are_rejected_not_emulated():
    class Odd:
        pass
    data, cfg, params, x, u_f, eta = util.make_problem(N=64, R=8, L=2, w=8, S=1)
    model = util.build_model(data, cfg, params, 2, 8)
    model.likelihood = Odd()
    with pytest.raises((NotImplementedError, _lib.CarelessHipError)):
        model.train_model(reference_inputs(data), 1, progress=False)


def test_double_wilson_prior_host_matches_oracle_and_validates_r():
    from careless_amd.models.priors.wilson import DoubleWilsonPrior
    from oracle import elbo_oracle as O
    data, cfg, params, x, u_f, eta = util.make_problem(N=200, R=40, S=2, double_wilson=True)
    out = O.elbo_forward(params, x, cfg, torch.as_tensor(u_f, dtype=torch.float64), torch.as_tensor(eta, dtype=torch.float64))
    prior = DoubleWilsonPrior(data["centric"], data["multiplicity"], data["parent_ids"], data["root"], data["asu_ids"], data["dw_r"])
    ref = O.double_wilson_log_prob(out["z_f"], x.centric, x.multiplicity, x.sigma, x.parent_ids, x.root, x.asu_ids, x.dw_r)
    assert np.allclose(prior.log_prob(out["z_f"].numpy()), ref.numpy(), rtol=1e-5, atol=1e-5)
    assert (data["parent_ids"] == -1).any()
    with pytest.raises(ValueError):                       # reference io/manager.py:415-419 (test_cli.py:92-110)
        DoubleWilsonPrior(data["centric"], data["multiplicity"], data["parent_ids"], data["root"], data["asu_ids"], [0.0, 1.0])


def test_laue_likelihood_convolve_known_answer():
    """reference tests/models/likelihoods/test_laue.py:11-36: convolve(iobs[hid] / count[hid]) reproduces iobs on the slots"""
    from careless_amd.models.likelihoods import laue
    from scipy import stats
    data = util.make_problem(N=120, R=16, laue=True)[0]
    inputs = util.reference_inputs(data)
    assert BaseModel.is_laue(inputs) and len(inputs) == 8
    hid = data["harmonic_id"]
    G = data["n_groups"]
    fake = (data["iobs"][hid] / np.bincount(hid)[hid]).astype(np.float32)
    lk = laue.NormalLikelihood()(inputs)
    conv = lk.convolve(fake)
    assert np.allclose(conv[:G], data["iobs"][:G], rtol=1e-5) and np.all(conv[G:] == 0)
    lp = lk.log_prob(fake)
    assert np.allclose(lp[:G], stats.norm.logpdf(data["iobs"][:G], data["iobs"][:G], data["sigiobs"][:G]), rtol=1e-5)
    assert np.allclose(lp[G:], stats.norm.logpdf(0.0, 1.0, 1.0))          # the padded-slot constant (formatter.py:637-640)
    assert np.allclose(lk.convolve(np.stack([fake] * 3)), conv[None, :])   # batched (reference :33-36)
    lt = laue.StudentTLikelihood(4.0)(inputs).log_prob(fake)
    assert np.allclose(lt[:G], stats.t.logpdf(data["iobs"][:G], 4.0, data["iobs"][:G], data["sigiobs"][:G]), rtol=1e-5)


def test_get_results_needs_the_hip_library_and_a_gpu():
    """F / SigF / <F^4> of the output step are `cl_tn_moments` (tests/test_output_step.py holds them against scipy on the GPU): on a
    machine without one the call fails loudly instead of computing the moments some other way."""
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: tests/test_output_step.py::test_get_results_matches_scipy_moments covers the call")
    from careless_amd._lib import CarelessHipError
    from careless_amd.results import get_results
    data, cfg, params, x, u_f, eta = util.make_problem(N=120, R=30, S=1)
    model = util.build_model(data, cfg, params, 2, 32)
    with pytest.raises(CarelessHipError):
        get_results(model.surrogate_posterior, util.reference_inputs(data))


def test_ev11_host_likelihood_matches_scipy():
    """reference likelihoods/mono.py:39-73: scale = Sdfac sqrt(sig^2 + SdB softplus(x) + Sdadd softplus(x)^2)"""
    from scipy import stats
    from careless_amd.models.likelihoods.mono import NormalEv11Likelihood, StudentTEv11Likelihood
    data = util.make_problem(N=50, R=8)[0]
    inputs = reference_inputs(data)
    x = np.asarray(data["iobs"], dtype=np.float64) + 2.0
    lk = NormalEv11Likelihood()
    assert abs(lk.Sdfac - 1.0) < 1e-6 and abs(lk.Sdadd - 1.0) < 1e-6 and abs(lk.SdB - 1.0) < 1e-6 and len(lk.trainable_variables) == 1
    sp = np.logaddexp(0, x)
    sc = np.sqrt(np.asarray(data["sigiobs"], dtype=np.float64) ** 2 + sp + sp * sp)
    assert np.allclose(lk(inputs).log_prob(x), stats.norm.logpdf(x, data["iobs"], sc), rtol=1e-5, atol=1e-5)
    lt = StudentTEv11Likelihood(4.0)
    assert np.allclose(lt(inputs).log_prob(x), stats.t.logpdf(x, 4.0, data["iobs"], sc), rtol=1e-5, atol=1e-5)


def test_pack_by_image_and_pack_laue_layouts():
    """Host-side packing of the observation axis for the per-image-layer and single-pass Laue kernels (pure numpy)."""
    from careless_amd.engine import GRANULE, TILE, pack_by_image, pack_laue
    rng = np.random.default_rng(0)
    img = rng.integers(0, 7, size=1000)
    pos, n_pad, tile_img, row_map = pack_by_image(img)
    assert n_pad % TILE == 0 and len(np.unique(pos)) == 1000 and np.all(tile_img[pos // TILE] == img)
    assert np.all(row_map[pos] == np.arange(1000)) and (row_map >= 0).sum() == 1000
    for by_image in (False, True):
        sizes = rng.choice([1, 2, 3, 5, 16], p=[.7, .15, .05, .05, .05], size=300)
        hid = np.repeat(np.arange(300), sizes)
        gimg = np.sort(rng.integers(0, 4, size=300))
        img = gimg[hid]
        perm = rng.permutation(len(hid))
        hid, img = hid[perm], img[perm]
        pos, n_pad, gmeta, tile_gmax, row_map, tile_img = pack_laue(hid, img, by_image)
        assert len(np.unique(pos)) == len(hid) and pos.max() < n_pad and n_pad % TILE == 0
        for g in range(300):                       # members consecutive, inside one 16-row granule, member index / size recorded
            p = np.sort(pos[hid == g])
            assert np.all(np.diff(p) == 1) and p[0] // GRANULE == p[-1] // GRANULE
            assert list(gmeta[p] & 0xff) == list(range(len(p))) and np.all(gmeta[p] >> 8 == len(p))
        assert np.all(row_map[pos] == np.arange(len(hid))) and (gmeta[row_map < 0] == 0).all()
        assert np.all(tile_gmax[pos // TILE] >= (gmeta[pos] >> 8)) and tile_gmax.max() == 16
        if by_image:
            assert np.all(tile_img[pos // TILE] == img)
        else:
            assert tile_img is None
    assert pack_laue(np.zeros(17, int), np.zeros(17, int), False) is None      # a group larger than a wave: two-pass path


def test_kernel_name_follows_the_librarys_routing():
    """`cl_mlp_kernel_name` restates `cl_launch_mlp`'s routing inside the library (bench.py labels its roofline row with it): the
    CLI-default scaler runs lane-per-observation up to 31 metadata columns and any number of MC samples, other depths / widths 11-15
    on the narrow kernel, everything else on the 16- / 32- / 64-wide fused instances.  Host function: no GPU needed."""
    import ctypes as C
    from careless_amd import _lib
    lib = _lib.get_lib()

    def name(mode=0, **kw):
        a = _lib.MlpArgs()
        for k, v in kw.items():
            setattr(a, k, v)
        buf = C.create_string_buffer(128)
        n = lib.cl_mlp_kernel_name(C.byref(a), mode, buf, 128)
        assert n == len(buf.value)
        return buf.value.decode()

    # (last argument: the instance with / without the optional inputs and outputs -- injected noise, ipred_out, Ev11)
    assert name(d=5, w=10, L=20, S=1) == "elbo_lane_kernel<10, 8, false, false>"
    assert name(d=5, w=10, L=20, S=1, eta=1) == "elbo_lane_kernel<10, 8, false, true>"
    assert name(d=12, w=7, L=20, S=3) == "elbo_lane_kernel<8, 15, false, false>"
    assert name(d=21, w=10, L=20, S=8) == "elbo_lane_kernel<10, 0, false, false>"   # positional encodings: rows of an LDS buffer
    assert name(d=31, w=4, L=20, S=40, row_map=1) == "elbo_lane_kernel<4, 0, true, true>"
    assert name(d=32, w=10, L=20, S=1).startswith("elbo_mlp_kernel<16, 32, 20, 0")
    assert name(d=5, w=13, L=12, S=8) == "elbo_narrow_kernel<2, 4, 8, false>"
    # other depths than the default at widths 7 .. 10 (round 6): the lane kernel compiled for that depth, widest instance
    assert name(d=5, w=10, L=12, S=1) == "elbo_lane_kernel<10, 15, false, false, false, 0, 12>"
    assert name(d=12, w=7, L=2, S=3, eta=1) == "elbo_lane_kernel<8, 15, false, true, false, 0, 2>"
    assert name(d=6, w=9, L=19, S=1, row_map=1) == "elbo_lane_kernel<10, 15, true, true, false, 0, 19>"
    assert name(d=5, w=6, L=12, S=1) == "elbo_lane_kernel<8, 15, false, false, false, 0, 12>"
    assert name(d=5, w=4, L=12, S=1).startswith("elbo_narrow_kernel<")                 # narrower than 5: the narrow kernel's two-step instance
    assert name(d=21, w=10, L=12, S=1).startswith("elbo_mlp_kernel<16, 32, 20, 0")     # (more than 15 columns without the engine's peeled first layer)
    assert name(d=5, w=10, L=1, S=1).startswith("elbo_narrow_kernel<")
    # widths 11 and 12 with the metadata in registers (round 6): the twelve-wide instances; on 16 .. 31 columns the next kernel down
    assert name(d=5, w=12, L=20, S=1) == "elbo_lane_kernel<12, 8, false, false>"
    assert name(d=12, w=11, L=14, S=2, row_map=1) == "elbo_lane_kernel<12, 15, true, true, false, 0, 14>"
    assert name(d=21, w=12, L=20, S=1).startswith("elbo_mlp_kernel<16, 32, 20, 0")
    # per-image layers (packed by image): the lane kernel's instances at the default depth and (round 6) at 2 .. 19 layers of width 5 .. 10
    imgl = dict(n_imgl=2, row_map=1, imgl=1, d_imgl=1, tile_img=1, n_images=7, use_img=0)
    assert name(d=5, w=10, L=20, S=2, **imgl) == "elbo_lane_kernel<10, 8, true, false, false, 2> (image layers)"
    assert name(d=10, w=10, L=20, S=2, dZ0_out=1, **imgl) == "elbo_lane_kernel<10, 15, true, false, true, 2> (image layers)"
    assert name(d=5, w=10, L=10, S=2, **imgl) == "elbo_lane_kernel<10, 15, true, false, false, 2, 10> (image layers)"
    assert name(d=7, w=7, L=3, S=1, ev11=1, **dict(imgl, n_imgl=1)) == "elbo_lane_kernel<10, 15, true, true, false, 1, 3> (image layers)"
    assert name(d=5, w=4, L=10, S=2, **imgl).startswith("elbo_mlp_kernel<16, 8, 24, 0, image layers")
    assert name(d=5, w=10, L=20, S=2, **dict(imgl, n_imgl=3)) == "elbo_lane_kernel<10, 15, true, false, false, 3> (image layers)"     # (three: the default depth only)
    assert name(d=5, w=10, L=12, S=2, **dict(imgl, n_imgl=3)) == "elbo_lane_kernel<10, 15, true, false, false, 3, 12> (image layers)"
    assert name(d=8, w=9, L=12, S=2, dZ0_out=1, **dict(imgl, n_imgl=3)) == "elbo_lane_kernel<10, 15, true, true, false, 3, 12> (image layers)"
    assert name(d=5, w=10, L=20, S=2, **dict(imgl, n_imgl=4)).startswith("elbo_mlp_kernel<16, 8, 24, 0, image layers")
    # width <= 15 on more than 32 columns: the training launch takes the 32-wide instance (the 16-wide one is withdrawn), the forward-only launch keeps it
    assert name(d=36, w=11, L=2, S=2, **imgl) == "elbo_mlp_kernel<32, 64, 5, 0, image layers, KS=4>"
    assert name(d=50, w=13, L=8, S=2, **dict(imgl, n_imgl=1)) == "elbo_mlp_kernel<32, 64, 10, 0, image layers, KS=4>"
    assert name(d=36, w=11, L=2, S=2, mode=1, **imgl) == "elbo_mlp_kernel<16, 64, 24, 1, image layers, KS=4>"
    # ... in deterministic mode (round 6): the lane instances only
    assert name(d=5, w=10, L=20, S=2, dzf_obs=1, **imgl) == "elbo_lane_kernel<10, 8, true, true, false, 2> (image layers) (deterministic stores)"
    assert name(d=5, w=10, L=7, S=2, dzf_obs=1, **imgl) == "elbo_lane_kernel<10, 15, true, true, false, 2, 7> (image layers) (deterministic stores)"
    assert name(d=5, w=32, L=2, S=2, dzf_obs=1, **imgl) == "(unsupported)"
    assert name(d=21, w=64, L=5, S=8) == "elbo_mlp_kernel<64, 32, 5, 0, KS=4>"
    assert name(d=21, w=64, L=5, S=8, mode=1) == "elbo_mlp_kernel<64, 32, 5, 1, KS=4>"
    assert name(d=5, w=10, L=20, S=1, act_out=1, mode=1).startswith("elbo_mlp_kernel<16, 8, 20, 1, chain")
    assert lib.cl_mlp_kernel_name(None, 0, C.create_string_buffer(8), 8) < 0


def test_wide_path_envelope_queries():
    """Host functions of the layer-by-layer path (no GPU needed): which shapes take the recomputed first layer, the head's backward pass
    fused into the top layer's kernels and the square-layer kernel's envelope, how many partials the fused dgrad + first-layer weight
    gradient writes, the row pitch of the activation buffers."""
    lib = _lib.get_lib()
    assert [lib.cl_wide_ld(w) for w in (1, 4, 70, 128, 129)] == [4, 4, 72, 128, 132] and lib.cl_wide_ld(0) == 0
    # first layer recomputed: at most 15 metadata columns, hidden width at most 128
    assert lib.cl_wide_pre_supported(5, 128) == 1 and lib.cl_wide_pre_supported(15, 65) == 1
    assert lib.cl_wide_pre_supported(16, 128) == 0 and lib.cl_wide_pre_supported(5, 129) == 0 and lib.cl_wide_pre_supported(0, 64) == 0
    # fused head backward: square layers of 5 .. 8 sixteen-column blocks on both sides (widths 65 .. 128, same block count)
    assert lib.cl_wide_head_bwd_supported(128, 128) == 1 and lib.cl_wide_head_bwd_supported(65, 80) == 1 and lib.cl_wide_head_bwd_supported(96, 90) == 1
    assert lib.cl_wide_head_bwd_supported(64, 64) == 0 and lib.cl_wide_head_bwd_supported(129, 129) == 0 and lib.cl_wide_head_bwd_supported(96, 128) == 0
    # one partial per workgroup of the streaming kernel: eight 16-row blocks per workgroup, two workgroups per CU at most
    assert lib.cl_wide_dgrad_wgrad0_parts(0) == 0 and lib.cl_wide_dgrad_wgrad0_parts(1) == 1 and lib.cl_wide_dgrad_wgrad0_parts(16 * 8 * 3 + 1) == 4
    big = lib.cl_wide_dgrad_wgrad0_parts(10_000_000)
    assert big % 2 == 0 and 2 <= big <= 4096 and lib.cl_wide_dgrad_wgrad0_parts(100_000_000) == big
    # weight-gradient splits: at least 512 observations each, at most 512 of them
    assert lib.cl_wide_wgrad_splits(1) == 1 and lib.cl_wide_wgrad_splits(513) == 2 and lib.cl_wide_wgrad_splits(10_000_000) == 512
    # the new entry points refuse missing buffers before touching a device
    assert lib.cl_slot_rows(None, None) == -1
    assert lib.cl_wide_dense_dgrad_head(None, 128, None, None, None, None, 100, 128, 128, None, 128, 0.01, None, 128, None, None) == -1
    assert lib.cl_wide_dense_wgrad_head(None, 128, None, None, None, 0.01, None, 128, 100, 128, 128, None, None, 1, None, None) == -1
    assert lib.cl_wide_dense_dgrad_pre_wgrad0(None, 128, None, 100, 128, 128, None, 8, 5, None, None, 0.01, None, None, None) == -1


def test_round6_entry_points_host_side():
    """`cl_frozen_rows` / `cl_chain_dx` (round 6): workspace queries and argument checks answer without a device."""
    import ctypes as C
    lib = _lib.get_lib()
    # two edge records per 64-row wave, S floats each; at most CL_LAUE_LIK_MAX_BLOCKS workgroups of 256 rows
    assert lib.cl_frozen_edge_floats(0, 3) == 0 and lib.cl_frozen_edge_floats(1, 3) == 6 and lib.cl_frozen_edge_floats(64, 1) == 2
    assert lib.cl_frozen_edge_floats(65, 8) == 32 and lib.cl_frozen_edge_floats(10_000_000, 8) == 2 * 156250 * 8
    assert lib.cl_frozen_grid(1) == 1 and lib.cl_frozen_grid(257) == 2 and lib.cl_frozen_grid(10_000_000) == _lib.CL_LAUE_LIK_MAX_BLOCKS
    assert int(lib.cl_frozen_args_size()) == C.sizeof(_lib.FrozenArgs)
    assert lib.cl_frozen_rows(None, None) == -1
    fa = _lib.FrozenArgs()
    fa.n, fa.S = 100, 2
    assert lib.cl_frozen_rows(C.byref(fa), None) == -1                    # no buffers
    fa.gmeta = 1                                                          # (harmonic groups, first call: still no buffers)
    assert lib.cl_frozen_rows(C.byref(fa), None) == -1
    assert lib.cl_chain_dx(None, None, 10, 128, 10, 10, None, None, None) == -1
    assert lib.cl_chain_dx(1, 1, 10, 128, 16, 10, 1, None, None) == -2    # widths beyond 15
    assert lib.cl_chain_dx(1, 1, 200, 128, 10, 10, 1, None, None) == -1   # n_pad < n_obs
