"""Pins the CPU oracle (oracle/elbo_oracle.py) BEFORE it is trusted as the checker of the HIP kernels:
  * the reference's own closed-form known-answer tests, restated (reference tests cited per test);
  * scipy.stats closed forms for every density on the path;
  * finite differences for the truncated-normal pathwise gradient and the full ELBO gradient.
CPU only."""
import math

import numpy as np
import pytest
import torch
from scipy import special, stats

from oracle import elbo_oracle as O

T = lambda a: torch.as_tensor(np.asarray(a, dtype=np.float64))


def test_wilson_centric_closed_form():
    """reference tests/models/priors/test_wilson.py:13-20: Centric pdf = sqrt(2/pi) exp(-E^2/2)"""
    E = np.linspace(0.1, 3.0, 100)
    p = (2.0 / np.pi) ** 0.5 * np.exp(-0.5 * E ** 2)
    lp = O.wilson_log_prob(T(E), torch.ones(100, dtype=torch.bool), T(np.ones(100)), T(1.0)).numpy()
    assert np.allclose(np.log(p), lp, rtol=1e-12, atol=1e-12)
    assert np.allclose(stats.halfnorm.logpdf(E), lp)


def test_wilson_acentric_closed_form():
    """reference tests/models/priors/test_wilson.py:22-29: Acentric pdf = 2 E exp(-E^2)"""
    E = np.linspace(0.1, 3.0, 100)
    p = 2.0 * E * np.exp(-E ** 2)
    lp = O.wilson_log_prob(T(E), torch.zeros(100, dtype=torch.bool), T(np.ones(100)), T(1.0)).numpy()
    assert np.allclose(np.log(p), lp, rtol=1e-12, atol=1e-12)
    assert np.allclose(stats.weibull_min.logpdf(E, 2.0), lp)


def test_wilson_multiplicity_sigma_and_moments():
    rng = np.random.default_rng(0)
    eps = rng.integers(1, 6, 50).astype(float)
    sig = rng.uniform(0.2, 2.0, 50)
    z = rng.uniform(0.05, 3.0, 50)
    c = rng.random(50) < 0.5
    lp = O.wilson_log_prob(T(z), torch.as_tensor(c), T(eps), T(sig)).numpy()
    ref = np.where(c, stats.halfnorm.logpdf(z, scale=np.sqrt(eps * sig)),
                   stats.weibull_min.logpdf(z, 2.0, scale=np.sqrt(eps * sig)))
    assert np.allclose(lp, ref)
    m, s = O.wilson_mean(c, eps, 1.0), O.wilson_stddev(c, eps, 1.0)
    assert np.allclose(m, np.where(c, stats.halfnorm.mean(scale=np.sqrt(eps)), stats.weibull_min.mean(2.0, scale=np.sqrt(eps))))
    assert np.allclose(s, np.where(c, stats.halfnorm.std(scale=np.sqrt(eps)), stats.weibull_min.std(2.0, scale=np.sqrt(eps))))


def test_truncated_normal_moment_4_vs_scipy():
    """reference tests/models/merging/test_truncated_normal.py:29-42 (rtol 1e-5), closed form of surrogate_posteriors.py:55-73"""
    rng = np.random.default_rng(1)
    loc, scale = rng.random((2, 100))
    scale = scale + 1e-3
    mom4 = O.tn_moment_4(T(loc), T(scale), T(np.zeros(100))).numpy()
    a, b = (0.0 - loc) / scale, np.inf
    assert np.allclose(mom4, stats.truncnorm.moment(4, a, b, loc, scale), rtol=1e-5)


def test_truncated_normal_logprob_mean_var_vs_scipy():
    rng = np.random.default_rng(2)
    loc = rng.uniform(0.05, 3.0, 200)
    scale = loc * 10 ** rng.uniform(-2, 0.3, 200)
    low = np.where(rng.random(200) < 0.2, 0.0, 1e-32)
    z = loc + scale * rng.normal(size=200)
    z = np.abs(z) + 1e-3
    a, b = (low - loc) / scale, (1e10 - loc) / scale
    lp = O.tn_log_prob(T(z), T(loc), T(scale), T(low), T(1e10)).numpy()
    assert np.allclose(lp, stats.truncnorm.logpdf(z, a, b, loc, scale), rtol=1e-9, atol=1e-9)
    assert np.allclose(O.tn_mean(T(loc), T(scale), T(low), T(1e10)).numpy(), stats.truncnorm.mean(a, b, loc, scale), rtol=1e-8)
    assert np.allclose(O.tn_variance(T(loc), T(scale), T(low), T(1e10)).numpy(), stats.truncnorm.var(a, b, loc, scale), rtol=1e-6)


def test_truncated_normal_sampler_is_inverse_cdf_and_distribution():
    rng = np.random.default_rng(3)
    loc, scale, low = np.array([0.8]), np.array([0.5]), np.array([0.0])
    u = rng.random((20000, 1))
    z = O.tn_sample(T(loc), T(scale), T(low), T(1e10), T(u)).numpy()[:, 0]
    a = (low - loc) / scale
    assert np.allclose(z, stats.truncnorm.ppf(u[:, 0], a[0], np.inf, loc[0], scale[0]), rtol=1e-7, atol=1e-9)
    assert stats.kstest(z, lambda x: stats.truncnorm.cdf(x, a[0], np.inf, loc[0], scale[0])).pvalue > 1e-3
    assert z.min() >= 0.0


def test_truncated_normal_pathwise_gradient_vs_finite_differences():
    """TFP's sample-gradient formula (dl, du) equals the derivative of the inverse-CDF map at fixed u."""
    rng = np.random.default_rng(4)
    n = 300
    loc = rng.uniform(0.05, 3.0, n)
    scale = loc * 10 ** rng.uniform(-1.5, 0.3, n)
    low = np.full(n, 1e-32)
    u = rng.uniform(0.01, 0.99, (1, n))
    tl, ts = T(loc).requires_grad_(True), T(scale).requires_grad_(True)
    z = O.tn_sample(tl, ts, T(low), T(1e10), T(u))
    gl, gs = torch.autograd.grad(z.sum(), [tl, ts])
    f = lambda l, s: stats.truncnorm.ppf(u[0], (low - l) / s, np.inf, l, s)
    h = 1e-6
    fd_l = (f(loc + h, scale) - f(loc - h, scale)) / (2 * h)
    fd_s = (f(loc, scale + h) - f(loc, scale - h)) / (2 * h)
    assert np.allclose(gl.numpy(), fd_l, rtol=1e-5, atol=1e-7)
    assert np.allclose(gs.numpy(), fd_s, rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize("dof", [1.0, 2.0, 4.0, 16.0])
def test_likelihoods_vs_scipy(dof):
    """reference tests/models/likelihoods/test_mono.py:12-51: likelihood.log_prob == Normal / StudentT density"""
    rng = np.random.default_rng(5)
    iobs, sig = rng.normal(size=300) * 50, rng.uniform(0.5, 20, 300)
    x = iobs + sig * rng.standard_t(3, size=(2, 300))
    assert np.allclose(O.normal_log_prob(T(x), T(iobs), T(sig)).numpy(), stats.norm.logpdf(x, iobs, sig))
    assert np.allclose(O.studentt_log_prob(T(x), dof, T(iobs), T(sig)).numpy(), stats.t.logpdf(x, dof, iobs, sig))


def test_laue_convolve_sums_harmonics():
    """reference tests/models/likelihoods/test_laue.py:11-36: convolve(iobs[hid]/count[hid]) reproduces iobs, also batched"""
    rng = np.random.default_rng(6)
    hid = np.sort(rng.integers(0, 40, 100))
    _, hid = np.unique(hid, return_inverse=True)
    G = hid.max() + 1
    iobs = np.zeros(100)
    iobs[:G] = rng.uniform(1, 10, G)
    ipred = iobs[hid] / np.bincount(hid)[hid]
    conv = O.laue_convolve(T(ipred), torch.as_tensor(hid)).numpy()
    assert np.allclose(conv[:G], iobs[:G]) and np.all(conv[G:] == 0)
    conv3 = O.laue_convolve(T(np.stack([ipred] * 3)), torch.as_tensor(hid)).numpy()
    assert np.allclose(conv3, conv[None, :])


def test_rice_and_folded_normal_vs_scipy():
    """careless/utils/distributions.py:278-283 (Rice), :333-335 (FoldedNormal)"""
    rng = np.random.default_rng(7)
    x, nu, s = rng.uniform(0.1, 4, 200), rng.uniform(0.0, 3, 200), rng.uniform(0.3, 1.5, 200)
    assert np.allclose(O.rice_log_prob(T(x), T(nu), T(s)).numpy(), stats.rice.logpdf(x, nu / s, scale=s), rtol=1e-8, atol=1e-8)
    assert np.allclose(O.folded_normal_log_prob(T(x), T(nu), T(s)).numpy(), stats.foldnorm.logpdf(x, nu / s, scale=s), rtol=1e-8, atol=1e-8)


def test_mlp_identity_init_and_forward():
    """nn.py:62-78: identity kernels (also non-square), zero bias => output = first two metadata columns after LeakyReLU"""
    ws, bs = O.mlp_identity_init(5, 8, 3)
    assert [w.shape for w in ws] == [(5, 8), (8, 8), (8, 8), (8, 2)]
    x = np.random.default_rng(8).normal(size=(10, 5))
    out = O.mlp_forward(T(x), [T(w) for w in ws], [T(b) for b in bs], 0.01).numpy()
    lr = lambda v: np.where(v > 0, v, 0.01 * v)
    assert np.allclose(out, lr(lr(lr(x[:, :2]))))


def test_scale_bijectors():
    raw = T(np.linspace(-5, 5, 11))
    assert np.allclose(O.scale_bijector(raw, "exp", 1e-7).numpy(), np.exp(raw.numpy()) + 1e-7)
    assert np.allclose(O.scale_bijector(raw, "softplus", 1e-7).numpy(), np.log1p(np.exp(raw.numpy())) + 1e-7)
    with pytest.raises(ValueError):
        O.scale_bijector(raw, "tanh", 0.0)


def test_elbo_gradient_vs_finite_differences():
    from tests import util
    data, cfg, params, x, u_f, eta = util.make_problem(N=64, R=8, L=2, w=8, S=2, likelihood="studentt", dof=4.0)
    u, e = T(u_f), T(eta)
    out, grads = O.elbo_value_and_grads(params, x, cfg, u, e)
    f = lambda: float(O.elbo_forward(params, x, cfg, u, e)["loss"])
    rng = np.random.default_rng(9)
    for t, g in zip(params.tensors(), grads):
        for _ in range(3):
            idx = tuple(int(rng.integers(0, n)) for n in t.shape)
            h = 1e-6 * max(1.0, abs(float(t[idx])))
            v = float(t[idx])
            t[idx] = v + h; fp = f()
            t[idx] = v - h; fm = f()
            t[idx] = v
            fd = (fp - fm) / (2 * h)
            assert abs(fd - float(g[idx])) <= 1e-5 * max(abs(fd), 1.0) + 1e-6, (idx, fd, float(g[idx]))


def test_adam_first_step_and_clipping():
    """tf_keras Adam: first update = -lr * g / (|g| + eps * sqrt(...)) ~ -lr sign(g); clip variants"""
    cfg = O.ElboConfig()
    p = [T([1.0, -2.0, 3.0])]
    g = [T([0.5, -4.0, 0.0])]
    st = O.AdamState.zeros_like(p)
    O.adam_apply(p, g, st, cfg)
    alpha = 1e-3 * math.sqrt(1 - 0.99) / (1 - 0.9)
    m, v = 0.1 * g[0], 0.01 * g[0] ** 2
    assert torch.allclose(p[0], T([1.0, -2.0, 3.0]) - m * alpha / (torch.sqrt(v) + 1e-7))
    gs = [T([3.0, 4.0]), T([0.3])]
    c = O.clip_grads(gs, O.ElboConfig(clipnorm=1.0))
    assert torch.allclose(c[0], T([0.6, 0.8])) and torch.allclose(c[1], T([0.3]))
    c = O.clip_grads(gs, O.ElboConfig(global_clipnorm=1.0))
    n = math.sqrt(25 + 0.09)
    assert torch.allclose(c[0], T([3.0, 4.0]) / n) and torch.allclose(c[1], T([0.3]) / n)
    c = O.clip_grads(gs, O.ElboConfig(clipvalue=0.5))
    assert torch.allclose(c[0], T([0.5, 0.5])) and torch.allclose(c[1], T([0.3]))
    # tf_keras `_clip_gradients` returns after the first active mode: with two flags the value clip never runs [3P-recall]
    c = O.clip_grads(gs, O.ElboConfig(clipnorm=1.0, clipvalue=0.5))
    assert torch.allclose(c[0], T([0.6, 0.8])) and torch.allclose(c[1], T([0.3]))
    c = O.clip_grads(gs, O.ElboConfig(global_clipnorm=1.0, clipvalue=0.1))
    assert torch.allclose(c[0], T([3.0, 4.0]) / n)
    with pytest.raises(ValueError):
        O.clip_grads(gs, O.ElboConfig(clipnorm=1.0, global_clipnorm=1.0))


def test_adam_trajectory_vs_an_independent_implementation():
    """Thirty steps of the oracle's tf_keras Adam against torch.optim.Adam, an implementation the oracle shares no code with.  The
    two differ only in where epsilon enters (Keras: sqrt(v) + eps under the bias-corrected step size; torch: sqrt(v / (1 - b2^t)) + eps),
    i.e. torch with eps / sqrt(1 - b2^t) per step IS the Keras update: checked exactly that way, and with the plain eps to the
    accuracy the eps placement allows."""
    cfg = O.ElboConfig()
    g = torch.Generator().manual_seed(3)
    p0 = torch.randn(40, generator=g, dtype=torch.float64)
    grads = [torch.randn(40, generator=g, dtype=torch.float64) * (1.0 + 0.1 * i) for i in range(30)]
    p = [p0.clone()]
    st = O.AdamState.zeros_like(p)
    q = torch.nn.Parameter(p0.clone())
    q2 = torch.nn.Parameter(p0.clone())
    opt2 = torch.optim.Adam([q2], lr=cfg.learning_rate, betas=(cfg.beta_1, cfg.beta_2), eps=cfg.adam_epsilon)
    for t, gr in enumerate(grads, start=1):
        O.adam_apply(p, [gr], st, cfg)
        # torch, with the epsilon that makes its formula the Keras one at this step
        opt = torch.optim.Adam([q], lr=cfg.learning_rate, betas=(cfg.beta_1, cfg.beta_2), eps=cfg.adam_epsilon / math.sqrt(1.0 - cfg.beta_2 ** t))
        if t > 1:
            opt.load_state_dict({"state": state, "param_groups": opt.state_dict()["param_groups"]})
        q.grad = gr.clone()
        opt.step()
        state = opt.state_dict()["state"]
        q2.grad = gr.clone()
        opt2.step()
        assert torch.allclose(p[0], q.detach(), rtol=0, atol=1e-13), t
    assert torch.allclose(p[0], q2.detach(), rtol=0, atol=1e-5)          # (an element with |g| ~ 1e-3 feels the epsilon placement at 1e-6 per step)


def test_loss_reductions_sum_vs_kl_weight():
    """variational.py:172-177: default = sums / S; with kl_weight = means and weighted KL"""
    from tests import util
    data, cfg, params, x, u_f, eta = util.make_problem(N=64, R=8, L=2, w=8, S=3)
    a = O.elbo_forward(params, x, cfg, T(u_f), T(eta))
    cfg2 = O.ElboConfig(mc_samples=3, kl_weight=0.25)
    b = O.elbo_forward(params, x, cfg2, T(u_f), T(eta))
    assert np.isclose(float(a["nll"]) / 64, float(b["nll"]))
    assert np.isclose(float(a["kl"]) / 8, float(b["kl"]))
    assert np.isclose(float(b["loss"]), float(b["nll"]) + 0.25 * float(b["kl"]))


def test_positional_encoding_layout():
    """careless/utils/positional_encoding.py:3-17"""
    x = np.array([[0.0, 10.0], [1.0, 20.0], [2.0, 30.0]])
    pe = O.positional_encoding(x, 2)
    p = np.array([[-1.0, -1.0], [0.0, 0.0], [1.0, 1.0]])
    ang = np.stack([np.pi * p[:, 0], 2 * np.pi * p[:, 0], np.pi * p[:, 1], 2 * np.pi * p[:, 1]], axis=1)
    assert pe.shape == (3, 8)
    assert np.allclose(pe, np.concatenate([np.cos(ang), np.sin(ang)], axis=1))


def test_output_step_restatement_vs_monte_carlo_moments():
    """oracle `prediction_mean_stddev` / `scale_mean_stddev` (reference variational.py:47-121): the closed-form moments of
    I = Sigma * F^2 per observation -- and their per-harmonic-slot sums for Laue data -- agree with the sample moments of the
    oracle's own forward pass (`elbo_forward`'s ipred, which follows variational.py:154-167 + laue.py:20-25 independently)."""
    from tests import util
    for laue in (False, True):
        S = 20000
        data, cfg, params, x, _, _ = util.make_problem(N=60, R=12, d0=5, L=2, w=16, S=S, laue=laue, perturb=0.02, seed=3)
        rng = np.random.default_rng(0)
        u = torch.as_tensor(rng.random((S, 12))); eta = torch.as_tensor(rng.normal(size=(S, 60)))
        ipred = O.elbo_forward(params, x, cfg, u, eta)["ipred"]
        if laue:
            ipred = O.laue_convolve(ipred, x.harmonic_id)
        ie, isd = O.prediction_mean_stddev(params, x, cfg)
        m, sd = ipred.mean(0), ipred.std(0)
        assert bool(((m - ie).abs() <= 5.0 * isd / math.sqrt(S) + 1e-12).all()), ((m - ie).abs() * math.sqrt(S) / isd).max()
        assert torch.allclose(sd, isd, rtol=0.1, atol=1e-9)       # heavy-tailed (I ~ F^2): the sample sd converges slowly
        if laue:
            G = int(x.harmonic_id.max()) + 1
            assert G < 60 and not ie[G:].any() and not isd[G:].any()
            sm, ssd = O.scale_mean_stddev(params, x, cfg)
            cfg1 = O.ElboConfig(**{**cfg.__dict__, "laue": False})
            sm1, ssd1 = O.scale_mean_stddev(params, x, cfg1)
            assert torch.allclose(sm, torch.zeros(60, dtype=sm.dtype).index_add(0, x.harmonic_id, sm1))
            assert torch.allclose(ssd ** 2, torch.zeros(60, dtype=sm.dtype).index_add(0, x.harmonic_id, ssd1 ** 2))


def test_merged_results_follow_the_reference_floor_on_sigi():
    """`get_results` numerics (io/manager.py:188-197): I = SigF^2 + F^2, SigI^2 = max((I * 1e-5)^2, <F^4> - I^2); F, SigF vs scipy"""
    from tests import util
    data, cfg, params, x, _, _ = util.make_problem(N=60, R=12, d0=5, L=2, w=16, S=1)
    r = O.merged_results(params, x, cfg)
    loc, scale = O.tn_loc_scale(params.q_loc_raw, params.q_scale_raw, cfg.epsilon)
    a = ((x.low - loc) / scale).numpy()
    assert np.allclose(r["F"].numpy(), stats.truncnorm.mean(a, np.inf, loc.numpy(), scale.numpy()), rtol=1e-9)
    assert np.allclose(r["SigF"].numpy(), stats.truncnorm.std(a, np.inf, loc.numpy(), scale.numpy()), rtol=1e-7)
    assert torch.allclose(r["I"], r["F"] ** 2 + r["SigF"] ** 2)
    assert bool((r["SigI"] >= 1e-5 * r["I"]).all())
    hi = O.merged_results(params, x, cfg, max_intensity_snr=10.0)       # a huge floor takes over
    assert torch.allclose(hi["SigI"], 10.0 * hi["I"])


def test_forced_leaky_relu_branches_and_the_rounding_candidates():
    """The gradient gate of the GPU parity tests (tests/test_gpu_parity.py: _assert_grads) re-runs the oracle with LeakyReLU units
    forced onto the other branch.  Pin those hooks on the CPU: no flips = the plain oracle, bit for bit; a pre-activation placed at
    1e-9 is the first candidate `near` reports; forcing it changes the loss by O(1e-9) and the gradient of its own bias from
    leak * g to g -- the derivative of the other branch."""
    from tests import util
    data, cfg, params, x, u_f, eta = util.make_problem(N=40, R=8, d0=5, L=3, w=8, S=2, use_image_scales=False)
    u, e = torch.as_tensor(u_f, dtype=torch.float64), torch.as_tensor(eta, dtype=torch.float64)
    # put pre-activation (layer 1, row 5, unit 3) at -1e-9 by shifting that unit's bias by the amount row 5 needs (other rows move too)
    h = x.metadata.double()
    h = torch.nn.functional.leaky_relu(h @ params.mlp_w[0] + params.mlp_b[0], negative_slope=cfg.leakiness)
    z = h @ params.mlp_w[1] + params.mlp_b[1]
    params.mlp_b[1][3] += -1e-9 - z[5, 3]
    out0, g0 = O.elbo_value_and_grads(params, x, cfg, u, e)
    near = []
    out1, g1 = O.elbo_value_and_grads(params, x, cfg, u, e, flips=(), near=near)
    assert float(out0["loss"]) == float(out1["loss"]) and all(torch.equal(a, b) for a, b in zip(g0, g1))
    near.sort()
    assert near and near[0][1:] == (1, 5, 3) and near[0][0] < 1e-2
    out2, g2 = O.elbo_value_and_grads(params, x, cfg, u, e, flips=[(1, 5, 3)])
    assert abs(float(out2["loss"]) - float(out0["loss"])) < 1e-6 * abs(float(out0["loss"]))
    d = [float((a - b).abs().max()) for a, b in zip(g0, g2)]
    assert max(d) > 0.0
    # tensors: q_loc, q_scale, W0, b0, W1, b1, ...: the flipped unit's bias gradient gains row 5's share scaled 1 / leak
    # d loss / d z[5, 3] on the negative branch is leak * G, on the positive one G: the two bias gradients differ by (1 - leak) G
    diff_b1 = (g2[5] - g0[5])
    others = torch.cat([diff_b1[:3], diff_b1[4:]]).abs().max()
    assert diff_b1.abs().argmax().item() == 3 and float(others) < 1e-8 * float(diff_b1[3].abs())      # (the others move with the 1e-9 of activation)
    # the layers ABOVE the unit see the same activations to 1e-9: their gradients do not move beyond that
    assert float((g2[-1] - g0[-1]).abs().max()) <= 1e-6 * float(g0[-1].abs().max())
