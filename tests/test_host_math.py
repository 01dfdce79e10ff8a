"""careless_amd/csrc/cl_math.h (the scalar fp32 formulas every HIP kernel uses) compiled for the HOST with g++ and checked
against the fp64 oracle / scipy: a formula error shows up here without a GPU.  CPU only."""
import ctypes
import os
import shutil
import subprocess
import tempfile

import numpy as np
import pytest
import torch
from scipy import special

from oracle import elbo_oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))
fp = lambda a: a.ctypes.data_as(ctypes.c_void_p)


@pytest.fixture(scope="module")
def hm():
    if shutil.which("g++") is None:
        pytest.skip("g++ not available")
    d = tempfile.mkdtemp(prefix="cl_hm_")
    so = os.path.join(d, "libhm.so")
    subprocess.check_call(["g++", "-O2", "-shared", "-fPIC", "-o", so, os.path.join(HERE, "host_math_check.cpp")])
    return ctypes.CDLL(so)


def test_ndtri_lower(hm):
    rng = np.random.default_rng(0)
    p = np.concatenate([10 ** rng.uniform(-30, -0.31, 5000), rng.uniform(0.001, 0.5, 5000)]).astype(np.float32)
    out = np.empty_like(p)
    hm.hm_ndtri_lower(len(p), fp(p), fp(out))
    ref = special.ndtri(p.astype(np.float64))
    assert np.max(np.abs(out - ref) / np.maximum(np.abs(ref), 1e-3)) < 2e-6


def test_truncated_normal_element(hm):
    rng = np.random.default_rng(1)
    n = 20000
    loc0 = rng.uniform(0.05, 3, n)
    sc0 = loc0 * 10 ** rng.uniform(-2.5, 0.3, n)
    a, b = np.log(loc0).astype(np.float32), np.log(sc0).astype(np.float32)
    low = np.where(rng.random(n) < 0.1, 0.0, 1e-32).astype(np.float32)
    u = rng.random(n).astype(np.float32)
    out = np.empty((n, 10), np.float32)
    hm.hm_tn(n, fp(a), fp(b), fp(low), ctypes.c_float(1e10), ctypes.c_float(1e-7), fp(u), fp(out))
    T = lambda v: torch.as_tensor(np.asarray(v, dtype=np.float64))
    loc, scale = O.tn_loc_scale(T(a), T(b), 1e-7)
    ll, sl = loc.clone().requires_grad_(True), scale.clone().requires_grad_(True)
    z = O.tn_sample(ll, sl, T(low), T(1e10), T(u)[None, :])[0]
    gl, gs = torch.autograd.grad(z.sum(), [ll, sl])
    zd = z.detach().clone().requires_grad_(True)
    lq = O.tn_log_prob(zd, ll, sl, T(low), T(1e10))
    g = torch.autograd.grad(lq.sum(), [zd, ll, sl])
    sc = scale.numpy()
    def err(k, ref, nat):
        return np.max(np.abs(out[:, k] - ref) / nat)
    assert err(0, z.detach().numpy(), loc0) < 1e-5
    assert err(1, lq.detach().numpy(), np.maximum(np.abs(lq.detach().numpy()), 1.0)) < 1e-4
    assert err(2, gl.numpy(), 1.0) < 1e-5
    assert err(3, gs.numpy(), np.maximum(np.abs(gs.numpy()), 1.0)) < 1e-5
    assert err(4, g[0].numpy(), np.maximum(np.abs(g[0].numpy()), 1.0 / sc)) < 1e-4
    assert err(5, g[1].numpy(), np.maximum(np.abs(g[1].numpy()), 1.0 / sc)) < 1e-4
    assert err(6, g[2].numpy(), np.maximum(np.abs(g[2].numpy()), 1.0 / sc)) < 2e-4


def test_wilson_likelihood_bijector(hm):
    rng = np.random.default_rng(2)
    n = 5000
    T = lambda v: torch.as_tensor(np.asarray(v, dtype=np.float64))
    z = rng.uniform(0.01, 4, n).astype(np.float32)
    c = (rng.random(n) < 0.3).astype(np.int32)
    es = rng.choice([1.0, 2.0, 3.0, 4.0, 6.0], n).astype(np.float32)
    lp, dlp = np.empty(n, np.float32), np.empty(n, np.float32)
    hm.hm_wilson(n, fp(z), fp(c), fp(es), fp(lp), fp(dlp))
    zt = T(z).requires_grad_(True)
    ref = O.wilson_log_prob(zt, torch.as_tensor(c.astype(bool)), T(es), T(1.0))
    gref, = torch.autograd.grad(ref.sum(), zt)
    assert np.allclose(lp, ref.detach().numpy(), rtol=2e-5, atol=2e-6)
    assert np.allclose(dlp, gref.numpy(), rtol=2e-5, atol=2e-6)
    ip, io, sg = rng.normal(size=n).astype(np.float32) * 30, rng.normal(size=n).astype(np.float32) * 30, rng.uniform(0.5, 9, n).astype(np.float32)
    for kind, dof in ((0, 0.0), (1, 4.0), (1, 16.0)):
        const = 0.0 if kind == 0 else float(special.gammaln((dof + 1) / 2) - special.gammaln(dof / 2) - 0.5 * np.log(dof * np.pi))
        ll, dll = np.empty(n, np.float32), np.empty(n, np.float32)
        hm.hm_lik(n, fp(ip), fp(io), fp(sg), kind, ctypes.c_float(dof), ctypes.c_float(const), fp(ll), fp(dll))
        it = T(ip).requires_grad_(True)
        r = O.normal_log_prob(it, T(io), T(sg)) if kind == 0 else O.studentt_log_prob(it, dof, T(io), T(sg))
        gr, = torch.autograd.grad(r.sum(), it)
        assert np.allclose(ll, r.detach().numpy(), rtol=2e-5, atol=2e-5)
        assert np.allclose(dll, gr.numpy(), rtol=2e-5, atol=1e-6)
    raw = rng.uniform(-8, 8, n).astype(np.float32)
    for kind, name in ((0, "exp"), (1, "softplus")):
        s, ds = np.empty(n, np.float32), np.empty(n, np.float32)
        hm.hm_bij(n, fp(raw), kind, ctypes.c_float(1e-7), fp(s), fp(ds))
        rt = T(raw).requires_grad_(True)
        r = O.scale_bijector(rt, name, 1e-7)
        gr, = torch.autograd.grad(r.sum(), rt)
        assert np.allclose(s, r.detach().numpy(), rtol=1e-5) and np.allclose(ds, gr.numpy(), rtol=1e-5)


def test_noise_generator_statistics(hm):
    n = 200000
    for s in (0, 3, 4, 7):
        un, nr = np.empty(n, np.float32), np.empty(n, np.float32)
        hm.hm_noise(n, ctypes.c_ulonglong(1234), 7, s, ctypes.c_ulonglong(0), fp(un), fp(nr))
        assert 0.0 < un.min() and un.max() < 1.0
        assert abs(un.mean() - 0.5) < 3e-3 and abs(nr.mean()) < 8e-3 and abs(nr.std() - 1.0) < 8e-3
    a, b = np.empty(n, np.float32), np.empty(n, np.float32)
    hm.hm_noise(n, ctypes.c_ulonglong(1234), 7, 1, ctypes.c_ulonglong(0), fp(un), fp(a))
    hm.hm_noise(n, ctypes.c_ulonglong(1234), 7, 5, ctypes.c_ulonglong(0), fp(un), fp(b))   # the sine partner of sample 1
    assert abs(np.corrcoef(a, b)[0, 1]) < 0.01 and abs(np.corrcoef(a * a, b * b)[0, 1]) < 0.01


def test_bessel_and_double_wilson_conditional(hm):
    rng = np.random.default_rng(3)
    x = np.concatenate([np.linspace(0, 8, 3000), np.linspace(8, 3000, 3000)]).astype(np.float32)
    a, b = np.empty_like(x), np.empty_like(x)
    hm.hm_bessel(len(x), fp(x), fp(a), fp(b))
    assert np.max(np.abs(a - special.i0e(x.astype(float))) / special.i0e(x.astype(float))) < 2e-6
    assert np.max(np.abs(b - special.i1e(x.astype(float))) / np.maximum(special.i1e(x.astype(float)), 1e-3)) < 5e-6
    n = 10000
    z, zp = rng.uniform(0.05, 4, n).astype(np.float32), rng.uniform(0.05, 4, n).astype(np.float32)
    has = (rng.random(n) < 0.9).astype(np.int32)
    r = rng.uniform(0.0, 0.97, n).astype(np.float32)
    c = (rng.random(n) < 0.3).astype(np.int32)
    es = rng.choice([1.0, 2.0, 3.0], n).astype(np.float32)
    lp, dz, dzp, dr = np.empty(n, np.float32), np.empty(n, np.float32), np.empty(n, np.float32), np.empty(n, np.float32)
    hm.hm_dw(n, fp(z), fp(zp), fp(has), fp(r), fp(c), fp(es), fp(lp), fp(dz), fp(dzp), fp(dr))
    T = lambda v: torch.as_tensor(np.asarray(v, dtype=np.float64))
    zt, zpt, rt = T(z).requires_grad_(True), T(zp).requires_grad_(True), T(r).requires_grad_(True)
    cb, hb = torch.as_tensor(c.astype(bool)), torch.as_tensor(has.astype(bool))
    loc = torch.where(hb, zpt * rt, torch.zeros(n, dtype=torch.float64))
    scale = torch.where(cb, torch.sqrt(T(es) * (1 - rt ** 2)), torch.sqrt(0.5 * T(es) * (1 - rt ** 2)))
    ref = torch.where(cb, O.folded_normal_log_prob(zt, loc, scale), O.rice_log_prob(zt, loc, scale))
    g = torch.autograd.grad(ref.sum(), [zt, zpt, rt])
    nat = lambda v: np.maximum(np.abs(v), 1.0)
    assert np.max(np.abs(lp - ref.detach().numpy()) / nat(ref.detach().numpy())) < 1e-4
    assert np.max(np.abs(dz - g[0].numpy()) / nat(g[0].numpy())) < 1e-4
    assert np.max(np.abs(dzp - g[1].numpy()) / nat(g[1].numpy())) < 1e-4
    assert np.max(np.abs(dr - g[2].numpy()) / nat(g[2].numpy())) < 2e-4
