"""Oracle-independent internal consistency of the engine (round 5): the gradient the backward kernels produce is the derivative of the loss
the forward kernels produce.  Central finite differences of the ENGINE's own loss along random directions of every trainable tensor, on
fixed injected noise (the pathwise sample gradient of q included: the same uniforms give a differentiable z(a, b)) -- no function of
`oracle/` is called.  It cannot see a shared misreading of the reference (tests/test_recovery.py looks at that from the other side), it does
see any disagreement between a forward and a backward kernel, for every kernel the routing table holds.
Reference: the gradient is `tape.gradient(loss, trainable_variables)`, careless/models/merging/variational.py:197-202."""
import numpy as np
import pytest
import torch

from tests import util

pytestmark = pytest.mark.gpu

CASES = {
    "mono_5x64_studentt_S3": dict(N=700, R=50, d0=5, posenc=True, L=5, w=64, S=3, likelihood="studentt", dof=16.0),
    "cli_default_20x10": dict(N=600, R=40, d0=5, L=20, w=10, S=2, perturb=0.02),
    "peeled_first_layer_20x10_d37": dict(N=600, R=40, d0=37, L=20, w=10, S=2, perturb=0.02),
    "narrow_7x12": dict(N=500, R=40, d0=5, L=7, w=12, S=2, perturb=0.03),
    "width_16_exactly_9x16": dict(N=500, R=40, d0=5, L=9, w=16, S=2, perturb=0.03),
    "laue_single_pass_2x32": dict(N=600, R=50, L=2, w=32, S=2, laue=True),
    "double_wilson_2x32": dict(N=500, R=60, d0=5, L=2, w=32, S=2, double_wilson=True),
    "chained_12x32": dict(N=500, R=40, d0=5, L=12, w=32, S=2),
    "wide_2x96_ev11": dict(N=500, R=40, d0=5, L=2, w=96, S=2, ev11=True),
    "image_layers1_2x32": dict(N=600, R=40, d0=5, L=2, w=32, S=2, n_images=5, image_layers=1),
}


@pytest.mark.parametrize("name", list(CASES))
def test_engine_gradient_is_the_derivative_of_its_own_loss(name):
    from careless_amd.engine import ElboEngine
    kw = dict(CASES[name])
    data, cfg, params, x, u_f, eta = util.make_problem(**kw)
    inputs = util.reference_inputs(data)
    model = util.build_model(data, cfg, params, kw["L"], kw["w"])
    model.deterministic = not (kw.get("image_layers") or kw.get("two_pass"))       # fixed summation order where the mode exists: less noise in the differences
    eng = ElboEngine(model, inputs, seed=3)
    du, de = eng._noise_to_device(u_f, eta)              # the same injected noise for every evaluation

    def loss():
        eng.forward_backward(0, du, de)
        torch.cuda.synchronize()
        return eng.loss_terms()["loss"]

    l0 = loss()
    g = eng.grads.clone().double()
    p0 = eng.params.clone()
    rng = np.random.default_rng(5)
    lay = eng.layout
    checked = 0
    for lo, hi in zip(lay.seg_off[:-1], lay.seg_off[1:]):                          # one direction per trainable tensor
        if hi - lo == 0 or float(g[lo:hi].abs().max()) == 0.0:
            continue
        v = torch.zeros_like(p0, dtype=torch.float64)
        # along the gradient, plus (tensors of four or more entries) a random part: neither tiny nor blind to the tensor's small entries
        v[lo:hi] = g[lo:hi] / float(g[lo:hi].norm())
        if hi - lo >= 4:
            r = torch.as_tensor(rng.normal(size=hi - lo), device=p0.device)
            v[lo:hi] += r / float(r.norm())
        slope = abs(float((g * v).sum()))
        vmax = float(v[lo:hi].abs().max())
        best = None
        # fp32 parameters, loss known to ~1e-7 relative, LeakyReLU kinks (a bias step of 0.05 flips units): steps that move the loss by a
        # few 1e-4 of itself and no entry by more than `cap`; the best of three sizes counts
        for target, cap in ((4e-4, 0.05), (1e-4, 0.01), (3e-5, 0.003)):
            h = min(target * abs(l0) / max(slope, 1e-30), cap / vmax)
            eng.params.copy_((p0.double() + h * v).float())
            hp = eng.params.double() - p0.double()                                 # the step actually taken in fp32
            lp = loss()
            eng.params.copy_((p0.double() - h * v).float())
            hm = eng.params.double() - p0.double()
            lm = loss()
            exp = float((g * (hp - hm)).sum())
            err = abs((lp - lm) - exp) / max(abs(exp), 1e-6 * abs(l0))
            best = err if best is None else min(best, err)
        eng.params.copy_(p0)
        assert best < 3e-2, (name, lo, hi, best, slope, l0)
        checked += 1
    assert checked >= 4
