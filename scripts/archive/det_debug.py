"""Which tensors differ between two runs of the same step (deterministic mode on / off)?"""
import sys, numpy as np, torch
sys.path.insert(0, ".")
from tests import util
from careless_amd.engine import ElboEngine
kw = dict(N=1500, R=60, d0=5, L=5, w=64, S=3, likelihood="studentt", dof=8.0, n_images=7)
data, cfg, params, x, u_f, eta = util.make_problem(**kw)
inputs = util.reference_inputs(data)
for det in (True, False):
    gs = []
    for rep in range(3):
        m = util.build_model(data, cfg, params, kw["L"], kw["w"]); m.deterministic = det
        e = ElboEngine(m, inputs, seed=5)
        e.forward_backward(1); torch.cuda.synchronize()
        gs.append(([g.clone() for g in e.grad_tensors()], e.loss_terms(), e.dz_f.clone()))
    for rep in (1, 2):
        d = [(i, int((a != b).sum()), float((a - b).abs().max())) for i, (a, b) in enumerate(zip(gs[0][0], gs[rep][0])) if not torch.equal(a, b)]
        print("det", det, "rep", rep, "differing tensors (index, count, maxabs):", d, "dz_f differs:", int((gs[0][2] != gs[rep][2]).sum()), "nll equal", gs[0][1]["nll"] == gs[rep][1]["nll"], "kl equal", gs[0][1]["kl"] == gs[rep][1]["kl"])
