"""Round 6: where does the run-to-run defect of the dZ_0-storing lane instances live?  (NOTEBOOK R5.9 / R5.12, R6.1)

Runs ONE configuration `--runs` times on fresh engines with identical inputs and seed through the library named by CARELESS_HIP_LIB (a
diagnostic build: scripts/probe/build_lane_variants.py) and reports, against the first run, WHICH outputs move and where they sit:
the NLL, dZ_0 per (packed row, feature) -> (wave tile, lane), dz_f per (reflection, sample), the scaler's gradient partials per workgroup.

    CARELESS_HIP_LIB=... python scripts/probe/lane_defect_probe.py --config image_layers2_peeled_d21 --runs 12
"""
from __future__ import annotations

import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import util  # noqa: E402

CONFIGS = {
    "image_layers2_peeled_d21": dict(R=40, d0=5, posenc=True, L=20, w=10, S=2, perturb=0.02, image_layers=2, n_images=17),
    "peeled_dZ0_out": dict(R=40, d0=37, L=20, w=10, S=3, perturb=0.02),
    "image_layers2": dict(R=40, d0=5, L=20, w=10, S=2, perturb=0.02, image_layers=2, n_images=23),
    "cli_default": dict(R=40, d0=5, L=20, w=10, S=1, perturb=0.02),
    # (the other fused kernels carried the same inline-assembly LeakyReLU until round 6: soak them as well)
    "headline_5x64": dict(R=40, d0=5, posenc=True, L=5, w=64, S=8, likelihood="studentt", dof=16.0),
    "narrow_12x12": dict(R=40, d0=5, L=12, w=12, S=2, perturb=0.03),
    "depth10_10x10": dict(R=40, d0=5, L=10, w=10, S=1, perturb=0.03),
    "chain_24x10": dict(R=40, d0=5, L=24, w=10, S=1, perturb=0.01),
    "image_layers2_depth10": dict(R=40, d0=5, L=10, w=10, S=2, perturb=0.03, image_layers=2, n_images=23),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="image_layers2_peeled_d21")
    ap.add_argument("--runs", type=int, default=12)
    ap.add_argument("--N", type=int, default=5000)
    ap.add_argument("--images", type=int, default=0)
    ap.add_argument("--same-engine", action="store_true", help="repeat the step on ONE engine instead of fresh ones")
    ap.add_argument("--tag", default="")
    a = ap.parse_args()
    from careless_amd.engine import ElboEngine
    kw = dict(CONFIGS[a.config], N=a.N)
    if a.images:
        kw["n_images"] = a.images
    data, cfg, params, x, _, _ = util.make_problem(**kw)
    inputs = util.reference_inputs(data)
    ref = None
    rec = dict(tag=a.tag, lib=os.environ.get("CARELESS_HIP_LIB", "default"), config=a.config, N=a.N, runs=a.runs, kernel=None, nll=[], n_bad_runs=0,
               per_run=[])
    eng = None
    for r in range(a.runs):
        if eng is None or not a.same_engine:
            eng = ElboEngine(util.build_model(data, cfg, params, kw["L"], kw["w"]), inputs, seed=99)
        eng.forward_backward(3)
        torch.cuda.synchronize()
        rec["kernel"] = eng.kernel_name()
        obs = eng.obs
        out = dict(nll=float(eng.loss_terms()["nll"]), grads=eng.grads.clone(), dz_f=eng.dz_f.clone(), partials=obs.partials.clone())
        pb = getattr(obs, "peel", None)
        if pb is not None:
            out["dz0"] = pb["dz0"].clone().view(-1, obs.n_pad)
            out["u"] = pb["u"].clone().view(-1, obs.n_pad)
        rec["nll"].append(out["nll"])
        if ref is None:
            ref = out
            rec["n_pad"], rec["grid"] = int(obs.n_pad), int(obs.grid)
            continue
        pr = dict(run=r, nll_diff=out["nll"] - ref["nll"])
        gmax = float(ref["grads"].abs().max())
        pr["grad_maxdiff_rel"] = float((out["grads"] - ref["grads"]).abs().max()) / gmax
        pr["dzf_n_diff"] = int(((out["dz_f"] - ref["dz_f"]).abs() > 1e-5 * ref["dz_f"].abs().max()).sum())
        dp = (out["partials"] - ref["partials"]).abs().view(rec["grid"], -1)
        pr["partials_blocks_diff"] = torch.nonzero(dp.max(dim=1).values > 0).flatten().tolist()[:40]
        if "dz0" in out:
            pr["u_diff"] = int((out["u"] != ref["u"]).sum())
            dd = (out["dz0"] - ref["dz0"]).abs()
            scale = float(ref["dz0"].abs().max())
            bad = dd > 1e-6 * scale
            rows = torch.nonzero(bad.any(dim=0)).flatten().cpu().numpy()
            pr["dz0_bad_rows"] = int(rows.size)
            pr["dz0_maxdiff_rel"] = float(dd.max()) / scale
            if rows.size:
                tiles, lanes = rows // 64, rows % 64
                ut, ct = np.unique(tiles, return_counts=True)
                pr["tiles"] = {int(t): int(c) for t, c in zip(ut[:40], ct[:40])}
                pr["n_tiles_touched"] = int(ut.size)
                pr["lane_hist16"] = np.bincount(lanes % 16, minlength=16).tolist()
                pr["lane_hist_q"] = np.bincount(lanes // 16, minlength=4).tolist()
                pr["feat_hist"] = bad.sum(dim=1).tolist()
                pr["first_rows"] = rows[:24].tolist()
                k = rows[0]
                pr["example"] = dict(row=int(k), ref=ref["dz0"][:, k].tolist(), got=out["dz0"][:, k].tolist())
        if abs(pr["nll_diff"]) > 0 or pr["grad_maxdiff_rel"] > 2e-6:
            rec["n_bad_runs"] += 1
        rec["per_run"].append(pr)
    rec["distinct_nll"] = len(set(round(v, 6) for v in rec["nll"]))
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
