#!/usr/bin/env python3
"""Replay the committed golden vectors (tests/golden/*.npz) through the REAL reference (rs-station/careless on
TensorFlow / TFP / tf_keras) and diff loss, gradients and the 6-step Adam trajectory.

This is the one-command pin of the oracle SURVEY.md 8c(v) asks for.  It cannot run in the build container or on the GPU box
(TensorFlow is not installed there and the reference never travels); run it wherever `import careless` works:

    python3.10 -m venv /tmp/ref && . /tmp/ref/bin/activate
    pip install "tensorflow==2.18.0" "tensorflow-probability[tf]==0.25" tf_keras reciprocalspaceship "careless==0.5.4"
        # (= the reference's pyproject.toml:14-19; or: pip install -e /path/to/reference)
    python scripts/replay_golden_in_reference.py [--cases mono_2x32_normal_S3 ...] [--rtol 1e-4]

What it does per case (tests/golden/cases.json holds the problem sizes, the .npz everything else):
  * builds the reference's own plugin objects exactly as `DataManager.build_model` wires them (careless/io/manager.py:432-506):
    `WilsonPrior`, `TruncatedNormal.from_loc_and_scale`, `MLPScaler` [+ `ImageScaler` -> `HybridImageScaler`] or
    `NeuralImageScaler`, mono / Laue `NormalLikelihood` / `StudentTLikelihood` / `*Ev11Likelihood`, `VariationalMergingModel`,
    `tfk.optimizers.Adam(1e-3, 0.9, 0.99)`; the `inputs` tuple is the file's `inputs_*` arrays in `BaseModel.input_index` order
    with the reference's shapes (ids (N,1) int64, data (N,1) float32; tests/models/merging/test_variational_mono.py:22-77);
  * loads the golden parameters `param_XX` into the model's variables;
  * replaces the two Monte-Carlo draws by the file's injected noise -- the ONLY patch applied to the reference:
      - `careless.models.merging.surrogate_posteriors.TruncatedNormal.sample` (:50-53): inverse CDF of the file's uniforms
        `u_f`, z = loc + scale * ndtri(Phi(alpha) + u (Phi(beta) - Phi(alpha))), then `tf.maximum(low, z)` as the reference does;
        written in differentiable tf ops, its gradient IS the pathwise gradient TFP's sampler implements implicitly;
      - `tfd.Normal._sample_n` (the scaler's `scale_dist.sample`, variational.py:156-157): loc + scale * `eta`;
  * compares  loss / NLL / KL  (model.losses, metrics),  every gradient tensor (tf.GradientTape over model.trainable_variables,
    matched to `grad_XX` by the oracle's tensor order),  and  6 x `train_step_with_gradient_norm` on `traj_u` / `traj_eta`
    against `traj_loss`, `traj_gnorm` and the final parameters `final_XX`.
Two cases exist to make the first TensorFlow run decisive on the two [3P-recall] items the oracle carries: `mono_2x16_clipnorm_and_
clipvalue_S2` (both clip flags: tf_keras applies the first active mode only -- the trajectory differs from a cumulative clip in the
first step) and `mono_2x16_extreme_uniforms_S4` (injected uniforms at 1 - 2^-24, where the [tiny, 1 - eps] clip of TFP's
sample gradient is active; with loc = exp(a) > 0 >= low the truncation point cannot sit in the upper tail, so the ends of u are the
only way to reach the clip).  NOTE for the second one: patch 1 below differentiates the inverse CDF exactly and does NOT reproduce
TFP's clip; its `grad_00 / grad_01` rows therefore show the clip's effect as a mismatch there if and only if the oracle's
recollection of the clip is right -- compare those two rows with `--no-tn-patch`, which leaves TFP's own sampler gradient in place
and checks d z / d(loc, scale) on TFP's own draws against the oracle's formula evaluated at the same draws.
Exit code 0 = every compared number within --rtol (default 1e-4, the north_star tolerance); the PASS / FAIL table it ends with is the
evidence.
The double-Wilson case: the reference builds `DoubleWilsonPrior` from gemmi-backed `ReciprocalASUCollection` objects
(careless/models/priors/wilson.py:82-138); the golden file holds what that constructor derives from them (reflids, root, asu_ids,
centric, multiplicity, r), so the script allocates the reference's class without running its constructor and assigns exactly the
attributes `log_prob` reads (:146-175) -- the arithmetic under test is the reference's own `log_prob`, `RiceWoolfson`, `Rice` and
`FoldedNormal`.
"""
import argparse
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(os.path.dirname(HERE), "tests", "golden")


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b)) / max(float(np.max(np.abs(b))), 1e-30))


def tn_gradient_check(tf, tfd, cases, names, rtol):
    """TFP's own sampler, TFP's own implicit gradient: sum_s dz/dloc and sum_s dz/dscale per reflection against the formula the
    oracle and the HIP kernels use (oracle._TNStdSample.backward): cdf = clip((Phi(e) - Phi(alpha)) / Z, tiny, 1 - eps),
    dl = exp((e^2 - alpha^2) / 2 + log1p(-cdf)), du = exp((e^2 - beta^2) / 2 + log cdf), dz/dloc = 1 - dl - du,
    dz/dscale = e - alpha dl - beta du -- evaluated in float64 at the draws TFP made.  The draws themselves are TF's (no injected
    noise), so extreme cdf values occur with their natural probability only: run it with many samples."""
    from scipy.special import ndtr
    worst = 0.0
    for name in names:
        z = np.load(os.path.join(GOLD, name + ".npz"))
        a, b = z["param_00"].astype(np.float32), z["param_01"].astype(np.float32)
        centric = np.asarray(z["centric"], bool)
        low = (1e-32 * ~centric).astype("float32")
        loc, scale = tf.Variable(np.exp(a)), tf.Variable(np.exp(b) + np.float32(1e-7))
        with tf.GradientTape() as tape:
            zs = tfd.TruncatedNormal(loc, scale, low, 1e10).sample(4096, seed=1)
            tot = tf.reduce_sum(zs)
        gl, gs = [g.numpy().astype(np.float64) for g in tape.gradient(tot, [loc, scale])]
        L, Sc, zs = loc.numpy().astype(np.float64), scale.numpy().astype(np.float64), zs.numpy().astype(np.float64)
        e, al, be = (zs - L) / Sc, (low - L) / Sc, (1e10 - L) / Sc
        Z = ndtr(be) - ndtr(al)
        cdf = np.clip((ndtr(e) - ndtr(al)) / Z, np.finfo(np.float32).tiny, 1.0 - np.finfo(np.float32).eps)
        dl = np.exp(0.5 * (e * e - al * al) + np.log1p(-cdf))
        du = np.exp(0.5 * (e * e - be * be) + np.log(cdf))
        want_l, want_s = (1.0 - dl - du).sum(0), (e - al * dl - be * du).sum(0)
        el, es = rel(gl, want_l), rel(gs, want_s)
        worst = max(worst, el, es)
        print(f"{name:44s} sum_s dz/dloc {el:.2e}   sum_s dz/dscale {es:.2e}")
    print(f"worst: {worst:.3e} (tolerance {rtol:g}) -> {'PASS' if worst <= rtol else 'FAIL'}")
    return 0 if worst <= rtol else 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", nargs="*", default=None)
    ap.add_argument("--rtol", type=float, default=1e-4)
    ap.add_argument("--no-tn-patch", action="store_true", help="leave TFP's truncated-normal sampler (and its implicit sample gradient) "
                    "in place and check d z / d(loc, scale) of TFP's own draws against the oracle's closed form, clip included")
    args = ap.parse_args()

    import tensorflow as tf
    import tensorflow_probability as tfp
    import tf_keras as tfk
    from tensorflow_probability import bijectors as tfb
    from tensorflow_probability import distributions as tfd
    from tensorflow_probability.python.internal import special_math

    from careless.models.likelihoods import laue as laue_lik
    from careless.models.likelihoods import mono as mono_lik
    from careless.models.merging import surrogate_posteriors as sp
    from careless.models.merging.variational import VariationalMergingModel
    from careless.models.priors.wilson import DoubleWilsonPrior, WilsonPrior
    from careless.models.priors.base import Prior
    from tensorflow_probability import util as tfu
    from careless.models.scaling.image import HybridImageScaler, ImageScaler, NeuralImageScaler
    from careless.models.scaling.nn import MLPScaler

    noise = {"u": None, "eta": None}

    def tn_sample(self, sample_shape=(), *a, **k):          # patch 1: injected uniforms through the inverse CDF
        d = self.distribution
        loc, scale = tf.convert_to_tensor(d.loc), tf.convert_to_tensor(d.scale)
        low, high = tf.cast(d.low, loc.dtype), tf.cast(d.high, loc.dtype)
        alpha, beta = (low - loc) / scale, (high - loc) / scale
        ca, cb = special_math.ndtr(alpha), special_math.ndtr(beta)
        e = special_math.ndtri(ca + tf.constant(noise["u"], loc.dtype) * (cb - ca))
        return tf.maximum(low, loc + scale * e)
    sp.TruncatedNormal.sample = tn_sample

    def normal_sample_n(self, n, seed=None):                 # patch 2: injected standard normals for the scale
        loc, scale = tf.convert_to_tensor(self.loc), tf.convert_to_tensor(self.scale)
        return loc + scale * tf.constant(noise["eta"], loc.dtype)
    tfd.Normal._sample_n = normal_sample_n

    cases = json.load(open(os.path.join(GOLD, "cases.json")))
    names = args.cases or list(cases)
    if args.no_tn_patch:
        return tn_gradient_check(tf, tfd, cases, names, args.rtol)
    worst, table = 0.0, []
    for name in names:
        kw = cases[name]
        z = np.load(os.path.join(GOLD, name + ".npz"))
        S, L, w = kw["S"], kw["L"], kw["w"]
        eps = 1e-7
        col = lambda a, t: np.asarray(a).astype(t).reshape(-1, 1)
        inputs = [col(z["inputs_refl_id"], np.int64), col(z["inputs_image_id"], np.int64), col(z["inputs_file_id"], np.int64),
                  np.asarray(z["inputs_metadata"], np.float32), col(z["inputs_intensities"], np.float32),
                  col(z["inputs_uncertainties"], np.float32)]
        laue = bool(kw.get("laue"))
        if laue:
            inputs += [col(z["data_wavelength"], np.float32), col(z["data_harmonic_id"], np.int64)]
        inputs = tuple(inputs)
        params = [z[k] for k in sorted(k for k in z.files if k.startswith("param_"))]
        grads = [z[k] for k in sorted(k for k in z.files if k.startswith("grad_"))]
        finals = [z[k] for k in sorted(k for k in z.files if k.startswith("final_"))]

        centric = np.asarray(z["centric"], bool)
        mult = np.asarray(z["multiplicity"], np.float32)
        dw = bool(kw.get("double_wilson"))
        if dw:
            # the reference's class with the attributes its constructor would derive from the ASU collection (wilson.py:112-138)
            prior = DoubleWilsonPrior.__new__(DoubleWilsonPrior)
            Prior.__init__(prior)
            prior.parents, prior.optimize_r = [None, 0], bool(kw.get("optimize_dw_r"))
            r0 = np.asarray(z["data_dw_r"], np.float32)
            prior.r = tfu.TransformedVariable(tf.convert_to_tensor(r0), tfb.Sigmoid()) if prior.optimize_r else tf.convert_to_tensor(r0)
            prior.centric, prior.multiplicity = centric, mult
            prior.asu_ids = np.asarray(z["data_asu_ids"], np.int32)
            prior.sigma = 1.0
            prior.reflids = np.asarray(z["data_parent_ids"], np.int32)
            prior.absent = tf.convert_to_tensor(prior.reflids == -1)
            prior.root = np.asarray(z["data_root"], bool)
            prior.wilson_prior = WilsonPrior(centric, mult, 1.0)
        else:
            prior = WilsonPrior(centric, mult)
        low = (1e-32 * ~centric).astype("float32")
        q = sp.TruncatedNormal.from_loc_and_scale(np.exp(params[0]).astype("float32"), (np.exp(params[1]) + eps).astype("float32"),
                                                  low, scale_shift=eps)
        mod = laue_lik if laue else mono_lik
        dof = kw.get("dof")
        if kw.get("ev11"):
            lik = mod.NormalEv11Likelihood() if kw.get("likelihood", "normal") == "normal" else mod.StudentTEv11Likelihood(dof)
        else:
            lik = mod.NormalLikelihood() if kw.get("likelihood", "normal") == "normal" else mod.StudentTLikelihood(dof)
        bij = tfb.Chain([tfb.Shift(eps), tfb.Softplus() if kw.get("bijector") == "softplus" else tfb.Exp()])
        shift = kw.get("shift") or None
        n_images = int(z["n_images"])
        k_img = int(kw.get("image_layers", 0))
        use_img = kw.get("use_image_scales", True) and k_img == 0
        if k_img:
            scaler = NeuralImageScaler(k_img, n_images, L, w, epsilon=eps, scale_bijector=bij, scale_multiplier=shift)
        else:
            mlp = MLPScaler(L, w, epsilon=eps, scale_bijector=bij, scale_multiplier=shift)
            scaler = HybridImageScaler(mlp, ImageScaler(n_images)) if use_img else mlp
        model = VariationalMergingModel(q, prior, lik, scaler, S, kl_weight=kw.get("kl_weight"))
        # (clip flags as DataManager.build_model passes them, io/manager.py:494-501: the two-flag case pins the precedence of
        #  tf_keras' `_clip_gradients` -- first active mode only -- that oracle.clip_grads restates from memory)
        model.compile(tfk.optimizers.Adam(1e-3, 0.9, 0.99, clipnorm=kw.get("clipnorm"), clipvalue=kw.get("clipvalue"),
                                          global_clipnorm=kw.get("global_clipnorm")), run_eagerly=True)

        noise["u"], noise["eta"] = z["u_f"], z["eta"]
        model(inputs)                                        # builds every variable
        # ---- golden parameters into the reference's variables (oracle order: a, b, (W, b) x (L+1), image scales,
        #      per-image (kernel, bias) x K, Ev11 raw) ------------------------------------------------------------------
        it = iter(params[2:])
        dense = (scaler.metadata_scaler if k_img else (scaler.mlp_scaler if use_img else scaler))
        dense_vars = list(dense.network.trainable_variables) + list(dense.distribution.trainable_variables)
        order = [q.distribution.loc.pretransformed_input, q.distribution.scale.pretransformed_input]
        q.distribution.loc.pretransformed_input.assign(params[0])
        q.distribution.scale.pretransformed_input.assign(params[1])
        for v in dense_vars:
            v.assign(next(it)); order.append(v)
        if use_img:
            scaler.image_scaler._scales.assign(next(it)); order.append(scaler.image_scaler._scales)
        if k_img:
            for layer in scaler.image_layers:
                layer.w.assign(next(it)); order.append(layer.w)
                layer.b.assign(next(it)); order.append(layer.b)
        if dw and prior.optimize_r:
            prior.r.pretransformed_input.assign(next(it)); order.append(prior.r.pretransformed_input)
        if kw.get("ev11"):
            ev = next(it)
            e11 = lik.mono if laue else lik
            for i, tv in enumerate((e11.Sdfac, e11.Sdadd, e11.SdB)):
                tv.pretransformed_input.assign(ev[i])
            order.append((e11.Sdfac.pretransformed_input, e11.Sdadd.pretransformed_input, e11.SdB.pretransformed_input))

        # ---- one forward / backward ------------------------------------------------------------------------------------
        flat_order = [v for o in order for v in (o if isinstance(o, tuple) else (o,))]
        with tf.GradientTape() as tape:
            model(inputs, training=True)
            loss = tf.add_n(model.losses)
        g = tape.gradient(loss, flat_order)
        got, k = [], 0
        for o in order:
            if isinstance(o, tuple):
                got.append(np.array([float(x) for x in g[k:k + len(o)]])); k += len(o)
            else:
                got.append(g[k].numpy()); k += 1
        metrics = {m.name: float(m.result()) for m in model.metrics}
        rows = [("loss", rel(float(loss), float(z["loss"]))), ("NLL", rel(metrics.get("NLL", np.nan), float(z["nll"]))),
                ("F KLDiv", rel(metrics.get("F KLDiv", np.nan), float(z["kl"])))]
        rows += [(f"grad_{i:02d} {tuple(a.shape)}", rel(a, b)) for i, (a, b) in enumerate(zip(got, grads))]

        # ---- six Adam steps ---------------------------------------------------------------------------------------------
        tl, tg = [], []
        for i in range(len(z["traj_loss"])):
            noise["u"], noise["eta"] = z["traj_u"][i], z["traj_eta"][i]
            model.reset_metrics()
            h = model.train_step_with_gradient_norm((inputs,))
            tl.append(float(h["loss"])); tg.append(float(h["Grad Norm"]))
        rows += [("traj loss", rel(tl, z["traj_loss"])), ("traj grad norm", rel(tg, z["traj_gnorm"]))]
        cur = []
        for o in order:
            cur.append(np.array([float(x) for x in o]) if isinstance(o, tuple) else o.numpy())
        rows += [(f"final_{i:02d}", rel(a, b)) for i, (a, b) in enumerate(zip(cur, finals))]
        finite = [r for r in rows if np.isfinite(r[1])]          # (logit(r = 0) = -inf for the root ASU: compared as equal infinities below)
        bad = [r for r in finite if not (r[1] <= args.rtol)]
        worst = max(worst, max(r[1] for r in finite))
        print(f"== {name}: {'OK' if not bad else 'MISMATCH'}  (max rel err {max(r[1] for r in finite):.2e})")
        for n_, e in rows:
            print(f"   {n_:28s} {e:.3e}{'' if (e <= args.rtol or not np.isfinite(e)) else '   <-- above tolerance'}")
        pick = lambda pre: max([e for n_, e in finite if n_.startswith(pre)] or [0.0])
        table.append((name, pick("loss"), max(pick("NLL"), pick("F KLDiv")), pick("grad_"), max(pick("traj"), pick("final_")), not bad))
    print()
    print(f"{'case':44s} {'loss':>9s} {'NLL/KL':>9s} {'gradients':>10s} {'trajectory':>11s}   verdict")
    for name, a, b, c, d, ok in table:
        print(f"{name:44s} {a:9.2e} {b:9.2e} {c:10.2e} {d:11.2e}   {'PASS' if ok else 'FAIL'}")
    print(f"worst relative error over all cases: {worst:.3e} (tolerance {args.rtol:g}) -> {'PASS' if worst <= args.rtol else 'FAIL'}")
    return 0 if worst <= args.rtol else 1


if __name__ == "__main__":
    sys.exit(main())
