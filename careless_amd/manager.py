"""Model assembly and data splitting around the ELBO path: the array-level counterpart of `careless.io.manager.DataManager`.

What is mirrored (reference careless/io/manager.py):
  * `build_model` (:380-507) -- the wiring of a parsed flag Namespace into prior, surrogate posterior, likelihood, scaling model and
    Adam: same flag names (reference careless/args/*.py), same defaults (`default_args`), same error behaviour
    (`ValueError` for a double-Wilson r outside (-1, 1), for an unknown scale bijector);
  * the cross-validation splits `split_data_by_refl` (:273-297), `split_laue_data_by_mask` (:299-343), `split_data_by_image`
    (:345-377), including the Laue re-packing of `harmonic_id` and the padded intensity slots;
  * `get_results` / `get_predictions` (:89-250) at array level (see careless_amd/results.py).
What is NOT: reading reflection files, the reciprocal-ASU bookkeeping (gemmi / reciprocalspaceship), MTZ output -- the caller
supplies `inputs` in `BaseModel.input_index` order plus the per-reflection arrays an ASU collection provides
(`centric`, `multiplicity`, optionally `dHKL` for `--wilson-prior-b` and the double-Wilson lookup arrays).
"""
from __future__ import annotations

from argparse import Namespace
from typing import Optional

import numpy as np

from careless_amd.models.base import BaseModel


def default_args(**overrides) -> Namespace:
    """The reference CLI's defaults for every flag `build_model` / `train_model` consume (careless/args/*.py)."""
    ns = Namespace(
        type="mono",
        mc_samples=1, structure_factor_init_scale=1.0, epsilon=1e-7,                         # args/common.py
        mlp_layers=20, mlp_width=10, image_layers=0, use_image_scales=True, scale_bijector="exp",   # args/scaling.py
        iterations=10_000, learning_rate=1e-3, beta_1=0.9, beta_2=0.99,                      # args/optimizer.py
        clipnorm=None, clipvalue=None, global_clipnorm=None,
        studentt_likelihood_dof=None, refine_uncertainties=False,                            # args/likelihood.py
        kl_weight=None, wilson_prior_b=None, parents=None, dwr=None, reindexing_ops=None,    # args/prior.py
        optimize_double_wilson_r=False,
        test_fraction=None, merge_half_datasets=False, half_dataset_repeats=1, validation_frequency=10,   # args/crossvalidation.py
        freeze_structure_factors=False, freeze_scales=False, disable_progress_bar=False,
        run_eagerly=False, seed=1234,                                                        # args/tf_options.py
    )
    for k, v in overrides.items():
        if not hasattr(ns, k):
            raise ValueError(f"unknown careless flag {k!r}")
        setattr(ns, k, v)
    return ns


class DataManager:
    """Data manipulation methods plus model construction (reference `DataManager`, array inputs)."""

    def __init__(self, inputs, centric, multiplicity=None, parser: Optional[Namespace] = None, dHKL=None, double_wilson=None):
        """
        inputs        : tuple in BaseModel.input_index order (numpy arrays, reference shapes / dtypes)
        centric       : (R,) bool       -- `asu_collection.centric`; or a `careless_amd.io.asu.ReciprocalASUCollection`, which then
                        also supplies multiplicity, dHKL and the double-Wilson parent lookup (the reference's call signature
                        `DataManager(inputs, asu_collection, parser=parser)`, careless/careless.py:39)
        multiplicity  : (R,) float      -- `asu_collection.multiplicity`
        dHKL          : (R,) float, resolution of every reflection, only needed with --wilson-prior-b (manager.py:43-52)
        double_wilson : dict(reflids=, root=, asu_ids=) -- the parent lookup the reference derives with gemmi (priors/wilson.py:112-138)
        """
        self.inputs = tuple(inputs)
        self.asu_collection = None
        if hasattr(centric, "reciprocal_asus"):
            self.asu_collection = rac = centric
            centric, multiplicity = rac.centric, rac.multiplicity
            dHKL = rac.dHKL if dHKL is None else dHKL
        self.centric = np.asarray(centric, dtype=bool)
        self.multiplicity = np.asarray(multiplicity, dtype=np.float32)
        self.parser = parser
        self.dHKL = None if dHKL is None else np.asarray(dHKL, dtype=np.float64)
        self.double_wilson = double_wilson

    # -- priors ----------------------------------------------------------------------------------------------------------
    @staticmethod
    def wilson_sigma(b, dHKL):
        return np.exp(-0.25 * b * np.reciprocal(dHKL * dHKL))                       # manager.py:43-46

    def get_wilson_sigma(self, b=None):
        if b is None:
            return 1.0
        if self.dHKL is None:
            raise ValueError("--wilson-prior-b needs the resolution dHKL of every reflection")
        return self.wilson_sigma(b, self.dHKL)

    def get_wilson_prior(self, b=None, k=1.0):
        from careless_amd.models.priors.wilson import WilsonPrior
        if b is None:
            sigma = 1.0
        elif isinstance(b, float):
            sigma = self.get_wilson_sigma(b)
        else:
            raise ValueError(f"parameter b has type{type(b)} but float was expected")   # manager.py:60-61
        return WilsonPrior(self.centric, self.multiplicity, sigma * k)

    # -- cross-validation splits -------------------------------------------------------------------------------------------
    def split_mono_data_by_mask(self, test_idx):
        test_idx = np.asarray(test_idx).reshape(-1)
        return tuple(a[~test_idx, ...] for a in self.inputs), tuple(a[test_idx, ...] for a in self.inputs)

    def split_laue_data_by_mask(self, test_idx):
        """Split Laue data; `harmonic_id` is re-packed and the intensity slots re-padded (manager.py:299-343)."""
        harmonic_id = BaseModel.get_harmonic_id(self.inputs)
        test_idx = np.asarray(test_idx).reshape(harmonic_id.shape)
        isect = np.intersect1d(harmonic_id[test_idx].flatten(), harmonic_id[~test_idx].flatten())
        if len(isect) > 0:
            raise ValueError(f"test_idx splits harmonic observations with harmonic_id : {isect}")

        def split(inputs, idx):
            hid = BaseModel.get_harmonic_id(inputs)
            uni, inv = np.unique(hid[idx], return_inverse=True)
            out = ()
            for i, v in enumerate(inputs):
                name = BaseModel.get_name_by_index(i)
                if name in ("intensities", "uncertainties"):
                    v = v[uni]
                    v = np.pad(v, [[0, len(inv) - len(v)], [0, 0]], constant_values=1.0)
                elif name == "harmonic_id":
                    v = inv.reshape(-1)[:, None]
                else:
                    v = v[idx.flatten(), ...]
                out += (v,)
            return out

        return split(self.inputs, ~test_idx), split(self.inputs, test_idx)

    def split_data_by_refl(self, test_fraction=0.5):
        if BaseModel.is_laue(self.inputs):
            harmonic_id = BaseModel.get_harmonic_id(self.inputs)
            test_idx = (np.random.random(harmonic_id.max() + 1) <= test_fraction)[harmonic_id]
            return self.split_laue_data_by_mask(test_idx)
        test_idx = np.random.random(len(self.inputs[0])) <= test_fraction
        return self.split_mono_data_by_mask(test_idx)

    def split_data_by_image(self, test_fraction=0.5):
        image_id = BaseModel.get_image_id(self.inputs)
        test_idx = np.random.random(image_id.max() + 1) <= test_fraction
        if True not in test_idx:                       # low image count edge case (manager.py:363-367)
            test_idx[0] = True
        elif False not in test_idx:
            test_idx[0] = False
        test_idx = test_idx[image_id]
        if BaseModel.is_laue(self.inputs):
            return self.split_laue_data_by_mask(test_idx)
        return self.split_mono_data_by_mask(test_idx)

    # -- results -----------------------------------------------------------------------------------------------------------
    def get_results(self, surrogate_posterior, inputs=None, output_parameters=True, max_intensity_snr=1e-5):
        from careless_amd.results import get_results
        return get_results(surrogate_posterior, self.inputs if inputs is None else inputs, output_parameters, max_intensity_snr)

    def get_predictions(self, model, inputs=None):
        from careless_amd.results import get_predictions
        return get_predictions(model, self.inputs if inputs is None else inputs)

    # -- model assembly ----------------------------------------------------------------------------------------------------
    def build_model(self, parser=None, surrogate_posterior=None, prior=None, likelihood=None, scaling_model=None,
                    mc_sample_size=None):
        """Build the model specified in `parser` (reference manager.py:380-507); any component may be overridden."""
        from careless_amd.models.merging.surrogate_posteriors import TruncatedNormal
        from careless_amd.models.merging.variational import VariationalMergingModel
        from careless_amd.models.priors.wilson import DoubleWilsonPrior
        from careless_amd.models.scaling.image import HybridImageScaler, ImageScaler
        from careless_amd.models.scaling.nn import MLPScaler
        from careless_amd.optimizers import Adam

        parser = self.parser if parser is None else parser
        if parser is None:
            raise ValueError("No parser supplied, but self.parser is unset")

        if parser.type == "poly":
            from careless_amd.models.likelihoods import laue as lik_mod
        elif parser.type == "mono":
            from careless_amd.models.likelihoods import mono as lik_mod
        else:
            raise ValueError(f"unknown experiment type {parser.type}")
        if parser.refine_uncertainties:
            Normal, StudentT = lik_mod.NormalEv11Likelihood, lik_mod.StudentTEv11Likelihood
        else:
            Normal, StudentT = lik_mod.NormalLikelihood, lik_mod.StudentTLikelihood

        parents, r_values = parser.parents, parser.dwr
        if prior is None and parents is None:
            prior = self.get_wilson_prior(parser.wilson_prior_b)
        elif prior is None:
            parents = [None if i == "None" else int(i) for i in parents.split(",")]
            r_values = [float(i) for i in r_values.split(",")]
            for r in r_values:
                if (r >= 1.0) or (r <= -1.0):
                    raise ValueError(f"Supplied --double-wilson-r value {r} outside of allowed range (-1, 1)")
                if r < 0:
                    from warnings import warn
                    warn(f"Supplied --double-wilson-r value {r} is negative")
            if self.double_wilson is None and self.asu_collection is not None:
                self.double_wilson = double_wilson_lookup(self.asu_collection, parents, parser.reindexing_ops)
            if self.double_wilson is None:
                raise ValueError("the double-Wilson prior needs the parent lookup arrays (reflids, root, asu_ids)")
            dw = self.double_wilson
            prior = DoubleWilsonPrior(self.centric, self.multiplicity, dw["reflids"], dw["root"], dw["asu_ids"], r_values,
                                      parents=parents, sigma=self.get_wilson_sigma(parser.wilson_prior_b),
                                      optimize_r=parser.optimize_double_wilson_r)

        loc, scale = prior.mean(), prior.stddev()
        scale = scale * parser.structure_factor_init_scale
        low = (1e-32 * ~self.centric).astype("float32")
        if surrogate_posterior is None:
            surrogate_posterior = TruncatedNormal.from_loc_and_scale(loc, scale, low, scale_shift=parser.epsilon)

        if likelihood is None:
            dof = parser.studentt_likelihood_dof
            likelihood = Normal() if dof is None else StudentT(dof)

        if scaling_model is None:
            mlp_width = parser.mlp_width
            if mlp_width is None:
                mlp_width = BaseModel.get_metadata(self.inputs).shape[-1]
            bij = parser.scale_bijector.lower()
            if bij == "softplus":
                istd = float(np.asarray(BaseModel.get_intensities(self.inputs)).std())
            elif bij == "exp":
                istd = None
            else:
                raise ValueError(f"Unsupported scale bijector type, {parser.scale_bijector}")
            if parser.image_layers > 0:                     # manager.py:467-478
                from careless_amd.models.scaling.image import NeuralImageScaler
                n_images = int(np.max(BaseModel.get_image_id(self.inputs))) + 1
                scaling_model = NeuralImageScaler(parser.image_layers, n_images, parser.mlp_layers, mlp_width,
                                                  epsilon=parser.epsilon, scale_bijector=bij, scale_multiplier=istd)
            elif parser.use_image_scales:
                mlp_scaler = MLPScaler(parser.mlp_layers, mlp_width, epsilon=parser.epsilon, scale_bijector=bij, scale_multiplier=istd)
                n_images = int(np.max(BaseModel.get_image_id(self.inputs))) + 1
                scaling_model = HybridImageScaler(mlp_scaler, ImageScaler(n_images))
            else:
                scaling_model = MLPScaler(parser.mlp_layers, mlp_width, epsilon=parser.epsilon, scale_bijector=bij, scale_multiplier=istd)

        model = VariationalMergingModel(surrogate_posterior, prior, likelihood, scaling_model,
                                        parser.mc_samples if mc_sample_size is None else mc_sample_size, kl_weight=parser.kl_weight)
        model.seed = getattr(parser, "seed", 1234)
        model.compile(Adam(parser.learning_rate, parser.beta_1, parser.beta_2, clipnorm=parser.clipnorm,
                           clipvalue=parser.clipvalue, global_clipnorm=parser.global_clipnorm),
                      run_eagerly=getattr(parser, "run_eagerly", False))
        return model


def double_wilson_lookup(rac, parents, reindexing_ops=None):
    """reflids / root / asu_ids of `DoubleWilsonPrior` from an ASU collection (reference priors/wilson.py:112-138): every reflection
    of a child ASU is re-indexed into the parent's setting (`--double-wilson-reindexing-ops`, ';'-separated, 'x,y,z' = identity),
    mapped to the parent's ASU and looked up there; -1 = no such parent reflection (absent or beyond the resolution limit)."""
    from careless_amd.io.asu import parse_symop
    ops = None
    if reindexing_ops is not None:
        ops = [parse_symop(o)[0] for o in (reindexing_ops.split(";") if isinstance(reindexing_ops, str) else reindexing_ops)]
    reflids, root = [], []
    for child, parent in enumerate(parents):
        casu = rac.reciprocal_asus[child]
        if parent is None:
            reflids.append(np.arange(len(casu), dtype=np.int64))           # as the reference: the ASU-local ids, unused for roots
            root.append(np.ones(len(casu), dtype=bool))
            continue
        root.append(np.zeros(len(casu), dtype=bool))
        pasu = rac.reciprocal_asus[parent]
        h = casu.Hall
        if ops is not None:
            h = h @ ops[child]                                             # rs.utils.apply_to_hkl: h' = h R
        h = pasu.ops.to_asu(h, pasu.anomalous)
        ids = np.full(len(h), -1, dtype=np.int64)
        from careless_amd.io.asu import _key
        k = _key(h)
        pos = np.clip(np.searchsorted(pasu._keys, k), 0, len(pasu._keys) - 1)
        hit = pasu._keys[pos] == k
        ids[hit] = pasu._sort[pos[hit]] + rac.offsets[parent]
        reflids.append(ids)
    return dict(reflids=np.concatenate(reflids), root=np.concatenate(root), asu_ids=rac.asu_ids)


def merge_half_datasets(dm: DataManager, parser: Namespace, scaling_model, iterations: int, repeats: int = 1, progress=False):
    """`--merge-half-datasets` (reference careless/careless.py:102-128): for every repeat split the images in two halves, train a
    model with the (already fitted) scaling model FROZEN on each half and return the merged results of every half.
    Returns a list of (repeat, half, results-dict)."""
    scaling_model.trainable = False
    out = []
    for repeat in range(repeats):
        halves = dm.split_data_by_image(0.5)
        for half_id, half in enumerate(halves):
            model = dm.build_model(parser, scaling_model=scaling_model)
            model.train_model(half, iterations, progress=progress)
            res = dm.get_results(model.surrogate_posterior, inputs=half)
            out.append((repeat, half_id, res))
    return out
