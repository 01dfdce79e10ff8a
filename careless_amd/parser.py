"""Command line of `careless_amd` -- the flags of `careless mono` / `careless poly` that reach the ELBO path, with the reference's
spellings, destinations and defaults (reference careless/parser.py, careless/args/*.py).  TensorFlow-only switches
(--run-eagerly, --jit-compile, --reduce-retracing, --disable-gpu, --gpu-id, --disable-memory-growth, --tf-debug) are accepted
and ignored so existing command lines keep working."""
from __future__ import annotations

import argparse

from careless_amd.manager import default_args


def _common(p: argparse.ArgumentParser):
    d = default_args()
    p.add_argument("metadata_keys", type=str, help="comma separated metadata column names, e.g. dHKL,Hobs,Kobs,Lobs,BATCH")
    p.add_argument("reflection_files", type=str, nargs="+", metavar="reflections.{mtz,npz}")
    p.add_argument("output_base", type=str, metavar="out")
    # args/common.py
    p.add_argument("--mc-samples", type=int, default=d.mc_samples)
    p.add_argument("--structure-factor-file", type=str, default=None)
    p.add_argument("--freeze-structure-factors", action="store_true")
    p.add_argument("--structure-factor-init-scale", type=float, default=d.structure_factor_init_scale)
    p.add_argument("--epsilon", type=float, default=d.epsilon)
    p.add_argument("--disable-metadata-standardization", action="store_false", dest="standardize_metadata")
    p.add_argument("--disable-progress-bar", action="store_true", default=False)
    # args/filtration.py, interpretation.py
    p.add_argument("-c", "--isigi-cutoff", type=float, default=None)
    p.add_argument("-d", "--dmin", type=float, default=None)
    p.add_argument("--spacegroups", type=str, default=None)
    p.add_argument("--image-key", type=str, default=None)
    p.add_argument("--intensity-key", type=str, default=None)
    p.add_argument("--uncertainty-key", type=str, default=None)
    p.add_argument("--anomalous", action="store_true", default=False)
    p.add_argument("--separate-files", action="store_true", default=False)
    # args/likelihood.py
    p.add_argument("--studentt-likelihood-dof", type=float, metavar="DOF", default=None)
    p.add_argument("--refine-uncertainties", action="store_true", default=False)
    # args/optimizer.py
    p.add_argument("--iterations", type=int, default=d.iterations)
    p.add_argument("--learning-rate", type=float, default=d.learning_rate)
    p.add_argument("--beta-1", type=float, default=d.beta_1)
    p.add_argument("--beta-2", type=float, default=d.beta_2)
    p.add_argument("--clipnorm", type=float, default=None)
    p.add_argument("--clipvalue", type=float, default=None)
    p.add_argument("--global-clipnorm", type=float, default=None)
    # args/positional_encoding.py
    p.add_argument("--positional-encoding-keys", type=str, default=None)
    p.add_argument("--positional-encoding-frequencies", "-L", type=int, default=4)
    # args/prior.py
    p.add_argument("--kl-weight", type=float, default=None)
    p.add_argument("--wilson-prior-b", type=float, default=None)
    p.add_argument("--double-wilson-r", type=str, default=None, dest="dwr")
    p.add_argument("--double-wilson-parents", type=str, default=None, dest="parents")
    p.add_argument("--double-wilson-reindexing-ops", type=str, default=None, dest="reindexing_ops")
    p.add_argument("--optimize-double-wilson-r", action="store_true")
    # args/scaling.py
    p.add_argument("--scale-file", type=str, default=None)
    p.add_argument("--freeze-scales", action="store_true")
    p.add_argument("--mlp-layers", type=int, default=d.mlp_layers)
    p.add_argument("--mlp-width", type=int, default=d.mlp_width)
    p.add_argument("--image-layers", type=int, default=d.image_layers)
    p.add_argument("--disable-image-scales", action="store_false", dest="use_image_scales", default=True)
    p.add_argument("--scale-bijector", type=str, default=d.scale_bijector, choices=["exp", "softplus"])
    # args/crossvalidation.py
    p.add_argument("--test-fraction", type=float, default=None)
    p.add_argument("--merge-half-datasets", action="store_true", default=False)
    p.add_argument("--half-dataset-repeats", type=int, default=1)
    p.add_argument("--validation-frequency", type=int, default=10)
    # args/tf_options.py
    p.add_argument("--seed", type=int, default=1234)
    for flag in ("--run-eagerly", "--jit-compile", "--reduce-retracing", "--disable-gpu", "--disable-memory-growth", "--tf-debug",
                 "--embed", "--save-data-manager"):
        p.add_argument(flag, action="store_true", default=False)
    p.add_argument("--gpu-id", type=int, default=0)


def make_parser() -> argparse.ArgumentParser:
    parser = argparse.ArgumentParser(prog="careless_amd", description="Scale and merge crystallographic data by approximate inference "
                                                                      "(MI355X engine)")
    sub = parser.add_subparsers(dest="type", required=True)
    mono = sub.add_parser("mono", help="monochromatic data")
    _common(mono)
    poly = sub.add_parser("poly", help="polychromatic (Laue) data")
    _common(poly)
    poly.add_argument("-l", "--wavelength-range", type=float, default=None, nargs=2, metavar=("lambda_min", "lambda_max"))
    poly.add_argument("-w", "--wavelength-key", type=str, default="Wavelength")
    return parser


parser = make_parser()
