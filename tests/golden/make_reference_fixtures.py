#!/usr/bin/env python3
"""Fixtures produced BY THE REFERENCE ITSELF (the parts of /root/reference that import without TensorFlow).

Most of the reference's ELBO path needs tensorflow / tensorflow-probability / tf_keras, which are not installable in the build
container, so the ELBO golden vectors come from the oracle (make_golden.py).  Two pieces on either side of the path are plain
Python / numpy and ARE executed here, loaded by file path (the package's own `__init__` asks setuptools for an installed
`careless` distribution):

  * `careless/utils/positional_encoding.py: positional_encoding`  -> reference/positional_encoding.npz (inputs + the reference's outputs)
  * `careless/args/*.py` (the argparse tables behind `careless mono|poly`) -> reference/cli_flags.json (flag -> dest/default/type/action/...)

The fixtures are data (inputs and expected outputs); nothing of the reference's source text is stored.  Run in the build container:

    python tests/golden/make_reference_fixtures.py [/root/reference]
"""
import argparse
import importlib.util
import json
import os
import sys

import numpy as np

sys.dont_write_bytecode = True          # the reference tree is read-only: no __pycache__ beside its sources
HERE = os.path.dirname(os.path.abspath(__file__))


def load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def positional_encoding_fixture(ref):
    pe = load(os.path.join(ref, "careless", "utils", "positional_encoding.py"), "ref_positional_encoding").positional_encoding
    rng = np.random.default_rng(20261003)
    out = {}
    cases = {
        "xy_f32_L4": (rng.uniform(0.0, 2048.0, size=(257, 2)).astype(np.float32), 4),        # --positional-encoding-keys X,Y (default L)
        "xy_f64_L5": (rng.uniform(-3.0, 7.0, size=(64, 2)), 5),
        "one_key_L1": (rng.normal(size=(33, 1)).astype(np.float32), 1),
        "three_keys_L6": (rng.uniform(0.0, 1.0, size=(19, 3)).astype(np.float32), 6),
        "constant_plus_ramp_L2": (np.stack([np.linspace(-5.0, 5.0, 21), np.linspace(10.0, 0.0, 21)], 1).astype(np.float32), 2),
    }
    for k, (x, L) in cases.items():
        out[f"{k}__x"] = x
        out[f"{k}__L"] = np.int64(L)
        out[f"{k}__y"] = pe(x, L)
    np.savez_compressed(os.path.join(HERE, "reference", "positional_encoding.npz"), **out)
    return len(cases)


def cli_flags_fixture(ref):
    """Feed the reference's (args, kwargs) tables to a plain ArgumentParser -- exactly what careless/parser.py does with them, minus
    the TensorFlow set-up of its parse_args -- and record what argparse made of every flag."""
    adir = os.path.join(ref, "careless", "args")
    groups = ["required", "poly", "common", "crossvalidation", "filtration", "interpretation", "likelihood", "optimizer",
              "positional_encoding", "prior", "scaling", "tf_options"]
    table = {}
    for g in groups:
        mod = load(os.path.join(adir, g + ".py"), "ref_args_" + g)
        p = argparse.ArgumentParser()
        for a, kw in mod.args_and_kwargs:
            p.add_argument(*a, **kw)
        entries = []
        for act in p._actions:
            if isinstance(act, argparse._HelpAction):
                continue
            t = act.type.__name__ if act.type is not None else None
            default = act.default
            if not isinstance(default, (type(None), bool, int, float, str, list)):
                default = repr(default)
            entries.append(dict(flags=list(act.option_strings), dest=act.dest, default=default, type=t,
                                action=type(act).__name__, nargs=act.nargs, choices=list(act.choices) if act.choices else None,
                                const=act.const if isinstance(act.const, (type(None), bool, int, float, str)) else repr(act.const)))
        table[g] = entries
    with open(os.path.join(HERE, "reference", "cli_flags.json"), "w") as f:
        json.dump(table, f, indent=1, sort_keys=True)
    return sum(len(v) for v in table.values())


if __name__ == "__main__":
    ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    print("positional_encoding cases:", positional_encoding_fixture(ref))
    print("cli flags:", cli_flags_fixture(ref))
