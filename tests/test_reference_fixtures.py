"""Fixtures the REFERENCE produced (tests/golden/make_reference_fixtures.py ran the TensorFlow-free parts of /root/reference in the
build container): the positional encoding that builds the metadata of BASELINE configs[2], and the argparse tables behind
`careless mono|poly`.  The repository's oracle, host code and command line are held against them."""
import json
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _pe_cases():
    z = np.load(os.path.join(GOLD, "reference", "positional_encoding.npz"))
    names = sorted({k.split("__")[0] for k in z.files})
    return [(n, z[f"{n}__x"], int(z[f"{n}__L"]), z[f"{n}__y"]) for n in names]


@pytest.mark.parametrize("name,x,L,y", _pe_cases(), ids=[c[0] for c in _pe_cases()])
def test_positional_encoding_equals_the_reference_output(name, x, L, y):
    """careless/utils/positional_encoding.py:3-17, executed by the generator: oracle restatement and product host code."""
    from oracle import elbo_oracle as O
    from careless_amd.synthetic import positional_encoding
    tol = dict(rtol=0.0, atol=2e-6 * 2 ** L) if x.dtype == np.float32 else dict(rtol=0.0, atol=1e-12)   # sin / cos of 2^L pi p at the input's precision
    for f in (O.positional_encoding, positional_encoding):
        got = np.asarray(f(x, L))
        assert got.shape == y.shape, (f.__module__, got.shape, y.shape)
        assert np.allclose(got, y, **tol), (f.__module__, np.abs(got - y).max())


def _ref_flags():
    return json.load(open(os.path.join(GOLD, "reference", "cli_flags.json")))


def _our_actions(sub):
    import argparse
    from careless_amd.parser import make_parser
    p = make_parser()
    subs = [a for a in p._actions if isinstance(a, argparse._SubParsersAction)][0]
    acts = {}
    for a in subs.choices[sub]._actions:
        if isinstance(a, argparse._HelpAction):
            continue
        for k in (a.option_strings or [a.dest]):
            acts[k] = a
    return acts


@pytest.mark.parametrize("sub", ["mono", "poly"])
def test_command_line_flags_match_the_reference_tables(sub):
    """Every flag of careless/args/*.py exists on `careless_amd mono|poly` with the same spellings, destination, default, type,
    action, nargs and choices (reference parser.py builds its sub-parsers from exactly these tables)."""
    ref, ours = _ref_flags(), _our_actions(sub)
    groups = [g for g in ref if g != "poly" or sub == "poly"]
    checked = 0
    for g in groups:
        for e in ref[g]:
            keys = e["flags"] or [e["dest"]]
            for k in keys:
                assert k in ours, f"{sub}: reference flag {k} ({g}) is missing"
            a = ours[keys[0]]
            assert all(ours[k] is a for k in keys), f"{keys} are not aliases of one option"
            assert a.dest == e["dest"], (keys, a.dest, e["dest"])
            assert type(a).__name__ == e["action"], (keys, type(a).__name__, e["action"])
            assert a.nargs == e["nargs"], (keys, a.nargs, e["nargs"])
            assert (list(a.choices) if a.choices else None) == e["choices"], keys
            if e["type"] is not None:
                assert a.type is not None and a.type.__name__ == e["type"], (keys, a.type, e["type"])
            if e["flags"]:                                            # positionals have no default
                d = a.default
                if e["dest"] == "jit_compile":                        # TensorFlow-only switch, accepted and ignored: None or False
                    assert not d
                else:
                    assert d == e["default"] and type(d) is type(e["default"]), (keys, d, e["default"])
            checked += 1
    assert checked >= 55
    if sub == "mono":                                                 # the Laue-only flags stay off the mono sub-command
        for e in ref["poly"]:
            assert not any(k in ours for k in e["flags"])


def test_parsed_defaults_equal_the_reference_defaults():
    """An argv with only the positionals parses to the reference's defaults (what DataManager.from_parser / build_model read)."""
    from careless_amd.parser import make_parser
    ns = make_parser().parse_args(["mono", "dHKL", "a.mtz", "out"])
    ref = _ref_flags()
    for g, entries in ref.items():
        if g in ("required", "poly"):
            continue
        for e in entries:
            if e["dest"] == "jit_compile":
                continue
            assert getattr(ns, e["dest"]) == e["default"], (e["dest"], getattr(ns, e["dest"]), e["default"])
