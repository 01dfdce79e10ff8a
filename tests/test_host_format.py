"""The formatter's native host entry points (careless_amd/csrc/host_format.cpp: cl_host_asu_map, cl_host_dense_ids) against their numpy
restatement (tests/ref_asu.py = the implementation the reference's MTZ fixtures had pinned): bit-exact, every space group of the built-in
table, both Friedel conventions, settings outside the CCP4 sets, any thread count.  Host code: runs without a GPU."""
import ctypes as C

import numpy as np
import pytest

from tests import ref_asu
from careless_amd._lib import get_lib
from careless_amd.io.asu import SymmetryOps
from careless_amd.io.formatter import _ngroup
from careless_amd.io.spacegroups import lookup


@pytest.mark.parametrize("anomalous", [False, True])
def test_asu_map_matches_numpy_for_every_space_group(anomalous):
    rng = np.random.default_rng(7)
    n_checked = 0
    for num in range(1, 231):
        try:
            symops, name, _ = lookup(str(num))
        except (NotImplementedError, KeyError, ValueError):
            continue
        ops = SymmetryOps(symops)
        h = rng.integers(-9, 10, size=(3000, 3))
        h[:50] = rng.integers(-2, 3, size=(50, 3))                      # axes, zones, the origin
        hasu, centric, eps, absent = ops.map_rows(h, anomalous)
        assert np.array_equal(hasu, ref_asu.to_asu(ops.R, ops.asu_case(), h, anomalous)), name
        c2, e2, a2 = ref_asu.describe(ops.R, ops.t, h)
        assert np.array_equal(centric, c2) and np.array_equal(eps, e2) and np.array_equal(absent, a2), name
        assert np.array_equal(ops.to_asu(h, anomalous), hasu)
        n_checked += 1
    assert n_checked == 65


def test_asu_map_without_a_ccp4_set_takes_the_largest_index():
    # monoclinic, unique axis c: none of the CCP4 sets (reference settings) is an asymmetric unit for these operators
    ops = SymmetryOps(["X, Y, Z", "-X, -Y, Z+1/2"])
    assert ops.asu_case() is None
    h = np.random.default_rng(3).integers(-7, 8, size=(5000, 3))
    for anomalous in (False, True):
        assert np.array_equal(ops.to_asu(h, anomalous), ref_asu.to_asu(ops.R, None, h, anomalous))
    c, e, a = ops.describe(h)
    c2, e2, a2 = ref_asu.describe(ops.R, ops.t, h)
    assert np.array_equal(c, c2) and np.array_equal(e, e2) and np.array_equal(a, a2)
    assert a[(h[:, 0] == 0) & (h[:, 1] == 0) & (h[:, 2] % 2 != 0)].all()          # 0 0 l, l odd: the screw axis' absences


def test_asu_map_is_independent_of_the_thread_count():
    symops, _, _ = lookup("96")
    ops = SymmetryOps(symops)
    h = np.ascontiguousarray(np.random.default_rng(5).integers(-40, 41, size=(200_000, 3)), dtype=np.int32)
    rot, trans = np.ascontiguousarray(ops.R, dtype=np.int32), np.ascontiguousarray(ops.t, dtype=np.float64)
    lib, p = get_lib(), lambda a: a.ctypes.data_as(C.c_void_p)
    outs = []
    for nthreads in (1, 3, 0):
        hasu, cen, eps, ab = np.empty((len(h), 3), np.int32), np.empty(len(h), np.uint8), np.empty(len(h), np.int32), np.empty(len(h), np.uint8)
        assert lib.cl_host_asu_map(p(h), len(h), p(rot), p(trans), len(rot), ops.asu_case(), 1, p(hasu), p(cen), p(eps), p(ab), nthreads) == 0
        outs.append((hasu, cen, eps, ab))
    for o in outs[1:]:
        assert all(np.array_equal(x, y) for x, y in zip(o, outs[0]))
    assert lib.cl_host_asu_map(p(h), len(h), p(rot), p(trans), 0, 0, 0, None, None, None, None, 0) == -1          # no operators
    assert lib.cl_host_asu_map(p(h), len(h), p(rot), None, len(rot), 0, 0, None, None, None, p(outs[0][3]), 0) == -1    # absences need translations


def test_ngroup_matches_sorted_unique_rows():
    rng = np.random.default_rng(11)
    a, b = rng.integers(0, 3, size=50_000), rng.integers(-500, 4000, size=50_000)                 # presence table (image ids)
    assert np.array_equal(_ngroup(a, b), ref_asu.ngroup(a, b))
    img, h0 = rng.integers(0, 20_000, size=50_000), rng.integers(-60, 61, size=(50_000, 3))        # 3.5e10 keys: sorted
    assert np.array_equal(_ngroup(img, h0[:, 0], h0[:, 1], h0[:, 2]), ref_asu.ngroup(img, h0[:, 0], h0[:, 1], h0[:, 2]))
    big = rng.integers(-(1 << 40), 1 << 40, size=(2000, 2))                                        # the folded key would overflow: row sort
    assert np.array_equal(_ngroup(big[:, 0], big[:, 1]), ref_asu.ngroup(big[:, 0], big[:, 1]))
    f = rng.integers(0, 9, size=3000).astype(np.float32)                                            # non-integer column: row sort
    assert np.array_equal(_ngroup(f, a[:3000]), ref_asu.ngroup(f, a[:3000]))
    assert len(_ngroup(np.zeros(0, dtype=np.int64))) == 0
    one = _ngroup(np.full(10, 7))
    assert np.array_equal(one, np.zeros(10, dtype=np.int64))


def test_dense_ids_rejects_keys_outside_the_range():
    lib, p = get_lib(), lambda a: a.ctypes.data_as(C.c_void_p)
    key, ids = np.array([0, 5, 2, 9], dtype=np.int64), np.empty(4, dtype=np.int64)
    ng = C.c_longlong(0)
    assert lib.cl_host_dense_ids(p(key), 4, 0, 9, p(ids), C.byref(ng), 1) == 0 and ng.value == 4 and ids.tolist() == [0, 2, 1, 3]
    assert lib.cl_host_dense_ids(p(key), 4, 0, 8, p(ids), None, 1) == -1
    assert lib.cl_host_dense_ids(p(key), 4, 0, 1 << 40, p(ids), None, 1) == -2


# ---- CrystFEL streams: the native parser against the line-by-line Python loop -----------------------------------------------------------
def _synthetic_stream(path, n_crystals=40, seed=0, crlf=False, odd=True):
    rng = np.random.default_rng(seed)
    nl = "\r\n" if crlf else "\n"
    out = ["CrystFEL stream format 2.3", "----- Begin unit cell -----", "a = 79.10 A", "b = 79.10 A", "c = 38.20 A", "al = 90.00 deg", "be = 90.00 deg",
           "ga = 90.00 deg", "----- End unit cell -----"]
    for c in range(n_crystals):
        out += ["----- Begin chunk -----", "Image filename: x.h5", "--- Begin crystal", "Cell parameters 7.9 7.9 3.8 nm, 90 90 90 deg"]
        if odd and c % 7 == 3:                       # a crystal without a reflection list
            out += ["--- End crystal", "----- End chunk -----"]
            continue
        out += ["num_reflections = 5", "Reflections measured after indexing", "   h    k    l          I   sigma(I)       peak background  fs/px  ss/px panel"]
        for _ in range(int(rng.integers(0, 60))):
            h, k, l = rng.integers(-40, 41, 3)
            out.append("%4d %4d %4d %10.2f %10.2f %10.2f %10.2f %6.1f %6.1f p0" % (h, k, l, rng.normal(50, 200), abs(rng.normal(20, 5)), rng.uniform(0, 99),
                                                                                    rng.uniform(0, 30), rng.uniform(0, 1440), rng.uniform(0, 1440)))
        if odd and c % 5 == 1:
            out += ["  1   2   3  4.0", "", "   \t ", "+7 -8 +9 1e3 2.5E-1 nan inf 0.5 .25 q1 extra fields here"]       # short lines are skipped; a long one is a row
        out += ["End of reflections", "--- End crystal"]
        if odd and c % 9 == 4:                       # a second lattice in the same chunk
            out += ["--- Begin crystal", "Reflections measured after indexing", "   h    k    l  ...", " 1 1 1 5.0 1.0 2.0 3.0 4.0 5.0 p1", "End of reflections", "--- End crystal"]
        out += ["----- End chunk -----"]
    with open(path, "w", newline="") as f:
        f.write(nl.join(out) + (nl if seed % 2 == 0 else ""))


def _same_table(a, b):
    assert a.keys() == b.keys() and a.cell == b.cell and a.types == b.types
    for k in a.columns:
        assert a.columns[k].dtype == np.float32 and np.array_equal(a.columns[k], b.columns[k], equal_nan=True), k


def test_crystfel_parser_matches_the_python_loop(tmp_path):
    import os
    from careless_amd.io.crystfel import read_crystfel
    from tests import ref_crystfel
    fixture = os.path.join(os.path.dirname(__file__), "golden", "crystfel.stream")
    _same_table(read_crystfel(fixture), ref_crystfel.read_crystfel(fixture))
    for seed, crlf in ((0, False), (1, False), (2, True), (3, True)):
        p = str(tmp_path / f"s{seed}.stream")
        _synthetic_stream(p, n_crystals=60, seed=seed, crlf=crlf)
        got, ref = read_crystfel(p), ref_crystfel.read_crystfel(p)
        _same_table(got, ref)
        assert len(got) > 500 and got.columns["BATCH"].max() >= 59


def test_crystfel_parser_thread_counts_and_errors(tmp_path):
    from careless_amd.io.crystfel import COLUMNS, read_crystfel
    p = str(tmp_path / "t.stream")
    _synthetic_stream(p, n_crystals=300, seed=4, odd=False)
    buf = np.fromfile(p, dtype=np.uint8)
    lib, ptr = get_lib(), lambda a: a.ctypes.data_as(C.c_void_p)
    nc = C.c_longlong(0)
    n = lib.cl_host_crystfel_count(ptr(buf), buf.size, C.byref(nc), 0)
    assert n > 5000 and nc.value == 300
    tabs = []
    for nthreads in (1, 3, 0):
        t = np.empty((len(COLUMNS), n), dtype=np.float32)
        assert lib.cl_host_crystfel_parse(ptr(buf), buf.size, n, ptr(t), nthreads) == 0
        tabs.append(t)
    assert all(np.array_equal(t, tabs[0]) for t in tabs[1:])
    assert lib.cl_host_crystfel_parse(ptr(buf), buf.size, n + 1, ptr(tabs[0]), 0) == -1                  # a table of another size
    bad = str(tmp_path / "bad.stream")
    with open(p) as f, open(bad, "w") as g:
        g.write(f.read().replace("End of reflections", " 1 2 3 4.0 5.0 x6 7.0 8.0 9.0 p0\nEnd of reflections", 1))
    with pytest.raises(ValueError, match="not a number"):
        read_crystfel(bad)
    empty = str(tmp_path / "none.stream")
    with open(empty, "w") as g:
        g.write("CrystFEL stream format 2.3\n----- Begin unit cell -----\na = 1 A\n----- End unit cell -----\n")
    with pytest.raises(ValueError, match="no indexed reflections"):
        read_crystfel(empty)
