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
