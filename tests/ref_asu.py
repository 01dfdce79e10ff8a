"""numpy restatement of the symmetry bookkeeping that `cl_host_asu_map` / `cl_host_dense_ids` (careless_amd/csrc/host_format.cpp) do
natively -- the checker of tests/test_host_format.py; test infrastructure only.  It is the formatter's former implementation
(rounds 1-5), which tests/test_io.py and tests/test_reference_fixtures.py had pinned against the reference's MTZ fixtures: h' = h R for
every operator, the first orbit member inside the CCP4 inequality set (careless_amd/io/asu.py: _CCP4_ASU), centric / epsilon / absent
by comparing the orbit with +-h (reference: DataSet.hkl_to_asu / remove_absences, careless/io/formatter.py:285-302, 319)."""
import numpy as np

from careless_amd.io.asu import _CCP4_ASU, _key


def orbit(R, hkl):
    return np.einsum("ni,oij->onj", np.asarray(hkl, dtype=np.int64), R)


def to_asu(R, case, hkl, anomalous=False):
    hkl = np.asarray(hkl, dtype=np.int64)
    rot = orbit(R, hkl)
    orb = np.concatenate([rot, -rot], axis=0)
    if case is None:
        best = np.argmax(_key(orb), axis=0)
    else:
        best = np.argmax(_CCP4_ASU[case](orb[..., 0], orb[..., 1], orb[..., 2]), axis=0)
    rep = np.take_along_axis(orb, best[None, :, None], axis=0)[0]
    if anomalous:
        minus = ~np.any(np.all(rot == rep[None], axis=2), axis=0)
        rep = np.where(minus[:, None], -rep, rep)
    return rep


def describe(R, t, hkl):
    h = np.asarray(hkl, dtype=np.int64)
    orb = orbit(R, h)
    same = np.all(orb == h[None], axis=2)
    eps = same.sum(0)
    phase = np.einsum("ni,oi->on", h.astype(np.float64), t)
    absent = np.any(same & (np.abs(phase - np.round(phase)) > 1e-6), axis=0)
    centric = np.any(np.all(orb == -h[None], axis=2), axis=0)
    return centric, eps, absent


def ngroup(*cols):
    keys = np.stack([np.asarray(c) for c in cols], axis=1)
    _, inv = np.unique(keys, axis=0, return_inverse=True)
    return inv.reshape(-1).astype(np.int64)
