"""Reciprocal-space bookkeeping from symmetry operators: resolution, canonical asymmetric-unit representative, centric flag,
multiplicity epsilon and systematic absences.  Stands in for what the reference gets from gemmi / reciprocalspaceship in
`careless/io/asu.py:5-143` (`ReciprocalASU`, `ReciprocalASUCollection`) and `careless/io/formatter.py:285-302`.

The operators are the `SYMM` records of the MTZ header (all of them, centring translations included).  The ASU representative of
a reflection follows the CCP4 convention that gemmi / reciprocalspaceship use (`DataSet.hkl_to_asu`, reference
`io/formatter.py:319,562`): one inequality set per Laue class in its reference setting (`_CCP4_ASU` below).  Which set applies is
not looked up by space-group number but found by test: the set must select exactly one member of every orbit of a small index
grid under the operators at hand.  For a setting none of the sets fits (e.g. monoclinic with unique axis c, where gemmi changes
basis first) the representative is the lexicographically largest index of the orbit: a valid asymmetric unit, but then merged
reflections may be listed under a symmetry mate of the index the reference prints.  Anomalous: an acentric reflection that reaches
the ASU only through the inversion is a Friedel-minus and keeps the NEGATED representative, as `hkl_to_asu(anomalous=True)` does."""
from __future__ import annotations

import re
from typing import List, Sequence

import numpy as np


def parse_symop(s: str):
    """'X-Y, X, Z+1/2' -> (3x3 integer rotation acting on fractional coordinates, translation)."""
    R = np.zeros((3, 3), dtype=np.int64)
    t = np.zeros(3)
    for i, part in enumerate(s.replace(" ", "").upper().split(",")):
        for sign, num, den, ax in re.findall(r"([+-]?)(?:(\d+)/(\d+)|([XYZ]))", part):
            sg = -1 if sign == "-" else 1
            if ax:
                R[i, "XYZ".index(ax)] += sg
            else:
                t[i] += sg * int(num) / int(den)
    return R, t


def reciprocal_metric(cell) -> np.ndarray:
    a, b, c, al, be, ga = [float(v) for v in cell]
    al, be, ga = np.deg2rad([al, be, ga])
    G = np.array([[a * a, a * b * np.cos(ga), a * c * np.cos(be)],
                  [a * b * np.cos(ga), b * b, b * c * np.cos(al)],
                  [a * c * np.cos(be), b * c * np.cos(al), c * c]])
    return np.linalg.inv(G)


def inv_d2(hkl: np.ndarray, cell) -> np.ndarray:
    """1 / d^2 of every Miller index (any crystal system)."""
    h = np.asarray(hkl, dtype=np.float64)
    return np.einsum("ni,ij,nj->n", h, reciprocal_metric(cell), h)


def _key(h: np.ndarray) -> np.ndarray:
    B = 1 << 20
    return (h[..., 0] + B // 2) * B * B + (h[..., 1] + B // 2) * B + (h[..., 2] + B // 2)


# CCP4 reciprocal asymmetric units of the Laue classes in their reference settings (the `asuset` conventions; gemmi's
# `ReciprocalAsu::is_in` evaluates the same conditions).  Order: -1, 2/m (unique b), mmm, 4/m and 6/m, 4/mmm and 6/mmm, -3,
# -31m, -3m1, m-3, m-3m.
_CCP4_ASU = (
    lambda h, k, l: (l > 0) | ((l == 0) & ((h > 0) | ((h == 0) & (k >= 0)))),
    lambda h, k, l: (k >= 0) & ((l > 0) | ((l == 0) & (h >= 0))),
    lambda h, k, l: (h >= 0) & (k >= 0) & (l >= 0),
    lambda h, k, l: (l >= 0) & (((h >= 0) & (k > 0)) | ((h == 0) & (k == 0))),
    lambda h, k, l: (h >= k) & (k >= 0) & (l >= 0),
    lambda h, k, l: ((h >= 0) & (k > 0)) | ((h == 0) & (k == 0) & (l >= 0)),
    lambda h, k, l: (h >= k) & (k >= 0) & ((k > 0) | (l >= 0)),
    lambda h, k, l: (h >= k) & (k >= 0) & ((h > k) | (l >= 0)),
    lambda h, k, l: (h >= 0) & (((l >= h) & (k > h)) | ((l == h) & (k == h))),
    lambda h, k, l: (k >= l) & (l >= h) & (h >= 0),
)
_UNSET = object()


class SymmetryOps:
    def __init__(self, symops: Sequence[str]):
        self._case = _UNSET
        parsed = [parse_symop(s) for s in symops]
        self.R = np.stack([p[0] for p in parsed])         # (nops, 3, 3)
        self.t = np.stack([p[1] for p in parsed])         # (nops, 3)

    def orbit(self, hkl: np.ndarray) -> np.ndarray:
        """(nops, N, 3): h' = h R for every operator (row-vector convention for reciprocal space)."""
        return np.einsum("ni,oij->onj", np.asarray(hkl, dtype=np.int64), self.R)

    def asu_case(self):
        """Index into `_CCP4_ASU` of the inequality set that is an asymmetric unit for these operators, or None."""
        if self._case is _UNSET:
            ax = np.arange(-3, 4)
            g = np.stack(np.meshgrid(ax, ax, ax, indexing="ij"), axis=-1).reshape(-1, 3)
            orb = self.orbit(g)
            orb = np.concatenate([orb, -orb], axis=0)                      # (2 nops, N, 3)
            k = _key(orb)
            self._case = None
            for i, inside in enumerate(_CCP4_ASU):
                sel = inside(orb[..., 0], orb[..., 1], orb[..., 2])
                if not sel.any(axis=0).all():
                    continue
                lo = np.where(sel, k, np.iinfo(np.int64).max).min(axis=0)
                hi = np.where(sel, k, np.iinfo(np.int64).min).max(axis=0)
                if np.all(lo == hi):                                       # one DISTINCT member of every orbit
                    self._case = i
                    break
        return self._case

    def _map(self, hkl: np.ndarray, anomalous: bool, want_asu: bool, want_flags: bool):
        """One pass of `cl_host_asu_map` (careless_amd/csrc/host_format.cpp; host threads) over the rows of `hkl`."""
        import ctypes as C
        from careless_amd._lib import check, get_lib
        h = np.asarray(hkl).reshape(-1, 3)
        if len(h) and (int(h.max()) >= (1 << 19) or int(h.min()) <= -(1 << 19)):
            raise ValueError("Miller index beyond +-2^19")
        h32 = np.ascontiguousarray(h, dtype=np.int32)
        n = len(h32)
        rot = np.ascontiguousarray(self.R, dtype=np.int32)
        trans = np.ascontiguousarray(self.t, dtype=np.float64)
        case = self.asu_case() if want_asu else None
        hasu = np.empty((n, 3), dtype=np.int32) if want_asu else None
        centric = np.empty(n, dtype=np.uint8) if want_flags else None
        eps = np.empty(n, dtype=np.int32) if want_flags else None
        absent = np.empty(n, dtype=np.uint8) if want_flags else None
        p = lambda a: None if a is None else a.ctypes.data_as(C.c_void_p)
        check(get_lib().cl_host_asu_map(p(h32), n, p(rot), p(trans), len(rot), -1 if case is None else int(case), int(bool(anomalous)),
                                        p(hasu), p(centric), p(eps), p(absent), 0), "cl_host_asu_map")
        return hasu, centric, eps, absent

    def to_asu(self, hkl: np.ndarray, anomalous: bool = False) -> np.ndarray:
        return self._map(hkl, anomalous, True, False)[0].astype(np.int64)

    def describe(self, hkl: np.ndarray):
        """centric (N,) bool, epsilon (N,) int, absent (N,) bool."""
        _, centric, eps, absent = self._map(hkl, False, False, True)
        return centric.astype(bool), eps.astype(np.int64), absent.astype(bool)

    def map_rows(self, hkl: np.ndarray, anomalous: bool = False):
        """(ASU representative int32 (N, 3), centric, epsilon, absent) of every row in ONE pass (the formatter's per-observation call)."""
        hasu, centric, eps, absent = self._map(hkl, anomalous, True, True)
        return hasu, centric.view(bool), eps, absent.view(bool)


class ReciprocalASU:
    """All unique reflections to `dmin` (reference careless/io/asu.py:5-83)."""

    def __init__(self, cell, symops: Sequence[str], dmin: float, anomalous: bool = False, spacegroup_name: str = "P 1",
                 spacegroup_number: int = 1):
        self.cell, self.symops, self.dmin, self.anomalous = tuple(cell), list(symops), float(dmin), bool(anomalous)
        self.spacegroup_name, self.spacegroup_number = spacegroup_name, int(spacegroup_number)
        self.ops = SymmetryOps(symops)
        s2max = 1.0 / (self.dmin * self.dmin) * (1.0 + 1e-6)
        Gs = reciprocal_metric(cell)
        # |h_i| <= sqrt(s2max * G_ii) with G the direct metric: bound of the index range
        G = np.linalg.inv(Gs)
        lim = [int(np.floor(np.sqrt(s2max * G[i, i]))) + 1 for i in range(3)]
        ax = [np.arange(-m, m + 1) for m in lim]
        g = np.stack(np.meshgrid(*ax, indexing="ij"), axis=-1).reshape(-1, 3)
        g = g[np.any(g != 0, axis=1)]
        g = g[inv_d2(g, cell) <= s2max]
        g = g[np.all(self.ops.to_asu(g, anomalous) == g, axis=1)]          # keep the representatives
        centric, eps, absent = self.ops.describe(g)
        g, centric, eps = g[~absent], centric[~absent], eps[~absent]
        order = np.lexsort((g[:, 2], g[:, 1], g[:, 0]))
        self.Hall = g[order]
        self._centric, self._eps = centric[order], eps[order]
        self._dHKL = (1.0 / np.sqrt(inv_d2(self.Hall, cell))).astype(np.float32)
        k = _key(self.Hall)
        self._sort = np.argsort(k)
        self._keys = k[self._sort]
        # reflection id by direct lookup over the index box (|h_i| <= lim_i): one gather per observation instead of a binary search
        self._lim = np.asarray(lim, dtype=np.int64)
        self._lut = None                                   # (built at the first lookup; never pickled)

    @property
    def centric(self):
        return self._centric

    @property
    def multiplicity(self):
        return self._eps.astype(np.float32)

    @property
    def dHKL(self):
        return self._dHKL

    def __len__(self):
        return len(self.Hall)

    def _flat(self, H: np.ndarray) -> np.ndarray:
        """Position of every Miller index (inside the box) in the lookup table."""
        lh, lk, ll = (int(v) for v in self._lim)
        H = np.asarray(H)
        return ((H[:, 0].astype(np.int64) + lh) * (2 * lk + 1) + (H[:, 1] + lk)) * (2 * ll + 1) + (H[:, 2] + ll)

    def __getstate__(self):
        state = dict(self.__dict__)
        state["_lut"] = None                               # (a table over the index box: rebuilt on demand, not shipped with the pickle)
        return state

    def _table(self):
        if self._lut is None and int((2 * self._lim + 1).prod()) <= (1 << 25):
            self._lut = np.full(int((2 * self._lim + 1).prod()), -1, dtype=np.int32)
            self._lut[self._flat(self.Hall)] = np.arange(len(self.Hall), dtype=np.int32)
        return self._lut

    def to_refl_id(self, H: np.ndarray) -> np.ndarray:
        """Reflection ids of Miller indices that are already ASU representatives; raises KeyError for anything else."""
        if self._table() is not None:
            H = np.asarray(H)
            H = H.reshape(-1, 3)
            inside = np.ones(len(H), dtype=bool)
            for j in range(3):
                inside &= (H[:, j] >= -self._lim[j]) & (H[:, j] <= self._lim[j])
            if not inside.all():
                raise KeyError("Miller index outside the reciprocal asymmetric unit (absent, beyond dmin, or not mapped to the ASU)")
            ids = self._lut[self._flat(H)]
            if len(ids) and int(ids.min()) < 0:
                raise KeyError("Miller index outside the reciprocal asymmetric unit (absent, beyond dmin, or not mapped to the ASU)")
            return ids.astype(np.int64)
        k = _key(np.asarray(H, dtype=np.int64))
        pos = np.searchsorted(self._keys, k)
        pos = np.clip(pos, 0, len(self._keys) - 1)
        if not np.all(self._keys[pos] == k):
            raise KeyError("Miller index outside the reciprocal asymmetric unit (absent, beyond dmin, or not mapped to the ASU)")
        return self._sort[pos].astype(np.int64)

    def to_miller_index(self, refl_id):
        return self.Hall[np.asarray(refl_id, dtype=np.int64)]


class ReciprocalASUCollection:
    """Several ASUs addressed by one contiguous reflection id (reference careless/io/asu.py:85-143)."""

    def __init__(self, reciprocal_asus: List[ReciprocalASU]):
        self.reciprocal_asus = list(reciprocal_asus)
        sizes = [len(a) for a in self.reciprocal_asus]
        self.offsets = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
        self.asu_ids = np.concatenate([np.full(n, i, dtype=np.int64) for i, n in enumerate(sizes)])
        self.centric = np.concatenate([a.centric for a in self.reciprocal_asus])
        self.multiplicity = np.concatenate([a.multiplicity for a in self.reciprocal_asus])
        self.dHKL = np.concatenate([a.dHKL for a in self.reciprocal_asus])
        self.Hall = np.concatenate([a.Hall for a in self.reciprocal_asus])

    def __len__(self):
        return len(self.reciprocal_asus)

    def __iter__(self):
        return iter(self.reciprocal_asus)

    def to_refl_id(self, asu_ids, H) -> np.ndarray:
        asu_ids = np.asarray(asu_ids, dtype=np.int64).reshape(-1)
        H = np.asarray(H, dtype=np.int64)
        if len(self.reciprocal_asus) == 1 and (len(asu_ids) == 0 or (asu_ids.min() == 0 and asu_ids.max() == 0)):
            return self.reciprocal_asus[0].to_refl_id(H)                 # (one ASU: no row selection, no copies)
        out = np.empty(len(asu_ids), dtype=np.int64)
        for i, a in enumerate(self.reciprocal_asus):
            m = asu_ids == i
            if m.any():
                out[m] = a.to_refl_id(H[m]) + self.offsets[i]
        return out

    def to_asu_id_and_miller_index(self, refl_id):
        refl_id = np.asarray(refl_id, dtype=np.int64).reshape(-1)
        return self.asu_ids[refl_id], self.Hall[refl_id]
