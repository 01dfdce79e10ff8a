"""Test-side reader of the reference's own MTZ fixture `tests/data/pyp_off.mtz` (copied verbatim to tests/golden/pyp_off.mtz: 9 280
bytes, 166 observations x 10 columns of real PYP Laue data the reference uses as its mono AND Laue test input,
reference tests/conftest.py:98-219), plus the minimum of reciprocal-space bookkeeping needed to turn it into careless `inputs`
(BASELINE.json configs[0]).  TEST INFRASTRUCTURE: the product does not parse MTZ files (formatter scope).

What the reference's formatter does with gemmi/reciprocalspaceship (careless/io/formatter.py:87-146,354-400; io/asu.py:5-83) and
what is restated here with numpy from the symmetry operators in the file header:
  * reciprocal ASU to dmin: every hkl with d >= dmin mapped to the Laue-group ASU, systematic absences removed;
  * centric flag (some rotation maps h to -h), multiplicity epsilon (number of rotations that fix h);
  * refl_id = index into that ASU list; metadata = z-scored [1/d^2, image id] (the reference's `dHKL,image_id` keys).
"""
from __future__ import annotations

import os
import re
import struct

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
PYP = os.path.join(HERE, "golden", "pyp_off.mtz")


def read_mtz(path=PYP):
    b = open(path, "rb").read()
    assert b[:4] == b"MTZ "
    hdr_off = (struct.unpack("<i", b[4:8])[0] - 1) * 4
    recs = [b[i:i + 80].decode("latin1").rstrip() for i in range(hdr_off, len(b), 80)]
    ncol = nrow = None
    cols, symm, cell = [], [], None
    for r in recs:
        t = r.split()
        if not t:
            continue
        if t[0] == "NCOL":
            ncol, nrow = int(t[1]), int(t[2])
        elif t[0] == "COLUMN":
            cols.append(t[1])
        elif t[0] == "CELL":
            cell = tuple(float(v) for v in t[1:7])
        elif t[0] == "SYMM":
            symm.append(r[4:].strip())
        elif t[0] == "END":
            break
    data = np.frombuffer(b, dtype="<f4", count=ncol * nrow, offset=80).reshape(nrow, ncol)
    return {c: data[:, i].copy() for i, c in enumerate(cols)}, cell, symm


def parse_symop(s):
    """'X-Y,X,Z+1/2' -> (3x3 int rotation acting on real-space coordinates, translation)"""
    R = np.zeros((3, 3), dtype=int)
    t = np.zeros(3)
    for i, part in enumerate(s.replace(" ", "").upper().split(",")):
        for sign, num, den, ax in re.findall(r"([+-]?)(?:(\d+)/(\d+)|([XYZ]))", part):
            sg = -1 if sign == "-" else 1
            if ax:
                R[i, "XYZ".index(ax)] += sg
            else:
                t[i] += sg * int(num) / int(den)
    return R, t


def inv_d2_hex(hkl, cell):
    a, _, c = cell[:3]
    h, k, l = hkl[:, 0].astype(float), hkl[:, 1].astype(float), hkl[:, 2].astype(float)
    return 4.0 / 3.0 * (h * h + h * k + k * k) / (a * a) + l * l / (c * c)


def in_asu_6m(h):
    """Laue group 6/m reciprocal ASU (the convention of the file's own H, K, L columns): l >= 0 and (h >= 0, k > 0 or h = k = 0)"""
    return (h[:, 2] >= 0) & (((h[:, 0] >= 0) & (h[:, 1] > 0)) | ((h[:, 0] == 0) & (h[:, 1] == 0)))


def build_inputs(path=PYP):
    cols, cell, symm = read_mtz(path)
    ops = [parse_symop(s) for s in symm]
    hkl = np.stack([cols["H"], cols["K"], cols["L"]], axis=1).astype(int)
    assert np.all(in_asu_6m(hkl))
    s2 = inv_d2_hex(hkl, cell)
    s2max = s2.max() * (1 + 1e-6)
    # enumerate the ASU to dmin
    hm = int(np.ceil(np.sqrt(s2max) * cell[0])) + 1
    lm = int(np.ceil(np.sqrt(s2max) * cell[2])) + 1
    g = np.array([(h, k, l) for h in range(-hm, hm + 1) for k in range(-hm, hm + 1) for l in range(0, lm + 1)], dtype=int)
    g = g[in_asu_6m(g) & (inv_d2_hex(g, cell) <= s2max) & np.any(g != 0, axis=1)]
    rots = [R for R, _ in ops]
    absent = np.zeros(len(g), bool)
    eps = np.zeros(len(g), int)
    centric = np.zeros(len(g), bool)
    for R, t in ops:
        hR = g @ R                                    # h' = h R  (row vector convention for reciprocal space)
        same = np.all(hR == g, axis=1)
        eps += same
        phase = g @ t
        absent |= same & (np.abs(phase - np.round(phase)) > 1e-6)
        centric |= np.all(hR == -g, axis=1)
    g, eps, centric = g[~absent], eps[~absent], centric[~absent]
    order = np.lexsort((g[:, 2], g[:, 1], g[:, 0]))
    g, eps, centric = g[order], eps[order], centric[order]
    lut = {tuple(v): i for i, v in enumerate(g)}
    refl_id = np.array([lut[tuple(v)] for v in hkl], dtype=np.int64)
    image_id = cols["BATCH"].astype(np.int64)
    raw = np.stack([s2, image_id.astype(float)], axis=1)          # dHKL -> 1/d^2 (formatter.py:370), then z-scored
    meta = ((raw - raw.mean(0)) / raw.std(0)).astype(np.float32)
    return dict(refl_id=refl_id, image_id=image_id, file_id=np.zeros(len(hkl), np.int64), metadata=meta,
                iobs=cols["I"].astype(np.float32), sigiobs=cols["SigI"].astype(np.float32), centric=centric,
                multiplicity=eps.astype(np.float32), n_images=int(image_id.max()) + 1, n_refl=len(g), hkl_asu=g)
