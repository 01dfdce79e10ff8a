"""Line-by-line Python restatement of the CrystFEL stream reader (the package's implementation of rounds 1-5) -- the checker of the native
parser `cl_host_crystfel_count / _parse` (careless_amd/csrc/host_format.cpp) in tests/test_host_format.py; test infrastructure only.
Minimal CrystFEL `.stream` reader (what careless needs of `reciprocalspaceship.read_crystfel`, reference
careless/io/formatter.py:179-184): the indexed reflection lists of every crystal as one unmerged table -- H, K, L, I, SigI,
BATCH (crystal number), peak, background, XDET / YDET (fs / ss pixel coordinates) -- plus the target unit cell of the stream
header.  A stream carries no symmetry: the space group comes from `--spacegroups` (the reference needs that flag too)."""
from __future__ import annotations

import re

import numpy as np

from careless_amd.io.mtz import Mtz


def read_crystfel(path: str, symops=("X, Y, Z",), spacegroup_name: str = "P 1", spacegroup_number: int = 1) -> Mtz:
    cell = {}
    rows, batch = [], -1
    in_cell = in_refl = False
    with open(path) as f:
        for line in f:
            if line.startswith("----- Begin unit cell"):
                in_cell = True
                continue
            if line.startswith("----- End unit cell"):
                in_cell = False
                continue
            if in_cell:
                m = re.match(r"\s*(a|b|c|al|be|ga)\s*=\s*([-+0-9.eE]+)", line)
                if m:
                    cell[m.group(1)] = float(m.group(2))
                continue
            if line.startswith("--- Begin crystal"):
                batch += 1
                continue
            if line.startswith("Reflections measured after indexing"):
                in_refl = True
                next(f)                                   # column header
                continue
            if line.startswith("End of reflections"):
                in_refl = False
                continue
            if in_refl:
                t = line.split()
                if len(t) >= 9:
                    rows.append((int(t[0]), int(t[1]), int(t[2]), float(t[3]), float(t[4]), float(t[5]), float(t[6]), float(t[7]),
                                 float(t[8]), batch))
    if not rows or len(cell) < 6:
        raise ValueError(f"{path}: no indexed reflections or no unit cell in the stream")
    a = np.array(rows, dtype=np.float64)
    cols = {"H": a[:, 0], "K": a[:, 1], "L": a[:, 2], "I": a[:, 3], "SigI": a[:, 4], "peak": a[:, 5], "background": a[:, 6],
            "XDET": a[:, 7], "YDET": a[:, 8], "BATCH": a[:, 9]}
    cols = {k: v.astype(np.float32) for k, v in cols.items()}
    types = {"H": "H", "K": "H", "L": "H", "I": "J", "SigI": "Q", "peak": "R", "background": "R", "XDET": "R", "YDET": "R", "BATCH": "B"}
    return Mtz(cols, types, tuple(cell[k] for k in ("a", "b", "c", "al", "be", "ga")), list(symops), spacegroup_name, spacegroup_number,
               title=f"CrystFEL stream {path}")
