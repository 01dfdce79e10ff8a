"""Minimal MTZ reader / writer (CCP4 MTZ format v1.1, little-endian IEEE) -- what careless needs of
`reciprocalspaceship.read_mtz` / `DataSet.write_mtz` (reference careless/io/formatter.py:179-184, careless/careless.py:72-74):
column data as float32 with their one-letter MTZ types, unit cell, space-group name / number and the symmetry operators.
No external crystallography library: the header is plain 80-character text records."""
from __future__ import annotations

import struct
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np


@dataclass
class Mtz:
    columns: Dict[str, np.ndarray]                 # label -> float32 [nrow] (NaN = missing), in file order
    types: Dict[str, str]                          # label -> MTZ column type: H index, J intensity, Q sigma, F, B batch, ...
    cell: Tuple[float, float, float, float, float, float]
    symops: List[str] = field(default_factory=lambda: ["X, Y, Z"])
    spacegroup_name: str = "P 1"
    spacegroup_number: int = 1
    title: str = ""

    def __len__(self):
        return len(next(iter(self.columns.values()))) if self.columns else 0

    def keys(self):
        return list(self.columns.keys())

    def first_key_of_type(self, t: str) -> Optional[str]:
        for k, v in self.types.items():
            if v == t:
                return k
        return None

    def hkl(self) -> np.ndarray:
        return np.stack([self.columns[k] for k in ("H", "K", "L")], axis=1).astype(np.int64)


def read_mtz(path: str) -> Mtz:
    b = open(path, "rb").read()
    if b[:4] != b"MTZ ":
        raise ValueError(f"{path}: not an MTZ file")
    stamp = b[8:12]
    little = not (len(stamp) >= 1 and (stamp[0] >> 4) == 1)          # 0x44 = little-endian IEEE reals, 0x11 = big-endian
    e = "<" if little else ">"
    hdr_off = (struct.unpack(e + "i", b[4:8])[0] - 1) * 4
    recs = [b[i:i + 80].decode("latin1").rstrip() for i in range(hdr_off, len(b), 80)]
    ncol = nrow = None
    labels, types, symm = [], [], []
    cell, sg_name, sg_num, title = None, "P 1", 1, ""
    for r in recs:
        t = r.split()
        if not t:
            continue
        key = t[0].upper()
        if key == "TITLE":
            title = r[5:].strip()
        elif key == "NCOL":
            ncol, nrow = int(t[1]), int(t[2])
        elif key == "CELL":
            cell = tuple(float(v) for v in t[1:7])
        elif key == "COLUMN":
            labels.append(t[1]); types.append(t[2])
        elif key == "SYMINF":
            # SYMINF nsym nprim lattice number 'name' pointgroup
            sg_num = int(t[4])
            q = r.split("'")
            sg_name = q[1] if len(q) >= 3 else t[5]
        elif key == "SYMM":
            symm.append(r[4:].strip())
        elif key == "END":
            break
    if ncol is None or cell is None or len(labels) != ncol:
        raise ValueError(f"{path}: malformed MTZ header")
    # one pass over the data block: transposed (and byte-swapped when the file is big-endian) into [column][row], whose rows are the columns
    data = np.frombuffer(b, dtype=e + "f4", count=ncol * nrow, offset=80).reshape(nrow, ncol)
    by_col = np.empty((ncol, nrow), dtype=np.float32)
    for a in range(0, nrow, 1 << 16):                      # (row blocks that stay in cache: 10 x faster than one strided transpose)
        by_col[:, a:a + (1 << 16)] = data[a:a + (1 << 16)].T
    cols = {c: by_col[i] for i, c in enumerate(labels)}
    return Mtz(cols, dict(zip(labels, types)), cell, symm or ["X, Y, Z"], sg_name, sg_num, title)


def _rec(s: str) -> bytes:
    return s[:80].ljust(80).encode("latin1")


def write_mtz(path: str, columns: Dict[str, Sequence[float]], types: Dict[str, str], cell, symops: Sequence[str] = ("X, Y, Z",),
              spacegroup_name: str = "P 1", spacegroup_number: int = 1, title: str = "careless_amd", wavelength: float = 0.0):
    """Write one crystal / one dataset.  `columns` must start with H, K, L; NaN marks missing values (VALM NAN)."""
    labels = list(columns.keys())
    if labels[:3] != ["H", "K", "L"]:
        raise ValueError("an MTZ file starts with the H, K, L columns")
    arr = np.stack([np.asarray(columns[k], dtype=np.float32) for k in labels], axis=1)
    nrow, ncol = arr.shape
    a, b_, c, al, be, ga = [float(v) for v in cell]
    from careless_amd.io.asu import inv_d2
    s2 = inv_d2(arr[:, :3].astype(np.int64), cell) if nrow else np.zeros(1)
    lattice = spacegroup_name.strip()[0].upper() if spacegroup_name.strip() else "P"
    ncent = {"P": 1, "A": 2, "B": 2, "C": 2, "I": 2, "R": 3, "H": 3, "F": 4}.get(lattice, 1)
    hdr = [
        "VERS MTZ:V1.1",
        f"TITLE {title}",
        f"NCOL {ncol:8d} {nrow:12d} {0:8d}",
        f"CELL  {a:9.4f} {b_:9.4f} {c:9.4f} {al:9.4f} {be:9.4f} {ga:9.4f}",
        "SORT    0   0   0   0   0",
        f"SYMINF {len(symops):3d} {max(1, len(symops) // ncent):2d} {lattice} {spacegroup_number:5d} '{spacegroup_name}' PG1",
    ]
    hdr += [f"SYMM {s}" for s in symops]
    hdr += [f"RESO {float(s2.min()):.6f} {float(s2.max()):.6f}", "VALM NAN"]
    for i, k in enumerate(labels):
        v = arr[:, i]
        ok = np.isfinite(v)
        lo, hi = (float(v[ok].min()), float(v[ok].max())) if ok.any() else (0.0, 0.0)
        hdr.append(f"COLUMN {k:<30s} {types[k]} {lo:17.9g} {hi:17.9g} {0 if i < 3 else 1:4d}")
    hdr += ["NDIF        2",
            "PROJECT       0 HKL_base", "CRYSTAL       0 HKL_base", "DATASET       0 HKL_base",
            f"DCELL         0 {a:9.4f} {b_:9.4f} {c:9.4f} {al:9.4f} {be:9.4f} {ga:9.4f}", "DWAVEL        0    0.000000",
            "PROJECT       1 careless", "CRYSTAL       1 careless", "DATASET       1 careless",
            f"DCELL         1 {a:9.4f} {b_:9.4f} {c:9.4f} {al:9.4f} {be:9.4f} {ga:9.4f}", f"DWAVEL        1 {wavelength:11.6f}",
            "END", "MTZHIST   1", "written by careless_amd", "MTZENDOFHEADERS"]
    with open(path, "wb") as f:
        f.write(b"MTZ ")
        f.write(struct.pack("<i", 20 + nrow * ncol + 1))          # 1-based word index of the header block
        f.write(bytes([0x44, 0x41, 0x00, 0x00]))                  # machine stamp: little-endian IEEE
        f.write(b"\0" * (80 - 12))
        f.write(arr.astype("<f4").tobytes())
        for r in hdr:
            f.write(_rec(r))
