"""Minimal CrystFEL `.stream` reader (what careless needs of `reciprocalspaceship.read_crystfel`, reference
careless/io/formatter.py:179-184): the indexed reflection lists of every crystal as one unmerged table -- H, K, L, I, SigI,
BATCH (crystal number), peak, background, XDET / YDET (fs / ss pixel coordinates) -- plus the target unit cell of the stream
header.  A stream carries no symmetry: the space group comes from `--spacegroups` (the reference needs that flag too).

Serial-crystallography streams are text files of 10^7 .. 10^8 reflection lines: the lists are parsed natively, in parallel on host
threads, straight from a read-only memory map of the file (`cl_host_crystfel_count / _parse`, careless_amd/csrc/host_format.cpp;
round 5 -- the line-by-line Python loop of rounds 1-4, 8 s and 1 GB of tuples per million reflections, is the checker in
tests/ref_crystfel.py).  Only the unit cell of the header is read here."""
from __future__ import annotations

import ctypes as C
import mmap
import re

import numpy as np

from careless_amd.io.mtz import Mtz

_CELL_LINE = re.compile(rb"\s*(a|b|c|al|be|ga)\s*=\s*([-+0-9.eE]+)")
COLUMNS = ("H", "K", "L", "I", "SigI", "peak", "background", "XDET", "YDET", "BATCH")          # the table's columns in the parser's order


def _unit_cell(buf) -> dict:
    """a, b, c, al, be, ga of every "----- Begin unit cell" block of the stream, later blocks overriding earlier ones as lines are met."""
    cell, pos = {}, 0
    while True:
        a = buf.find(b"----- Begin unit cell", pos)
        if a < 0 or (a > 0 and buf[a - 1:a] != b"\n"):
            if a < 0:
                break
            pos = a + 1
            continue
        # (a unit-cell block is a dozen short lines: the end marker is looked for in the 4 KiB behind the begin marker, so that a header cut
        #  off before it does not make this copy and split a stream of tens of GB)
        b = buf.find(b"\n----- End unit cell", a, a + 4096)
        b = min(len(buf), a + 4096) if b < 0 else b
        for line in buf[a:b].split(b"\n")[1:]:
            m = _CELL_LINE.match(line)
            if m:
                cell[m.group(1).decode()] = float(m.group(2))
        pos = b + 1
    return cell


def read_crystfel(path: str, symops=("X, Y, Z",), spacegroup_name: str = "P 1", spacegroup_number: int = 1) -> Mtz:
    from careless_amd._lib import check, get_lib
    lib = get_lib()
    with open(path, "rb") as f:
        try:
            mm = mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ)
        except ValueError:                                     # an empty file cannot be mapped
            raise ValueError(f"{path}: no indexed reflections or no unit cell in the stream")
        try:
            cell = _unit_cell(mm)
            view = np.frombuffer(mm, dtype=np.uint8)
            ptr, nbytes = view.ctypes.data_as(C.c_void_p), int(view.size)
            n = int(lib.cl_host_crystfel_count(ptr, nbytes, None, 0))
            if n < 0:
                check(n, "cl_host_crystfel_count")
            if n == 0 or len(cell) < 6:
                raise ValueError(f"{path}: no indexed reflections or no unit cell in the stream")
            table = np.empty((len(COLUMNS), n), dtype=np.float32)
            rc = int(lib.cl_host_crystfel_parse(ptr, nbytes, n, table.ctypes.data_as(C.c_void_p), 0))
            if rc == -5:
                raise ValueError(f"{path}: a reflection line holds a field that is not a number")
            check(rc, "cl_host_crystfel_parse")
        finally:
            view = ptr = None                                  # (the map cannot close while an array exports its buffer)
            mm.close()
    cols = {k: table[i] for i, k in enumerate(COLUMNS)}
    cols = {k: cols[k] for k in ("H", "K", "L", "I", "SigI", "peak", "background", "XDET", "YDET", "BATCH")}
    types = {"H": "H", "K": "H", "L": "H", "I": "J", "SigI": "Q", "peak": "R", "background": "R", "XDET": "R", "YDET": "R", "BATCH": "B"}
    return Mtz(cols, types, tuple(cell[k] for k in ("a", "b", "c", "al", "be", "ga")), list(symops), spacegroup_name, spacegroup_number,
               title=f"CrystFEL stream {path}")
