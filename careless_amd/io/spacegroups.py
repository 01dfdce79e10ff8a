"""Symmetry operators of the 65 Sohncke (chiral) space groups by name or number, for `--spacegroups` (reference
careless/io/formatter.py:254-263 hands the names to gemmi.SpaceGroup).  Reference settings only: monoclinic unique axis b,
rhombohedral groups on hexagonal axes.  Each entry lists generators; the full operator list is their closure modulo lattice
translations, and the table carries the expected number of operators so that a mistyped generator cannot go unnoticed (it is
checked for every entry by `tests/test_io.py`).  Anything else (non-chiral groups, other settings) has to come from the reflection
file's own header.

What the formatter needs from the operators: the rotation parts (asymmetric-unit mapping, centric flags, multiplicities) and the
intrinsic screw / centring translations (systematic absences); origin-dependent translation parts do not enter either."""
from __future__ import annotations

from fractions import Fraction
from typing import Dict, List, Tuple

import numpy as np

_C = ["x+1/2,y+1/2,z"]
_I = ["x+1/2,y+1/2,z+1/2"]
_F = ["x,y+1/2,z+1/2", "x+1/2,y,z+1/2"]
_R = ["x+2/3,y+1/3,z+1/3"]
_222 = ["-x,-y,z", "-x,y,-z"]
_23 = _222 + ["z,x,y"]
_213 = ["-x+1/2,-y,z+1/2", "-x,y+1/2,-z+1/2", "z,x,y"]

# number: (Hermann-Mauguin name, generators, number of operators)
_TABLE: Dict[int, Tuple[str, List[str], int]] = {
    1: ("P 1", [], 1),
    3: ("P 1 2 1", ["-x,y,-z"], 2), 4: ("P 1 21 1", ["-x,y+1/2,-z"], 2), 5: ("C 1 2 1", ["-x,y,-z"] + _C, 4),
    16: ("P 2 2 2", _222, 4), 17: ("P 2 2 21", ["-x,-y,z+1/2", "-x,y,-z+1/2"], 4),
    18: ("P 21 21 2", ["-x,-y,z", "-x+1/2,y+1/2,-z"], 4), 19: ("P 21 21 21", ["-x+1/2,-y,z+1/2", "-x,y+1/2,-z+1/2"], 4),
    20: ("C 2 2 21", ["-x,-y,z+1/2", "-x,y,-z+1/2"] + _C, 8), 21: ("C 2 2 2", _222 + _C, 8), 22: ("F 2 2 2", _222 + _F, 16),
    23: ("I 2 2 2", _222 + _I, 8), 24: ("I 21 21 21", ["-x+1/2,-y,z+1/2", "-x,y+1/2,-z+1/2"] + _I, 8),
    75: ("P 4", ["-y,x,z"], 4), 76: ("P 41", ["-y,x,z+1/4"], 4), 77: ("P 42", ["-y,x,z+1/2"], 4), 78: ("P 43", ["-y,x,z+3/4"], 4),
    79: ("I 4", ["-y,x,z"] + _I, 8), 80: ("I 41", ["-y,x+1/2,z+1/4"] + _I, 8),
    89: ("P 4 2 2", ["-y,x,z", "-x,y,-z"], 8), 90: ("P 4 21 2", ["-y+1/2,x+1/2,z", "-x+1/2,y+1/2,-z"], 8),
    91: ("P 41 2 2", ["-y,x,z+1/4", "-x,y,-z"], 8), 92: ("P 41 21 2", ["-y+1/2,x+1/2,z+1/4", "-x+1/2,y+1/2,-z+1/4"], 8),
    93: ("P 42 2 2", ["-y,x,z+1/2", "-x,y,-z"], 8), 94: ("P 42 21 2", ["-y+1/2,x+1/2,z+1/2", "-x+1/2,y+1/2,-z+1/2"], 8),
    95: ("P 43 2 2", ["-y,x,z+3/4", "-x,y,-z"], 8), 96: ("P 43 21 2", ["-y+1/2,x+1/2,z+3/4", "-x+1/2,y+1/2,-z+3/4"], 8),
    97: ("I 4 2 2", ["-y,x,z", "-x,y,-z"] + _I, 16), 98: ("I 41 2 2", ["-y,x+1/2,z+1/4", "-x+1/2,y,-z+3/4"] + _I, 16),
    143: ("P 3", ["-y,x-y,z"], 3), 144: ("P 31", ["-y,x-y,z+1/3"], 3), 145: ("P 32", ["-y,x-y,z+2/3"], 3),
    146: ("R 3", ["-y,x-y,z"] + _R, 9),
    149: ("P 3 1 2", ["-y,x-y,z", "-y,-x,-z"], 6), 150: ("P 3 2 1", ["-y,x-y,z", "y,x,-z"], 6),
    151: ("P 31 1 2", ["-y,x-y,z+1/3", "-y,-x,-z+2/3"], 6), 152: ("P 31 2 1", ["-y,x-y,z+1/3", "y,x,-z"], 6),
    153: ("P 32 1 2", ["-y,x-y,z+2/3", "-y,-x,-z+1/3"], 6), 154: ("P 32 2 1", ["-y,x-y,z+2/3", "y,x,-z"], 6),
    155: ("R 3 2", ["-y,x-y,z", "y,x,-z"] + _R, 18),
    168: ("P 6", ["x-y,x,z"], 6), 169: ("P 61", ["x-y,x,z+1/6"], 6), 170: ("P 65", ["x-y,x,z+5/6"], 6),
    171: ("P 62", ["x-y,x,z+1/3"], 6), 172: ("P 64", ["x-y,x,z+2/3"], 6), 173: ("P 63", ["x-y,x,z+1/2"], 6),
    177: ("P 6 2 2", ["x-y,x,z", "y,x,-z"], 12), 178: ("P 61 2 2", ["x-y,x,z+1/6", "y,x,-z+1/3"], 12),
    179: ("P 65 2 2", ["x-y,x,z+5/6", "y,x,-z+2/3"], 12), 180: ("P 62 2 2", ["x-y,x,z+1/3", "y,x,-z+2/3"], 12),
    181: ("P 64 2 2", ["x-y,x,z+2/3", "y,x,-z+1/3"], 12), 182: ("P 63 2 2", ["x-y,x,z+1/2", "y,x,-z"], 12),
    195: ("P 2 3", _23, 12), 196: ("F 2 3", _23 + _F, 48), 197: ("I 2 3", _23 + _I, 24),
    198: ("P 21 3", _213, 12), 199: ("I 21 3", _213 + _I, 24),
    207: ("P 4 3 2", _23 + ["y,x,-z"], 24), 208: ("P 42 3 2", _23 + ["y+1/2,x+1/2,-z+1/2"], 24),
    209: ("F 4 3 2", _23 + ["y,x,-z"] + _F, 96),
    210: ("F 41 3 2", ["-x,-y+1/2,z+1/2", "-x+1/2,y+1/2,-z", "z,x,y", "y+3/4,x+1/4,-z+3/4"] + _F, 96),
    211: ("I 4 3 2", _23 + ["y,x,-z"] + _I, 48),
    212: ("P 43 3 2", _213 + ["y+1/4,x+3/4,-z+3/4"], 24), 213: ("P 41 3 2", _213 + ["y+3/4,x+1/4,-z+1/4"], 24),
    214: ("I 41 3 2", _213 + ["y+3/4,x+1/4,-z+1/4"] + _I, 48),
}


def _parse(op: str):
    from careless_amd.io.asu import parse_symop
    R, t = parse_symop(op)
    return R, tuple(Fraction(float(v)).limit_denominator(12) % 1 for v in t)


def _format(R: np.ndarray, t) -> str:
    parts = []
    for i in range(3):
        terms = "".join(("+" if c > 0 else "-") + ax for ax, c in zip("XYZ", R[i]) if c)
        if t[i]:
            terms += f"+{t[i].numerator}/{t[i].denominator}"
        parts.append(terms.lstrip("+"))
    return ", ".join(parts)


def operators(number: int) -> List[str]:
    """All operators (centring included) of space group `number`, identity first, as 'X, Y, Z' strings like an MTZ header's."""
    name, gens, order = _TABLE[number]
    ops = {(tuple(np.eye(3, dtype=np.int64).ravel()), (Fraction(0),) * 3)}
    gens = [_parse(g) for g in gens]
    frontier = list(ops)
    while frontier:
        new = []
        for Ra, ta in frontier:
            Ra = np.array(Ra).reshape(3, 3)
            for Rg, tg in gens:
                R = Rg @ Ra                                                   # (g o a)(x) = Rg (Ra x + ta) + tg
                t = tuple((sum(Fraction(int(Rg[i, k])) * ta[k] for k in range(3)) + tg[i]) % 1 for i in range(3))
                key = (tuple(R.ravel()), t)
                if key not in ops:
                    ops.add(key)
                    new.append(key)
        frontier = new
        if len(ops) > 192:
            break
    if len(ops) != order:
        raise RuntimeError(f"space-group table entry {number} ({name}): {len(ops)} operators instead of {order}")
    ident = (tuple(np.eye(3, dtype=np.int64).ravel()), (Fraction(0),) * 3)
    rest = sorted(ops - {ident}, key=lambda k: (k[1], k[0]))
    return [_format(np.array(R).reshape(3, 3), t) for R, t in [ident] + rest]


def _norm(name: str) -> str:
    s = str(name).replace(" ", "").replace("_", "").upper()
    if s.endswith(":H"):
        s = s[:-2]
    if s.startswith("H3"):                        # 'H 3', 'H 3 2': the hexagonal-axes names of R 3, R 3 2
        s = "R" + s[1:]
    return s


_BY_NAME: Dict[str, int] = {}
for _n, (_hm, _g, _o) in _TABLE.items():
    _BY_NAME[_norm(_hm)] = _n
    _parts = _hm.split()
    if len(_parts) == 4 and _parts[1] == "1" and _parts[3] == "1":          # 'P 1 21 1' is also written 'P 21'
        _BY_NAME[_norm(_parts[0] + _parts[2])] = _n


def lookup(spec) -> Tuple[List[str], str, int]:
    """(operators, Hermann-Mauguin name, number) of a space group given by number or name ('19', 'P 21 21 21', 'P212121', 'H 3')."""
    s = str(spec).strip()
    number = int(s) if s.isdigit() else _BY_NAME.get(_norm(s))
    if number not in _TABLE:
        raise NotImplementedError(f"--spacegroups={spec}: not in the built-in table of the 65 chiral space groups (reference settings); "
                                  "careless_amd otherwise takes the symmetry operators from the reflection file header")
    return operators(number), _TABLE[number][0], number
