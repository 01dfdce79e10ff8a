"""A large synthetic Laue (polychromatic) unmerged MTZ for timing the `poly` formatter: N observations of random Miller indices with a
wavelength consistent with one fixed crystal orientation per image is not needed for timing -- wavelengths are drawn in [1.0, 1.2]."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from careless_amd.io.mtz import write_mtz
from careless_amd.io.spacegroups import lookup
n = int(sys.argv[1]); rng = np.random.default_rng(0)
cell = (34., 45., 99., 90., 90., 90.)
H = rng.integers(-30, 31, size=(n, 3)); H = H[np.abs(H).sum(1) > 0]; n = len(H)
batch = np.sort(rng.integers(1, 2001, size=n))
I = rng.gamma(1.0, 100.0, size=n); sig = np.sqrt(I) + 5
cols = {"H": H[:, 0], "K": H[:, 1], "L": H[:, 2], "BATCH": batch, "I": I, "SIGI": sig, "Wavelength": rng.uniform(1.0, 1.2, n),
        "X": rng.uniform(0, 2000, n), "Y": rng.uniform(0, 2000, n)}
types = {"H": "H", "K": "H", "L": "H", "BATCH": "B", "I": "J", "SIGI": "Q", "Wavelength": "R", "X": "R", "Y": "R"}
symops, name, num = lookup("19")
write_mtz(sys.argv[2], cols, types, cell, spacegroup_name=name, spacegroup_number=num, symops=symops)
print("wrote", n)
