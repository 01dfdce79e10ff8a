"""Reflection tables -> the `inputs` tuple of the merging model.

Restates what `MonoFormatter` / `LaueFormatter` do (reference careless/io/formatter.py:60-653) on plain column dictionaries read
by `careless_amd.io.mtz` -- resolution cut, systematic absences, observed Miller indices as metadata, mapping to the reciprocal
ASU, key guessing by MTZ column type, I/sigma cut, harmonic expansion for Laue data (reference careless/utils/laue.py:9-81),
image / harmonic group ids, 1/d^2, z-scoring, positional encoding, padded per-group intensities -- without reciprocalspaceship
or gemmi: the symmetry comes from the operators in the file header (`careless_amd.io.asu`).  CrystFEL `.stream` files go through `careless_amd.io.crystfel`."""
from __future__ import annotations

import warnings
from typing import Dict, List, Optional, Sequence

import numpy as np

from careless_amd.io.asu import ReciprocalASU, ReciprocalASUCollection, SymmetryOps, inv_d2
from careless_amd.io.mtz import Mtz, read_mtz
from careless_amd.models.base import BaseModel
from careless_amd.synthetic import positional_encoding


def standardize_metadata(metadata: np.ndarray, metadata_keys: Optional[Sequence[str]] = None) -> np.ndarray:
    """z-score every column, leaving zero-variance columns alone (reference formatter.py:41-57)."""
    metadata = np.array(metadata, dtype=np.float32)
    std = metadata.std(0)
    zeros = std == 0.0
    for k, v in enumerate(std):
        if v == 0.0:
            name = metadata_keys[k] if metadata_keys is not None else k
            warnings.warn(f'Metadata column "{name}" with zero standard deviation will not be standardized.')
    nz = ~zeros
    metadata[:, nz] = (metadata[:, nz] - metadata[:, nz].mean(0)) / metadata[:, nz].std(0)
    return metadata


def _ngroup(*cols) -> np.ndarray:
    """pandas `groupby(cols).ngroup()`: dense ids in sorted key order (reference formatter.py:203, 617).  Integer columns fold into ONE
    int64 key whose order is the columns' lexicographic order; a key range of the order of the row count is ranked through a presence
    table (`cl_host_dense_ids`, careless_amd/csrc/host_format.cpp: no sort -- the image ids), a larger one by sorting the key."""
    cols = [np.asarray(c).reshape(-1) for c in cols]
    n = len(cols[0])
    if n == 0:
        return np.zeros(0, dtype=np.int64)
    if all(np.issubdtype(c.dtype, np.integer) for c in cols):
        lo = [int(c.min()) for c in cols]
        span = [int(c.max()) - l + 1 for c, l in zip(cols, lo)]
        total = 1
        for v in span:
            total *= v
        if total < (1 << 62):
            key = np.zeros(n, dtype=np.int64)
            for c, l, v in zip(cols, lo, span):
                key *= v
                key += c.astype(np.int64) - l
            if total <= max(1 << 22, 4 * n):
                import ctypes as C
                from careless_amd._lib import check, get_lib
                ids = np.empty(n, dtype=np.int64)
                rc = get_lib().cl_host_dense_ids(key.ctypes.data_as(C.c_void_p), n, 0, total - 1, ids.ctypes.data_as(C.c_void_p), None, 0)
                if rc == 0:
                    return ids
                if rc not in (-2, -3):                     # (-2 / -3: no table of that size -- the key is sorted below)
                    check(rc, "cl_host_dense_ids")
            _, inv = np.unique(key, return_inverse=True)
            return inv.reshape(-1).astype(np.int64)
    keys = np.stack(cols, axis=1)
    _, inv = np.unique(keys, axis=0, return_inverse=True)
    return inv.reshape(-1).astype(np.int64)


def _key_error(key, kind, flag, table):
    if key is None:
        msg = f"Unable to determine the {kind} column key. Please use {flag} to specify the {kind} key name."
    else:
        msg = f"User supplied {kind} column key {key}, but {key} is not available in the input data. "
    raise ValueError(msg + " Available keys are: \n" + ",".join(table.keys()))


def expand_harmonics(cols: Dict[str, np.ndarray], cell, dmin: Optional[float] = None, wavelength_key: str = "Wavelength"):
    """Every observation repeated once per harmonic n = 1..floor(d_0 / dmin) of its central ray, with H = n H_0, wavelength
    lambda_0 / n and the new columns H_0, K_0, L_0 (reference careless/utils/laue.py:9-81)."""
    H = np.stack([cols["H"], cols["K"], cols["L"]], axis=1).astype(np.int64)
    d = 1.0 / np.sqrt(inv_d2(H, cell))
    if dmin is None:
        dmin = d.min() - 1e-12
    nobs = np.gcd.reduce(H, axis=-1)
    H0 = (H // nobs[:, None]).astype(np.int64)
    d0 = d * nobs
    wl0 = np.asarray(cols[wavelength_key], dtype=np.float64) * nobs
    n_max = np.floor_divide(d0, dmin).astype(np.int64)
    n = np.arange(max(int(n_max.max()), 0)) + 1
    idx, nn = np.where(n[None, :] <= n_max[:, None])
    nn = nn + 1
    out = {k: np.asarray(v)[idx] for k, v in cols.items()}
    out["H_0"], out["K_0"], out["L_0"] = H0[idx].T
    out[wavelength_key] = (wl0[idx] / nn).astype(np.float32)
    Hn = nn[:, None] * H0[idx]
    out["H"], out["K"], out["L"] = Hn.T
    return out


def parse_spacegroups(spec, n_files: int):
    """`--spacegroups` (reference formatter.py:254-263): one entry or one per file, by number or Hermann-Mauguin name.  The
    operators come from the built-in table of the 65 chiral space groups (`careless_amd.io.spacegroups`); other groups and
    non-reference settings raise NotImplementedError (a reflection file's own header is always honoured)."""
    from careless_amd.io.spacegroups import lookup
    if spec is None:
        return None
    names = [v.strip() for v in str(spec).split(",")]
    if len(names) == 1:
        names = names * n_files
    elif len(names) != n_files:
        raise ValueError("Multiple values provided for --spacegroups=, but the number of provided values does not match the number of "
                         "reflection files. Either provide a single spacegroup or one per reflection file as a comma-separated list. ")
    return [lookup(v) for v in names]


class DataFormatter:
    wavelength_key = None
    spacegroups = None

    def __init__(self, intensity_key=None, uncertainty_key=None, image_key=None, metadata_keys=("dHKL",), separate_outputs=False,
                 anomalous=False, dmin=0.0, isigi_cutoff=None, positional_encoding_keys=None, encoding_bit_depth=5, standardize=True):
        self.intensity_key, self.uncertainty_key, self.image_key = intensity_key, uncertainty_key, image_key
        self.metadata_keys = list(metadata_keys)
        self.separate_outputs, self.anomalous = bool(separate_outputs), bool(anomalous)
        self.dmin, self.isigi_cutoff = dmin, isigi_cutoff
        self.positional_encoding_keys = positional_encoding_keys
        self.encoding_bit_depth = encoding_bit_depth
        self.standardize = standardize

    # -- per file -----------------------------------------------------------------------------------------
    def _guess_keys(self, mtz: Mtz):
        image_key = self.image_key or mtz.first_key_of_type("B")
        if image_key not in mtz.columns:
            _key_error(self.image_key, "Batch", "--image-key", mtz)
        intensity_key = self.intensity_key or mtz.first_key_of_type("J")
        if intensity_key not in mtz.columns:
            _key_error(self.intensity_key, "Intensity", "--intensity-key", mtz)
        uncertainty_key = self.uncertainty_key
        if uncertainty_key is None:
            for prefix in ("Sig", "SIG"):
                if prefix + intensity_key in mtz.columns:
                    uncertainty_key = prefix + intensity_key
        if uncertainty_key is None:
            uncertainty_key = mtz.first_key_of_type("Q")
        if uncertainty_key not in mtz.columns:
            _key_error(self.uncertainty_key, "Stddev", "--uncertainty-key", mtz)
        return image_key, intensity_key, uncertainty_key

    def prep_dataset(self, mtz: Mtz) -> Dict[str, np.ndarray]:
        raise NotImplementedError("Formatter classes should implement `prep_dataset`")

    def _common_prep(self, cols: Dict[str, np.ndarray], mtz: Mtz, keys) -> Dict[str, np.ndarray]:
        ops = SymmetryOps(mtz.symops)
        H = np.empty((len(cols["H"]), 3), dtype=np.int32)                 # (filled column by column: no int64 / float64 copies of the table)
        for j, k in enumerate(("H", "K", "L")):
            c = np.asarray(cols[k])
            if c.dtype.kind == "f" and not np.isfinite(c).all():       # (a NaN cast to int32 is INT_MIN: say which column, not "index beyond 2^19")
                raise ValueError(f"{getattr(mtz, 'path', 'reflection file')}: column {k} holds {int((~np.isfinite(c)).sum())} missing / non-finite Miller indices")
            H[:, j] = c
        Hasu, _, _, absent = ops.map_rows(H, self.anomalous)            # ds.remove_absences + ds.hkl_to_asu in one native pass
        if absent.any():
            keep = ~absent
            cols = {k: np.asarray(v)[keep] for k, v in cols.items()}
            H, Hasu = H[keep], Hasu[keep]
        cols = dict(cols)
        cols["Hobs"], cols["Kobs"], cols["Lobs"] = H.T.astype(np.float32)
        cols["H"], cols["K"], cols["L"] = Hasu.T
        cols["dHKL"] = (1.0 / np.sqrt(inv_d2(Hasu, mtz.cell))).astype(np.float32)
        image_key, intensity_key, uncertainty_key = keys
        cols["intensity"] = np.asarray(cols[intensity_key], dtype=np.float32)
        cols["uncertainty"] = np.asarray(cols[uncertainty_key], dtype=np.float32)
        cols["image_id"] = np.asarray(cols[image_key]).astype(np.int64)
        if self.isigi_cutoff is not None:
            keep = ~(cols["intensity"] / cols["uncertainty"] < self.isigi_cutoff)
            if not keep.all():
                cols = {k: v[keep] for k, v in cols.items()}
        return cols

    # -- all files ----------------------------------------------------------------------------------------
    def get_data_and_asu_collection(self, datasets: Sequence[Mtz]):
        tables, cells, syms, sgs = [], [], [], []
        for file_id, mtz in enumerate(datasets):
            cols = self.prep_dataset(mtz)
            n = len(cols["H"])
            cols["file_id"] = np.full(n, file_id, dtype=np.int64)
            cols["asu_id"] = np.full(n, file_id if self.separate_outputs else 0, dtype=np.int64)
            tables.append(cols)
            cells.append(mtz.cell); syms.append(mtz.symops); sgs.append((mtz.spacegroup_name, mtz.spacegroup_number))
        common = set(tables[0])
        for t in tables[1:]:
            common &= set(t)
        data = ({k: v for k, v in tables[0].items() if k in common} if len(tables) == 1 else
                {k: np.concatenate([t[k] for t in tables]) for k in tables[0] if k in common})
        dmin = float(data["dHKL"].min())
        if self.separate_outputs:
            asus = [ReciprocalASU(c, s, dmin, self.anomalous, *g) for c, s, g in zip(cells, syms, sgs)]
        else:
            asus = [ReciprocalASU(cells[0], syms[0], dmin, self.anomalous, *sgs[0])]
        data["image_id"] = _ngroup(data["file_id"], data["image_id"])
        return data, ReciprocalASUCollection(asus)

    def _metadata(self, data):
        data = dict(data)
        data["dHKL"] = data["dHKL"].astype(np.float64) ** -2.0
        missing = [k for k in self.metadata_keys if k not in data]
        if missing:
            raise ValueError("".join(f'Metadata key "{k}" not found in input data. \n' for k in missing) +
                             "Available keys are: \n" + ",".join(data.keys()))
        metadata = np.stack([np.asarray(data[k], dtype=np.float32) for k in self.metadata_keys], axis=1)
        if self.standardize:
            metadata = standardize_metadata(metadata, self.metadata_keys)
        if self.positional_encoding_keys is not None:
            enc = np.stack([np.asarray(data[k], dtype=np.float32) for k in self.positional_encoding_keys], axis=1)
            metadata = np.concatenate([metadata, positional_encoding(enc, self.encoding_bit_depth, dtype=np.float32)], axis=1)
        return metadata.astype(np.float32)

    @staticmethod
    def pack_inputs(inputs_dict):
        inputs = ()
        for i in range(len(BaseModel.input_index)):
            k = BaseModel.get_name_by_index(i)
            if k not in inputs_dict:
                break
            inputs += (inputs_dict[k],)
        return inputs

    def __call__(self, datasets):
        data, rac = self.get_data_and_asu_collection(list(datasets))
        return self.finalize(data, rac)

    def load(self, filename: str, file_id: int) -> Mtz:
        if str(filename).endswith(".mtz"):
            ds = read_mtz(filename)
        elif str(filename).endswith(".stream"):
            from careless_amd.io.crystfel import read_crystfel
            if self.spacegroups is None:
                raise ValueError("Could not determine spacegroups. Please supply the --spacegroups flag")     # reference formatter.py:113
            ds = read_crystfel(filename)
        else:
            raise ValueError(f"{filename}: reflection files are .mtz or .stream")
        if self.spacegroups is not None:
            ds.symops, ds.spacegroup_name, ds.spacegroup_number = self.spacegroups[file_id]
        return ds

    def format_files(self, files):
        return self([self.load(f, i) for i, f in enumerate(files)])


class MonoFormatter(DataFormatter):
    @classmethod
    def from_parser(cls, parser):
        pe = parser.positional_encoding_keys.split(",") if parser.positional_encoding_keys is not None else None
        fmt = cls(parser.intensity_key, parser.uncertainty_key, parser.image_key, parser.metadata_keys.split(","), parser.separate_files,
                  parser.anomalous, 0.0 if parser.dmin is None else parser.dmin, parser.isigi_cutoff, pe,
                  parser.positional_encoding_frequencies, standardize=parser.standardize_metadata)
        fmt.spacegroups = parse_spacegroups(getattr(parser, "spacegroups", None), len(parser.reflection_files))
        return fmt

    def prep_dataset(self, mtz: Mtz):
        keys = self._guess_keys(mtz)
        cols = dict(mtz.columns)
        if self.dmin is not None and self.dmin > 0.0:                     # resolution cut (formatter.py:296-297); no d < 0 without one
            d = 1.0 / np.sqrt(inv_d2(mtz.hkl(), mtz.cell))
            keep = ~(d < self.dmin)
            if not keep.all():
                cols = {k: v[keep] for k, v in cols.items()}
        return self._common_prep(cols, mtz, keys)

    def finalize(self, data, rac):
        metadata = self._metadata(data)
        H = np.stack([data["H"], data["K"], data["L"]], axis=1)
        refl_id = rac.to_refl_id(data["asu_id"], H)
        col = lambda v, t: np.asarray(v).astype(t)[:, None]
        inputs = {"refl_id": col(refl_id, np.int64), "file_id": col(data["file_id"], np.int64), "image_id": col(data["image_id"], np.int64),
                  "metadata": metadata, "intensities": col(data["intensity"], np.float32), "uncertainties": col(data["uncertainty"], np.float32)}
        return self.pack_inputs(inputs), rac


class LaueFormatter(DataFormatter):
    def __init__(self, wavelength_key="Wavelength", intensity_key=None, uncertainty_key=None, image_key=None, metadata_keys=("dHKL",),
                 separate_outputs=False, anomalous=False, lam_min=None, lam_max=None, dmin=None, isigi_cutoff=None,
                 positional_encoding_keys=None, encoding_bit_depth=5, standardize=True):
        super().__init__(intensity_key, uncertainty_key, image_key, metadata_keys, separate_outputs, anomalous, dmin, isigi_cutoff,
                         positional_encoding_keys, encoding_bit_depth, standardize)
        self.wavelength_key, self.lam_min, self.lam_max = wavelength_key, lam_min, lam_max

    @classmethod
    def from_parser(cls, parser):
        lmin = lmax = None
        if parser.wavelength_range is not None:
            lmin, lmax = parser.wavelength_range
        pe = parser.positional_encoding_keys.split(",") if parser.positional_encoding_keys is not None else None
        fmt = cls(parser.wavelength_key, parser.intensity_key, parser.uncertainty_key, parser.image_key, parser.metadata_keys.split(","),
                  parser.separate_files, parser.anomalous, lmin, lmax, parser.dmin, parser.isigi_cutoff, pe,
                  parser.positional_encoding_frequencies, standardize=parser.standardize_metadata)
        fmt.spacegroups = parse_spacegroups(getattr(parser, "spacegroups", None), len(parser.reflection_files))
        return fmt

    def prep_dataset(self, mtz: Mtz):
        keys = self._guess_keys(mtz)
        wk = self.wavelength_key
        if wk not in mtz.columns:
            _key_error(wk, "Wavelength", "--wavelength-key", mtz)
        d = 1.0 / np.sqrt(inv_d2(mtz.hkl(), mtz.cell))
        dmin = self.dmin if self.dmin is not None else float(d.min())
        lam_min = self.lam_min if self.lam_min is not None else float(mtz.columns[wk].min())
        lam_max = self.lam_max if self.lam_max is not None else float(mtz.columns[wk].max())
        cols = expand_harmonics(dict(mtz.columns), mtz.cell, dmin, wk)
        keep = ~((cols[wk] < lam_min) | (cols[wk] > lam_max))
        cols = {k: v[keep] for k, v in cols.items()}
        return self._common_prep(cols, mtz, keys)

    def format_files(self, files):
        for f in files:
            if str(f).endswith(".stream"):
                raise ValueError("careless poly does not support .stream files. Use careless mono instead.")   # reference formatter.py:655-661
        return super().format_files(files)

    def finalize(self, data, rac):
        data = dict(data)
        data["harmonic_id"] = _ngroup(data["image_id"], data["H_0"], data["K_0"], data["L_0"])     # formatter.py:617
        metadata = self._metadata(data)
        H = np.stack([data["H"], data["K"], data["L"]], axis=1)
        refl_id = rac.to_refl_id(data["asu_id"], H)
        _, idx = np.unique(data["harmonic_id"], return_index=True)
        n = len(refl_id)
        iobs = np.ones((n, 1), dtype=np.float32); sigma = np.ones((n, 1), dtype=np.float32)           # padded with 1 (:637-640)
        iobs[: len(idx), 0] = data["intensity"][idx]
        sigma[: len(idx), 0] = data["uncertainty"][idx]
        col = lambda v, t: np.asarray(v).astype(t)[:, None]
        inputs = {"refl_id": col(refl_id, np.int64), "file_id": col(data["file_id"], np.int64), "image_id": col(data["image_id"], np.int64),
                  "metadata": metadata, "intensities": iobs, "uncertainties": sigma,
                  "wavelength": col(data[self.wavelength_key], np.float32), "harmonic_id": col(data["harmonic_id"], np.int64)}
        return self.pack_inputs(inputs), rac
