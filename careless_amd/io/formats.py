"""Flat-file containers around the path: the pre-formatted `.npz` input (what the formatters produce, so that formatting can run
anywhere and the GPU job starts from arrays), the training-history CSV (reference careless/careless.py:76-77) and the per-ASU
result / prediction tables as MTZ (careless.py:72-74, 86-100)."""
from __future__ import annotations

from typing import Dict, List, Optional

import numpy as np

from careless_amd.io.asu import ReciprocalASU, ReciprocalASUCollection
from careless_amd.io.mtz import write_mtz
from careless_amd.models.base import BaseModel


def save_inputs_npz(path: str, inputs, rac: ReciprocalASUCollection, **extra):
    """inputs tuple (BaseModel.input_index order) + everything needed to rebuild the ASU collection."""
    d = {f"inputs_{BaseModel.get_name_by_index(i)}": np.asarray(v) for i, v in enumerate(inputs)}
    d["n_asu"] = np.int64(len(rac))
    for i, a in enumerate(rac):
        d[f"asu{i}_cell"] = np.asarray(a.cell, dtype=np.float64)
        d[f"asu{i}_symops"] = np.asarray(a.symops)
        d[f"asu{i}_dmin"] = np.float64(a.dmin)
        d[f"asu{i}_anomalous"] = np.bool_(a.anomalous)
        d[f"asu{i}_spacegroup"] = np.asarray([a.spacegroup_name, str(a.spacegroup_number)])
    d.update(extra)
    np.savez_compressed(path, **d)


def load_inputs_npz(path: str):
    z = np.load(path, allow_pickle=False)
    inputs = ()
    for i in range(len(BaseModel.input_index)):
        k = f"inputs_{BaseModel.get_name_by_index(i)}"
        if k not in z:
            break
        inputs += (z[k],)
    asus = [ReciprocalASU(tuple(z[f"asu{i}_cell"]), [str(s) for s in z[f"asu{i}_symops"]], float(z[f"asu{i}_dmin"]),
                          bool(z[f"asu{i}_anomalous"]), str(z[f"asu{i}_spacegroup"][0]), int(z[f"asu{i}_spacegroup"][1]))
            for i in range(int(z["n_asu"]))]
    return inputs, ReciprocalASUCollection(asus)


def write_history_csv(path: str, history: Dict[str, List[float]]):
    """`rs.DataSet(history).to_csv(filename, index_label='step')` (careless.py:76-77)."""
    keys = list(history.keys())
    n = max((len(history[k]) for k in keys), default=0)
    with open(path, "w") as f:
        f.write(",".join(["step"] + keys) + "\n")
        for i in range(n):
            f.write(",".join([str(i)] + [repr(float(history[k][i])) if i < len(history[k]) else "" for k in keys]) + "\n")


RESULT_TYPES = {"H": "H", "K": "H", "L": "H", "F": "F", "SigF": "Q", "I": "J", "SigI": "Q", "N": "I"}
PREDICTION_TYPES = {"H": "H", "K": "H", "L": "H", "asu_id": "I", "image_id": "I", "file_id": "I", "test": "I", "Iobs": "J",
                    "SigIobs": "Q", "Ipred": "J", "SigIpred": "Q", "Scale": "J", "SigScale": "Q", "repeat": "I", "half": "I"}


ANOM_KEYS = ["F(+)", "SigF(+)", "F(-)", "SigF(-)", "I(+)", "SigI(+)", "I(-)", "SigI(-)", "N(+)", "N(-)"]   # the order PHENIX expects


def unstack_anomalous(table: Dict[str, np.ndarray], asu: ReciprocalASU) -> Dict[str, np.ndarray]:
    """One row per reflection of the NON-anomalous ASU with `X(+)` / `X(-)` columns (reciprocalspaceship's
    `DataSet.unstack_anomalous`, used by reference manager.py:238-248).  A reflection of the anomalous ASU is a Friedel-plus
    when it is also the representative of its orbit under rotations AND inversion; otherwise it is the minus mate of that
    representative.  Centric reflections carry the same values in both columns; a missing mate gives NaN."""
    H = np.stack([table["H"], table["K"], table["L"]], axis=1).astype(np.int64)
    rep = asu.ops.to_asu(H, anomalous=False)
    plus = np.all(rep == H, axis=1)
    centric, _, _ = asu.ops.describe(H)
    from careless_amd.io.asu import _key
    keys = _key(rep)
    uk, inv = np.unique(keys, return_inverse=True)
    first = np.zeros(len(uk), dtype=np.int64)
    first[inv] = np.arange(len(inv))
    out = {"H": rep[first, 0], "K": rep[first, 1], "L": rep[first, 2]}
    names = [k for k in table if k not in ("H", "K", "L")]
    for k in names:
        v = np.asarray(table[k], dtype=np.float32)
        vp = np.full(len(uk), np.nan, dtype=np.float32)
        vm = np.full(len(uk), np.nan, dtype=np.float32)
        vp[inv[plus]] = v[plus]
        vm[inv[~plus]] = v[~plus]
        cm = np.zeros(len(uk), dtype=bool)
        cm[inv[plus & centric]] = True
        vm[cm] = vp[cm]
        out[f"{k}(+)"], out[f"{k}(-)"] = vp, vm
    order = [k for k in ANOM_KEYS if k in out] + [k for k in out if k not in ANOM_KEYS and k not in ("H", "K", "L")]
    return {**{k: out[k] for k in ("H", "K", "L")}, **{k: out[k] for k in order}}


def results_tables(results: Dict[str, np.ndarray], rac: ReciprocalASUCollection) -> List[Dict[str, np.ndarray]]:
    """Split the per-reflection result arrays by ASU, attach Miller indices, drop unobserved reflections; anomalous ASUs are
    unstacked into `F(+)`, `F(-)`, ... columns (reference manager.py:205-250)."""
    out = []
    for i, asu in enumerate(rac):
        m = (rac.asu_ids == i) & (np.asarray(results["N"]) > 0)
        t = {"H": rac.Hall[m, 0], "K": rac.Hall[m, 1], "L": rac.Hall[m, 2]}
        for k in ("F", "SigF", "I", "SigI", "N"):
            t[k] = np.asarray(results[k])[m]
        for k in sorted(results):
            if k not in t and k != "observed":
                t[k] = np.asarray(results[k])[m]
        out.append(unstack_anomalous(t, asu) if asu.anomalous else t)
    return out


def write_table_mtz(path: str, table: Dict[str, np.ndarray], asu: ReciprocalASU, types: Optional[Dict[str, str]] = None):
    types = dict(RESULT_TYPES if types is None else types)
    anom = {"F": "G", "SigF": "L", "I": "K", "SigI": "M", "N": "I"}     # MTZ types of Friedel-separated columns
    for k in table:
        if k in types:
            continue
        base = k[:-3] if k.endswith(("(+)", "(-)")) else None
        types[k] = anom.get(base, "R") if base is not None else "R"
    write_mtz(path, table, types, asu.cell, asu.symops, getattr(asu, "spacegroup_name", "P 1"), getattr(asu, "spacegroup_number", 1))
