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


def results_tables(results: Dict[str, np.ndarray], rac: ReciprocalASUCollection) -> List[Dict[str, np.ndarray]]:
    """Split the per-reflection result arrays by ASU, attach Miller indices, drop unobserved reflections
    (reference manager.py:205-236; anomalous data stay one row per Friedel mate)."""
    out = []
    for i, _ in enumerate(rac):
        m = (rac.asu_ids == i) & (np.asarray(results["N"]) > 0)
        t = {"H": rac.Hall[m, 0], "K": rac.Hall[m, 1], "L": rac.Hall[m, 2]}
        for k in ("F", "SigF", "I", "SigI", "N"):
            t[k] = np.asarray(results[k])[m]
        for k in sorted(results):
            if k not in t and k != "observed":
                t[k] = np.asarray(results[k])[m]
        out.append(t)
    return out


def write_table_mtz(path: str, table: Dict[str, np.ndarray], asu: ReciprocalASU, types: Optional[Dict[str, str]] = None):
    types = dict(RESULT_TYPES if types is None else types)
    for k in table:
        types.setdefault(k, "R")
    write_mtz(path, table, types, asu.cell, asu.symops, getattr(asu, "spacegroup_name", "P 1"), getattr(asu, "spacegroup_number", 1))
