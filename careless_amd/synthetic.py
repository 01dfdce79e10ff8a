"""Deterministic synthetic careless inputs: plain numpy, shared by bench.py, the smoke test
and the parity tests.  Produces the reference's `inputs` columns with the reference's dtypes (ids int64, data float32,
careless/io/formatter.py:382-394) plus the per-reflection constants the prior needs.

`positional_encoding` and `standardize_metadata` restate careless/utils/positional_encoding.py:3-17 and
careless/io/formatter.py:41-57 (host-side preprocessing that defines the metadata columns the scaler sees).
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np


def positional_encoding(x: np.ndarray, L: int, dtype=np.float64) -> np.ndarray:
    """`careless/utils/positional_encoding.py:3-17`: each column min-max scaled to [-1, 1]; angles
    pi 2^l p ordered column-major-then-frequency; output = [cos(all angles), sin(all angles)].
    The arithmetic is float64 whatever `dtype` the result is stored in (the formatter keeps float32 metadata); row blocks run on a few
    host threads (numpy's ufuncs release the interpreter lock) -- 10 M rows x 2 keys x 4 frequencies took 2.4 s on one."""
    p = np.asarray(x, dtype=np.float64)
    n, c = p.shape
    lo, hi = p.min(0), p.max(0)
    freqs = np.pi * 2.0 ** np.arange(L, dtype=np.float64)
    K = c * L
    out = np.empty((n, 2 * K), dtype=dtype)

    def block(a, b):
        q = 2.0 * (p[a:b] - lo) / (hi - lo) - 1.0
        ang = (q[:, :, None] * freqs[None, None, :]).reshape(b - a, K)
        np.cos(ang, out=out[a:b, :K], casting="same_kind")
        np.sin(ang, out=out[a:b, K:], casting="same_kind")

    step = 1 << 16
    if n <= 4 * step:
        block(0, n)
    else:
        import os
        from concurrent.futures import ThreadPoolExecutor
        try:
            cores = len(os.sched_getaffinity(0))
        except AttributeError:
            cores = os.cpu_count() or 1
        with ThreadPoolExecutor(max(1, min(16, cores))) as ex:
            list(ex.map(lambda a: block(a, min(n, a + step)), range(0, n, step)))
    return out


def standardize_metadata(m: np.ndarray) -> np.ndarray:
    """`careless/io/formatter.py:41-57`: per-column (x - mean) / std."""
    m = np.asarray(m, dtype=np.float64)
    return (m - m.mean(0)) / m.std(0)


def make_synthetic(N: int, R: Optional[int] = None, d0: int = 5, posenc: bool = False, posenc_L: int = 4,
                   outliers: bool = False, seed: int = 1234, n_images: Optional[int] = None, posenc_keys: int = 2):
    """Deterministic synthetic mono problem of SURVEY 8(d).  Returns a dict of numpy arrays in the reference's
    dtypes: ids int64, data float32 (io/formatter.py:382-394)."""
    rng = np.random.default_rng(seed)
    if R is None:
        R = max(1, N // 32)
    centric = rng.random(R) < 0.1
    mult = rng.choice(np.array([1.0, 2.0, 3.0, 4.0, 6.0]), size=R, p=[0.9, 0.05, 0.02, 0.02, 0.01])
    # F_true ~ Wilson(centric, eps, Sigma=1)
    sig = np.sqrt(mult)
    f_c = np.abs(rng.normal(size=R)) * sig
    f_a = sig * np.sqrt(-np.log1p(-rng.random(R)))
    f_true = np.where(centric, f_c, f_a)
    refl_id = np.concatenate([np.arange(min(R, N)), rng.integers(0, R, size=max(0, N - R))]).astype(np.int64)
    M = n_images if n_images is not None else max(1, N // 1000)
    image_id = np.sort(rng.integers(0, M, size=N)).astype(np.int64)
    inv_d2 = rng.uniform(0.01, 0.25, size=N)
    hkl = rng.integers(-40, 40, size=(N, 3)).astype(np.float64)
    extra = rng.uniform(0, 1, size=(N, max(0, d0 - 4)))
    raw = np.concatenate([inv_d2[:, None], hkl, extra], axis=1)[:, :d0]
    meta = standardize_metadata(raw)
    if posenc:
        xy = rng.uniform(0, 2048, size=(N, posenc_keys))       # (2 keys: X, Y -- + 16 columns at L = 4; 4 keys: + 32)
        meta = np.concatenate([meta, positional_encoding(xy, posenc_L)], axis=1)
    g = np.exp(rng.normal(0.0, 0.2, size=M))
    K = np.exp(-5.0 * inv_d2) * g[image_id]
    i_true = K * f_true[refl_id] ** 2 * 1e3
    sigi = np.sqrt(i_true + 25.0)
    iobs = i_true + sigi * rng.normal(size=N)
    if outliers:
        n_out = int(0.02 * N)
        idx = rng.choice(N, size=n_out, replace=False)
        iobs[idx] = i_true[idx] + 10.0 * sigi[idx] * rng.standard_t(2.0, size=n_out)
    return dict(
        refl_id=refl_id, image_id=image_id, file_id=np.zeros(N, dtype=np.int64),
        metadata=meta.astype(np.float32), iobs=iobs.astype(np.float32), sigiobs=sigi.astype(np.float32),
        centric=centric, multiplicity=mult.astype(np.float32), n_images=M, n_refl=R,
        f_true=f_true.astype(np.float32),        # the amplitudes the intensities were generated from (recovery tests; no kernel sees them)
    )


def make_synthetic_double_wilson(N: int, R_half: Optional[int] = None, r: float = 0.9, p_absent: float = 0.05, seed: int = 1234,
                                 **kw) -> Dict:
    """Two ASUs of equal size: ASU 0 is the root, every reflection of ASU 1 has its ASU-0 twin as parent
    with probability 1 - p_absent (else -1); observations are split between the two by `file_id`."""
    R_half = R_half if R_half is not None else max(1, N // 64)
    d = make_synthetic(N, R=2 * R_half, seed=seed, **kw)
    rng = np.random.default_rng(seed + 17)
    R = 2 * R_half
    # ASU 1 shares centric / multiplicity with its parent twin
    d["centric"] = np.concatenate([d["centric"][:R_half], d["centric"][:R_half]])
    d["multiplicity"] = np.concatenate([d["multiplicity"][:R_half], d["multiplicity"][:R_half]])
    parent = np.arange(R_half)
    parent = np.where(rng.random(R_half) < p_absent, -1, parent)
    d["parent_ids"] = np.concatenate([np.arange(R_half), parent]).astype(np.int64)   # root rows hold their own id (reference :114)
    d["root"] = np.concatenate([np.ones(R_half, bool), np.zeros(R_half, bool)])
    d["asu_ids"] = np.concatenate([np.zeros(R_half, np.int64), np.ones(R_half, np.int64)])
    d["dw_r"] = np.array([0.0, r], dtype=np.float32)
    d["file_id"] = (d["refl_id"] >= R_half).astype(np.int64)
    return d


def make_synthetic_laue(N: int, R: Optional[int] = None, seed: int = 1234, n_images: Optional[int] = None, pad: float = 1.0) -> Dict:
    """Polychromatic problem: N expanded rows, harmonic multiplicity {1: .80, 2: .15, 3: .05}; the rows of a
    group share image and observed intensity; Iobs / SigIobs live in slots [0,G) ordered by harmonic id and are padded with `pad`
    beyond (careless/io/formatter.py:617,637-640); metadata gets the wavelength as a sixth column."""
    rng = np.random.default_rng(seed)
    ks = rng.choice(np.array([1, 2, 3]), size=N, p=[0.8, 0.15, 0.05])       # more than enough groups; cut where the rows run out
    cum = np.cumsum(ks)
    G = int(np.searchsorted(cum, N, side="left")) + 1
    sizes = ks[:G].copy()
    sizes[-1] -= int(cum[G - 1]) - N
    hid = np.repeat(np.arange(G), sizes).astype(np.int64)
    d = make_synthetic(N, R=R, d0=5, seed=seed, n_images=n_images)
    # rows of one harmonic group come from one image
    img_g = np.sort(rng.integers(0, d["n_images"], size=G))
    d["image_id"] = img_g[hid].astype(np.int64)
    wl = rng.uniform(1.0, 1.2, size=N)
    meta = np.concatenate([np.asarray(d["metadata"], dtype=np.float64), ((wl - wl.mean()) / wl.std())[:, None]], axis=1)
    d["metadata"] = meta.astype(np.float32)
    d["wavelength"] = wl.astype(np.float32)
    i_row = np.asarray(d["iobs"], dtype=np.float64)
    i_grp = np.bincount(hid, weights=np.maximum(i_row, 0.0), minlength=G)
    s_grp = np.sqrt(np.abs(i_grp) + 25.0)
    iobs = np.full(N, pad, dtype=np.float32)
    sig = np.full(N, pad, dtype=np.float32)
    iobs[:G] = (i_grp + s_grp * rng.normal(size=G)).astype(np.float32)
    sig[:G] = s_grp.astype(np.float32)
    d["iobs"], d["sigiobs"] = iobs, sig
    d["harmonic_id"] = hid
    d["n_groups"] = G
    return d
