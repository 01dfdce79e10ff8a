"""Device images of observation sets and the data-parallel splits of the observation axis (host side, numpy + torch tensors):
`Shard` / `make_shard` / `owner_shard` / `laue_group_shard` (who takes which rows), `pack_by_image` / `pack_laue` (the packed orders
of the per-image-layer and single-pass Laue kernels), `ObsData` (one launch's arrays in HBM), `ObsChunks` (a shard cut into several
launches), `launch_row_limit`.  Split out of careless_amd/engine.py in round 4; the engine re-exports every name.

What the arrays replace in the reference: the `inputs` tuple of `BaseModel.input_index` order (careless/models/base.py:22-121) as
`formatter.py:354-400, 599-653` builds it -- the engine narrows the ids to int32 and stores the metadata feature-major.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional

import numpy as np
import torch

from careless_amd import _lib
from careless_amd.models.base import BaseModel

TILE = _lib.CL_MLP_TILE


def _np(x) -> np.ndarray:
    if torch.is_tensor(x):
        return x.detach().cpu().numpy()
    return np.asarray(x)


# ------------------------------------------------------------------------------------------------------------
# data-parallel sharding of the observation axis
# ------------------------------------------------------------------------------------------------------------
@dataclass
class Shard:
    rank: int
    world: int
    start: int        # first global observation of this rank
    stop: int
    kl_begin: int     # reflections whose KL term this rank owns
    kl_end: int
    owner: bool = False          # reflection-owner sharding: the rank holds EVERY observation of reflections [kl_begin, kl_end) and
    rows: Optional[np.ndarray] = None   # nothing else (`rows`: their global row numbers, ascending); start / stop are then 0 / len(rows)


def make_shard(n_obs: int, n_refl: int, rank: int = 0, world: int = 1) -> Shard:
    """Contiguous, near-equal split of the observations; the KL over the R reflections is split the same way so it
    is counted exactly once after the gradient all-reduce."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world of size {world}")
    from careless_amd.distributed import check_world
    check_world(n_obs, world)
    per = (n_obs + world - 1) // world
    if (world - 1) * per >= n_obs:          # ceil-sized chunks would leave the last ranks empty: floor-sized, remainder spread
        base, extra = divmod(n_obs, world)
        start = rank * base + min(rank, extra)
        stop = start + base + (1 if rank < extra else 0)
        rper = (n_refl + world - 1) // world
        return Shard(rank, world, start, stop, min(rank * rper, n_refl), min((rank + 1) * rper, n_refl))
    start, stop = min(rank * per, n_obs), min((rank + 1) * per, n_obs)
    rper = (n_refl + world - 1) // world
    return Shard(rank, world, start, stop, min(rank * rper, n_refl), min((rank + 1) * rper, n_refl))


def owner_bounds(refl_id: np.ndarray, n_refl: int, world: int) -> Optional[np.ndarray]:
    """Reflection ranges of a reflection-owner split: world + 1 boundaries in [0, n_refl] such that the ranges hold about the same
    number of OBSERVATIONS (the work) and every rank gets at least one reflection with at least one observation.  None when no such
    split exists (fewer observed reflections than ranks) -- the same answer on every rank, which then all use the row split."""
    counts = np.bincount(np.asarray(refl_id).reshape(-1).astype(np.int64), minlength=n_refl)
    cum = np.concatenate([[0], np.cumsum(counts)])
    n = int(cum[-1])
    b = np.array([int(np.searchsorted(cum, n * r / world, side="left")) for r in range(world)] + [n_refl], dtype=np.int64)
    b[0] = 0
    b = np.clip(b, 0, n_refl)
    if np.any(np.diff(b) <= 0) or np.any(np.diff(cum[b]) <= 0):
        return None
    return b


def owner_shard(refl_id: np.ndarray, n_refl: int, rank: int, world: int) -> Optional[Shard]:
    """The shard of `rank` in a reflection-owner split: reflections [r0, r1) and every observation of theirs.  All the
    terms of the loss that touch q(F_h) of an owned reflection -- its KL, its observations' likelihoods -- are then local: no other
    rank samples it, adds to its gradient or updates it, and the step's all-reduce carries the scaler's gradient only."""
    b = owner_bounds(refl_id, n_refl, world)
    if b is None:
        return None
    r0, r1 = int(b[rank]), int(b[rank + 1])
    rid = np.asarray(refl_id).reshape(-1)
    rows = np.nonzero((rid >= r0) & (rid < r1))[0]
    return Shard(rank, world, 0, int(len(rows)), r0, r1, True, rows)


def laue_group_shard(harmonic_id: np.ndarray, rank: int, world: int):
    """Laue shards must keep every harmonic group on one rank (the invariant the reference enforces for its train/test split,
    careless/io/manager.py:317-324).  Groups [g0, g1) go to `rank`, balanced by row count; the padded slots [G, N) are dealt out so
    that every rank has exactly as many slots as rows.  Returns (g0, g1, pad0, pad1)."""
    hid = np.asarray(harmonic_id).astype(np.int64)
    N = len(hid)
    G = int(hid.max()) + 1
    counts = np.bincount(hid, minlength=G)
    cum = np.concatenate([[0], np.cumsum(counts)])
    bounds = [int(np.searchsorted(cum, N * r / world, side="left")) for r in range(world)] + [G]
    bounds = np.clip(bounds, 0, G)
    if G < world or np.any(np.diff(bounds) <= 0):      # same test on every rank: all raise together, none waits in the collective
        raise ValueError(f"cannot shard {G} harmonic groups over {world} ranks: every rank needs at least one whole group")
    g0, g1 = int(bounds[rank]), int(bounds[rank + 1])
    pads = [int(cum[bounds[r + 1]] - cum[bounds[r]]) - int(bounds[r + 1] - bounds[r]) for r in range(world)]
    pad0 = G + int(sum(pads[:rank]))
    return g0, g1, pad0, pad0 + pads[rank]


# ------------------------------------------------------------------------------------------------------------
# device image of one set of observations (the training shard, or a validation set)
# ------------------------------------------------------------------------------------------------------------
def pack_by_image(image_id: np.ndarray):
    """Packed observation order of the per-image-layer kernel: rows grouped by image, every image padded to whole tiles.
    Returns (pos, n_pad, tile_img, row_map): pos[i] = packed position of row i; row_map[p] = row of packed position p or -1."""
    image_id = np.asarray(image_id).astype(np.int64)
    order = np.argsort(image_id, kind="stable")
    ids, counts = np.unique(image_id, return_counts=True)
    tiles = (counts + TILE - 1) // TILE
    base = np.concatenate([[0], np.cumsum(tiles)[:-1]]) * TILE          # first packed position of each image
    first = np.concatenate([[0], np.cumsum(counts)[:-1]])              # first sorted row of each image
    within = np.arange(len(image_id)) - np.repeat(first, counts)
    pos = np.empty(len(image_id), dtype=np.int64)
    pos[order] = np.repeat(base, counts) + within
    n_pad = int(tiles.sum()) * TILE
    row_map = np.full(n_pad, -1, dtype=np.int32)
    row_map[pos] = np.arange(len(image_id), dtype=np.int32)
    tile_img = np.repeat(ids, tiles).astype(np.int32)
    return pos, n_pad, tile_img, row_map


GRANULE = 16      # observations of one wave of the fused kernel


def pack_laue(harmonic_id: np.ndarray, image_id: np.ndarray, by_image: bool):
    """Packed order of the single-pass Laue kernel: the rows of a harmonic group are consecutive and inside one 16-row granule.
    Groups are laid out class by class (all groups of s rows of an image are contiguous: floor(16 / s) of them per granule, s rows
    apart, the rest of the granule padding -- triplets fill 15 of 16 rows, where padding them to four rows would fill 12), classes
    aligned to granules, images aligned to tiles when `by_image` (per-image layers).
    Returns (pos, n_pad, gmeta, tile_gmax, row_map, tile_img) or None when a group has more than 16 rows."""
    hid = np.asarray(harmonic_id).astype(np.int64)
    img = np.asarray(image_id).astype(np.int64)
    n = len(hid)
    order = np.argsort(hid, kind="stable")
    gid, first, size = np.unique(hid[order], return_index=True, return_counts=True)
    if size.max() > GRANULE:
        return None
    member = np.arange(n) - np.repeat(first, size)                       # member index of every sorted row
    gimg = img[order][first] if by_image else np.zeros(len(gid), dtype=np.int64)
    cls = size.astype(np.int64)                                          # class = exact group size
    per = GRANULE // cls                                                 # groups of that class per granule
    # regions = (image, class) pairs in sorted order; groups ranked inside their region
    key = gimg * 32 + cls
    gorder = np.argsort(key, kind="stable")
    rkey, rfirst, rcount = np.unique(key[gorder], return_index=True, return_counts=True)
    rcls = rkey % 32
    rimg = rkey // 32
    rrows = -(-rcount // (GRANULE // rcls)) * GRANULE                    # rows of a region: whole granules
    # image blocks aligned to tiles when the tiles must be single-image
    if by_image:
        ids, ifirst = np.unique(rimg, return_index=True)
        irows = np.add.reduceat(rrows, ifirst)
        irows = -(-irows // TILE) * TILE
        ibase = np.concatenate([[0], np.cumsum(irows)[:-1]])
        within = np.cumsum(rrows) - rrows - np.repeat((np.cumsum(rrows) - rrows)[ifirst], np.diff(np.concatenate([ifirst, [len(rimg)]])))
        rbase = np.repeat(ibase, np.diff(np.concatenate([ifirst, [len(rimg)]]))) + within
        n_pad = int(irows.sum())
        tile_img = np.repeat(ids, irows // TILE).astype(np.int32)
    else:
        rbase = np.cumsum(rrows) - rrows
        n_pad = int(-(-int(rrows.sum()) // TILE) * TILE)
        tile_img = None
    grank = np.empty(len(gid), dtype=np.int64)                           # rank of a group inside its region
    grank[gorder] = np.arange(len(gid)) - np.repeat(rfirst, rcount)
    gregion = np.empty(len(gid), dtype=np.int64)
    gregion[gorder] = np.repeat(np.arange(len(rkey)), rcount)
    gstart = rbase[gregion] + (grank // per) * GRANULE + (grank % per) * cls      # packed position of member 0
    pos = np.empty(n, dtype=np.int64)
    pos[order] = np.repeat(gstart, size) + member
    gmeta = np.zeros(n_pad, dtype=np.int32)
    gmeta[pos[order]] = (member | (np.repeat(size, size) << 8)).astype(np.int32)
    row_map = np.full(n_pad, -1, dtype=np.int32)
    row_map[pos] = np.arange(n, dtype=np.int32)
    tile_gmax = np.zeros(n_pad // TILE, dtype=np.int32)
    np.maximum.at(tile_gmax, pos[order] // TILE, np.repeat(size, size).astype(np.int32))
    return pos, n_pad, gmeta, tile_gmax, row_map, tile_img


class _ShardRows:
    """`x[sl]` of a (possibly memory-mapped) 2-D array as float32: the conversion happens on the selected rows only."""

    def __init__(self, a: np.ndarray, sl):
        self.a, self.shape = a, a.shape

    def __getitem__(self, sl) -> np.ndarray:
        return np.asarray(self.a[sl], dtype=np.float32)


class ObsData:
    """refl_id / image_id int32 [N], meta_t fp32 [rows][n_pad], iobs / sig fp32 [N], optional harmonic_id + Laue work
    buffers, and the per-launch workspace of the fused kernel (grid, gradient partials).  With `pack_images` (per-image
    layers) the arrays the fused kernel streams are in the packed order of `pack_by_image`."""

    def __init__(self, lib, inputs, start: int, stop: int, S: int, P: int, device, grid=None, n_refl=None, n_images=None,
                 laue_groups=None, pack_images: bool = False, laue_single_pass: bool = True, wide: bool = False, sort_images: bool = False,
                 rows: Optional[np.ndarray] = None):
        # Views of the caller's arrays (possibly memory-mapped files shared by the ranks of a node): only this shard's rows are
        # ever copied / converted -- a rank of an 8-GPU job does not hold eight copies' worth of the 50 M-observation problem
        refl_all = _np(BaseModel.get_refl_id(inputs)).reshape(-1)
        image_all = _np(BaseModel.get_image_id(inputs)).reshape(-1)
        meta_all = _np(BaseModel.get_metadata(inputs))
        meta_all = meta_all.reshape(len(refl_all), -1)
        iobs_all = _np(BaseModel.get_intensities(inputs)).reshape(-1)
        sig_all = _np(BaseModel.get_uncertainties(inputs)).reshape(-1)
        self.N_total = int(len(refl_all))
        stop = self.N_total if stop is None else stop
        self.laue = BaseModel.is_laue(inputs)
        self.rows = None                      # explicit row list when the shard is not a contiguous range
        if self.laue and laue_groups is not None:
            hid_all = _np(BaseModel.get_harmonic_id(inputs)).reshape(-1)
            g0, g1, pad0, pad1 = laue_groups
            self.rows = np.nonzero((hid_all >= g0) & (hid_all < g1))[0]
            sl = self.rows
            start, stop = 0, len(self.rows)
            # per-slot arrays of this shard: its own groups first, then its share of the padded slots (formatter.py:637-640)
            slot_idx = np.concatenate([np.arange(g0, g1), np.arange(pad0, pad1)])
            assert len(slot_idx) == len(self.rows)
            iobs_l, sig_l = iobs_all[slot_idx].astype(np.float32), sig_all[slot_idx].astype(np.float32)
        elif rows is not None:
            # monochromatic rows that are not a contiguous range (reflection-owner shard): stored in ascending row order; the global
            # row numbers key the in-kernel noise (noise_row) and pick the columns of injected noise
            if self.laue:
                raise ValueError("explicit rows are for monochromatic data (Laue shards go by harmonic group)")
            self.rows = np.asarray(rows, dtype=np.int64)
            sl = self.rows
            start, stop = 0, len(self.rows)
            iobs_l, sig_l = iobs_all[sl].astype(np.float32), sig_all[sl].astype(np.float32)
        else:
            sl = slice(start, stop)
            iobs_l, sig_l = iobs_all[sl].astype(np.float32), sig_all[sl].astype(np.float32)
        self.start, self.N = int(start), int(stop - start)
        if self.N <= 0:
            raise ValueError("empty observation shard")
        if n_refl is not None and refl_all.size and (refl_all.min() < 0 or refl_all.max() >= n_refl):
            raise ValueError("refl_id outside the range of the surrogate posterior")
        if n_images is not None and image_all.size and image_all.max() >= n_images:
            raise ValueError("image_id exceeds ImageScaler.max_images")
        metadata = _ShardRows(meta_all, sl)       # metadata[sl] -> this shard's rows as float32
        self.d = int(metadata.shape[1])
        self.tile_img = self.row_map = self.gmeta = self.tile_gmax = self.noise_row = None
        self.fused_laue = False
        rid_l, img_l = refl_all[sl].astype(np.int32), image_all[sl].astype(np.int32)
        lp = None
        if self.laue:
            hid_all0 = _np(BaseModel.get_harmonic_id(inputs)).reshape(-1).astype(np.int64)
            hl0 = hid_all0[sl] - (laue_groups[0] if (laue_groups is not None and self.rows is not None) else 0)
            if laue_single_pass:
                lp = pack_laue(hl0, img_l, by_image=pack_images)       # None: a group larger than a wave -> two-pass path
        if lp is not None:
            # single-pass Laue: everything the fused kernel streams is packed so that a harmonic group sits in one wave; the
            # group's observed intensity is replicated on its member rows; the padded slots keep their own small arrays
            pos, self.n_pad, gmeta, tile_gmax, row_map, tile_img = lp
            G = int(hl0.max()) + 1
            meta_t = np.zeros((int(lib.cl_mlp_meta_rows(self.d)), self.n_pad), dtype=np.float32)
            meta_t[: self.d, pos] = metadata[sl].T

            def packed(v, fill):
                out = np.full(self.n_pad, fill, dtype=v.dtype)
                out[pos] = v
                return out
            iobs_s, sig_s = np.asarray(iobs_l), np.asarray(sig_l)
            self.pad_iobs = torch.as_tensor(np.ascontiguousarray(iobs_s[G:]), device=device)
            self.pad_sig = torch.as_tensor(np.ascontiguousarray(sig_s[G:]), device=device)
            self.pad_iconv = torch.zeros(max(1, (len(iobs_s) - G) * S), dtype=torch.float32, device=device)
            # the formatter pads every empty slot with the same (1.0, 1.0) (reference io/formatter.py:637-640): their terms are one term times
            # their number -- the engine then launches the slot kernel on ONE slot with the weight multiplied (round 5: 29 us per step at 5 M rows)
            pi, ps = iobs_s[G:], sig_s[G:]
            self.pad_uniform = bool(len(pi) > 1 and np.all(pi == pi[0]) and np.all(ps == ps[0]))
            rid_l, img_l = packed(rid_l, -1), packed(img_l, 0)
            iobs_l, sig_l = packed(iobs_s[hl0], 0.0), packed(sig_s[hl0], 1.0)
            self.gmeta = torch.as_tensor(gmeta, device=device)
            self.tile_gmax = torch.as_tensor(tile_gmax, device=device)
            self.row_map = torch.as_tensor(row_map, device=device)
            self.tile_img = torch.as_tensor(tile_img, device=device) if tile_img is not None else None
            self.noise_row = None
            if self.rows is not None:                 # a shard of whole harmonic groups: rows are not a contiguous range
                nr = np.full(self.n_pad, 0, dtype=np.int32)
                nr[pos] = self.rows.astype(np.int32)
                self.noise_row = torch.as_tensor(nr, device=device)
            self.fused_laue = True
        elif pack_images:
            pos, self.n_pad, tile_img, row_map = pack_by_image(img_l)
            meta_t = np.zeros((int(lib.cl_mlp_meta_rows(self.d)), self.n_pad), dtype=np.float32)
            meta_t[: self.d, pos] = metadata[sl].T
            self.tile_img = torch.as_tensor(tile_img, device=device)
            self.row_map = torch.as_tensor(row_map, device=device)
            if not self.laue:           # the mono likelihood runs inside the fused kernel: its inputs are packed too
                def packed(v, fill):
                    out = np.full(self.n_pad, fill, dtype=v.dtype)
                    out[pos] = v
                    return out
                rid_l, img_l = packed(rid_l, -1), packed(img_l, 0)
                iobs_l, sig_l = packed(np.asarray(iobs_l), 0.0), packed(np.asarray(sig_l), 1.0)
        elif wide:
            # scaler wider than the fused kernel holds: layer-by-layer GEMMs on the row-major metadata (ElboEngine._data_term_wide)
            self.n_pad = ((self.N + TILE - 1) // TILE) * TILE
            meta_t = np.zeros((4, 4), dtype=np.float32)
            self.meta_ld = int(lib.cl_wide_ld(self.d))                 # [rows][ld]: the features, zero padding to a multiple of four
            rm = np.zeros((self.N, self.meta_ld), dtype=np.float32)
            rm[:, : self.d] = metadata[sl]
            self.perm = None
            if sort_images:
                # per-image layers on this path: the rows of an image must be consecutive (grouped GEMM kernels); everything per row
                # is stored in image order, `perm` maps the local order back to the caller's
                self.perm = np.argsort(img_l, kind="stable")
                rid_l, img_l, rm = rid_l[self.perm], img_l[self.perm], rm[self.perm]
                if not self.laue:                       # (mono: a row is its own slot; Laue keeps iobs / sig per slot)
                    iobs_l, sig_l = np.asarray(iobs_l)[self.perm], np.asarray(sig_l)[self.perm]
                self.img_seg = np.concatenate([[0], np.cumsum(np.bincount(img_l, minlength=int(n_images or (img_l.max() + 1))))]).astype(np.int64)
            self.meta_rm = torch.as_tensor(rm, device=device)
        else:
            self.n_pad = ((self.N + TILE - 1) // TILE) * TILE
            meta_t = np.zeros((int(lib.cl_mlp_meta_rows(self.d)), self.n_pad), dtype=np.float32)
            meta_t[: self.d, : self.N] = metadata[sl].T
            if rows is not None:             # (every plain-layout kernel reads the per-row noise key when it is given)
                nr = np.zeros(self.n_pad, dtype=np.int32)
                nr[: self.N] = self.rows.astype(np.int32)
                self.noise_row = torch.as_tensor(nr, device=device)
        self.refl_id = torch.as_tensor(rid_l, device=device)
        self.image_id = torch.as_tensor(img_l, device=device)
        self.meta_t = torch.as_tensor(meta_t, device=device)
        self.iobs = torch.as_tensor(np.ascontiguousarray(iobs_l), device=device)
        self.sig = torch.as_tensor(np.ascontiguousarray(sig_l), device=device)
        self.row_index = None
        if self.laue:
            hid = _np(BaseModel.get_harmonic_id(inputs)).reshape(-1).astype(np.int64)
            if hid.size and (hid.min() < 0 or hid.max() >= self.N_total):
                raise ValueError("harmonic_id outside [0, N)")
            hl = hid[sl]
            if self.rows is not None:
                hl = hl - laue_groups[0]
                self.row_index = torch.as_tensor(self.rows.astype(np.int64), device=device)
        if wide and not self.laue:
            hl = None                          # every row its own "harmonic group" (harmonic_id NULL): the slot kernels then ARE the mono likelihood
        elif wide and getattr(self, "perm", None) is not None:
            hl = hl[self.perm]
        if wide and getattr(self, "perm", None) is not None:
            # global rows in the stored (image) order: the noise key of every row, and which columns of an injected eta are its
            base_rows = self.rows if self.rows is not None else np.arange(self.start, self.start + self.N)
            self.rows = np.asarray(base_rows)[self.perm]
            self.row_index = torch.as_tensor(self.rows.astype(np.int64), device=device)
        if (self.laue and not self.fused_laue) or wide:
            self.harmonic_id = torch.as_tensor(hl.astype(np.int32), device=device) if hl is not None else None
            self.laue_loc = torch.empty(self.N, dtype=torch.float32, device=device)
            self.laue_sig = torch.empty(self.N, dtype=torch.float32, device=device)
            self.laue_iconv = torch.empty(self.N * S, dtype=torch.float32, device=device)
            self.laue_dO = torch.empty(self.N * 2, dtype=torch.float32, device=device)
        g = int(grid) if grid is not None else max(1, int(lib.cl_mlp_default_grid()))
        self.grid = min(g, self.n_pad // TILE)
        self.partials = torch.empty(0 if wide else self.grid * P, dtype=torch.float32, device=device)
        self.chain_act = self.chain_dact = None       # activations / their gradients at the block boundaries of a chained scaler

    def alloc_chain(self, lib, blocks, w, device):
        rows = int(lib.cl_mlp_meta_rows(w))
        n = len(blocks) - 1
        self.chain_act = [torch.zeros(rows, self.n_pad, dtype=torch.float32, device=device) for _ in range(n)]
        self.chain_dact = [torch.zeros(rows, self.n_pad, dtype=torch.float32, device=device) for _ in range(n)]


class ObsChunks:
    """A shard whose metadata image does not fit one launch of the fused kernels (their per-lane offsets are 32-bit: a launch
    addresses < 4 GiB of metadata; 50 M observations with positional encodings are 4.8 GB): consecutive `ObsData` pieces that are
    launched one after the other.  The reference is full-batch at any N (careless/models/merging/variational.py:255-256) and so is
    this: the weight-gradient partials reduce per launch into the same gradient, NLL and dz_f accumulate, the noise is keyed by
    the global row.  Only the plain observation layout is cut (packed layouts keep whole groups / images per launch)."""

    def __init__(self, children: List[ObsData]):
        self.children = children
        c0 = children[0]
        self.start, self.N, self.N_total, self.d = c0.start, sum(c.N for c in children), c0.N_total, c0.d
        self.n_pad, self.grid, self.partials = sum(c.n_pad for c in children), c0.grid, c0.partials
        self.laue, self.fused_laue, self.rows, self.row_map, self.row0 = False, False, None, None, 0
        for c in children:
            if getattr(c, "rows", None) is None:
                c.row0 = c.start - self.start       # first row of the piece inside the shard's eta / ipred arrays

    def alloc_chain(self, lib, blocks, w, device):
        self.children[0].alloc_chain(lib, blocks, w, device)         # the pieces run one after the other: one set of buffers
        for c in self.children[1:]:
            c.chain_act, c.chain_dact = self.children[0].chain_act, self.children[0].chain_dact


class _EmptyObs:
    """An owner-mode rank's share of a validation set in which none of its reflections occurs: nothing to launch."""

    def __init__(self, n_total: int):
        self.N, self.N_total, self.rows, self.empty = 0, n_total, np.zeros(0, dtype=np.int64), True


def launch_row_limit(d: int, S: int = 0) -> int:
    """Most rows of the plain layout one launch takes: 4 * cl_mlp_meta_rows(d) * n_pad bytes of metadata must stay below 4 GiB
    (include/careless_hip.h: return code -4), and -- `S` given: launches that address per-(row, sample) arrays with 32-bit lane
    offsets (the deterministic mode's dzf_obs) -- 4 * S * n_pad bytes as well.  CARELESS_HIP_MAX_LAUNCH_BYTES lowers the bound
    (tests of the chunked path)."""
    import os
    lim = int(os.environ.get("CARELESS_HIP_MAX_LAUNCH_BYTES", str((1 << 32) - (1 << 24))))
    per_row = 4 * max((d + 3) // 4 * 4, int(S))
    return max(TILE, lim // per_row // TILE * TILE)


