"""Oracle: tile enumeration and stitching (integer / copy work, numpy).

Follows ``data/tiling_manager.py`` and ``data/tile_stitcher.py``.  Test
infrastructure only (see ``oracle/__init__.py``).
"""
import math

import numpy as np

TRIM, PAD, SHIFT = 0, 1, 2  # tiling_manager.py:6-12


class TilePlan:
    """Restatement of ``TileIndexManager`` (tiling_manager.py:14-154)."""

    def __init__(self, data_shape, grid_shape, patch_shape, mode=SHIFT):
        self.data_shape = tuple(int(v) for v in data_shape)
        self.grid_shape = tuple(int(v) for v in grid_shape)
        self.patch_shape = tuple(int(v) for v in patch_shape)
        self.mode = mode
        assert len(self.data_shape) == len(self.grid_shape) == len(self.patch_shape)
        for d, (p, g) in enumerate(zip(self.patch_shape, self.grid_shape)):  # :21-29
            if p - g < 0 or (p - g) % 2:
                raise ValueError(f"bad patch/grid in dim {d}")

    def patch_offset(self):  # :31-32
        return tuple((p - g) // 2 for p, g in zip(self.patch_shape, self.grid_shape))

    def dim_count(self, dim):  # :34-50
        D, g, p = self.data_shape[dim], self.grid_shape[dim], self.patch_shape[dim]
        if g == 1 and p == 1:
            return D
        if self.mode == PAD:
            return int(math.ceil(D / g))
        if self.mode == SHIFT:
            return int(math.ceil((D - (p - g)) / g))
        return int(math.floor((D - (p - g)) / g))

    def grid_count(self, dim):  # :58-68 (tiles per unit index of `dim`)
        n = 1
        for d in range(dim + 1, len(self.data_shape)):
            n *= self.dim_count(d)
        return n

    def total(self):  # :52-56
        return self.grid_count(0) * self.dim_count(0)

    def grid_start(self, dim, k):  # :121-143
        D, g, p = self.data_shape[dim], self.grid_shape[dim], self.patch_shape[dim]
        ex = (p - g) // 2
        if g == 1 and p == 1:
            return k
        if self.mode == PAD:
            return k * g
        if self.mode == TRIM:
            return k * g + ex
        if k < self.dim_count(dim) - 1:
            return k * g + ex
        return D - g - ex

    def location(self, idx):  # :145-154
        loc = []
        for d in range(len(self.data_shape)):
            gc = self.grid_count(d)
            loc.append(self.grid_start(d, idx // gc))
            idx %= gc
        return tuple(loc)

    def patch_location(self, idx):  # :106-112
        return tuple(l - o for l, o in zip(self.location(idx), self.patch_offset()))

    def valid_region(self, idx):
        """tile_stitcher.py:26-56: (vgs, vge, rs, re) for one tile."""
        gs = np.array(self.location(idx), dtype=int)
        ge = gs + np.array(self.grid_shape)
        ps = gs - np.array(self.patch_offset())
        pe = ps + np.array(self.patch_shape)
        vgs, vge = gs.copy(), ge.copy()
        if self.mode == SHIFT:
            for d in range(len(gs)):
                if ps[d] == 0:
                    vgs[d] = 0
                if pe[d] == self.data_shape[d]:
                    vge[d] = self.data_shape[d]
        rs = vgs - ps
        re = rs + (vge - vgs)
        return vgs, vge, rs, re


def paste_region(plan, idx):
    """The part of tile ``idx``'s valid region that survives stitch_predictions' sequential paste
    (tile_stitcher.py:68-80: a later tile overwrites): the shifted last tile of a ragged extent re-covers a strip of
    the tile before it along that dimension.  Returns (vgs, vge, rs, re) like valid_region."""
    vgs, vge, rs, re = plan.valid_region(idx)
    vge = vge.copy()
    cy, cx = plan.dim_count(1), plan.dim_count(2)
    iy, ix = (idx // cx) % cy, idx % cx
    if cy >= 2 and iy == cy - 2:
        nxt = plan.valid_region(idx + cx)[0]
        vge[1] = max(vgs[1], min(vge[1], nxt[1]))
    if cx >= 2 and ix == cx - 2:
        nxt = plan.valid_region(idx + 1)[0]
        vge[2] = max(vgs[2], min(vge[2], nxt[2]))
    return vgs, vge, rs, rs + (vge - vgs)


def stitch(predictions, plan):
    """tile_stitcher.py:10-81 for 3-D data (N,H,W): predictions (T,C,ph,pw) -> (N,H,W,C)."""
    out = np.zeros(list(plan.data_shape) + [predictions.shape[1]], dtype=predictions.dtype)
    for i in range(predictions.shape[0]):
        vgs, vge, rs, re = plan.valid_region(i)
        for c in range(predictions.shape[1]):
            out[vgs[0]:vge[0], vgs[1]:vge[1], vgs[2]:vge[2], c] = \
                predictions[i][c, rs[1]:re[1], rs[2]:re[2]]
    return out


def extract_patch(frames, plan, idx):
    """What SplitDatasetTiledPred yields for tile ``idx`` before normalisation:
    the (C, p, p) crop at ``patch_location`` (split_dataset_tiledpred.py:30-32 with
    split_dataset.py patch extraction).  ``frames``: (N,H,W,C)."""
    n, y, x = plan.patch_location(idx)
    p = plan.patch_shape
    return np.moveaxis(frames[n, y:y + p[1], x:x + p[2], :], -1, 0)


# ---- the cropped multi-rank exchange: valid regions packed per rank, one all-gather, paste --------------
def pack_layout(plan, world):
    """Pixel offset of every tile inside its rank's packed run (rank = id % world, tiles in id order) and the pixels
    of every rank's run: the crop of tile_stitcher.py:38-56 applied before the collective."""
    T = plan.total()
    off = np.zeros(T, dtype=np.int64)
    runs = np.zeros(world, dtype=np.int64)
    for i in range(T):
        vgs, vge, _, _ = paste_region(plan, i)
        off[i] = runs[i % world]
        runs[i % world] += int(vge[1] - vgs[1]) * int(vge[2] - vgs[2])
    return off, runs


def pack_rank(predictions, ids, plan, off, run_elems):
    """predictions (len(ids), C, ph, pw) of the tiles ``ids`` of one rank -> its flat run ([C][h][w] per tile)."""
    C = predictions.shape[1] if len(ids) else 1
    flat = np.zeros(run_elems, dtype=np.float32)
    for k, i in enumerate(ids):
        _, _, rs, re = paste_region(plan, i)
        reg = predictions[k][:, rs[1]:re[1], rs[2]:re[2]]
        flat[off[i] * C:off[i] * C + reg.size] = reg.reshape(-1)
    return flat


def paste_packed(flat_all, plan, off, C):
    """flat_all (world, run_elems) -> canvas (N,H,W,C)."""
    world = flat_all.shape[0]
    out = np.zeros(list(plan.data_shape) + [C], dtype=np.float32)
    for i in range(plan.total()):
        vgs, vge, _, _ = paste_region(plan, i)
        h, w = int(vge[1] - vgs[1]), int(vge[2] - vgs[2])
        reg = flat_all[i % world, off[i] * C:off[i] * C + C * h * w].reshape(C, h, w)
        out[vgs[0], vgs[1]:vge[1], vgs[2]:vge[2], :] = np.moveaxis(reg, 0, -1)
    return out
