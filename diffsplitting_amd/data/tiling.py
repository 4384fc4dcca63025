"""Tile planning, device-side tile gather and stitch (C ABI: dsx_tile_plan,
dsx_tile_regions, dsx_tiles_gather, dsx_stitch)."""
import ctypes as C

import numpy as np
import torch

from .. import _lib
from .._lib import DsxError, check, lib

TRIM, PAD, SHIFT = _lib.TILING_TRIM, _lib.TILING_PAD, _lib.TILING_SHIFT


def _i64x3(v):
    return (C.c_int64 * 3)(*[int(x) for x in v])


class TilePlan:
    """All tiles of data (N,H,W) for a (1,g,g) grid and (1,p,p) patches —
    what ``SplitDatasetTiledPred.__init__`` sets up (split_dataset_tiledpred.py:9-24).
    Host-only integer math in the library; works without a GPU."""

    def __init__(self, data_shape, grid_shape, patch_shape, mode=SHIFT):
        if not (len(data_shape) == len(grid_shape) == len(patch_shape) == 3):
            raise DsxError("TilePlan handles 3-D data (N,H,W)")
        self.data_shape = tuple(int(v) for v in data_shape)
        self.grid_shape = tuple(int(v) for v in grid_shape)
        self.patch_shape = tuple(int(v) for v in patch_shape)
        self.mode = mode
        ds, gs, ps = _i64x3(data_shape), _i64x3(grid_shape), _i64x3(patch_shape)
        n = check(lib.dsx_tile_plan(ds, gs, ps, mode, None, None, 0))
        self.total = int(n)
        self.grid_start = np.zeros((self.total, 3), dtype=np.int64)
        self.patch_start = np.zeros((self.total, 3), dtype=np.int64)
        self.regions = np.zeros((self.total, 8), dtype=np.int32)
        if self.total:
            check(lib.dsx_tile_plan(ds, gs, ps, mode, self.grid_start.ctypes.data_as(C.POINTER(C.c_int64)),
                                    self.patch_start.ctypes.data_as(C.POINTER(C.c_int64)), self.total))
            check(lib.dsx_tile_regions(ds, gs, ps, mode, self.regions.ctypes.data_as(C.POINTER(C.c_int32)),
                                       self.total))

    def gather(self, frames, tile_ids=None):
        """frames: (N,H,W) fp32 CUDA tensor -> (count, ph, pw) tiles (all tiles by default)."""
        _lib.require_gpu()
        if not frames.is_cuda or frames.dtype != torch.float32 or tuple(frames.shape) != self.data_shape:
            raise DsxError(f"frames must be a float32 CUDA tensor of shape {self.data_shape}")
        frames = frames.contiguous()
        ids = np.arange(self.total, dtype=np.int64) if tile_ids is None else \
            np.ascontiguousarray(np.asarray(tile_ids, dtype=np.int64))
        if ids.size and (ids.min() < 0 or ids.max() >= self.total):
            raise DsxError("tile id out of range")
        out = torch.empty((len(ids), self.patch_shape[1], self.patch_shape[2]), dtype=torch.float32,
                          device=frames.device)
        if len(ids):
            check(lib.dsx_tiles_gather(C.c_void_p(frames.data_ptr()), _i64x3(self.data_shape),
                                       _i64x3(self.patch_shape),
                                       self.patch_start.ctypes.data_as(C.POINTER(C.c_int64)),
                                       ids.ctypes.data_as(C.POINTER(C.c_int64)), len(ids),
                                       C.c_void_p(out.data_ptr()),
                                       C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        return out

    def stitch(self, tiles, tile_ids=None, canvas=None):
        """tiles: (count, C, ph, pw) fp32 CUDA -> canvas (N,H,W,C), zero-initialised
        unless an existing canvas is passed (multi-GPU: every rank pastes its share)."""
        _lib.require_gpu()
        if not tiles.is_cuda or tiles.dtype != torch.float32 or tiles.dim() != 4:
            raise DsxError("tiles must be a (count,C,ph,pw) float32 CUDA tensor")
        tiles = tiles.contiguous()
        ids = np.arange(self.total, dtype=np.int64) if tile_ids is None else np.asarray(tile_ids, dtype=np.int64)
        if len(ids) != tiles.shape[0]:
            raise DsxError("one tile id per tile")
        Cn = tiles.shape[1]
        if tuple(tiles.shape[2:]) != self.patch_shape[1:]:
            raise DsxError("tile size does not match the plan")
        if canvas is None:
            canvas = torch.zeros(self.data_shape + (Cn,), dtype=torch.float32, device=tiles.device)
        reg = np.ascontiguousarray(self.regions[ids])
        if len(ids):
            check(lib.dsx_stitch(C.c_void_p(tiles.data_ptr()), len(ids), Cn, self.patch_shape[1],
                                 self.patch_shape[2], reg.ctypes.data_as(C.POINTER(C.c_int32)),
                                 C.c_void_p(canvas.data_ptr()), _i64x3(self.data_shape),
                                 C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        return canvas

    def stitch_with_psnr(self, tiles, gt):
        """Stitch ALL tiles (count == total, id order) and compute RangeInvariantPsnr (core/psnr.py:70-82) of every
        (frame, channel) against ``gt`` (N,H,W,C fp32 CUDA) from sums accumulated while pasting: no second pass
        over the canvas.  Returns (canvas (N,H,W,C), psnr (N,C) float64 on the device)."""
        _lib.require_gpu()
        if tiles.shape[0] != self.total:
            raise DsxError("stitch_with_psnr needs every tile of the plan")
        tiles = tiles.contiguous()
        Cn = tiles.shape[1]
        gt = gt.to(torch.float32).contiguous()
        if tuple(gt.shape) != self.data_shape + (Cn,) or not gt.is_cuda:
            raise DsxError(f"gt must be a CUDA tensor of shape {self.data_shape + (Cn,)}")
        canvas = torch.zeros(self.data_shape + (Cn,), dtype=torch.float32, device=tiles.device)
        gx = int(lib.dsx_stitch_psnr_blocks(self.patch_shape[1], self.patch_shape[2]))
        part = torch.zeros((self.total, gx, Cn, 8), dtype=torch.float64, device=tiles.device)
        check(lib.dsx_stitch_psnr(C.c_void_p(tiles.data_ptr()), self.total, Cn, self.patch_shape[1], self.patch_shape[2],
                                  self.regions.ctypes.data_as(C.POINTER(C.c_int32)), C.c_void_p(canvas.data_ptr()),
                                  _i64x3(self.data_shape), C.c_void_p(gt.data_ptr()), C.c_void_p(part.data_ptr()),
                                  C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        N = self.data_shape[0]
        per = self.total // N                                       # tile ids are frame-major (tiling_manager.py:145-154)
        p = part.view(N, per * gx, Cn, 8)
        s = p[..., :5].sum(dim=1)                                    # fixed order: reproducible
        gmin, gmax = p[..., 5].amin(dim=1), p[..., 6].amax(dim=1)
        return canvas, range_invariant_psnr_from_sums(s[..., 0], s[..., 1], s[..., 2], s[..., 3], s[..., 4], gmin, gmax,
                                                      float(self.data_shape[1] * self.data_shape[2]))


def range_invariant_psnr_from_sums(sp, spp, sg, sgg, sgp, gmin, gmax, n):
    """core/psnr.py:70-82 in closed form.  With g_ = (g - mean g) / std g (unbiased std, as torch.std) and
    p0 = p - mean p:  alpha = <g_, p0> / <p0, p0>,  mse = (<g_, g_> - <g_, p0>^2 / <p0, p0>) / n,
    PSNR = 20 log10( ((max g - min g) / std g) / sqrt(mse) )."""
    mean_g, mean_p = sg / n, sp / n
    var_g = (sgg - n * mean_g * mean_g) / (n - 1.0)
    std_g = torch.sqrt(var_g)
    gg = (sgg - n * mean_g * mean_g) / var_g                        # = n - 1
    gp = (sgp - n * mean_g * mean_p) / std_g
    pp = spp - n * mean_p * mean_p
    mse = (gg - gp * gp / pp) / n
    ra = (gmax - gmin) / std_g
    return 20.0 * torch.log10(ra / torch.sqrt(mse))
