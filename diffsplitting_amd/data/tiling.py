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


def as_sequence(ids):
    """(first, stride, count) if ``ids`` is an arithmetic sequence with a positive stride (a rank's shard or a batch of
    it), else None."""
    ids = np.asarray(ids, dtype=np.int64).reshape(-1)
    if ids.size == 0:
        return 0, 1, 0
    if ids.size == 1:
        return int(ids[0]), 1, 1
    d = int(ids[1] - ids[0])
    if d < 1 or np.any(np.diff(ids) != d):
        return None
    return int(ids[0]), d, int(ids.size)


class TilePlan:
    """All tiles of data (N,H,W) for a (1,g,g) grid and (1,p,p) patches —
    what ``SplitDatasetTiledPred.__init__`` sets up (split_dataset_tiledpred.py:9-24).
    Host-only integer math in the library; works without a GPU.

    The device methods go through a ``dsx_tileplan`` handle: the patch starts and valid regions of every tile are
    uploaded once, calls name their tiles as an arithmetic id sequence (``ids`` of a shard or batch), nothing is
    allocated, copied or synchronised per call.  Arbitrary id lists fall back to the per-call upload forms."""

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
        self._h = None
        self._layouts = {}
        # the regions every paste uses: the valid regions, clipped where a later tile overwrites (a ragged extent's
        # shifted last tile), so that all tiles can be pasted at once with the sequential loop's result
        self.valid_regions = self.regions
        try:
            clipped = np.zeros((self.total, 8), dtype=np.int32)
            if self.total:
                check(lib.dsx_tileplan_regions(self.handle, clipped.ctypes.data_as(C.POINTER(C.c_int32)), self.total))
            self.regions = clipped
        except DsxError:
            pass                                     # tiling modes whose tiles leave the frames have no device plan

    # ---- the device-resident plan ------------------------------------------------------------------------
    @property
    def handle(self):
        if self._h is None:
            h = C.c_void_p()
            check(lib.dsx_tileplan_create(_i64x3(self.data_shape), _i64x3(self.grid_shape), _i64x3(self.patch_shape),
                                          self.mode, C.byref(h)))
            self._h = h
        return self._h

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h is not None:
            try:
                lib.dsx_tileplan_destroy(h)
            except Exception:
                pass

    @staticmethod
    def _stream():
        return C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def _ids(self, tile_ids):
        ids = np.arange(self.total, dtype=np.int64) if tile_ids is None else \
            np.ascontiguousarray(np.asarray(tile_ids, dtype=np.int64).reshape(-1))
        if ids.size and (ids.min() < 0 or ids.max() >= self.total):
            raise DsxError("tile id out of range")
        return ids

    def gather(self, frames, tile_ids=None):
        """frames: (N,H,W) fp32 CUDA tensor -> (count, ph, pw) tiles (all tiles by default)."""
        _lib.require_gpu()
        if not frames.is_cuda or frames.dtype != torch.float32 or tuple(frames.shape) != self.data_shape:
            raise DsxError(f"frames must be a float32 CUDA tensor of shape {self.data_shape}")
        frames = frames.contiguous()
        ids = self._ids(tile_ids)
        out = torch.empty((len(ids), self.patch_shape[1], self.patch_shape[2]), dtype=torch.float32,
                          device=frames.device)
        seq = as_sequence(ids)
        if len(ids) and seq is not None:
            check(lib.dsx_tileplan_gather(self.handle, C.c_void_p(frames.data_ptr()), seq[0], seq[1], seq[2],
                                          C.c_void_p(out.data_ptr()), self._stream()))
        elif len(ids):
            check(lib.dsx_tiles_gather(C.c_void_p(frames.data_ptr()), _i64x3(self.data_shape),
                                       _i64x3(self.patch_shape),
                                       self.patch_start.ctypes.data_as(C.POINTER(C.c_int64)),
                                       ids.ctypes.data_as(C.POINTER(C.c_int64)), len(ids),
                                       C.c_void_p(out.data_ptr()), self._stream()))
        return out

    def _check_tiles(self, tiles):
        if not tiles.is_cuda or tiles.dtype != torch.float32 or tiles.dim() != 4:
            raise DsxError("tiles must be a (count,C,ph,pw) float32 CUDA tensor")
        if tuple(tiles.shape[2:]) != self.patch_shape[1:]:
            raise DsxError("tile size does not match the plan")
        return tiles.contiguous()

    def stitch(self, tiles, tile_ids=None, canvas=None):
        """tiles: (count, C, ph, pw) fp32 CUDA -> canvas (N,H,W,C), zero-initialised
        unless an existing canvas is passed (every batch / rank pastes its share)."""
        _lib.require_gpu()
        tiles = self._check_tiles(tiles)
        ids = self._ids(tile_ids)
        if len(ids) != tiles.shape[0]:
            raise DsxError("one tile id per tile")
        Cn = tiles.shape[1]
        if canvas is None:
            canvas = torch.zeros(self.data_shape + (Cn,), dtype=torch.float32, device=tiles.device)
        seq = as_sequence(ids)
        if len(ids) and seq is not None:
            check(lib.dsx_tileplan_stitch(self.handle, C.c_void_p(tiles.data_ptr()), Cn, seq[0], seq[1], seq[2],
                                          C.c_void_p(canvas.data_ptr()), None, None, self._stream()))
        elif len(ids):
            reg = np.ascontiguousarray(self.regions[ids])
            check(lib.dsx_stitch(C.c_void_p(tiles.data_ptr()), len(ids), Cn, self.patch_shape[1],
                                 self.patch_shape[2], reg.ctypes.data_as(C.POINTER(C.c_int32)),
                                 C.c_void_p(canvas.data_ptr()), _i64x3(self.data_shape), self._stream()))
        return canvas

    # ---- RangeInvariantPsnr sums accumulated while pasting ---------------------------------------------------
    def psnr_blocks(self):
        return int(lib.dsx_stitch_psnr_blocks(self.patch_shape[1], self.patch_shape[2]))

    def new_psnr_partials(self, Cn, device):
        """Partial-sum rows of every tile of the plan, [total][blocks][C][8] float64 (filled batch by batch)."""
        return torch.zeros((self.total, self.psnr_blocks(), Cn, 8), dtype=torch.float64, device=device)

    def _check_gt(self, gt, Cn):
        gt = gt.to(torch.float32).contiguous()
        if tuple(gt.shape) != self.data_shape + (Cn,) or not gt.is_cuda:
            raise DsxError(f"gt must be a CUDA tensor of shape {self.data_shape + (Cn,)}")
        if Cn > 4:
            raise DsxError("the fused PSNR sums handle at most 4 channels")
        return gt

    def stitch_psnr_into(self, tiles, tile_ids, canvas, gt, partials):
        """Paste the tiles ``tile_ids`` (an arithmetic id sequence) into ``canvas`` and write their PSNR partial sums
        to the rows ``partials[tile_ids]``: a batch of a longer run, no (total, C, ph, pw) buffer needed."""
        _lib.require_gpu()
        tiles = self._check_tiles(tiles)
        seq = as_sequence(self._ids(tile_ids))
        if seq is None or seq[2] != tiles.shape[0]:
            raise DsxError("stitch_psnr_into needs one id per tile, ids in arithmetic sequence")
        if seq[2] == 0:
            return
        Cn = tiles.shape[1]
        gt = self._check_gt(gt, Cn)
        rows = torch.empty((seq[2],) + tuple(partials.shape[1:]), dtype=torch.float64, device=tiles.device)
        check(lib.dsx_tileplan_stitch(self.handle, C.c_void_p(tiles.data_ptr()), Cn, seq[0], seq[1], seq[2],
                                      C.c_void_p(canvas.data_ptr()), C.c_void_p(gt.data_ptr()),
                                      C.c_void_p(rows.data_ptr()), self._stream()))
        partials[seq[0]:seq[0] + (seq[2] - 1) * seq[1] + 1:seq[1]] = rows

    def psnr_from_partials(self, partials):
        """(N, C) RangeInvariantPsnr (core/psnr.py:70-82) from the partial-sum rows of all tiles."""
        N = self.data_shape[0]
        per = self.total // N                                       # tile ids are frame-major (tiling_manager.py:145-154)
        gx, Cn = partials.shape[1], partials.shape[2]
        p = partials.view(N, per * gx, Cn, 8)
        s = p[..., :5].sum(dim=1)                                    # fixed order: reproducible
        gmin, gmax = p[..., 5].amin(dim=1), p[..., 6].amax(dim=1)
        return range_invariant_psnr_from_sums(s[..., 0], s[..., 1], s[..., 2], s[..., 3], s[..., 4], gmin, gmax,
                                              float(self.data_shape[1] * self.data_shape[2]))

    def stitch_with_psnr(self, tiles, gt):
        """Stitch ALL tiles (count == total, id order) and compute RangeInvariantPsnr (core/psnr.py:70-82) of every
        (frame, channel) against ``gt`` (N,H,W,C fp32 CUDA) from sums accumulated while pasting: no second pass
        over the canvas.  Returns (canvas (N,H,W,C), psnr (N,C) float64 on the device)."""
        _lib.require_gpu()
        if tiles.shape[0] != self.total:
            raise DsxError("stitch_with_psnr needs every tile of the plan")
        Cn = tiles.shape[1]
        canvas = torch.zeros(self.data_shape + (Cn,), dtype=torch.float32, device=tiles.device)
        part = self.new_psnr_partials(Cn, tiles.device)
        self.stitch_psnr_into(tiles, np.arange(self.total), canvas, gt, part)
        return canvas, self.psnr_from_partials(part)

    # ---- cropped exchange for multi-GPU tiled prediction (tile_stitcher.py:38-56 before the collective) ----
    def pack_layout(self, world):
        """Host only: (pixel offset of every tile inside its rank's packed run [total], pixels of every rank's run
        [world]) for the sharding ``id % world``; multiply by C for elements."""
        world = int(world)
        if world not in self._layouts:
            off = np.zeros(self.total, dtype=np.int64)
            rp = np.zeros(world, dtype=np.int64)
            check(lib.dsx_tileplan_pack_layout(self.handle, world, off.ctypes.data_as(C.POINTER(C.c_int64)),
                                               rp.ctypes.data_as(C.POINTER(C.c_int64))))
            self._layouts[world] = (off, rp)
        return self._layouts[world]

    def rank_stride(self, world, Cn):
        """Elements of one rank's slot in the exchange buffer: the longest run (equal counts, as RCCL wants)."""
        return int(self.pack_layout(world)[1].max()) * int(Cn) if self.total else 0

    def pack(self, tiles, world, first, flat_rank):
        """Valid regions of the predicted tiles ``first, first + world, ...`` (tiles.shape[0] of them) -> their places
        in ``flat_rank``, the packed run of rank ``first % world`` (a 1-D fp32 CUDA tensor of rank_stride elements)."""
        _lib.require_gpu()
        tiles = self._check_tiles(tiles)
        Cn = tiles.shape[1]
        if flat_rank.dtype != torch.float32 or not flat_rank.is_cuda or not flat_rank.is_contiguous() or \
                flat_rank.numel() < int(self.pack_layout(world)[1][int(first) % int(world)]) * Cn:
            raise DsxError("flat_rank must be a contiguous float32 CUDA tensor that holds this rank's run")
        check(lib.dsx_tileplan_pack(self.handle, C.c_void_p(tiles.data_ptr()), Cn, int(world), int(first),
                                    int(tiles.shape[0]), C.c_void_p(flat_rank.data_ptr()), self._stream()))

    def paste_packed(self, flat_all, Cn, world, gt=None):
        """Every tile of the plan from the gathered exchange buffer (world, rank_stride) -> canvas (N,H,W,C); with
        ``gt`` also the (N, C) RangeInvariantPsnr.  Returns canvas or (canvas, psnr)."""
        _lib.require_gpu()
        flat_all = flat_all.contiguous()
        stride = flat_all.numel() // int(world)
        if flat_all.dtype != torch.float32 or not flat_all.is_cuda or stride < self.rank_stride(world, Cn):
            raise DsxError("flat_all must be a float32 CUDA tensor of world * rank_stride elements")
        canvas = torch.zeros(self.data_shape + (Cn,), dtype=torch.float32, device=flat_all.device)
        part = None
        if gt is not None:
            gt = self._check_gt(gt, Cn)
            part = self.new_psnr_partials(Cn, flat_all.device)
        check(lib.dsx_tileplan_paste_packed(self.handle, C.c_void_p(flat_all.data_ptr()), int(Cn), int(world), stride,
                                            C.c_void_p(canvas.data_ptr()),
                                            C.c_void_p(gt.data_ptr()) if gt is not None else None,
                                            C.c_void_p(part.data_ptr()) if part is not None else None, self._stream()))
        return canvas if gt is None else (canvas, self.psnr_from_partials(part))


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
