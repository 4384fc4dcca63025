"""Batched tiled prediction: the notebook loop of the reference
(notebooks/EvaluateJointIndi.ipynb cells 23-26: one tile per ``test()`` call,
490 strictly serial calls, numpy stitch on the host) as a device-resident
pipeline: gather tiles on the GPU -> batches of tiles through the sampler ->
(multi-GPU: one all-gather) -> HIP stitch into the (N,H,W,C) canvas."""
import torch

from .. import parallel
from .tiling import TilePlan


class TileExchange:
    """The stitched output of a sharded run.  One rank: every batch is pasted straight into the canvas.  Several ranks:
    every batch's VALID REGIONS are packed into this rank's run of the exchange buffer (the crop of
    tile_stitcher.py:38-56 before the collective), one all-gather of the equal-sized runs (RCCL over xGMI) moves
    canvas bytes + padding instead of whole (C, p, p) tiles, and every rank pastes from the packed layout."""

    def __init__(self, plan, channels, device, group=None, gt=None):
        self.plan, self.C, self.group, self.gt = plan, int(channels), group, gt
        self.rank, self.world = parallel.rank(), parallel.world_size()
        self.canvas = self.part = self.flat = None
        if self.world == 1:
            self.canvas = torch.zeros(plan.data_shape + (self.C,), dtype=torch.float32, device=device)
            if gt is not None:
                self.part = plan.new_psnr_partials(self.C, device)
        else:
            self.flat = torch.zeros(plan.rank_stride(self.world, self.C), dtype=torch.float32, device=device)

    def add(self, tiles, ids):
        """``tiles`` (b, C, p, p): predictions of this rank's tile ids ``ids`` (a batch of its shard, in order)."""
        if len(ids) == 0:
            return
        if self.world == 1:
            if self.gt is not None:
                self.plan.stitch_psnr_into(tiles, ids, self.canvas, self.gt, self.part)
            else:
                self.plan.stitch(tiles, ids, self.canvas)
        else:
            self.plan.pack(tiles, self.world, ids[0], self.flat)

    def gathered_bytes(self):
        """Bytes every rank receives from the collective (world * rank_stride * 4)."""
        return 0 if self.world == 1 else self.world * self.flat.numel() * 4

    def finish(self):
        """-> canvas (N,H,W,C), or (canvas, psnr (N,C)) when a ground truth was given."""
        if self.world == 1:
            return self.canvas if self.gt is None else (self.canvas, self.plan.psnr_from_partials(self.part))
        full = parallel.all_gather_flat(self.flat, self.group)       # the path's only collective
        return self.plan.paste_packed(full, self.C, self.world, self.gt)


@torch.no_grad()
def predict_tiled(netG, frames_input, patch_size, grid_size=None, batch_tiles=8, sampler_kwargs=None,
                  group=None):
    """``frames_input``: (N,H,W) fp32 CUDA tensor, already normalised (the network input channel).
    Returns the stitched prediction (N,H,W,C) on every rank and the ``TilePlan``.

    ``netG`` is what ``define_G`` returns (InDI / JointIndi sampler); its full-batch
    output (``last_full_batch``) is used, not the single element the reference API returns."""
    if grid_size is None:
        grid_size = patch_size // 2                                   # split_dataset_tiledpred.py:13-14
    N, H, W = frames_input.shape
    plan = TilePlan((N, H, W), (1, grid_size, grid_size), (1, patch_size, patch_size))
    rank, world = parallel.rank(), parallel.world_size()
    ids = parallel.shard_ids(plan.total, rank, world)
    kw = dict(sampler_kwargs or {})
    ex = TileExchange(plan, netG.prediction_channels, frames_input.device, group)
    for i in range(0, len(ids), batch_tiles):
        chunk = ids[i:i + batch_tiles]
        tiles = plan.gather(frames_input, chunk).unsqueeze(1)         # (b,1,p,p)
        netG.inference(tiles, continuous=False, **kw)
        ex.add(netG.last_full_batch, chunk)
    return ex.finish(), plan
