"""Batched tiled prediction: the notebook loop of the reference
(notebooks/EvaluateJointIndi.ipynb cells 23-26: one tile per ``test()`` call,
490 strictly serial calls, numpy stitch on the host) as a device-resident
pipeline: gather tiles on the GPU -> batches of tiles through the sampler ->
(multi-GPU: one all-gather) -> HIP stitch into the (N,H,W,C) canvas."""
import torch

from .. import parallel
from .tiling import TilePlan


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
    outs = []
    for i in range(0, len(ids), batch_tiles):
        chunk = ids[i:i + batch_tiles]
        tiles = plan.gather(frames_input, chunk).unsqueeze(1)         # (b,1,p,p)
        netG.inference(tiles, continuous=False, **kw)
        outs.append(netG.last_full_batch.clone())
    C = netG.last_full_batch.shape[1] if outs else 1
    local = torch.cat(outs, dim=0) if outs else torch.zeros((0, C, patch_size, patch_size),
                                                            device=frames_input.device)
    full = parallel.all_gather_tiles(local, plan.total, group)
    return plan.stitch(full), plan
