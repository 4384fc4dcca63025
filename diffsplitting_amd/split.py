#!/usr/bin/env python3
"""``split.py`` — the reference's entry point (split.py:75-85 flags) for the
validation / tiled-prediction path on MI355X:

    python -m diffsplitting_amd.split -c config/splitting_hagen_indi.json -p val -gpu 0
    torchrun --nproc-per-node 8 -m diffsplitting_amd.split -c <config> -p val -gpu 0,1,2,3,4,5,6,7

The config file is consumed unchanged (JSON with // comments).  Frames come
from ``--frames <file.npy>`` ((N,H,W,2) raw channels) or are synthesised when
the config's data paths do not exist on this machine.  ``-p train`` is refused:
the engine is inference-only.
"""
import argparse
import logging
import os
import sys
import time

import numpy as np
import torch

from . import parallel
from .core import logger as Logger
from .data.split_dataset import DataLocation, SplitDatasetTiledPred
from .data.tiled_predict import TileExchange
from .model import create_model


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("-c", "--config", type=str, required=True, help="JSON file for configuration")
    ap.add_argument("-p", "--phase", type=str, choices=["train", "val"], default="val")
    ap.add_argument("-gpu", "--gpu_ids", type=str, default="0")
    ap.add_argument("-debug", "-d", action="store_true")
    ap.add_argument("-enable_wandb", action="store_true")
    ap.add_argument("-rootdir", type=str, default=".")
    ap.add_argument("--frames", type=str, default=None, help=".npy with (N,H,W,2) raw channel frames")
    ap.add_argument("--synthetic", type=str, default="2,512,512", help="N,H,W of synthetic frames")
    ap.add_argument("--steps", type=int, default=None, help="override beta_schedule.val.n_timestep")
    ap.add_argument("--batch-tiles", type=int, default=8)
    ap.add_argument("--dtype", type=str, default=None, choices=["f32", "bf16", "f16"])
    ap.add_argument("--gpus", type=int, default=None,
                    help="ranks to run (one process per GPU); default: the number of ids in -gpu.  Without a "
                         "launcher (torchrun) the ranks are started here")
    args = ap.parse_args(argv)
    if args.phase == "train":
        raise SystemExit("training is out of scope of the MI355X sampling engine; use -p val")
    n_ranks = args.gpus if args.gpus is not None else len(str(args.gpu_ids).split(","))
    if argv is None and parallel.needs_self_launch(n_ranks):
        # fresh child processes, started before anything here touches the GPU (never an exec after HIP init)
        raise SystemExit(parallel.self_launch(n_ranks, ["-m", "diffsplitting_amd.split"] + sys.argv[1:]))

    rank, world = parallel.init()
    logging.basicConfig(level=logging.INFO if rank == 0 else logging.WARNING, format="%(asctime)s %(message)s")
    log = logging.getLogger("base")
    opt = Logger.parse(args)
    if args.dtype:
        opt["model"]["compute_dtype"] = args.dtype
    # -gpu selects the device(s): rank r of a torchrun launch drives gpu_ids[r] (the reference exports
    # CUDA_VISIBLE_DEVICES=gpu_ids instead, core/logger.py:59-65; mapping the index keeps one process per GPU
    # working without touching the environment after HIP may have been initialised)
    ids = list(opt["gpu_ids"] or [0])
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(ids[local % len(ids)])
    dev = torch.device("cuda", torch.cuda.current_device())
    torch.backends.cudnn.benchmark = True

    diffusion = create_model(opt)
    diffusion.set_new_noise_schedule(opt["model"]["beta_schedule"]["val"], schedule_phase="val")
    netG = diffusion.netG
    n_steps = args.steps or opt["model"]["beta_schedule"]["val"]["n_timestep"]

    if args.frames:
        frames = np.load(args.frames, allow_pickle=False).astype(np.float32)
    else:
        n, h, w = (int(v) for v in args.synthetic.split(","))
        rng = np.random.default_rng(0)
        frames = (rng.random((n, h, w, 2), dtype=np.float32) * 1000.0).astype(np.float32)    # raw detector counts
        log.info("no --frames given: using synthetic frames %s", frames.shape)
    # the validation dataset of split.get_datasets(opt, tiled_pred=True) (reference split.py:30-71) with the frames
    # resident on the GPU: quantile normalisation (compute_normalization_dict), ShiftBoundary tiling, normalised
    # tile batches cut by one HIP launch
    dsopt = opt["datasets"] or {}
    patch = (dsopt["val"] or {}).get("patch_size") if dsopt.get("val") else None
    patch = int(patch or dsopt.get("patch_size") or 512)
    patch = min(patch, frames.shape[1], frames.shape[2])
    which = opt["model"]["which_model_G"]
    val_set = SplitDatasetTiledPred("Hagen", DataLocation(arrays=(frames[..., 0], frames[..., 1])), patch,
                                    grid_size=patch // 2, target_channel_idx=dsopt.get("target_channel_idx"),
                                    max_qval=dsopt.get("max_qval") or 0.98, upper_clip=bool(dsopt.get("upper_clip")),
                                    channel_weights=dsopt.get("channel_weights"), enable_transforms=False,
                                    random_patching=False, input_from_normalized_target=(which == "joint_indi"),
                                    device=dev)
    plan = val_set.plan
    ids = parallel.shard_ids(plan.total, rank, world)

    torch.cuda.synchronize()
    t0 = time.perf_counter()
    # every rank holds the frames: the ground truth needs no tiles and no collective
    C_out = netG.prediction_channels
    gt = val_set.normalized_target_frames()                           # (N,H,W,C_target), normalised
    score = gt.shape[-1] == C_out and C_out <= 4
    ex = TileExchange(plan, C_out, dev, gt=gt if score else None)
    for i in range(0, len(ids), args.batch_tiles):
        chunk = ids[i:i + args.batch_tiles]
        batch = val_set.tiles(chunk)
        netG.inference(batch["input"], continuous=False, num_timesteps=n_steps)
        ex.add(netG.last_full_batch, chunk)
    res = ex.finish()                                                 # the path's only collective: cropped tiles
    pred, ps = res if score else (res, None)                          # metric accumulated while pasting
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if rank == 0:
        log.info("tiled prediction: %d tiles of %d^2, %d steps, %d GPU(s): %.3f s (%.1f tiles/s)",
                 plan.total, patch, n_steps, world, dt, plan.total / dt)
        if ps is not None:
            for c in range(ps.shape[1]):
                log.info("channel %d: RangeInvariantPsnr %.2f +- %.2f dB (random-init weights unless a checkpoint "
                         "was given in path.resume_state)", c, ps[:, c].mean().item(),
                         ps[:, c].std().item() if ps.shape[0] > 1 else 0.0)
    return pred


if __name__ == "__main__":
    main()
