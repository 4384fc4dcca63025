"""Child program of test_two_rank_predict_tiled (one process per GPU, RCCL): every rank predicts its share of the
tiles, one all-gather, every rank stitches; rank 0 checks the result against the same prediction done on one GPU."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from diffsplitting_amd import parallel
    from diffsplitting_amd.data.tiled_predict import predict_tiled
    from diffsplitting_amd.model import networks
    from tests.test_gpu_boundary import _opt, _tiny_indi_section
    from tests.util import golden_state_dict
    torch.set_grad_enabled(False)
    rank, world = parallel.init(os.environ.get("DSX_DIST_BACKEND") or "nccl")   # (sets the rank's device)
    sd, _ = golden_state_dict("unet_hagen_64")
    sec = _tiny_indi_section()
    sec["unet"]["channel_multiplier"] = [1, 2, 4, 8]
    netG = networks.define_G(_opt(sec)).cuda()
    netG.load_state_dict({"denoise_fn." + k: v for k, v in sd.items()})
    netG.e = 0.0                                                   # no noise: the result does not depend on RNG streams
    frames = torch.from_numpy(np.random.default_rng(5).standard_normal((3, 96, 160)).astype(np.float32)).cuda()
    pred, plan = predict_tiled(netG, frames, patch_size=64, grid_size=32, batch_tiles=4, sampler_kwargs=dict(num_timesteps=2))
    # the same tiles on this rank alone (world-size-1 semantics): every id, no collective
    import torch.distributed as dist
    outs = []
    for i in range(0, plan.total, 4):
        tiles = plan.gather(frames, list(range(i, min(i + 4, plan.total)))).unsqueeze(1)
        netG.inference(tiles, continuous=False, num_timesteps=2)
        outs.append(netG.last_full_batch.clone())
    ref = plan.stitch(torch.cat(outs))
    ok = torch.equal(pred, ref)
    flag = torch.tensor([1 if ok else 0], device="cuda" if dist.get_backend() == "nccl" else "cpu")
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if rank == 0:
        print("PREDICT_TILED_OK" if int(flag.item()) == 1 else "PREDICT_TILED_MISMATCH", plan.total, world)
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if int(flag.item()) == 1 else 1)


if __name__ == "__main__":
    main()
