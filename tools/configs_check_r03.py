"""End-to-end timing of the tiled BASELINE configs on one MI355X (synthetic weights / frames), round 3:
  C3  config/splitting_hagen_indi.json: 10 x 2048^2 frames -> 490 tiles of 512^2, InDI n = 3, 8 tiles per batch:
      dataset tiles (fused crop + normalise) -> sampler -> stitch + RangeInvariantPsnr sums, everything inside the
      timed region, bf16 and fp32; the CPU oracle's tile rate on this host beside it.
  C5  config/splitting_hagen_indi_joint.json: JointIndi (two 1 -> 1 UNets) on 8 tiles, n = 3: the two loops on two HIP
      streams vs back to back, fp16 and fp32; the TimePredictor on 8 tiles.
Writes gpurun_out/r03_configs_check.json."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
torch.set_grad_enabled(False)
from diffsplitting_amd.core.logger import dict_to_nonedict
from diffsplitting_amd.data.split_dataset import DataLocation, SplitDatasetTiledPred
from diffsplitting_amd.data.tiled_predict import TileExchange
from diffsplitting_amd.model import networks
import bench

dev = torch.device("cuda:0")
blob = json.load(open(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "state_dict_keys.json")))
res = {}


def netg(cfgname, dtype):
    sec = json.loads(json.dumps(blob["model"][cfgname]))
    sec["compute_dtype"] = dtype
    opt = dict_to_nonedict({"model": sec, "phase": "val", "gpu_ids": [0], "distributed": False, "path": {"resume_state": None}})
    g = networks.define_G(opt).to(dev)
    sd = g.state_dict()
    rnd = bench.random_init_state_dict(list(sd.keys()), [tuple(v.shape) for v in sd.values()])
    g.load_state_dict({k: rnd.get(k, v) for k, v in sd.items()}, strict=False)
    g.set_new_noise_schedule({"n_timestep": 3}, dev)
    return g


rng = np.random.default_rng(0)
frames = (rng.random((10, 2048, 2048, 2), dtype=np.float32) * 1000.0).astype(np.float32)
ds = SplitDatasetTiledPred("Hagen", DataLocation(arrays=(frames[..., 0], frames[..., 1])), 512, grid_size=256,
                           max_qval=0.98, upper_clip=False, channel_weights=[1, 1], enable_transforms=False,
                           random_patching=False, input_from_normalized_target=False, device=dev)
plan = ds.plan
assert plan.total == 490
for dtype in ("bf16", "f32"):
    g = netg("splitting_hagen_indi", dtype)
    ids = list(range(plan.total))

    def run(score=True, stitch=True):
        gt = ds.normalized_target_frames() if score else None
        ex = TileExchange(plan, g.prediction_channels, dev, gt=gt) if stitch else None
        for i in range(0, len(ids), 8):
            chunk = ids[i:i + 8]
            b = ds.tiles(chunk)
            g.inference(b["input"], continuous=False, num_timesteps=3)
            if ex is not None:
                ex.add(g.last_full_batch, chunk)
        out = ex.finish() if ex is not None else None
        torch.cuda.synchronize()
        return out
    run()                                                      # warm: executors, graphs, device tables
    t0 = time.perf_counter(); out = run(); dt = time.perf_counter() - t0
    t0 = time.perf_counter(); run(score=False, stitch=False); dt_loops = time.perf_counter() - t0
    canvas, ps = out
    res[f"C3 hagen_indi {dtype}: 490 tiles of 512^2 (10 x 2048^2), n=3, 8 tiles/batch, END TO END (tiles + loops + stitch + PSNR)"] = dict(
        sec=dt, tiles_per_s=490 / dt, sec_without_stitch_and_gt=dt_loops, tiles_per_s_loops_only=490 / dt_loops,
        finite=bool(torch.isfinite(canvas).all()), psnr_db_mean=[float(v) for v in ps.mean(dim=0)])
    print(list(res.items())[-1], flush=True)
    del g

# CPU oracle on this host: one 512^2 tile, n = 3 (the reference runs tile after tile)
from oracle import samplers
from tests.util import golden_state_dict
sd, _ = golden_state_dict("unet_hagen_64")
osd = {"denoise_fn." + k: v for k, v in sd.items()}
cfg = dict(in_channel=2, out_channel=2, inner_channel=16, norm_groups=16, channel_mults=(1, 2, 4, 8), attn_res=(), res_blocks=1, image_size=32)
torch.set_num_threads(bench.host_threads())
x = torch.randn(1, 1, 512, 512)
samplers.indi_inference(osd, cfg, x, 1, 2)
t0 = time.perf_counter(); samplers.indi_inference(osd, cfg, x, 3, 2); dt = time.perf_counter() - t0
res["C3 CPU oracle (torch fp32), one 512^2 tile, n=3"] = dict(sec=dt, tiles_per_s=1 / dt, threads=torch.get_num_threads())
print(list(res.items())[-1], flush=True)

# C5: JointIndi, two streams vs back to back
x8 = torch.randn(8, 1, 512, 512, device=dev)
for dtype in ("f16", "f32"):
    g = netg("splitting_hagen_indi_joint", dtype)
    for conc in (True, False):
        g.concurrent = conc
        g.inference(x8, num_timesteps=3); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            g.inference(x8, num_timesteps=3)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 5
        res[f"C5 hagen_indi_joint {dtype}: JointIndi 8 tiles of 512^2, n=3, {'two streams' if conc else 'back to back'}"] = dict(
            sec=dt, tiles_per_s=8 / dt)
        print(list(res.items())[-1], flush=True)
    del g
from diffsplitting_amd.model.ddpm_modules.time_predictor import TimePredictor
tp = TimePredictor(in_channel=1, out_channel=1, inner_channel=16, norm_groups=16, channel_mults=(1, 2, 4, 8), attn_res=(),
                   res_blocks=1, image_size=32).to(dev)
tp(x8); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    tp(x8)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 5
res["C5 TimePredictor f32: 8 tiles of 512^2"] = dict(sec=dt, tiles_per_s=8 / dt)
print(list(res.items())[-1], flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/r03_configs_check.json", "w"), indent=1)
