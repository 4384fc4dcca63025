"""Functional + timing check of the other BASELINE configs on one MI355X (synthetic weights/inputs)."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import bench
from diffsplitting_amd import engine
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")

def build(flavour, unet, dtype):
    cfg = engine.make_cfg(flavour, unet["in_channel"], unet["out_channel"], unet["inner_channel"], unet["norm_groups"],
                          unet["channel_mults"], unet["attn_res"], unet["res_blocks"], unet["image_size"])
    eng = engine.UNetEngine(cfg, flavour)
    eng.load_state_dict(bench.random_init_state_dict(eng.param_names, eng.param_shapes))
    eng.finalize(dtype)
    return eng

def timeit(fn, n=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n

res = {}
# C3: splitting_hagen_indi — UNet 2->2, inner 16, mults [1,2,4,8], GN16, 512^2 tiles, n=3 InDI steps
hagen = dict(in_channel=2, out_channel=2, inner_channel=16, norm_groups=16, channel_mults=(1, 2, 4, 8), attn_res=(), res_blocks=1, image_size=32)
for dtype in ("f32", "bf16"):
    eng = build("ddpm", hagen, dtype)
    for B in (1, 8):
        x = torch.randn(B, 2, 512, 512, device=dev)
        tab = engine.indi_step_table(3, 1.0)
        dt = timeit(lambda: eng.sample_loop(tab, x.clone(), seed=1))
        gf = eng.flops(512, 512) * 3 * B / 1e9
        res[f"C3 hagen_indi {dtype} B={B} tiles of 512^2, 3 steps"] = dict(sec=dt, tiles_per_s=B / dt, tflops=gf / dt / 1e3,
                                                                       ws_gb=eng.workspace_bytes(B, 512, 512) / 1e9)
        print(list(res.items())[-1], flush=True)
    del eng
# C4: sr_sr3_64_512 — fp32, B=2, 512^2, inner 64, mults [1,2,4,8,16], GN16, rb=1, no attn_res
c4 = dict(in_channel=6, out_channel=3, inner_channel=64, norm_groups=16, channel_mults=(1, 2, 4, 8, 16), attn_res=(), res_blocks=1, image_size=512)
for dtype in ("f32", "bf16"):
    eng = build("sr3", c4, dtype)
    B = 2
    x = torch.randn(B, 6, 512, 512, device=dev); t = torch.rand(B, 1, device=dev)
    dt = timeit(lambda: eng.forward(x, t, cond_channels=3))
    y = eng.forward(x, t, cond_channels=3)
    gf = eng.flops(512, 512) * B / 1e9
    res[f"C4 sr_sr3_64_512 {dtype} B={B} one UNet forward"] = dict(sec=dt, ms=dt * 1e3, tflops=gf / dt / 1e3, finite=bool(torch.isfinite(y).all()),
                                                                 ws_gb=eng.workspace_bytes(B, 512, 512, 3) / 1e9, img_per_s_2000=B / (dt * 2000))
    print(list(res.items())[-1], flush=True)
    del eng
# C2 in fp32 (parity dtype)
eng = build("sr3", bench.UNET, "f32")
x = torch.randn(16, 6, 128, 128, device=dev); t = torch.rand(16, 1, device=dev)
dt = timeit(lambda: eng.forward(x, t, cond_channels=3), 5)
res["C2 sr_sr3_16_128 f32 B=16 one UNet forward"] = dict(ms=dt * 1e3, tflops=eng.flops(128, 128) * 16 / dt / 1e12, img_per_s_2000=16 / (dt * 2000))
print(list(res.items())[-1], flush=True)
# C1: cifar10_indi — 32^2, B=4, n=100
c1 = dict(in_channel=6, out_channel=6, inner_channel=16, norm_groups=16, channel_mults=(1, 2, 4, 8), attn_res=(), res_blocks=1, image_size=32)
eng = build("ddpm", c1, "f32")
x = torch.randn(4, 6, 32, 32, device=dev)
tab = engine.indi_step_table(100, 1.0)
dt = timeit(lambda: eng.sample_loop(tab, x.clone(), seed=1))
res["C1 cifar10_indi f32 B=4 32^2 100 steps"] = dict(sec=dt, img_per_s=4 / dt)
print(list(res.items())[-1], flush=True)
json.dump(res, open("gpurun_out/configs_check.json", "w"), indent=1)
