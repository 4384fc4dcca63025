import sys, os, torch
sys.path.insert(0, os.getcwd())
import bench
from diffsplitting_amd import engine
torch.set_grad_enabled(False)
cfg = engine.make_cfg("sr3", **{k: bench.UNET[k] for k in ("in_channel", "out_channel", "inner_channel", "norm_groups", "channel_mults", "attn_res", "res_blocks", "image_size")})
for dtype in ("bf16", "f32"):
    eng = engine.UNetEngine(cfg, "sr3")
    eng.load_state_dict(bench.random_init_state_dict(eng.param_names, eng.param_shapes)); eng.finalize(dtype)
    g = torch.Generator().manual_seed(3)
    x = torch.randn(16, 6, 128, 128, generator=g).cuda(); t = (0.05 + 0.9 * torch.rand(16, 1, generator=g)).cuda()
    ref = eng.forward(x, t, cond_channels=3).clone()
    bad = 0
    for i in range(30):
        y = eng.forward(x, t, cond_channels=3)
        if not torch.equal(y, ref): bad += 1
    print(dtype, "forward B=16 x30 bitwise mismatches:", bad, "finite:", bool(torch.isfinite(ref).all()))
    # sampling loop determinism (graph), 40 steps, twice
    bufs, gam = engine.gaussian_buffers(bench.SCHEDULE)
    full = engine.gaussian_step_table(bufs, gam, "sr3", clip_denoised=True)
    import numpy as np
    idx = np.arange(40)
    tab = engine.StepTableHost(full.tcond[idx], c1=full.c1[idx], c2=full.c2[idx], sigma=full.sigma[idx], a=full.a[idx], b=full.b[idx], predict_eps=True, clip=True)
    cond = x[:, :3].contiguous(); x0 = engine.randn((16, 3, 128, 128), seed=5)
    o1 = eng.sample_loop(tab, x0.clone(), cond=cond, seed=7)[0].clone()
    o2 = eng.sample_loop(tab, x0.clone(), cond=cond, seed=7)[0].clone()
    print(dtype, "40-step loop bitwise equal:", bool(torch.equal(o1, o2)))
    del eng
