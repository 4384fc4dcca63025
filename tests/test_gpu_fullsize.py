"""Parity at the sizes BASELINE.json names (`-m gpu`; the oracle runs on the GPU box's host cores):

  C4  config/sr_sr3_64_512.json            the mults-[1,2,4,8,16] UNet at 512^2, B = 1 (2048->1024 convs, d = 1024
                                           attention over 1024 tokens, GN16)
  C3  config/splitting_hagen_indi.json     InDI.inference n = 3 on one 512^2 tile (attention over 4096 tokens, d = 128)
  C5  config/splitting_hagen_indi_joint.json   JointIndi n = 3 + TimePredictor on 512^2 tiles, fp32 and fp16

fp32: <= 1e-3 max-abs per pixel against the oracle (whole tensors) and against the digests of the reference's
own outputs (tests/golden/full_*.npz, written by oracle/gen_golden.py from /root/reference).  bf16 / fp16: PSNR of
the engine's output against the fp32 oracle output, threshold 35 dB (measured values are printed).
Also here: loop-level reduced-precision evidence and the regression tests for the two bugs of round 1.
"""
import numpy as np
import pytest
import torch

from oracle import cases, samplers
from oracle.unet import time_predictor_forward, unet_forward
from tests.gpu_util import DrawRecorder, build_engine, maxabs, psnr
from tests.util import golden_state_dict

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)
FP32_TOL = 1e-3
PSNR_MIN = 35.0


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


def _check_digest(y, g, atol=FP32_TOL):
    d = cases.digest(y)
    assert list(d["shape"]) == list(g["shape"])
    for k in ("crop", "grid", "corner"):
        assert maxabs(d[k], g[k]) <= atol, (k, maxabs(d[k], g[k]))
    npix = float(np.prod(d["shape"][-2:]))
    assert maxabs(d["chsum"] / npix, g["chsum"] / npix) <= atol


# ----------------------------------------------------------------------------- C4
def test_c4_unet_512(dev):
    sd, g = golden_state_dict("full_c4_unet")
    case = cases.FULLSIZE_CASES["c4_sr3_512"]
    x = cases.make_fullsize_input("c4_x", (1, 6, 512, 512))
    t = torch.tensor([[0.613]])
    ref = unet_forward(sd, case["cfg"], "sr3", x, t).numpy()
    eng = build_engine(case["cfg"], "sr3", sd)
    y = eng.forward(x.to(dev), t.to(dev), cond_channels=3).cpu().numpy()
    err = maxabs(y, ref)
    print(f"\nC4 sr_sr3_64_512 B=1 512^2 fp32: max|hip-oracle| = {err:.3e}, launches = {eng.num_launches(1, 512, 512, 3)}")
    assert err <= FP32_TOL
    _check_digest(y, g)
    for dt in ("bf16", "f16"):
        e16 = build_engine(case["cfg"], "sr3", sd, dtype=dt)
        y16 = e16.forward(x.to(dev), t.to(dev), cond_channels=3).cpu().numpy()
        p = psnr(ref, y16)
        print(f"C4 {dt}: PSNR vs fp32 oracle = {p:.1f} dB, max-abs {maxabs(y16, ref):.3e}")
        assert p > PSNR_MIN
        del e16


# ----------------------------------------------------------------------------- C3
def test_c3_indi_512_through_the_sampler_class(dev):
    """define_G('indi') -> InDISampler.inference(n = 3) with host-injected draws in the reference's order."""
    from diffsplitting_amd.model.ddpm_modules.unet import UNet
    from diffsplitting_amd.model.samplers import InDISampler
    sd, g = golden_state_dict("full_c3_indi")
    case = cases.FULLSIZE_CASES["c3_hagen_512"]
    x_in = cases.make_fullsize_input("c3_x", (1, 1, 512, 512))
    osd = {"denoise_fn." + k: v for k, v in sd.items()}
    rec = DrawRecorder(cases.LOOP_SEED)
    ref = samplers.indi_inference(osd, case["cfg"], x_in, 3, 2, randn=rec, continuous=True, t_float_start=1.0).numpy()
    c = case["cfg"]
    net = UNet(in_channel=c["in_channel"], out_channel=c["out_channel"], inner_channel=c["inner_channel"],
               norm_groups=c["norm_groups"], channel_mults=c["channel_mults"], attn_res=c["attn_res"],
               res_blocks=c["res_blocks"], image_size=c["image_size"])
    smp = InDISampler(net, 32, channels=2, out_channel=2, conditional=False, val_schedule_opt={"n_timestep": 3}).cuda()
    smp.load_state_dict(osd, strict=True)
    smp.set_new_noise_schedule({"n_timestep": 3}, "cuda")
    torch.manual_seed(cases.LOOP_SEED)                       # the reference's own draws, in its order
    smp.noise_source = lambda shape: torch.randn(shape)
    ret = smp.inference(x_in.to(dev), continuous=True, t_float_start=1.0).cpu().numpy()
    err = maxabs(ret, ref)
    print(f"\nC3 hagen 512^2 InDI n=3 (L=4096 attention) fp32: max|hip-oracle| = {err:.3e}")
    assert ret.shape == (4, 2, 512, 512) and err <= FP32_TOL
    _check_digest(ret, g)
    for dt in ("bf16", "f16"):
        net.compute_dtype = dt
        torch.manual_seed(cases.LOOP_SEED)
        r16 = smp.inference(x_in.to(dev), continuous=True, t_float_start=1.0).cpu().numpy()
        p = psnr(ref[-1], r16[-1])
        print(f"C3 {dt}: final-frame PSNR vs fp32 oracle = {p:.1f} dB")
        assert p > PSNR_MIN


# ----------------------------------------------------------------------------- C5
def test_c5_joint_indi_and_time_predictor_512(dev):
    from diffsplitting_amd.model import networks
    from diffsplitting_amd.model.ddpm_modules.time_predictor import TimePredictor
    from tests.test_gpu_boundary import _opt, _tiny_indi_section
    sd, g = golden_state_dict("full_c5_joint")
    case = cases.FULLSIZE_CASES["c5_joint_512"]
    x_in = cases.make_fullsize_input("c5_x", (1, 1, 512, 512))
    rec = DrawRecorder(cases.LOOP_SEED)
    ref = samplers.joint_indi_inference(sd, case["cfg"], x_in, 3, 1, randn=rec, continuous=True, t_float_start=0.5).numpy()
    netG = networks.define_G(_opt(_tiny_indi_section(1, 1, "joint_indi"))).cuda()
    netG.load_state_dict(sd, strict=True)
    netG.set_new_noise_schedule({"n_timestep": 3}, "cuda")
    torch.manual_seed(cases.LOOP_SEED)
    netG.noise_source = lambda shape: torch.randn(shape)
    ret = netG.inference(x_in.to(dev), continuous=True, t_float_start=0.5).cpu().numpy()
    err = maxabs(ret, ref)
    print(f"\nC5 joint 512^2 n=3 fp32: max|hip-oracle| = {err:.3e}")
    assert ret.shape == (4, 2, 512, 512) and err <= FP32_TOL
    _check_digest(ret, g)
    # the config's dtype: fp16 operands (PSNR of the engine output against the fp32 oracle)
    for m in netG.modules():
        if hasattr(m, "compute_dtype"):
            m.compute_dtype = "f16"
    torch.manual_seed(cases.LOOP_SEED)
    r16 = netG.inference(x_in.to(dev), continuous=True, t_float_start=0.5).cpu().numpy()
    p = psnr(ref[-1], r16[-1])
    print(f"C5 f16: final-frame PSNR vs fp32 oracle = {p:.1f} dB")
    assert p > PSNR_MIN
    # device-noise mode: the two loops on two HIP streams, hipGraph-captured steps
    netG.noise_source = None
    out = netG.inference(x_in.to(dev), continuous=False)
    torch.cuda.synchronize()
    assert out.shape == (1, 2, 512, 512) and torch.isfinite(out).all()

    sdt, gt = golden_state_dict("full_c5_timepred")
    x = cases.make_fullsize_input("c5_tp_x", (2, 1, 512, 512))
    tp = TimePredictor(**case["cfg"]).cuda()
    tp.load_state_dict(sdt, strict=True)
    t = tp(x.to(dev)).cpu().numpy()
    tref = time_predictor_forward(sdt, case["cfg"], x).numpy()
    print(f"C5 TimePredictor 512^2: t = {t}, max|hip-oracle| = {maxabs(t, tref):.3e}")
    assert maxabs(t, tref) <= FP32_TOL and maxabs(t, gt["t"]) <= FP32_TOL
    tp.unet.compute_dtype = "f16"
    t16 = tp(x.to(dev)).cpu().numpy()
    assert maxabs(t16, tref) <= 2e-2, maxabs(t16, tref)


# ----------------------------------------------------------------------------- loop-level reduced precision
def _sr3_loop(eng, sch, cond, draws, dev):
    from diffsplitting_amd import engine
    bufs, gam = engine.gaussian_buffers(sch)
    tab = engine.gaussian_step_table(bufs, gam, "sr3", True)
    x0 = draws[0].to(dev)
    noise = torch.zeros((tab.n_steps,) + tuple(x0.shape))
    for s, d in enumerate(draws[1:]):
        noise[s] = d
    x, _ = eng.sample_loop(tab, x0.clone(), cond=cond.to(dev), noise=noise.to(dev))
    torch.cuda.synchronize()
    return x.cpu()


@pytest.mark.parametrize("sched,shape", [("lin_25", (2, 3, 32, 32)), ("sr3_2000", (1, 3, 16, 16))])
def test_reduced_precision_loop_psnr(sched, shape, dev):
    """The headline dtype over WHOLE loops: bf16 / fp16 operands and activation storage for every step, identical
    injected noise, PSNR of the final images against the fp32 oracle loop (incl. the full 2000-step schedule)."""
    from tests.gpu_util import oracle_sr3_loop_tiny
    sd, case, sch, cond, draws, full = oracle_sr3_loop_tiny(sched, shape)
    for dt in ("bf16", "f16"):
        eng = build_engine(case["cfg"], "sr3", sd, dtype=dt)
        x = _sr3_loop(eng, sch, cond, draws, dev)
        p = psnr(full.numpy(), x.numpy())
        print(f"\n{sched} ({sch['n_timestep']} steps) {dt}: PSNR of the final images vs the fp32 oracle loop = {p:.1f} dB, "
              f"max-abs {maxabs(x, full):.3e}")
        assert torch.isfinite(x).all() and p > PSNR_MIN


def test_headline_unet_loop_psnr_bf16(dev):
    """The real sr_sr3_16_128 UNet, 12 reverse steps from the tail of the T = 2000 schedule (where the image forms),
    bf16 vs the fp32 engine path (itself <= 1e-3 from the oracle) on identical injected noise."""
    from diffsplitting_amd import engine
    sd, _ = golden_state_dict("unet_sr3_128")
    case = cases.UNET_CASES["sr3_128"]
    bufs, gam = engine.gaussian_buffers(cases.SCHEDULES["sr3_2000"])
    full = engine.gaussian_step_table(bufs, gam, "sr3", True)
    idx = np.arange(2000 - 12, 2000)
    tab = engine.StepTableHost(full.tcond[idx], c1=full.c1[idx], c2=full.c2[idx], sigma=full.sigma[idx], a=full.a[idx],
                               b=full.b[idx], predict_eps=True, clip=True)
    g = torch.Generator().manual_seed(5)
    cond = torch.randn((2, 3, 128, 128), generator=g).to(dev)
    x0 = (0.3 * torch.randn((2, 3, 128, 128), generator=g)).to(dev)
    noise = torch.randn((12, 2, 3, 128, 128), generator=g).to(dev)
    outs = {}
    for dt in ("f32", "bf16", "f16"):
        eng = build_engine(case["cfg"], "sr3", sd, dtype=dt)
        outs[dt] = eng.sample_loop(tab, x0.clone(), cond=cond, noise=noise)[0].cpu().numpy()
        del eng
    for dt in ("bf16", "f16"):
        p = psnr(outs["f32"], outs[dt])
        print(f"\nsr_sr3_16_128 UNet, last 12 steps of T=2000, {dt}: PSNR vs fp32 = {p:.1f} dB")
        assert p > PSNR_MIN


# ----------------------------------------------------------------------------- regression: round-1 bugs
def test_headline_plan_bitwise_repeatable(dev):
    """B = 16 bf16 (the benchmark's plan: persistent warp-specialised kernels, fused statistics, split-K): two
    forwards and two 40-step graph loops must agree bitwise (a scale/shift slot race shipped once in round 1)."""
    from diffsplitting_amd import engine
    sd, _ = golden_state_dict("unet_sr3_128")
    case = cases.UNET_CASES["sr3_128"]
    eng = build_engine(case["cfg"], "sr3", sd, dtype="bf16")
    g = torch.Generator().manual_seed(3)
    x = torch.randn(16, 6, 128, 128, generator=g).to(dev)
    t = (0.05 + 0.9 * torch.rand(16, 1, generator=g)).to(dev)
    y1 = eng.forward(x, t, cond_channels=3).clone()
    y2 = eng.forward(x, t, cond_channels=3).clone()
    assert torch.isfinite(y1).all() and torch.equal(y1, y2)
    bufs, gam = engine.gaussian_buffers(cases.SCHEDULES["sr3_2000"])
    full = engine.gaussian_step_table(bufs, gam, "sr3", True)
    idx = np.arange(40)
    tab = engine.StepTableHost(full.tcond[idx], c1=full.c1[idx], c2=full.c2[idx], sigma=full.sigma[idx], a=full.a[idx],
                               b=full.b[idx], predict_eps=True, clip=True)
    cond = x[:, :3].contiguous()
    x0 = engine.randn((16, 3, 128, 128), seed=5)
    o1 = eng.sample_loop(tab, x0.clone(), cond=cond, seed=7)[0].clone()
    o2 = eng.sample_loop(tab, x0.clone(), cond=cond, seed=7)[0].clone()
    o3 = eng.sample_loop(tab, x0.clone(), cond=cond, seed=8)[0].clone()   # same captured graph, other seed
    assert torch.equal(o1, o2) and not torch.equal(o1, o3)


def test_unconditional_sample_continuous_keeps_initial_noise(dev):
    """sample(continous=True) of an unconditional model: slot 0 of the returned stack is the INITIAL noise
    (sr3 diffusion.py:183-185), not the final sample (the loop updates its state in place)."""
    from diffsplitting_amd.model.samplers import GaussianSampler
    from diffsplitting_amd.model.sr3_modules.unet import UNet
    c = dict(cases.UNET_CASES["sr3_tiny"]["cfg"])
    c["in_channel"] = 3
    net = UNet(in_channel=3, out_channel=3, inner_channel=c["inner_channel"], norm_groups=c["norm_groups"],
               channel_mults=c["channel_mults"], attn_res=c["attn_res"], res_blocks=c["res_blocks"], image_size=32)
    smp = GaussianSampler(net, 32, channels=3, conditional=False).cuda()
    smp.set_new_noise_schedule(cases.SCHEDULES["lin_8"], "cuda")
    rec = DrawRecorder(4)
    smp.noise_source = rec
    ret = smp.sample(batch_size=2, continous=True)
    torch.cuda.synchronize()
    n_snap = len([i for i in range(8) if i % (1 | (8 // 10)) == 0])
    assert ret.shape == (2 * (1 + n_snap), 3, 32, 32)
    assert torch.equal(ret[:2].cpu(), rec.draws[0])                   # the initial noise, untouched
    assert torch.equal(ret[-2:], smp.last_full_batch)                 # last snapshot = final state (i == 0)
    assert not torch.equal(ret[:2], ret[-2:])
