"""The oracle (oracle/) against fixtures generated from the reference's own
classes (oracle/gen_golden.py) and against the reference's known-answer test
for the tiling path.  CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import cases, samplers, tiling
from oracle.unet import time_predictor_forward, unet_forward
from tests.util import golden_state_dict, load_golden

torch.set_grad_enabled(False)


@pytest.mark.parametrize("name", list(cases.UNET_CASES))
def test_unet_forward(name):
    sd, g = golden_state_dict("unet_" + name)
    case = cases.UNET_CASES[name]
    x, t = cases.make_unet_inputs(name)
    y = unet_forward(sd, case["cfg"], case["flavour"], x, t)
    # same ATen ops in the same order as the reference modules -> bitwise on one machine,
    # 1e-5 allows for a different CPU's vectorised kernels
    np.testing.assert_allclose(y.numpy(), g["y"], rtol=0, atol=2e-5)


@pytest.mark.parametrize("name", list(cases.SCHEDULES))
def test_gaussian_schedule_bit_exact(name):
    g = load_golden("schedule_" + name)
    sch = samplers.gaussian_schedule(cases.SCHEDULES[name])
    for k, v in g.items():
        if k == "sqrt_alphas_cumprod_prev_f64":
            assert np.array_equal(sch["sqrt_alphas_cumprod_prev"], v)
        else:
            assert np.array_equal(sch[k].numpy(), v), k


@pytest.mark.parametrize("sched", ["lin_8", "lin_25"])
def test_sr3_loop(sched):
    sd, g = golden_state_dict("loop_sr3_" + sched)
    sd = {"denoise_fn." + k: v for k, v in sd.items()}
    case = cases.UNET_CASES["sr3_tiny"]
    sch = samplers.gaussian_schedule(cases.SCHEDULES[sched])
    cond = cases.make_cond("sr3_loop")
    torch.manual_seed(cases.LOOP_SEED)
    ret = samplers.sr3_p_sample_loop(sd, case["cfg"], sch, cond, continous=True)
    np.testing.assert_allclose(ret.numpy(), g["ret"], rtol=0, atol=5e-5)
    torch.manual_seed(cases.LOOP_SEED)
    last = samplers.sr3_p_sample_loop(sd, case["cfg"], sch, cond, continous=False)
    assert last.shape == g["last"].shape  # (C,H,W): last element of last snapshot (Q1)
    np.testing.assert_allclose(last.numpy(), g["last"], rtol=0, atol=5e-5)


def test_ddpm_loop():
    sd, g = golden_state_dict("loop_ddpm_lin_8")
    sd = {"denoise_fn." + k: v for k, v in sd.items()}
    sch = samplers.gaussian_schedule(cases.SCHEDULES["lin_8"])
    cond = cases.make_cond("ddpm_loop")
    torch.manual_seed(cases.LOOP_SEED)
    ret = samplers.ddpm_p_sample_loop(sd, cases.DDPM_COND_CASE["cfg"], sch, cond, continous=True, channels=1)
    np.testing.assert_allclose(ret.numpy(), g["ret"], rtol=0, atol=5e-5)


@pytest.mark.parametrize("n,t0", [(1, 1.0), (3, 1.0), (10, 1.0), (20, 1.0), (4, 0.6)])
def test_indi_loop(n, t0):
    sd, g = golden_state_dict(f"loop_indi_n{n}_t{t0}")
    sd = {"denoise_fn." + k: v for k, v in sd.items()}
    case = cases.UNET_CASES["ddpm_tiny"]
    x_in = cases.make_cond("indi_loop")
    torch.manual_seed(cases.LOOP_SEED)
    ret = samplers.indi_inference(sd, case["cfg"], x_in, n, 2, continuous=True, t_float_start=t0)
    np.testing.assert_allclose(ret.numpy(), g["ret"], rtol=0, atol=5e-5)
    torch.manual_seed(cases.LOOP_SEED)
    last = samplers.indi_inference(sd, case["cfg"], x_in, n, 2, continuous=False, t_float_start=t0)
    assert last.shape == g["last"].shape == (1, 2, 32, 48)  # ret[-1:] keeps one batch element (Q1)
    np.testing.assert_allclose(last.numpy(), g["last"], rtol=0, atol=5e-5)


@pytest.mark.parametrize("n", cases.C1_STEPS)
def test_c1_cifar_indi_loop(n):
    """BASELINE C1 (config/splitting_cifar10_indi.json:43-44,65,71): UNet 6 -> 6, 1-channel x_in replicated x6
    (indi.py:80), batch 4 at 32^2, n = 20 and n = 100 -- both trip the reference's drift assert (generated under -O)."""
    sd, g = golden_state_dict(f"loop_c1_cifar_n{n}")
    sd = {"denoise_fn." + k: v for k, v in sd.items()}
    x_in = cases.make_cond("c1_cifar")
    torch.manual_seed(cases.LOOP_SEED)
    ret = samplers.indi_inference(sd, cases.C1_CASE["cfg"], x_in, n, 6, continuous=True, t_float_start=1.0)
    blocks = ret.numpy().reshape(-1, 4, 6, 32, 32)
    assert blocks.shape[0] == int(g["nblocks"])
    np.testing.assert_allclose(blocks[cases.c1_keep(blocks.shape[0])], g["blocks"], rtol=0, atol=5e-5)
    torch.manual_seed(cases.LOOP_SEED)
    last = samplers.indi_inference(sd, cases.C1_CASE["cfg"], x_in, n, 6, continuous=False, t_float_start=1.0)
    assert last.shape == g["last"].shape == (1, 6, 32, 32)
    np.testing.assert_allclose(last.numpy(), g["last"], rtol=0, atol=5e-5)


def test_indi_frame_count_invariant():
    """tests/test_joint_indi.py:9-25 — n_timestep+1 frames for n in {1,2,10} (with the
    working entry point)."""
    ident = {}
    for n in (1, 2, 10):
        x = torch.randn(1, 1, 16, 16)
        # identity-ish denoiser through the sampler arithmetic only
        ts, c0, c1, cn = samplers.indi_schedule(n, 0.5)
        assert len(ts) == n
        ident[n] = ts
    g = load_golden("loop_joint_n3")
    assert g["ret"].shape[0] == (3 + 1) * 2  # batch 2, n=3 -> 4 frames of 2


@pytest.mark.parametrize("n,t0", cases.INDI_T_CASES)
def test_indi_t_sequence_bit_exact(n, t0):
    g = load_golden("indi_tseq")
    ts, c0, c1, cn = samplers.indi_schedule(n, t0)
    assert np.array_equal(ts, g[f"t_n{n}_t{t0}"])
    # coefficient pin: replay the reference's stub-denoiser run with the schedule scalars
    torch.manual_seed(5)
    x_in = torch.randn(1, 1, 4, 4)
    x = x_in + torch.randn(x_in.shape) * (0.01 * torch.Tensor([t0]))
    for i in range(n):
        x0 = 0.5 * x
        noise = torch.randn(x.shape) * (0.01 * torch.tensor([cn[i]]))
        x = torch.tensor([c0[i]]) * x0 + torch.tensor([c1[i]]) * x + noise
    assert np.array_equal(x.numpy(), g[f"x_n{n}_t{t0}"])


def test_joint_indi_loop():
    sd, g = golden_state_dict("loop_joint_n3")
    case = cases.UNET_CASES["joint_32"]
    x_in = cases.make_cond("joint_loop")
    torch.manual_seed(cases.LOOP_SEED)
    ret = samplers.joint_indi_inference(sd, case["cfg"], x_in, 3, 1, continuous=True, t_float_start=0.5)
    np.testing.assert_allclose(ret.numpy(), g["ret"], rtol=0, atol=5e-5)
    torch.manual_seed(cases.LOOP_SEED)
    last = samplers.joint_indi_inference(sd, case["cfg"], x_in, 3, 1, continuous=False, t_float_start=0.3)
    np.testing.assert_allclose(last.numpy(), g["last_t03"], rtol=0, atol=5e-5)


def test_time_predictor():
    sd, g = golden_state_dict("time_predictor")
    x = cases.make_cond("time_pred")
    t = time_predictor_forward(sd, cases.TIME_PRED_CFG, x)
    np.testing.assert_allclose(t.numpy(), g["t"], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("name,data_shape,grid_shape,patch_shape", cases.TILE_CASES)
def test_tile_plan_matches_reference(name, data_shape, grid_shape, patch_shape):
    g = load_golden("tiles_" + name)
    plan = tiling.TilePlan(data_shape, grid_shape, patch_shape)
    assert plan.total() == int(g["total"])
    locs = np.array([plan.location(i) for i in range(plan.total())])
    plocs = np.array([plan.patch_location(i) for i in range(plan.total())])
    assert np.array_equal(locs, g["locs"]) and np.array_equal(plocs, g["plocs"])
    if "stitched" in g:
        rng = np.random.default_rng(3)
        pred = rng.standard_normal((plan.total(), 2, patch_shape[1], patch_shape[2])).astype(np.float32)
        assert np.array_equal(tiling.stitch(pred, plan), g["stitched"])


def test_stitch_known_answer():
    """The reference's own known-answer test (tests/test_tiling_setup.py:35-55):
    tiles cut from arange frames, stitched back, equal the frames exactly."""
    n, H, W, C = 5, 512, 512, 2
    data = np.arange(n * H * W * C).reshape(n, H, W, C)
    plan = tiling.TilePlan((n, H, W), (1, 128, 128), (1, 256, 256))
    assert plan.total() == 45
    preds = np.stack([tiling.extract_patch(data, plan, i) for i in range(plan.total())])
    assert (tiling.stitch(preds, plan) == data).all()


# ----------------------------------------------------------------------------- full-size cases (C3 / C4 / C5 shapes)
def _check_digest(y, g, atol):
    d = cases.digest(y)
    assert list(d["shape"]) == list(g["shape"])
    for k in ("crop", "grid", "corner"):
        np.testing.assert_allclose(d[k], g[k], rtol=0, atol=atol, err_msg=k)
    npix = float(np.prod(d["shape"][-2:]))
    np.testing.assert_allclose(d["chsum"] / npix, g["chsum"] / npix, rtol=0, atol=atol)
    np.testing.assert_allclose(d["chsq"], g["chsq"], rtol=1e-4)


def test_fullsize_c4_unet():
    """The sr_sr3_64_512 UNet (config 4) at 512^2, B = 1: oracle vs the reference's digest."""
    sd, g = golden_state_dict("full_c4_unet")
    case = cases.FULLSIZE_CASES["c4_sr3_512"]
    x = cases.make_fullsize_input("c4_x", (1, 6, 512, 512))
    y = unet_forward(sd, case["cfg"], "sr3", x, torch.tensor([[0.613]]))
    _check_digest(y.numpy(), g, 2e-4)


def test_fullsize_c3_indi():
    """InDI n = 3 on one 512^2 Hagen tile (config 3; bottleneck attention over 4096 tokens)."""
    sd, g = golden_state_dict("full_c3_indi")
    case = cases.FULLSIZE_CASES["c3_hagen_512"]
    x_in = cases.make_fullsize_input("c3_x", (1, 1, 512, 512))
    osd = {"denoise_fn." + k: v for k, v in sd.items()}
    torch.manual_seed(cases.LOOP_SEED)
    ret = samplers.indi_inference(osd, case["cfg"], x_in, 3, 2, continuous=True, t_float_start=1.0)
    _check_digest(ret.numpy(), g, 2e-4)


def test_fullsize_c5_joint_and_time_predictor():
    sd, g = golden_state_dict("full_c5_joint")
    case = cases.FULLSIZE_CASES["c5_joint_512"]
    x_in = cases.make_fullsize_input("c5_x", (1, 1, 512, 512))
    torch.manual_seed(cases.LOOP_SEED)
    ret = samplers.joint_indi_inference(sd, case["cfg"], x_in, 3, 1, continuous=True, t_float_start=0.5)
    _check_digest(ret.numpy(), g, 2e-4)
    sd, g = golden_state_dict("full_c5_timepred")
    x = cases.make_fullsize_input("c5_tp_x", (2, 1, 512, 512))
    t = time_predictor_forward(sd, case["cfg"], x)
    np.testing.assert_allclose(t.numpy(), g["t"], rtol=1e-5, atol=1e-6)


def test_cached_oracle_run_is_the_oracle():
    """tests/golden/oracle_run_sr3_2000_tiny.npz (oracle/gen_oracle_cache.py) is the oracle's own 2000-step SR3 loop on
    the tiny UNet, cached so that the GPU tests need not spend minutes of the GPU box's host on it: recompute it here
    and compare (1e-5: hosts sum convolutions in different orders; the GPU tests that use it allow 1e-3)."""
    import numpy as np
    from tests import gpu_util
    from oracle import gen_oracle_cache as gc
    gpu_util.oracle_sr3_loop_tiny.cache_clear()
    sd, case, sch, cond, draws, full = gpu_util.oracle_sr3_loop_tiny(gc.SCHED, gc.SHAPE, gc.COND_SEED, gc.DRAW_SEED)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_run_sr3_2000_tiny.npz")
    assert os.path.exists(path), "run python oracle/gen_oracle_cache.py"
    _, _, _, cond2, draws2, full2 = gpu_util.compute_oracle_sr3_loop_tiny(gc.SCHED, gc.SHAPE, gc.COND_SEED, gc.DRAW_SEED)
    assert len(draws) == len(draws2) and all(bool((a == b).all()) for a, b in zip(draws[:5] + draws[-5:], draws2[:5] + draws2[-5:]))
    assert bool((cond == cond2).all())
    err = float(np.max(np.abs(full.numpy().astype(np.float64) - full2.numpy().astype(np.float64))))
    assert err <= 1e-5, err
