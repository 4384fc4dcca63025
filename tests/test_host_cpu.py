"""CPU-side checks of the product's host logic: the C-ABI library loads and
exports everything include/dsx.h declares, its parameter table equals the
reference's state_dict key lists, the tile planner equals the reference's
TileIndexManager, and the host schedule tables are bit-exact.  No GPU compute."""
import ctypes as C
import json
import os
import re

import numpy as np
import pytest
import torch

from oracle import cases, samplers
from tests.util import GOLDEN, load_golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from diffsplitting_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "dsx.h")).read()
    declared = sorted(set(re.findall(r"\b(dsx_[a-z0-9_]+)\s*\(", hdr)))
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(_lib.lib, name), name
    assert sorted(_lib.SIGNATURES) == declared
    assert _lib.lib.dsx_abi_version() == 2


def test_compute_fails_loudly_without_gpu():
    from diffsplitting_amd import _lib, engine
    if _lib.lib.dsx_device_count() > 0:
        pytest.skip("GPU present")
    c = cases.UNET_CASES["ddpm_tiny"]["cfg"]
    eng = engine.UNetEngine(engine.make_cfg("ddpm", c["in_channel"], c["out_channel"], c["inner_channel"],
                                            c["norm_groups"], c["channel_mults"], c["attn_res"],
                                            c["res_blocks"], c["image_size"]), "ddpm")
    with pytest.raises(_lib.DsxError):
        eng.finalize("f32")
    with pytest.raises(_lib.DsxError):
        engine.randn((4,), 0)


def _engine_for(cfg, flavour, with_time_emb=True):
    from diffsplitting_amd import engine
    return engine.UNetEngine(engine.make_cfg(flavour, cfg["in_channel"], cfg["out_channel"], cfg["inner_channel"],
                                             cfg["norm_groups"], cfg["channel_mults"], cfg["attn_res"],
                                             cfg["res_blocks"], cfg["image_size"], with_time_emb), flavour)


@pytest.mark.parametrize("name", list(cases.UNET_CASES))
def test_param_table_equals_reference_state_dict(name):
    g = load_golden("unet_" + name)
    case = cases.UNET_CASES[name]
    eng = _engine_for(case["cfg"], case["flavour"])
    assert list(zip(eng.param_names, eng.param_shapes)) == [(k, tuple(s)) for k, s in g["keys"]]


def test_param_table_baseline_configs():
    """Key lists of the reference's define_G / UNets for the BASELINE configs."""
    blob = json.load(open(os.path.join(GOLDEN, "state_dict_keys.json")))
    keys = blob["keys"]
    for cfgname, prefixes in (("sr_sr3_16_128", ["denoise_fn."]), ("sr_sr3_64_512", ["denoise_fn."]),
                              ("sr_ddpm_16_128", ["denoise_fn."]), ("splitting_hagen_indi", ["denoise_fn."]),
                              ("splitting_cifar10_indi", ["denoise_fn."]),
                              ("splitting_hagen_indi_joint", ["indi1.denoise_fn.", "indi2.denoise_fn."])):
        opt = {"model": blob["model"][cfgname]}
        u = opt["model"]["unet"]
        cfg = dict(in_channel=u["in_channel"], out_channel=u["out_channel"], inner_channel=u["inner_channel"],
                   norm_groups=u.get("norm_groups") or 32, channel_mults=u["channel_multiplier"],
                   attn_res=u["attn_res"], res_blocks=u["res_blocks"],
                   image_size=opt["model"]["diffusion"]["image_size"])
        flavour = "sr3" if opt["model"]["which_model_G"] == "sr3" else "ddpm"
        eng = _engine_for(cfg, flavour)
        ref = [(k, tuple(s)) for k, s in keys[cfgname]]
        for p in prefixes:
            mine = [(p + n, s) for n, s in zip(eng.param_names, eng.param_shapes)]
            sub = [(k, s) for k, s in ref if k.startswith(p)]
            assert mine == sub, cfgname


def test_time_predictor_param_table():
    g = load_golden("time_predictor")
    eng = _engine_for(cases.TIME_PRED_CFG, "ddpm", with_time_emb=False)
    mine = [("unet." + n, s) for n, s in zip(eng.param_names, eng.param_shapes)]
    ref = [(k, tuple(s)) for k, s in g["keys"] if k.startswith("unet.")]
    assert mine == ref


def test_model_flops_match_survey():
    """Algorithmic FLOPs per image-step: 92.353 GF (C2), 1246.11 GF (C4), 70.867 GF (C3)."""
    c2 = _engine_for(cases.UNET_CASES["sr3_128"]["cfg"], "sr3")
    assert abs(c2.flops(128, 128) / 1e9 - 92.353) < 0.05
    c3 = _engine_for(cases.UNET_CASES["hagen_64"]["cfg"], "ddpm")
    assert abs(c3.flops(512, 512) / 1e9 - 70.867) < 0.05
    c4 = _engine_for(dict(in_channel=6, out_channel=3, inner_channel=64, norm_groups=16,
                          channel_mults=(1, 2, 4, 8, 16), attn_res=(), res_blocks=1, image_size=512), "sr3")
    assert abs(c4.flops(512, 512) / 1e9 - 1246.11) < 0.5


@pytest.mark.parametrize("name,data_shape,grid_shape,patch_shape", cases.TILE_CASES)
def test_tile_plan_equals_reference(name, data_shape, grid_shape, patch_shape):
    from diffsplitting_amd.data.tiling import TilePlan
    g = load_golden("tiles_" + name)
    plan = TilePlan(data_shape, grid_shape, patch_shape)
    assert plan.total == int(g["total"])
    assert np.array_equal(plan.grid_start, g["locs"]) and np.array_equal(plan.patch_start, g["plocs"])
    # regions against the oracle's restatement of tile_stitcher.py:26-56
    from oracle.tiling import TilePlan as OPlan
    op = OPlan(data_shape, grid_shape, patch_shape)
    from oracle.tiling import paste_region
    covered = 0
    for i in range(plan.total):
        if i % max(1, plan.total // 50) == 0:
            vgs, vge, rs, re = op.valid_region(i)
            r = plan.valid_regions[i]
            assert (r[0], r[1], r[2]) == tuple(vgs) and (r[3], r[4]) == (vge[1] - vgs[1], vge[2] - vgs[2])
            assert (r[5], r[6]) == (rs[1], rs[2])
            # the paste regions: clipped where the sequential paste of tile_stitcher.py:68-80 lets a later tile overwrite
            vgs, vge, rs, re = paste_region(op, i)
            r = plan.regions[i]
            assert (r[0], r[1], r[2]) == tuple(vgs) and (r[3], r[4]) == (vge[1] - vgs[1], vge[2] - vgs[2])
            assert (r[5], r[6]) == (rs[1], rs[2])
        covered += int(plan.regions[i][3]) * int(plan.regions[i][4])
    assert covered == int(np.prod(data_shape))                        # every canvas pixel is pasted exactly once


def test_tile_plan_rejects_bad_shapes():
    from diffsplitting_amd.data.tiling import TilePlan
    from diffsplitting_amd._lib import DsxError
    with pytest.raises(DsxError):
        TilePlan((2, 64, 64), (1, 32, 32), (1, 33, 33))   # odd padding (tiling_manager.py:27-29)
    with pytest.raises(DsxError):
        TilePlan((2, 64, 64), (1, 32, 32), (1, 16, 16))   # patch < grid


@pytest.mark.parametrize("name", list(cases.SCHEDULES))
def test_host_gaussian_buffers_bit_exact(name):
    from diffsplitting_amd import engine
    g = load_golden("schedule_" + name)
    bufs, gam = engine.gaussian_buffers(cases.SCHEDULES[name])
    assert np.array_equal(gam, g["sqrt_alphas_cumprod_prev_f64"])
    for k, v in bufs.items():
        assert np.array_equal(v.numpy(), g[k]), k
    tab = engine.gaussian_step_table(bufs, gam, "sr3")
    T = tab.n_steps
    # step s handles i = T-1-s; gamma index is i+1 (diffusion.py:153-154)
    for s in (0, 1, T // 2, T - 1):
        i = T - 1 - s
        assert tab.tcond[s] == torch.FloatTensor([g["sqrt_alphas_cumprod_prev_f64"][i + 1]]).item()
        assert tab.a[s] == g["sqrt_recip_alphas_cumprod"][i] and tab.c2[s] == g["posterior_mean_coef2"][i]
    assert tab.sigma[T - 1] == 0.0 and tab.sigma[0] > 0


@pytest.mark.parametrize("n,t0", cases.INDI_T_CASES)
def test_host_indi_table_bit_exact(n, t0):
    from diffsplitting_amd import engine
    g = load_golden("indi_tseq")
    tab = engine.indi_step_table(n, t0)
    assert np.array_equal(tab.tcond, g[f"t_n{n}_t{t0}"])          # the reference's own t sequence
    ts, c0, c1, cn = samplers.indi_schedule(n, t0)
    assert np.array_equal(tab.c1, c0) and np.array_equal(tab.c2, c1)
    assert np.array_equal(tab.sigma, (torch.tensor(cn) * 0.01).numpy())


def test_snapshot_schedules():
    from diffsplitting_amd import engine
    # sr3: i % (1|(T//10)) == 0 on the descending index -> 10 snapshots at T=2000
    s = engine.gaussian_snapshot_steps(2000)
    assert len(s) == 10 and s[-1] == 1999 and [1999 - k for k in s][::-1][:3] == [0, 201, 402]
    assert engine.indi_snapshot_steps(3) == [0, 1, 2] and engine.indi_snapshot_steps(1) == [0]
    assert engine.indi_snapshot_steps(100)[-1] == 99


def test_psnr_matches_reference_cpu():
    from diffsplitting_amd.core.psnr import PSNR, RangeInvariantPsnr
    g = load_golden("psnr")
    gt, pred = torch.from_numpy(g["gt"]), torch.from_numpy(g["pred"])
    assert np.allclose(PSNR(gt, pred).numpy(), g["psnr"], atol=1e-4)
    assert np.allclose(RangeInvariantPsnr(gt, pred).numpy(), g["ri_psnr"], atol=1e-4)


def test_define_G_state_dict_keys_all_baseline_configs():
    """define_G(opt) builds every config of the reference (incl. sr3/ddpm: rot R1) and its
    state_dict carries exactly the reference's keys and shapes."""
    from diffsplitting_amd.core.logger import dict_to_nonedict
    from diffsplitting_amd.model import networks
    blob = json.load(open(os.path.join(GOLDEN, "state_dict_keys.json")))
    for name, section in blob["model"].items():
        opt = dict_to_nonedict({"model": section, "phase": "val", "gpu_ids": None, "distributed": False})
        netG = networks.define_G(opt)
        mine = [(k, list(v.shape)) for k, v in netG.state_dict().items()]
        assert mine == [(k, list(s)) for k, s in blob["keys"][name]], name
        assert opt["model"]["unet"]["norm_groups"] is not None          # define_G fills it (Q9)


def test_model_refuses_cpu_inference():
    from diffsplitting_amd._lib import DsxError
    from diffsplitting_amd.model.ddpm_modules.unet import UNet
    net = UNet(in_channel=2, out_channel=2, inner_channel=16, norm_groups=16, channel_mults=(1, 2), attn_res=(),
               res_blocks=1, image_size=32)
    with pytest.raises(DsxError):
        net(torch.zeros(1, 2, 32, 32), torch.tensor([0.5]))


def test_config_loader_strips_comments(tmp_path):
    from diffsplitting_amd.core.logger import dict_to_nonedict, load_json
    p = tmp_path / "c.json"
    p.write_text('{\n "a": 1, // trailing comment\n "b": {"c": [1, 2]} // another\n}\n')
    opt = dict_to_nonedict(load_json(str(p)))
    assert opt["a"] == 1 and opt["b"]["c"] == [1, 2] and opt["missing"] is None and opt["b"]["zzz"] is None


def test_reference_module_paths_via_compat(tmp_path):
    """with diffsplitting_amd/compat on PYTHONPATH the reference's own import lines
    (`import model as Model`, `import model.networks`, `from data.tiling_manager import ...`, split.py:1-20) resolve to
    the engine-backed packages, from a working directory outside the repo, and define_G builds from a
    reference-schema opt."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r"""
import json, sys
import model as Model
import model.networks, model.model, data.tiling_manager, data.tile_stitcher, data.split_dataset_tiledpred
import core.logger as Logger, core.psnr
from model.ddpm_modules.indi import InDI
from model.ddpm_modules.joint_indi import JointIndi
from model.sr3_modules.unet import UNet
from model.sr3_modules.diffusion import GaussianDiffusion
from data.tiling_manager import TileIndexManager, TilingMode
import diffsplitting_amd.model.networks as real
assert model.networks is real and Model.create_model is not None
blob = json.load(open(sys.argv[1]))
for name in ("splitting_cifar10_indi", "sr_sr3_16_128", "splitting_hagen_indi_joint"):
    opt = Logger.dict_to_nonedict({"model": blob["model"][name], "phase": "val", "gpu_ids": None, "distributed": False})
    netG = model.networks.define_G(opt)
    assert [k for k in netG.state_dict()] == [k for k, _ in blob["keys"][name]], name
mng = TileIndexManager((5, 512, 512), (1, 128, 128), (1, 256, 256), TilingMode.ShiftBoundary)
assert mng.total_grid_count() == 45
print("compat ok")
"""
    env = dict(os.environ)
    env["PYTHONPATH"] = os.pathsep.join([os.path.join(root, "diffsplitting_amd", "compat"), root])
    r = subprocess.run([sys.executable, "-c", code, os.path.join(GOLDEN, "state_dict_keys.json")], cwd=str(tmp_path),
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "compat ok" in r.stdout, r.stderr[-2000:]


def test_affine_film_checkpoint_is_rejected_by_name():
    """use_affine_level=True (sr3 unet.py:34-50: Linear to 2 C rows, (1 + gamma) x + beta) is not implemented -- no
    reference config enables it; such a checkpoint must be refused with a message that says so, not a bare shape error."""
    from diffsplitting_amd._lib import DsxError
    from oracle.weights import synth_state_dict
    g = load_golden("unet_sr3_tiny")
    eng = _engine_for(cases.UNET_CASES["sr3_tiny"]["cfg"], "sr3")
    sd = synth_state_dict(g["keys"], 0)
    k = next(k for k in sd if "noise_func" in k and k.endswith("weight"))
    sd[k] = torch.cat([sd[k], sd[k]], dim=0)
    with pytest.raises(DsxError, match="use_affine_level"):
        eng.load_state_dict(sd)
