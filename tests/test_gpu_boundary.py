"""The drop-in boundary on the MI355X: define_G / create_model / DDPM.test() /
tiled prediction / split.py behave like the reference's (checked against the
CPU oracle and golden vectors).  `-m gpu`."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import cases, samplers, tiling
from oracle.weights import synth_state_dict
from tests.gpu_util import DrawRecorder, maxabs
from tests.util import GOLDEN, golden_state_dict, load_golden

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)
FP32_TOL = 1e-3


def _opt(model_section, **extra):
    from diffsplitting_amd.core.logger import dict_to_nonedict
    d = {"model": json.loads(json.dumps(model_section)), "phase": "val", "gpu_ids": [0], "distributed": False,
         "path": {"resume_state": None}}
    d.update(extra)
    return dict_to_nonedict(d)


def _tiny_indi_section(in_ch=2, out_ch=2, which="indi"):
    return {"which_model_G": which, "loss_type": "l1", "lr_reduction": "mean", "finetune_norm": False,
            "w_input_loss": 0.0,
            "unet": {"in_channel": in_ch, "out_channel": out_ch, "inner_channel": 16, "norm_groups": 16,
                     "channel_multiplier": [1, 2, 4] if which == "indi" else [1, 2, 4, 8], "attn_res": [],
                     "res_blocks": 1, "dropout": 0},
            "beta_schedule": {"train": {"schedule": "linear", "n_timestep": 20, "linear_start": 1e-6, "linear_end": 1e-2},
                              "val": {"schedule": "linear", "n_timestep": 3, "linear_start": 1e-6, "linear_end": 1e-2}},
            "diffusion": {"image_size": 32, "channels": out_ch, "conditional": False}}


def test_create_model_indi_test_matches_oracle():
    """create_model -> load_state_dict(reference keys) -> feed_data -> test(): same output as the oracle's
    InDI.inference, including the 'last batch element only' return (Q1) and the full-batch accessor."""
    from diffsplitting_amd.model import create_model
    sd, g = golden_state_dict("loop_indi_n3_t1.0")
    model = create_model(_opt(_tiny_indi_section()))
    model.netG.load_state_dict({"denoise_fn." + k: v for k, v in sd.items()}, strict=True)
    model.set_new_noise_schedule({"n_timestep": 3}, schedule_phase="val")
    x_in = cases.make_cond("indi_loop")
    torch.manual_seed(cases.LOOP_SEED)                 # same CPU draws as the golden run
    model.netG.noise_source = lambda shape: torch.randn(shape)
    model.feed_data({"input": x_in.clone(), "target": torch.zeros(3, 2, 32, 48)})
    model.test(continuous=False)
    vis = model.get_current_visuals()
    assert vis["prediction"].shape == (1, 2, 32, 48)
    assert maxabs(vis["prediction"].numpy(), g["last"]) <= FP32_TOL
    torch.manual_seed(cases.LOOP_SEED)
    model.test(continuous=True)
    assert maxabs(model.get_current_visuals()["prediction"].numpy(), g["ret"]) <= FP32_TOL
    assert model.netG.last_full_batch.shape == (3, 2, 32, 48)


def test_define_G_sr3_superset_route():
    """R1/R2: define_G builds 'sr3' and DDPM.test() routes to super_resolution; output equals the oracle loop."""
    from diffsplitting_amd.model import create_model
    sd, g = golden_state_dict("loop_sr3_lin_8")
    case = cases.UNET_CASES["sr3_tiny"]["cfg"]
    sec = {"which_model_G": "sr3", "finetune_norm": False,
           "unet": {"in_channel": 6, "out_channel": 3, "inner_channel": 32, "norm_groups": 32,
                    "channel_multiplier": [1, 2, 4], "attn_res": [16], "res_blocks": 2, "dropout": 0.2},
           "beta_schedule": {"train": cases.SCHEDULES["lin_8"], "val": cases.SCHEDULES["lin_8"]},
           "diffusion": {"image_size": 32, "channels": 3, "conditional": True}}
    model = create_model(_opt(sec))
    missing, unexpected = model.netG.load_state_dict({"denoise_fn." + k: v for k, v in sd.items()}, strict=False)
    assert not unexpected and all(not k.startswith("denoise_fn.") for k in missing)
    cond = cases.make_cond("sr3_loop")
    torch.manual_seed(cases.LOOP_SEED)
    model.netG.noise_source = lambda shape: torch.randn(shape)
    model.feed_data({"input": cond.clone(), "target": cond.clone()})
    model.test(continuous=True)
    assert maxabs(model.get_current_visuals()["prediction"].numpy(), g["ret"]) <= FP32_TOL
    torch.manual_seed(cases.LOOP_SEED)
    model.test(continuous=False)
    pred = model.get_current_visuals()["prediction"].numpy()
    assert pred.shape == (3, 32, 32) and maxabs(pred, g["last"]) <= FP32_TOL
    # the schedule buffers are registered under the reference's names
    keys = set(model.netG.state_dict())
    assert {"betas", "posterior_mean_coef1", "sqrt_recipm1_alphas_cumprod"} <= keys


def test_joint_indi_module():
    from diffsplitting_amd.model import networks
    sd, g = golden_state_dict("loop_joint_n3")
    sec = _tiny_indi_section(1, 1, "joint_indi")
    netG = networks.define_G(_opt(sec)).cuda()
    netG.load_state_dict(sd, strict=True)
    netG.set_new_noise_schedule({"n_timestep": 3}, "cuda")
    x_in = cases.make_cond("joint_loop").cuda()
    torch.manual_seed(cases.LOOP_SEED)
    netG.noise_source = lambda shape: torch.randn(shape)
    ret = netG.inference(x_in, continuous=True, t_float_start=0.5)
    assert maxabs(ret.cpu().numpy(), g["ret"]) <= FP32_TOL
    torch.manual_seed(cases.LOOP_SEED)
    last = netG.inference(x_in, continuous=False, t_float_start=0.3)
    assert maxabs(last.cpu().numpy(), g["last_t03"]) <= FP32_TOL
    # device-noise mode: both loops on two HIP streams, full batch available
    netG.noise_source = None
    out = netG.inference(x_in, continuous=False)
    torch.cuda.synchronize()
    assert out.shape == (1, 2, 32, 32) and netG.last_full_batch.shape == (2, 2, 32, 32)
    assert torch.isfinite(netG.last_full_batch).all()


def test_time_predictor_module():
    from diffsplitting_amd.model.ddpm_modules.time_predictor import TimePredictor
    sd, g = golden_state_dict("time_predictor")
    tp = TimePredictor(**cases.TIME_PRED_CFG).cuda()
    tp.load_state_dict(sd, strict=True)
    t = tp(cases.make_cond("time_pred").cuda())
    assert maxabs(t.cpu().numpy(), g["t"]) <= FP32_TOL


def test_checkpoint_round_trip(tmp_path):
    """save_network / load_network (model/model.py:131-173) with the reference's file naming."""
    from diffsplitting_amd.model import create_model
    sd, _ = golden_state_dict("loop_indi_n3_t1.0")
    opt = _opt(_tiny_indi_section())
    opt["path"]["checkpoint"] = str(tmp_path)
    m1 = create_model(opt)
    m1.netG.load_state_dict({"denoise_fn." + k: v for k, v in sd.items()})
    m1.save_network(epoch=1, iter_step=10)
    assert os.path.exists(tmp_path / "I10_E1_gen.pth")
    opt2 = _opt(_tiny_indi_section())
    opt2["path"]["resume_state"] = str(tmp_path / "I10_E1")
    m2 = create_model(opt2)
    x = cases.make_cond("indi_loop").cuda()
    for m in (m1, m2):
        m.netG.e = 0.0
    a = m1.netG.inference(x, num_timesteps=2)
    b = m2.netG.inference(x, num_timesteps=2)
    assert torch.equal(a, b)


def test_tiled_prediction_matches_reference_procedure():
    """predict_tiled (batched gather -> sampler -> stitch on the device) equals the reference's
    procedure (one tile per inference call, numpy stitch) done with the oracle; e = 0 so no RNG."""
    from diffsplitting_amd.data.tiled_predict import predict_tiled
    from diffsplitting_amd.model import networks
    sd, _ = golden_state_dict("unet_hagen_64")
    sec = _tiny_indi_section()
    sec["unet"]["channel_multiplier"] = [1, 2, 4, 8]
    netG = networks.define_G(_opt(sec)).cuda()
    netG.load_state_dict({"denoise_fn." + k: v for k, v in sd.items()})
    netG.e = 0.0
    rng = np.random.default_rng(5)
    frames = rng.standard_normal((2, 96, 160)).astype(np.float32)
    pred, plan = predict_tiled(netG, torch.from_numpy(frames).cuda(), patch_size=64, grid_size=32, batch_tiles=5,
                               sampler_kwargs=dict(num_timesteps=2))
    oplan = tiling.TilePlan((2, 96, 160), (1, 32, 32), (1, 64, 64))
    assert plan.total == oplan.total() == 2 * 2 * 4
    osd = {"denoise_fn." + k: v for k, v in sd.items()}
    cfg = cases.UNET_CASES["hagen_64"]["cfg"]
    tiles = []
    for i in range(oplan.total()):
        n, y, x = oplan.patch_location(i)
        t_in = torch.from_numpy(frames[n, y:y + 64, x:x + 64])[None, None]
        out = samplers.indi_inference(osd, cfg, t_in, 2, 2, randn=lambda s: torch.zeros(s), e=0.0)
        tiles.append(out[0].numpy())
    ref = tiling.stitch(np.stack(tiles), oplan)
    assert pred.shape == (2, 96, 160, 2)
    assert maxabs(pred.cpu().numpy(), ref) <= FP32_TOL


def test_stitch_predictions_mirror_numpy():
    from diffsplitting_amd.data.split_dataset_tiledpred import SplitDatasetTiledPred
    from diffsplitting_amd.data.tile_stitcher import stitch_predictions
    data = np.arange(3 * 128 * 128 * 2).reshape(3, 128, 128, 2).astype(np.float32)
    dset = SplitDatasetTiledPred(data, patch_size=64, grid_size=32)
    preds = np.stack([dset[i]["target"] for i in range(len(dset))])
    out = stitch_predictions(preds, dset.tile_manager)
    assert isinstance(out, np.ndarray) and np.array_equal(out, data)   # tests/test_tiling_setup.py invariant


def test_split_entry_point(tmp_path):
    """`split.py -c <config> -p val` on a config in the reference's schema (with // comments), synthetic frames."""
    from diffsplitting_amd import split
    sec = _tiny_indi_section()
    cfg = {"name": "tiny_hagen_indi", "phase": "train", "gpu_ids": [0],
           "path": {"log": "logs", "results": "results", "checkpoint": "checkpoint", "resume_state": None},
           "datasets": {"val": {"name": "Hagen", "patch_size": 64, "datatype": "img"}}, "model": sec}
    text = json.dumps(cfg, indent=2).replace('"name": "tiny_hagen_indi",', '"name": "tiny_hagen_indi", // comment')
    p = tmp_path / "tiny.json"
    p.write_text(text)
    pred = split.main(["-c", str(p), "-p", "val", "-gpu", "0", "-rootdir", str(tmp_path), "--synthetic", "2,128,128",
                       "--steps", "2", "--batch-tiles", "4"])
    assert pred.shape == (2, 128, 128, 2) and torch.isfinite(pred).all()


def test_psnr_metrics_on_device():
    from diffsplitting_amd.core.psnr import PSNR, RangeInvariantPsnr
    g = load_golden("psnr")
    gt, pred = torch.from_numpy(g["gt"]).cuda(), torch.from_numpy(g["pred"]).cuda()
    assert maxabs(PSNR(gt, pred).cpu().numpy(), g["psnr"]) < 1e-3
    assert maxabs(RangeInvariantPsnr(gt, pred).cpu().numpy(), g["ri_psnr"]) < 1e-3


def test_two_rank_predict_tiled():
    """End to end on 2 GPUs (skipped on a 1-GPU box): two fresh ranks started by parallel.self_launch, tiles sharded
    rank-strided, one RCCL all-gather, stitched result bitwise equal to the one-GPU prediction on every rank."""
    import subprocess
    import sys
    if torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r)\nfrom diffsplitting_amd import parallel\n"
            "sys.exit(parallel.self_launch(2, [%r]))\n" % (root, os.path.join(root, "tests", "multi_gpu_predict_tiled.py")))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "PREDICT_TILED_OK" in r.stdout, (r.stdout[-500:], r.stderr[-2000:])


@pytest.mark.parametrize("world", [2, 3])
def test_ranks_share_the_one_gpu_gloo_transport(world):
    """The N > 1 tiled path on the hardware that is there: `world` fresh ranks (parallel.self_launch), every rank its own
    libdsx / executors / tile shard on the ONE GPU, valid regions packed on the device, the exchange over gloo (staged
    through the host: RCCL refuses two ranks on one device), paste from the packed layout -- stitched result bitwise
    equal to the one-rank prediction on every rank.  What stays unexecuted on hardware is the RCCL transport itself."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r)\nfrom diffsplitting_amd import parallel\n"
            "sys.exit(parallel.self_launch(%d, [%r], timeout=500))\n" % (root, world, os.path.join(root, "tests", "multi_gpu_predict_tiled.py")))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["DSX_DIST_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "PREDICT_TILED_OK" in r.stdout, (r.stdout[-500:], r.stderr[-2000:])


def test_bench_two_ranks_share_the_one_gpu_gloo_transport():
    """`python bench.py --gpus 2` on the hardware that is there: self-launch, per-rank device / library state, barriers,
    MAX over ranks and the final gather of the images all run (gloo through the host instead of RCCL); the line says it
    is a rehearsal.  Both ranks run the B = 16 loop on the one GPU at once."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["DSX_DIST_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "2",
                        "--no-roofline", "--no-cpu-baseline", "--no-fp32-parity"], env=env, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["steps"] == 20 and "REHEARSAL" in line["config"]["parallelism"]
    assert line["config"]["global_batch"] == 32 and line["value"] > 0


def test_stitch_with_fused_range_invariant_psnr():
    """N1: the quality metric accumulated while stitching (no second pass over the canvas).  Fixture: psnr.npz holds
    gt / pred and RangeInvariantPsnr as the reference's core/psnr.py computed them; the prediction is cut into tiles,
    stitched back by the HIP kernel, and the fused metric must equal the reference's value."""
    from diffsplitting_amd.core.psnr import RangeInvariantPsnr
    from diffsplitting_amd.data.tiling import TilePlan
    g = load_golden("psnr")
    gt = torch.from_numpy(g["gt"]).cuda()
    pred = torch.from_numpy(g["pred"]).cuda()
    plan = TilePlan(tuple(gt.shape), (1, 8, 8), (1, 16, 16))
    tiles = plan.gather(pred).unsqueeze(1)
    canvas, ps = plan.stitch_with_psnr(tiles, gt.unsqueeze(-1))
    assert torch.equal(canvas[..., 0], pred)
    assert maxabs(ps[:, 0].cpu().numpy(), g["ri_psnr"]) < 1e-3, (ps.cpu().numpy(), g["ri_psnr"])
    # two channels, bigger frames, against the (golden-pinned) torch metric on the stitched canvas
    rng = np.random.default_rng(2)
    gt2 = torch.from_numpy(rng.standard_normal((2, 160, 224, 2)).astype(np.float32) * 3 + 1).cuda()
    pr2 = 0.6 * gt2 + 0.4 * torch.from_numpy(rng.standard_normal((2, 160, 224, 2)).astype(np.float32)).cuda() - 0.3
    plan2 = TilePlan((2, 160, 224), (1, 32, 32), (1, 64, 64))
    tiles2 = torch.stack([plan2.gather(pr2[..., c].contiguous()) for c in range(2)], dim=1)
    canvas2, ps2 = plan2.stitch_with_psnr(tiles2, gt2)
    assert torch.equal(canvas2, pr2)
    for c in range(2):
        ref = RangeInvariantPsnr(gt2[..., c], canvas2[..., c])
        assert maxabs(ps2[:, c].cpu().numpy(), ref.cpu().numpy()) < 1e-3
    _, ps3 = plan2.stitch_with_psnr(tiles2, gt2)
    assert torch.equal(ps2, ps3)                                     # fixed reduction order: bitwise reproducible


@pytest.mark.parametrize("nsteps", [1, 2])
def test_time_predictor_refinement_batched(nsteps):
    """N3 (core/psnr_based_t_refinement.py:14-57): TimePredictor -> per-tile start time -> both InDI samplers ->
    RangeInvariantPsnr scan, batched (one loop per sampler with per-sample step tables) against the fixture the
    reference's own classes produced sample by sample (oracle/gen_golden.py).  The reference draws its noise per
    sample (indi_1 start, steps; indi_2 start, steps); the draws are replayed in that order."""
    from diffsplitting_amd.core import psnr_based_t_refinement as R
    from diffsplitting_amd.model.ddpm_modules.time_predictor import TimePredictor
    from diffsplitting_amd.model.ddpm_modules.unet import UNet
    from diffsplitting_amd.model.samplers import InDISampler
    g = load_golden(f"refine_n{nsteps}")
    k1 = [(k, tuple(s)) for k, s in json.loads(bytes(g["keys1"]).decode())]
    k2 = [(k, tuple(s)) for k, s in json.loads(bytes(g["keys2"]).decode())]
    kt = [(k, tuple(s)) for k, s in json.loads(bytes(g["keys_tp"]).decode())]
    c = cases.UNET_CASES["joint_32"]["cfg"]

    def sampler(keys, seed):
        net = UNet(in_channel=1, out_channel=1, inner_channel=c["inner_channel"], norm_groups=c["norm_groups"],
                   channel_mults=c["channel_mults"], attn_res=c["attn_res"], res_blocks=c["res_blocks"], image_size=32)
        s = InDISampler(net, 32, channels=1, out_channel=1, conditional=False, val_schedule_opt={"n_timestep": nsteps}).cuda()
        s.load_state_dict(synth_state_dict(keys, seed), strict=True)
        s.set_new_noise_schedule({"n_timestep": nsteps}, "cuda")
        return s

    i1, i2 = sampler(k1, 1), sampler(k2, 2)
    tp = TimePredictor(**cases.TIME_PRED_CFG).cuda()
    tp.load_state_dict(synth_state_dict(kt, 0), strict=True)
    inp = cases.make_cond("time_pred")
    B = inp.shape[0]
    # the reference's draw sequence: for b: [i1 start, i1 steps..., i2 start, i2 steps...]
    torch.manual_seed(cases.LOOP_SEED)
    seq = {}
    for b in range(B):
        for name in ("i1", "i2"):
            seq[(name, b)] = [torch.randn(1, 1, 32, 32) for _ in range(1 + nsteps)]

    def source_for(name):
        order = [seq[(name, b)][k] for b in range(B) for k in range(1 + nsteps)]   # the batched sampler asks per sample
        it = iter(order)
        return lambda shape: next(it)

    i1.noise_source, i2.noise_source = source_for("i1"), source_for("i2")
    t_hat = R.get_time_prediction_from_classifier(inp, tp)
    assert maxabs(t_hat.cpu().numpy(), g["pred_t"]) <= FP32_TOL
    pred1, pred2 = R.get_channel_estimates(inp, i1, i2, tp, num_timesteps=nsteps)
    assert pred1.shape == g["pred1"].shape
    assert maxabs(pred1, g["pred1"]) <= FP32_TOL and maxabs(pred2, g["pred2"]) <= FP32_TOL, (maxabs(pred1, g["pred1"]), maxabs(pred2, g["pred2"]))
    i1.noise_source, i2.noise_source = source_for("i1"), source_for("i2")
    per_sample_t, concensus_t = R.estimate_time_using_PSNR(inp, i1, i2, tp, num_timesteps=nsteps)
    assert np.allclose(per_sample_t, g["per_sample_t"]) and abs(concensus_t - float(g["concensus_t"])) < 1e-9
    # MMSE over repeats with device noise: runs batched, finite, and averages (cells 60-62 of the notebook)
    i1.noise_source = i2.noise_source = None
    m1, m2 = R.get_channel_estimates(inp, i1, i2, tp, num_timesteps=nsteps, mmse_count=3)
    assert m1.shape == pred1.shape and np.isfinite(m1).all() and np.isfinite(m2).all()
    assert maxabs(m1, g["pred1"]) < 0.2                                 # e = 0.01 noise: close to the single estimate


def test_packed_weight_cache(tmp_path):
    """N4: load_network (model/model.py:153-166) repacks a checkpoint once; the second load of the same `*_gen.pth`
    uploads the cached image (keyed by the checkpoint hash, dtype, configuration, ABI) and computes bitwise the same."""
    from diffsplitting_amd.model import create_model
    from diffsplitting_amd.model.engine_unet import EngineUNet
    sd, _ = golden_state_dict("loop_indi_n3_t1.0")
    opt = _opt(_tiny_indi_section())
    opt["path"]["checkpoint"] = str(tmp_path)
    m1 = create_model(opt)
    m1.netG.load_state_dict({"denoise_fn." + k: v for k, v in sd.items()})
    m1.save_network(epoch=1, iter_step=10)
    x = cases.make_cond("indi_loop").cuda()

    def load():
        o = _opt(_tiny_indi_section())
        o["path"]["resume_state"] = str(tmp_path / "I10_E1")
        m = create_model(o)
        m.netG.e = 0.0
        out = m.netG.inference(x, num_timesteps=2)
        unet = [u for u in m.netG.modules() if isinstance(u, EngineUNet)][0]
        return out, unet.pack_cache_hit

    a, hit_a = load()
    assert hit_a is False and (tmp_path / "I10_E1_gen.unet0.f32.dsxpack").exists()
    b, hit_b = load()
    assert hit_b is True and torch.equal(a, b)
    # a different checkpoint under the same name must not reuse the image
    sd2 = {k: v + 0.01 for k, v in m1.netG.state_dict().items()}
    torch.save({k: v.cpu() for k, v in sd2.items()}, tmp_path / "I10_E1_gen.pth")
    c, hit_c = load()
    assert hit_c is False and not torch.equal(a, c)
    # a damaged payload of the right size is refused (SHA-256 of the payload in the header) and rewritten
    pk = tmp_path / "I10_E1_gen.unet0.f32.dsxpack"
    raw = bytearray(pk.read_bytes())
    raw[-5] ^= 0xFF
    pk.write_bytes(bytes(raw))
    d, hit_d = load()
    assert hit_d is False and torch.equal(c, d)
    # weights edited in place after load_network: the cached image no longer describes the module and is not used
    o = _opt(_tiny_indi_section())
    o["path"]["resume_state"] = str(tmp_path / "I10_E1")
    m = create_model(o)
    m.netG.e = 0.0
    unet = [u for u in m.netG.modules() if isinstance(u, EngineUNet)][0]
    with torch.no_grad():
        next(iter(unet.parameters())).add_(0.05)
    e = m.netG.inference(x, num_timesteps=2)
    assert unet.pack_cache_hit is not True and not torch.equal(e, d)


def _numpy_item(ch0, ch1, loc, p, nd, w, from_norm_target):
    """SplitDataset.__getitem__ (data/split_dataset.py:237-278) restated with numpy for one grid patch."""
    n, y, x = loc
    patch1 = ch0[n, y:y + p, x:x + p].astype(np.float32)[None]
    patch2 = ch1[n, y:y + p, x:x + p].astype(np.float32)[None]
    target = np.concatenate([patch1, patch2], axis=0)
    target = ((target - nd["mean_target"].reshape(-1, 1, 1)) / nd["std_target"].reshape(-1, 1, 1)).astype(np.float32)
    if from_norm_target:
        inp = w[0] * target[0:1] + w[1] * target[1:2]
    else:
        inp = w[0] * patch1 + w[1] * patch2
        inp = ((inp - nd["mean_input"]) / nd["std_input"]).astype(np.float32)
    return {"input": inp, "target": target}


def test_device_split_dataset_matches_numpy_restatement():
    """N2: frames resident on the GPU; normalisation statistics (compute_normalization_dict :29-74: quantiles) on the
    device equal numpy's; batches of normalised tiles from one HIP launch are bit-exact with __getitem__'s arithmetic.
    PARITY UNPINNED: data/split_dataset.py cannot be imported here (albumentations, skimage), so the comparison is
    with a numpy restatement of its arithmetic written in this file (`_numpy_item`), not with the reference itself;
    the reference's only fixture for this path (tests/test_tiling_setup.py) is the next test."""
    from diffsplitting_amd.data.split_dataset import (DataLocation, SplitDataset, SplitDatasetTiledPred,
                                                      compute_normalization_dict)
    rng = np.random.default_rng(11)
    # square frames: the reference's grid formula (patch_location :215-225) divides and wraps by h // patch only
    ch0 = (rng.gamma(2.0, 120.0, size=(3, 128, 128))).astype(np.float32)
    ch1 = (rng.gamma(3.0, 60.0, size=(3, 128, 128))).astype(np.float32)
    w = [1, 1]
    nd = compute_normalization_dict({0: torch.from_numpy(ch0).cuda(), 1: torch.from_numpy(ch1).cuda()}, w, q_val=0.98)
    d0, d1 = ch0.reshape(-1).astype(np.float64), ch1.reshape(-1).astype(np.float64)
    assert nd["target0_max"] == np.quantile(d0, 0.98)                          # numpy's linear-interpolation quantile
    assert nd["target1_max"] == np.quantile(d1, 0.98)
    assert nd["input_max"] == np.quantile(d0 * 1 + d1 * 1, 0.98) and isinstance(nd["mean_input"], np.float64)
    # the reference's frames are integer-typed .tif counts: numpy then works in float64, exactly this path
    i0, i1 = np.floor(ch0).astype(np.uint16), np.floor(ch1).astype(np.uint16)
    ndi = compute_normalization_dict({0: torch.from_numpy(i0.astype(np.int32)).cuda(), 1: torch.from_numpy(i1.astype(np.int32)).cuda()}, w, q_val=0.98)
    assert ndi["target0_max"] == np.quantile(i0.reshape(-1), 0.98) and ndi["target1_max"] == np.quantile(i1.reshape(-1), 0.98)
    for from_norm in (False, True):
        ds = SplitDataset("Hagen", DataLocation(arrays=(ch0, ch1)), 32, max_qval=0.98, channel_weights=w,
                          input_from_normalized_target=from_norm)
        assert len(ds) == 3 * 4 * 4
        ids = [0, 7, 14, 47]
        batch = ds.tiles(ids)
        assert batch["input"].shape == (4, 1, 32, 32) and batch["target"].shape == (4, 2, 32, 32)
        for k, i in enumerate(ids):
            ref = _numpy_item(ch0, ch1, ds.patch_location(i), 32, ds.get_normalization_dict(), w, from_norm)
            assert np.array_equal(batch["target"][k].cpu().numpy(), ref["target"])
            assert np.array_equal(batch["input"][k].cpu().numpy(), ref["input"])
            item = ds[i]
            assert np.array_equal(item["input"], ref["input"]) and np.array_equal(item["target"], ref["target"])
    # upper_clip (:147-150) and the tiled-prediction subclass
    dt = SplitDatasetTiledPred("Hagen", DataLocation(arrays=(ch0, ch1)), 64, grid_size=32, max_qval=0.98, upper_clip=True)
    assert len(dt) == dt.plan.total == 3 * 3 * 3
    c0 = np.clip(ch0, 0, dt.get_normalization_dict()["target0_max"])
    c1 = np.clip(ch1, 0, dt.get_normalization_dict()["target1_max"])
    ref = _numpy_item(c0, c1, dt.patch_location(5), 64, dt.get_normalization_dict(), [1, 1], False)
    got = dt.tiles([5])
    assert np.array_equal(got["target"][0].cpu().numpy(), ref["target"]) and np.array_equal(got["input"][0].cpu().numpy(), ref["input"])
    # a shard (arithmetic id sequence) goes through the plan's device tables, an arbitrary list through a per-call
    # upload: same tiles; the whole-frame ground truth equals the stitched target tiles
    shard = list(range(1, len(dt), 4))
    by_seq = dt.tiles(shard)
    by_loc = dt.tiles_at([dt.patch_location(i) for i in shard])
    assert torch.equal(by_seq["input"], by_loc["input"]) and torch.equal(by_seq["target"], by_loc["target"])
    gt = dt.normalized_target_frames()
    assert gt.shape == (3, 128, 128, 2)
    assert torch.equal(gt, dt.plan.stitch(dt.tiles(range(len(dt)))["target"]))


def test_reference_known_answer_through_the_device_dataset():
    """tests/test_tiling_setup.py of the reference (arange frames, identity normalisation, every tile's own target as
    the prediction, stitched == data exactly) with the device-resident dataset and the HIP stitch."""
    from diffsplitting_amd.data.split_dataset import DataLocation, SplitDatasetTiledPred
    from diffsplitting_amd.data.tile_stitcher import stitch_predictions
    data = np.arange(5 * 512 * 512 * 2).reshape(5, 512, 512, 2).astype(np.float32)      # < 2^24: exact in fp32
    nd = {"mean_input": 0, "std_input": 1, "mean_target": np.array([0, 0]), "std_target": np.array([1, 1]),
          "target0_max": 1, "target1_max": 1, "input_max": 1}
    ds = SplitDatasetTiledPred("Hagen", DataLocation(arrays=(data[..., 0], data[..., 1])), 256, grid_size=128,
                               max_qval=0.98, upper_clip=False, normalization_dict=nd)
    assert len(ds) == 45
    preds = ds.tiles(range(len(ds)))["target"]
    out = ds.plan.stitch(preds)
    assert np.array_equal(out.cpu().numpy(), data)
    assert np.array_equal(stitch_predictions(preds.cpu().numpy(), ds.tile_manager), data)
