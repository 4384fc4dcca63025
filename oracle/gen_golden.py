"""Generate tests/golden/*.npz by running the REFERENCE's own classes.

Runs only in the build container (needs /root/reference); never shipped to or
run on the GPU box.  Run as:  ``python -O oracle/gen_golden.py``  (``-O`` strips
the reference's drift assert, indi.py:64, so n=20/100 can be recorded; the
arithmetic is unchanged — SURVEY §8c).

Fixtures hold only inputs' seeds/shapes, state-dict key lists and expected
outputs; weights are re-synthesised from the key lists (oracle/weights.py).
"""
import json
import os
import sys

import numpy as np
import torch

REF = os.environ.get("DSX_REFERENCE", "/root/reference")
sys.path.insert(0, REF)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import cases  # noqa: E402
from oracle.weights import synth_state_dict  # noqa: E402

from model.sr3_modules.unet import UNet as UNetSr3  # noqa: E402
from model.sr3_modules.diffusion import GaussianDiffusion as GDSr3  # noqa: E402
from model.ddpm_modules.unet import UNet as UNetDdpm  # noqa: E402
from model.ddpm_modules.diffusion import GaussianDiffusion as GDDdpm  # noqa: E402
from model.ddpm_modules.indi import InDI  # noqa: E402
from model.ddpm_modules.joint_indi import JointIndi  # noqa: E402
from model.ddpm_modules.time_predictor import TimePredictor  # noqa: E402
from data.tiling_manager import TileIndexManager, TilingMode  # noqa: E402
from data.tile_stitcher import stitch_predictions  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
os.makedirs(OUT, exist_ok=True)
torch.set_grad_enabled(False)


def key_shapes(module, skip_prefixes=()):
    return [(k, list(v.shape)) for k, v in module.state_dict().items()
            if not any(k.startswith(p) for p in skip_prefixes)]


def load_synth(module, ks, seed=0):
    sd = synth_state_dict(ks, seed)
    missing, unexpected = module.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    return sd


def build_unet(case):
    cls = UNetSr3 if case["flavour"] == "sr3" else UNetDdpm
    net = cls(**case["cfg"]).eval()
    ks = key_shapes(net)
    load_synth(net, ks)
    return net, ks


def save(name, **arrs):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrs)
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB")


def jstr(obj):
    return np.frombuffer(json.dumps(obj).encode(), dtype=np.uint8)


# ---------------------------------------------------------------- UNet forwards
for name, case in cases.UNET_CASES.items():
    net, ks = build_unet(case)
    x, t = cases.make_unet_inputs(name)
    y = net(x, t)
    save(f"unet_{name}", keys=jstr(ks), y=y.numpy())

# ---------------------------------------------------------------- schedules
for name, opt in cases.SCHEDULES.items():
    gd = GDSr3(None, 32, channels=3, conditional=True)
    gd.set_new_noise_schedule(opt, "cpu")
    arrs = {k: v.numpy() for k, v in gd.state_dict().items()}
    arrs["sqrt_alphas_cumprod_prev_f64"] = gd.sqrt_alphas_cumprod_prev
    save(f"schedule_{name}", **arrs)

# ---------------------------------------------------------------- SR3 loops
case = cases.UNET_CASES["sr3_tiny"]
for sched in ("lin_8", "lin_25"):
    net, ks = build_unet(case)
    gd = GDSr3(net, 32, channels=3, conditional=True).eval()
    gd.set_new_noise_schedule(cases.SCHEDULES[sched], "cpu")
    cond = cases.make_cond("sr3_loop")
    torch.manual_seed(cases.LOOP_SEED)
    ret = gd.p_sample_loop(cond, clip_denoised=True, continous=True)
    torch.manual_seed(cases.LOOP_SEED)
    last = gd.super_resolution(cond)
    save(f"loop_sr3_{sched}", keys=jstr(ks), ret=ret.numpy(), last=last.numpy())

# ---------------------------------------------------------------- DDPM loop (conditional)
case = cases.DDPM_COND_CASE
net = UNetDdpm(**case["cfg"]).eval()
ks = key_shapes(net)
load_synth(net, ks)
gd = GDDdpm(net, 32, channels=1, conditional=True).eval()
gd.set_new_noise_schedule(cases.SCHEDULES["lin_8"], "cpu")
cond = cases.make_cond("ddpm_loop")
torch.manual_seed(cases.LOOP_SEED)
ret = gd.p_sample_loop(cond, clip_denoised=True, continous=True)
save("loop_ddpm_lin_8", keys=jstr(ks), ret=ret.numpy())

# ---------------------------------------------------------------- InDI loops
case = cases.UNET_CASES["ddpm_tiny"]
for n, t0 in ((1, 1.0), (3, 1.0), (10, 1.0), (20, 1.0), (4, 0.6)):
    net, ks = build_unet(case)
    indi = InDI(net, 32, channels=2, out_channel=2, conditional=False,
                val_schedule_opt={"n_timestep": n}).eval()
    indi.set_new_noise_schedule({"n_timestep": n}, "cpu")
    x_in = cases.make_cond("indi_loop")
    torch.manual_seed(cases.LOOP_SEED)
    ret = indi.inference(x_in, continuous=True, t_float_start=t0)
    torch.manual_seed(cases.LOOP_SEED)
    last = indi.inference(x_in, continuous=False, t_float_start=t0)
    save(f"loop_indi_n{n}_t{t0}", keys=jstr(ks), ret=ret.numpy(), last=last.numpy())

# BASELINE C1 (config/splitting_cifar10_indi.json): UNet 6 -> 6, x_in (4, 1, 32, 32) replicated x6, n = 20 / 100
case = cases.C1_CASE
for n in cases.C1_STEPS:
    net, ks = build_unet(case)
    indi = InDI(net, 32, channels=6, out_channel=6, conditional=False, val_schedule_opt={"n_timestep": n}).eval()
    indi.set_new_noise_schedule({"n_timestep": n}, "cpu")
    x_in = cases.make_cond("c1_cifar")
    torch.manual_seed(cases.LOOP_SEED)
    ret = indi.inference(x_in, continuous=True, t_float_start=1.0)
    torch.manual_seed(cases.LOOP_SEED)
    last = indi.inference(x_in, continuous=False, t_float_start=1.0)
    # ret stacks (1 + snapshots) batches of 4: the fixture keeps the noisy input, the first snapshot, a middle one and
    # the final state (cases.C1_KEEP blocks of 4 images) plus the non-continuous return
    blocks = ret.numpy().reshape(-1, 4, 6, 32, 32)
    keep = cases.c1_keep(blocks.shape[0])
    save(f"loop_c1_cifar_n{n}", keys=jstr(ks), nblocks=np.int64(blocks.shape[0]), blocks=blocks[keep], last=last.numpy())

# InDI t sequence + coefficient pin with a stub denoiser
tseq = {}
for n, t0 in cases.INDI_T_CASES:
    rec = []

    def stub(x, t, rec=rec):
        rec.append(t.clone())
        return 0.5 * x

    indi = InDI(stub, 8, channels=1, out_channel=1, conditional=False,
                val_schedule_opt={"n_timestep": n})
    indi.set_new_noise_schedule({"n_timestep": n}, "cpu")
    torch.manual_seed(5)
    x_in = torch.randn(1, 1, 4, 4)
    out = indi.inference(x_in, continuous=False, t_float_start=t0)
    tseq[f"t_n{n}_t{t0}"] = torch.cat(rec).numpy()
    tseq[f"x_n{n}_t{t0}"] = out.numpy()
save("indi_tseq", **tseq)

# ---------------------------------------------------------------- JointIndi
case = cases.UNET_CASES["joint_32"]
n1 = UNetDdpm(**case["cfg"]).eval()
n2 = UNetDdpm(**case["cfg"]).eval()
joint = JointIndi(None, 32, channels=1, out_channel=1, denoise_fn_ch1=n1, denoise_fn_ch2=n2,
                  conditional=False, val_schedule_opt={"n_timestep": 3}).eval()
ks = key_shapes(joint)
load_synth(joint, ks)
joint.set_new_noise_schedule({"n_timestep": 3}, "cpu")
x_in = cases.make_cond("joint_loop")
torch.manual_seed(cases.LOOP_SEED)
ret = joint.inference(x_in, continuous=True, t_float_start=0.5)
torch.manual_seed(cases.LOOP_SEED)
last = joint.inference(x_in, continuous=False, t_float_start=0.3)
save("loop_joint_n3", keys=jstr(ks), ret=ret.numpy(), last_t03=last.numpy())

# ---------------------------------------------------------------- TimePredictor
tp = TimePredictor(**cases.TIME_PRED_CFG).eval()
ks = key_shapes(tp)
load_synth(tp, ks)
x = cases.make_cond("time_pred")
save("time_predictor", keys=jstr(ks), t=tp(x).numpy())

# ---------------------------------------------------------------- state-dict key lists of the BASELINE configs
from core.logger import dict_to_nonedict  # noqa: E402
import model.networks as networks  # noqa: E402

keylists = {}
for cfgname in ("splitting_cifar10_indi", "splitting_hagen_indi", "splitting_hagen_indi_joint",
                "splitting_hagen_indi_single_ch"):
    raw = cases.load_config_json(os.path.join(REF, "config", cfgname + ".json"))
    raw["phase"] = "val"
    raw["gpu_ids"] = None
    raw["distributed"] = False
    opt = dict_to_nonedict(raw)
    netG = networks.define_G(opt)
    keylists[cfgname] = key_shapes(netG)
for cfgname in ("sr_sr3_16_128", "sr_sr3_64_512", "sr_ddpm_16_128"):
    raw = cases.load_config_json(os.path.join(REF, "config", cfgname + ".json"))
    u = raw["model"]["unet"]
    cls = UNetSr3 if raw["model"]["which_model_G"] == "sr3" else UNetDdpm
    net = cls(in_channel=u["in_channel"], out_channel=u["out_channel"],
              norm_groups=u.get("norm_groups") or 32, inner_channel=u["inner_channel"],
              channel_mults=u["channel_multiplier"], attn_res=u["attn_res"],
              res_blocks=u["res_blocks"], dropout=u["dropout"],
              image_size=raw["model"]["diffusion"]["image_size"])
    keylists[cfgname] = [("denoise_fn." + k, s) for k, s in key_shapes(net)]
# the hyper-parameters the key lists were built from (model section of each config)
modelcfgs = {}
for cfgname in keylists:
    raw = cases.load_config_json(os.path.join(REF, "config", cfgname + ".json"))
    modelcfgs[cfgname] = raw["model"]
with open(os.path.join(OUT, "state_dict_keys.json"), "w") as f:
    json.dump({"keys": keylists, "model": modelcfgs}, f)
print("state_dict_keys.json:", {k: len(v) for k, v in keylists.items()})

# ---------------------------------------------------------------- tiling
for name, data_shape, grid_shape, patch_shape in cases.TILE_CASES:
    mng = TileIndexManager(data_shape, grid_shape, patch_shape, TilingMode.ShiftBoundary)
    T = mng.total_grid_count()
    locs = np.array([mng.get_location_from_dataset_idx(i) for i in range(T)], dtype=np.int64)
    plocs = np.array([mng.get_patch_location_from_dataset_idx(i) for i in range(T)], dtype=np.int64)
    arrs = dict(total=np.int64(T), locs=locs, plocs=plocs)
    if name in ("ragged", "single_tile", "grid_eq_patch"):
        rng = np.random.default_rng(3)
        pred = rng.standard_normal((T, 2, patch_shape[1], patch_shape[2])).astype(np.float32)
        arrs["stitched"] = stitch_predictions(pred, mng)
    save(f"tiles_{name}", **arrs)

# ---------------------------------------------------------------- PSNR metrics (core/psnr.py)
from core.psnr import PSNR, RangeInvariantPsnr  # noqa: E402
gp = torch.Generator().manual_seed(9)
gt = torch.randn((3, 40, 56), generator=gp) * 2 + 1
pred = 0.7 * gt + 0.3 * torch.randn((3, 40, 56), generator=gp) - 0.5
save("psnr", gt=gt.numpy(), pred=pred.numpy(), psnr=PSNR(gt, pred).numpy(),
     ri_psnr=RangeInvariantPsnr(gt, pred).numpy())


# ---------------------------------------------------------------- full-size cases (BASELINE C3 / C4 / C5 shapes)
# Outputs are 512^2: the fixtures keep digests (cases.digest) of the reference's outputs.
def save_digest(name, ks, y, **extra):
    d = cases.digest(y)
    save(name, keys=jstr(ks), **d, **extra)


# C4: the sr_sr3_64_512 UNet, one forward, B = 1
case = cases.FULLSIZE_CASES["c4_sr3_512"]
net, ks = build_unet(case)
x = cases.make_fullsize_input("c4_x", (1, 6, 512, 512))
t = torch.tensor([[0.613]])
save_digest("full_c4_unet", ks, net(x, t).numpy())
del net

# C3: the Hagen UNet, InDI.inference n = 3 on one 512^2 tile
case = cases.FULLSIZE_CASES["c3_hagen_512"]
net, ks = build_unet(case)
indi = InDI(net, 32, channels=2, out_channel=2, conditional=False, val_schedule_opt={"n_timestep": 3}).eval()
indi.set_new_noise_schedule({"n_timestep": 3}, "cpu")
x_in = cases.make_fullsize_input("c3_x", (1, 1, 512, 512))
torch.manual_seed(cases.LOOP_SEED)
ret = indi.inference(x_in, continuous=True, t_float_start=1.0)
save_digest("full_c3_indi", ks, ret.numpy())
del net, indi

# C5: JointIndi n = 3 on one 512^2 tile + the TimePredictor on two tiles
case = cases.FULLSIZE_CASES["c5_joint_512"]
n1 = UNetDdpm(**case["cfg"]).eval()
n2 = UNetDdpm(**case["cfg"]).eval()
joint = JointIndi(None, 32, channels=1, out_channel=1, denoise_fn_ch1=n1, denoise_fn_ch2=n2,
                  conditional=False, val_schedule_opt={"n_timestep": 3}).eval()
ks = key_shapes(joint)
load_synth(joint, ks)
joint.set_new_noise_schedule({"n_timestep": 3}, "cpu")
x_in = cases.make_fullsize_input("c5_x", (1, 1, 512, 512))
torch.manual_seed(cases.LOOP_SEED)
ret = joint.inference(x_in, continuous=True, t_float_start=0.5)
save_digest("full_c5_joint", ks, ret.numpy())
del joint, n1, n2

tp = TimePredictor(**case["cfg"]).eval()     # config/splitting_hagen_time_predictor.json: the same 1->1 UNet, no time embedding
ks = key_shapes(tp)
load_synth(tp, ks)
x = cases.make_fullsize_input("c5_tp_x", (2, 1, 512, 512))
save("full_c5_timepred", keys=jstr(ks), t=tp(x).numpy())

# ---------------------------------------------------------------- TimePredictor-driven refinement (N3)
# core/psnr_based_t_refinement.py cannot be imported (it imports the external `disentangle` package, :10, and calls
# InDI.p_sample_loop with the signature of what is now InDI.inference).  Its 30 lines are followed here literally
# with the reference's own classes: TimePredictor, InDI.inference (batch 1, sample by sample), core.psnr.
case = cases.UNET_CASES["joint_32"]
for nsteps in (1, 2):
    i1 = InDI(UNetDdpm(**case["cfg"]).eval(), 32, channels=1, out_channel=1, conditional=False,
              val_schedule_opt={"n_timestep": nsteps}).eval()
    i2 = InDI(UNetDdpm(**case["cfg"]).eval(), 32, channels=1, out_channel=1, conditional=False,
              val_schedule_opt={"n_timestep": nsteps}).eval()
    ks1, ks2 = key_shapes(i1), key_shapes(i2)
    load_synth(i1, ks1, seed=1)
    load_synth(i2, ks2, seed=2)
    tp = TimePredictor(**cases.TIME_PRED_CFG).eval()
    kst = key_shapes(tp)
    load_synth(tp, kst)
    inp = cases.make_cond("time_pred")                              # (3, 1, 32, 32)
    pred_t_2 = tp(inp)                                              # get_time_prediction_from_classifier (:14-17)
    pred_t_1 = 1 - pred_t_2
    torch.manual_seed(cases.LOOP_SEED)
    p1, p2 = [], []
    for b in range(inp.shape[0]):                                   # get_channel_estimates (:20-39)
        p1.append(i1.inference(inp[b:b + 1], continuous=False, num_timesteps=nsteps, t_float_start=pred_t_1[b].item()).numpy())
        p2.append(i2.inference(inp[b:b + 1], continuous=False, num_timesteps=nsteps, t_float_start=pred_t_2[b].item()).numpy())
    pred1, pred2 = np.concatenate(p1, axis=0), np.concatenate(p2, axis=0)
    gt = inp.numpy()[:, 0]                                          # estimate_time_using_PSNR (:41-57)
    t_list = np.arange(0, 1.0, 0.05)
    psnr_list = [RangeInvariantPsnr(gt, pred1[:, 0] * t + pred2[:, 0] * (1 - t)) for t in t_list]
    psnr_matrix = torch.stack(psnr_list)
    per_sample_t = t_list[psnr_matrix.argmax(dim=0)]
    concensus_t = t_list[psnr_matrix.mean(dim=1).argmax()]
    save(f"refine_n{nsteps}", keys1=jstr(ks1), keys2=jstr(ks2), keys_tp=jstr(kst), pred_t=pred_t_2.numpy(),
         pred1=pred1, pred2=pred2, psnr=psnr_matrix.numpy(), per_sample_t=np.asarray(per_sample_t),
         concensus_t=np.float64(concensus_t))
