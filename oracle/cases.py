"""Shared case definitions for golden generation and tests (test infrastructure only).

UNet configs use the keyword names of the reference ``UNet.__init__``
(sr3 unet.py:162-174): in_channel, out_channel, inner_channel, norm_groups,
channel_mults, attn_res, res_blocks, image_size.
"""

UNET_CASES = {
    # straddling GroupNorm groups on concatenated inputs (192/32, 96/32), attention at 16² and 8²
    "sr3_tiny": dict(flavour="sr3", B=2, H=32, W=32,
                     cfg=dict(in_channel=6, out_channel=3, inner_channel=32, norm_groups=32,
                              channel_mults=(1, 2, 4), attn_res=(16,), res_blocks=2, image_size=32)),
    # InDI-style: one scalar t for the whole batch, 16 groups, no attn_res (mid attention only)
    "ddpm_tiny": dict(flavour="ddpm", B=3, H=32, W=48,
                      cfg=dict(in_channel=2, out_channel=2, inner_channel=16, norm_groups=16,
                               channel_mults=(1, 2, 4), attn_res=(), res_blocks=1, image_size=32)),
    # the headline config (config/sr_sr3_16_128.json), one image
    "sr3_128": dict(flavour="sr3", B=1, H=128, W=128,
                    cfg=dict(in_channel=6, out_channel=3, inner_channel=64, norm_groups=32,
                             channel_mults=(1, 2, 4, 8, 8), attn_res=(16,), res_blocks=2, image_size=128)),
    # config/splitting_hagen_indi.json UNet on a 64² tile (bottleneck 8² attention, d=128)
    "hagen_64": dict(flavour="ddpm", B=2, H=64, W=64,
                     cfg=dict(in_channel=2, out_channel=2, inner_channel=16, norm_groups=16,
                              channel_mults=(1, 2, 4, 8), attn_res=(), res_blocks=1, image_size=32)),
    # config/splitting_hagen_indi_joint.json UNet (1 -> 1 channel)
    "joint_32": dict(flavour="ddpm", B=2, H=32, W=32,
                     cfg=dict(in_channel=1, out_channel=1, inner_channel=16, norm_groups=16,
                              channel_mults=(1, 2, 4, 8), attn_res=(), res_blocks=1, image_size=32)),
}

# BASELINE C1: config/splitting_cifar10_indi.json:43-44,45-62,71 -- UNet 6 -> 6, inner 16, mults [1,2,4,8], GN16, rb 1,
# no attn_res; InDI replicates the 1-channel x_in out_channel = 6 times (indi.py:80); batch 4 at 32^2, n = 20 (the file)
# and n = 100 (BASELINE.json); both step counts trip the reference's drift assert: generated under -O
C1_CASE = dict(flavour="ddpm", B=4, H=32, W=32,
               cfg=dict(in_channel=6, out_channel=6, inner_channel=16, norm_groups=16,
                        channel_mults=(1, 2, 4, 8), attn_res=(), res_blocks=1, image_size=32))
C1_STEPS = (20, 100)


def c1_keep(nblocks):
    """Blocks (of 4 images) of the continuous C1 return that the fixture keeps."""
    return [0, 1, nblocks // 2, nblocks - 1]

# Full-size cases at the shapes BASELINE.json names (C3 / C4 / C5): fixtures hold digests of the reference's
# outputs (crops, strided grids, per-channel sums), the GPU tests compare whole tensors with the oracle.
_HAGEN = dict(inner_channel=16, norm_groups=16, channel_mults=(1, 2, 4, 8), attn_res=(), res_blocks=1, image_size=32)
FULLSIZE_CASES = {
    # config/sr_sr3_64_512.json: mults [1,2,4,8,16], 2048->1024 convs, bottleneck attention L=1024 d=1024, GN16, rb=1
    "c4_sr3_512": dict(flavour="sr3", B=1, H=512, W=512,
                       cfg=dict(in_channel=6, out_channel=3, inner_channel=64, norm_groups=16,
                                channel_mults=(1, 2, 4, 8, 16), attn_res=(), res_blocks=1, image_size=512)),
    # config/splitting_hagen_indi.json on one 512^2 tile: bottleneck attention L=4096 d=128
    "c3_hagen_512": dict(flavour="ddpm", B=1, H=512, W=512, cfg=dict(in_channel=2, out_channel=2, **_HAGEN)),
    # config/splitting_hagen_indi_joint.json: two 1->1 UNets; config/splitting_hagen_time_predictor.json: the TimePredictor
    "c5_joint_512": dict(flavour="ddpm", B=1, H=512, W=512, cfg=dict(in_channel=1, out_channel=1, **_HAGEN)),
}


def digest(y):
    """Size-bounded digest of an (N, C, H, W) array: centre crop, strided grid over the whole image (borders
    included), a corner, and float64 per-(n, c) sums / sums of squares."""
    import numpy as np
    y = np.asarray(y)
    H, W = y.shape[-2:]
    cy, cx = H // 2 - 16, W // 2 - 16
    y64 = y.astype(np.float64)
    return dict(crop=y[..., cy:cy + 32, cx:cx + 32].copy(), grid=y[..., ::16, ::16].copy(),
                corner=y[..., :16, :16].copy(), chsum=y64.sum(axis=(-2, -1)), chsq=(y64 * y64).sum(axis=(-2, -1)),
                shape=np.asarray(y.shape, dtype=np.int64))


def make_fullsize_input(name, shape):
    import torch
    import zlib
    g = torch.Generator().manual_seed(zlib.crc32(("full_" + name).encode()) & 0x7FFFFFFF)
    return torch.randn(shape, generator=g)


SCHEDULES = {
    "sr3_2000": dict(schedule="linear", n_timestep=2000, linear_start=1e-6, linear_end=1e-2),
    "lin_8": dict(schedule="linear", n_timestep=8, linear_start=1e-4, linear_end=2e-1),
    "lin_25": dict(schedule="linear", n_timestep=25, linear_start=1e-4, linear_end=5e-2),
    "cos_12": dict(schedule="cosine", n_timestep=12, linear_start=1e-4, linear_end=2e-2),
    "warm_20": dict(schedule="warmup10", n_timestep=20, linear_start=1e-4, linear_end=2e-2),
}

# (name, data_shape, grid_shape, patch_shape) — ShiftBoundary, as SplitDatasetTiledPred builds them
TILE_CASES = [
    ("ref_test_45", (5, 512, 512), (1, 128, 128), (1, 256, 256)),       # tests/test_tiling_setup.py
    ("hagen_490", (10, 2048, 2048), (1, 256, 256), (1, 512, 512)),     # EvaluateJointIndi cell 6
    ("ragged", (2, 150, 210), (1, 16, 16), (1, 32, 32)),
    ("single_tile", (2, 64, 64), (1, 32, 32), (1, 64, 64)),
    ("grid_eq_patch", (2, 96, 128), (1, 32, 32), (1, 32, 32)),
]

INDI_T_CASES = [(1, 1.0), (2, 1.0), (3, 1.0), (3, 0.5), (5, 1.0), (10, 1.0), (20, 1.0), (100, 1.0),
                (7, 0.3), (2000, 1.0)]

DDPM_COND_CASE = dict(flavour="ddpm", B=2, H=32, W=32,
                      cfg=dict(in_channel=2, out_channel=1, inner_channel=16, norm_groups=8,
                               channel_mults=(1, 2), attn_res=(16,), res_blocks=1, image_size=32))

TIME_PRED_CFG = dict(in_channel=1, out_channel=1, inner_channel=16, norm_groups=16,
                     channel_mults=(1, 2, 4), attn_res=(), res_blocks=1, image_size=32)

LOOP_SEED = 20250225

_COND_SHAPES = {
    "sr3_loop": (2, 3, 32, 32),
    "ddpm_loop": (2, 1, 32, 32),
    "indi_loop": (3, 1, 32, 48),
    "joint_loop": (2, 1, 32, 32),
    "time_pred": (3, 1, 32, 32),
    "c1_cifar": (4, 1, 32, 32),
}


def make_cond(name):
    import torch
    import zlib
    g = torch.Generator().manual_seed(zlib.crc32(name.encode()) & 0x7FFFFFFF)
    return torch.randn(_COND_SHAPES[name], generator=g)


def make_unet_inputs(name):
    """Seeded (x, time) for UNET_CASES[name]."""
    import torch
    import zlib
    case = UNET_CASES[name]
    g = torch.Generator().manual_seed(zlib.crc32(("unet_" + name).encode()) & 0x7FFFFFFF)
    B, H, W = case["B"], case["H"], case["W"]
    x = torch.randn((B, case["cfg"]["in_channel"], H, W), generator=g)
    if case["flavour"] == "sr3":
        t = 0.05 + 0.95 * torch.rand((B, 1), generator=g)          # gamma per sample (B,1)
    elif name == "joint_32":
        t = torch.tensor([3, 1700][:B], dtype=torch.long)          # DDPM-style integer t (B,)
    else:
        t = torch.tensor([0.37 if name == "ddpm_tiny" else 0.8])   # InDI-style one scalar (1,)
    return x, t


def load_config_json(path):
    """core/logger.py:20-27,35-40: JSON with '//' comments stripped line-wise."""
    import json
    from collections import OrderedDict
    s = ""
    with open(path) as f:
        for line in f:
            s += line.split("//")[0] + "\n"
    return json.loads(s, object_pairs_hook=OrderedDict)
