"""``define_G(opt)`` — same signature and config keys as model/networks.py:91-180
of the reference; builds engine-backed UNet(s) and the sampler around them.

Superset of the reference (SURVEY rot list): ``sr3`` / ``ddpm`` models build too
(R1), and an extra ``model.compute_dtype`` key ("f32" default, "bf16") selects
the MFMA operand type.
"""
import logging

from .ddpm_modules.unet import UNet as UNetDdpm
from .samplers import GaussianSampler, GaussianSamplerDdpm, InDISampler, JointIndiSampler
from .sr3_modules.unet import UNet as UNetSr3

logger = logging.getLogger("base")

_FAMILIES = {
    "sr3": (GaussianSampler, UNetSr3),
    "ddpm": (GaussianSamplerDdpm, UNetDdpm),
    "indi": (InDISampler, UNetDdpm),
    "joint_indi": (JointIndiSampler, UNetDdpm),
}


def _get(d, key, default=None):
    try:
        v = d[key]
    except (KeyError, TypeError):
        return default
    return default if v is None else v


def _build_unet(unet_cls, model_opt):
    u = model_opt["unet"]
    return unet_cls(in_channel=u["in_channel"], out_channel=u["out_channel"], norm_groups=u["norm_groups"],
                    inner_channel=u["inner_channel"], channel_mults=u["channel_multiplier"],
                    attn_res=u["attn_res"], res_blocks=u["res_blocks"], dropout=u["dropout"],
                    image_size=model_opt["diffusion"]["image_size"])


def define_G(opt):
    model_opt = opt["model"]
    if _get(model_opt["unet"], "norm_groups") is None:
        model_opt["unet"]["norm_groups"] = 32                    # networks.py:95-96 (mutates opt, Q9)
    which = model_opt["which_model_G"]
    if which not in _FAMILIES:
        raise NotImplementedError("Generator model [{:s}] not recognized".format(str(which)))
    sampler_cls, unet_cls = _FAMILIES[which]
    kwargs = {}
    if which == "joint_indi":
        kwargs["allow_full_translation"] = _get(model_opt, "allow_full_translation", False)
        kwargs["w_input_loss"] = _get(model_opt, "w_input_loss", 0.0)
        kwargs["denoise_fn_ch1"] = _build_unet(unet_cls, model_opt)
        kwargs["denoise_fn_ch2"] = _build_unet(unet_cls, model_opt)
        unet = None
    else:
        unet = _build_unet(unet_cls, model_opt)
    netG = sampler_cls(unet, image_size=model_opt["diffusion"]["image_size"],
                       channels=model_opt["diffusion"]["channels"], loss_type=_get(model_opt, "loss_type", "l1"),
                       out_channel=model_opt["unet"]["out_channel"], lr_reduction=_get(model_opt, "lr_reduction"),
                       conditional=model_opt["diffusion"]["conditional"],
                       schedule_opt=model_opt["beta_schedule"]["train"],
                       val_schedule_opt=model_opt["beta_schedule"]["val"], **kwargs)
    dtype = _get(model_opt, "compute_dtype", "f32")
    for m in netG.modules():
        if hasattr(m, "compute_dtype"):
            m.compute_dtype = dtype
    if _get(opt, "phase") == "train":
        logger.warning("phase == 'train': the MI355X engine is inference-only; weights keep their initial values")
    # nn.DataParallel (networks.py:177-179) is never applied: multi-GPU inference is one process per
    # GPU (torchrun) with tile / batch sharding, see diffsplitting_amd/parallel.py
    return netG
