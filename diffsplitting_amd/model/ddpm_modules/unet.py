"""model/ddpm_modules/unet.py of the reference: the t-conditioned UNet (HIP engine)."""
from ..engine_unet import EngineUNet


class UNet(EngineUNet):
    flavour = "ddpm"

    def __init__(self, in_channel=6, out_channel=3, inner_channel=32, norm_groups=32,
                 channel_mults=(1, 2, 4, 8, 8), attn_res=(8,), res_blocks=3, dropout=0,
                 with_time_emb=True, image_size=128):
        super().__init__(in_channel, out_channel, inner_channel, norm_groups, channel_mults, attn_res,
                         res_blocks, dropout, with_time_emb, image_size)
