"""model/ddpm_modules/time_predictor.py of the reference on the HIP engine:
t_hat = sum(relu(unet(x)) * sigmoid(conv7x7(x))) / sum(sigmoid(conv7x7(x))) per image."""
import ctypes as C

import torch
from torch import nn

from .unet import UNet
from ..._lib import DsxError, check, lib


class ForegroundMask(nn.Module):
    def __init__(self, in_channel, out_channel):
        super().__init__()
        self.layer = nn.Conv2d(in_channel, out_channel, 7, padding=3)  # parameter holder only


class TimePredictor(nn.Module):
    def __init__(self, in_channel=6, out_channel=3, inner_channel=32, norm_groups=32,
                 channel_mults=(1, 2, 4, 8, 8), attn_res=(8,), res_blocks=3, dropout=0, image_size=128):
        super().__init__()
        if out_channel != 1:
            raise DsxError("the engine's TimePredictor head supports out_channel == 1 (the Hagen configs)")
        self.unet = UNet(in_channel=in_channel, out_channel=out_channel, inner_channel=inner_channel,
                         norm_groups=norm_groups, channel_mults=channel_mults, attn_res=attn_res,
                         res_blocks=res_blocks, dropout=dropout, image_size=image_size, with_time_emb=False)
        self.foreground_mask = ForegroundMask(in_channel, out_channel)

    @torch.no_grad()
    def forward(self, x):
        if not x.is_cuda:
            raise DsxError("TimePredictor runs on the MI355X only (no CPU fallback)")
        x = x.float().contiguous()
        B, _, H, W = x.shape
        eng = self.unet.engine()
        ex = eng.executor(B, H, W)
        w = self.foreground_mask.layer.weight.detach().to("cpu", torch.float32).contiguous()
        b = self.foreground_mask.layer.bias.detach().to("cpu", torch.float32).contiguous()
        check(lib.dsx_time_predictor_set_mask(ex, C.c_void_p(w.data_ptr()), C.c_void_p(b.data_ptr())))
        out = torch.empty(B, dtype=torch.float32, device=x.device)
        check(lib.dsx_time_predictor_forward(ex, C.c_void_p(x.data_ptr()), C.c_void_p(out.data_ptr()),
                                             C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        return out
