"""Oracle: functional UNet forward over a reference-keyed state dict.

Follows ``model/sr3_modules/unet.py`` (flavour ``"sr3"``) and
``model/ddpm_modules/unet.py`` (flavour ``"ddpm"``).  Test infrastructure only
(see ``oracle/__init__.py``).
"""
import math

import torch
import torch.nn.functional as F


def _swish(x):
    # sr3 unet.py:53-55 / ddpm unet.py:37-39
    return x * torch.sigmoid(x)


def unet_topology(cfg):
    """Enumerate the module list exactly as ``UNet.__init__`` builds it.

    sr3 unet.py:161-233 / ddpm unet.py:150-218.  Returns a list of dicts in
    execution order with keys ``kind`` ∈ {conv_in, res, down, up, final},
    ``prefix`` (state-dict prefix), ``cin``/``cout``, ``attn`` and
    ``section`` ∈ {downs, mid, ups}.
    """
    inner = cfg["inner_channel"]
    mults = list(cfg["channel_mults"])
    attn_res = cfg.get("attn_res") or []
    if isinstance(attn_res, int):
        attn_res = [attn_res]
    res_blocks = cfg["res_blocks"]
    now_res = cfg["image_size"]
    mods = []
    pre = inner
    feat_channels = [pre]
    idx = 0
    mods.append(dict(kind="conv_in", section="downs", prefix=f"downs.{idx}",
                     cin=cfg["in_channel"], cout=inner))
    idx += 1
    n = len(mults)
    for ind in range(n):
        is_last = ind == n - 1
        use_attn = now_res in attn_res
        ch = inner * mults[ind]
        for _ in range(res_blocks):
            mods.append(dict(kind="res", section="downs", prefix=f"downs.{idx}",
                             cin=pre, cout=ch, attn=use_attn))
            idx += 1
            feat_channels.append(ch)
            pre = ch
        if not is_last:
            mods.append(dict(kind="down", section="downs", prefix=f"downs.{idx}",
                             cin=pre, cout=pre))
            idx += 1
            feat_channels.append(pre)
            now_res //= 2
    mods.append(dict(kind="res", section="mid", prefix="mid.0", cin=pre, cout=pre, attn=True))
    mods.append(dict(kind="res", section="mid", prefix="mid.1", cin=pre, cout=pre, attn=False))
    idx = 0
    for ind in reversed(range(n)):
        is_last = ind < 1
        use_attn = now_res in attn_res
        ch = inner * mults[ind]
        for _ in range(res_blocks + 1):
            skip = feat_channels.pop()
            mods.append(dict(kind="res", section="ups", prefix=f"ups.{idx}",
                             cin=pre + skip, cout=ch, attn=use_attn, skip=skip))
            idx += 1
            pre = ch
        if not is_last:
            mods.append(dict(kind="up", section="ups", prefix=f"ups.{idx}", cin=pre, cout=pre))
            idx += 1
            now_res *= 2
    out_ch = cfg["out_channel"] if cfg.get("out_channel") is not None else cfg["in_channel"]
    mods.append(dict(kind="final", section="final", prefix="final_conv", cin=pre, cout=out_ch))
    return mods


def time_embedding(sd, cfg, flavour, time, p=""):
    """sr3 unet.py:18-31,177-187 / ddpm unet.py:19-34,163-173."""
    inner = cfg["inner_channel"]
    if flavour == "sr3":
        if (p + "noise_level_mlp.1.weight") not in sd:
            return None
        # PositionalEncoding: noise_level (B,1) -> (B,1,inner)
        count = inner // 2
        step = torch.arange(count, dtype=time.dtype) / count
        enc = time.unsqueeze(1) * torch.exp(-math.log(1e4) * step.unsqueeze(0))
        enc = torch.cat([torch.sin(enc), torch.cos(enc)], dim=-1)
        h = F.linear(enc, sd[p + "noise_level_mlp.1.weight"], sd[p + "noise_level_mlp.1.bias"])
        h = _swish(h)
        return F.linear(h, sd[p + "noise_level_mlp.3.weight"], sd[p + "noise_level_mlp.3.bias"])
    else:
        if (p + "time_mlp.1.weight") not in sd:
            return None
        inv_freq = sd[p + "time_mlp.0.inv_freq"]
        shape = time.shape
        sinus = torch.ger(time.view(-1).float(), inv_freq)
        pos = torch.cat([sinus.sin(), sinus.cos()], dim=-1).view(*shape, inner)
        h = F.linear(pos, sd[p + "time_mlp.1.weight"], sd[p + "time_mlp.1.bias"])
        h = _swish(h)
        return F.linear(h, sd[p + "time_mlp.3.weight"], sd[p + "time_mlp.3.bias"])


def _block(sd, pfx, x, groups):
    # Block: GN -> Swish -> (Dropout: identity in eval) -> Conv3x3  (unet.py:80-91)
    h = F.group_norm(x, groups, sd[pfx + ".block.0.weight"], sd[pfx + ".block.0.bias"], eps=1e-5)
    h = _swish(h)
    return F.conv2d(h, sd[pfx + ".block.3.weight"], sd[pfx + ".block.3.bias"], padding=1)


def resnet_block(sd, pfx, x, t_emb, groups, flavour):
    """sr3 unet.py:94-110 / ddpm unet.py:78-96."""
    h = _block(sd, pfx + ".block1", x, groups)
    if flavour == "sr3":
        if t_emb is not None:
            # FeatureWiseAffine, additive (use_affine_level=False) unet.py:34-50
            b = x.shape[0]
            h = h + F.linear(t_emb, sd[pfx + ".noise_func.noise_func.0.weight"],
                             sd[pfx + ".noise_func.noise_func.0.bias"]).view(b, -1, 1, 1)
    else:
        if t_emb is not None and (pfx + ".mlp.1.weight") in sd:
            h = h + F.linear(_swish(t_emb), sd[pfx + ".mlp.1.weight"],
                             sd[pfx + ".mlp.1.bias"])[:, :, None, None]
    h = _block(sd, pfx + ".block2", h, groups)
    if (pfx + ".res_conv.weight") in sd:
        return h + F.conv2d(x, sd[pfx + ".res_conv.weight"], sd[pfx + ".res_conv.bias"])
    return h + x


def self_attention(sd, pfx, x, groups):
    """unet.py:113-142 (n_head=1): GN -> qkv 1x1 -> softmax(QK^T/sqrt(C)) V -> out 1x1 + x."""
    b, c, hh, ww = x.shape
    n = F.group_norm(x, groups, sd[pfx + ".norm.weight"], sd[pfx + ".norm.bias"], eps=1e-5)
    qkv = F.conv2d(n, sd[pfx + ".qkv.weight"])
    q, k, v = qkv.view(b, 1, 3 * c, hh, ww).chunk(3, dim=2)
    attn = torch.einsum("bnchw, bncyx -> bnhwyx", q, k).contiguous() / math.sqrt(c)
    attn = torch.softmax(attn.view(b, 1, hh, ww, -1), -1).view(b, 1, hh, ww, hh, ww)
    out = torch.einsum("bnhwyx, bncyx -> bnchw", attn, v).contiguous()
    out = F.conv2d(out.view(b, c, hh, ww), sd[pfx + ".out.weight"], sd[pfx + ".out.bias"])
    return out + x


def unet_forward(sd, cfg, flavour, x, time, prefix=""):
    """sr3 unet.py:235-259 / ddpm unet.py:220-243.

    ``sd`` holds the UNet's own keys under ``prefix`` (e.g. ``"denoise_fn."``).
    ``x`` NCHW fp32; ``time``: sr3 γ ``(B,1)``; ddpm ``(B,)`` or ``(1,)``; None
    when the UNet has no time embedding (TimePredictor).
    """
    p = prefix
    groups = cfg["norm_groups"]
    t = time_embedding(sd, cfg, flavour, time, p) if time is not None else None
    if flavour == "sr3" and t is not None:
        t = t  # (B,1,inner): FeatureWiseAffine views to (B,-1,1,1)
    feats = []
    for m in unet_topology(cfg):
        pf = p + m["prefix"]
        if m["kind"] == "conv_in":
            x = F.conv2d(x, sd[pf + ".weight"], sd[pf + ".bias"], padding=1)
            feats.append(x)
        elif m["kind"] == "down":
            x = F.conv2d(x, sd[pf + ".conv.weight"], sd[pf + ".conv.bias"], stride=2, padding=1)
            feats.append(x)
        elif m["kind"] == "up":
            x = F.interpolate(x, scale_factor=2, mode="nearest")
            x = F.conv2d(x, sd[pf + ".conv.weight"], sd[pf + ".conv.bias"], padding=1)
        elif m["kind"] == "res":
            if m["section"] == "ups":
                x = torch.cat((x, feats.pop()), dim=1)
            x = resnet_block(sd, pf + ".res_block", x, t, groups, flavour)
            if m["attn"]:
                x = self_attention(sd, pf + ".attn", x, groups)
            if m["section"] == "downs":
                feats.append(x)
        elif m["kind"] == "final":
            x = _block(sd, pf, x, groups)
    return x


def time_predictor_forward(sd, cfg, x, prefix=""):
    """ddpm_modules/time_predictor.py:5-44."""
    out = unet_forward(sd, cfg, "ddpm", x, None, prefix + "unet.")
    out = F.relu(out)
    att = torch.sigmoid(F.conv2d(x, sd[prefix + "foreground_mask.layer.weight"],
                                 sd[prefix + "foreground_mask.layer.bias"], padding=3))
    out = (out * att).reshape(out.shape[0], -1)
    return out.sum(dim=1) / att.reshape(out.shape).sum(dim=1)
