"""``nn.Module`` face of one engine UNet.

It holds the weights as ordinary parameters under the reference's state-dict
key names (so ``*_gen.pth`` files load unchanged, model/model.py:153-166) and
forwards through ``libdsx.so``; nothing here computes on the CPU.
"""
import math

import torch
from torch import nn

from .. import engine
from .._lib import DsxError


def _mangle(name):
    return name.replace(".", "__")


class EngineUNet(nn.Module):
    """Keyword arguments are those of the reference ``UNet.__init__``
    (sr3 unet.py:161-174 / ddpm unet.py:150-162)."""

    flavour = "sr3"

    def __init__(self, in_channel=6, out_channel=3, inner_channel=32, norm_groups=32,
                 channel_mults=(1, 2, 4, 8, 8), attn_res=(8,), res_blocks=3, dropout=0,
                 with_time_emb=True, image_size=128):
        super().__init__()
        self.dropout = dropout  # identity at inference (Q6): the engine has eval semantics always
        self.cfg = engine.make_cfg(self.flavour, in_channel, out_channel, inner_channel, norm_groups,
                                   channel_mults, attn_res, res_blocks, image_size, with_time_emb)
        self._eng = engine.UNetEngine(self.cfg, self.flavour)
        self.compute_dtype = "f32"
        self._synced = None
        # packed-weight cache (set by DDPM.load_network): file prefix and the checkpoint's hash
        self.pack_cache = None          # (path_prefix, key[, tensor fingerprint]) or None; consumed by the next sync
        self.pack_cache_hit = None      # True / False after the last engine() sync that consulted the cache
        g = torch.Generator().manual_seed(0)
        self._ref_names = list(self._eng.param_names)
        for name, shape in zip(self._eng.param_names, self._eng.param_shapes):
            if name.endswith("inv_freq"):  # ddpm unet.py:22-26 (a buffer in the reference too)
                dim = shape[0] * 2
                buf = torch.exp(torch.arange(0, dim, 2, dtype=torch.float32) * (-math.log(10000) / dim))
                self.register_buffer(_mangle(name), buf)
            elif len(shape) == 1:
                init = torch.ones(shape) if name.endswith("weight") else torch.zeros(shape)
                self.register_parameter(_mangle(name), nn.Parameter(init))
            else:
                fan_in = 1
                for s in shape[1:]:
                    fan_in *= s
                w = (torch.rand(shape, generator=g) * 2 - 1) / math.sqrt(fan_in)
                self.register_parameter(_mangle(name), nn.Parameter(w))

    # ---- state dict under the reference's key names --------------------------------------
    def _tensors(self):
        return {n: getattr(self, _mangle(n)) for n in self._ref_names}

    def _save_to_state_dict(self, destination, prefix, keep_vars):
        for n, t in self._tensors().items():
            destination[prefix + n] = t if keep_vars else t.detach()

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys,
                              unexpected_keys, error_msgs):
        mine = self._tensors()
        for n, t in mine.items():
            key = prefix + n
            if key not in state_dict:
                if strict and not n.endswith("inv_freq"):
                    missing_keys.append(key)
                continue
            src = state_dict[key]
            if tuple(src.shape) != tuple(t.shape):
                error_msgs.append(f"size mismatch for {key}: {tuple(src.shape)} vs {tuple(t.shape)}")
                continue
            with torch.no_grad():
                t.copy_(src)
        self.pack_cache = None      # new weights: a cache, if any, is re-attached by DDPM.load_network
        if strict:
            for key in state_dict:
                if key.startswith(prefix) and key[len(prefix):] not in mine:
                    unexpected_keys.append(key)
        self._synced = None

    # ---- engine sync -----------------------------------------------------------------------
    def attach_pack_cache(self, prefix, key):
        """DDPM.load_network: a packed image of exactly the weights just loaded may be found at / written to
        ``<prefix>.<dtype>.dsxpack``."""
        self.pack_cache = (prefix, key, self._fingerprint()[1:])

    def _fingerprint(self):
        return (self.compute_dtype,) + tuple((t.data_ptr(), t._version) for t in self._tensors().values())

    def engine(self):
        """The finalized ``UNetEngine`` with the module's current weights."""
        fp = self._fingerprint()
        if self._synced != fp:
            hit = False
            # the cache describes the checkpoint as load_network read it: honoured once, and only while the module's
            # tensors are still the ones it was attached to (an in-place edit afterwards repacks from the tensors)
            cache, self.pack_cache = self.pack_cache, None
            if cache is not None and len(cache) == 3 and cache[2] != fp[1:]:
                cache = None
            if cache is not None:
                prefix, key = cache[0], cache[1]
                path = f"{prefix}.{self.compute_dtype}.dsxpack"
                hit = self._eng.finalize_from_packed(path, self.compute_dtype, key)
                if not hit:
                    self._eng.load_state_dict(self._tensors())
                    self._eng._finalized_dtype = None
                    self._eng.finalize(self.compute_dtype)
                    try:
                        self._eng.save_packed(path, key)
                    except OSError:
                        pass                                   # read-only checkpoint directory: no cache
                self.pack_cache_hit = hit
            else:
                self._eng.load_state_dict(self._tensors())
                self._eng.finalize(self.compute_dtype)
            self._synced = fp
        return self._eng

    def forward(self, x, time=None):
        if not x.is_cuda:
            raise DsxError("the UNet runs on the MI355X only; move the input to the GPU (no CPU fallback)")
        with torch.no_grad():
            return self.engine().forward(x.float(), None if time is None else time.float())
