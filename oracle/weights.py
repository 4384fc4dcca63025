"""Deterministic synthetic weights keyed by reference state-dict names.

Test infrastructure only.  No checkpoints exist offline, so both
the golden generator (which loads these into the *reference* modules) and the
tests / bench (which load them into the oracle and the HIP engine) build the
same tensors from ``(key, shape)`` lists with torch's CPU generator, which is
reproducible across machines for one torch version.
"""
import zlib

import torch


def synth_tensor(key, shape, seed=0):
    g = torch.Generator().manual_seed((zlib.crc32(key.encode()) + 7919 * seed) & 0x7FFFFFFF)
    shape = tuple(shape)
    leaf = key.rsplit(".", 1)[-1]
    if leaf == "inv_freq":
        raise KeyError("inv_freq is a derived buffer, not synthesised")
    if len(shape) == 0:
        return torch.randn((), generator=g) * 0.1
    if len(shape) == 1:
        if leaf == "weight":  # GroupNorm gamma
            return 1.0 + 0.2 * torch.randn(shape, generator=g)
        return 0.1 * torch.randn(shape, generator=g)  # biases / GN beta
    fan_in = 1
    for s in shape[1:]:
        fan_in *= s
    return torch.randn(shape, generator=g) * (1.0 / fan_in) ** 0.5


def synth_state_dict(key_shapes, seed=0, inner_channel=None):
    """``key_shapes``: list of ``(key, shape)``; schedule buffers and
    ``inv_freq`` are skipped / derived (ddpm unet.py:22-26)."""
    import math
    sd = {}
    for key, shape in key_shapes:
        if key.endswith("inv_freq"):
            dim = int(shape[0]) * 2
            sd[key] = torch.exp(torch.arange(0, dim, 2, dtype=torch.float32) * (-math.log(10000) / dim))
        else:
            sd[key] = synth_tensor(key, shape, seed)
    return sd
