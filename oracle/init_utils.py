"""Deterministic, name-keyed weight initialisation shared by tests, the golden-vector minting
script and bench.py's cpu_baseline — TEST INFRASTRUCTURE ONLY.

Zero-initialised modules of the reference (connector blocks.py:83, zero convs
unet_addon_rawbox.py:230-281, embedder conv_out map_embedder.py:110-112) get NON-zero seeded
values here, otherwise their paths would be untested."""
import zlib

import torch


def seeded_state_dict(module, seed=0):
    """name -> tensor, each drawn from its own generator keyed by crc32(name) ^ seed."""
    out = {}
    for name, t in module.state_dict().items():
        g = torch.Generator().manual_seed((zlib.crc32(name.encode()) ^ (seed * 2654435761)) & 0x7FFFFFFF)
        if not t.is_floating_point():
            out[name] = t.clone()
            continue
        if t.dim() >= 2:
            fan_in = t[0].numel()
            v = torch.randn(t.shape, generator=g) * (fan_in ** -0.5)
        elif name.endswith("weight"):          # norm scales
            v = 1.0 + 0.1 * torch.randn(t.shape, generator=g)
        else:                                  # biases, null features
            v = 0.05 * torch.randn(t.shape, generator=g)
        out[name] = v.to(t.dtype)
    return out


def seeded_init_(module, seed=0):
    module.load_state_dict(seeded_state_dict(module, seed), strict=True)
    return module


def seeded_tensor(shape, seed, scale=1.0, dtype=torch.float32):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dtype)
