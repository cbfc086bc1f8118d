"""Reference-numerics emulation for the oracle — TEST INFRASTRUCTURE ONLY.

The reference's eval path keeps every activation in fp16 between torch ops while each op
accumulates in fp32 (SURVEY.md §8c "Numerics of the reference path").  `storage_emulation` makes
the fp32 CPU oracle do the same: forward hooks round the OUTPUT of every leaf nn.Module (Conv2d,
Linear, GroupNorm, LayerNorm, SiLU, Embedding ...) to the storage dtype.  Functional ops (silu,
residual adds, softmax) are not hooked, so this is a LOWER bound of the reference path's own
rounding noise — the yardstick the HIP path's error is compared with in tests/test_model_gpu.py.
"""
import contextlib

import torch


@contextlib.contextmanager
def storage_emulation(module, dtype):
    handles = []

    def hook(_m, _inp, out):
        if torch.is_tensor(out) and out.is_floating_point():
            return out.to(dtype).to(out.dtype)
        return out

    for m in module.modules():
        if len(list(m.children())) == 0:
            handles.append(m.register_forward_hook(hook))
    try:
        yield module
    finally:
        for h in handles:
            h.remove()
