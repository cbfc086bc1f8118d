"""Reference-numerics emulation for the oracle — TEST INFRASTRUCTURE ONLY.

The reference's eval path keeps every activation in fp16 between torch ops while each op
accumulates in fp32 (SURVEY.md §8c "Numerics of the reference path").  `storage_emulation` makes
the fp32 CPU oracle do the same: hooks round every floating-point tensor that ENTERS or LEAVES any
nn.Module (leaf or container) to the storage dtype — conv / linear / norm outputs, and also the
results of functional ops (SiLU, residual adds, `sample += controlnet_cond`, concatenations) at the
point where they are handed to the next module.  Tensors that live only inside a functional
expression (softmax probabilities, q.k^T scores) stay fp32, so this is still a LOWER bound of the
reference path's own rounding noise — the yardstick the HIP path's error is compared with in
tests/test_model_gpu.py.
"""
import contextlib

import torch


def _round_tree(x, dtype):
    if torch.is_tensor(x):
        return x.to(dtype).to(x.dtype) if x.is_floating_point() else x
    if isinstance(x, tuple):
        return tuple(_round_tree(v, dtype) for v in x)
    if isinstance(x, list):
        return [_round_tree(v, dtype) for v in x]
    if isinstance(x, dict):
        return {k: _round_tree(v, dtype) for k, v in x.items()}
    return x


@contextlib.contextmanager
def storage_emulation(module, dtype):
    handles = []

    def pre(_m, args, kwargs):
        return _round_tree(args, dtype), _round_tree(kwargs, dtype)

    def post(_m, _args, out):
        if hasattr(out, "sample") and torch.is_tensor(out.sample):
            out.sample = _round_tree(out.sample, dtype)
            return out
        return _round_tree(out, dtype)

    for m in module.modules():
        handles.append(m.register_forward_pre_hook(pre, with_kwargs=True))
        handles.append(m.register_forward_hook(post))
    try:
        yield module
    finally:
        for h in handles:
            h.remove()
