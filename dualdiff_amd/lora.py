"""LoRA deltas folded into the attention projections (EXTENSION, BASELINE configs[4] "LoRA-fused QKV").

The reference mentions LoRA only as stage-2 fine-tuning prose (README.md:47-49); it ships no LoRA code or
weights, so there are no reference semantics to match.  What is implemented is the standard diffusers-0.17
attention LoRA (`LoRAAttnProcessor`: `to_q(h) + scale * up(down(h))` for q, k, v and out), folded OFFLINE:

        W' = W + scale * up.weight @ down.weight                      (rank r, once, in fp32)

so that inference runs the unchanged fused Q|K|V projection kernels — no extra launches, no rank-r GEMMs.
State-dict keys follow diffusers: `<attention path>.processor.to_{q,k,v,out}_lora.{down,up}.weight`.
"""
import torch

from .networks.layers import Attention


def lora_keys(model, rank=4, cross_attention_dim=None):
    """{key: shape} of a diffusers attention-LoRA state dict for `model` (for building synthetic adapters)."""
    out = {}
    for name, mod in model.named_modules():
        if isinstance(mod, Attention):
            for proj, lin in (("to_q", mod.to_q), ("to_k", mod.to_k), ("to_v", mod.to_v), ("to_out", mod.to_out[0])):
                out["%s.processor.%s_lora.down.weight" % (name, proj)] = (rank, lin.in_features)
                out["%s.processor.%s_lora.up.weight" % (name, proj)] = (lin.out_features, rank)
    return out


def fold_lora_(model, lora_state_dict, scale=1.0):
    """Adds scale * up @ down to the matching projection weights of `model` in place (fp32 arithmetic, result
    rounded to the parameter dtype) and drops the packed-weight caches.  Returns the number of folded matrices."""
    mods = {name: mod for name, mod in model.named_modules() if isinstance(mod, Attention)}
    n = 0
    with torch.no_grad():
        for key, down in lora_state_dict.items():
            if not key.endswith("_lora.down.weight"):
                continue
            path, proj = key[: -len("_lora.down.weight")].rsplit(".processor.", 1)
            up = lora_state_dict[key.replace(".down.", ".up.")]
            attn = mods.get(path)
            if attn is None:
                raise KeyError("LoRA key %s names no attention layer of the model" % key)
            lin = attn.to_out[0] if proj == "to_out" else getattr(attn, proj)
            delta = float(scale) * (up.to(lin.weight.device, torch.float32) @ down.to(lin.weight.device, torch.float32))
            w = lin.weight.detach().float().reshape(lin.out_features, -1) + delta
            lin.weight.copy_(w.reshape(lin.weight.shape).to(lin.weight.dtype))    # bumps _version: folded products refresh
            lin._drop_cache()
            attn._drop_cache()
            n += 1
    for mod in model.modules():        # folded connector / proj_out products are keyed on parameter versions
        if hasattr(mod, "_invalidate"):
            mod._invalidate()
    return n
