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


def fold_lora_(model, lora_state_dict, scale=1.0, network_alphas=None):
    """Adds scale * (alpha / rank) * up @ down to the matching projection weights of `model` in place (fp32
    arithmetic, result rounded to the parameter dtype) and drops the packed-weight caches.  `alpha` comes from an
    optional `<...>_lora.alpha` (or `.network_alpha`) entry next to the down / up pair — diffusers / PEFT
    `network_alpha` — or from the separate `network_alphas` mapping ({<stem> or <down key>: alpha}, the form diffusers'
    loaders hand over) and defaults to the rank (factor 1).  Malformed keys, a missing up-weight, mismatched ranks or
    shapes raise with the offending key; entries that are not LoRA matrices are rejected rather than ignored.
    Returns the number of folded matrices."""
    mods = {name: mod for name, mod in model.named_modules() if isinstance(mod, Attention)}
    suffix_down, known = "_lora.down.weight", ("_lora.down.weight", "_lora.up.weight", "_lora.alpha", "_lora.network_alpha")
    stray = [k for k in lora_state_dict if not k.endswith(known)]
    if stray:
        raise KeyError("not attention-LoRA entries (expected <attn>.processor.to_{q,k,v,out}_lora.{down,up}.weight): %s"
                       % stray[:4])
    ups = [k for k in lora_state_dict if k.endswith("_lora.up.weight")
           and k[: -len("_lora.up.weight")] + suffix_down not in lora_state_dict]
    if ups:
        raise KeyError("LoRA up-weights without a down-weight: %s" % ups[:4])
    # `network_alphas` as diffusers' loaders hand it over: keys '<prefix>.<attention path>.processor.<proj>_lora.down.weight
    # .alpha', '<...>.<proj>_lora.alpha' or the bare stem, with an optional 'unet.' prefix — normalised to the stem;
    # entries that match no folded pair are an error (a silently ignored alpha is a silently wrong LoRA scale)
    # The loader hands over the WHOLE mapping, other networks included: entries under a foreign prefix ('text_encoder.',
    # 'text_encoder_2.', ...) are not this model's and are skipped; the strictness applies to 'unet.' / bare keys.
    alphas = {}
    for k, v in (network_alphas or {}).items():
        if k.startswith(("text_encoder", "te.", "lora_te", "vae.", "controlnet.")):
            continue
        kk = k[len("unet."):] if k.startswith("unet.") else k
        for suf in (".alpha", ".network_alpha"):
            if kk.endswith(suf):
                kk = kk[: -len(suf)]
        for suf in (".lora.down.weight", ".lora.up.weight", "_lora.down.weight", "_lora.up.weight", "_lora", ".lora"):
            if kk.endswith(suf):
                kk = kk[: -len(suf)]
        if kk.endswith(".to_out.0"):                  # 'attn1.to_out.0[.lora ...]': the Linear inside the ModuleList
            kk = kk[: -len(".0")]
        if ".processor." not in kk and kk.rsplit(".", 1)[-1] in ("to_q", "to_k", "to_v", "to_out"):
            path, proj = kk.rsplit(".", 1)
            kk = path + ".processor." + proj
        alphas[kk] = v
    used = set()
    plan = []                                # validate EVERYTHING first: a bad entry must not leave a half-folded model
    for key, down in lora_state_dict.items():
        if not key.endswith(suffix_down):
            continue
        stem = key[: -len(suffix_down)]
        if ".processor." not in stem:
            raise KeyError("LoRA key %s: expected '<attention path>.processor.<proj>_lora.down.weight'" % key)
        path, proj = stem.rsplit(".processor.", 1)
        if proj not in ("to_q", "to_k", "to_v", "to_out"):
            raise KeyError("LoRA key %s: unknown projection '%s'" % (key, proj))
        up_key = stem + "_lora.up.weight"
        if up_key not in lora_state_dict:
            raise KeyError("LoRA key %s has no matching %s" % (key, up_key))
        up = lora_state_dict[up_key]
        attn = mods.get(path)
        if attn is None:
            raise KeyError("LoRA key %s names no attention layer of the model" % key)
        lin = attn.to_out[0] if proj == "to_out" else getattr(attn, proj)
        if down.dim() != 2 or up.dim() != 2 or up.shape[1] != down.shape[0] or down.shape[1] != lin.in_features \
                or up.shape[0] != lin.out_features:
            raise ValueError("LoRA %s: down %s / up %s do not fit a %d -> %d projection"
                             % (stem, tuple(down.shape), tuple(up.shape), lin.in_features, lin.out_features))
        rank = down.shape[0]
        # alpha key convention: `<stem>_lora.alpha` or `<stem>_lora.network_alpha` beside the down / up pair (a 0-d or
        # 1-element tensor, or a number); a separate mapping goes through `network_alphas`
        alpha = lora_state_dict.get(stem + "_lora.alpha", lora_state_dict.get(stem + "_lora.network_alpha"))
        if alpha is None and stem in alphas:
            alpha = alphas[stem]
            used.add(stem)
        plan.append((attn, lin, down, up, float(scale) * (float(alpha) / rank if alpha is not None else 1.0)))
    unmatched = sorted(set(alphas) - used)
    if unmatched:
        raise KeyError("network_alphas entries that match no LoRA pair of the state dict: %s" % unmatched[:4])
    n = 0
    with torch.no_grad():
        for attn, lin, down, up, factor in plan:
            delta = factor * (up.to(lin.weight.device, torch.float32) @ down.to(lin.weight.device, torch.float32))
            w = lin.weight.detach().float().reshape(lin.out_features, -1) + delta
            lin.weight.copy_(w.reshape(lin.weight.shape).to(lin.weight.dtype))    # bumps _version: folded products refresh
            lin._drop_cache()
            attn._drop_cache()
            n += 1
    for mod in model.modules():        # folded connector / proj_out products are keyed on parameter versions
        if hasattr(mod, "_invalidate"):
            mod._invalidate()
    return n
