"""Model-level plumbing shared by the UNet and the ControlNet: the slice of diffusers'
`ModelMixin` / `ConfigMixin` surface that the reference's runner, pipeline and build_pipe touch
(SURVEY.md §8b "B2 — model surface"): `.config`, `.dtype`, `.device`, `from_pretrained` /
`save_pretrained` on diffusers-layout folders, attention-processor registry, and accepted-but-
no-op memory knobs (`enable_xformers_memory_efficient_attention`, gradient checkpointing).
"""
import json
import os

import torch
import torch.nn as nn

from .layers import HIPAttnProcessor, TimeEmbProjBank


class Config(dict):
    """`model.config`: attribute and mapping access (`unet.config.in_channels`, `**unet.config`)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)


class ModelBase(nn.Module):
    config_name = "config.json"
    weights_name = "diffusion_pytorch_model.bin"
    safetensors_name = "diffusion_pytorch_model.safetensors"
    _supports_gradient_checkpointing = True

    def _register_config(self, **kw):
        cfg = Config(kw)
        cfg["_class_name"] = type(self).__name__
        cfg["_diffusers_version"] = "0.17.1"
        self.__dict__["_config"] = cfg
        self.__dict__["_temb_bank"] = None

    @property
    def config(self):
        return self.__dict__["_config"]

    @property
    def dtype(self):
        return next(self.parameters()).dtype

    @property
    def device(self):
        return next(self.parameters()).device

    # -- caches ------------------------------------------------------------------------------
    @property
    def temb_bank(self):
        if self.__dict__.get("_temb_bank") is None:
            self.__dict__["_temb_bank"] = TimeEmbProjBank(self)
        return self.__dict__["_temb_bank"]

    def _invalidate(self):
        if self.__dict__.get("_temb_bank") is not None:
            self.__dict__["_temb_bank"].invalidate()

    def _apply(self, fn, *a, **k):
        self._invalidate()
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self._invalidate()
        return super().load_state_dict(*a, **k)

    # -- attention-processor registry (unet_addon_rawbox.py:523-593) --------------------------
    @property
    def attn_processors(self):
        return {name + ".processor": m.processor for name, m in self.named_modules()
                if hasattr(m, "set_processor")}

    def set_attn_processor(self, processor):
        count = len(self.attn_processors)
        if isinstance(processor, dict) and len(processor) != count:
            raise ValueError(
                f"A dict of processors was passed, but the number of processors {len(processor)} does not "
                f"match the number of attention layers: {count}.")
        for name, m in [(n, mm) for n, mm in self.named_modules() if hasattr(mm, "set_processor")]:
            m.set_processor(processor.pop(name + ".processor") if isinstance(processor, dict) else processor)

    def set_default_attn_processor(self):
        self.set_attn_processor(HIPAttnProcessor())

    # -- knobs the reference flips; nothing to do on this implementation ------------------------
    def enable_xformers_memory_efficient_attention(self, attention_op=None):
        return None          # attention is always the fused HIP kernel

    def disable_xformers_memory_efficient_attention(self):
        return None

    def enable_gradient_checkpointing(self, flag=None):
        return None          # inference-only path

    def set_attention_slice(self, slice_size):
        return None

    # -- diffusers-layout checkpoints -----------------------------------------------------------
    def save_pretrained(self, save_directory, safe_serialization=False, **unused):
        os.makedirs(save_directory, exist_ok=True)
        cfg = {k: (list(v) if isinstance(v, tuple) else v) for k, v in self.config.items()}
        with open(os.path.join(save_directory, self.config_name), "w") as f:
            json.dump(cfg, f, indent=2, sort_keys=True)
        sd = {k: v.detach().cpu().contiguous() for k, v in self.state_dict().items()}
        if safe_serialization:
            from safetensors.torch import save_file
            save_file(sd, os.path.join(save_directory, self.safetensors_name))
        else:
            torch.save(sd, os.path.join(save_directory, self.weights_name))

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, torch_dtype=None, subfolder=None,
                        ignore_mismatched_sizes=False, **unused):
        """Loads `config.json` + `diffusion_pytorch_model.{safetensors,bin}` (diffusers key names).
        Like diffusers, returns the model in eval mode; extra kwargs (low_cpu_mem_usage,
        device_map, ...) are accepted and ignored (misc/test_utils.py:111-113)."""
        path = pretrained_model_name_or_path
        if subfolder:
            path = os.path.join(path, subfolder)
        with open(os.path.join(path, cls.config_name)) as f:
            cfg = {k: v for k, v in json.load(f).items() if not k.startswith("_")}
        model = cls(**cfg)
        st = os.path.join(path, cls.safetensors_name)
        if os.path.exists(st):
            from safetensors.torch import load_file
            sd = load_file(st)
        else:
            sd = torch.load(os.path.join(path, cls.weights_name), map_location="cpu")
        own = model.state_dict()
        if ignore_mismatched_sizes:
            sd = {k: v for k, v in sd.items() if k not in own or own[k].shape == v.shape}
        missing, unexpected = model.load_state_dict(sd, strict=False)
        allowed = tuple(getattr(cls, "_keys_to_ignore_on_load_missing", ()))
        hard_missing = [k for k in missing if not k.startswith(allowed)] if allowed else list(missing)
        if hard_missing and not ignore_mismatched_sizes:
            raise RuntimeError("checkpoint %s is missing keys: %s" % (path, hard_missing[:8]))
        if torch_dtype is not None:
            model = model.to(torch_dtype)
        return model.eval()
