"""CPU restatement of the diffusers-0.17.1 building blocks the DualDiff hot path instantiates —
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

PARITY UNPINNED for the arithmetic in this file: diffusers 0.17.1 (pin:
MD_txt_con_fusion/sd-controlnet-seg/config.json:3, MagicDrive fork installed from an
un-vendored third_party/ submodule, README.md:70-72) is absent from /root/reference and from
this image, and the reference holds no test or golden vector at this boundary.
What follows restates the published diffusers-0.17.1 algorithm of each block; it is anchored on
the reference's own call sites (cited per class) and on the structural contract of the SD-v1.5
state dict (parameter names and shapes, checked in tests/test_oracle_structure.py).

Modules are NCHW / torch.nn, parameter names equal the diffusers state-dict keys, constructor
signatures follow diffusers so that the reference's subclasses
(networks/unet_2d_condition_multiview.py, networks/blocks.py, networks/unet_addon_rawbox.py)
can be executed *verbatim* on top of them when minting golden vectors (tests/golden/mint.py).
"""
import math
from collections import OrderedDict
from dataclasses import dataclass
from typing import Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from .numerics import prob_round, stor


# --------------------------------------------------------------------------- config glue --
class FrozenConfig(OrderedDict):
    """`model.config`: attribute + mapping access (the reference uses both
    `unet.config.in_channels` and `cls(**unet.config)`, unet_2d_condition_multiview.py:311-315)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)


def register_to_config(init):
    """Records the constructor arguments in `self.config` (diffusers.configuration_utils)."""
    import functools
    import inspect
    sig = inspect.signature(init)

    @functools.wraps(init)
    def wrapped(self, *args, **kwargs):
        bound = sig.bind(self, *args, **kwargs)
        bound.apply_defaults()
        cfg = FrozenConfig((k, v) for k, v in bound.arguments.items() if k not in ("self", "kwargs"))
        init(self, *args, **kwargs)
        self._config = cfg      # the outermost (sub)class constructor wins
    return wrapped


class ConfigMixin:
    @property
    def config(self):
        return self._config


class ModelMixin(nn.Module):
    @property
    def dtype(self):
        return next(self.parameters()).dtype

    @property
    def device(self):
        return next(self.parameters()).device


@dataclass
class UNet2DConditionOutput:
    sample: torch.Tensor


def zero_module(module):
    """diffusers.models.controlnet.zero_module: zero every parameter, return the module."""
    for p in module.parameters():
        nn.init.zeros_(p)
    return module


# ----------------------------------------------------------------------------- embeddings --
def get_timestep_embedding(timesteps, embedding_dim, flip_sin_to_cos=False, downscale_freq_shift=1.0,
                           scale=1.0, max_period=10000):
    half = embedding_dim // 2
    exponent = -math.log(max_period) * torch.arange(half, dtype=torch.float32, device=timesteps.device)
    exponent = exponent / (half - downscale_freq_shift)
    emb = timesteps[:, None].float() * torch.exp(exponent)[None, :] * scale
    emb = torch.cat([torch.sin(emb), torch.cos(emb)], dim=-1)
    if flip_sin_to_cos:
        emb = torch.cat([emb[:, half:], emb[:, :half]], dim=-1)
    if embedding_dim % 2 == 1:
        emb = F.pad(emb, (0, 1, 0, 0))
    return emb


class Timesteps(nn.Module):
    """Instantiated at networks/unet_addon_rawbox.py:142-144 (320, True, 0)."""

    def __init__(self, num_channels, flip_sin_to_cos, downscale_freq_shift):
        super().__init__()
        self.num_channels, self.flip_sin_to_cos = num_channels, flip_sin_to_cos
        self.downscale_freq_shift = downscale_freq_shift

    def forward(self, timesteps):
        return get_timestep_embedding(timesteps, self.num_channels, self.flip_sin_to_cos,
                                      self.downscale_freq_shift)


class TimestepEmbedding(nn.Module):
    """linear_1 -> SiLU -> linear_2 (networks/unet_addon_rawbox.py:147-151)."""

    def __init__(self, in_channels, time_embed_dim, act_fn="silu", out_dim=None, post_act_fn=None,
                 cond_proj_dim=None):
        super().__init__()
        assert act_fn == "silu" and post_act_fn is None and cond_proj_dim is None
        self.linear_1 = nn.Linear(in_channels, time_embed_dim)
        self.act = nn.SiLU()
        self.linear_2 = nn.Linear(time_embed_dim, out_dim or time_embed_dim)

    def forward(self, sample, condition=None):
        return self.linear_2(self.act(self.linear_1(sample)))


# ------------------------------------------------------------------------------- resnet ---
class ResnetBlock2D(nn.Module):
    """GN -> SiLU -> conv3x3 -> (+ Linear(SiLU(temb))) -> GN -> SiLU -> conv3x3 -> + shortcut.
    eps 1e-5 from config norm_eps, output_scale_factor 1."""

    def __init__(self, *, in_channels, out_channels=None, temb_channels=512, groups=32, eps=1e-6,
                 output_scale_factor=1.0, dropout=0.0, non_linearity="swish",
                 time_embedding_norm="default", pre_norm=True):
        super().__init__()
        out_channels = out_channels or in_channels
        assert time_embedding_norm == "default"
        self.in_channels, self.out_channels = in_channels, out_channels
        self.output_scale_factor = output_scale_factor
        self.norm1 = nn.GroupNorm(groups, in_channels, eps=eps, affine=True)
        self.conv1 = nn.Conv2d(in_channels, out_channels, 3, padding=1)
        self.time_emb_proj = nn.Linear(temb_channels, out_channels) if temb_channels else None
        self.norm2 = nn.GroupNorm(groups, out_channels, eps=eps, affine=True)
        self.conv2 = nn.Conv2d(out_channels, out_channels, 3, padding=1)
        self.conv_shortcut = nn.Conv2d(in_channels, out_channels, 1) if in_channels != out_channels else None

    def forward(self, x, temb):
        h = self.conv1(F.silu(self.norm1(x)))
        if temb is not None:
            h = stor(h + self.time_emb_proj(F.silu(temb))[:, :, None, None])
        h = self.conv2(F.silu(self.norm2(h)))
        if self.conv_shortcut is not None:
            x = self.conv_shortcut(x)
        return stor(x + h) / self.output_scale_factor


class Downsample2D(nn.Module):
    def __init__(self, channels, use_conv=True, out_channels=None, padding=1, name="conv"):
        super().__init__()
        assert use_conv and padding == 1
        self.conv = nn.Conv2d(channels, out_channels or channels, 3, stride=2, padding=1)

    def forward(self, x):
        return self.conv(x)


class Upsample2D(nn.Module):
    """nearest resize (x2, or to an explicit size when the UNet forwards one) then conv3x3."""

    def __init__(self, channels, use_conv=True, out_channels=None):
        super().__init__()
        self.conv = nn.Conv2d(channels, out_channels or channels, 3, padding=1)

    def forward(self, x, output_size=None):
        if output_size is None:
            x = F.interpolate(x, scale_factor=2.0, mode="nearest")
        else:
            x = F.interpolate(x, size=tuple(output_size), mode="nearest")
        return self.conv(x)


# ------------------------------------------------------------------------------ attention --
class AttnProcessor:
    """Default processor: softmax(q k^T * scale) v, to_out (diffusers AttnProcessor)."""

    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, temb=None):
        assert attention_mask is None
        q = attn.to_q(hidden_states)
        ctx = hidden_states if encoder_hidden_states is None else encoder_hidden_states
        k, v = attn.to_k(ctx), attn.to_v(ctx)
        b, lq, c = q.shape
        h = attn.heads

        def split(t):
            return t.reshape(t.shape[0], t.shape[1], h, c // h).permute(0, 2, 1, 3)

        s = (split(q).float() @ split(k).float().transpose(-1, -2)) * attn.scale
        o = (prob_round(torch.softmax(s, dim=-1)) @ split(v).float()).to(q.dtype)
        o = o.permute(0, 2, 1, 3).reshape(b, lq, c)
        return attn.to_out[1](attn.to_out[0](o))


class Attention(nn.Module):
    """diffusers.models.attention_processor.Attention as used at networks/blocks.py:72-80 and by
    the processor protocol of networks/box_adapter.py:33-175."""

    def __init__(self, query_dim, cross_attention_dim=None, heads=8, dim_head=64, dropout=0.0,
                 bias=False, upcast_attention=False, upcast_softmax=False, processor=None, **unused):
        super().__init__()
        inner = heads * dim_head
        self.heads, self.scale = heads, dim_head ** -0.5
        self.upcast_attention = upcast_attention
        self.norm_cross = None
        self.group_norm = None
        self.spatial_norm = None
        self.residual_connection = False
        self.rescale_output_factor = 1.0
        self.to_q = nn.Linear(query_dim, inner, bias=bias)
        self.to_k = nn.Linear(cross_attention_dim or query_dim, inner, bias=bias)
        self.to_v = nn.Linear(cross_attention_dim or query_dim, inner, bias=bias)
        self.to_out = nn.ModuleList([nn.Linear(inner, query_dim), nn.Dropout(dropout)])
        self.processor = processor or AttnProcessor()

    def set_processor(self, processor):
        if isinstance(getattr(self, "processor", None), nn.Module) and not isinstance(processor, nn.Module):
            self._modules.pop("processor")
        self.processor = processor

    def prepare_attention_mask(self, attention_mask, target_length, batch_size=None, out_dim=3):
        assert attention_mask is None
        return None

    def head_to_batch_dim(self, t):
        b, l, c = t.shape
        return t.reshape(b, l, self.heads, c // self.heads).permute(0, 2, 1, 3).reshape(b * self.heads, l, c // self.heads)

    def batch_to_head_dim(self, t):
        bh, l, d = t.shape
        b = bh // self.heads
        return t.reshape(b, self.heads, l, d).permute(0, 2, 1, 3).reshape(b, l, d * self.heads)

    def forward(self, hidden_states, encoder_hidden_states=None, attention_mask=None, **kw):
        return self.processor(self, hidden_states, encoder_hidden_states=encoder_hidden_states,
                              attention_mask=attention_mask, **kw)


class GEGLU(nn.Module):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out * 2)

    def forward(self, x):
        h, gate = self.proj(x).chunk(2, dim=-1)
        return stor(h * stor(F.gelu(gate)))


class FeedForward(nn.Module):
    def __init__(self, dim, dim_out=None, mult=4, dropout=0.0, activation_fn="geglu", final_dropout=False):
        super().__init__()
        assert activation_fn == "geglu" and not final_dropout
        self.net = nn.ModuleList([GEGLU(dim, dim * mult), nn.Dropout(dropout), nn.Linear(dim * mult, dim_out or dim)])

    def forward(self, x):
        for m in self.net:
            x = m(x)
        return x


class AdaLayerNorm(nn.Module):  # name only; never instantiated on this path (blocks.py:10,68)
    def __init__(self, *a, **k):
        raise NotImplementedError


class BasicTransformerBlock(nn.Module):
    """LN->self-attn->+ ; LN->cross-attn->+ ; LN->GEGLU FF->+.  `_args` is the attribute of the
    MagicDrive diffusers fork read at networks/unet_2d_condition_multiview.py:226."""

    def __init__(self, dim, num_attention_heads, attention_head_dim, dropout=0.0, cross_attention_dim=None,
                 activation_fn="geglu", num_embeds_ada_norm=None, attention_bias=False,
                 only_cross_attention=False, double_self_attention=False, upcast_attention=False,
                 norm_elementwise_affine=True, norm_type="layer_norm", final_dropout=False):
        super().__init__()
        self._args = dict(dim=dim, num_attention_heads=num_attention_heads, attention_head_dim=attention_head_dim,
                          dropout=dropout, cross_attention_dim=cross_attention_dim, activation_fn=activation_fn,
                          num_embeds_ada_norm=num_embeds_ada_norm, attention_bias=attention_bias,
                          only_cross_attention=only_cross_attention, double_self_attention=double_self_attention,
                          upcast_attention=upcast_attention, norm_elementwise_affine=norm_elementwise_affine,
                          norm_type=norm_type, final_dropout=final_dropout)
        assert num_embeds_ada_norm is None and norm_type == "layer_norm" and not double_self_attention
        self.only_cross_attention = only_cross_attention
        self.use_ada_layer_norm = False
        self.use_ada_layer_norm_zero = False
        self.norm1 = nn.LayerNorm(dim, elementwise_affine=norm_elementwise_affine)
        self.attn1 = Attention(query_dim=dim, heads=num_attention_heads, dim_head=attention_head_dim,
                               dropout=dropout, bias=attention_bias,
                               cross_attention_dim=cross_attention_dim if only_cross_attention else None,
                               upcast_attention=upcast_attention)
        if cross_attention_dim is not None:
            self.norm2 = nn.LayerNorm(dim, elementwise_affine=norm_elementwise_affine)
            self.attn2 = Attention(query_dim=dim, cross_attention_dim=cross_attention_dim,
                                   heads=num_attention_heads, dim_head=attention_head_dim, dropout=dropout,
                                   bias=attention_bias, upcast_attention=upcast_attention)
        else:
            self.norm2, self.attn2 = None, None
        self.norm3 = nn.LayerNorm(dim, elementwise_affine=norm_elementwise_affine)
        self.ff = FeedForward(dim, dropout=dropout, activation_fn=activation_fn, final_dropout=final_dropout)

    def forward(self, hidden_states, attention_mask=None, encoder_hidden_states=None,
                encoder_attention_mask=None, timestep=None, cross_attention_kwargs=None, class_labels=None):
        kw = cross_attention_kwargs or {}
        h = hidden_states
        h = stor(self.attn1(self.norm1(h), encoder_hidden_states=encoder_hidden_states if self.only_cross_attention else None,
                            attention_mask=attention_mask, **kw) + h)
        if self.attn2 is not None:
            h = stor(self.attn2(self.norm2(h), encoder_hidden_states=encoder_hidden_states,
                                attention_mask=encoder_attention_mask, **kw) + h)
        return stor(self.ff(self.norm3(h)) + h)


class Transformer2DModel(nn.Module):
    """GN(32, C, 1e-6) -> 1x1 conv -> tokens -> blocks -> 1x1 conv -> + residual
    (use_linear_projection False, sd-controlnet-seg/config.json:86)."""

    def __init__(self, num_attention_heads=16, attention_head_dim=88, in_channels=None, num_layers=1,
                 dropout=0.0, norm_num_groups=32, cross_attention_dim=None, use_linear_projection=False,
                 only_cross_attention=False, upcast_attention=False, **unused):
        super().__init__()
        assert not use_linear_projection
        inner = num_attention_heads * attention_head_dim
        self.norm = nn.GroupNorm(norm_num_groups, in_channels, eps=1e-6, affine=True)
        self.proj_in = nn.Conv2d(in_channels, inner, 1)
        self.transformer_blocks = nn.ModuleList([
            BasicTransformerBlock(inner, num_attention_heads, attention_head_dim, dropout=dropout,
                                  cross_attention_dim=cross_attention_dim,
                                  only_cross_attention=only_cross_attention, upcast_attention=upcast_attention)
            for _ in range(num_layers)])
        self.proj_out = nn.Conv2d(inner, in_channels, 1)

    def forward(self, hidden_states, encoder_hidden_states=None, timestep=None, class_labels=None,
                cross_attention_kwargs=None, attention_mask=None, encoder_attention_mask=None, return_dict=True):
        b, _, hh, ww = hidden_states.shape
        res = hidden_states
        h = self.proj_in(self.norm(hidden_states))
        c = h.shape[1]
        h = h.permute(0, 2, 3, 1).reshape(b, hh * ww, c)
        for blk in self.transformer_blocks:
            h = blk(h, attention_mask=attention_mask, encoder_hidden_states=encoder_hidden_states,
                    encoder_attention_mask=encoder_attention_mask, timestep=timestep,
                    cross_attention_kwargs=cross_attention_kwargs, class_labels=class_labels)
        h = h.reshape(b, hh, ww, c).permute(0, 3, 1, 2).contiguous()
        out = stor(self.proj_out(h) + res)
        return (out,)


# ------------------------------------------------------------------------------- blocks ---
class CrossAttnDownBlock2D(nn.Module):
    has_cross_attention = True

    def __init__(self, in_channels, out_channels, temb_channels, num_layers=1, resnet_eps=1e-6,
                 resnet_groups=32, attn_num_head_channels=1, cross_attention_dim=1280, add_downsample=True,
                 only_cross_attention=False, upcast_attention=False, **unused):
        super().__init__()
        self.attn_num_head_channels = attn_num_head_channels
        self.gradient_checkpointing = False
        self.resnets = nn.ModuleList([
            ResnetBlock2D(in_channels=in_channels if i == 0 else out_channels, out_channels=out_channels,
                          temb_channels=temb_channels, eps=resnet_eps, groups=resnet_groups)
            for i in range(num_layers)])
        # `attn_num_head_channels` is the NUMBER of heads (attention_head_dim: 8 -> 8 heads)
        self.attentions = nn.ModuleList([
            Transformer2DModel(attn_num_head_channels, out_channels // attn_num_head_channels,
                               in_channels=out_channels, num_layers=1, cross_attention_dim=cross_attention_dim,
                               norm_num_groups=resnet_groups, only_cross_attention=only_cross_attention,
                               upcast_attention=upcast_attention)
            for _ in range(num_layers)])
        self.downsamplers = nn.ModuleList([Downsample2D(out_channels, out_channels=out_channels, name="op")]) \
            if add_downsample else None

    def forward(self, hidden_states, temb=None, encoder_hidden_states=None, attention_mask=None,
                cross_attention_kwargs=None, encoder_attention_mask=None):
        outs = ()
        h = hidden_states
        for resnet, attn in zip(self.resnets, self.attentions):
            h = resnet(h, temb)
            h = attn(h, encoder_hidden_states=encoder_hidden_states,
                     cross_attention_kwargs=cross_attention_kwargs)[0]
            outs += (h,)
        if self.downsamplers is not None:
            for d in self.downsamplers:
                h = d(h)
            outs += (h,)
        return h, outs


class DownBlock2D(nn.Module):
    def __init__(self, in_channels, out_channels, temb_channels, num_layers=1, resnet_eps=1e-6,
                 resnet_groups=32, add_downsample=True, **unused):
        super().__init__()
        self.gradient_checkpointing = False
        self.resnets = nn.ModuleList([
            ResnetBlock2D(in_channels=in_channels if i == 0 else out_channels, out_channels=out_channels,
                          temb_channels=temb_channels, eps=resnet_eps, groups=resnet_groups)
            for i in range(num_layers)])
        self.downsamplers = nn.ModuleList([Downsample2D(out_channels, out_channels=out_channels, name="op")]) \
            if add_downsample else None

    def forward(self, hidden_states, temb=None):
        outs = ()
        h = hidden_states
        for resnet in self.resnets:
            h = resnet(h, temb)
            outs += (h,)
        if self.downsamplers is not None:
            for d in self.downsamplers:
                h = d(h)
            outs += (h,)
        return h, outs


class UNetMidBlock2DCrossAttn(nn.Module):
    has_cross_attention = True

    def __init__(self, in_channels, temb_channels, num_layers=1, resnet_eps=1e-6, resnet_groups=32,
                 attn_num_head_channels=1, output_scale_factor=1.0, cross_attention_dim=1280,
                 upcast_attention=False, **unused):
        super().__init__()
        self.resnets = nn.ModuleList([
            ResnetBlock2D(in_channels=in_channels, out_channels=in_channels, temb_channels=temb_channels,
                          eps=resnet_eps, groups=resnet_groups, output_scale_factor=output_scale_factor)
            for _ in range(num_layers + 1)])
        self.attentions = nn.ModuleList([
            Transformer2DModel(attn_num_head_channels, in_channels // attn_num_head_channels,
                               in_channels=in_channels, num_layers=1, cross_attention_dim=cross_attention_dim,
                               norm_num_groups=resnet_groups, upcast_attention=upcast_attention)
            for _ in range(num_layers)])

    def forward(self, hidden_states, temb=None, encoder_hidden_states=None, attention_mask=None,
                cross_attention_kwargs=None, encoder_attention_mask=None):
        h = self.resnets[0](hidden_states, temb)
        for attn, resnet in zip(self.attentions, self.resnets[1:]):
            h = attn(h, encoder_hidden_states=encoder_hidden_states,
                     cross_attention_kwargs=cross_attention_kwargs)[0]
            h = resnet(h, temb)
        return h


def _up_resnets(in_channels, out_channels, prev_output_channel, temb_channels, num_layers, eps, groups):
    rs = []
    for i in range(num_layers):
        skip = in_channels if i == num_layers - 1 else out_channels
        rin = prev_output_channel if i == 0 else out_channels
        rs.append(ResnetBlock2D(in_channels=rin + skip, out_channels=out_channels, temb_channels=temb_channels,
                                eps=eps, groups=groups))
    return nn.ModuleList(rs)


class UpBlock2D(nn.Module):
    def __init__(self, in_channels, prev_output_channel, out_channels, temb_channels, num_layers=1,
                 resnet_eps=1e-6, resnet_groups=32, add_upsample=True, **unused):
        super().__init__()
        self.gradient_checkpointing = False
        self.resnets = _up_resnets(in_channels, out_channels, prev_output_channel, temb_channels, num_layers,
                                   resnet_eps, resnet_groups)
        self.upsamplers = nn.ModuleList([Upsample2D(out_channels, out_channels=out_channels)]) if add_upsample else None

    def forward(self, hidden_states, res_hidden_states_tuple, temb=None, upsample_size=None):
        h = hidden_states
        for resnet in self.resnets:
            skip = res_hidden_states_tuple[-1]
            res_hidden_states_tuple = res_hidden_states_tuple[:-1]
            h = resnet(torch.cat([h, skip], dim=1), temb)
        if self.upsamplers is not None:
            for u in self.upsamplers:
                h = u(h, upsample_size)
        return h


class CrossAttnUpBlock2D(nn.Module):
    has_cross_attention = True

    def __init__(self, in_channels, out_channels, prev_output_channel, temb_channels, num_layers=1,
                 resnet_eps=1e-6, resnet_groups=32, attn_num_head_channels=1, cross_attention_dim=1280,
                 add_upsample=True, only_cross_attention=False, upcast_attention=False, **unused):
        super().__init__()
        self.gradient_checkpointing = False
        self.resnets = _up_resnets(in_channels, out_channels, prev_output_channel, temb_channels, num_layers,
                                   resnet_eps, resnet_groups)
        self.attentions = nn.ModuleList([
            Transformer2DModel(attn_num_head_channels, out_channels // attn_num_head_channels,
                               in_channels=out_channels, num_layers=1, cross_attention_dim=cross_attention_dim,
                               norm_num_groups=resnet_groups, only_cross_attention=only_cross_attention,
                               upcast_attention=upcast_attention)
            for _ in range(num_layers)])
        self.upsamplers = nn.ModuleList([Upsample2D(out_channels, out_channels=out_channels)]) if add_upsample else None

    def forward(self, hidden_states, res_hidden_states_tuple, temb=None, encoder_hidden_states=None,
                cross_attention_kwargs=None, upsample_size=None, attention_mask=None, encoder_attention_mask=None):
        h = hidden_states
        for resnet, attn in zip(self.resnets, self.attentions):
            skip = res_hidden_states_tuple[-1]
            res_hidden_states_tuple = res_hidden_states_tuple[:-1]
            h = resnet(torch.cat([h, skip], dim=1), temb)
            h = attn(h, encoder_hidden_states=encoder_hidden_states,
                     cross_attention_kwargs=cross_attention_kwargs)[0]
        if self.upsamplers is not None:
            for u in self.upsamplers:
                h = u(h, upsample_size)
        return h


def get_down_block(down_block_type, num_layers, in_channels, out_channels, temb_channels, add_downsample,
                   resnet_eps, resnet_act_fn, attn_num_head_channels, resnet_groups=None,
                   cross_attention_dim=None, downsample_padding=None, dual_cross_attention=False,
                   use_linear_projection=False, only_cross_attention=False, upcast_attention=False,
                   resnet_time_scale_shift="default", **unused):
    """diffusers.models.unet_2d_blocks.get_down_block as called at
    networks/unet_addon_rawbox.py:240-257."""
    assert resnet_act_fn == "silu" and resnet_time_scale_shift == "default" and not use_linear_projection
    if down_block_type == "DownBlock2D":
        return DownBlock2D(in_channels, out_channels, temb_channels, num_layers=num_layers,
                           resnet_eps=resnet_eps, resnet_groups=resnet_groups, add_downsample=add_downsample)
    if down_block_type == "CrossAttnDownBlock2D":
        return CrossAttnDownBlock2D(in_channels, out_channels, temb_channels, num_layers=num_layers,
                                    resnet_eps=resnet_eps, resnet_groups=resnet_groups,
                                    attn_num_head_channels=attn_num_head_channels,
                                    cross_attention_dim=cross_attention_dim, add_downsample=add_downsample,
                                    only_cross_attention=only_cross_attention, upcast_attention=upcast_attention)
    raise ValueError(down_block_type)


def get_up_block(up_block_type, num_layers, in_channels, out_channels, prev_output_channel, temb_channels,
                 add_upsample, resnet_eps, resnet_groups, attn_num_head_channels, cross_attention_dim,
                 only_cross_attention=False, upcast_attention=False):
    if up_block_type == "UpBlock2D":
        return UpBlock2D(in_channels, prev_output_channel, out_channels, temb_channels, num_layers=num_layers,
                         resnet_eps=resnet_eps, resnet_groups=resnet_groups, add_upsample=add_upsample)
    if up_block_type == "CrossAttnUpBlock2D":
        return CrossAttnUpBlock2D(in_channels, out_channels, prev_output_channel, temb_channels,
                                  num_layers=num_layers, resnet_eps=resnet_eps, resnet_groups=resnet_groups,
                                  attn_num_head_channels=attn_num_head_channels,
                                  cross_attention_dim=cross_attention_dim, add_upsample=add_upsample,
                                  only_cross_attention=only_cross_attention, upcast_attention=upcast_attention)
    raise ValueError(up_block_type)


# ---------------------------------------------------------------------------------- UNet ---
class UNet2DConditionModel(ModelMixin, ConfigMixin):
    """SD-v1.5 topology; parent class of the reference's UNet2DConditionModelMultiview
    (networks/unet_2d_condition_multiview.py:44,181-216)."""

    @register_to_config
    def __init__(self, sample_size=None, in_channels=4, out_channels=4, center_input_sample=False,
                 flip_sin_to_cos=True, freq_shift=0,
                 down_block_types=("CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "DownBlock2D"),
                 mid_block_type="UNetMidBlock2DCrossAttn",
                 up_block_types=("UpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D"),
                 only_cross_attention=False, block_out_channels=(320, 640, 1280, 1280), layers_per_block=2,
                 downsample_padding=1, mid_block_scale_factor=1, act_fn="silu", norm_num_groups=32,
                 norm_eps=1e-5, cross_attention_dim=1280, encoder_hid_dim=None, encoder_hid_dim_type=None,
                 attention_head_dim=8, dual_cross_attention=False, use_linear_projection=False,
                 class_embed_type=None, addition_embed_type=None, num_class_embeds=None,
                 upcast_attention=False, resnet_time_scale_shift="default", resnet_skip_time_act=False,
                 resnet_out_scale_factor=1.0, time_embedding_type="positional", time_embedding_dim=None,
                 time_embedding_act_fn=None, timestep_post_act=None, time_cond_proj_dim=None,
                 conv_in_kernel=3, conv_out_kernel=3, projection_class_embeddings_input_dim=None,
                 class_embeddings_concat=False, mid_block_only_cross_attention=None,
                 cross_attention_norm=None, addition_embed_type_num_heads=64):
        super().__init__()
        assert class_embed_type is None and addition_embed_type is None and encoder_hid_dim is None
        assert time_embedding_type == "positional" and not use_linear_projection
        n = len(down_block_types)
        if isinstance(only_cross_attention, bool):
            only_cross_attention = [only_cross_attention] * n
        if isinstance(attention_head_dim, int):
            attention_head_dim = (attention_head_dim,) * n
        if isinstance(layers_per_block, int):
            layers_per_block = [layers_per_block] * n
        if isinstance(cross_attention_dim, int):
            cross_attention_dim = (cross_attention_dim,) * n
        self.sample_size = sample_size
        self.conv_in = nn.Conv2d(in_channels, block_out_channels[0], conv_in_kernel, padding=(conv_in_kernel - 1) // 2)
        ted = block_out_channels[0] * 4
        self.time_proj = Timesteps(block_out_channels[0], flip_sin_to_cos, freq_shift)
        self.time_embedding = TimestepEmbedding(block_out_channels[0], ted, act_fn=act_fn)
        self.class_embedding = None
        self.encoder_hid_proj = None
        self.time_embed_act = None
        self.down_blocks = nn.ModuleList()
        self.up_blocks = nn.ModuleList()
        oc = block_out_channels[0]
        for i, t in enumerate(down_block_types):
            ic, oc = oc, block_out_channels[i]
            self.down_blocks.append(get_down_block(
                t, num_layers=layers_per_block[i], in_channels=ic, out_channels=oc, temb_channels=ted,
                add_downsample=i != n - 1, resnet_eps=norm_eps, resnet_act_fn=act_fn, resnet_groups=norm_num_groups,
                cross_attention_dim=cross_attention_dim[i], attn_num_head_channels=attention_head_dim[i],
                downsample_padding=downsample_padding, only_cross_attention=only_cross_attention[i],
                upcast_attention=upcast_attention))
        self.mid_block = UNetMidBlock2DCrossAttn(
            in_channels=block_out_channels[-1], temb_channels=ted, resnet_eps=norm_eps, resnet_groups=norm_num_groups,
            output_scale_factor=mid_block_scale_factor, cross_attention_dim=cross_attention_dim[-1],
            attn_num_head_channels=attention_head_dim[-1], upcast_attention=upcast_attention)
        self.num_upsamplers = 0
        rev = list(reversed(block_out_channels))
        rheads = list(reversed(attention_head_dim))
        rlayers = list(reversed(layers_per_block))
        rxd = list(reversed(cross_attention_dim))
        roca = list(reversed(only_cross_attention))
        oc = rev[0]
        for i, t in enumerate(up_block_types):
            final = i == n - 1
            prev, oc = oc, rev[i]
            ic = rev[min(i + 1, n - 1)]
            if not final:
                self.num_upsamplers += 1
            self.up_blocks.append(get_up_block(
                t, num_layers=rlayers[i] + 1, in_channels=ic, out_channels=oc, prev_output_channel=prev,
                temb_channels=ted, add_upsample=not final, resnet_eps=norm_eps, resnet_groups=norm_num_groups,
                attn_num_head_channels=rheads[i], cross_attention_dim=rxd[i],
                only_cross_attention=roca[i], upcast_attention=upcast_attention))
        self.conv_norm_out = nn.GroupNorm(norm_num_groups, block_out_channels[0], eps=norm_eps)
        self.conv_act = nn.SiLU()
        self.conv_out = nn.Conv2d(block_out_channels[0], out_channels, conv_out_kernel, padding=(conv_out_kernel - 1) // 2)

    @property
    def attn_processors(self):
        return {name + ".processor": m.processor for name, m in self.named_modules() if hasattr(m, "set_processor")}

    def set_attn_processor(self, processor):
        for name, m in self.named_modules():
            if hasattr(m, "set_processor"):
                m.set_processor(processor.pop(name + ".processor") if isinstance(processor, dict) else processor)

    def forward(self, sample, timestep, encoder_hidden_states, class_labels=None, timestep_cond=None,
                attention_mask=None, cross_attention_kwargs=None, down_block_additional_residuals=None,
                mid_block_additional_residual=None, return_dict=True):
        """diffusers UNet2DConditionModel.forward (the body the reference re-states at
        networks/unet_2d_condition_multiview.py:327-527)."""
        forward_size = any(s % (2 ** self.num_upsamplers) != 0 for s in sample.shape[-2:])
        t = timestep
        if not torch.is_tensor(t):
            t = torch.tensor([t], dtype=torch.float64 if isinstance(t, float) else torch.int64, device=sample.device)
        elif t.dim() == 0:
            t = t[None].to(sample.device)
        t = t.expand(sample.shape[0])
        emb = self.time_embedding(self.time_proj(t).to(dtype=self.dtype), timestep_cond)
        sample = self.conv_in(sample)
        skips = (sample,)
        for blk in self.down_blocks:
            if getattr(blk, "has_cross_attention", False):
                sample, res = blk(hidden_states=sample, temb=emb, encoder_hidden_states=encoder_hidden_states,
                                  attention_mask=attention_mask, cross_attention_kwargs=cross_attention_kwargs)
            else:
                sample, res = blk(hidden_states=sample, temb=emb)
            skips += res
        if down_block_additional_residuals is not None:
            skips = tuple(s + r for s, r in zip(skips, down_block_additional_residuals))
        sample = self.mid_block(sample, emb, encoder_hidden_states=encoder_hidden_states,
                                attention_mask=attention_mask, cross_attention_kwargs=cross_attention_kwargs)
        if mid_block_additional_residual is not None:
            sample = sample + mid_block_additional_residual
        for i, blk in enumerate(self.up_blocks):
            k = len(blk.resnets)
            res, skips = skips[-k:], skips[:-k]
            up_size = skips[-1].shape[2:] if (i != len(self.up_blocks) - 1 and forward_size) else None
            if getattr(blk, "has_cross_attention", False):
                sample = blk(hidden_states=sample, temb=emb, res_hidden_states_tuple=res,
                             encoder_hidden_states=encoder_hidden_states,
                             cross_attention_kwargs=cross_attention_kwargs, upsample_size=up_size,
                             attention_mask=attention_mask)
            else:
                sample = blk(hidden_states=sample, temb=emb, res_hidden_states_tuple=res, upsample_size=up_size)
        sample = self.conv_out(self.conv_act(self.conv_norm_out(sample)))
        if not return_dict:
            return (sample,)
        return UNet2DConditionOutput(sample=sample)
