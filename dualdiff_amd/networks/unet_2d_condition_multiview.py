"""`UNet2DConditionModelMultiview` on hand-written gfx950 kernels.

Drop-in for /root/reference/MD_txt_con_fusion/magicdrive/networks/unet_2d_condition_multiview.py
(class at :44, forward at :327-527): same constructor/config keys, same `forward()` keyword
names and return type, diffusers-layout state dict (incl. the `norm4 / attn4 / connector`
additions, blocks.py:67-90), loadable by dotted path
(`model.unet_module: dualdiff_amd.networks.unet_2d_condition_multiview.UNet2DConditionModelMultiview`).

Internally the network runs NHWC end to end: fused GroupNorm+SiLU, implicit-GEMM 3x3 convs with
bias / time-embedding / residual epilogues, fused-QKV flash attention, GEGLU in the GEMM epilogue,
the up-path channel concat and the nearest upsample folded into their consumers.
"""
from dataclasses import dataclass
from typing import Any, Dict, Optional, Tuple, Union

import torch
import torch.nn as nn

from .. import ops as O
from .blocks import BasicMultiviewTransformerBlock
from .layers import (BasicTransformerBlock, prefetch_cross_kv, drop_prefetched_kv, CrossKVBank, Conv3x3, CrossAttnDownBlock2D, CrossAttnUpBlock2D, DownBlock2D, GroupNorm,
                     TimestepEmbedding, Timesteps, UNetMidBlock2DCrossAttn, UpBlock2D, as_nchw_view,
                     run_down_block, run_up_block, to_nhwc, CTX_BASE, context_keys, ctx_capacity, lk_const)
from .model_base import ModelBase, sibling_overlap


def _at_capacity(ctx, cap):
    """(m, lc, C) context -> (m, cap, C) with the same first lc tokens per instance.  A view of a capacity layout (the
    tokens our ControlNet hands back: `buffer[:, :lc]`) is widened IN PLACE; anything else is copied into a zeroed
    buffer.  The tokens past lc are never read by an attention (layers.context_keys); the K / V projection GEMM does
    multiply them, so they must be memory this process owns."""
    m, lc, c = ctx.shape
    if lc == cap:
        return ctx
    if ctx.stride(2) == 1 and ctx.stride(1) == c and ctx.stride(0) == cap * c and \
            (ctx.storage_offset() + m * cap * c) * ctx.element_size() <= ctx.untyped_storage().nbytes():
        return ctx.as_strided((m, cap, c), ctx.stride())
    buf = ctx.new_zeros((m, cap, c))
    buf[:, :lc] = ctx
    return buf


@dataclass
class UNet2DConditionOutput:
    sample: torch.Tensor


_DEFAULT_PAIR = {0: [5, 1], 1: [0, 2], 2: [1, 3], 3: [2, 4], 4: [3, 5], 5: [4, 0]}


class UNet2DConditionModelMultiview(ModelBase):
    # True: `BasicMultiviewTransformerBlock` (attn4 + connector) in every transformer; the plain SD-v1.5
    # subclass below turns it off
    multiview = True
    video = False      # True: VideoMultiviewTransformerBlock (UNet2DConditionModelMultiviewVideo below)
    # Side-stream projection of the cross-attention K/V (layers.prefetch_cross_kv).  Measured on MI355X
    # (config 2, graph replay): 66.0 steps/s off vs 62.4 on — the extra stream's small GEMMs delay the
    # main chain more than they shorten it — so it is off; DD_PREFETCH_KV=1 turns it on.
    prefetch_kv = __import__('os').environ.get('DD_PREFETCH_KV', '0') == '1'
    # all attn2 K/V projections of the model as one GEMM per forward (layers.CrossKVBank); DD_KV_BANK=0 disables
    kv_bank = __import__('os').environ.get('DD_KV_BANK', '1') != '0'

    _WARN_ONCE = 0

    def __init__(
        self,
        sample_size: Optional[int] = None,
        in_channels: int = 4,
        out_channels: int = 4,
        center_input_sample: bool = False,
        flip_sin_to_cos: bool = True,
        freq_shift: int = 0,
        down_block_types: Tuple[str] = ("CrossAttnDownBlock2D", "CrossAttnDownBlock2D",
                                        "CrossAttnDownBlock2D", "DownBlock2D"),
        mid_block_type: Optional[str] = "UNetMidBlock2DCrossAttn",
        up_block_types: Tuple[str] = ("UpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D",
                                      "CrossAttnUpBlock2D"),
        only_cross_attention: Union[bool, Tuple[bool]] = False,
        block_out_channels: Tuple[int] = (320, 640, 1280, 1280),
        layers_per_block: Union[int, Tuple[int]] = 2,
        downsample_padding: int = 1,
        mid_block_scale_factor: float = 1,
        act_fn: str = "silu",
        norm_num_groups: Optional[int] = 32,
        norm_eps: float = 1e-5,
        cross_attention_dim: Union[int, Tuple[int]] = 1280,
        attention_head_dim: Union[int, Tuple[int]] = 8,
        use_linear_projection: bool = False,
        # parameters added by the reference (unet_2d_condition_multiview.py:173-179)
        trainable_state="only_new",
        neighboring_view_pair: Optional[dict] = None,
        neighboring_attn_type: str = "add",
        zero_module_type: str = "zero_linear",
        crossview_attn_type: str = "basic",
        img_size: Optional[Tuple[int, int]] = None,
        **other_diffusers_config,
    ):
        super().__init__()
        cfg = dict(locals())
        for k in ("self", "__class__", "other_diffusers_config"):
            cfg.pop(k, None)
        cfg.update(other_diffusers_config)
        self._register_config(**cfg)
        # the SD-v1.5 / DualDiff configuration space this implementation covers
        unsupported = {
            "act_fn": act_fn != "silu", "use_linear_projection": use_linear_projection,
            "only_cross_attention": bool(only_cross_attention) and only_cross_attention is not False,
            "center_input_sample": center_input_sample, "crossview_attn_type": crossview_attn_type != "basic",
            "mid_block_type": mid_block_type != "UNetMidBlock2DCrossAttn",
            "class_embed_type": other_diffusers_config.get("class_embed_type") is not None,
            "addition_embed_type": other_diffusers_config.get("addition_embed_type") is not None,
            "encoder_hid_dim": other_diffusers_config.get("encoder_hid_dim") is not None,
            "mid_block_scale_factor": mid_block_scale_factor != 1, "downsample_padding": downsample_padding != 1,
        }
        bad = [k for k, v in unsupported.items() if v]
        if bad:
            raise NotImplementedError("UNet2DConditionModelMultiview (HIP): unsupported config: %s" % bad)
        n = len(down_block_types)
        heads = (attention_head_dim,) * n if isinstance(attention_head_dim, int) else tuple(attention_head_dim)
        layers = (layers_per_block,) * n if isinstance(layers_per_block, int) else tuple(layers_per_block)
        xdim = (cross_attention_dim,) * n if isinstance(cross_attention_dim, int) else tuple(cross_attention_dim)
        self.crossview_attn_type = crossview_attn_type
        self.img_size = [int(s) for s in img_size] if img_size is not None else None
        self.trainable_state = trainable_state
        self._new_module = {}
        if self.multiview:
            pair = neighboring_view_pair if neighboring_view_pair is not None else _DEFAULT_PAIR
            blk_kw = dict(neighboring_view_pair=pair, neighboring_attn_type=neighboring_attn_type,
                          zero_module_type=zero_module_type)
            bcls = BasicMultiviewTransformerBlock
            nf = other_diffusers_config.get("n_frames")
            if self.video:                                   # extension: ST-Attn + temporal attention per block
                from .video_blocks import VideoMultiviewTransformerBlock
                bcls = VideoMultiviewTransformerBlock
                blk_kw["n_frames"] = int(nf or 1)
        else:
            blk_kw, bcls = {}, BasicTransformerBlock

        c0 = block_out_channels[0]
        ted = c0 * 4
        self.conv_in = Conv3x3(in_channels, c0)
        self.time_proj = Timesteps(c0, flip_sin_to_cos, freq_shift)
        self.time_embedding = TimestepEmbedding(c0, ted)
        self.class_embedding = None
        self.down_blocks = nn.ModuleList()
        oc = c0
        for i, t in enumerate(down_block_types):
            ic, oc = oc, block_out_channels[i]
            final = i == n - 1
            if t == "CrossAttnDownBlock2D":
                self.down_blocks.append(CrossAttnDownBlock2D(ic, oc, ted, layers[i], heads[i], xdim[i], not final,
                                                             norm_num_groups, norm_eps, bcls, blk_kw))
            elif t == "DownBlock2D":
                self.down_blocks.append(DownBlock2D(ic, oc, ted, layers[i], not final, norm_num_groups, norm_eps))
            else:
                raise NotImplementedError(t)
        self.mid_block = UNetMidBlock2DCrossAttn(block_out_channels[-1], ted, heads[-1], xdim[-1],
                                                 norm_num_groups, norm_eps, bcls, blk_kw)
        self.up_blocks = nn.ModuleList()
        self.num_upsamplers = 0
        rev = list(reversed(block_out_channels))
        rheads, rlayers, rxd = list(reversed(heads)), list(reversed(layers)), list(reversed(xdim))
        oc = rev[0]
        for i, t in enumerate(up_block_types):
            final = i == n - 1
            prev, oc = oc, rev[i]
            ic = rev[min(i + 1, n - 1)]
            if not final:
                self.num_upsamplers += 1
            if t == "CrossAttnUpBlock2D":
                self.up_blocks.append(CrossAttnUpBlock2D(ic, oc, prev, ted, rlayers[i] + 1, rheads[i], rxd[i],
                                                         not final, norm_num_groups, norm_eps, bcls, blk_kw))
            elif t == "UpBlock2D":
                self.up_blocks.append(UpBlock2D(ic, prev, oc, ted, rlayers[i] + 1, not final,
                                                norm_num_groups, norm_eps))
            else:
                raise NotImplementedError(t)
        self.conv_norm_out = GroupNorm(norm_num_groups, c0, norm_eps)
        self.conv_out = Conv3x3(c0, out_channels)

    # -- training-surface stubs (the released reference trains no UNet parameters, :221,233-242) --
    @property
    def trainable_module(self) -> Dict[str, nn.Module]:
        if self.trainable_state == "all":
            return {self.__class__: self}
        if self.trainable_state == "only_new":
            return self._new_module
        raise ValueError(f"Unknown trainable_state: {self.trainable_state}")

    @property
    def trainable_parameters(self):
        return [p for mod in self.trainable_module.values() for p in mod.parameters()]

    @classmethod
    def from_unet_2d_condition(cls, unet, load_weights_from_unet: bool = True, **kwargs):
        """unet_2d_condition_multiview.py:294-325: build from a plain SD UNet's config/weights."""
        model = cls(**{k: v for k, v in dict(unet.config).items() if not k.startswith("_")}, **kwargs)
        if load_weights_from_unet:
            model.load_state_dict(unet.state_dict(), strict=False)
        return model

    def set_view_shard(self, shard):
        """Spread the views of a scene over several GPUs (dualdiff_amd.parallel.ViewShard; None = off): the
        model then runs on this rank's view-instances only and every `BasicMultiviewTransformerBlock`
        fetches the neighbour views' K/V through `shard.exchange` (blocks.py:106-142)."""
        for mod in self.modules():
            if isinstance(mod, BasicMultiviewTransformerBlock):
                mod.view_shard = shard
        self.view_shard = shard

    # -- forward -----------------------------------------------------------------------------------
    def _timesteps(self, timestep, m, device):
        t = timestep
        if not torch.is_tensor(t):
            t = torch.tensor([float(t)], dtype=torch.float32, device=device)
        t = t.to(device=device, dtype=torch.float32).reshape(-1)
        return t.expand(m).contiguous()

    @sibling_overlap
    def forward(
        self,
        sample: torch.Tensor,
        timestep: Union[torch.Tensor, float, int],
        encoder_hidden_states: torch.Tensor,
        class_labels: Optional[torch.Tensor] = None,
        timestep_cond: Optional[torch.Tensor] = None,
        attention_mask: Optional[torch.Tensor] = None,
        cross_attention_kwargs: Optional[Dict[str, Any]] = None,
        down_block_additional_residuals: Optional[Tuple[torch.Tensor]] = None,
        mid_block_additional_residual: Optional[torch.Tensor] = None,
        return_dict: bool = True,
    ):
        """Same contract as the reference forward (:327-339): `sample` (M, 4, h, w) NCHW, `timestep`
        scalar / 0-d / (M,), `encoder_hidden_states` (M, Lc, 768), optional ControlNet residuals
        (NCHW-shaped tensors; channels_last strides are consumed zero-copy)."""
        if attention_mask is not None or timestep_cond is not None or class_labels is not None:
            raise NotImplementedError("attention_mask / timestep_cond / class_labels are None on the "
                                      "denoising path (pipeline_bev_controlnet.py:476-484)")
        if cross_attention_kwargs:
            raise NotImplementedError("cross_attention_kwargs is unused by the DualDiff pipeline")
        if not sample.is_cuda:
            raise RuntimeError("dualdiff_amd runs on the GPU only; got a %s tensor" % sample.device)
        m = sample.shape[0]
        n_down = None if down_block_additional_residuals is None else len(down_block_additional_residuals)
        # Context length (round 6): a context of CTX_BASE + N_box tokens is laid out at its bucket's capacity and its real
        # length travels as one int32 in device memory (layers.context_keys) — one graph per bucket, not per box count
        # (the ControlNet's tokens arrive as a view of such a layout and are read in place).
        ctx, lk = encoder_hidden_states, None
        if ctx.dim() == 3 and ctx.shape[1] >= CTX_BASE and self._varlen_ok():
            lk = lk_const(ctx.shape[1], sample.device)
            ctx = _at_capacity(ctx, ctx_capacity(ctx.shape[1]))
        tensors = [sample, self._timesteps(timestep, m, sample.device), ctx, lk]
        tensors += list(down_block_additional_residuals or ()) + [mid_block_additional_residual]
        graphs = self._graphs()
        if graphs is None:
            out = self._forward_flat(tensors, n_down)[0]
        else:
            # (the noise prediction is small, 4 channels: always the caller's own copy, never a view of graph memory)
            out = graphs.call(("unet", n_down, graphs.flags(self)), tensors, lambda ts: self._forward_flat(ts, n_down))[0]
        if not return_dict:
            return (out,)
        return UNet2DConditionOutput(sample=out)

    def _forward_flat(self, tensors, n_down):
        """forward() on a flat tensor list [sample, t (m,) fp32, encoder_hidden_states, real context length (int32 [1]) or
        None, *down residuals, mid residual] (what ForwardGraphs records): NCHW in, NCHW out, residuals NCHW-shaped
        (channels_last strides are zero-copy)."""
        sample, t_f32, encoder_hidden_states, lk_dev = tensors[:4]
        down = None if n_down is None else tensors[4:4 + n_down]
        mid = tensors[-1]
        dt = self.dtype
        x, m, h, w = to_nhwc(sample.to(dt))
        if x.shape[1] != self.conv_in.cin_pad:
            x = torch.nn.functional.pad(x, (0, self.conv_in.cin_pad - x.shape[1]))
        down_res = None if down is None else [to_nhwc(r.to(dt))[0] for r in down]
        mid_res = None if mid is None else to_nhwc(mid.to(dt))[0]
        ctx = encoder_hidden_states.to(dt)
        lc = ctx.shape[1]
        ctx2d = ctx.reshape(m * lc, ctx.shape[2])
        if not ctx2d.is_contiguous():
            ctx2d = ctx2d.contiguous()
        with context_keys(ctx2d, lc, lk_dev):
            return [self.forward_nhwc(x, m, h, w, t_f32, ctx2d, lc, down_res, mid_res)]

    def forward_nhwc(self, x, m, h, w, t_f32, ctx2d, lc, down_res=None, mid_res=None):
        """x: (m*h*w, 8) NHWC latents (4 channels zero-padded to 8); residuals: NHWC 2-D tensors in
        skip order (each entry a tensor or a tuple of tensors to be summed).  Returns eps as
        (m, 4, h, w) NCHW."""
        state = self.encode_nhwc(x, m, h, w, t_f32, ctx2d, lc)
        return self.decode_nhwc(state, down_res, mid_res)

    def encode_nhwc(self, x, m, h, w, t_f32, ctx2d, lc):
        """conv_in + down path + mid block — everything that does NOT depend on the ControlNet
        residuals, so a sampler can overlap it with the ControlNet branches on other streams."""
        dt = self.dtype
        if self.kv_bank:
            if self.__dict__.get("_kv_bank") is None:
                self.__dict__["_kv_bank"] = CrossKVBank(self)
            self.__dict__["_kv_bank"].run(ctx2d)
        elif self.prefetch_kv:
            if self.__dict__.get("_kv_stream") is None:
                self.__dict__["_kv_stream"] = torch.cuda.Stream()
            prefetch_cross_kv(self, ctx2d, self.__dict__["_kv_stream"])
        # 1. time (unet_2d_condition_multiview.py:404-411)
        emb = self.time_embedding.run(self.time_proj.run(t_f32, dt))
        temb = self.temb_bank.run(O.silu(emb))
        # 28 % 8 != 0 -> explicit upsample sizes (:363-374, :500-501)
        forward_size = any(s % (2 ** self.num_upsamplers) != 0 for s in (h, w))
        # 2./3. conv_in + down path (:443-462)
        x = self.conv_in.run(x, m, h, w)
        skips = [(x, h, w)]
        for blk in self.down_blocks:
            x, h, w, s = run_down_block(blk, x, m, h, w, temb, ctx2d, lc)
            skips += s
        # 4. mid (:476-485); its input is the un-augmented down output, the residual is added after
        x = self.mid_block.run(x, m, h, w, temb, ctx2d, lc)
        return {"x": x, "m": m, "h": h, "w": w, "skips": skips, "temb": temb, "ctx2d": ctx2d, "lc": lc,
                "forward_size": forward_size}

    def decode_nhwc(self, state, down_res=None, mid_res=None):
        x, m, h, w = state["x"], state["m"], state["h"], state["w"]
        skips, temb, ctx2d, lc = list(state["skips"]), state["temb"], state["ctx2d"], state["lc"]

        def plus(t, r):
            """t + sum(r): any number of branches (pipeline_bev_controlnet.py:421-429 sums them all),
            two operands per 3-input add."""
            if not isinstance(r, (tuple, list)):
                r = (r,)
            if len(r) == 0:
                return t
            i = 0
            while i < len(r):
                t = O.add(t, r[i], r[i + 1]) if i + 1 < len(r) else O.add(t, r[i])
                i += 2
            return t

        # ControlNet residual add on the skips (:464-473) and on the mid output (:487-488)
        if down_res is not None:
            if len(down_res) != len(skips):
                raise ValueError("expected %d down-block residuals, got %d" % (len(skips), len(down_res)))
            skips = [(plus(s, r), sh, sw) for (s, sh, sw), r in zip(skips, down_res)]
        if mid_res is not None:
            x = plus(x, mid_res)
        # 5. up (:491-516)
        for i, blk in enumerate(self.up_blocks):
            final = i == len(self.up_blocks) - 1
            k = len(blk.resnets)
            up_size = None
            if not final:
                nxt = skips[-k - 1]
                up_size = (nxt[1], nxt[2]) if state["forward_size"] else None
            x, h, w = run_up_block(blk, x, m, h, w, skips, temb, ctx2d, lc, up_size)
        # 6. post-process (:519-522): GN + SiLU fused, conv_out writes NCHW directly
        a = self.conv_norm_out.run(x, m, h * w, True)
        if self.__dict__.get("_kv_stream") is not None:
            drop_prefetched_kv(self, self.__dict__["_kv_stream"])
        if self.__dict__.get("_kv_bank") is not None:
            self.__dict__["_kv_bank"].drop()
        return O.conv3x3_small_cout(a, self.conv_out.packed, self.conv_out.bias, m, h, w)



class UNet2DConditionModel(UNet2DConditionModelMultiview):
    """The stock SD-v1.5 UNet (diffusers `UNet2DConditionModel`): same network without the neighbour-view
    attention — `BasicTransformerBlock` instead of `BasicMultiviewTransformerBlock`, any number of views
    (BASELINE configs[0]: one view, null text, no ControlNet).  It is what the reference loads from
    `pretrained_model_name_or_path` and hands to `from_unet_2d_condition` (unet_2d_condition_multiview.py:
    294-325); state-dict names are the diffusers ones, so `UNet2DConditionModelMultiview.
    from_unet_2d_condition(plain_unet)` copies every shared weight."""
    multiview = False



class UNet2DConditionModelMultiviewVideo(UNet2DConditionModelMultiview):
    """EXTENSION (BASELINE configs[3]; no reference semantics — see networks/video_blocks.py): the multiview
    UNet with ST-Attn and temporal attention in every transformer block.  `n_frames` (config key) frames per
    scene; instances are ordered scene-major, then frame, then view.  State dict = the image model's keys plus
    `...transformer_blocks.0.{norm_temp, attn_temp.*}`, so `load_state_dict(image_sd, strict=False)` starts a
    video model from an image checkpoint (attn_temp.to_out zero-initialised by the trainer)."""
    video = True

    def set_n_frames(self, n_frames):
        from .video_blocks import VideoMultiviewTransformerBlock
        for mod in self.modules():
            if isinstance(mod, VideoMultiviewTransformerBlock):
                mod.n_frames = int(n_frames)
        self.config["n_frames"] = int(n_frames)

    def set_frame_shard(self, shard):
        """Frame split (parallel.FrameShard): this rank holds `shard.plan.local` of the n_frames frames
        of every scene (all views of a frame local); None restores the unsharded model."""
        from .video_blocks import VideoMultiviewTransformerBlock
        blocks = [mod for mod in self.modules() if isinstance(mod, VideoMultiviewTransformerBlock)]
        if shard is not None and any(b.n_frames != shard.plan.n_frames for b in blocks):
            raise ValueError("frame shard plans %d frames, the model is configured for %d"
                             % (shard.plan.n_frames, blocks[0].n_frames))
        for mod in blocks:
            mod.frame_shard = shard
        self.frame_shard = shard
