"""CPU restatement of the reference-OWNED parts of the hot path — TEST INFRASTRUCTURE ONLY.

Every class cites the reference source it restates (paths relative to
/root/reference/MD_txt_con_fusion/).  Unlike oracle/diffusers_restated.py, the code in this
file IS pinned: tests/golden/mint.py executes the reference's own source files (under a stub
`diffusers`/`xformers` namespace) on seeded inputs and tests/test_oracle_golden.py checks
these restatements against those outputs (tests/golden/*.npz).

Written as plain maths (project once, attend, sum), not as a transcription of the reference's
control flow: e.g. the neighbour-view attention projects Q/K/V once per view instead of once
per (view, neighbour) pair — the two are algebraically identical and the golden vectors prove
it numerically.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from .numerics import prob_round, stor

from . import diffusers_restated as D


def sdpa(q, k, v, heads, scale):
    """xformers.ops.memory_efficient_attention on (B, L, C) tensors split into `heads`
    (networks/box_adapter.py:115-156, networks/txt_con_fusion.py:121-162)."""
    b, lq, c = q.shape
    d = c // heads

    def split(t):
        return t.reshape(t.shape[0], t.shape[1], heads, d).permute(0, 2, 1, 3).float()

    s = (split(q) @ split(k).transpose(-1, -2)) * scale
    o = prob_round(torch.softmax(s, dim=-1)) @ split(v)
    return o.permute(0, 2, 1, 3).reshape(b, lq, c).to(q.dtype)


# ----------------------------------------------------------------------------- SFA (A12/13)
class TxtConFusion(nn.Module):
    """Semantic Fusion Attention, networks/txt_con_fusion.py:18-181 (`txt_con_XFormersAttn`):
    Q from the 320-ch ORS condition map, K/V from the 768-d text tokens, 8 heads x 40,
    out-projection with bias, residual add (residual_connection=True :40,:176-177)."""

    def __init__(self, con_dim=320, txt_dim=768, hidden_size=320):
        super().__init__()
        self.to_q = nn.Linear(con_dim, hidden_size, bias=False)
        self.to_k = nn.Linear(txt_dim, hidden_size, bias=False)
        self.to_v = nn.Linear(txt_dim, hidden_size, bias=False)
        self.to_out = nn.ModuleList([nn.Linear(hidden_size, hidden_size, bias=True), nn.Dropout(0.0)])
        self.heads = 8
        self.scale = (hidden_size // self.heads) ** -0.5

    def forward(self, hidden_states, encoder_hidden_states):
        b, c, h, w = hidden_states.shape
        x = hidden_states.reshape(b, c, h * w).transpose(1, 2)
        o = sdpa(self.to_q(x), self.to_k(encoder_hidden_states), self.to_v(encoder_hidden_states),
                 self.heads, self.scale)
        o = self.to_out[0](o)
        return o.transpose(1, 2).reshape(b, c, h, w) + hidden_states


class TxtConFusionPlus(nn.Module):
    """networks/txt_con_fusion.py:184-337 (`txt_con_XFormersAttn_plus`): q' = Attn(q_occ, k_txt,
    v_txt); out = Attn(q', k_occ, v_occ); out-proj; + residual (:313-333)."""

    def __init__(self, con_dim=320, txt_dim=768, hidden_size=320):
        super().__init__()
        self.to_q_occ = nn.Linear(con_dim, hidden_size, bias=False)
        self.to_k_occ = nn.Linear(con_dim, hidden_size, bias=False)
        self.to_v_occ = nn.Linear(con_dim, hidden_size, bias=False)
        self.to_k_txt = nn.Linear(txt_dim, hidden_size, bias=False)
        self.to_v_txt = nn.Linear(txt_dim, hidden_size, bias=False)
        self.to_out = nn.ModuleList([nn.Linear(hidden_size, hidden_size, bias=True), nn.Dropout(0.0)])
        self.heads = 8
        self.scale = (hidden_size // self.heads) ** -0.5

    def forward(self, hidden_states, encoder_hidden_states):
        b, c, h, w = hidden_states.shape
        x = hidden_states.reshape(b, c, h * w).transpose(1, 2)
        e = encoder_hidden_states
        q1 = sdpa(self.to_q_occ(x), self.to_k_txt(e), self.to_v_txt(e), self.heads, self.scale)
        o = sdpa(q1, self.to_k_occ(x), self.to_v_occ(x), self.heads, self.scale)
        o = self.to_out[0](o)
        return o.transpose(1, 2).reshape(b, c, h, w) + hidden_states


# ------------------------------------------------------------------- embedders (A11, A16) --
def fourier_embed(x, num_freqs):
    """networks/embedder.py:18-67 `get_embedder(input_dims, num_freqs)` (include_input, log
    sampling): [x, sin(2^0 x), cos(2^0 x), ..., sin(2^(F-1) x), cos(2^(F-1) x)] on the last dim."""
    outs = [x]
    for f in range(num_freqs):
        freq = 2.0 ** f
        outs += [torch.sin(x * freq), torch.cos(x * freq)]
    return torch.cat(outs, dim=-1)


class ControlNetConditioningEmbedding(nn.Module):
    """networks/map_embedder.py:81-138: panorama (b,3,224,2400) -> 6 views -> conv 3->16, then
    [c->c, c->c' s2] x3, SiLU after each, zero-init conv 256->320."""

    def __init__(self, conditioning_embedding_channels, conditioning_channels=3,
                 block_out_channels=(16, 32, 96, 256), conditioning_size=None):
        super().__init__()
        self.conv_in = nn.Conv2d(conditioning_channels, block_out_channels[0], 3, padding=1)
        self.blocks = nn.ModuleList()
        for i in range(len(block_out_channels) - 1):
            ci, co = block_out_channels[i], block_out_channels[i + 1]
            self.blocks.append(nn.Conv2d(ci, ci, 3, padding=1))
            self.blocks.append(nn.Conv2d(ci, co, 3, padding=1, stride=2))
        self.conv_out = D.zero_module(nn.Conv2d(block_out_channels[-1], conditioning_embedding_channels, 3, padding=1))

    def forward(self, conditioning):
        b, c, h, pw = conditioning.shape
        w = pw // 6
        x = conditioning.reshape(b, c, h, 6, w).permute(0, 3, 1, 2, 4).reshape(b * 6, c, h, w)
        x = F.silu(self.conv_in(x))
        for blk in self.blocks:
            x = F.silu(blk(x))
        return self.conv_out(x)


class BEVControlNetConditioningEmbedding(nn.Module):
    """networks/map_embedder.py:10-77 (vanilla MagicDrive's BEV-map embedder; unused by the DualDiff ORS configs):
    the (b, 25, 200, 200) map is repeated for the 6 views, conv 25->32, then [c->c pad 1 | c->c' pad (2, 1) stride 2]
    for the first len-2 stages, [c->c pad (2, 1) | c->c' pad (2, 1) stride (2, 1)] for the last one — 200x200 ->
    101x100 -> 52x50 -> 54x50 -> 28x50 — SiLU after every conv, zero-init conv 256->320."""

    def __init__(self, conditioning_embedding_channels=320, conditioning_size=(25, 200, 200),
                 block_out_channels=(32, 64, 128, 256)):
        super().__init__()
        self.conv_in = nn.Conv2d(conditioning_size[0], block_out_channels[0], 3, padding=1)
        self.blocks = nn.ModuleList()
        for i in range(len(block_out_channels) - 2):
            ci, co = block_out_channels[i], block_out_channels[i + 1]
            self.blocks.append(nn.Conv2d(ci, ci, 3, padding=1))
            self.blocks.append(nn.Conv2d(ci, co, 3, padding=(2, 1), stride=2))
        ci, co = block_out_channels[-2], block_out_channels[-1]
        self.blocks.append(nn.Conv2d(ci, ci, 3, padding=(2, 1)))
        self.blocks.append(nn.Conv2d(ci, co, 3, padding=(2, 1), stride=(2, 1)))
        self.conv_out = D.zero_module(nn.Conv2d(co, conditioning_embedding_channels, 3, padding=1))

    def forward(self, conditioning):
        x = conditioning.repeat_interleave(6, dim=0)                  # 'b ... -> (b repeat) ...'
        x = F.silu(self.conv_in(x))
        for blk in self.blocks:
            x = F.silu(blk(x))
        return self.conv_out(x)


class BBoxEmbedder(nn.Module):
    """networks/bbox_embedder.py:28-203 `ContinuousBBoxWithTextEmbedding` (mode 'all-xyz',
    minmax_normalize False): Fourier(8 corners x 3) -> Linear -> SiLU -> cat class token ->
    3-layer MLP; masked boxes use the learned null features (:186-193)."""

    def __init__(self, n_classes=10, class_token_dim=768, embedder_num_freq=4,
                 proj_dims=(768, 512, 512, 768), n_points=8, **unused):
        super().__init__()
        self.num_freq = embedder_num_freq
        fdim = 3 * (1 + 2 * embedder_num_freq) * n_points
        self.bbox_proj = nn.Linear(fdim, proj_dims[0])
        self.second_linear = nn.Sequential(
            nn.Linear(proj_dims[0] + class_token_dim, proj_dims[1]), nn.SiLU(),
            nn.Linear(proj_dims[1], proj_dims[2]), nn.SiLU(),
            nn.Linear(proj_dims[2], proj_dims[3]))
        self.register_buffer("_class_tokens", torch.randn(n_classes, class_token_dim))
        self.null_class_feature = nn.Parameter(torch.zeros(class_token_dim))
        self.null_pos_feature = nn.Parameter(torch.zeros(fdim))

    def forward_feature(self, pos_emb, cls_emb):
        e = F.silu(self.bbox_proj(pos_emb))
        return self.second_linear(torch.cat([e, cls_emb], dim=-1))

    def forward(self, bboxes, classes, masks=None, return_cls_emb=False):
        b, n = classes.shape
        pts = bboxes.reshape(b * n, -1, 3)
        m = torch.ones(b * n) if masks is None else masks.reshape(-1)
        m = m.unsqueeze(-1).type_as(self.null_pos_feature)
        pos = fourier_embed(pts, self.num_freq).reshape(b * n, -1).type_as(self.null_pos_feature)
        pos = pos * m + self.null_pos_feature[None] * (1 - m)
        cls = self._class_tokens[classes.reshape(-1)]
        cls = cls * m + self.null_class_feature[None] * (1 - m)
        emb = self.forward_feature(pos, cls).reshape(b, n, -1)
        if return_cls_emb:                                                  # :199-203
            return emb, cls.reshape(b, n, -1)
        return emb


# ----------------------------------------------------------- box / class adapter (N1) --
class AdapterAttnProcessor(nn.Module):
    """networks/box_adapter.py:177-411 `Adapter_XFormersAttnProcessor`: the context is
    [text(+cam) | box tokens | class tokens]; the text part feeds the layer's own K/V; the box tokens
    get their own K/V projections, each enriched by an attention over the class tokens' K/V
    (:349-357), and a second attention of the SAME queries over them is added with `scale` (:380-388)
    before the out-projection."""

    def __init__(self, hidden_size, cross_attention_dim=None, scale=1.0, num_tokens=200):
        super().__init__()
        self.scale, self.num_tokens = scale, num_tokens
        d = cross_attention_dim or hidden_size
        self.to_k_box = nn.Linear(d, hidden_size, bias=False)
        self.to_v_box = nn.Linear(d, hidden_size, bias=False)
        self.to_k_cls = nn.Linear(d, hidden_size, bias=False)
        self.to_v_cls = nn.Linear(d, hidden_size, bias=False)

    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, temb=None):
        assert attention_mask is None and encoder_hidden_states is not None
        q = attn.to_q(hidden_states)
        e = encoder_hidden_states
        end = e.shape[1] - self.num_tokens                                   # :276-284
        e, cls_h = e[:, :end], e[:, end:]
        end = e.shape[1] - self.num_tokens
        e, box_h = e[:, :end], e[:, end:]
        out = sdpa(q, attn.to_k(e), attn.to_v(e), attn.heads, attn.scale)    # :291-292,:339-341
        bk, bv = self.to_k_box(box_h), self.to_v_box(box_h)                  # :294-298
        ck, cv = self.to_k_cls(cls_h), self.to_v_cls(cls_h)
        bk = bk + sdpa(bk, ck, cv, attn.heads, attn.scale)                   # :349-353
        bv = bv + sdpa(bv, ck, cv, attn.heads, attn.scale)                   # :354-357
        out = out + self.scale * sdpa(q, bk, bv, attn.heads, attn.scale)     # :380-388
        return attn.to_out[0](out)                                           # :391


def box_adapter(net):
    """networks/box_adapter.py:414-444 (use_box_token False): the adapter on every text cross-attention
    (attn2), initialised from that layer's own to_k / to_v; plain processors elsewhere."""
    sd = net.state_dict()
    for name, m in [(n, mm) for n, mm in net.named_modules() if hasattr(mm, "set_processor")]:
        if name.endswith("attn1") or name.endswith("attn4"):
            continue
        p = AdapterAttnProcessor(hidden_size=m.to_q.out_features, cross_attention_dim=m.to_k.in_features)
        p.load_state_dict({"to_k_box.weight": sd[name + ".to_k.weight"], "to_v_box.weight": sd[name + ".to_v.weight"],
                           "to_k_cls.weight": sd[name + ".to_k.weight"], "to_v_cls.weight": sd[name + ".to_v.weight"]})
        m.set_processor(p)          # an nn.Module processor becomes the sub-module `<attn>.processor`
    return net


# -------------------------------------------------------------- multiview block (A4, A6) --
class GatedConnector(nn.Module):
    """networks/blocks.py:24-32: tanh(alpha) * x with a zero-initialised per-channel alpha."""

    def __init__(self, dim):
        super().__init__()
        self.alpha = nn.Parameter(torch.zeros(dim))

    def forward(self, x):
        return torch.tanh(self.alpha) * x


class BasicMultiviewTransformerBlock(D.BasicTransformerBlock):
    """networks/blocks.py:35-238.  neighboring_attn_type (:106-142): 'add' — for view v and each neighbour u in
    pair[v]: to_out(Attn(Wq x_v, Wk x_u, Wv x_u)), the out-projection (with its bias) applied per pair and the
    pairs summed (:203-217); 'concat' — one attention of view v over the concatenated tokens of its neighbours;
    'self' — one self-attention over the tokens of ALL views of a scene.  zero_module_type (:81-90):
    'zero_linear' Linear connector, 'gated' tanh(alpha) gate, 'none' identity.  Then residual (:220-222)."""

    def __init__(self, *args, neighboring_view_pair=None, neighboring_attn_type="add",
                 zero_module_type="zero_linear", **kw):
        super().__init__(*args, **kw)
        assert neighboring_attn_type in ("add", "concat", "self")
        dim, heads, hd = self._args["dim"], self._args["num_attention_heads"], self._args["attention_head_dim"]
        self.neighboring_view_pair = {int(k): [int(x) for x in v] for k, v in neighboring_view_pair.items()}
        self.neighboring_attn_type = neighboring_attn_type
        self.norm4 = nn.LayerNorm(dim)
        self.attn4 = D.Attention(query_dim=dim, cross_attention_dim=dim, heads=heads, dim_head=hd)
        if zero_module_type == "zero_linear":
            self.connector = D.zero_module(nn.Linear(dim, dim))
        elif zero_module_type == "gated":
            self.connector = GatedConnector(dim)
        elif zero_module_type == "none":
            self.connector = lambda x: x
        else:
            raise TypeError(zero_module_type)

    def forward(self, hidden_states, attention_mask=None, encoder_hidden_states=None,
                encoder_attention_mask=None, timestep=None, cross_attention_kwargs=None, class_labels=None):
        h = hidden_states
        h = stor(self.attn1(self.norm1(h)) + h)
        h = stor(self.attn2(self.norm2(h), encoder_hidden_states=encoder_hidden_states) + h)
        n_cam = len(self.neighboring_view_pair)
        x = self.norm4(h)
        xv = x.reshape(-1, n_cam, x.shape[1], x.shape[2])          # (b, view, tokens, C)
        a = self.attn4
        if self.neighboring_attn_type == "self":                     # :135-139: all views in one sequence
            xs = xv.reshape(xv.shape[0], n_cam * xv.shape[2], xv.shape[3])
            out = a.to_out[0](sdpa(a.to_q(xs), a.to_k(xs), a.to_v(xs), a.heads, a.scale)).reshape_as(xv)
        else:
            q, k, v = a.to_q(xv), a.to_k(xv), a.to_v(xv)
            out = torch.zeros_like(xv)
            for view, neighbours in self.neighboring_view_pair.items():
                if self.neighboring_attn_type == "concat":           # :122-134: keys of all neighbours together
                    kc = torch.cat([k[:, u] for u in neighbours], dim=1)
                    vc = torch.cat([v[:, u] for u in neighbours], dim=1)
                    out[:, view] = a.to_out[0](sdpa(q[:, view], kc, vc, a.heads, a.scale))
                    continue
                for u in neighbours:
                    o = sdpa(q[:, view], k[:, u], v[:, u], a.heads, a.scale)
                    out[:, view] = stor(out[:, view] + a.to_out[0](o))
        h = stor(self.connector(out.reshape_as(x)) + h)
        return stor(self.ff(self.norm3(h)) + h)


class UNet2DConditionModelMultiview(D.UNet2DConditionModel):
    """networks/unet_2d_condition_multiview.py:44-527: SD-v1.5 UNet with every
    BasicTransformerBlock replaced by the multiview block (:222-234); forward = parent forward
    (incl. ControlNet residual adds :464-473,:487-488)."""

    def __init__(self, neighboring_view_pair=None, **kw):
        super().__init__(**kw)
        self._config["neighboring_view_pair"] = neighboring_view_pair
        for name, mod in list(self.named_modules()):
            if type(mod) is D.BasicTransformerBlock:
                parent = self
                *path, leaf = name.split(".")
                for p in path:
                    parent = getattr(parent, p)
                setattr(parent, leaf, BasicMultiviewTransformerBlock(
                    **mod._args, neighboring_view_pair=neighboring_view_pair))


# ---------------------------------------------------------------------- ControlNet (A10) --
class BEVControlNetModel(D.ModelMixin):
    """networks/unet_addon_rawbox.py:39-1082, eval path (no condition dropout), flags as
    build_pipe sets them (misc/test_utils.py:123-136): use_cam_in_temb False, box adapter off,
    SFA (`use_txt_con_fusion`) on/off, `use_occ_3d` selects raw 320-ch ORS-3D input vs the
    panorama embedder."""

    def __init__(self, in_channels=4, block_out_channels=(320, 640, 1280, 1280), layers_per_block=2,
                 cross_attention_dim=768, attention_head_dim=8, norm_num_groups=32, norm_eps=1e-5,
                 camera_in_dim=189, camera_out_dim=768, uncond_cam_in_dim=(3, 7), cam_num_freqs=4,
                 conditioning_embedding_out_channels=(16, 32, 96, 256), n_box_points=8,
                 use_txt_con_fusion=True, use_occ_3d=False):
        super().__init__()
        c0 = block_out_channels[0]
        ted = c0 * 4
        self.cam_num_freqs = cam_num_freqs
        self.cam2token = nn.Linear(camera_in_dim, camera_out_dim)
        self.uncond_cam = nn.Embedding(1, uncond_cam_in_dim[0] * uncond_cam_in_dim[1])
        self.uncond_cam_num = uncond_cam_in_dim[1]
        self.conv_in = nn.Conv2d(in_channels, c0, 3, padding=1)
        self.time_proj = D.Timesteps(c0, True, 0)
        self.time_embedding = D.TimestepEmbedding(c0, ted)
        self.use_occ_3d, self.use_txt_con_fusion = use_occ_3d, use_txt_con_fusion
        self.controlnet_cond_embedding = None if use_occ_3d else ControlNetConditioningEmbedding(
            c0, block_out_channels=conditioning_embedding_out_channels)
        self.bbox_embedder = BBoxEmbedder(n_points=n_box_points)
        self.down_blocks = nn.ModuleList()
        self.controlnet_down_blocks = nn.ModuleList([D.zero_module(nn.Conv2d(c0, c0, 1))])
        types = ("CrossAttnDownBlock2D",) * 3 + ("DownBlock2D",)
        oc = c0
        for i, t in enumerate(types):
            ic, oc = oc, block_out_channels[i]
            final = i == len(types) - 1
            self.down_blocks.append(D.get_down_block(
                t, num_layers=layers_per_block, in_channels=ic, out_channels=oc, temb_channels=ted,
                add_downsample=not final, resnet_eps=norm_eps, resnet_act_fn="silu",
                resnet_groups=norm_num_groups, cross_attention_dim=cross_attention_dim,
                attn_num_head_channels=attention_head_dim, downsample_padding=1))
            for _ in range(layers_per_block + (0 if final else 1)):
                self.controlnet_down_blocks.append(D.zero_module(nn.Conv2d(oc, oc, 1)))
        self.controlnet_mid_block = D.zero_module(nn.Conv2d(oc, oc, 1))
        self.mid_block = D.UNetMidBlock2DCrossAttn(
            in_channels=oc, temb_channels=ted, resnet_eps=norm_eps, resnet_groups=norm_num_groups,
            cross_attention_dim=cross_attention_dim, attn_num_head_channels=attention_head_dim)
        self.txt_con_fusion = TxtConFusion() if use_txt_con_fusion else None

    # :327-335
    def uncond_cam_param(self, repeat_size=1):
        if isinstance(repeat_size, int):
            repeat_size = [1, repeat_size]
        n = int(math.prod(repeat_size))
        p = self.uncond_cam(torch.zeros(n, dtype=torch.long, device=self.device))
        return p.reshape(*repeat_size, -1, self.uncond_cam_num)

    # :308-325 — Fourier-embed each of the 7 column 3-vectors, concatenate per view (189 dims)
    def _embed_camera(self, camera_param):
        b, n, _, k = camera_param.shape
        e = fourier_embed(camera_param.permute(0, 1, 3, 2), self.cam_num_freqs)   # (b, n, 7, 27)
        return e.reshape(b, n, -1)

    def forward(self, sample, timestep, camera_param, bboxes_3d_data, encoder_hidden_states,
                controlnet_cond, conditioning_scale=1.0, guess_mode=False):
        b, n_cam = camera_param.shape[:2]
        cam_tok = self.cam2token(self._embed_camera(camera_param))                      # :349
        txt = encoder_hidden_states[:, None].expand(-1, n_cam, -1, -1)                    # :354
        ctx = torch.cat([cam_tok[:, :, None], txt], dim=2).reshape(b * n_cam, -1, txt.shape[-1])   # :355-360,:945
        box = None
        if bboxes_3d_data is not None:                                                    # :852-896
            bb = bboxes_3d_data["bboxes"]
            nb = bb.shape[1]
            flat = {k: v.reshape(-1, *v.shape[2:]) for k, v in bboxes_3d_data.items()}
            cls = None
            if getattr(self, "use_box_adapter", False):                                   # :873-878
                box, cls = self.bbox_embedder(flat["bboxes"], flat["classes"], flat["masks"], return_cls_emb=True)
                cls = cls.reshape(b, nb, *cls.shape[1:])
            else:
                box = self.bbox_embedder(flat["bboxes"], flat["classes"], flat["masks"])
            box = box.reshape(b, nb, *box.shape[1:])
            if nb != n_cam:
                box = box.expand(-1, n_cam, -1, -1)
                cls = None if cls is None else cls.expand(-1, n_cam, -1, -1)
            box = box.reshape(b * n_cam, *box.shape[2:])
            cls = None if cls is None else cls.reshape(b * n_cam, *cls.shape[2:])
            for m in self.modules():                                                      # :898-900
                if isinstance(getattr(m, "processor", None), AdapterAttnProcessor):
                    m.processor.num_tokens = box.shape[1]
        else:
            cls = None
        t = timestep.reshape(-1)
        emb = self.time_embedding(self.time_proj(t).to(self.dtype))                       # :921-929
        x = sample.reshape(b * n_cam, *sample.shape[2:])
        if len(emb) < len(x):
            emb = emb.repeat_interleave(n_cam, dim=0)                                     # :951-952
        x = self.conv_in(x)                                                               # :965
        cond = controlnet_cond if self.use_occ_3d else self.controlnet_cond_embedding(controlnet_cond)
        if self.use_txt_con_fusion:
            cond = self.txt_con_fusion(cond, ctx[:, 1:])                                  # :973-978 (no cam token)
        x = x + cond                                                                      # :990
        full_ctx = ctx if box is None else torch.cat([ctx, box], dim=1)                   # :1065-1068
        blk_ctx = full_ctx if cls is None else torch.cat([full_ctx, cls], dim=1)          # :1006,:1021 (+ class tokens)
        skips = (x,)
        for blk in self.down_blocks:
            if getattr(blk, "has_cross_attention", False):
                x, res = blk(hidden_states=x, temb=emb, encoder_hidden_states=blk_ctx)
            else:
                x, res = blk(hidden_states=x, temb=emb)
            skips += res
        x = self.mid_block(x, emb, encoder_hidden_states=blk_ctx)
        down = [zc(s) for s, zc in zip(skips, self.controlnet_down_blocks)]              # :1031-1039
        mid = self.controlnet_mid_block(x)
        if guess_mode:                                                                    # :1042-1050
            scales = torch.logspace(-1, 0, len(down) + 1) * conditioning_scale            # 0.1 ... 1.0
            down = [d * sc for d, sc in zip(down, scales)]
            mid = mid * scales[-1]
        else:                                                                             # :1051-1055
            down = [d * conditioning_scale for d in down]
            mid = mid * conditioning_scale
        return down, mid, full_ctx                                                        # :1066-1076


# ------------------------------------------------------------------ sampler loop (A14/15) --
def ddim_alphas(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012):
    """SD-v1.5 scheduler config: scaled-linear betas, alphas_cumprod."""
    betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
    return torch.cumprod(1.0 - betas, dim=0)


def ddim_timesteps(num_inference_steps, num_train_timesteps=1000, steps_offset=1):
    """diffusers DDIMScheduler.set_timesteps ('leading' spacing, steps_offset=1)."""
    ratio = num_train_timesteps // num_inference_steps
    ts = (torch.arange(0, num_inference_steps) * ratio).round().flip(0).to(torch.int64) + steps_offset
    return ts, ratio


def ddim_coefs(alphas_cumprod, t, ratio, set_alpha_to_one=False):
    """{sqrt(a_t), sqrt(1-a_t), sqrt(a_prev), sqrt(1-a_prev)} for DDIMScheduler.step (eta=0)."""
    a_t = alphas_cumprod[t]
    prev = t - ratio
    a_p = alphas_cumprod[prev] if prev >= 0 else (torch.tensor(1.0) if set_alpha_to_one else alphas_cumprod[0])
    return [float(a_t.sqrt()), float((1 - a_t).sqrt()), float(a_p.sqrt()), float((1 - a_p).sqrt())]


def denoise_step(unet, controlnets, latents, t, prompt_embeds, camera_param, bboxes_list, conds,
                 guidance_scale, coef):
    """One iteration of pipeline/pipeline_bev_controlnet.py:381-504 (CFG on, guess_mode off):
    latents (b, n, 4, h, w); prompt_embeds / camera_param / boxes / conds already hold the uncond
    half first (as add_uncond_to_kwargs builds them :349-375)."""
    n_cam = latents.shape[1]
    lat_in = torch.cat([latents] * 2)                                                   # :384-386
    tt = torch.as_tensor(t).reshape(1).repeat(len(lat_in))                               # :392,:403
    down = mid = ctx = None
    for i, cn in enumerate(controlnets):                                                 # :405-431
        d, m, c = cn(lat_in, tt, camera_param, bboxes_list[i], prompt_embeds, conds[i])
        if i == 0:
            down, mid, ctx = d, m, c
        else:
            down = [a + b for a, b in zip(down, d)]
            mid = mid + m
    x = lat_in.reshape(-1, *lat_in.shape[2:])                                            # :470-472
    eps = unet(x, torch.as_tensor(t), encoder_hidden_states=ctx,
               down_block_additional_residuals=down, mid_block_additional_residual=mid).sample   # :476-484
    eu, ec = eps.chunk(2)
    eps = eu + guidance_scale * (ec - eu)                                                # :487-492
    sa_t, s1a_t, sa_p, s1a_p = coef
    flat = latents.reshape(-1, *latents.shape[2:])
    x0 = (flat - s1a_t * eps) / sa_t                                                     # DDIMScheduler.step, eta=0
    prev = sa_p * x0 + s1a_p * eps
    return prev.reshape_as(latents)                                                      # :504
