"""`ContinuousBBoxWithTextEmbedding` — mirrors magicdrive/networks/bbox_embedder.py:28-203.

Fourier features of the 8 box corners -> Linear -> SiLU -> concat class token -> 3-layer MLP,
masked boxes replaced by learned null features.  The Linear layers run on the HIP GEMM kernel
(SiLU fused into the epilogue); the Fourier features / masking are small device tensor ops.
Step-invariant: the pipeline harness may evaluate it once per sample."""
import logging

import torch
import torch.nn as nn

from .. import ops as O
from .embedder import get_embedder
from .layers import Linear

FUSED_BOX_TOKENS = __import__("os").environ.get("DD_FUSED_TOKENS", "1") != "0"      # one launch for both MLP operands (ops.box_tokens); False = the tensor-op chain
XYZ_MIN = [-200, -300, -20]
XYZ_RANGE = [350, 650, 80]


class _MLP(nn.Module):
    """Sequential(Linear, SiLU, Linear, SiLU, Linear) with diffusers-style indices 0 / 2 / 4."""

    def __init__(self, dims):
        super().__init__()
        self.add_module("0", Linear(dims[0], dims[1]))
        self.add_module("2", Linear(dims[1], dims[2]))
        self.add_module("4", Linear(dims[2], dims[3]))

    def run(self, x):
        m = self._modules
        x = m["0"].run(x, epilogue=O.DD_EPI_SILU)
        x = m["2"].run(x, epilogue=O.DD_EPI_SILU)
        return m["4"].run(x)


class ContinuousBBoxWithTextEmbedding(nn.Module):
    def __init__(self, n_classes, class_token_dim=768, trainable_class_token=False, embedder_num_freq=4,
                 proj_dims=(768, 512, 512, 768), mode="cxyz", minmax_normalize=True,
                 use_text_encoder_init=True, **kwargs):
        super().__init__()
        self.mode = mode
        if mode == "cxyz":
            output_num = 4
        elif mode == "all-xyz":
            output_num = 8
        else:
            raise NotImplementedError(f"Wrong mode {mode}")
        self.minmax_normalize = minmax_normalize
        self.use_text_encoder_init = use_text_encoder_init
        self.fourier_embedder = get_embedder(3, embedder_num_freq)
        fdim = self.fourier_embedder.out_dim * output_num
        self.bbox_proj = Linear(fdim, proj_dims[0])
        self.second_linear = _MLP([proj_dims[0] + class_token_dim, proj_dims[1], proj_dims[2], proj_dims[3]])
        tokens = torch.randn(n_classes, class_token_dim)
        if trainable_class_token:
            self.register_parameter("_class_tokens", nn.Parameter(tokens))
        else:
            self.register_buffer("_class_tokens", tokens)
        self.null_class_feature = nn.Parameter(torch.zeros([class_token_dim]))
        self.null_pos_feature = nn.Parameter(torch.zeros([fdim]))

    @property
    def class_tokens(self):
        return self._class_tokens

    def prepare(self, cfg, **kwargs):
        if self.use_text_encoder_init:
            self.set_category_token(kwargs["tokenizer"], kwargs["text_encoder"], cfg.dataset.object_classes)

    def reinitialize(self):
        """40-point map-vector variant (bbox_embedder.py:122-130)."""
        logging.info("[ContinuousBBoxWithTextEmbedding] Reinitialize 40pts ")
        fdim = self.fourier_embedder.out_dim * 40
        like = self.bbox_proj.weight
        self.bbox_proj = Linear(fdim, self.bbox_proj.out_features).to(device=like.device, dtype=like.dtype)
        self.null_pos_feature = nn.Parameter(torch.zeros([fdim], device=like.device, dtype=like.dtype))

    @torch.no_grad()
    def set_category_token(self, tokenizer, text_encoder, class_names):
        device = self.class_tokens.device
        for idx, name in enumerate(class_names):
            ids = tokenizer([name], padding="do_not_pad", return_tensors="pt").input_ids.to(device)
            self.class_tokens[idx].copy_(text_encoder(ids).pooler_output[0])

    def forward_feature(self, pos_emb, cls_emb):
        emb = self.bbox_proj.run(pos_emb.contiguous(), epilogue=O.DD_EPI_SILU)
        return self.second_linear.run(torch.cat([emb, cls_emb], dim=-1))

    def add_n_uncond_tokens(self, hidden_states, token_num):
        b = hidden_states.shape[0]
        tok = self.forward_feature(self.null_pos_feature[None], self.null_class_feature[None])
        return torch.cat([hidden_states, tok[None].expand(b, token_num, -1)], dim=1)

    def forward(self, bboxes, classes, masks=None, return_cls_emb=False, **kwargs):
        b, n = classes.shape
        dt = self.null_pos_feature.dtype
        pts = bboxes.reshape(b * n, *bboxes.shape[2:])
        fe = self.fourier_embedder
        if FUSED_BOX_TOKENS and pts.is_cuda and pts.dim() == 3 and pts.shape[-1] == 3 and len(fe.freq_bands) <= 16 and dt in (torch.float16, torch.bfloat16) \
                and pts.dtype in (torch.float16, torch.bfloat16, torch.float32) and self.class_tokens.dtype == dt:
            # GPU path: ONE launch writes both operands of the MLP — the (masked) Fourier features and the (masked) class
            # token in the right half of the concat buffer; bbox_proj's SiLU epilogue fills the left half in place.
            # Bit-identical to the tensor-op chain below (a 0 / 1 mask blend of finite values is a select).
            rows, pd = b * n, self.bbox_proj.out_features
            ctd = self.class_tokens.shape[1]
            pos = torch.empty((rows, self.bbox_proj.in_features), dtype=dt, device=pts.device)
            cat = torch.empty((rows, pd + ctd), dtype=dt, device=pts.device)
            cls = torch.empty((rows, ctd), dtype=dt, device=pts.device) if return_cls_emb else None
            O.box_tokens(pts, classes.reshape(-1), None if masks is None else masks.reshape(-1), self.class_tokens,
                         self.null_pos_feature, self.null_class_feature, fe.freq_bands, fe.include_input, pos, cat, pd,
                         cls_out=cls, normalize=(XYZ_MIN, XYZ_RANGE) if self.minmax_normalize else None)
            self.bbox_proj.run(pos, epilogue=O.DD_EPI_SILU, out=cat[:, :pd])
            emb = self.second_linear.run(cat).reshape(b, n, -1)
            if return_cls_emb:
                return emb, cls.reshape(b, n, -1)
            return emb
        if masks is None:
            masks = torch.ones(len(pts), device=pts.device)
        masks = masks.reshape(-1).unsqueeze(-1).to(dt)
        if self.minmax_normalize:
            mins = torch.as_tensor(XYZ_MIN, dtype=pts.dtype, device=pts.device)[None, None]
            div = torch.as_tensor(XYZ_RANGE, dtype=pts.dtype, device=pts.device)[None, None]
            pts = (pts - mins) / div
        pos = self.fourier_embedder(pts).reshape(b * n, -1).to(dt)
        pos = pos * masks + self.null_pos_feature[None] * (1 - masks)
        cls = self.class_tokens[classes.reshape(-1)].to(dt)
        cls = cls * masks + self.null_class_feature[None] * (1 - masks)
        emb = self.forward_feature(pos, cls).reshape(b, n, -1)
        if return_cls_emb:
            return emb, cls.reshape(b, n, -1)
        return emb
