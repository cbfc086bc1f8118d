"""Semantic Fusion Attention (SFA) — mirrors magicdrive/networks/txt_con_fusion.py.

`txt_con_XFormersAttn` (:18-181): Q = 320-ch ORS condition map (28x50 tokens), K/V = 768-d text
tokens, 8 heads x 40, out-projection + bias, residual add.  `txt_con_XFormersAttn_plus`
(:184-337): occ->text attention followed by an occ self-style attention.  Both follow the
attention-processor call protocol with `attn=None` (unet_addon_rawbox.py:974-978).

K and V are projected by one fused GEMM; the residual is folded into the out-projection epilogue.
HBM/launch-bound (0.79 GFLOP per instance), not MFMA-bound.
"""
import torch
import torch.nn as nn

from .. import ops as O
from .layers import Linear, _Dropout, as_nchw_view, to_nhwc


class _SFABase(nn.Module):
    heads = 8

    def _io(self, hidden_states, encoder_hidden_states):
        if encoder_hidden_states is None:
            raise AssertionError("SFA is a cross-attention: encoder_hidden_states is required")
        if hidden_states.dim() == 4:
            x, b, h, w = to_nhwc(hidden_states)
            shape = (b, h, w)
        else:
            b, l, c = hidden_states.shape
            x, shape = hidden_states.reshape(b * l, c), None
        e = encoder_hidden_states
        return x, b, x.shape[0] // b, e.reshape(-1, e.shape[-1]).contiguous(), e.shape[1], shape

    def _out(self, y, b, shape, like):
        if shape is not None:
            return as_nchw_view(y, *shape)
        return y.reshape(b, -1, y.shape[1])


class txt_con_XFormersAttn(_SFABase):
    def __init__(self, con_dim=320, txt_dim=768, hidden_size=320):
        super().__init__()
        self.inner_dim = self.out_dim = hidden_size
        self.to_q = Linear(con_dim, hidden_size, bias=False)
        self.to_k = Linear(txt_dim, hidden_size, bias=False)
        self.to_v = Linear(txt_dim, hidden_size, bias=False)
        self.to_out = nn.ModuleList([Linear(hidden_size, hidden_size, bias=True), _Dropout()])
        self.scale = (hidden_size // self.heads) ** -0.5
        self.rescale_output_factor = 1.0
        self.residual_connection = True

    def run(self, x, b, lq, e2d, lk):
        c = self.inner_dim
        if "_pk_kv" not in self.__dict__ or self.__dict__["_pk_kv"].dtype != x.dtype \
                or self.__dict__["_pk_kv"].device != x.device:
            self.__dict__["_pk_kv"] = torch.cat([self.to_k.weight.detach(), self.to_v.weight.detach()], 0).contiguous()
        from . import layers
        if layers.XATTN_FUSED and O.xattn320_ok(c, self.heads, lk, x.shape[0]) and x.shape[1] == c:
            # to_q -> attention over the text keys -> to_out + bias + residual in ONE launch (csrc/xattn.hip); K | V of
            # the text tokens as one contiguous [keys][40] block per head (the form the kernel streams fastest)
            hd = self.heads
            kvh = O.gemm(e2d, self.__dict__["_pk_kv"], head_major=(c // hd, 0, 1.0))        # (16, b * lk, 40)
            return O.xattn320(x, self.to_q.wx, self.to_out[0].wx, self.to_out[0].bias, kvh[:hd], kvh[hd:], b, lq, lk,
                              self.scale, res=x if self.residual_connection else None)
        kv = O.gemm(e2d, self.__dict__["_pk_kv"])
        q = self.to_q.run(x)
        o = O.attention(q, kv[:, :c], kv[:, c:], b, lq, lk, self.heads, c // self.heads, self.scale)
        return self.to_out[0].run(o, res=x if self.residual_connection else None)

    def _apply(self, fn, *a, **k):
        self.__dict__.pop("_pk_kv", None)
        return super()._apply(fn, *a, **k)

    def _load_from_state_dict(self, *a, **k):
        self.__dict__.pop("_pk_kv", None)
        return super()._load_from_state_dict(*a, **k)

    def forward(self, attn=None, hidden_states=None, encoder_hidden_states=None, attention_mask=None, temb=None):
        assert attention_mask is None and temb is None
        x, b, lq, e2d, lk, shape = self._io(hidden_states, encoder_hidden_states)
        return self._out(self.run(x, b, lq, e2d, lk), b, shape, hidden_states)


class txt_con_XFormersAttn_plus(_SFABase):
    def __init__(self, con_dim=320, txt_dim=768, hidden_size=320):
        super().__init__()
        self.inner_dim = self.out_dim = hidden_size
        self.to_q_occ = Linear(con_dim, hidden_size, bias=False)
        self.to_k_occ = Linear(con_dim, hidden_size, bias=False)
        self.to_v_occ = Linear(con_dim, hidden_size, bias=False)
        self.to_k_txt = Linear(txt_dim, hidden_size, bias=False)
        self.to_v_txt = Linear(txt_dim, hidden_size, bias=False)
        self.to_out = nn.ModuleList([Linear(hidden_size, hidden_size, bias=True), _Dropout()])
        self.scale = (hidden_size // self.heads) ** -0.5
        self.rescale_output_factor = 1.0
        self.residual_connection = True

    def run(self, x, b, lq, e2d, lk):
        d = self.inner_dim // self.heads
        q, ko, vo = self.to_q_occ.run(x), self.to_k_occ.run(x), self.to_v_occ.run(x)
        kt, vt = self.to_k_txt.run(e2d), self.to_v_txt.run(e2d)
        q1 = O.attention(q, kt, vt, b, lq, lk, self.heads, d, self.scale)      # :313-315
        o = O.attention(q1, ko, vo, b, lq, lq, self.heads, d, self.scale)      # :316-318
        return self.to_out[0].run(o, res=x if self.residual_connection else None)

    def forward(self, attn=None, hidden_states=None, encoder_hidden_states=None, attention_mask=None, temb=None):
        assert attention_mask is None, "we don't use attention mask here"
        x, b, lq, e2d, lk, shape = self._io(hidden_states, encoder_hidden_states)
        return self._out(self.run(x, b, lq, e2d, lk), b, shape, hidden_states)
