"""Attention-processor plugins — mirror magicdrive/networks/box_adapter.py.

`XFormersAttnProcessor` (:17-175) is the processor the reference installs on every attention
layer; here it is the same call protocol in front of the fused HIP attention path (optional batch
chunking via the SPLIT_SIZE environment variable, :11,41-64, is honoured).  The IP-adapter style
`Adapter_XFormersAttnProcessor` (:177-411) runs on the same GEMM / attention kernels.
"""
import math
import os

import torch

from .. import ops as O
from .layers import HIPAttnProcessor, Linear

SPLIT_SIZE = int(os.getenv("SPLIT_SIZE", -1))


class XFormersAttnProcessor(HIPAttnProcessor):
    def __init__(self, attention_op=None):
        self.attention_op = attention_op

    @property
    def chunked(self):
        """True when SPLIT_SIZE asks for batch chunking: the transformer blocks then call the processor
        through the diffusers protocol (so that __call__ can chunk) instead of their fused fast path."""
        return SPLIT_SIZE != -1

    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, temb=None, **kw):
        n = hidden_states.shape[0]
        if SPLIT_SIZE != -1 and n > SPLIT_SIZE:
            steps = min(math.ceil(n / SPLIT_SIZE), n)
            assert attention_mask is None and temb is None
            hs = hidden_states.chunk(steps)
            es = [None] * steps if encoder_hidden_states is None else encoder_hidden_states.chunk(steps)
            return torch.cat([self._real_call(attn, h.contiguous(), None if e is None else e.contiguous())
                              for h, e in zip(hs, es)], dim=0)
        return self._real_call(attn, hidden_states, encoder_hidden_states, attention_mask, temb, **kw)

    def _real_call(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, temb=None, **kw):
        return HIPAttnProcessor.__call__(self, attn, hidden_states, encoder_hidden_states, attention_mask, temb, **kw)


class Adapter_XFormersAttnProcessor(torch.nn.Module):
    """Box / class adapter on a text cross-attention (box_adapter.py:177-411), on the HIP kernels.

    The context is [cam + text | box tokens | class tokens] (`num_tokens` each for the last two,
    set by the ControlNet per call, unet_addon_rawbox.py:898-900).  Text tokens feed the layer's own
    K/V; the box tokens get their own K/V projections, each enriched by an attention over the class
    tokens' K/V (:349-357); a second attention of the SAME queries over them is added with `scale`
    (:380-388) before the out-projection.  Everything is GEMM + flash-attention launches; the
    `box + Attn(box, cls)` sums and (for scale == 1) the final sum ride on the attention kernel's
    `accumulate` epilogue."""

    def __init__(self, hidden_size, cross_attention_dim=None, scale=1.0, num_tokens=200):
        super().__init__()
        self.attention_op = None
        self.hidden_size, self.cross_attention_dim = hidden_size, cross_attention_dim
        self.scale, self.num_tokens = scale, num_tokens
        d = cross_attention_dim or hidden_size
        self.to_k_box = Linear(d, hidden_size, bias=False)
        self.to_v_box = Linear(d, hidden_size, bias=False)
        self.to_k_cls = Linear(d, hidden_size, bias=False)
        self.to_v_cls = Linear(d, hidden_size, bias=False)

    def _fused(self, a, b):
        key = "_pk_" + a + b
        if key not in self.__dict__:
            self.__dict__[key] = torch.cat([getattr(self, a).weight.detach(), getattr(self, b).weight.detach()],
                                           dim=0).contiguous()
        return self.__dict__[key]

    def _apply(self, fn, *a, **k):
        for key in [k_ for k_ in self.__dict__ if k_.startswith("_pk_")]:
            del self.__dict__[key]
        return super()._apply(fn, *a, **k)

    def _load_from_state_dict(self, *a, **k):
        for key in [k_ for k_ in self.__dict__ if k_.startswith("_pk_")]:
            del self.__dict__[key]
        return super()._load_from_state_dict(*a, **k)

    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, temb=None, **kw):
        n = hidden_states.shape[0]
        if SPLIT_SIZE != -1 and n > SPLIT_SIZE:                       # :201-225
            steps = min(math.ceil(n / SPLIT_SIZE), n)
            assert attention_mask is None and temb is None
            hs = hidden_states.chunk(steps)
            es = encoder_hidden_states.chunk(steps)
            return torch.cat([self._real_call(attn, h.contiguous(), e.contiguous()) for h, e in zip(hs, es)], dim=0)
        return self._real_call(attn, hidden_states, encoder_hidden_states, attention_mask, temb)

    def _real_call(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, temb=None):
        if attention_mask is not None or encoder_hidden_states is None or isinstance(encoder_hidden_states, list):
            raise NotImplementedError("adapter attention: text cross-attention without mask / box-token list "
                                      "(box_adapter.py:268-272 raises for the list form too)")
        b, lq, c = hidden_states.shape
        nt = int(self.num_tokens)
        e = encoder_hidden_states
        lt = e.shape[1] - 2 * nt
        if lt <= 0 or nt <= 0:
            raise ValueError("context of %d tokens cannot hold 2 x %d adapter tokens" % (e.shape[1], nt))
        kdim = e.shape[2]
        txt = e[:, :lt].reshape(b * lt, kdim) if lt == e.shape[1] else e[:, :lt].contiguous().reshape(b * lt, kdim)
        box = e[:, lt:lt + nt].contiguous().reshape(b * nt, kdim)
        cls = e[:, lt + nt:].contiguous().reshape(b * nt, kdim)
        h, d, sc = attn.heads, attn.dim_head, attn.scale
        ci = attn.inner_dim
        q = attn.to_q.run(hidden_states.reshape(b * lq, c))
        kv = attn.project_kv(txt)
        o = O.attention(q, kv[:, :ci], kv[:, ci:], b, lq, lt, h, d, sc)                       # :339-341
        bkv = O.gemm(box, self._fused("to_k_box", "to_v_box"))                                 # :294-296
        ckv = O.gemm(cls, self._fused("to_k_cls", "to_v_cls"))                                 # :297-298
        ck, cv = ckv[:, :ci], ckv[:, ci:]
        bkv2 = bkv.clone()                       # box_key + Attn(box_key, cls), box_val + Attn(box_val, cls)
        O.attention(bkv[:, :ci], ck, cv, b, nt, nt, h, d, sc, out=bkv2[:, :ci], accumulate=True)   # :349-353
        O.attention(bkv[:, ci:], ck, cv, b, nt, nt, h, d, sc, out=bkv2[:, ci:], accumulate=True)   # :354-357
        if self.scale == 1.0:                                                                      # :380-388
            O.attention(q, bkv2[:, :ci], bkv2[:, ci:], b, lq, nt, h, d, sc, out=o, accumulate=True)
        else:
            bo = O.attention(q, bkv2[:, :ci], bkv2[:, ci:], b, lq, nt, h, d, sc)
            o = O.add(o, O.scale(bo, float(self.scale)))
        y = attn.to_out[0].run(o)                                                                   # :391
        return y.reshape(b, lq, -1)


def box_adapter(net, use_box_token=False):
    """Installs the adapter on every text cross-attention of `net`, initialised from that layer's own
    to_k / to_v (box_adapter.py:414-444); attn1 / attn4 keep the plain processor.  use_box_token (:441-442): the
    adapter projections take 128-wide boxworld feature tokens and keep nn.Linear's default initialisation, as in the
    reference — whose processor then raises NotImplementedError for the token-list context (:269-270), as this one does."""
    sd = net.state_dict()
    procs = {}
    for name in list(net.attn_processors.keys()):
        if name.endswith("attn1.processor") or name.endswith("attn4.processor"):
            procs[name] = XFormersAttnProcessor()
            continue
        layer = name[: -len(".processor")]
        wk, wv = sd[layer + ".to_k.weight"], sd[layer + ".to_v.weight"]
        if use_box_token:
            p = Adapter_XFormersAttnProcessor(hidden_size=wk.shape[0], cross_attention_dim=128)
            for lin in (p.to_k_box, p.to_v_box, p.to_k_cls, p.to_v_cls):
                torch.nn.init.kaiming_uniform_(lin.weight, a=math.sqrt(5))          # nn.Linear.reset_parameters
            procs[name] = p.to(device=wk.device, dtype=wk.dtype)
            continue
        p = Adapter_XFormersAttnProcessor(hidden_size=wk.shape[0], cross_attention_dim=wk.shape[1])
        p.load_state_dict({"to_k_box.weight": wk, "to_v_box.weight": wv, "to_k_cls.weight": wk, "to_v_cls.weight": wv})
        procs[name] = p.to(device=wk.device, dtype=wk.dtype)
    net.set_attn_processor(procs)
    return net
