"""Attention-processor plugins — mirror magicdrive/networks/box_adapter.py.

`XFormersAttnProcessor` (:17-175) is the processor the reference installs on every attention
layer; here it is the same call protocol in front of the fused HIP attention path (optional batch
chunking via the SPLIT_SIZE environment variable, :11,41-64, is honoured).  The IP-adapter style
`Adapter_XFormersAttnProcessor` (:177-411) is a "next" row of the scope table.
"""
import math
import os

import torch

from .layers import HIPAttnProcessor

SPLIT_SIZE = int(os.getenv("SPLIT_SIZE", -1))


class XFormersAttnProcessor(HIPAttnProcessor):
    def __init__(self, attention_op=None):
        self.attention_op = attention_op

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
    def __init__(self, *a, **k):
        super().__init__()
        raise NotImplementedError("box-adapter attention (use_box_adapter) is a 'next' row of the scope "
                                  "table (SURVEY.md §8f N1); the shipped dual-branch config does not use it")


def box_adapter(net, use_box_token=False):
    raise NotImplementedError("box_adapter(): see Adapter_XFormersAttnProcessor")
