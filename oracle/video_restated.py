"""CPU definition of the VIDEO transformer block (ST-Attn + temporal attention) — TEST INFRASTRUCTURE ONLY.

EXTENSION WITHOUT REFERENCE SEMANTICS.  The released reference contains no video code: ST-Attn / temporal
attention exist only as prose and as four boxes in a figure (README.md:47-49, media/framework.jpg "Video
Transformer Block": ST-Attn -> Cross-Attn -> Cross View Attn -> Temporal Attn).  BASELINE.json configs[3] asks
for them, so this file DEFINES what the build implements; there is nothing to pin it to ("parity unpinned" by
construction) and it is labelled as an extension everywhere it is reported.

Definition (instances ordered scene-major, then FRAME, then view: i = (b * T + t) * n_cam + v):

  1. ST-Attn (sparse spatio-temporal self-attention, the Tune-A-Video form the MagicDrive video branch uses;
     weights = attn1 / norm1, so an image checkpoint loads unchanged):
         h += to_out( Attn( Q = Wq LN1(h[t]),  K,V = Wk,v LN1([h[0] ; h[max(t-1, 0)]]) ) )   per view,
     i.e. every frame attends to the FIRST frame and to its PREVIOUS frame (2 n keys).
  2. text / box cross-attention (attn2)                         — as in the image block
  3. cross-view attention (attn4 + connector), inside a frame  — as in the image block (blocks.py:190-222)
  4. temporal attention (new weights norm_temp / attn_temp, `attn_temp.to_out` zero-initialised):
         h[:, p] += to_out( SelfAttn over the T frames of token position p ( LN_temp(h)[:, p] ) )   per view
  5. GEGLU feed-forward                                          — as in the image block
"""
import torch
import torch.nn as nn

from . import diffusers_restated as D
from . import dualdiff_restated as R
from .dualdiff_restated import sdpa
from .numerics import stor


class VideoMultiviewTransformerBlock(R.BasicMultiviewTransformerBlock):
    def __init__(self, *args, n_frames=1, **kw):
        super().__init__(*args, **kw)
        dim, heads, hd = self._args["dim"], self._args["num_attention_heads"], self._args["attention_head_dim"]
        self.n_frames = n_frames
        self.norm_temp = nn.LayerNorm(dim)
        self.attn_temp = D.Attention(query_dim=dim, heads=heads, dim_head=hd)
        D.zero_module(self.attn_temp.to_out[0])

    def forward(self, hidden_states, attention_mask=None, encoder_hidden_states=None,
                encoder_attention_mask=None, timestep=None, cross_attention_kwargs=None, class_labels=None):
        h = hidden_states
        n_cam, t_n = len(self.neighboring_view_pair), self.n_frames
        m, n, c = h.shape
        b = m // (t_n * n_cam)
        # 1. ST-Attn
        x = self.norm1(h).reshape(b, t_n, n_cam, n, c)
        a = self.attn1
        q = a.to_q(x)
        prev = torch.tensor([max(t - 1, 0) for t in range(t_n)])
        src = torch.cat([x[:, :1].expand(-1, t_n, -1, -1, -1), x[:, prev]], dim=3)        # (b, T, v, 2n, C)
        k, v = a.to_k(src), a.to_v(src)
        o = sdpa(q.reshape(m, n, c), k.reshape(m, 2 * n, c), v.reshape(m, 2 * n, c), a.heads, a.scale)
        h = stor(a.to_out[0](o) + h)
        # 2. cross-attention
        h = stor(self.attn2(self.norm2(h), encoder_hidden_states=encoder_hidden_states) + h)
        # 3. cross-view attention within each frame
        x = self.norm4(h)
        xv = x.reshape(-1, n_cam, n, c)
        a = self.attn4
        q, k, v = a.to_q(xv), a.to_k(xv), a.to_v(xv)
        out = torch.zeros_like(xv)
        for view, neighbours in self.neighboring_view_pair.items():
            for u in neighbours:
                out[:, view] = stor(out[:, view] + a.to_out[0](sdpa(q[:, view], k[:, u], v[:, u], a.heads, a.scale)))
        h = stor(self.connector(out.reshape_as(x)) + h)
        # 4. temporal attention: sequences of T frames per (scene, view, token position)
        a = self.attn_temp
        x = self.norm_temp(h).reshape(b, t_n, n_cam * n, c).permute(0, 2, 1, 3).reshape(b * n_cam * n, t_n, c)
        o = a.to_out[0](sdpa(a.to_q(x), a.to_k(x), a.to_v(x), a.heads, a.scale))
        h = stor(o.reshape(b, n_cam * n, t_n, c).permute(0, 2, 1, 3).reshape(m, n, c) + h)
        # 5. feed-forward
        return stor(self.ff(self.norm3(h)) + h)


class UNet2DConditionModelMultiviewVideo(D.UNet2DConditionModel):
    """SD-v1.5 UNet with every transformer block replaced by the video block; the resnets / samplers / ControlNet
    residual adds are per-instance as in the image model (frames are just more instances)."""

    def __init__(self, neighboring_view_pair=None, n_frames=1, **kw):
        super().__init__(**kw)
        for name, mod in list(self.named_modules()):
            if type(mod) is D.BasicTransformerBlock:
                parent = self
                *path, leaf = name.split(".")
                for p in path:
                    parent = getattr(parent, p)
                setattr(parent, leaf, VideoMultiviewTransformerBlock(
                    **mod._args, neighboring_view_pair=neighboring_view_pair, n_frames=n_frames))
