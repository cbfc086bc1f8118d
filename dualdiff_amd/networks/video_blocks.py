"""VIDEO transformer block: ST-Attn + temporal attention around the multiview block — an EXTENSION.

The released reference has no video code (ST-Attn / temporal attention appear only in README.md:47-49 and in
media/framework.jpg, "Video Transformer Block": ST-Attn -> Cross-Attn -> Cross View Attn -> Temporal Attn);
BASELINE.json configs[3] asks for it, so the semantics are DEFINED by this build (oracle/video_restated.py is
the definition) and every report labels it "extension, no reference semantics".

Instances are ordered scene-major, then FRAME, then view: i = (b * T + t) * n_cam + v, so a frame is just more
instances for every per-instance layer (resnets, ControlNet, cross-attention) and the cross-view attention,
the view split and its halo exchange work unchanged (each (scene, frame) is one more batch entry of a view).

  ST-Attn      attn1 / norm1 weights (an image checkpoint loads unchanged): each frame's queries attend to the
               K/V of the FIRST frame and of its PREVIOUS frame (2n keys).  K/V are projected ONCE per frame by
               the fused QKV GEMM; the two source frames' head-major K/V are gathered into one
               (instance, [K heads | V heads], 2n, d) buffer (the attention kernel's batch-major operand form).
  temporal     new weights norm_temp / attn_temp (to_out zero-initialised): self-attention over the T frames
               of every token position — the flash kernel walks the frame axis IN PLACE through
               `seq_strides` (row stride = tokens of a frame x C, batch stride = C): no transpose.
"""
import torch

from .. import ops as O
from .blocks import BasicMultiviewTransformerBlock
from .layers import Attention, LayerNorm


class VideoMultiviewTransformerBlock(BasicMultiviewTransformerBlock):
    def __init__(self, dim, num_attention_heads, attention_head_dim, cross_attention_dim=None, n_frames=1, **kw):
        super().__init__(dim, num_attention_heads, attention_head_dim, cross_attention_dim=cross_attention_dim, **kw)
        self.n_frames = int(n_frames)
        self.frame_shard = None          # parallel.FrameShard when the T frames are split over ranks
        self.norm_temp = LayerNorm(dim)
        self.attn_temp = Attention(dim, None, num_attention_heads, attention_head_dim)

    @property
    def new_module(self):
        d = dict(super().new_module)
        d.update({"norm_temp": self.norm_temp, "attn_temp": self.attn_temp})
        return d

    def _views(self):
        return self.view_shard.n_local if self.view_shard is not None else self.n_cam

    def _frames(self):
        """Frames held by this rank: all T, or the local range of a frame split (parallel.FrameShard)."""
        return self.frame_shard.n_local if self.frame_shard is not None else self.n_frames

    def _st_attn(self, h, batch, l):
        a = self.attn1
        hd, d, t_n, v_n = a.heads, a.dim_head, self._frames(), self._views()
        if batch % (t_n * v_n):
            raise ValueError("%d instances are not scenes x %d frames x %d views" % (batch, t_n, v_n))
        nb = batch // (t_n * v_n)
        qkv = a.project_qkv(h, self.norm1, head_major=True)                         # (3 hd, batch * l, d)
        kv = qkv[hd:].reshape(2 * hd, nb, t_n, v_n, l, d)
        if self.frame_shard is None:
            first = prev0 = kv[:, :, 0]                                             # frame 0's previous frame is frame 0
        else:       # frame split: frame 0 lives on rank 0, the frame before my first one on the rank before me
            first, prev0 = self.frame_shard.exchange.st_sources(kv[:, :, 0].contiguous(), kv[:, :, t_n - 1].contiguous())
        # K/V of [first frame ; previous frame] per instance: (nb, T, V, 2 hd, 2, l, d)
        kv2 = torch.empty((nb, t_n, v_n, 2 * hd, 2, l, d), dtype=h.dtype, device=h.device)
        kv2[:, :, :, :, 0] = first.unsqueeze(2).permute(1, 2, 3, 0, 4, 5)           # broadcast over the frames
        kv2[:, 0, :, :, 1] = prev0.permute(1, 2, 0, 3, 4)
        if t_n > 1:
            kv2[:, 1:, :, :, 1] = kv[:, :, :t_n - 1].permute(1, 2, 3, 0, 4, 5)
        flat = kv2.reshape(batch, 2 * hd, 2 * l, d)
        o = O.attention(qkv[:hd], flat[:, :hd], flat[:, hd:], batch, l, 2 * l, hd, d, q_prescaled=True)
        return a.to_out[0].run(o, res=h)

    def _temporal(self, h, batch, l):
        a = self.attn_temp
        hd, d, c, t_n, v_n = a.heads, a.dim_head, a.inner_dim, self._frames(), self._views()
        nb = batch // (t_n * v_n)
        qkv = a.project_qkv(h, self.norm_temp)                                      # (batch * l, 3 C) row-major
        o = torch.empty((batch * l, c), dtype=h.dtype, device=h.device)
        per = t_n * v_n * l                                                         # rows of one scene
        if self.frame_shard is None:
            for bi in range(nb):                                                    # sequences of T rows, V*l apart
                rows = slice(bi * per, (bi + 1) * per)
                O.attention(qkv[rows, :c], qkv[rows, c:2 * c], qkv[rows, 2 * c:], v_n * l, t_n, t_n, hd, d, a.scale,
                            out=o[rows], seq_strides=(v_n * l * 3 * c, 3 * c), out_seq_strides=(v_n * l * c, c))
            return a.to_out[0].run(o, res=h)
        # frame split: local queries against the K|V rows of ALL frames, gathered frame-major (T, nb, V*l, 2C)
        kv_loc = qkv.view(nb, t_n, v_n * l, 3 * c)[..., c:].permute(1, 0, 2, 3).contiguous()        # one copy
        kv_all = self.frame_shard.exchange.gather_frames(kv_loc)
        t_all = kv_all.shape[0]
        if t_all != self.n_frames:
            raise ValueError("gathered %d frames, the block is configured for %d" % (t_all, self.n_frames))
        fs = nb * v_n * l * 2 * c                                                   # frame pitch of kv_all
        for bi in range(nb):
            rows = slice(bi * per, (bi + 1) * per)
            O.attention(qkv[rows, :c], kv_all[0, bi, :, :c], kv_all[0, bi, :, c:], v_n * l, t_n, t_all, hd, d, a.scale,
                        out=o[rows], seq_strides=(v_n * l * 3 * c, 3 * c), out_seq_strides=(v_n * l * c, c),
                        kv_seq_strides=(fs, 2 * c))
        return a.to_out[0].run(o, res=h)

    def run(self, h, batch, l, ctx2d, lc, defer_ff_out=False):
        h = self._st_attn(h, batch, l)
        h = self._attn(self.attn2, self.norm2, h, batch, l, ctx2d, lc)
        h = self._cross_view(h, batch, l)
        h = self._temporal(h, batch, l)
        if defer_ff_out:
            return self.ff.run(h, norm=self.norm3, defer_out=True), h
        return self.ff.run(h, res=h, norm=self.norm3)
