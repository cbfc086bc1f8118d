"""ControlNet condition embedders — mirror magicdrive/networks/map_embedder.py.

`ControlNetConditioningEmbedding` (:81-138): the ORS panorama (b, 3, 224, 2400) is split into 6
views, then conv 3->16, [c->c, c->c' stride 2] x3 with SiLU after every conv, and a zero-init conv
256->320.  All convs are the implicit-GEMM HIP kernel with SiLU fused in the epilogue; the view
split is folded into the NCHW->NHWC boundary copy.
"""
from typing import Tuple

import torch
import torch.nn as nn

from .. import ops as O
from .layers import Conv3x3, as_nchw_view


class ControlNetConditioningEmbedding(nn.Module):
    def __init__(self, conditioning_embedding_channels: int, conditioning_channels: int = 3,
                 block_out_channels: Tuple[int, ...] = (16, 32, 96, 256), conditioning_size=None):
        super().__init__()
        self.conv_in = Conv3x3(conditioning_channels, block_out_channels[0])
        self.blocks = nn.ModuleList([])
        for i in range(len(block_out_channels) - 1):
            ci, co = block_out_channels[i], block_out_channels[i + 1]
            self.blocks.append(Conv3x3(ci, ci))
            self.blocks.append(Conv3x3(ci, co, stride=2))
        self.conv_out = Conv3x3(block_out_channels[-1], conditioning_embedding_channels)

    def run(self, conditioning, n_views=6):
        """(b, c, h, n_views*w) NCHW panorama -> ((b*n_views*h'*w'), C) NHWC rows, b*n_views, h', w'.
        The reference hard-codes 6 views (map_embedder.py:116-125); a view-sharded rank passes the columns of
        its own views only."""
        b, c, h, pw = conditioning.shape
        nv = int(n_views)
        w = pw // nv
        dt = self.conv_in.weight.dtype
        # view split (map_embedder.py:116-125) + NHWC + channel pad to 8: one layout launch
        x = O.nchw_to_nhwc(conditioning.to(dt), self.conv_in.cin_pad, views=nv)
        m = b * nv
        x = self.conv_in.run(x, m, h, w, epilogue=O.DD_EPI_SILU)
        for blk in self.blocks:
            x = blk.run(x, m, h, w, epilogue=O.DD_EPI_SILU)
            h, w = blk.out_hw(h, w)
        return self.conv_out.run(x, m, h, w), m, h, w

    def forward(self, conditioning):
        x, m, h, w = self.run(conditioning)
        return as_nchw_view(x, m, h, w)


class BEVControlNetConditioningEmbedding(nn.Module):
    """BEV-map embedder of vanilla MagicDrive (map_embedder.py:10-77); the DualDiff ORS branches
    (configs/exp/dual_branch_augloss_fusion.yaml) use ControlNetConditioningEmbedding instead.  The (b, 25, 200, 200) map is
    shared by the 6 views (`repeat(... repeat=6)`, :65): it is embedded ONCE per scene and the result is repeated.

    The reference's asymmetric paddings / strides map onto the pad-1 HIP conv without new arithmetic:
      padding (2, 1)            one extra zero row above and below the NHWC activation, then the pad-1 conv;
      stride 2, padding (2, 1)  the same extra rows, then the stride-2 pad-1 conv (output row i reads input rows 2i-2..2i);
      stride (2, 1)             the stride-1 conv, every second output ROW kept (the same taps in the same order, so the kept
                                rows carry the bits a dedicated kernel would produce; the embedder runs on 54x50 maps).
    200x200 -> 101x100 -> 52x50 -> 54x50 -> 28x50 with the default widths; SiLU after every conv but the last."""

    def __init__(self, conditioning_embedding_channels: int = 320, conditioning_size: Tuple[int, int, int] = (25, 200, 200),
                 block_out_channels: Tuple[int, ...] = (32, 64, 128, 256)):
        super().__init__()
        self.conv_in = Conv3x3(conditioning_size[0], block_out_channels[0])
        self.blocks = nn.ModuleList([])
        self._geom = []                                   # per block: (extra zero rows, keep every k-th output row)
        for i in range(len(block_out_channels) - 2):
            ci, co = block_out_channels[i], block_out_channels[i + 1]
            self.blocks.append(Conv3x3(ci, ci))
            self.blocks.append(Conv3x3(ci, co, stride=2))
            self._geom += [(0, 1), (1, 1)]
        ci, co = block_out_channels[-2], block_out_channels[-1]
        self.blocks.append(Conv3x3(ci, ci))
        self.blocks.append(Conv3x3(ci, co))
        self._geom += [(1, 1), (1, 2)]
        self.conv_out = Conv3x3(co, conditioning_embedding_channels)

    def run(self, conditioning, n_views=6):
        """(b, c, h, w) NCHW map -> ((b * n_views * h' * w'), C) NHWC rows, b * n_views, h', w'."""
        b, c, h, w = conditioning.shape
        dt = self.conv_in.weight.dtype
        xp = conditioning.new_zeros((b, h, w, self.conv_in.cin_pad), dtype=dt)
        xp[..., :c] = conditioning.to(dt).permute(0, 2, 3, 1)
        x = self.conv_in.run(xp.reshape(b * h * w, -1), b, h, w, epilogue=O.DD_EPI_SILU)
        for blk, (rows, keep) in zip(self.blocks, self._geom):
            ch = x.shape[1]
            if rows:
                x = torch.nn.functional.pad(x.reshape(b, h, w, ch), (0, 0, 0, 0, rows, rows)).reshape(-1, ch)
                h += 2 * rows
            x = blk.run(x, b, h, w, epilogue=O.DD_EPI_SILU)
            h, w = blk.out_hw(h, w)
            if keep > 1:
                x = x.reshape(b, h, w, -1)[:, ::keep].contiguous()
                h = x.shape[1]
                x = x.reshape(b * h * w, -1)
        y = self.conv_out.run(x, b, h, w)
        nv = int(n_views)
        y = y.reshape(b, 1, h * w, -1).expand(b, nv, h * w, y.shape[-1]).reshape(b * nv * h * w, -1)
        return y, b * nv, h, w

    def forward(self, conditioning):
        x, m, h, w = self.run(conditioning)
        return as_nchw_view(x, m, h, w)
