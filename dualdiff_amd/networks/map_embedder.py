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
        # view split (map_embedder.py:116-125) + NHWC + channel pad to 8 in one boundary copy
        x = conditioning.to(dt).reshape(b, c, h, nv, w).permute(0, 3, 2, 4, 1)      # b, view, h, w, c
        xp = x.new_zeros((b, nv, h, w, self.conv_in.cin_pad))
        xp[..., :c] = x
        x = xp.reshape(b * nv * h * w, self.conv_in.cin_pad)
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
    """BEV-map embedder of vanilla MagicDrive (map_embedder.py:10-77); not used by the DualDiff
    ORS branches (configs/exp/dual_branch_augloss_fusion.yaml) — kept as a named stub so configs
    that reference it fail with a clear message."""

    def __init__(self, *a, **k):
        super().__init__()
        raise NotImplementedError("BEVControlNetConditioningEmbedding is outside the DualDiff hot path "
                                  "(SURVEY.md §8a A11); use ControlNetConditioningEmbedding")
