"""Return object of `BEVControlNetModel.forward(return_dict=True)`.

The reference hands the sampler an object with three attributes (13 residuals for the UNet skips,
the mid-block residual, and the camera/text/box token sequence the UNet must attend to —
magicdrive/networks/unet_addon_rawbox.py:1078-1082, consumed at pipeline_bev_controlnet.py:405-446).
This container also behaves like the `return_dict=False` tuple (indexing, unpacking, `to_tuple()`),
which is what diffusers' BaseOutput offers the reference's callers.
"""
from typing import Iterator, Sequence

import torch

_FIELDS = ("down_block_res_samples", "mid_block_res_sample", "encoder_hidden_states_with_cam")


class BEVControlNetOutput:
    __slots__ = _FIELDS

    def __init__(self, down_block_res_samples: Sequence[torch.Tensor], mid_block_res_sample: torch.Tensor,
                 encoder_hidden_states_with_cam: torch.Tensor):
        self.down_block_res_samples = tuple(down_block_res_samples)
        self.mid_block_res_sample = mid_block_res_sample
        self.encoder_hidden_states_with_cam = encoder_hidden_states_with_cam

    def to_tuple(self):
        return tuple(getattr(self, f) for f in _FIELDS)

    def __iter__(self) -> Iterator:
        return iter(self.to_tuple())

    def __len__(self):
        return len(_FIELDS)

    def __getitem__(self, key):
        if isinstance(key, str):
            if key not in _FIELDS:
                raise KeyError(key)
            return getattr(self, key)
        return self.to_tuple()[key]

    def __repr__(self):
        return "BEVControlNetOutput(%d residuals, mid %s, tokens %s)" % (
            len(self.down_block_res_samples), tuple(self.mid_block_res_sample.shape),
            tuple(self.encoder_hidden_states_with_cam.shape))
