"""Output container of BEVControlNetModel.forward — mirrors magicdrive/networks/output_cls.py."""
from dataclasses import dataclass
from typing import Tuple

import torch


@dataclass
class BEVControlNetOutput:
    down_block_res_samples: Tuple[torch.Tensor]
    mid_block_res_sample: torch.Tensor
    encoder_hidden_states_with_cam: torch.Tensor

    def __getitem__(self, i):
        return (self.down_block_res_samples, self.mid_block_res_sample, self.encoder_hidden_states_with_cam)[i]
