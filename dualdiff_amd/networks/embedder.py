"""NeRF-style Fourier feature embedder — mirrors magicdrive/networks/embedder.py:18-67.
Tiny token preparation: one dd_fourier_embed launch on the GPU (the sin / cos / cat
chain of the reference is ~20 launches per call on the sampler's critical stream); plain tensor ops for
host tensors."""
import torch

from .. import ops as O


class Embedder:
    def __init__(self, input_dims, num_freqs, include_input=True, log_sampling=True, **unused):
        self.input_dims, self.num_freqs, self.include_input = input_dims, num_freqs, include_input
        max_freq = num_freqs - 1
        if log_sampling:
            self.freq_bands = [2.0 ** (max_freq * i / max(num_freqs - 1, 1)) for i in range(num_freqs)]
        else:
            lo, hi = 1.0, 2.0 ** max_freq
            self.freq_bands = [lo + (hi - lo) * i / max(num_freqs - 1, 1) for i in range(num_freqs)]
        self.out_dim = input_dims * ((1 if include_input else 0) + 2 * num_freqs)

    def __call__(self, inputs):
        if inputs.is_cuda and len(self.freq_bands) <= 16 and inputs.dtype in (torch.float16, torch.bfloat16, torch.float32):
            return O.fourier_embed(inputs, self.freq_bands, self.include_input)
        outs = [inputs] if self.include_input else []
        for f in self.freq_bands:
            outs += [torch.sin(inputs * f), torch.cos(inputs * f)]
        return torch.cat(outs, dim=-1)


def get_embedder(input_dims, num_freqs, include_input=True, log_sampling=True):
    return Embedder(input_dims=input_dims, num_freqs=num_freqs, include_input=include_input,
                    log_sampling=log_sampling)
