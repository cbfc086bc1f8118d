"""MI355X-native denoising hot path for DualDiff-style multi-view diffusion.

Host code is Python on PyTorch-ROCm (device memory, streams, torch.distributed); every
operator on the denoising path is a hand-written gfx950 HIP kernel behind the C-ABI declared
in include/dualdiff_hip.h.  The model classes under
`dualdiff_amd.networks` mirror the reference's `magicdrive.networks` surface so that the
reference's runner / pipeline can load them by dotted path.
"""
__version__ = "0.1.0"
