"""Launches the hot kernels on their real shapes a few times (for rocprofv3 --pmc passes)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dualdiff_amd import ops as O

M = 12
dt = torch.bfloat16
which = sys.argv[1] if len(sys.argv) > 1 else "all"


def r(*shape, s=1.0):
    return (torch.randn(*shape, device="cuda") * s).to(dt)


if which in ("all", "attn"):
    q, k, v = r(M * 1400, 320), r(M * 1400, 320), r(M * 1400, 320)
    for _ in range(3):
        O.attention(q, k, v, M, 1400, 1400, 8, 40)
if which in ("all", "conv"):
    x, w, b = r(M * 1400, 320), r(320, 2880, s=0.02), r(320)
    for tile in (11, 17, 1):
        for _ in range(3):
            O.conv3x3(x, w, b, M, 28, 50, tile=tile, split_k=1)
    x, w, b = r(M * 1400, 960), r(320, 8640, s=0.02), r(320)
    for tile in (11, 12, 16):
        for _ in range(3):
            O.conv3x3(x, w, b, M, 28, 50, tile=tile, split_k=1)
    x, w, b = r(M * 350, 1280), r(640, 11520, s=0.02), r(640)
    for tile, sp in ((18, 1), (11, 1), (11, 3), (17, 1)):
        for _ in range(3):
            O.conv3x3(x, w, b, M, 14, 25, tile=tile, split_k=sp)
if which in ("all", "gemm"):
    a, w, b = r(M * 1400, 1280), r(320, 1280, s=0.02), r(320)
    for _ in range(3):
        O.gemm(a, w, b, tile=17, split_k=1)
    a, w = r(M * 1400, 320), r(320, 320, s=0.05)
    for _ in range(3):
        O.gemm(a, w, b, tile=17, split_k=1)
if which in ("all", "gn"):
    x, g, b = r(M * 1400, 320), r(320), r(320)
    for _ in range(3):
        O.groupnorm(x, g, b, M, 1400, 32, 1e-5, True)
torch.cuda.synchronize()
