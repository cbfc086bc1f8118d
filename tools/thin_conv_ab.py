"""The condition embedder's conv chain (map_embedder.py:79-113, 12 view-instances of 224x400): per-layer hot time with
dd_conv3x3_thin (DD_THIN_CONV=1, default) or the implicit-GEMM families (DD_THIN_CONV=0); run once per setting."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dualdiff_amd import ops as O
from tools.attn_variants import graph_time
dt = torch.float16
print("DD_THIN_CONV =", os.environ.get("DD_THIN_CONV", "1"))
m, h, w = 12, 224, 400
tot = 0.0
for (ci, co, st) in ((8, 16, 1), (16, 16, 1), (16, 32, 2), (32, 32, 1), (32, 96, 2), (96, 96, 1), (96, 256, 2), (256, 320, 1)):
    x = torch.randn(m * h * w, ci, device="cuda").to(dt)
    wt = (torch.randn(co, 9 * ci, device="cuda") * (9 * ci) ** -0.5).to(dt)
    b = torch.randn(co, device="cuda").to(dt)
    t = graph_time(lambda: O.conv3x3(x, wt, b, m, h, w, stride=st, epilogue=O.DD_EPI_SILU))
    ho, wo = (h - 1) // st + 1, (w - 1) // st + 1
    nbytes = 2.0 * (m * h * w * ci + m * ho * wo * co)
    print("  %3d -> %3d stride %d on %dx%d: %6.1f us  %5.2f TB/s of algorithmic bytes" % (ci, co, st, h, w, t, nbytes / t / 1e6))
    tot += t
    h, w = ho, wo
print("  chain: %.1f us" % tot)
