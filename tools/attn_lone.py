"""Lone-workgroup latency of the attention kernels: one q-block per (batch, head), <= 1 workgroup per CU."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dualdiff_amd import ops as O
from tools.attn_variants import graph_time  # noqa
dt = torch.bfloat16
for (b, lq, lk, h, d, vs) in ((32, 128, 1400, 8, 40, (0, 5, 7)), (32, 192, 1400, 8, 40, (11, 12)), (32, 128, 2800, 8, 40, (7,)),
                              (128, 128, 1400, 8, 40, (7,)), (32, 64, 1400, 8, 40, (8,))):
    q = torch.randn(b * lq, h * d, device="cuda").to(dt); k = torch.randn(b * lk, h * d, device="cuda").to(dt); v = torch.randn(b * lk, h * d, device="cuda").to(dt)
    out = torch.empty_like(q)
    print((b, lq, lk, h, d), " | ".join("v%d %.1f us" % (var, graph_time(lambda: O.attention(q, k, v, b, lq, lk, h, d, out=out, variant=var))) for var in vs))
