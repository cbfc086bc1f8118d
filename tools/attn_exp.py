import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dualdiff_amd import ops as O
from tools.attn_variants import graph_time
dt = torch.bfloat16
res = []
for (b, lq, lk, h, d) in ((128, 128, 1400, 8, 40), (32, 128, 1400, 8, 40), (32, 128, 2800, 8, 40)):
    q = torch.randn(b * lq, h * d, device="cuda").to(dt); k = torch.randn(b * lk, h * d, device="cuda").to(dt); v = torch.randn(b * lk, h * d, device="cuda").to(dt)
    out = torch.empty_like(q)
    res.append("%.1f" % graph_time(lambda: O.attention(q, k, v, b, lq, lk, h, d, out=out, variant=7)))
print(os.environ.get("DD_HIP_LIB", "base"), "sat/lone/lone2800:", " ".join(res))
