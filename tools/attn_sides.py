"""Attention main-loop diagnosis: the step's attention shapes timed hot with the product library and with diagnostic builds
(DD_HIP_LIB=.../libdd_attn_nomfma.so: matrix instructions removed; libdd_attn_noexp.so: exponentials become moves)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dualdiff_amd import ops as O
from tools._timing import graph_time
dt = torch.float16
print("lib:", os.path.basename(os.environ.get("DD_HIP_LIB", "product")))
for (b, lq, lk, h, d) in ((12, 1400, 1400, 8, 40), (12, 350, 350, 8, 80), (12, 91, 91, 8, 160), (12, 1400, 98, 8, 40)):
    q = torch.randn(3 * h, b * max(lq, lk), d, device="cuda").to(dt)
    out = torch.empty(b * lq, h * d, device="cuda", dtype=dt)
    t = graph_time(lambda: O.attention(q[:h, :b * lq], q[h:2 * h, :b * lk], q[2 * h:, :b * lk], b, lq, lk, h, d, out=out, q_prescaled=True))
    print("  b=%d lq=%d lk=%d d=%d: %7.1f us  %6.1f TFLOP/s" % (b, lq, lk, d, t, 4.0 * b * h * lq * lk * d / t / 1e6))
