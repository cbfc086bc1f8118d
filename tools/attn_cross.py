"""Cross-attention shapes of the step: row-major q vs head-major prescaled q (K/V row-major from the bank)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dualdiff_amd import ops as O
from tools.attn_variants import graph_time
dt = torch.bfloat16
for (b, lq, lk, h, d) in ((12, 1400, 98, 8, 40), (12, 350, 98, 8, 80), (12, 91, 98, 8, 160), (12, 28, 98, 8, 160),
                          (12, 350, 350, 8, 80), (12, 91, 91, 8, 160), (6, 350, 350, 8, 80), (6, 350, 98, 8, 80)):
    c = h * d
    q = torch.randn(b * lq, c, device="cuda").to(dt); kv = torch.randn(b * lk, 2 * c, device="cuda").to(dt)
    qh = q.view(b * lq, h, d).permute(1, 0, 2).contiguous()
    kh = kv[:, :c].reshape(b * lk, h, d).permute(1, 0, 2).contiguous(); vh = kv[:, c:].reshape(b * lk, h, d).permute(1, 0, 2).contiguous()
    out = torch.empty(b * lq, c, device="cuda", dtype=dt)
    t0 = graph_time(lambda: O.attention(q, kv[:, :c], kv[:, c:], b, lq, lk, h, d, out=out))
    t1 = graph_time(lambda: O.attention(qh, kv[:, :c], kv[:, c:], b, lq, lk, h, d, out=out, q_prescaled=True))
    t2 = graph_time(lambda: O.attention(qh, kh, vh, b, lq, lk, h, d, out=out, q_prescaled=True))
    t3 = graph_time(lambda: O.attention(q, kv[:, :c], kv[:, c:], b, lq, lk, h, d, out=out, q_prescaled=True))
    print((b, lq, lk, h, d), "row-major %.1f | q head-major+prescaled %.1f | all head-major+prescaled %.1f | row-major prescaled %.1f" % (t0, t1, t2, t3))
