"""How the hot kernels' in-graph duration scales when the instance count doubles (12 -> 24 view-instances): the
case for running the two ControlNet branches as ONE grouped launch sequence instead of two streams.
python tools/batch_scaling.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dualdiff_amd import ops as O
from tools._timing import graph_time
dt = torch.float16
LEVELS = ((28, 50, 320, 40), (14, 25, 640, 80), (7, 13, 1280, 160), (4, 7, 1280, 160))


def r(*s, scale=1.0):
    return (torch.randn(*s, device="cuda") * scale).to(dt)


tot = {12: 0.0, 24: 0.0}
for (h, w, c, d) in LEVELS:
    print("level %dx%d C=%d" % (h, w, c))
    for name in ("gemm CxC+res", "gemm qkv", "geglu", "ff2", "conv3", "attn self", "attn cross", "gn", "ln"):
        row = []
        for b in (12, 24):
            l = h * w
            rows = b * l
            x = r(rows, c)
            if name == "gemm CxC+res":
                wt, bi = r(c, c, scale=c ** -0.5), r(c)
                fn = lambda: O.gemm(x, wt, bi, res=x)
            elif name == "gemm qkv":
                wt = r(3 * c, c, scale=c ** -0.5)
                fn = lambda: O.gemm(x, wt, None, head_major=(d, 8, 0.2))
            elif name == "geglu":
                wt, bi = r(8 * c, c, scale=c ** -0.5), r(8 * c)
                fn = lambda: O.gemm(x, wt, bi, epilogue=O.DD_EPI_GEGLU)
            elif name == "ff2":
                x4 = r(rows, 4 * c)
                wt, bi = r(c, 4 * c, scale=(4 * c) ** -0.5), r(c)
                fn = lambda: O.gemm(x4, wt, bi, res=x)
            elif name == "conv3":
                wt, bi = r(c, 9 * c, scale=(9 * c) ** -0.5), r(c)
                fn = lambda: O.conv3x3(x, wt, bi, b, h, w)
            elif name == "attn self":
                q = r(3 * 8, rows, d)
                o = torch.empty(rows, c, device="cuda", dtype=dt)
                fn = lambda: O.attention(q[:8], q[8:16], q[16:], b, l, l, 8, d, out=o, q_prescaled=True)
            elif name == "attn cross":
                q, kv = r(8, rows, d), r(16, b * 98, d)
                o = torch.empty(rows, c, device="cuda", dtype=dt)
                fn = lambda: O.attention(q, kv[:8], kv[8:], b, l, 98, 8, d, out=o, q_prescaled=True)
            elif name == "gn":
                g, be = r(c), r(c)
                fn = lambda: O.groupnorm(x, g, be, b, l, 32, 1e-5, True)
            elif name == "ln":
                g, be = r(c), r(c)
                fn = lambda: O.layernorm(x, g, be)
            fn()
            row.append(graph_time(fn))
        print("  %-14s 12: %6.1f us   24: %6.1f us   x%.2f" % (name, row[0], row[1], row[1] / row[0]))
