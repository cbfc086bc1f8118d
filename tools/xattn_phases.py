"""dd_xattn320 with parts of its work skipped (DD_XATTN_DBG, read once per process: run once per mode):
0 product, 1 no attention phase, 2 no MFMAs in the products, 4 no softmax.  Hot graph chain, 12 instances x 1400 rows,
98 keys, head-major K / V.  python tools/xattn_phases.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dualdiff_amd import ops as O
from tools._timing import graph_time
dt, dev = torch.float16, torch.device("cuda")
inst, lq, lk = 12, 1400, int(os.environ.get("XLK", "98"))
x = torch.randn(inst * lq, 320, device=dev).to(dt)
wq = (torch.randn(320, 320, device=dev) * 320 ** -0.5).to(dt)
wo = (torch.randn(320, 320, device=dev) * 320 ** -0.5).to(dt)
bo = torch.randn(320, device=dev).to(dt)
k = torch.randn(8, inst * lk, 40, device=dev).to(dt)
v = torch.randn(8, inst * lk, 40, device=dev).to(dt)
wqp, wop = O.xattn_pack_weight(wq), O.xattn_pack_weight(wo)
g = torch.ones(320, device=dev).to(dt); b = torch.zeros(320, device=dev).to(dt)
us = min(graph_time(lambda: O.xattn320(x, wqp, wop, bo, k, v, inst, lq, lk, 40 ** -0.5, res=x), n=8) for _ in range(3))
us_ln = min(graph_time(lambda: O.xattn320(x, wqp, wop, bo, k, v, inst, lq, lk, 40 ** -0.5, res=x, ln_out=(g, b, 1e-5)), n=8) for _ in range(3))
print("DD_XATTN_DBG=%s lk=%d: %.1f us (with LayerNorm output %.1f us)" % (os.environ.get("DD_XATTN_DBG", "0"), lk, us, us_ln), flush=True)
