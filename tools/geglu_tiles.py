"""Per-tile timing of the GEGLU projection shapes of the step (graph chain; weights hot): python tools/geglu_tiles.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dualdiff_amd import ops as O
from tools._timing import graph_time
dt = torch.float16
for rows, c in ((16800, 320), (4200, 640), (1092, 1280), (336, 1280)):
    a = torch.randn(rows, c, device="cuda").to(dt); w = (torch.randn(8 * c, c, device="cuda") * c ** -0.5).to(dt)
    b = torch.randn(8 * c, device="cuda").to(dt)
    ref = O.gemm(a, w, b, epilogue=O.DD_EPI_GEGLU, tile=12)
    r = []
    for t in (11, 12, 14, 16, 20, 24, 25, 29, 44, 46, 50):
        try:
            y = O.gemm(a, w, b, epilogue=O.DD_EPI_GEGLU, tile=t)
            ok = torch.equal(y, ref)
            us = graph_time(lambda: O.gemm(a, w, b, epilogue=O.DD_EPI_GEGLU, tile=t))
            r.append((us, t, ok))
        except Exception as e:
            pass
    r.sort()
    gf = 2.0 * rows * 8 * c * c
    print("GEGLU %dx%dx%d:" % (rows, 8 * c, c), " ".join("t%d:%.1f%s" % (t, us, "" if ok else "!") for us, t, ok in r),
          " best %.0f TFLOP/s" % (gf / r[0][0] / 1e6))
