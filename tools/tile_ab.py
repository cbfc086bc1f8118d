"""Hot / cold timing of explicit GEMM tiles on given shapes: python tools/tile_ab.py  (GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dualdiff_amd import ops as O

dt = torch.float16
shapes = [(16800, 320, 320), (16800, 320, 640), (1092, 1280, 1280), (4200, 640, 640), (336, 1280, 1280), (1092, 1280, 2560)]
tiles = [40, 27, 28, 13, 14, 15, 52, 59]
flush = torch.empty(512 << 20, dtype=torch.uint8, device="cuda")
for rows, n, k in shapes:
    a = torch.randn(rows, k, device="cuda", dtype=dt)
    w = torch.randn(n, k, device="cuda", dtype=dt) * k ** -0.5
    b = torch.randn(n, device="cuda", dtype=dt)
    res = torch.randn(rows, n, device="cuda", dtype=dt)
    g = torch.ones(n, device="cuda", dtype=dt)
    ref = None
    for tile in tiles:
        for lno in ((False, True) if tile == 40 and n == 320 else (False,)):
            kw = dict(res=res, tile=tile, split_k=1)
            if lno:
                kw = dict(res=res, ln_out=(g, b, 1e-5))
            try:
                y = O.gemm(a, w, b, **kw)
            except Exception as e:
                print(rows, n, k, tile, "unsupported", str(e)[:60]); continue
            if ref is None:
                ref = y.clone()
            ok = torch.equal(y, ref)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for _ in range(5):
                O.gemm(a, w, b, **kw)
            e0.record()
            for _ in range(50):
                O.gemm(a, w, b, **kw)
            e1.record(); e1.synchronize()
            hot = e0.elapsed_time(e1) / 50 * 1e3
            cold = []
            for _ in range(7):
                flush.zero_()
                e0.record(); O.gemm(a, w, b, **kw); e1.record(); e1.synchronize()
                cold.append(e0.elapsed_time(e1) * 1e3)
            cold.sort()
            print("%5dx%4dx%4d tile %2d %s hot %6.1f us  cold %6.1f us  equal-to-first %s" % (rows, n, k, tile, "ln_out" if lno else "      ", hot, cold[3], ok))
