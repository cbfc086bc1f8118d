"""Cold-weight timing of one conv / GEMM shape over tile x split candidates (weights flushed from
L2 / Infinity Cache before every launch, activations re-touched): python tools/cold_conv.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dualdiff_amd import ops as O
M = 12; dt = torch.bfloat16; dev = torch.device("cuda:0")
def r(*shape, s=1.0): return (torch.randn(*shape, device="cuda") * s).to(dt)
def cold(fn, warm, n=5):
    tot = 0.0
    for _ in range(n):
        O._flush_and_warm(dev, warm)
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize()
        tot += e0.elapsed_time(e1)
    return tot / n * 1e3
names = {i: None for i in range(40)}
shapes = [("conv", 1280, 1280, 4, 7), ("conv", 2560, 1280, 4, 7), ("conv", 1280, 1280, 7, 13), ("gemm", 1092, 1280, 1280), ("gemm", 336, 1280, 1280), ("gemm", 4200, 640, 640)]
tiles = [int(t) for t in sys.argv[1].split(",")] if len(sys.argv) > 1 else [13, 17, 23, 20, 16, 26, 27, 11, 12, 15, 19, 21]
for sh in shapes:
    res = []
    if sh[0] == "conv":
        _, cin, cout, h, w = sh
        x, wt, b = r(M * h * w, cin), r(cout, 9 * cin, s=0.02), r(cout)
        for tile in tiles:
            for sp in (1, 2, 3, 4, 6, 8, 12, 16):
                try:
                    t = cold(lambda: O.conv3x3(x, wt, b, M, h, w, tile=tile, split_k=sp), (x,))
                except Exception as e:
                    continue
                res.append((t, tile, sp))
    else:
        _, rows, n, k = sh
        a, wt, b, rs = r(rows, k), r(n, k, s=0.05), r(n), r(rows, n)
        for tile in tiles:
            for sp in (1, 2, 4):
                try:
                    t = cold(lambda: O.gemm(a, wt, b, res=rs, tile=tile, split_k=sp), (a, rs))
                except Exception as e:
                    continue
                res.append((t, tile, sp))
    res.sort()
    print(sh, " | ".join("tile %d split %d: %.1f us" % (tl, sp, t) for t, tl, sp in res[:8]))
