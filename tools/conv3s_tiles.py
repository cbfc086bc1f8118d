"""Cold-weight timing of the deep-level conv shapes over the conv3s tiles x split-K."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dualdiff_amd import ops as O
M = 12; dt = torch.bfloat16; dev = torch.device("cuda:0")
def r(*shape, s=1.0): return (torch.randn(*shape, device="cuda") * s).to(dt)
def cold(fn, warm, n=5):
    ts = []
    for _ in range(n):
        O._flush_and_warm(dev, warm)
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort(); return ts[len(ts) // 2]
for (cin, cout, h, w) in ((1280, 1280, 4, 7), (2560, 1280, 4, 7), (1280, 1280, 7, 13), (2560, 1280, 7, 13), (1920, 1280, 7, 13)):
    x, wt, b = r(M * h * w, cin), r(cout, 9 * cin, s=0.02), r(cout)
    res = []
    for tile in (31, 33, 34, 35, 36, 37, 38, 13, 17, 20, 11):
        for sp in (1, 2, 3, 4, 5, 6, 8, 10, 12, 16, 20):
            try:
                t = cold(lambda: O.conv3x3(x, wt, b, M, h, w, tile=tile, split_k=sp), (x,))
            except Exception:
                continue
            res.append((t, tile, sp))
    res.sort()
    print((cin, cout, h, w), " | ".join("t%d/s%d %.1f" % (tl, sp, t) for t, tl, sp in res[:10]))
