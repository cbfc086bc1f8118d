"""Calibration: how fast can ANY kernel pull N MB of cold data (after the autotuner's cache flush)?"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dualdiff_amd import ops as O
dev = torch.device("cuda:0"); dt = torch.bfloat16
def cold(fn, n=6, flush=True):
    tot = 0.0
    for _ in range(n):
        if flush: O._flush_and_warm(dev, ())
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize(); tot += e0.elapsed_time(e1)
    return tot / n * 1e3
for mb in (1, 3.3, 7.4, 29.5, 59, 118):
    n = int(mb * 1e6 / 2) // 8 * 8
    x = torch.randn(n, device=dev).to(dt); y = torch.empty_like(x)
    small = torch.empty(1024, device=dev, dtype=dt)
    t_empty = cold(lambda: O.scale(small, 1.0, out=small))
    t = cold(lambda: O.scale(x, 1.0, out=y))
    th = cold(lambda: O.scale(x, 1.0, out=y), flush=False)
    s = cold(lambda: x.sum())
    print("%6.1f MB: scale (read+write) cold %.1f us, hot %.1f us; torch.sum (read only) cold %.1f us; tiny kernel %.1f us" % (mb, t, th, s, t_empty))
