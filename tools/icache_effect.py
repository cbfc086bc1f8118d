"""Does a launch pay for a cold INSTRUCTION cache?  The same GEMM launch (hot data) timed in a graph chain (a) back to back
and (b) alternating with launches of OTHER kernels with large code (three conv / GEMM instantiations on tiny inputs, whose
own time is measured separately and subtracted)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dualdiff_amd import ops as O
from tools.attn_variants import graph_time
dt = torch.float16


def r(*s, scale=1.0):
    return (torch.randn(*s, device="cuda") * scale).to(dt)


# "evictors": different symbols, tiny problems
xe = r(12 * 4 * 7, 128); we = r(128, 9 * 128, scale=0.03); be = r(128)
xg = r(64, 320); wg = r(320, 320, scale=0.05)
xq = r(96, 640); wq = r(1920, 640, scale=0.04)


def evict():
    O.conv3x3(xe, we, be, 12, 4, 7, tile=35, split_k=1)       # conv3s 128x64
    O.gemm(xg, wg, None, tile=40)                              # 80x320
    O.gemm(xq, wq, None, tile=13)                              # 128x64/dma3
    O.layernorm(xq, wq[0], wq[1])
    O.groupnorm(xe, be, be, 12, 28, 32, 1e-5, True)


def evict_small():                                               # small-code kernels only
    O.layernorm(xq, wq[0], wq[1])
    O.groupnorm(xe, be, be, 12, 28, 32, 1e-5, True)
    O.layernorm(xq, wq[0], wq[1])
    O.groupnorm(xe, be, be, 12, 28, 32, 1e-5, True)
    O.layernorm(xq, wq[0], wq[1])


evict(); evict_small()
t_ev = graph_time(evict)
t_evs = graph_time(evict_small)
print("evictor sequences alone: big-code %.1f us, small-code %.1f us" % (t_ev, t_evs))
for (rows, n, k, tile, hm) in ((1092, 3840, 1280, 20, 160), (4200, 1920, 640, 20, 80), (16800, 320, 320, 40, 0),
                               (1092, 1280, 1280, 15, 0), (4200, 640, 640, 18, 0), (16800, 960, 320, 13, 40)):
    a, w = r(rows, k), r(n, k, scale=k ** -0.5)
    kw = {"head_major": (hm, 8, 0.2)} if hm else {}
    fn = lambda: O.gemm(a, w, None, tile=tile, **kw)
    fn()
    t0 = graph_time(fn)

    def both():
        fn(); evict()
    t1 = graph_time(both) - t_ev

    def both_s():
        fn(); evict_small()
    t2 = graph_time(both_s) - t_evs
    print("gemm %5dx%4dx%4d tile %2d: back-to-back %6.1f us   between big-code kernels %6.1f (+%.1f)   between small-code kernels %6.1f (+%.1f)" %
          (rows, n, k, tile, t0, t1, t1 - t0, t2, t2 - t0))
