"""How much of a weight-bearing launch is the coldness of its weights: the same launch timed in a graph chain with ONE
weight buffer (resident in the 256 MB Infinity Cache after the first pass) and rotating over enough buffers to exceed it
(every launch streams its weights from HBM, as in the real step: 3.3 GB of weights per step)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dualdiff_amd import ops as O
from tools._timing import graph_time
dt = torch.float16


def r(*s, scale=1.0):
    return (torch.randn(*s, device="cuda") * scale).to(dt)


def ab(name, make_w, call, wbytes):
    nbuf = max(2, int(600e6 // wbytes) + 1)
    ws = [make_w() for _ in range(nbuf)]
    state = {"i": 0}

    def hot():
        call(ws[0])

    def cold():
        call(ws[state["i"] % nbuf]); state["i"] += 1
    call(ws[0])
    th = graph_time(hot, n=nbuf)
    tc = graph_time(cold, n=nbuf)
    # rotation over ~110 MB: beyond the 8 x 4 MB L2s, inside the 256 MB Infinity Cache
    nm = max(2, min(nbuf, int(110e6 // wbytes)))
    state["i"] = 0

    def mall():
        call(ws[state["i"] % nm]); state["i"] += 1
    tm = graph_time(mall, n=nm * 3)
    print("%-34s weights %5.1f MB: L2-hot %6.1f us   Infinity-Cache-hot (%3d bufs) %6.1f us   HBM-cold (%4d bufs) %6.1f us" %
          (name, wbytes / 1e6, th, nm, tm, nbuf, tc))


for (b, h, w, c) in ((12, 28, 50, 320), (12, 14, 25, 640), (12, 7, 13, 1280), (12, 4, 7, 1280)):
    rows = b * h * w
    x, bi = r(rows, c), r(c)
    ab("conv3 %dx%dx%d" % (rows, c, 9 * c), lambda: r(c, 9 * c, scale=(9 * c) ** -0.5),
       lambda wt: O.conv3x3(x, wt, bi, b, h, w), c * 9 * c * 2)
    ab("CxC+res %dx%dx%d" % (rows, c, c), lambda: r(c, c, scale=c ** -0.5), lambda wt: O.gemm(x, wt, bi, res=x), c * c * 2)
    b8 = r(8 * c)
    ab("geglu %dx%dx%d" % (rows, 8 * c, c), lambda: r(8 * c, c, scale=c ** -0.5),
       lambda wt: O.gemm(x, wt, b8, epilogue=O.DD_EPI_GEGLU), 8 * c * c * 2)
    x4 = r(rows, 4 * c)
    ab("ff2 %dx%dx%d" % (rows, c, 4 * c), lambda: r(c, 4 * c, scale=(4 * c) ** -0.5), lambda wt: O.gemm(x4, wt, bi, res=x),
       4 * c * c * 2)
