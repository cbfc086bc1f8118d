"""One line per library / switch (DD_HIP_LIB = the product or a diagnostic build of tools/gemm4_bound.sh; DD_PERSIST3=0 = one
tile per workgroup): hot HIP-graph-chain time of the pipelined dense tiles on the short-K shapes VERDICT r5 item 1 names,
at 12 and at 48 view-instances.  python tools/gemm4_sides.py [label]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dualdiff_amd import ops as O
from tools._timing import graph_time
label = sys.argv[1] if len(sys.argv) > 1 else "product"
dt, dev = torch.float16, torch.device("cuda")
O.workspace(512 << 20, dev)
# (rows, weight rows, K, GEGLU, tiles)
SHAPES = [(16800, 2560, 320, True, (75, 50)), (16800, 960, 320, False, (75, 72)), (4200, 5120, 640, True, (75, 50)),
          (16800, 320, 320, False, (72, 78)), (16800, 320, 1280, False, (72, 78)),
          (67200, 2560, 320, True, (75, 50)), (67200, 960, 320, False, (75, 72)), (67200, 320, 320, False, (72, 78)),
          (16800, 5120, 640, True, (75, 50)), (16800, 1920, 640, False, (75,)), (16800, 640, 640, False, (72, 75))]
out = "%-22s" % label
for rows, n, k, geglu, tiles in SHAPES:
    x = torch.randn(rows, k, device=dev).to(dt)
    w = (torch.randn(n, k, device=dev) * k ** -0.5).to(dt)
    bi = torch.randn(n, device=dev).to(dt)
    out += " | %dx%dx%d%s" % (rows, n, k, "g" if geglu else "")
    for t in tiles:
        kw = {"epilogue": O.DD_EPI_GEGLU} if geglu else {"split_k": 1}
        try:
            us = min(graph_time(lambda: O.gemm(x, w, bi, tile=t, **kw), n=8) for _ in range(3))
            out += " t%d %5.1f" % (t, us)
        except Exception:
            out += " t%d n/a" % t
print(out, flush=True)
if label == "product":          # the LayerNorm-emitting 80 x 320 tile: pipelined (74: persistent beyond 256 row tiles) vs dd_gemm2 (40)
    line = "%-22s" % "ln-out 80x320"
    for rows, k in ((16800, 320), (67200, 320), (67200, 1280)):
        x = torch.randn(rows, k, device=dev).to(dt)
        w = (torch.randn(320, k, device=dev) * k ** -0.5).to(dt)
        bi, ga, be = (torch.randn(320, device=dev).to(dt) for _ in range(3))
        res = torch.randn(rows, 320, device=dev).to(dt)
        line += " | %dx320x%d" % (rows, k)
        for t in (74, 40):
            us = min(graph_time(lambda: O.gemm(x, w, bi, res=res, ln_out=(ga, be, 1e-5), tile=t), n=8) for _ in range(3))
            line += " t%d %5.1f" % (t, us)
    print(line, flush=True)
