"""One line per library (DD_HIP_LIB = the product or a diagnostic build): hot HIP-graph-chain time of the GEGLU feed-forward
GEMMs of the step per tile.  python tools/geglu_sides.py [label]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dualdiff_amd import ops as O
from tools._timing import graph_time
label = sys.argv[1] if len(sys.argv) > 1 else "product"
dt, dev = torch.float16, torch.device("cuda")
O.workspace(512 << 20, dev)
SHAPES = [(16800, 2560, 320, (50, 75, 20)), (4200, 5120, 640, (50, 75, 20)), (1092, 10240, 1280, (50, 75, 20)), (67200, 2560, 320, (50, 75))]
out = "%-14s" % label
for rows, n, k, tiles in SHAPES:
    x = torch.randn(rows, k, device=dev).to(dt)
    w = (torch.randn(n, k, device=dev) * k ** -0.5).to(dt)
    bi = torch.randn(n, device=dev).to(dt)
    out += " | %dx%dx%d" % (rows, n, k)
    for t in tiles:
        try:
            us = min(graph_time(lambda: O.gemm(x, w, bi, epilogue=O.DD_EPI_GEGLU, tile=t), n=8) for _ in range(3))
            out += " t%d %5.1f" % (t, us)
        except Exception as e:
            out += " t%d n/a" % t
print(out, flush=True)
