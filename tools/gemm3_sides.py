"""One line per library (DD_HIP_LIB = the product or a diagnostic build of tools/gemm3_bound.sh): hot HIP-graph-chain time of
pipelined dense tiles on the dominant shapes.  python tools/gemm3_sides.py [label]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dualdiff_amd import ops as O
from tools._timing import graph_time
label = sys.argv[1] if len(sys.argv) > 1 else "product"
dt, dev = torch.float16, torch.device("cuda")
O.workspace(512 << 20, dev)
SHAPES = [(1092, 1280, 1280, 72), (1092, 1280, 1280, 73), (336, 1280, 1280, 77), (4200, 640, 640, 72), (4200, 640, 2560, 72),
          (16800, 320, 320, 78), (1092, 1280, 5120, 73), (4200, 1920, 640, 75), (16800, 960, 320, 75)]
out = "%-8s" % label
for rows, n, k, t in SHAPES:
    x = torch.randn(rows, k, device=dev).to(dt)
    w = (torch.randn(n, k, device=dev) * k ** -0.5).to(dt)
    bi = torch.randn(n, device=dev).to(dt)
    us = min(graph_time(lambda: O.gemm(x, w, bi, tile=t, split_k=1), n=8) for _ in range(3))
    out += " | %dx%dx%d t%d %5.1f" % (rows, n, k, t, us)
print(out, flush=True)
