"""One line per conv shape: hot HIP-graph-chain time of the direct / band 3x3 conv with the library DD_HIP_LIB names
(the product or one of the diagnostic builds of tools/conv3s_bound.sh).  python tools/conv3s_sides.py [label]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dualdiff_amd import ops as O
from tools._timing import graph_time
label = sys.argv[1] if len(sys.argv) > 1 else "product"
dt, dev = torch.float16, torch.device("cuda")
O.workspace(512 << 20, dev)
SHAPES = [(12, 28, 50, 320, 320, 39), (12, 14, 25, 640, 640, 31), (12, 7, 13, 1280, 1280, 31), (48, 28, 50, 320, 320, 39)]
out = "%-10s" % label
for m, h, w, cin, cout, t in SHAPES:
    rows = m * h * w
    x = torch.randn(rows, cin, device=dev).to(dt)
    bi = torch.randn(cout, device=dev).to(dt)
    wt = (torch.randn(cout, 9 * cin, device=dev) * (9 * cin) ** -0.5).to(dt)
    us = min(graph_time(lambda: O.conv3x3(x, wt, bi, m, h, w, tile=t, split_k=1), n=8) for _ in range(3))
    out += " | %dx%dx%d %d->%d t%d %6.1f us %5.0f TF" % (m, h, w, cin, cout, t, us, 2.0 * rows * cout * 9 * cin / us / 1e6)
print(out, flush=True)
