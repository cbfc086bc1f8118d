"""Per-tile timing of one 3x3 conv shape (graph chain, hot): python tools/conv_tiles.py cin cout h w [m]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dualdiff_amd import ops as O, _native
from tools._timing import graph_time
cin, cout, h, w = [int(x) for x in sys.argv[1:5]]
m = int(sys.argv[5]) if len(sys.argv) > 5 else 12
dt = torch.float16
x = torch.randn(m * h * w, cin, device="cuda").to(dt); wt = (torch.randn(cout, 9 * cin, device="cuda") * 0.02).to(dt)
b = torch.randn(cout, device="cuda").to(dt)
lib = _native.load()
tiles = [lib.dd_gemm_tile_id(i) for i in range(lib.dd_gemm_num_tiles())]
r = []
for t in tiles:
    for sp in (1, 2, 4):
        try:
            us = graph_time(lambda: O.conv3x3(x, wt, b, m, h, w, tile=t, split_k=sp))
            r.append((us, t, sp))
        except Exception:
            pass
r.sort()
fl = 2.0 * m * h * w * cout * 9 * cin
print("conv %dx%dx%d (%dx%d):" % (m * h * w, cout, 9 * cin, h, w), " ".join("t%d/%d:%.1f(%.0fTF)" % (t, sp, us, fl / us / 1e6) for us, t, sp in r[:10]))
