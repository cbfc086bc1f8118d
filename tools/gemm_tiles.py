"""Per-tile timing of one dense GEMM shape (graph chain, activations hot): python tools/gemm_tiles.py rows n k [res]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dualdiff_amd import ops as O, _native
from tools._timing import graph_time
rows, n, k = [int(x) for x in sys.argv[1:4]]
use_res = len(sys.argv) > 4
dt = torch.bfloat16
a = torch.randn(rows, k, device="cuda").to(dt); w = (torch.randn(n, k, device="cuda") * k ** -0.5).to(dt)
b = torch.randn(n, device="cuda").to(dt); res = torch.randn(rows, n, device="cuda").to(dt) if use_res else None
out = torch.empty(rows, n, device="cuda", dtype=dt)
lib = _native.load()
tiles = [lib.dd_gemm_tile_id(i) for i in range(lib.dd_gemm_num_tiles())]
r = []
for t in tiles:
    for sp in (1,):
        try:
            us = graph_time(lambda: O.gemm(a, w, b, res=res, out=out, tile=t, split_k=sp))
            r.append((us, t, sp))
        except Exception as e:
            pass
r.sort()
print("%dx%dx%d res=%s:" % (rows, n, k, use_res), " ".join("t%d:%.1f" % (t, us) for us, t, sp in r[:12]))
