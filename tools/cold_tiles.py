"""Per-tile timing of one dense GEMM shape the way the autotuner sees it (weights flushed by a 320 MB write, activations
re-touched, ONE launch per sample, median of 21) next to the hot graph-chain time: python tools/cold_tiles.py rows n k [hm_d]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dualdiff_amd import ops as O, _native
from tools.attn_variants import graph_time
rows, n, k = [int(v) for v in sys.argv[1:4]]
hm = int(sys.argv[4]) if len(sys.argv) > 4 else 0
dt = torch.float16
a = torch.randn(rows, k, device="cuda").to(dt)
w = (torch.randn(n, k, device="cuda") * k ** -0.5).to(dt)
lib = _native.load()
tiles = [lib.dd_gemm_tile_id(i) for i in range(lib.dd_gemm_num_tiles())]
kw = {"head_major": (hm, 8, 0.2)} if hm else {}
res = []
for t in tiles:
    for sp in (1, 2, 4):
        try:
            fn = lambda: O.gemm(a, w, None, tile=t, split_k=sp, **kw)
            fn()
        except Exception:
            continue
        s = []
        for _ in range(21):
            O._flush_and_warm(a.device, (a,))
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record(); e1.synchronize()
            s.append(e0.elapsed_time(e1) * 1e3)
        s.sort()
        res.append((s[10], s[2], s[18], graph_time(fn), t, sp))
res.sort()
print("gemm %dx%dx%d  (cold median [p10 p90] us | hot us)" % (rows, n, k))
for c, lo, hi, h, t, sp in res[:12]:
    print("  tile %2d split %d: cold %6.1f [%5.1f %5.1f] | hot %6.1f" % (t, sp, c, lo, hi, h))
