"""DIAGNOSTIC: global -> LDS staging rate of the LDS-DMA GEMM family.  With a -DDD_DBG_NOMFMA library the kernel time is
the staging + fragment-read time; bytes staged per workgroup and K-step are (BM + BN) * 128.  Adding -DDD_DBG_SAMEK makes
every step re-stage the same (L1-resident) bytes: if the rate rises, the cap is on the L2 side, otherwise L1 -> LDS."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dualdiff_amd import ops as O
from tools._timing import graph_time
dt = torch.float16
print("lib:", os.path.basename(os.environ.get("DD_HIP_LIB", "product")))
a = (torch.randn(8192, 8192, device="cuda")).to(dt)
w = (torch.randn(8192, 8192, device="cuda") * 0.01).to(dt)
for tile, bm, bn in ((19, 64, 64), (15, 64, 64), (12, 128, 128), (16, 256, 128), (28, 160, 160)):
    t = graph_time(lambda: O.gemm(a, w, None, tile=tile), n=3, reps=3)
    tiles = -(-8192 // bm) * -(-8192 // bn)
    staged = tiles * 128 * (bm + bn) * 128
    print("tile %2d (%3dx%3d): %8.1f us   staged %6.2f GB -> %5.1f TB/s = %5.1f GB/s per CU" %
          (tile, bm, bn, t, staged / 1e9, staged / t / 1e6, staged / t / 1e3 / 256))
