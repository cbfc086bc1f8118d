"""Main-loop diagnosis: the path's MFMA-bound launches timed (hot graph chain) with the product library and with the
diagnostic builds of tools/build_dbg_libs.sh (DD_HIP_LIB=.../libdd_nomfma.so: loads + LDS reads only;
libdd_nodma.so: LDS reads + MFMAs only).  python tools/loop_sides.py   (run once per library)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dualdiff_amd import ops as O
from tools.attn_variants import graph_time
dt = torch.float16


def r(*s, scale=1.0):
    return (torch.randn(*s, device="cuda") * scale).to(dt)


print("lib:", os.environ.get("DD_HIP_LIB", "product"))
for (b, h, w, c) in ((12, 28, 50, 320), (12, 14, 25, 640), (12, 7, 13, 1280)):
    rows = b * h * w
    x = r(rows, c)
    wt, bi = r(c, 9 * c, scale=(9 * c) ** -0.5), r(c)
    t = graph_time(lambda: O.conv3x3(x, wt, bi, b, h, w))
    print("conv3 %6dx%5dx%5d %7.1f us %7.1f TF/s  %s" % (rows, c, 9 * c, t, 2.0 * rows * c * 9 * c / t * 1e-6,
          O.gemm_kernel_name(rows, c, 9 * c, dt, conv=True, cin=c, hw=(h, w))))
    w8, b8 = r(8 * c, c, scale=c ** -0.5), r(8 * c)
    t = graph_time(lambda: O.gemm(x, w8, b8, epilogue=O.DD_EPI_GEGLU))
    print("geglu %6dx%5dx%5d %7.1f us %7.1f TF/s" % (rows, 8 * c, c, t, 2.0 * rows * c * 8 * c / t * 1e-6))
    x4, w4 = r(rows, 4 * c), r(c, 4 * c, scale=(4 * c) ** -0.5)
    t = graph_time(lambda: O.gemm(x4, w4, bi, res=x))
    print("ff2   %6dx%5dx%5d %7.1f us %7.1f TF/s" % (rows, c, 4 * c, t, 2.0 * rows * c * 4 * c / t * 1e-6))
    w3 = r(3 * c, c, scale=c ** -0.5)
    t = graph_time(lambda: O.gemm(x, w3, None))
    print("qkv   %6dx%5dx%5d %7.1f us %7.1f TF/s" % (rows, 3 * c, c, t, 2.0 * rows * c * 3 * c / t * 1e-6))
a, wb = r(8192, 8192), r(8192, 8192, scale=0.01)
for tile in (0, 16, 26):
    try:
        t = graph_time(lambda: O.gemm(a, wb, None, tile=tile), n=3, reps=3)
        print("8192^3 tile %d %7.1f us %7.1f TF/s" % (tile, t, 2.0 * 8192 ** 3 / t * 1e-6))
    except Exception as e:
        print("tile", tile, "n/a", e)
