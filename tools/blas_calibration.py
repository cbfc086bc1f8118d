"""Calibration only: what the vendor GEMM (torch.mm -> hipBLASLt / rocBLAS) reaches on this path's GEMM shapes,
next to dd_gemm on the same operands (plain GEMM, no epilogue).  Not part of the product path.
python tools/blas_calibration.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dualdiff_amd import ops as O
from tools._timing import graph_time
dt = torch.float16
SHAPES = ((16800, 320, 2880), (4200, 640, 5760), (1092, 1280, 11520), (336, 1280, 11520), (16800, 2560, 320),
          (4200, 5120, 640), (1092, 10240, 1280), (16800, 960, 320), (4200, 1920, 640), (1092, 3840, 1280),
          (16800, 320, 320), (4200, 640, 640), (1092, 1280, 1280), (16800, 320, 1280), (4200, 640, 2560),
          (1092, 1280, 5120), (8192, 8192, 8192))
for (m, n, k) in SHAPES:
    a = (torch.randn(m, k, device="cuda")).to(dt)
    w = (torch.randn(n, k, device="cuda") * k ** -0.5).to(dt)
    out = torch.empty(m, n, device="cuda", dtype=dt)
    t_blas = graph_time(lambda: torch.mm(a, w.t(), out=out))
    try:
        t_dd = graph_time(lambda: O.gemm(a, w, None, out=out))
    except Exception as e:
        t_dd = float("nan")
    fl = 2.0 * m * n * k
    print("%6dx%5dx%5d  blas %7.1f us %7.1f TF/s | dd_gemm %7.1f us %7.1f TF/s" %
          (m, n, k, t_blas, fl / t_blas * 1e-6, t_dd, fl / t_dd * 1e-6))
