"""fp8 (W8A8, dd_gemm8) vs 16-bit projection GEMMs on the attention-projection shapes (graph chains, hot):
LayerNorm + GEMM against rowquant(LayerNorm) + gemm8.  python tools/gemm8_bench.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dualdiff_amd import ops as O
from tools._timing import graph_time
dt = torch.float16
for rows, n, k in ((16800, 960, 320), (16800, 320, 320), (4200, 1920, 640), (4200, 640, 640), (1092, 3840, 1280), (1092, 1280, 1280),
                   (336, 3840, 1280), (336, 1280, 1280), (4200, 5120, 640), (1092, 10240, 1280), (16800, 2560, 320)):
    x = torch.randn(rows, k, device="cuda").to(dt)
    g_ = torch.ones(k, device="cuda", dtype=dt); b_ = torch.zeros(k, device="cuda", dtype=dt)
    w = (torch.randn(n, k, device="cuda") * k ** -0.5).to(dt)
    w8, sw = O.quantize_fp8_padded(w)
    a8, sa = O.rowquant_fp8(x, (g_, b_, 1e-5))
    xn = O.layernorm(x, g_, b_, 1e-5)
    t_ln = graph_time(lambda: O.layernorm(x, g_, b_, 1e-5))
    t_g16 = graph_time(lambda: O.gemm(xn, w))
    t_q = graph_time(lambda: O.rowquant_fp8(x, (g_, b_, 1e-5)))
    t_g8 = graph_time(lambda: O.gemm8(a8, sa, w8, sw, dtype=dt))
    print("%6dx%5dx%4d  16-bit: LN %5.1f + GEMM %5.1f = %5.1f us (%4.0f TF)   fp8: quant %5.1f + GEMM %5.1f = %5.1f us (%4.0f TF)" % (
        rows, n, k, t_ln, t_g16, t_ln + t_g16, 2.0 * rows * n * k / t_g16 / 1e6, t_q, t_g8, t_q + t_g8, 2.0 * rows * n * k / t_g8 / 1e6), flush=True)
