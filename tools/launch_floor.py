"""Per-node cost of a HIP-graph replay: chains of N trivial / small kernels on one stream."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dualdiff_amd import ops as O
dt = torch.bfloat16
def bench_graph(fn, n=200, reps=20):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn(); torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            for _ in range(n): fn()
    torch.cuda.current_stream().wait_stream(s)
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps / n * 1e6
x = torch.randn(1024, device="cuda").to(dt); y = torch.empty_like(x)
print("tiny scale kernel (1 block): %.2f us/node" % bench_graph(lambda: O.scale(x, 2.0, out=y)))
xb = torch.randn(16800 * 320, device="cuda").to(dt); yb = torch.empty_like(xb)
print("scale 16800x320 (10.7 MB r + w): %.2f us/node" % bench_graph(lambda: O.scale(xb, 2.0, out=yb)))
for rows, n, k in ((16800, 320, 320), (4200, 640, 640), (1092, 1280, 1280), (336, 1280, 1280), (1176, 640, 768), (12, 1280, 320)):
    a = torch.randn(rows, k, device="cuda").to(dt); w = (torch.randn(n, k, device="cuda") * 0.05).to(dt)
    b = torch.randn(n, device="cuda").to(dt); out = torch.empty(rows, n, device="cuda", dtype=dt)
    O.gemm(a, w, b, out=out)   # autotune
    print("gemm %dx%dx%d: %.2f us/node  (%s)" % (rows, n, k, bench_graph(lambda: O.gemm(a, w, b, out=out)), O.gemm_kernel_name(rows, n, k, dt)))
h = torch.randn(16800, 320, device="cuda").to(dt); gam = torch.ones(320, device="cuda", dtype=dt); bet = torch.zeros(320, device="cuda", dtype=dt)
print("layernorm 16800x320: %.2f us/node" % bench_graph(lambda: O.layernorm(h, gam, bet, 1e-5)))
print("groupnorm 12x1400x320 (2 kernels): %.2f us/call" % bench_graph(lambda: O.groupnorm(h, gam, bet, 12, 1400, 32, 1e-5, True)))
for (m, hw, c) in ((12, 350, 640), (12, 350, 1280), (12, 91, 1280), (12, 91, 2560), (12, 28, 1280), (12, 28, 2560), (12, 1400, 640)):
    hh = torch.randn(m * hw, c, device="cuda").to(dt); g2 = torch.ones(c, device="cuda", dtype=dt); b2 = torch.zeros(c, device="cuda", dtype=dt)
    print("groupnorm %dx%dx%d: %.2f us/call" % (m, hw, c, bench_graph(lambda: O.groupnorm(hh, g2, b2, m, hw, 32, 1e-5, True))))
for (rows, c) in ((4200, 640), (1092, 1280), (336, 1280)):
    hh = torch.randn(rows, c, device="cuda").to(dt); g2 = torch.ones(c, device="cuda", dtype=dt); b2 = torch.zeros(c, device="cuda", dtype=dt)
    print("layernorm %dx%d: %.2f us/node" % (rows, c, bench_graph(lambda: O.layernorm(hh, g2, b2, 1e-5))))
