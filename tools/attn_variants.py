"""Graph-chain timing of the attention kernel variants (rows per wave x key tile) on the step's shapes."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dualdiff_amd import ops as O
dt = torch.bfloat16
def graph_time(fn, n=20, reps=10):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn(); torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            for _ in range(n): fn()
    torch.cuda.current_stream().wait_stream(s)
    g.replay(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): g.replay()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps / n * 1e6
if __name__ == "__main__":
    for (b, lq, lk, h, d) in ((12, 1400, 1400, 8, 40), (12, 350, 350, 8, 80), (12, 1400, 98, 8, 40), (6, 1400, 1400, 8, 40), (12, 350, 98, 8, 80), (12, 91, 91, 8, 160)):
        q = torch.randn(b * lq, h * d, device="cuda").to(dt); k = torch.randn(b * lk, h * d, device="cuda").to(dt); v = torch.randn(b * lk, h * d, device="cuda").to(dt)
        out = torch.empty_like(q)
        res = []
        for var in (0, 13, 5, 7, 8, 11, 12):
            try:
                res.append("v%d %.1f us" % (var, graph_time(lambda: O.attention(q, k, v, b, lq, lk, h, d, out=out, variant=var))))
            except Exception as e:
                res.append("v%d n/a" % var)
        print((b, lq, lk, h, d), " | ".join(res))

    # head-major K/V (and Q) planes from the projection epilogue vs the fused row-major projection
    for (b, l, h, d) in ((12, 1400, 8, 40), (6, 1400, 8, 40), (12, 350, 8, 80)):
        c = h * d
        x = torch.randn(b * l, c, device="cuda").to(dt); w = (torch.randn(3 * c, c, device="cuda") * c ** -0.5).to(dt)
        qkv = O.gemm(x, w, None); hm = O.gemm(x, w, None, head_major=(d, h, d ** -0.5 * 1.4426950408889634))
        out = torch.empty(b * l, c, device="cuda", dtype=dt)
        t_rm = graph_time(lambda: O.attention(qkv[:, :c], qkv[:, c:2 * c], qkv[:, 2 * c:], b, l, l, h, d, out=out))
        t_hm = graph_time(lambda: O.attention(hm[:h], hm[h:2 * h], hm[2 * h:], b, l, l, h, d, out=out, q_prescaled=True))
        g_rm = graph_time(lambda: O.gemm(x, w, None, out=qkv))
        g_hm = graph_time(lambda: O.gemm(x, w, None, head_major=(d, h, 1.0)))
        print((b, l, h, d), "attention row-major %.1f us | head-major %.1f us ; QKV gemm row-major %.1f | head-major %.1f" % (t_rm, t_hm, g_rm, g_hm))
