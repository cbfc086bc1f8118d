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
