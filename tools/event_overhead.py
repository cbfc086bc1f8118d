import torch, sys
sys.path.insert(0, "/root/repo")
from dualdiff_amd import ops as O
st = torch.cuda.current_stream()
x = torch.zeros(1 << 20, device="cuda", dtype=torch.bfloat16)
def brackets(fn, n=200):
    ps = []
    for _ in range(n):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(st); fn(); e1.record(st); ps.append((e0, e1))
    torch.cuda.synchronize()
    d = sorted(a.elapsed_time(b) * 1e3 for a, b in ps)
    return d[0], d[len(d) // 4], d[len(d) // 2], d[-1]
print("empty      min/q1/med/max us", brackets(lambda: None))
y = torch.empty_like(x)
print("scale 2MB  min/q1/med/max us", brackets(lambda: O.scale(x, 2.0, out=y)))
x8 = x[:64]; y8 = y[:64]
print("scale 128B min/q1/med/max us", brackets(lambda: O.scale(x8, 2.0, out=y8)))
a = torch.randn(16800, 320, device="cuda").to(torch.bfloat16); w = torch.randn(320, 320, device="cuda").to(torch.bfloat16)
o = torch.empty(16800, 320, device="cuda", dtype=torch.bfloat16)
print("gemm       min/q1/med/max us", brackets(lambda: O.gemm(a, w, None, out=o, tile=17, split_k=1)))
