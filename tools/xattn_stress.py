import sys, torch
sys.path.insert(0, "/root/repo")
from dualdiff_amd import ops as O
dt = torch.float16; dev = "cuda"
g = torch.Generator(device=dev).manual_seed(3)
r = lambda *s, sc=1.0: (torch.randn(*s, generator=g, device=dev) * sc).to(dt)
n, C = 1400, 320
cases = []
for inst, lk in ((6, 15), (6, 9), (6, 98)):
    x = r(inst * n, C); res = r(inst * n, C); wq = r(C, C, sc=C ** -0.5); wo = r(C, C, sc=C ** -0.5); b = r(C)
    bank = r(inst * lk, 11520); k, v = bank[:, 640:960], bank[:, 960:1280]
    gm, bt = r(C), r(C)
    args = (x, wq, wo, b, k, v, inst, n, lk, 40 ** -0.5)
    ref = O.xattn320(*args, res=res, ln_out=(gm, bt, 1e-5))
    cases.append((args, res, (gm, bt, 1e-5), ref, ref._ln_out))
torch.cuda.synchronize()
streams = [torch.cuda.Stream() for _ in cases]
big = torch.empty(1 << 28, dtype=torch.float16, device=dev)
bad = 0
outs = []
for it in range(60):
    for (args, res, ln, ref, lnref), st in zip(cases, streams):
        with torch.cuda.stream(st):
            y = O.xattn320(*args, res=res, ln_out=ln)
            outs.append((y, y._ln_out, ref, lnref))
    if it % 3 == 0:
        big.add_(1)
torch.cuda.synchronize()
for y, yl, ref, lnref in outs:
    if not torch.equal(y, ref) or not torch.equal(yl, lnref):
        bad += 1
print("concurrent launches with wrong bits: %d of %d" % (bad, len(outs)))
