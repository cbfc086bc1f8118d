import sys, torch
sys.path.insert(0, "/root/repo")
from dualdiff_amd import ops as O
dt = torch.float16; dev = "cuda"
g = torch.Generator(device=dev).manual_seed(3)
r = lambda *s, sc=1.0: (torch.randn(*s, generator=g, device=dev) * sc).to(dt)
inst, n, lk, C = 12, 1400, 15, 320
x = r(inst * n, C); res = r(inst * n, C); wq = r(C, C, sc=C ** -0.5); wo = r(C, C, sc=C ** -0.5); b = r(C)
bank = r(inst * lk, 24960); k, v = bank[:, 640:960], bank[:, 960:1280]
gm, bt = r(C), r(C)
full = O.xattn320(x, wq, wo, b, k, v, inst, n, lk, 40 ** -0.5, res=res, ln_out=(gm, bt, 1e-5))
for sub in (1, 2, 4, 6):
    for rep in range(3):
        y = O.xattn320(x[:sub * n], wq, wo, b, k[:sub * lk], v[:sub * lk], sub, n, lk, 40 ** -0.5, res=res[:sub * n], ln_out=(gm, bt, 1e-5))
        d1 = (y.float() - full[:sub * n].float()).abs().max().item()
        d2 = (y._ln_out.float() - full._ln_out[:sub * n].float()).abs().max().item()
        print("instances %d rep %d: max |diff| vs the 12-instance launch: out %.3e  ln_out %.3e" % (sub, rep, d1, d2))
# offset subsets: instances 6..11 alone
y = O.xattn320(x[6 * n:], wq, wo, b, k[6 * lk:], v[6 * lk:], 6, n, lk, 40 ** -0.5, res=res[6 * n:])
print("instances 6..11 alone:", (y.float() - full[6 * n:].float()).abs().max().item())
