import os, sys, torch
sys.path.insert(0, "/root/repo")
from dualdiff_amd import ops as O
from tools.attn_variants import graph_time
dt=torch.float16; dev="cuda"
inst, LQ, LK, C = 12, 1400, 77, 320
r=lambda *s: torch.randn(*s, device=dev).to(dt)
x=r(inst*LQ,C); wq=r(C,C)*0.05; wo=r(C,C)*0.05; b=r(C); kvh=r(16,inst*LK,40)
print("dbg", os.environ.get("DD_XATTN_DBG","0"), "%.1f us" % graph_time(lambda: O.xattn320(x,wq,wo,b,kvh[:8],kvh[8:],inst,LQ,LK,40**-0.5,res=x)))
