"""One attention shape, one variant, a few plain launches — the target of rocprofv3 --pmc passes.
Usage: python3 tools/attn_one.py <variant> [b lq lk h d] [launches]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dualdiff_amd import ops as O
var = int(sys.argv[1]) if len(sys.argv) > 1 else 0
b, lq, lk, h, d = [int(x) for x in sys.argv[2:7]] if len(sys.argv) > 6 else (12, 1400, 1400, 8, 40)
n = int(sys.argv[7]) if len(sys.argv) > 7 else 10
dt = torch.bfloat16
q = torch.randn(b * lq, h * d, device="cuda").to(dt); k = torch.randn(b * lk, h * d, device="cuda").to(dt)
v = torch.randn(b * lk, h * d, device="cuda").to(dt); out = torch.empty_like(q)
for _ in range(n):
    O.attention(q, k, v, b, lq, lk, h, d, out=out, variant=var)
torch.cuda.synchronize()
print("ok", float(out.float().abs().mean()))
