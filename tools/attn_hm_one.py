"""The model's self-attention call form (head-major Q/K/V planes, Q pre-scaled): a few plain launches for rocprofv3.
Usage: python3 tools/attn_hm_one.py [b l h d] [launches]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dualdiff_amd import ops as O
b, l, h, d = [int(x) for x in sys.argv[1:5]] if len(sys.argv) > 4 else (12, 1400, 8, 40)
n = int(sys.argv[5]) if len(sys.argv) > 5 else 10
dt = torch.float16
qkv = torch.randn(3 * h, b * l, d, device="cuda").to(dt)
out = torch.empty(b * l, h * d, device="cuda", dtype=dt)
for _ in range(n):
    O.attention(qkv[:h], qkv[h:2 * h], qkv[2 * h:], b, l, l, h, d, out=out, q_prescaled=True)
torch.cuda.synchronize()
print("ok", float(out.float().abs().mean()))
