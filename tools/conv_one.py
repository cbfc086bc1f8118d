"""One conv / GEMM shape with the tuned (or given) tile, a few plain launches — target of rocprofv3 --pmc passes.
Usage: python3 tools/conv_one.py cin cout h w [tile split]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dualdiff_amd import ops as O
cin, cout, h, w = [int(x) for x in sys.argv[1:5]]
tile, split = (int(sys.argv[5]), int(sys.argv[6])) if len(sys.argv) > 6 else (0, 0)
M = 12; dt = torch.bfloat16
x = torch.randn(M * h * w, cin, device="cuda").to(dt); wt = (torch.randn(cout, 9 * cin, device="cuda") * 0.02).to(dt)
b = torch.randn(cout, device="cuda").to(dt)
for _ in range(10):
    y = O.conv3x3(x, wt, b, M, h, w, tile=tile, split_k=split)
torch.cuda.synchronize()
print("ok", float(y.float().abs().mean()))
