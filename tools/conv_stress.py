"""Do the LDS-DMA GEMM / conv kernels (counted vmcnt waits, out-of-range lanes for padding taps and tile tails) give the
idle-chip bits when several streams keep the memory system busy?  (Round 3: dd_xattn320 did not, until its counted waits
stopped spanning fully out-of-range instructions.)"""
import sys, torch
sys.path.insert(0, "/root/repo")
from dualdiff_amd import ops as O
dt = torch.float16; dev = "cuda"
g = torch.Generator(device=dev).manual_seed(3)
r = lambda *s, sc=1.0: (torch.randn(*s, generator=g, device=dev) * sc).to(dt)
cases = []
# (m, h, w, cin, cout, tile, split): implicit-GEMM conv tiles with many fully-padded tap rows (small images), band conv, direct conv
for (m, h, w, cin, cout, tile, split) in ((12, 28, 50, 320, 320, 28, 1), (12, 28, 50, 320, 320, 39, 1), (12, 14, 25, 640, 640, 12, 1),
                                          (12, 7, 13, 1280, 1280, 31, 4), (12, 4, 7, 1280, 1280, 37, 5), (12, 4, 7, 640, 1280, 13, 3),
                                          (3, 14, 25, 320, 640, 20, 1), (12, 28, 50, 960, 320, 0, 0)):
    x = r(m * h * w, cin); wt = r(cout, 9 * cin, sc=(9 * cin) ** -0.5); b = r(cout)
    fn = (lambda x=x, wt=wt, b=b, m=m, h=h, w=w, tile=tile, split=split: O.conv3x3(x, wt, b, m, h, w, tile=tile, split_k=split))
    cases.append((fn, fn()))
for (rows, n, k, tile, split) in ((1092, 1280, 1280, 13, 1), (16800, 320, 320, 27, 1), (77, 72, 200 * 8, 18, 1), (1003, 328, 2048, 12, 2)):
    a = r(rows, k); wt = r(n, k, sc=k ** -0.5)
    fn = (lambda a=a, wt=wt, tile=tile, split=split: O.gemm(a, wt, tile=tile, split_k=split))
    cases.append((fn, fn()))
torch.cuda.synchronize()
streams = [torch.cuda.Stream() for _ in range(3)]
big = torch.empty(1 << 28, dtype=torch.float16, device=dev)
outs = []
for it in range(30):
    for ci, (fn, ref) in enumerate(cases):
        with torch.cuda.stream(streams[(ci + it) % 3]):
            outs.append((ci, fn(), ref))
    if it % 2 == 0:
        big.add_(1)
torch.cuda.synchronize()
bad = {}
for ci, y, ref in outs:
    if not torch.equal(y, ref):
        bad[ci] = bad.get(ci, 0) + 1
print("launches with wrong bits per case (of 30 each):", bad if bad else "none")
