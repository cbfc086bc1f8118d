"""Fixed cost vs per-step cost of the direct small-image conv: time (hot, graph chain) over the number of 64-channel
chunks a workgroup walks (cin = 64 * chunks, split-K off), at the 7x13 and 4x7 levels.  The intercept is prologue +
pipeline fill + epilogue, the slope 9 (chunk, tap) steps.  python tools/conv3s_fixed_cost.py [tile]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dualdiff_amd import ops as O
from tools.attn_variants import graph_time
tile = int(sys.argv[1]) if len(sys.argv) > 1 else 31
dt = torch.float16
for (h, w) in ((7, 13), (4, 7)):
    row = []
    for chunks in (1, 2, 4, 8, 16):
        cin = 64 * chunks
        x = torch.randn(12 * h * w, cin, device="cuda").to(dt)
        wt = (torch.randn(1280, 9 * cin, device="cuda") * 0.02).to(dt)
        b = torch.randn(1280, device="cuda").to(dt)
        us = graph_time(lambda: O.conv3x3(x, wt, b, 12, h, w, tile=tile, split_k=1))
        row.append((chunks, us))
    slope = (row[-1][1] - row[0][1]) / (row[-1][0] - row[0][0])
    print("%dx%d tile %d: " % (h, w, tile) + "  ".join("%d:%.1fus" % r for r in row) +
          "   -> %.2f us per chunk (9 steps), intercept %.1f us" % (slope, row[0][1] - slope * row[0][0]))
