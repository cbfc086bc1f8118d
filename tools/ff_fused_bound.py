"""Fused feed-forward: the measurement behind the decision not to build it (VERDICT r1-r3 asked for it three times).

A fused FF kernel (x tile resident -> GEGLU(x W1^T) chunk by chunk -> accumulate into out with W2) owns WHOLE rows of the
320 / 640 / 1280-wide output, so its row tile is bounded by LDS and registers: x tile (BM x C x 2 B) + the hidden chunk
(BM x 160 x 2 B) + a weight ring, with the BM x C fp32 accumulator in registers — BM <= 64 at C = 320 (140 KB), less
beyond.  Every row tile then stages ALL of W1 and W2.  This tool times, on the L0 shape of the step (16800 x 320,
hidden 1280):
  (a) the production pair: GEGLU GEMM + out GEMM with the tuned tiles (what the step runs);
  (b) the same two GEMMs forced onto 64-ROW tiles — the staging pattern of the fused kernel's two phases (each 64-row
      block pulls the whole weight matrix through L2 -> LDS), i.e. a LOWER bound of its staging time: the fused kernel
      saves the round trip of the hidden tensor and x's re-staging per column tile, nothing else;
  (c) the round trip it would save: writing and re-reading the 16800 x 1280 hidden tensor at the measured copy rate.
It prints the staged bytes of each form next to the times.  python tools/ff_fused_bound.py [fp16|bf16]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dualdiff_amd import ops as O
from tools._timing import graph_time
dt = torch.bfloat16 if (len(sys.argv) > 1 and sys.argv[1] == "bf16") else torch.float16
dev = torch.device("cuda")


def r(*s, scale=1.0):
    return (torch.randn(*s, device=dev) * scale).to(dt)


for (rows, c) in ((16800, 320), (4200, 640), (1092, 1280)):
    hid = 4 * c
    x, w1, b1 = r(rows, c), r(2 * hid, c, scale=c ** -0.5), r(2 * hid)
    w2, b2 = r(c, hid, scale=hid ** -0.5), r(c)
    h = O.gemm(x, w1, b1, epilogue=O.DD_EPI_GEGLU)
    y = O.gemm(h, w2, b2, res=x)
    t1 = graph_time(lambda: O.gemm(x, w1, b1, epilogue=O.DD_EPI_GEGLU))
    t2 = graph_time(lambda: O.gemm(h, w2, b2, res=x))
    # 64-row tiles: id 14 = 64x128/dma3 (GEGLU-capable: 64 gated columns per tile), id 15 = 64x64/dma3
    f1 = graph_time(lambda: O.gemm(x, w1, b1, epilogue=O.DD_EPI_GEGLU, tile=14, split_k=1))
    f2 = graph_time(lambda: O.gemm(h, w2, b2, res=x, tile=14, split_k=1))
    hh = torch.empty_like(h)
    tc = graph_time(lambda: hh.copy_(h))
    tiles64 = (rows + 63) // 64
    w_bytes = (w1.numel() + w2.numel()) * 2
    st_fused = tiles64 * (64 * c * 2 + w_bytes)                     # x tile once + every weight byte, per 64-row tile
    print("FF %5d x %4d (hidden %d), %s:" % (rows, c, hid, str(dt).split(".")[-1]))
    print("  (a) production tiles: GEGLU %.1f us + out %.1f us = %.1f us" % (t1, t2, t1 + t2))
    print("  (b) both GEMMs on 64-row tiles (the fused kernel's staging pattern): %.1f + %.1f = %.1f us" % (f1, f2, f1 + f2))
    print("  (c) hidden-tensor round trip a fusion would save: %.1f MB written + read, copy kernel %.1f us" % (h.numel() * 2 / 1e6 * 2, tc))
    print("      fused kernel >= (b) - (c) - x re-staging ~ %.1f us vs (a) %.1f us;  it would stage %.0f MB through L2 -> LDS "
          "(%d row tiles x %.1f MB of weights)" % (f1 + f2 - tc, t1 + t2, st_fused / 1e6, tiles64, w_bytes / 1e6))
