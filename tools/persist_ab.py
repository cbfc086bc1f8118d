"""A/B of the persistent tile walk (DD_PERSIST=0 turns it off; run once per setting): times and a checksum of the
results of the small-K multi-round GEMMs of the step."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dualdiff_amd import ops as O
from tools.attn_variants import graph_time
dt = torch.float16
torch.manual_seed(0)


def r(*s, scale=1.0):
    return (torch.randn(*s, device="cuda") * scale).to(dt)


print("DD_PERSIST =", os.environ.get("DD_PERSIST", "(default on)"))
for (rows, c) in ((16800, 320), (4200, 640), (1092, 1280)):
    x, bi = r(rows, c), r(c)
    w3 = r(3 * c, c, scale=c ** -0.5)
    w8, b8 = r(8 * c, c, scale=c ** -0.5), r(8 * c)
    x4, w4 = r(rows, 4 * c), r(c, 4 * c, scale=(4 * c) ** -0.5)
    cases = (("qkv hm", lambda t: O.gemm(x, w3, None, head_major=(c // 8, 8, 0.2), tile=t), (19, 15, 12, 25, 16, 20, 28, 13, 14)),
             ("geglu", lambda t: O.gemm(x, w8, b8, epilogue=O.DD_EPI_GEGLU, tile=t), (12, 25, 16, 20, 29, 14, 24)),
             ("ff2+res", lambda t: O.gemm(x4, w4, bi, res=x, tile=t, split_k=1), (19, 15, 12, 16, 28)),
             ("CxC+res", lambda t: O.gemm(x, w4[:, :c].contiguous(), bi, res=x, tile=t, split_k=1), (19, 15, 12, 28)))
    for name, fn, tiles in cases:
        ref = fn(tiles[0]).float()
        out = []
        for t in tiles:
            y = fn(t)
            ok = torch.equal(y.float(), ref)
            tt = graph_time(lambda: fn(t))
            out.append("%d:%.1f%s" % (t, tt, "" if ok else "(!= tile %d: %.2e)" % (tiles[0], (y.float() - ref).abs().max().item())))
        print("%-8s %5dx%d  sum %.6e | %s" % (name, rows, c, ref.double().sum().item(), "  ".join(out)))
