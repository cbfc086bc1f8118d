"""GPU diagnostic: persistent pipelined tiles vs their dd_gemm2 twins, repeated; prints where results differ."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dualdiff_amd import ops as O
dev = torch.device("cuda")
torch.manual_seed(0)


def diff(a, b, tag):
    if torch.equal(a, b):
        return 0
    ne = (a != b)
    idx = ne.nonzero()
    print("  MISMATCH %s: %d of %d elements; first %s; a=%s b=%s" % (tag, int(ne.sum()), ne.numel(), idx[:6].tolist(),
          a[ne][:6].tolist(), b[ne][:6].tolist()))
    if idx.shape[1] == 3:
        print("    planes", sorted(set(idx[:, 0].tolist()))[:30], "rows range", int(idx[:, 1].min()), int(idx[:, 1].max()),
              "d range", int(idx[:, 2].min()), int(idx[:, 2].max()))
    else:
        print("    rows range", int(idx[:, 0].min()), int(idx[:, 0].max()), "cols", sorted(set(idx[:, 1].tolist()))[:40])
    return 1


for dt in (torch.float16, torch.bfloat16):
    for tile, twin, rows, n, k in ((75, 44, 67200, 960, 320), (72, 52, 67200, 960, 320), (75, 44, 50001, 328, 640), (78, 28, 67200, 320, 320),
                                   (73, 52, 67200, 320, 320)):
        x = torch.randn(rows, k, device=dev).to(dt)
        w = (torch.randn(n, k, device=dev) * k ** -0.5).to(dt)
        bias = torch.randn(n, device=dev).to(dt)
        res = torch.randn(rows, n, device=dev).to(dt)
        bad = 0
        for rep in range(4):
            for name, kw in (("plain", {}), ("bias", {"bias": bias}), ("res", {"bias": bias, "res": res, "alpha": 0.5})):
                kw = dict(kw)
                b = kw.pop("bias", None)
                bad += diff(O.gemm(x, w, b, tile=tile, split_k=1, **kw), O.gemm(x, w, b, tile=twin, split_k=1, **kw), "%s t%d %s rep%d" % (dt, tile, name, rep))
            if n % 40 == 0:
                for planes in (0, 1, n // 40 // 3, n // 40):
                    hm = (40, planes, 0.228)
                    bad += diff(O.gemm(x, w, None, tile=tile, split_k=1, head_major=hm), O.gemm(x, w, None, tile=twin, split_k=1, head_major=hm),
                                "%s t%d hm planes=%d rep%d" % (dt, tile, planes, rep))
                    bad += diff(O.gemm(x, w, bias, tile=tile, split_k=1, head_major=hm), O.gemm(x, w, bias, tile=twin, split_k=1, head_major=hm),
                                "%s t%d hm+bias planes=%d rep%d" % (dt, tile, planes, rep))
        print("%s tile %d vs %d  %dx%dx%d: %d mismatching calls" % (dt, tile, twin, rows, n, k, bad), flush=True)
