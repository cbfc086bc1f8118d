import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dualdiff_amd import ops as O
import tests.test_gemm4_gpu as T
DEV = torch.device("cuda:0")
dtype, tile, twin, rows, n, k = torch.bfloat16, 78, 28, 67200, 320, 320

def analyse(got, want, ref, tag):
    ne = (got != want)
    if not ne.any():
        print(tag, "equal"); return
    r = ne.any(dim=1).nonzero().flatten(); c = ne.any(dim=0).nonzero().flatten()
    print(tag, "ne elements", int(ne.sum()), "rows", r.numel(), int(r.min()), int(r.max()), "cols", c.numel(), int(c.min()), int(c.max()),
          "row tiles(160)", sorted(set((r // 160).tolist()))[:20])
    for nm, y in (("gemm4", got), ("twin", want)):
        e = (y.float() - ref).abs()
        print("   ", nm, "max err", float(e.max()), "rows>0.06:", int((e > 0.06).any(dim=1).sum()))

def body(order):
    x, w, g = T._mk(rows, n, k, dtype, 3)
    bias = torch.randn(n, device=DEV, generator=g).to(dtype)
    res = torch.randn(rows, n, device=DEV, generator=g).to(dtype)
    ref = x.float() @ w.float().t()
    for kw in ({}, {"bias": bias}, {"bias": bias, "res": res, "alpha": 0.5}, {"res": res, "epilogue": O.DD_EPI_SILU}):
        kw = dict(kw)
        b = kw.pop("bias", None)
        if order == "tile_first":
            got = O.gemm(x, w, b, tile=tile, split_k=1, **kw); want = O.gemm(x, w, b, tile=twin, split_k=1, **kw)
        else:
            want = O.gemm(x, w, b, tile=twin, split_k=1, **kw); got = O.gemm(x, w, b, tile=tile, split_k=1, **kw)
        r = x.float() @ w.float().t()
        if b is not None: r = r + b.float()
        if "alpha" in kw: r = r * 0.5
        if "res" in kw: r = r + res.float()
        if "epilogue" in kw: r = torch.nn.functional.silu(r)
        analyse(got, want, r, "%s %s" % (order, sorted(kw)))

body("tile_first")
body("twin_first")
