"""GPU diagnostic: LayerNorm-emitting and GEGLU persistent tiles vs their dd_gemm2 twins; difference statistics."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dualdiff_amd import ops as O
dev = torch.device("cuda")
torch.manual_seed(0)


def stats(a, b, tag):
    ne = a != b
    d = (a.float() - b.float()).abs()
    rows = ne.any(dim=1).nonzero().flatten()
    print("%s: %d of %d differ, max abs %.3e (values up to %.2f); rows %s ... cols of first bad row %s" % (
        tag, int(ne.sum()), ne.numel(), d.max().item(), a.float().abs().max().item(), rows[:8].tolist(),
        ne[rows[0]].nonzero().flatten()[:12].tolist() if rows.numel() else []), flush=True)


for dt in (torch.float16, torch.bfloat16):
    for rows, k in ((67200, 320), (20000, 320), (67200, 1280)):
        n = 320
        x = torch.randn(rows, k, device=dev).to(dt)
        w = (torch.randn(n, k, device=dev) * k ** -0.5).to(dt)
        bias, ga, be = (torch.randn(n, device=dev).to(dt) for _ in range(3))
        res = torch.randn(rows, n, device=dev).to(dt)
        o = {}
        for t in (74, 40):
            y = O.gemm(x, w, bias, res=res, ln_out=(ga, be, 1e-5), tile=t)
            o[t] = (y.clone(), y._ln_out.clone())
        stats(o[74][0], o[40][0], "%s LN rows=%d k=%d out   " % (dt, rows, k))
        stats(o[74][1], o[40][1], "%s LN rows=%d k=%d ln_out" % (dt, rows, k))
        ref = torch.nn.functional.layer_norm(o[40][0].float(), (n,), ga.float(), be.float(), 1e-5)
        print("   vs torch: t74 %.3e  t40 %.3e" % ((o[74][1].float() - ref).abs().max().item(), (o[40][1].float() - ref).abs().max().item()))
    for rows, n, k in ((67200, 2560, 320), (33001, 1296, 640)):
        x = torch.randn(rows, k, device=dev).to(dt)
        w = (torch.randn(n, k, device=dev) * k ** -0.5).to(dt)
        bias = torch.randn(n, device=dev).to(dt)
        for b in (None, bias):
            stats(O.gemm(x, w, b, tile=75, epilogue=O.DD_EPI_GEGLU), O.gemm(x, w, b, tile=44, epilogue=O.DD_EPI_GEGLU),
                  "%s GEGLU %dx%dx%d bias=%s" % (dt, rows, n, k, b is not None))
