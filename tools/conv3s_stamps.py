"""DIAGNOSTIC (needs DD_HIP_LIB=<a library built with -DDD_DBG_STAMP>, tools/build_dbg_libs.sh): phase timeline of the
direct small-image conv — per workgroup, cycles from kernel entry to: tables built, prologue DMAs issued, first 9 steps
done, main loop done, tile stored; plus the in-kernel clock (s_memtime over s_memrealtime)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dualdiff_amd import ops as O
dt = torch.float16
dev = torch.device("cuda")
O.workspace(512 << 20, dev)
for (b, h, w, c) in ((12, 28, 50, 320), (12, 14, 25, 640), (12, 7, 13, 1280), (12, 4, 7, 1280), (48, 28, 50, 320), (48, 14, 25, 640)):
    rows = b * h * w
    x = (torch.randn(rows, c, device=dev)).to(dt)
    wt = (torch.randn(c, 9 * c, device=dev) * (9 * c) ** -0.5).to(dt)
    bi = torch.randn(c, device=dev).to(dt)
    for _ in range(20):
        O.conv3x3(x, wt, bi, b, h, w)
    torch.cuda.synchronize()
    ws = O.workspace(1, dev)
    tail = ws.view(torch.int64)[-(1 << 17):].clone()
    tail.zero_(); ws.view(torch.int64)[-(1 << 17):].zero_()
    torch.cuda.synchronize()
    O.conv3x3(x, wt, bi, b, h, w)
    torch.cuda.synchronize()
    raw = ws.view(torch.int64)[-(1 << 17):].cpu()
    st = raw[:65536].reshape(-1, 8)
    sg = raw[65536:].reshape(-1, 8)
    sg = sg[sg[:, 7] != 0]
    st = st[st[:, 7] != 0]
    n = st.shape[0]
    t = (st[:, 1:6] - st[:, 0:1]).double()
    real = (st[:, 7] - st[:, 6]).double() * 10.0           # ns (100 MHz)
    clk = (st[:, 5] - st[:, 0]).double() / real             # GHz
    span = (st[:, 7].max() - st[:, 6].min()).item() * 10.0
    print("conv %dx%d C=%d: %d workgroups, kernel span %.1f us, in-kernel clock median %.2f GHz" %
          (h, w, c, n, span / 1e3, clk.median().item()))
    print("   start spread %.1f us" % ((st[:, 6].max() - st[:, 6].min()).item() / 100.0))
    names = ("tables", "prologue issued", "first 9 steps", "loop done", "stored")
    for i, nm in enumerate(names):
        print("   %-16s median %8.0f cyc = %5.1f us  (p10 %8.0f, p90 %8.0f)" % (nm, t[:, i].median().item(),
              t[:, i].median().item() / clk.median().item() / 1e3, t[:, i].quantile(0.1).item(), t[:, i].quantile(0.9).item()))
    if sg.shape[0]:
        for wv, nm in ((0, "early wave 0"), (4, "late  wave 4")):
            g = sg[sg[:, 6] == wv]
            if not g.shape[0] or g[0, 5] == 0:
                continue
            per = g[:, :5].double() / g[:, 5:6].double()
            med = per.median(dim=0).values
            lab = (("(-)", "weight reads + DMA issue", "MFMAs(s) + gathers(s+1)") if wv == 0 else
                   ("MFMAs(s-1) + gathers(s)", "DMA issue", "weight reads, lgkmcnt(0)"))
            print("   %s, cycles per steady-state step (median over workgroups): vmcnt wait %4.0f | barrier %4.0f | %s %4.0f | %s %4.0f | %s %4.0f | sum %4.0f" %
                  (nm, med[0], med[1], lab[0], med[2], lab[1], med[3], lab[2], med[4], med.sum().item()))
