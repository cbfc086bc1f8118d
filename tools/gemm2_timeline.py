"""DIAGNOSTIC (DD_HIP_LIB=<-DDD_DBG_STAMP build>, DD_DBG_STAMP_WS=1; tools/gemm2_timeline.sh): where one launch of the
dominant dense class spends its time.  Per shape, with the production tile of the tracked table, weights cold (rotation
over 600 MB of weight buffers, as in the step) and hot (same buffer again):
  * the launch's duration by HIP events (back-to-back launches of a graph chain, the way the step runs it);
  * from the per-workgroup s_memrealtime / cycle stamps (thread 0 of every workgroup): when the FIRST and the LAST workgroup
    started and ended relative to the first start (dispatch skew, total span), and the median workgroup's phases in
    microseconds: entry -> address tables built -> prologue DMAs issued -> first K-step consumed -> K loop done -> stored."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dualdiff_amd import ops as O
from tools._timing import graph_time
dt = torch.float16
dev = torch.device("cuda")
O.workspace(512 << 20, dev)


def r(*s, scale=1.0):
    return (torch.randn(*s, device=dev) * scale).to(dt)


def stamps(fn):
    ws = O.workspace(1, dev)
    ws.view(torch.int64)[-(1 << 17):].zero_()
    torch.cuda.synchronize()
    fn()
    torch.cuda.synchronize()
    st = ws.view(torch.int64)[-(1 << 17):].cpu().reshape(-1, 8)
    return st[st[:, 7] != 0]


def show(name, rows, n, k, res=True, tile=0):
    x, bi = r(rows, k), r(n)
    xr = r(rows, n)
    nbuf = max(3, int(600e6 // (n * k * 2)) + 1)
    ws_ = [r(n, k, scale=k ** -0.5) for _ in range(nbuf)]
    state = {"i": 0}

    def hot():
        return O.gemm(x, ws_[0], bi, res=xr if res else None, tile=tile, split_k=1 if tile else 0)

    def cold():
        state["i"] += 1
        return O.gemm(x, ws_[state["i"] % nbuf], bi, res=xr if res else None, tile=tile, split_k=1 if tile else 0)
    for _ in range(3):
        hot()
    t_hot = graph_time(hot, n=nbuf)
    t_cold = graph_time(cold, n=nbuf)
    for label, fn in (("cold", cold), ("hot", hot)):
        for _ in range(2):
            fn()
        st = stamps(fn)
        if st.shape[0] == 0:
            print(name, "no stamps (is DD_HIP_LIB the stamp build?)"); return
        start = (st[:, 6] - st[:, 6].min()).double() / 100.0            # us, 100 MHz real-time counter
        end = (st[:, 7] - st[:, 6].min()).double() / 100.0
        cyc = (st[:, 1:6] - st[:, 0:1]).double()
        real = (st[:, 7] - st[:, 6]).double() / 100.0
        clk = ((st[:, 5] - st[:, 0]).double() / real.clamp_min(1e-3)).median().item()          # MHz -> cycles per us
        ph = [(cyc[:, i] / clk).median().item() for i in range(5)]
        print("%-22s %-4s events %5.1f us | %3d WGs: last start +%4.1f us, first end %4.1f, last end %5.1f us (clk %.2f GHz) | "
              "median WG: tables %4.1f  issued %4.1f  1st K-step %4.1f  loop done %5.1f  stored %5.1f us  (K-steps %d: %.2f us each)"
              % (name, label, t_cold if label == "cold" else t_hot, st.shape[0], start.max().item(), end.min().item(),
                 end.max().item(), clk / 1e3, *ph, (k + 63) // 64, (ph[3] - ph[2]) / max(1, (k + 63) // 64 - 1)))


TILES = [int(t) for t in os.environ.get("DD_TIMELINE_TILES", "0").split(",")]       # 0 = the tracked table's tile
for (rows, n, k) in ((1092, 1280, 1280), (336, 1280, 1280), (4200, 640, 640), (4200, 640, 3200), (16800, 320, 320)):
    for t in TILES:
        show("%dx%dx%d%s" % (rows, n, k, " t%d" % t if t else ""), rows, n, k, tile=t)
