"""DIAGNOSTIC (DD_HIP_LIB=<-DDD_DBG_STAMP build>, DD_DBG_STAMP_WS=1): per-workgroup phase timeline of the LDS-DMA GEMM
family on the step's main shapes: entry -> tables -> prologue issued -> first K-step -> loop done -> stored."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dualdiff_amd import ops as O
dt = torch.float16
dev = torch.device("cuda")
O.workspace(512 << 20, dev)


def r(*s, scale=1.0):
    return (torch.randn(*s, device=dev) * scale).to(dt)


def show(name, fn):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    ws = O.workspace(1, dev)
    ws.view(torch.int64)[-(1 << 17):].zero_()
    torch.cuda.synchronize()
    fn()
    torch.cuda.synchronize()
    st = ws.view(torch.int64)[-(1 << 17):].cpu().reshape(-1, 8)
    st = st[st[:, 7] != 0]
    if st.shape[0] == 0:
        print(name, "no stamps"); return
    t = (st[:, 1:6] - st[:, 0:1]).double()
    real = (st[:, 7] - st[:, 6]).double() * 10.0
    clk = ((st[:, 5] - st[:, 0]).double() / real).median().item()
    span = (st[:, 7].max() - st[:, 6].min()).item() / 100.0
    med = [t[:, i].median().item() for i in range(5)]
    print("%-28s %4d WGs span %6.1f us clk %.2f | tables %5.0f  issued %5.0f  1st step %6.0f  loop %6.0f  stored %6.0f cyc | last start +%.1f us" %
          (name, st.shape[0], span, clk, *med, (st[:, 6].max() - st[:, 6].min()).item() / 100.0))


for (b, h, w, c) in ((12, 28, 50, 320), (12, 14, 25, 640), (12, 7, 13, 1280)):
    rows = b * h * w
    x = r(rows, c)
    bi = r(c)
    if c == 320:
        wt = r(c, 9 * c, scale=(9 * c) ** -0.5)
        show("conv3 %dx%dx%d" % (rows, c, 9 * c), lambda: O.conv3x3(x, wt, bi, b, h, w))
    w8, b8 = r(8 * c, c, scale=c ** -0.5), r(8 * c)
    show("geglu %dx%dx%d" % (rows, 8 * c, c), lambda: O.gemm(x, w8, b8, epilogue=O.DD_EPI_GEGLU))
    x4, w4 = r(rows, 4 * c), r(c, 4 * c, scale=(4 * c) ** -0.5)
    show("ff2 %dx%dx%d" % (rows, c, 4 * c), lambda: O.gemm(x4, w4, bi, res=x))
    w3 = r(3 * c, c, scale=c ** -0.5)
    show("qkv %dx%dx%d" % (rows, 3 * c, c), lambda: O.gemm(x, w3, None, head_major=(c // 8, 8, 0.2)))
    w1 = r(c, c, scale=c ** -0.5)
    show("CxC+res %dx%dx%d" % (rows, c, c), lambda: O.gemm(x, w1, bi, res=x))
