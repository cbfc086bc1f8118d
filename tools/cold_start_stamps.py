"""DIAGNOSTIC (DD_HIP_LIB=<-DDD_DBG_STAMP build>, DD_DBG_STAMP_WS=1): phase stamps of a GEMM launched right after a whole
eager denoising step (its code and weights as cold as inside the step) against the same launch repeated immediately."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from dualdiff_amd import ops as O
from dualdiff_amd.pipeline.pipeline_bev_controlnet import BEVDenoiser
dt, dev = torch.float16, torch.device("cuda:0")
O.workspace(512 << 20, dev)
unet, cns = bench.build_models(dt, dev)
den = BEVDenoiser(unet, cns, use_graph=False, parallel_branches=False)


def r(*s, scale=1.0):
    return (torch.randn(*s, device=dev) * scale).to(dt)


def stamps():
    ws = O.workspace(1, dev)
    st = ws.view(torch.int64)[-(1 << 17):].cpu().reshape(-1, 8)
    st = st[st[:, 7] != 0]
    t = (st[:, 1:6] - st[:, 0:1]).double()
    span = (st[:, 7].max() - st[:, 6].min()).item() / 100.0
    return [t[:, i].median().item() for i in range(5)], span, st.shape[0]


def clear():
    O.workspace(1, dev).view(torch.int64)[-(1 << 17):].zero_()


with torch.no_grad():
    den.set_inputs(*bench.synthetic_inputs(1, dt, dev, 1))
    den.step(0); den.step(1)
    for (rows, n, k, tile) in ((1092, 3840, 1280, 44), (4200, 1920, 640, 20), (1092, 1280, 1280, 13), (16800, 320, 320, 28)):
        a, w = r(rows, k), r(n, k, scale=k ** -0.5)
        fn = lambda: O.gemm(a, w, None, tile=tile)
        for _ in range(3):
            fn()
        out = []
        for mode in ("after a step", "repeated"):
            if mode == "after a step":
                den._step_body()
            torch.cuda.synchronize(); clear(); torch.cuda.synchronize()
            fn(); torch.cuda.synchronize()
            med, span, nwg = stamps()
            out.append("%-12s span %6.1f us | tables %5.0f issued %5.0f 1st-step %6.0f loop %6.0f stored %6.0f" % ((mode, span) + tuple(med)))
        print("gemm %dx%dx%d tile %d (%d WGs)\n   %s\n   %s" % (rows, n, k, tile, nwg, out[0], out[1]))
