"""HIP-graph chain time (hot / cold weights) of the GEGLU projections of the step, per library build (DD_HIP_LIB):
the one-scene shapes and their four-scene forms.  us per launch."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dualdiff_amd import ops as O
from tools._timing import graph_time
dt, dev = torch.float16, torch.device("cuda")
O.workspace(512 << 20, dev)
SHAPES = [(16800, 2560, 320, 75), (4200, 5120, 640, 75), (1092, 10240, 1280, 75), (67200, 2560, 320, 75), (16800, 5120, 640, 75), (4368, 10240, 1280, 75)]
out = "%-22s" % os.path.basename(os.environ.get("DD_HIP_LIB", "product"))
for rows, n, k, t in SHAPES:
    x = torch.randn(rows, k, device=dev).to(dt)
    nbuf = max(2, int(600e6 // (n * k * 2)) + 1)
    ws = [(torch.randn(n, k, device=dev) * k ** -0.5).to(dt) for _ in range(nbuf)]
    bi = torch.randn(n, device=dev).to(dt)
    st = {"i": 0}

    def cold():
        st["i"] += 1
        return O.gemm(x, ws[st["i"] % nbuf], bi, tile=t, epilogue=O.DD_EPI_GEGLU)
    hot = min(graph_time(lambda: O.gemm(x, ws[0], bi, tile=t, epilogue=O.DD_EPI_GEGLU), n=8) for _ in range(3))
    cld = min(graph_time(cold, n=nbuf) for _ in range(2))
    out += " | %dx%dx%d %5.1f/%5.1f" % (rows, n, k, hot, cld)
print(out, flush=True)
