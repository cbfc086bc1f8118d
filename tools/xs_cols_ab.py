"""(Needs docs/experiments/r06_xs_cols.patch.txt applied: the product has no DD_XS_COLS.)  A/B of the XCD-share column slicing (DD_XS_COLS = 1 row-major / 2 / 4 / 8 / 0 = the host's rule): hot and COLD (weights
rotated over 600 MB, as in the step) HIP-graph-chain time per launch of the dense shapes whose weight matrix does not fit an L2."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dualdiff_amd import ops as O
from tools._timing import graph_time
dt, dev = torch.float16, torch.device("cuda")
O.workspace(512 << 20, dev)
SHAPES = [(4200, 5120, 640, True, 75), (1092, 10240, 1280, True, 75), (16800, 5120, 640, True, 75), (16800, 5120, 640, True, 50), (4368, 10240, 1280, True, 75),
          (4200, 1920, 640, False, 75), (16800, 1920, 640, False, 75), (1092, 3840, 1280, False, 75), (4368, 3840, 1280, False, 75),
          (4200, 640, 2560, False, 72), (16800, 640, 2560, False, 75), (1092, 1280, 5120, False, 73), (4368, 1280, 5120, False, 75)]
out = "xs_cols=%-2s" % os.environ.get("DD_XS_COLS", "0")
for rows, n, k, geglu, t in SHAPES:
    x = torch.randn(rows, k, device=dev).to(dt)
    nbuf = max(2, int(600e6 // (n * k * 2)) + 1)
    ws = [(torch.randn(n, k, device=dev) * k ** -0.5).to(dt) for _ in range(nbuf)]
    bi = torch.randn(n, device=dev).to(dt)
    kw = {"epilogue": O.DD_EPI_GEGLU} if geglu else {"split_k": 1}
    st = {"i": 0}

    def cold():
        st["i"] += 1
        return O.gemm(x, ws[st["i"] % nbuf], bi, tile=t, **kw)
    hot = min(graph_time(lambda: O.gemm(x, ws[0], bi, tile=t, **kw), n=8) for _ in range(3))
    cld = min(graph_time(cold, n=nbuf) for _ in range(2))
    out += " | %dx%dx%d%s t%d %5.1f/%5.1f" % (rows, n, k, "g" if geglu else "", t, hot, cld)
print(out, flush=True)
