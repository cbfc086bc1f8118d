"""A/B of the pipelined dense family (dd_gemm3_kernel, tiles 72+) against the dd_gemm2_kernel tiles of the tracked table,
in ONE process, interleaved rounds (guide rule 24): per shape and tile the median over rounds of a graph chain of
launches, weights HOT (same buffer again) and COLD (rotation over ~600 MB of weight buffers, the state the step runs in).
    python tools/gemm3_ab.py [rounds] > gpurun_out/r05_gemm3_ab.txt"""
import os, sys, statistics, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dualdiff_amd import ops as O
from tools._timing import graph_time

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
dt = torch.float16
dev = torch.device("cuda")
O.workspace(512 << 20, dev)


def r(*s, scale=1.0):
    return (torch.randn(*s, device=dev) * scale).to(dt)


# (rows, n, k, kind, incumbent tile of the tracked table, challengers)
SHAPES = [
    (1092, 1280, 1280, "res", 52, (72, 73, 74)),
    (336, 1280, 1280, "res", 52, (72, 76, 77, 79)),
    (4200, 640, 640, "res", 52, (72, 73)),
    (4200, 640, 2560, "res", 52, (72, 73)),
    (1092, 1280, 5120, "res", 52, (72, 73, 74)),
    (1092, 3840, 1280, "hm", 44, (75, 72, 73)),
    (4200, 1920, 640, "hm", 44, (75, 72)),
    (16800, 320, 320, "res", 28, (78, 72)),
    (16800, 960, 320, "hm", 28, (78, 75, 72)),
    (1176, 640, 768, "plain", 52, (72, 76, 77)),
    (12, 1280, 1280, "plain", 60, (76, 77, 79)),
    (1092, 10240, 1280, "geglu", 44, (75,)),
]


def bench(rows, n, k, kind, tiles):
    x = r(rows, k)
    geglu = kind == "geglu"
    nw = n if not geglu else n          # n counts both halves for geglu rows of W
    bi = r(nw)
    xr = r(rows, n) if kind == "res" else None
    nbuf = max(3, int(600e6 // (nw * k * 2)) + 1)
    ws_ = [r(nw, k, scale=k ** -0.5) for _ in range(nbuf)]
    out = {}
    state = {"i": 0}

    def call(w, t):
        if geglu:
            return O.gemm(x, w, bi, epilogue=O.DD_EPI_GEGLU, tile=t)
        if kind == "hm":
            d = 40 if n // 3 == 320 else (80 if n // 3 == 640 else 160)
            return O.gemm(x, w, None, head_major=(d, 8, 0.3), tile=t, split_k=1)
        return O.gemm(x, w, bi, res=xr, tile=t, split_k=1)

    res = {t: {"hot": [], "cold": []} for t in tiles}
    for _ in range(rounds):
        for t in tiles:
            try:
                def hot():
                    return call(ws_[0], t)

                def cold():
                    state["i"] += 1
                    return call(ws_[state["i"] % nbuf], t)
                res[t]["hot"].append(graph_time(hot, n=nbuf))
                res[t]["cold"].append(graph_time(cold, n=nbuf))
            except Exception as e:
                res[t]["err"] = str(e)[:60]
    return res


for rows, n, k, kind, inc, ch in SHAPES:
    tiles = (inc,) + tuple(ch)
    res = bench(rows, n, k, kind, tiles)
    line = "%-18s %-5s" % ("%dx%dx%d" % (rows, n, k), kind)
    for t in tiles:
        if res[t]["cold"]:
            line += " | t%d cold %5.1f hot %5.1f" % (t, statistics.median(res[t]["cold"]), statistics.median(res[t]["hot"]))
        else:
            line += " | t%d n/a %s" % (t, res[t].get("err", ""))
    print(line, flush=True)
