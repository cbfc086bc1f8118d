"""A/B of direct-conv tile decompositions in ONE process (interleaved rounds, graph chains, weights hot / cold):
    python tools/conv3s_ab.py [rounds]"""
import os, sys, statistics, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dualdiff_amd import ops as O
from tools._timing import graph_time
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
dt, dev = torch.float16, torch.device("cuda")
O.workspace(512 << 20, dev)
# (instances, h, w, cin, cout, incumbent, challengers)
SHAPES = [(12, 28, 50, 320, 320, 39, ()), (12, 28, 50, 640, 320, 39, ()), (12, 28, 50, 960, 320, 39, ()),
          (12, 14, 25, 640, 640, 31, ()), (12, 14, 25, 1280, 640, 31, ()), (12, 14, 25, 1920, 640, 31, ()),
          (12, 7, 13, 1280, 1280, 31, ()), (12, 7, 13, 2560, 1280, 31, ()),
          (48, 28, 50, 320, 320, 39, ()), (48, 14, 25, 640, 640, 31, ()), (48, 7, 13, 1280, 1280, 31, ())]
for m, h, w, cin, cout, inc, ch in SHAPES:
    rows = m * h * w
    x = torch.randn(rows, cin, device=dev).to(dt)
    bi = torch.randn(cout, device=dev).to(dt)
    tv = torch.randn(m, cout, device=dev).to(dt)
    nbuf = max(3, int(600e6 // (cout * 9 * cin * 2)) + 1)
    ws_ = [(torch.randn(cout, 9 * cin, device=dev) * (9 * cin) ** -0.5).to(dt) for _ in range(nbuf)]
    st = {"i": 0}
    res = {t: {"hot": [], "cold": []} for t in (inc,) + ch}
    for _ in range(rounds):
        for t in res:
            try:
                def hot():
                    return O.conv3x3(x, ws_[0], bi, m, h, w, rowvec=tv, tile=t, split_k=1)

                def cold():
                    st["i"] += 1
                    return O.conv3x3(x, ws_[st["i"] % nbuf], bi, m, h, w, rowvec=tv, tile=t, split_k=1)
                res[t]["hot"].append(graph_time(hot, n=nbuf))
                res[t]["cold"].append(graph_time(cold, n=nbuf))
            except Exception as e:
                res[t]["err"] = str(e)[:50]
    line = "%-26s" % ("%dx%dx%d %d->%d" % (m, h, w, cin, cout))
    for t in res:
        line += (" | t%d cold %6.1f hot %6.1f" % (t, statistics.median(res[t]["cold"]), statistics.median(res[t]["hot"]))
                 if res[t]["cold"] else " | t%d n/a %s" % (t, res[t].get("err", "")))
    print((os.environ.get("AB_LABEL", "") + " " + line).strip(), flush=True)
