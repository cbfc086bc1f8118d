"""How much would running the two ControlNet forward graphs of the drop-in loop CONCURRENTLY save?  Timing only: the two
graphs replay on two side streams (their split-K workspaces may collide — results are not checked), against the same two
replays back to back on one stream, and against the UNet graph alone.  python tools/sibling_overlap_probe.py"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B
from dualdiff_amd.pipeline.pipeline_bev_controlnet import ddim_schedule
dev = torch.device("cuda:0")
dtype = torch.float16
unet, cns = B.build_models(dtype, dev)
inputs = B.synthetic_inputs(1, dtype, dev, seed=1234)
ts, coefs = ddim_schedule(50)
ts = ts.to(dev); coefs = coefs.tolist()
with torch.no_grad():
    B.dropin_loop(unet, cns, inputs, ts, coefs, 4)
torch.cuda.synchronize()
graphs = []
for m in cns + [unet]:
    fg = m.__dict__["_fwd_graphs"]
    e = list(fg.entries.values())[-1]
    graphs.append(e["graph"])
print("graphs:", len(graphs))


def timed(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
cur = torch.cuda.current_stream()


def serial():
    graphs[0].replay(); graphs[1].replay()


def concurrent():
    s1.wait_stream(cur); s2.wait_stream(cur)
    with torch.cuda.stream(s1):
        graphs[0].replay()
    with torch.cuda.stream(s2):
        graphs[1].replay()
    cur.wait_stream(s1); cur.wait_stream(s2)


for rep in range(2):
    a = timed(lambda: graphs[0].replay()); b = timed(lambda: graphs[1].replay()); u = timed(lambda: graphs[2].replay())
    se = timed(serial); co = timed(concurrent)
    print("cn0 %.3f ms  cn1 %.3f ms  unet %.3f ms | two ControlNets back to back %.3f ms, concurrent %.3f ms (saves %.3f ms of a %.2f ms drop-in step)"
          % (a, b, u, se, co, se - co, a + b + u))
