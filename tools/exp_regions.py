"""Region timing (each region as its own HIP graph, alone on the GPU): UNet encode, UNet decode,
ControlNet branch 0 / 1 (tokens + condition + forward)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from dualdiff_amd import ops as O
dtype = torch.bfloat16; dev = torch.device("cuda:0")
if os.environ.get("DD_TUNE_CACHE") and os.path.exists(os.environ["DD_TUNE_CACHE"]):
    O.load_tuned(os.environ["DD_TUNE_CACHE"])
unet, cns = bench.build_models(dtype, dev)
lat, prompt, cam, boxes, conds = bench.synthetic_inputs(1, dtype, dev, 1)
H, W, M = bench.H, bench.W, 12
lat2 = torch.cat([lat.reshape(6, 4, H, W)] * 2)
t = torch.full((M,), 500.0, device=dev)
def graph_time(fn, n=20):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn(); torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s): out = fn()
    torch.cuda.current_stream().wait_stream(s)
    for _ in range(3): g.replay()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, out
with torch.no_grad():
    x8 = O.nchw_to_nhwc(lat2, 8)
    tok0 = cns[0].prepare_tokens(cam, boxes[0], prompt, False)
    ms, _ = graph_time(lambda: cns[0].prepare_tokens(cam, boxes[0], prompt, False)); print("CNet0 tokens        %.3f ms" % ms)
    ms, _ = graph_time(lambda: cns[0].prepare_cond(tok0, conds[0])); print("CNet0 cond embed+SFA %.3f ms" % ms)
    p0 = cns[0].prepare_cond(tok0, conds[0])
    ms, r0 = graph_time(lambda: cns[0].forward_nhwc(x8, M, H, W, t, p0, 1.0)); print("CNet0 forward       %.3f ms" % ms)
    ms, _ = graph_time(lambda: cns[1].prepare_condition(cam, boxes[1], prompt, conds[1], False)); print("CNet1 prepare       %.3f ms" % ms)
    p1 = cns[1].prepare_condition(cam, boxes[1], prompt, conds[1], False)
    ms, r1 = graph_time(lambda: cns[1].forward_nhwc(x8, M, H, W, t, p1, 1.0)); print("CNet1 forward       %.3f ms" % ms)
    ms, st = graph_time(lambda: unet.encode_nhwc(x8, M, H, W, t, tok0["ctx2d"], tok0["lc"])); print("UNet encode         %.3f ms" % ms)
    down = [tuple((r0[j][0], r1[j][0])) for j in range(len(r0) - 1)]; mid = (r0[-1][0], r1[-1][0])
    ms, _ = graph_time(lambda: unet.decode_nhwc(st, down, mid)); print("UNet decode         %.3f ms" % ms)
