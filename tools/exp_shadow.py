"""Does work forked beside the UNet decoder inside a captured HIP graph run concurrently with it?  decode alone, the
two branches' conditioning (tokens + ORS embedder + SFA) alone, and decode with the conditioning forked on a side
stream — each as its own HIP graph.  (Round 3: cross-step pipelining of the invariant conditioning measured +-0.)"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from dualdiff_amd import ops as O
dtype = torch.float16; dev = torch.device("cuda:0")
unet, cns = bench.build_models(dtype, dev)
H, W, M = bench.H, bench.W, 12
lat, prompt, cam, boxes, conds = bench.synthetic_inputs(1, dtype, dev, 1)
lat2 = torch.cat([lat.reshape(6, 4, H, W)] * 2)
t = torch.full((M,), 500.0, device=dev)
def graph_time(fn, n=30):
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
    p = [cns[i].prepare_condition(cam, boxes[i], prompt, conds[i], False) for i in range(2)]
    r = [cns[i].forward_nhwc(x8, M, H, W, t, p[i], 1.0) for i in range(2)]
    st = unet.encode_nhwc(x8, M, H, W, t, p[0]["ctx2d"], p[0]["lc"])
    down = [tuple((r[0][j][0], r[1][j][0])) for j in range(len(r[0]) - 1)]; mid = (r[0][-1][0], r[1][-1][0])
    prep = lambda: [cns[i].prepare_condition(cam, boxes[i], prompt, conds[i], False) for i in range(2)]
    dec = lambda: unet.decode_nhwc(st, down, mid)
    ms_d, _ = graph_time(dec); print("decode alone            %.3f ms" % ms_d, flush=True)
    ms_p, _ = graph_time(prep); print("conditioning alone      %.3f ms" % ms_p, flush=True)
    side = torch.cuda.Stream()
    def both():
        main = torch.cuda.current_stream()
        side.wait_stream(main)
        with torch.cuda.stream(side):
            q = prep()
        e = dec()
        main.wait_stream(side)
        return e, q
    ms_b, _ = graph_time(both); print("decode || conditioning  %.3f ms (sum %.3f)" % (ms_b, ms_d + ms_p), flush=True)
    def enc():
        return unet.encode_nhwc(x8, M, H, W, t, p[0]["ctx2d"], p[0]["lc"])
    ms_e, _ = graph_time(enc); print("encode alone            %.3f ms" % ms_e, flush=True)
    sides = [torch.cuda.Stream() for _ in range(2)]
    def phase1(with_prep):
        main = torch.cuda.current_stream()
        outs = [None, None]
        for i in range(2):
            sides[i].wait_stream(main)
            with torch.cuda.stream(sides[i]):
                pi = cns[i].prepare_condition(cam, boxes[i], prompt, conds[i], False) if with_prep else p[i]
                outs[i] = cns[i].forward_nhwc(x8, M, H, W, t, pi, 1.0)
        e = enc()
        for s_ in sides:
            main.wait_stream(s_)
        return e, outs
    ms1, _ = graph_time(lambda: phase1(True)); print("phase 1 (3 streams) with conditioning     %.3f ms" % ms1, flush=True)
    ms2, _ = graph_time(lambda: phase1(False)); print("phase 1 (3 streams) without conditioning  %.3f ms" % ms2, flush=True)
