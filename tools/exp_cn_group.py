"""What would GROUPED ControlNet-branch launches buy?    A grouped launch of the two branches has the
row counts of ONE branch on 24 view-instances, so: T(one branch, 12 instances), T(two branches on two streams, 12
each — what the step does today), T(one branch, 24 instances = the grouped stand-in), each as its own HIP graph."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from dualdiff_amd import ops as O
dtype = torch.float16; dev = torch.device("cuda:0")
unet, cns = bench.build_models(dtype, dev)
del unet
H, W = bench.H, bench.W
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
    res = {}
    for scenes in (1, 2):
        M = 12 * scenes
        lat, prompt, cam, boxes, conds = bench.synthetic_inputs(scenes, dtype, dev, 1)
        lat2 = torch.cat([lat.reshape(6 * scenes, 4, H, W)] * 2)
        t = torch.full((M,), 500.0, device=dev)
        x8 = O.nchw_to_nhwc(lat2, 8)
        p = [cns[i].prepare_condition(cam, boxes[i], prompt, conds[i], False) for i in range(2)]
        ms, _ = graph_time(lambda: cns[1].forward_nhwc(x8, M, H, W, t, p[1], 1.0))
        res[("one", M)] = ms
        print("one branch, %2d instances: %.3f ms" % (M, ms), flush=True)
        side = torch.cuda.Stream()
        def both():
            main = torch.cuda.current_stream()
            side.wait_stream(main)
            with torch.cuda.stream(side):
                a = cns[0].forward_nhwc(x8, M, H, W, t, p[0], 1.0)
            b = cns[1].forward_nhwc(x8, M, H, W, t, p[1], 1.0)
            main.wait_stream(side)
            return a, b
        ms, _ = graph_time(both)
        res[("two", M)] = ms
        print("two branches on two streams, %2d instances each: %.3f ms" % (M, ms), flush=True)
        def serial():
            a = cns[0].forward_nhwc(x8, M, H, W, t, p[0], 1.0)
            b = cns[1].forward_nhwc(x8, M, H, W, t, p[1], 1.0)
            return a, b
        ms, _ = graph_time(serial)
        print("two branches back to back, %2d instances each: %.3f ms" % (M, ms), flush=True)
    print("grouped stand-in (one branch at 24) vs two streams at 12: %.3f vs %.3f ms" % (res[("one", 24)], res[("two", 12)]))
