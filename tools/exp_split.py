"""Experiment: one M=12 chain vs two concurrent M=6 chains (the two CFG halves as separate graphs)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from dualdiff_amd import ops as O
from dualdiff_amd.pipeline.pipeline_bev_controlnet import BEVDenoiser

dtype = torch.bfloat16
dev = torch.device("cuda:0")
unet, cns = bench.build_models(dtype, dev)
H, W = bench.H, bench.W
lat, prompt, cam, boxes, conds = bench.synthetic_inputs(1, dtype, dev, 1)

def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

class Half:
    def __init__(self, half):
        sl = slice(half, half + 1)
        self.m = 6
        self.lat = lat.reshape(6, 4, H, W).clone()
        self.prompt = prompt[sl]; self.cam = cam[sl]
        self.boxes = [{k: v[sl] for k, v in b.items()} for b in boxes]
        self.conds = [conds[0][sl], conds[1][half * 6: half * 6 + 6]]
        self.t = torch.full((6,), 500.0, device=dev)
        self.side = [torch.cuda.Stream() for _ in cns]
        self.graph = None
    def body(self):
        m = self.m
        x8 = O.nchw_to_nhwc(self.lat, 8)
        main = torch.cuda.current_stream()
        prep0 = cns[0].prepare_condition(self.cam, self.boxes[0], self.prompt, self.conds[0], False)
        results = [None] * len(cns)
        for i, cn in enumerate(cns):
            s = self.side[i]; s.wait_stream(main)
            with torch.cuda.stream(s):
                p = prep0 if i == 0 else cn.prepare_condition(self.cam, self.boxes[i], self.prompt, self.conds[i], False)
                results[i] = cn.forward_nhwc(x8, m, H, W, self.t, p, 1.0)
        state = unet.encode_nhwc(x8, m, H, W, self.t, prep0["ctx2d"], prep0["lc"])
        for s in self.side: main.wait_stream(s)
        down = [tuple(results[i][j][0] for i in range(len(cns))) for j in range(len(results[0]) - 1)]
        mid = tuple(results[i][-1][0] for i in range(len(cns)))
        return unet.decode_nhwc(state, down, mid)
    def capture(self):
        self.body(); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            self.body(); torch.cuda.synchronize()
            with torch.cuda.graph(g, stream=s):
                self.body()
        torch.cuda.current_stream().wait_stream(s)
        self.graph = g

with torch.no_grad():
    d12 = BEVDenoiser(unet, cns, use_graph=True)
    d12.set_inputs(lat, prompt, cam, boxes, conds)
    d12.capture()
    print("M=12 single chain (3 streams): %.2f ms/step" % timeit(lambda: d12.step(3)))
    a, b = Half(0), Half(1)
    a.capture(); b.capture()
    print("M=6 single chain: %.2f ms" % timeit(lambda: a.graph.replay()))
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    def both():
        cur = torch.cuda.current_stream()
        s1.wait_stream(cur); s2.wait_stream(cur)
        with torch.cuda.stream(s1): a.graph.replay()
        with torch.cuda.stream(s2): b.graph.replay()
        cur.wait_stream(s1); cur.wait_stream(s2)
    print("2 x M=6 concurrent chains (6 streams): %.2f ms/pair" % timeit(both))
