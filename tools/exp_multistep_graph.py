"""Experiment: k denoising steps captured into ONE HIP graph vs k single-step replays."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from dualdiff_amd import ops as O
from dualdiff_amd.pipeline.pipeline_bev_controlnet import BEVDenoiser
dtype = torch.bfloat16; dev = torch.device("cuda:0")
if os.environ.get("DD_TUNE_CACHE") and os.path.exists(os.environ["DD_TUNE_CACHE"]):
    O.load_tuned(os.environ["DD_TUNE_CACHE"])
unet, cns = bench.build_models(dtype, dev)
with torch.no_grad():
    den = BEVDenoiser(unet, cns, use_graph=True)
    den.set_inputs(*bench.synthetic_inputs(1, dtype, dev, 1))
    den.capture()
    def t_single(n=50):
        for i in range(5): den.step(i)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(n): den.step(i % 50)
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
    print("single-step graph: %.3f ms/step" % t_single())
    for k in (2, 5, 10):
        g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            den._step_body(); torch.cuda.synchronize()      # sizes the per-stream workspaces
            with torch.cuda.graph(g, stream=s):
                for _ in range(k): den._step_body()
        torch.cuda.current_stream().wait_stream(s)
        for _ in range(2): g.replay()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        reps = 50 // k
        for _ in range(reps): g.replay()
        torch.cuda.synchronize()
        print("%d-step graph: %.3f ms/step" % (k, (time.perf_counter() - t0) / (reps * k) * 1e3))
