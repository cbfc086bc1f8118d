"""Per-shape GEMM / conv time table of one eager denoising step (HIP events per launch)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from dualdiff_amd import ops as O
from dualdiff_amd.pipeline.pipeline_bev_controlnet import BEVDenoiser

if os.environ.get("DD_TUNE_CACHE") and os.path.exists(os.environ["DD_TUNE_CACHE"]):
    O.load_tuned(os.environ["DD_TUNE_CACHE"])
dtype = torch.float16           # the headline dtype of bench.py
dev = torch.device("cuda:0")
unet, cns = bench.build_models(dtype, dev)
den = BEVDenoiser(unet, cns, use_graph=False, parallel_branches=False)
with torch.no_grad():
    den.set_inputs(*bench.synthetic_inputs(1, dtype, dev, 1))
    den.step(0); den.step(1)
    t = O.KernelTimer(shapes=True)
    O.set_timer(t)
    den._step_body()
    O.set_timer(None)
summ = t.summary()
tot = sum(v["ms"] for v in summ.values())
print("total timed %.3f ms" % tot)
for k, v in sorted(summ.items(), key=lambda kv: -kv[1]["ms"]):
    print("%-95s n=%3d total=%7.3f ms avg=%7.1f us %6.1f TF/s" % (k[-95:], v["count"], v["ms"], v["ms"] / v["count"] * 1e3,
                                                              v["flops"] / (v["ms"] * 1e-3) / 1e12))
