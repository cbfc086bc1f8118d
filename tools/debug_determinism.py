"""Debug: run-to-run / eager-vs-graph / hoisted-vs-not determinism of one denoising step."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from dualdiff_amd.pipeline.pipeline_bev_controlnet import BEVDenoiser

dtype = torch.float16
dev = torch.device("cuda:0")
unet, cns = bench.build_models(dtype, dev)
inputs = bench.synthetic_inputs(1, dtype, dev, 1)
res = {}
with torch.no_grad():
    for name, graph, hoist in [("eager", False, False), ("eager2", False, False), ("eager_hoist", False, True),
                               ("graph", True, False), ("graph_hoist", True, True)]:
        den = BEVDenoiser(unet, cns, use_graph=graph, hoist_invariant=hoist)
        den.set_inputs(*inputs)
        den.run(2)
        res[name] = den.latents.float().clone()
        print(name, "finite", torch.isfinite(res[name]).all().item(), "norm", res[name].norm().item())
base = res["eager"]
for k, v in res.items():
    print("%-12s max|diff vs eager| = %.3e  equal=%s" % (k, (v - base).abs().max().item(), torch.equal(v, base)))
