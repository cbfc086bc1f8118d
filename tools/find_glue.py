"""Which Python lines of the step still launch torch kernels (copies, cats, elementwise)?"""
import os, sys, torch, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from dualdiff_amd.pipeline.pipeline_bev_controlnet import BEVDenoiser
from torch.profiler import profile, ProfilerActivity
dtype = torch.bfloat16; dev = torch.device("cuda:0")
unet, cns = bench.build_models(dtype, dev)
with torch.no_grad():
    den = BEVDenoiser(unet, cns, use_graph=False, parallel_branches=False)
    den.set_inputs(*bench.synthetic_inputs(1, dtype, dev, 1))
    den.step(0); den.step(1)
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        den._step_body()
        torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
for ev in prof.key_averages(group_by_stack_n=8):
    if not ev.key.startswith("aten::") or ev.device_time_total <= 0:
        continue
    st = [s for s in (ev.stack or []) if "dualdiff_amd" in s]
    key = (ev.key, st[0].strip()[-100:] if st else "?")
    agg[key][0] += ev.count; agg[key][1] += ev.self_device_time_total
for (name, where), (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:30]:
    if t > 0:
        print("%4d x %-24s %8.1f us  %s" % (n, name, t, where))
