"""Which Python lines of the step still call torch ops that launch kernels (copies, cats, elementwise)?
Uses a TorchDispatchMode to log every aten op executed during one eager step with its call site."""
import os, sys, torch, collections, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from dualdiff_amd.pipeline.pipeline_bev_controlnet import BEVDenoiser
from torch.utils._python_dispatch import TorchDispatchMode
dtype = torch.bfloat16; dev = torch.device("cuda:0")
unet, cns = bench.build_models(dtype, dev)
SKIP = ("aten.view", "aten.reshape", "aten._unsafe_view", "aten.expand", "aten.permute", "aten.slice", "aten.select",
        "aten.unsqueeze", "aten.squeeze", "aten.t.", "aten.transpose", "aten.detach", "aten.empty", "aten.alias",
        "aten.as_strided", "aten.split", "aten.unbind", "aten.is_", "aten.sym_", "aten._local_scalar")
class Log(TorchDispatchMode):
    def __init__(self):
        super().__init__(); self.c = collections.Counter()
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not name.startswith(SKIP):
            site = "?"
            for fr in reversed(traceback.extract_stack()):
                if "dualdiff_amd" in fr.filename and "ops.py" not in fr.filename:
                    site = "%s:%d" % (fr.filename.split("dualdiff_amd/")[-1], fr.lineno); break
            self.c[(name, site)] += 1
        return func(*args, **(kwargs or {}))
with torch.no_grad():
    den = BEVDenoiser(unet, cns, use_graph=False, parallel_branches=False)
    den.set_inputs(*bench.synthetic_inputs(1, dtype, dev, 1))
    den.step(0); den.step(1)
    log = Log()
    with log:
        den._step_body()
for (name, site), n in sorted(log.c.items(), key=lambda kv: -kv[1])[:45]:
    print("%3d x %-32s %s" % (n, name, site))
