"""Step time per 20-step window over a long run of the captured step (clock ramp / steady state).  GPU box."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from dualdiff_amd.pipeline.pipeline_bev_controlnet import BEVDenoiser

dt = torch.float16 if (len(sys.argv) < 2 or sys.argv[1] == "fp16") else torch.bfloat16
scenes = int(sys.argv[2]) if len(sys.argv) > 2 else 1
idle = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
tiny = int(sys.argv[4]) if len(sys.argv) > 4 else 0          # tiny kernel launches before the replays
reset = int(sys.argv[5]) if len(sys.argv) > 5 else 0         # restore the initial latents every `reset` steps (0: never)
dev = torch.device("cuda:0")
unet, cns = bench.build_models(dt, dev)
den = BEVDenoiser(unet, cns, guidance_scale=2.0, num_inference_steps=50)
with torch.no_grad():
    den.set_inputs(*bench.synthetic_inputs(scenes, dt, dev, seed=1234))
    den.capture()
    torch.cuda.synchronize()
    time.sleep(idle)
    z = torch.zeros(64, device=dev)
    for _ in range(tiny):
        z.add_(1.0)
    torch.cuda.synchronize()
    lat0 = den.lat2.clone()
    t_prev = time.perf_counter()
    out, fin = [], []
    for w in range(40):
        for i in range(20):
            k = w * 20 + i
            if reset and k % reset == 0:
                den.lat2.copy_(lat0)
            den.step(k % 50)
        torch.cuda.synchronize()
        t = time.perf_counter()
        out.append((t - t_prev) / 20 * 1e3)
        fin.append(int(torch.isfinite(den.lat2.float()).all().item()))
        t_prev = time.perf_counter()
print("finite per window:", "".join(str(f) for f in fin))
print(sys.argv[1:] , "ms/step per 20-step window:", " ".join("%.2f" % x for x in out))
