import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from dualdiff_amd import ops as O
from dualdiff_amd.pipeline.pipeline_bev_controlnet import BEVDenoiser
dt = torch.float16; dev = torch.device("cuda")
unet, cns = bench.build_models(dt, dev)
inp = bench.synthetic_inputs(1, dt, dev, 1)
def run(graph):
    box = {}
    halves = []
    for hf in (0, 1):
        d = BEVDenoiser(unet, cns, guidance_scale=2.0, num_inference_steps=50, use_graph=graph, cfg_half=hf,
                        cfg_exchange=lambda e: None)
        d._combine_halves = lambda: None
        d.set_inputs(*inp)
        halves.append(d)
    eps_log = []
    for i in range(2):
        for d in halves:
            d.step(i)
        eps2 = torch.stack([halves[0]._eps_half, halves[1]._eps_half])
        eps_log.append(eps2.float().clone())
        for d in halves:
            O.cfg_ddim_step(eps2, d.lat2[0], d.coef, d.guidance_scale, x_out=d.lat2[0], x_dup=d.lat2[1])
    torch.cuda.synchronize()
    return halves[0].latents.float().clone(), eps_log
with torch.no_grad():
    for graph in (False, True):
        ref, elog = run(graph)
        for rep in range(4):
            y, el = run(graph)
            print("graph=%s rep %d: latents max|diff| %.3e, eps step0 %.3e step1 %.3e" % (
                graph, rep, (y - ref).abs().max().item(), (el[0] - elog[0]).abs().max().item(), (el[1] - elog[1]).abs().max().item()), flush=True)
