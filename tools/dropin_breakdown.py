"""Where the `dropin` leg's step goes (GPU box): each model's forward graph alone, the torch glue alone, the whole loop."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from dualdiff_amd.pipeline.pipeline_bev_controlnet import ddim_schedule
dev = torch.device("cuda:0"); dt = torch.float16
unet, cns = bench.build_models(dt, dev)
inputs = bench.synthetic_inputs(1, dt, dev, seed=1234)
lat, prompt, cam, boxes, conds = inputs
ts, coefs = ddim_schedule(50); ts = ts.to(dev); coefs = coefs.tolist()
N = 30


def timed(fn, n=N):
    fn(); fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3


with torch.no_grad():
    t = ts[0]; lmi = torch.cat([lat] * 2)
    outs = []
    for j, cn in enumerate(cns):
        f = lambda j=j, cn=cn: cn(lmi, t.expand(2), cam, boxes[j], prompt, conds[j], conditioning_scale=1.0, guess_mode=False, return_dict=False, use_aug_text=False)
        print("controlnet %d forward graph  %.3f ms" % (j, timed(f)))
        outs.append(f())
    d0, m0, c0 = outs[0]; d1, m1, _ = outs[1]
    def glue():
        ds = [a.clone() for a in d0]; ms = m0.clone()
        ds = [a + b for a, b in zip(ds, d1)]; ms = ms + m1
        return ds, ms
    print("residual glue (13 clones + 13 adds) %.3f ms" % timed(glue))
    ds, ms = glue()
    x = lmi.reshape(12, *lmi.shape[2:])
    fu = lambda: unet(x, t, encoder_hidden_states=c0, down_block_additional_residuals=ds, mid_block_additional_residual=ms).sample
    print("unet forward graph (incl. 14 input copies) %.3f ms" % timed(fu))
    ds2 = [torch.empty_like(a) for a in ds]
    print("14 input copies alone (foreach) %.3f ms" % timed(lambda: torch._foreach_copy_(ds2, ds)))
    eps = fu()
    def cfg():
        e = eps.reshape(2, 1, 6, *eps.shape[1:]).float(); e = e[0] + 2.0 * (e[1] - e[0]); c = coefs[0]
        xx = lat.float(); x0 = (xx - c[1] * e) / c[0]; return (c[2] * x0 + c[3] * e).to(dt)
    print("cfg + ddim glue %.3f ms" % timed(cfg))
    print("whole loop %.3f ms/step" % (timed(lambda: bench.dropin_loop(unet, cns, inputs, ts, coefs, 10), n=3) / 10))
