"""BASELINE configs[2] as a whole step (VERDICT r5 missing 3): 4 scenes per GPU = 48 view-instances in ONE fused step
(HIP-graph replay: 2 ControlNet branches + multiview UNet + CFG + DDIM) against FOUR one-scene steps on the same seeds.

Scenes are independent on this path (the only cross-instance operator, attn4, couples the 6 views of ONE scene:
blocks.py:190-222), so the batched step must reproduce the per-scene steps up to the storage rounding of the few places
where the row count changes the kernel (other GEMM tiles / split-K at 48 instances: other summation order).  Bound: the
drop-in loop's (tests/test_dropin_loop_gpu.py) — two DDIM steps, rel-L2 of the latents."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_four_scene_step_equals_four_one_scene_steps(gpu, dtype):
    import bench
    from dualdiff_amd.pipeline.pipeline_bev_controlnet import BEVDenoiser
    from tests.parity_util import log_row
    dev = torch.device("cuda:0")
    b, n, steps = 4, bench.NCAM, 2
    unet, cns = bench.build_models(dtype, dev)
    lat, prompt, cam, boxes, conds = bench.synthetic_inputs(b, dtype, dev, seed=21)
    lat = lat + 0.1 * torch.arange(b, device=dev, dtype=dtype).reshape(b, 1, 1, 1, 1)     # scenes differ in their noise too
    with torch.no_grad():
        den = BEVDenoiser(unet, cns, guidance_scale=2.0, num_inference_steps=50, use_graph=True)
        den.set_inputs(lat, prompt, cam, boxes, conds)
        den.run(steps)
        torch.cuda.synchronize()
        batched = den.latents.clone()
        assert batched.shape == (b, n, 4, bench.H, bench.W) and torch.isfinite(batched.float()).all()
        del den
        worst = 0.0
        for s in range(b):
            rows = [s, b + s]                                        # [uncond ; cond] rows of scene s
            inst = torch.cat([torch.arange(r * n, (r + 1) * n) for r in rows]).to(dev)
            one = BEVDenoiser(unet, cns, guidance_scale=2.0, num_inference_steps=50, use_graph=True)
            one.set_inputs(lat[s:s + 1], prompt[rows], cam[rows], [{k: v[rows] for k, v in d.items()} for d in boxes],
                           [conds[0][rows], conds[1][inst]])
            one.run(steps)
            torch.cuda.synchronize()
            ref = one.latents[0].float()
            e = ((batched[s].float() - ref).norm() / ref.norm()).item()
            worst = max(worst, e)
            del one
        other = ((batched[0].float() - batched[1].float()).norm() / batched[0].float().norm()).item()
    bnd = 2e-3 if dtype == torch.float16 else 1.6e-2
    print("4-scene fused step vs four one-scene steps after %d steps, %s: worst rel-L2 %.3e (bound %.1e; scene 0 vs scene 1: %.2e)"
          % (steps, dtype, worst, bnd, other))
    log_row("configs[2]: 4-scene step vs 4 x one-scene steps (2 steps)", dtype, worst, float("nan"), bnd)
    assert worst <= bnd and other > 10 * bnd          # ... and the scenes are not trivially equal
