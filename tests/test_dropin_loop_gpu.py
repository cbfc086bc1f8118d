"""The reference sampler's loop body, written against the PUBLIC drop-in surfaces only, against the fused sampler.

`pipeline/pipeline_bev_controlnet.py:381-504` is: CFG-double the latents; call every ControlNet's `forward(sample[b,n,4,h,w],
timestep, camera_param, bboxes_3d_data, encoder_hidden_states, controlnet_cond, ..., return_dict=False)` and sum the 13
residuals of the branches (:405-431); call `unet(sample[(b n),4,h,w], t, encoder_hidden_states=<tokens of branch 0>,
down_block_additional_residuals=..., mid_block_additional_residual=...).sample` (:476-484); guidance (:487-492);
scheduler step (:497-499).  A maintainer who only swaps the classes by config override runs exactly this.  The test runs
it with torch tensor ops between the calls (NCHW in, NCHW out, nothing fused across the calls) and compares the latents
after two DDIM steps with `BEVDenoiser` (one HIP-graph replay per step, zero-conv sums and residual adds in epilogues,
CFG + DDIM in one kernel).  Same kernels underneath, different fusion: the two paths round at different places, so the
bound is the storage rounding of a few adds, not bit equality."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_reference_shaped_loop_equals_fused_sampler(gpu, dtype):
    import bench
    from dualdiff_amd.pipeline.pipeline_bev_controlnet import BEVDenoiser, ddim_schedule
    dev = torch.device("cuda:0")
    unet, cns = bench.build_models(dtype, dev)
    lat, prompt, cam, boxes, conds = bench.synthetic_inputs(1, dtype, dev, seed=7)
    g_scale, steps = 2.0, 2
    ts, coefs = ddim_schedule(50)
    with torch.no_grad():
        # ---- the reference-shaped loop through forward() only (bench.dropin_loop: the `dropin` leg times the same code) ----
        loop_latents = bench.dropin_loop(unet, cns, (lat, prompt, cam, boxes, conds), ts.to(dev), coefs.tolist(), steps,
                                         g_scale=g_scale)
        # ---- the fused sampler -------------------------------------------------------------------------------------
        den = BEVDenoiser(unet, cns, guidance_scale=g_scale, num_inference_steps=50, use_graph=True)
        den.set_inputs(lat, prompt, cam, boxes, conds)
        den.run(steps)
        torch.cuda.synchronize()
        fused = den.latents
    assert fused.shape == loop_latents.shape and torch.isfinite(fused.float()).all()
    e = ((fused.float() - loop_latents.float()).norm() / loop_latents.float().norm()).item()
    bnd = 2e-3 if dtype == torch.float16 else 1.6e-2
    print("reference-shaped loop vs fused sampler after %d steps, %s: rel-L2 %.3e (bound %.1e)" % (steps, dtype, e, bnd))
    from tests.parity_util import log_row
    log_row("drop-in loop vs fused sampler (2 steps)", dtype, e, float("nan"), bnd)
    assert e <= bnd, e
