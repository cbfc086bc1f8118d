"""Model-level A/B of the fused cross-attention kernel: UNet / ControlNet outputs with layers.XATTN_FUSED on vs off."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from dualdiff_amd.networks import layers
from dualdiff_amd import ops as O
dt = torch.float16; dev = torch.device("cuda")
unet, cns = bench.build_models(dt, dev)
M = int(sys.argv[1]) if len(sys.argv) > 1 else 12
LC = int(sys.argv[2]) if len(sys.argv) > 2 else 15
g = torch.Generator(device=dev).manual_seed(1)
x = torch.randn((M, 4, 28, 50), generator=g, device=dev).to(dt)
ctx = torch.randn((M, LC, 768), generator=g, device=dev).to(dt)
outs = {}
with torch.no_grad():
    for fused in (False, True, False, True):
        layers.XATTN_FUSED = fused
        y = unet(x, 481, encoder_hidden_states=ctx).sample.float()
        torch.cuda.synchronize()
        outs.setdefault(fused, []).append(y)
    a, b = outs[False][0], outs[True][0]
    print("UNet M=%d lc=%d: fused vs 3 launches rel-L2 %.3e; repeat off %.1e on %.1e" % (
        M, LC, ((a - b).norm() / a.norm()).item(), (outs[False][0] - outs[False][1]).abs().max().item(),
        (outs[True][0] - outs[True][1]).abs().max().item()))
    # one 28x50 transformer block alone
    blk = unet.down_blocks[0].attentions[0].transformer_blocks[0]
    h = torch.randn((M * 1400, 320), generator=g, device=dev).to(dt)
    c2 = ctx.reshape(-1, 768)
    res = {}
    for fused in (False, True):
        layers.XATTN_FUSED = fused
        res[fused] = blk.run(h, M, 1400, c2, LC).float()
    print("block: rel-L2 %.3e" % ((res[0] - res[1]).norm() / res[0].norm()).item())
    for fused in (False, True):
        layers.XATTN_FUSED = fused
        hh = blk._attn(blk.attn2, blk.norm2, h, M, 1400, c2, LC, next_norm=blk.norm4)
        res[fused] = hh.float()
        res[(fused, "ln")] = blk.norm4.run(hh).float()
    print("attn2 only: rel-L2 %.3e, next LN %.3e" % (((res[0] - res[1]).norm() / res[0].norm()).item(),
          ((res[(0, "ln")] - res[(1, "ln")]).norm() / res[(0, "ln")].norm()).item()))
    # ControlNet branches (SFA inside prepare_cond + attn2 of the 28x50 blocks)
    from dualdiff_amd.networks.txt_con_fusion import txt_con_XFormersAttn
    for ltxt in (9, 77):
        lat, prompt, cam, boxes, conds = bench.synthetic_inputs(1, dt, dev, 1)
        prompt = prompt[:, :ltxt].contiguous()
        for i, cn in enumerate(cns):
            r = {}
            for fused in (False, True):
                layers.XATTN_FUSED = fused
                p_ = cn.prepare_condition(cam, boxes[i], prompt, conds[i], False)
                r[fused] = p_["cond"].float()
            print("ControlNet %d ltxt=%d: SFA'd condition fused vs 3 launches rel-L2 %.3e" % (
                i, ltxt, ((r[0] - r[1]).norm() / r[0].norm()).item()))
    lat, prompt, cam, boxes, conds = bench.synthetic_inputs(1, dt, dev, 1)
    prompt = prompt[:, :9].contiguous()
    lat2 = torch.cat([lat.reshape(6, 4, 28, 50)] * 2)
    tt = torch.full((12,), 500.0, device=dev)
    x8 = O.nchw_to_nhwc(lat2, 8)
    for i, cn in enumerate(cns):
        layers.XATTN_FUSED = False
        p_ = cn.prepare_condition(cam, boxes[i], prompt, conds[i], False)
        r = {}
        for fused in (False, True):
            layers.XATTN_FUSED = fused
            r[fused] = [o[0].float() for o in cn.forward_nhwc(x8, 12, 28, 50, tt, p_, 1.0)]
        print("ControlNet %d forward (same conditioning): residuals fused vs 3 launches rel-L2 %s" % (
            i, " ".join("%.1e" % ((a - b).norm() / a.norm()).item() for a, b in zip(r[0], r[1]))))
    from dualdiff_amd.pipeline.pipeline_bev_controlnet import BEVDenoiser
    outs = {}
    for fused in (False, True):
        layers.XATTN_FUSED = fused
        den = BEVDenoiser(unet, cns, guidance_scale=2.0, num_inference_steps=50, use_graph=False)
        den.set_inputs(lat, prompt, cam, boxes, conds)
        den.run(2)
        outs[fused] = den.latents.float().clone()
    print("denoiser, 2 steps: latents fused vs 3 launches rel-L2 %.3e" % ((outs[0] - outs[1]).norm() / outs[0].norm()).item())
