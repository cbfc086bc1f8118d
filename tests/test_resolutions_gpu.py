"""Model-level parity at the reference's OTHER latent resolutions.

Every other model-level GPU test, the tracked tile table and several kernel preconditions (band conv's LDS budget at
W = 50, the direct conv's G*H*W <= 384 rows, the 80-row dd_xattn320 tiles) live on the 28x50 pyramid of configs/exp/*.
The reference also ships 256x704 (32x88 latents: configs/exp-hd/256x704.yaml:11) and 432x768 (54x96:
configs/exp-hd/432x768.yaml:11, dual_branch_augloss_fusion_8pts_432x768.yaml) and 192x384 (24x48:
configs/exp-drive-wm/192x384.yaml).  Here: one fp16 multiview-UNet forward
(with ControlNet residuals) and one ControlNet forward per branch kind against the fp32 oracle at those sizes, 6
view-instances, reduced context; shapes that are not in the tracked table are tuned at run time, dispatchers fall back
where a fast kernel's precondition fails (tests/test_dispatch_predicates.py checks those predicates without a GPU).
Same bound as tests/test_model_gpu.py: rel-L2 <= max(1e-3, 1.02 x storage floor)."""
import os

import pytest
import torch

from oracle import dualdiff_restated as R
from oracle.init_utils import seeded_state_dict, seeded_tensor
from oracle.numerics import storage_emulation
from tests.parity_util import report

pytestmark = pytest.mark.gpu
PAIR = {0: [5, 1], 1: [0, 2], 2: [1, 3], 3: [2, 4], 4: [3, 5], 5: [4, 0]}
NCAM, NBOX, LTXT = 6, 5, 9
torch.set_num_threads(min(32, os.cpu_count() or 1))
RES = [(32, 88), (54, 96), (24, 48)]


def bf16_round(t):
    return t.to(torch.bfloat16).to(torch.float32) if t.is_floating_point() else t


def pyramid(h, w):
    out = [(h, w)]
    for _ in range(3):
        h, w = (h - 1) // 2 + 1, (w - 1) // 2 + 1
        out.append((h, w))
    return out


def _dev(x, dtype):
    if isinstance(x, dict):
        return {k: _dev(v, dtype) for k, v in x.items()}
    x = x.cuda()
    return x.to(dtype) if x.is_floating_point() else x


@pytest.mark.parametrize("hw", RES, ids=["%dx%d" % r for r in RES])
def test_unet_forward_other_resolution(gpu, hw):
    from dualdiff_amd.networks.unet_2d_condition_multiview import UNet2DConditionModelMultiview
    dtype = torch.float16
    h, w = hw
    ora = R.UNet2DConditionModelMultiview(cross_attention_dim=768, neighboring_view_pair=PAIR).eval()
    sd = {k: bf16_round(v) for k, v in seeded_state_dict(ora, 21).items()}
    ora.load_state_dict(sd)
    l0, l1, l2, l3 = pyramid(h, w)
    sample = bf16_round(seeded_tensor((NCAM, 4, h, w), 1))
    ctx = bf16_round(seeded_tensor((NCAM, 1 + LTXT + NBOX, 768), 2))
    shapes = [(320,) + l0] * 3 + [(320,) + l1] + [(640,) + l1] * 2 + [(640,) + l2] + [(1280,) + l2] * 2 + [(1280,) + l3] * 3
    down = [bf16_round(seeded_tensor((NCAM,) + s, 100 + i, 0.3)) for i, s in enumerate(shapes)]
    mid = bf16_round(seeded_tensor((NCAM, 1280) + l3, 130, 0.3))

    def run():
        return ora(sample, torch.tensor(481), encoder_hidden_states=ctx, down_block_additional_residuals=down,
                   mid_block_additional_residual=mid).sample
    def oracle():
        with torch.no_grad():
            ref = run()
            with storage_emulation(ora, dtype):
                emul = run()
        return {"ref": ref, "emul": emul}
    from tests.parity_util import oracle_cache
    o = oracle_cache("unet_forward_%dx%d_f16" % (h, w), oracle)
    ref, emul = o["ref"], o["emul"]
    net = UNet2DConditionModelMultiview(cross_attention_dim=768, neighboring_view_pair=PAIR)
    net.load_state_dict(sd, strict=True)
    net = net.to("cuda", dtype).eval()
    rec = []
    with torch.no_grad():
        out = net(sample.cuda().to(dtype), torch.tensor(481, device="cuda"), encoder_hidden_states=ctx.cuda().to(dtype),
                  down_block_additional_residuals=[d.cuda().to(dtype) for d in down],
                  mid_block_additional_residual=mid.cuda().to(dtype)).sample
        again = net(sample.cuda().to(dtype), torch.tensor(481, device="cuda"), encoder_hidden_states=ctx.cuda().to(dtype),
                    down_block_additional_residuals=[d.cuda().to(dtype) for d in down],
                    mid_block_additional_residual=mid.cuda().to(dtype)).sample       # the forward graph's replay
    assert out.shape == (NCAM, 4, h, w) and torch.equal(out, again)
    assert report("unet eps at %dx%d latents" % (h, w), out, ref, dtype, rec, emul) <= 1.0, rec


@pytest.mark.parametrize("occ3d", [False, True], ids=["bg_panorama", "fg_occ3d"])
@pytest.mark.parametrize("hw", RES, ids=["%dx%d" % r for r in RES])
def test_controlnet_forward_other_resolution(gpu, hw, occ3d):
    from tests.test_model_gpu import _make_cnet
    dtype = torch.float16
    h, w = hw
    b = 1
    g = torch.Generator().manual_seed(7)
    nv = 1 if occ3d else NCAM
    inp = {"sample": bf16_round(seeded_tensor((b, NCAM, 4, h, w), 11)), "timestep": torch.tensor([981]),
           "camera_param": bf16_round(seeded_tensor((b, NCAM, 3, 7), 12)), "text": bf16_round(seeded_tensor((b, LTXT, 768), 13)),
           "boxes": {"bboxes": bf16_round((torch.rand((b, nv, NBOX, 8, 3), generator=g) - 0.5) * 20.0),
                     "classes": torch.randint(0, 10, (b, nv, NBOX), generator=g),
                     "masks": torch.rand((b, nv, NBOX), generator=g) > 0.3},
           "cond": (bf16_round(torch.randint(0, 18, (b * NCAM, 320, h, w), generator=g).float() / 17.0) if occ3d
                    else bf16_round(torch.rand((b, 3, 8 * h, 8 * w * NCAM), generator=g)))}
    ora = R.BEVControlNetModel(use_occ_3d=occ3d).eval()
    sd = {k: bf16_round(v) for k, v in seeded_state_dict(ora, 31 + int(occ3d)).items()}
    ora.load_state_dict(sd)

    def run():
        return ora(inp["sample"], inp["timestep"], inp["camera_param"], inp["boxes"], inp["text"], inp["cond"],
                   conditioning_scale=0.75)
    with torch.no_grad():
        rdown, rmid, rctx = run()
        with storage_emulation(ora, dtype):
            edown, emid, ectx = run()
    net = _make_cnet(sd, occ3d, dtype)
    d = _dev(inp, dtype)
    with torch.no_grad():
        down, mid, ctx = net(d["sample"], d["timestep"], d["camera_param"], d["boxes"], d["text"], d["cond"],
                             conditioning_scale=0.75, return_dict=False, use_aug_text=False)
    l3 = pyramid(h, w)[3]
    assert len(down) == 12 and down[0].shape == (NCAM, 320, h, w) and mid.shape == (NCAM, 1280) + l3
    rec = []
    tag = "%dx%d %s" % (h, w, "fg" if occ3d else "bg")
    errs = [report("cnet down[%d] at %s" % (i, tag), a, r_, dtype, rec, e) for i, (a, r_, e) in enumerate(zip(down, rdown, edown))]
    errs.append(report("cnet mid at %s" % tag, mid, rmid, dtype, rec, emid))
    errs.append(report("cnet ctx tokens at %s" % tag, ctx, rctx, dtype, rec, ectx))
    assert max(errs) <= 1.0, rec
