"""End-to-end parity of the HIP model path against the CPU oracle (full SD-v1.5 widths).

Weights: oracle.init_utils.seeded_state_dict (zero-init modules get non-zero values), rounded to
bf16 so the SAME fp32 oracle run is exact-weight for both the fp16 and the bf16 HIP runs.
Inputs likewise.  The oracle runs in fp32 on the CPU; the HIP path stores activations in
fp16 / bf16 with fp32 accumulation.

Tolerance.  Metric: relative L2 error e(y) = ||y - ref||_2 / ||ref||_2 per output tensor against
the fp32 oracle.  BASELINE.json's north star asks for "1e-3 rel" on fp16 outputs; an fp16-storage
network of this depth cannot meet that against exact arithmetic — the REFERENCE's own numerics
cannot either — so the bound is stated relative to the reference-dtype noise floor measured on the
same inputs: e_floor = e(oracle with every leaf-module output rounded to the storage dtype,
oracle/numerics.py; a lower bound of the reference path's rounding noise).  Required:
        e(HIP) <= max(1e-3, 1.02 * e_floor)           (fp16 and bf16 alike; tests/parity_util.py)
i.e. 1e-3 wherever the dtype allows it, and never more than the noise the reference's own
storage dtype produces.  Both numbers are printed for every tensor and appended to the parity CSV
(profiles/r05_parity.csv is the tracked copy).
"""
import os

import pytest
import torch

from oracle import dualdiff_restated as R
from oracle.init_utils import seeded_state_dict, seeded_tensor
from oracle.numerics import storage_emulation

pytestmark = pytest.mark.gpu

PAIR = {0: [5, 1], 1: [0, 2], 2: [1, 3], 3: [2, 4], 4: [3, 5], 5: [4, 0]}
DTYPES = [torch.float16, torch.bfloat16]
torch.set_num_threads(min(32, os.cpu_count() or 1))


from tests.parity_util import bound, rel_l2, report  # noqa: E402,F401  (1.02 x floor bound + tracked CSV)

H, W, NCAM, NBOX, LTXT = 28, 50, 6, 5, 9


def bf16_round(t):
    return t.to(torch.bfloat16).to(torch.float32) if t.is_floating_point() else t


@pytest.fixture(scope="module")
def unet_case(gpu):
    torch.manual_seed(0)
    ora = R.UNet2DConditionModelMultiview(cross_attention_dim=768, neighboring_view_pair=PAIR).eval()
    sd = {k: bf16_round(v) for k, v in seeded_state_dict(ora, 21).items()}
    ora.load_state_dict(sd)
    m = NCAM
    sample = bf16_round(seeded_tensor((m, 4, H, W), 1))
    ctx = bf16_round(seeded_tensor((m, 1 + LTXT + NBOX, 768), 2))
    shapes = [(320, 28, 50)] * 3 + [(320, 14, 25)] + [(640, 14, 25)] * 2 + [(640, 7, 13)] + \
             [(1280, 7, 13)] * 2 + [(1280, 4, 7)] * 3
    down = [bf16_round(seeded_tensor((m,) + s, 100 + i, 0.3)) for i, s in enumerate(shapes)]
    mid = bf16_round(seeded_tensor((m, 1280, 4, 7), 130, 0.3))
    def run():
        return ora(sample, torch.tensor(481), encoder_hidden_states=ctx, down_block_additional_residuals=down,
                   mid_block_additional_residual=mid).sample

    def oracle():
        out = {}
        with torch.no_grad():
            out["ref"] = run()
            for dt in DTYPES:
                tag = str(dt).split(".")[-1]
                with storage_emulation(ora, dt):
                    out["emul_" + tag] = run()
                with storage_emulation(ora, dt, legacy=True):      # the round-1 floor, logged beside the current one
                    out["r1_" + tag] = run()
        return out
    from tests.parity_util import oracle_cache
    o = oracle_cache("unet_multiview_forward", oracle)
    ref, emul = o["ref"], {}
    for dt in DTYPES:
        tag = str(dt).split(".")[-1]
        emul[dt], emul[("r1", dt)] = o["emul_" + tag], o["r1_" + tag]
    return sd, sample, ctx, down, mid, ref, emul


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_unet_multiview_forward(unet_case, dtype):
    from dualdiff_amd.networks.unet_2d_condition_multiview import UNet2DConditionModelMultiview
    sd, sample, ctx, down, mid, ref, emul = unet_case
    net = UNet2DConditionModelMultiview(cross_attention_dim=768, neighboring_view_pair=PAIR)
    net.load_state_dict(sd, strict=True)
    net = net.to("cuda", dtype).eval()
    rec = []
    with torch.no_grad():
        out = net(sample.cuda().to(dtype), torch.tensor(481, device="cuda"),
                  encoder_hidden_states=ctx.cuda().to(dtype),
                  down_block_additional_residuals=[d.cuda().to(dtype) for d in down],
                  mid_block_additional_residual=mid.cuda().to(dtype)).sample
        out2 = net(sample.cuda().to(dtype), 981, encoder_hidden_states=ctx.cuda().to(dtype), return_dict=False)[0]
    assert out.shape == (NCAM, 4, H, W) and out.dtype == dtype
    assert out2.shape == out.shape and torch.isfinite(out2).all()      # scalar timestep / tuple return surface
    r = report("unet eps (with residuals)", out, ref, dtype, rec, emul[dtype], emul[("r1", dtype)])
    assert r <= 1.0, rec


def _cnet_inputs(b):
    g = torch.Generator().manual_seed(7)
    return {
        "sample": bf16_round(seeded_tensor((b, NCAM, 4, H, W), 11)),
        "timestep": torch.tensor([981, 41][:b]),          # int64 like the scheduler hands them over (pipeline_bev_controlnet.py:381); a float tensor would be cast to the storage dtype with the other inputs (bf16: 981 -> 980)
        "camera_param": bf16_round(seeded_tensor((b, NCAM, 3, 7), 12)),
        "text": bf16_round(seeded_tensor((b, LTXT, 768), 13)),
        "boxes_bg": {"bboxes": bf16_round((torch.rand((b, NCAM, NBOX, 8, 3), generator=g) - 0.5) * 20.0),
                     "classes": torch.randint(0, 10, (b, NCAM, NBOX), generator=g),
                     "masks": torch.rand((b, NCAM, NBOX), generator=g) > 0.3},
        "boxes_fg": {"bboxes": bf16_round((torch.rand((b, 1, NBOX, 8, 3), generator=g) - 0.5) * 20.0),
                     "classes": torch.randint(0, 10, (b, 1, NBOX), generator=g),
                     "masks": torch.rand((b, 1, NBOX), generator=g) > 0.3},
        "cond_bg": bf16_round(torch.rand((b, 3, 224, 2400), generator=g)),
        "cond_fg": bf16_round(torch.randint(0, 18, (b * NCAM, 320, H, W), generator=g).float() / 17.0),
    }


def _to_dev(x, dtype):
    if isinstance(x, dict):
        return {k: _to_dev(v, dtype) for k, v in x.items()}
    x = x.cuda()
    return x.to(dtype) if x.is_floating_point() else x


def _make_cnet(sd, occ3d, dtype):
    from dualdiff_amd.networks.unet_addon_rawbox import BEVControlNetModel
    net = BEVControlNetModel(cross_attention_dim=768)
    missing, unexpected = net.load_state_dict(sd, strict=False)
    assert not unexpected
    assert all(k.startswith(("adm_proj", "txt_con_fusionp", "controlnet_cond_embedding")) for k in missing), missing
    # the attribute protocol of misc/test_utils.py:123-136
    net.use_cam_in_temb = False
    net.use_box_adapter = False
    net.adm_proj = None
    net.use_txt_con_fusion = True
    net.use_txt_con_fusionp = False
    net.txt_con_fusionp = None
    net.use_occ_3d = occ3d
    if occ3d:
        net.controlnet_cond_embedding = None
    return net.to("cuda", dtype).eval()


@pytest.fixture(scope="module")
def cnet_case(gpu):
    out = {}
    inp = _cnet_inputs(2)
    for occ3d in (False, True):
        ora = R.BEVControlNetModel(use_occ_3d=occ3d).eval()
        sd = {k: bf16_round(v) for k, v in seeded_state_dict(ora, 31 + int(occ3d)).items()}
        ora.load_state_dict(sd)
        def run():
            return ora(inp["sample"], inp["timestep"], inp["camera_param"],
                       inp["boxes_fg" if occ3d else "boxes_bg"], inp["text"],
                       inp["cond_fg" if occ3d else "cond_bg"], conditioning_scale=0.75)

        with torch.no_grad():
            ref = run()
            emul = {}
            for dt in DTYPES:
                with storage_emulation(ora, dt):
                    emul[dt] = run()
        out[occ3d] = (sd, ref, emul)
    return inp, out


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("occ3d", [False, True], ids=["bg_panorama", "fg_occ3d"])
def test_controlnet_forward(cnet_case, occ3d, dtype):
    inp, refs = cnet_case
    sd, (rdown, rmid, rctx), emul = refs[occ3d]
    edown, emid, ectx = emul[dtype]
    net = _make_cnet(sd, occ3d, dtype)
    d = _to_dev(inp, dtype)
    with torch.no_grad():
        down, mid, ctx = net(d["sample"], d["timestep"], d["camera_param"],
                             d["boxes_fg" if occ3d else "boxes_bg"], d["text"],
                             d["cond_fg" if occ3d else "cond_bg"], conditioning_scale=0.75,
                             return_dict=False, use_aug_text=False)
    rec = []
    assert len(down) == 12
    errs = [report("cnet down[%d]" % i, a, b, dtype, rec, e) for i, (a, b, e) in enumerate(zip(down, rdown, edown))]
    errs.append(report("cnet mid", mid, rmid, dtype, rec, emid))
    errs.append(report("cnet ctx tokens", ctx, rctx, dtype, rec, ectx))
    assert down[0].shape == (12, 320, H, W) and mid.shape == (12, 1280, 4, 7)
    assert max(errs) <= 1.0, rec


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("plus", [False, True], ids=["sfa", "sfa_plus"])
@pytest.mark.parametrize("m", [6, 48], ids=["golden_case", "config3_batch4"])
def test_sfa_standalone(gpu, plus, m, dtype):
    """A12 / A13 standalone (BASELINE config 3: batch 4 scenes -> 48 view-instances): SFA and SFA+ on
    the HIP kernels vs the oracle restatement that is pinned to the reference's own modules
    (tests/golden/sfa*.npz); NCHW in / NCHW out like unet_addon_rawbox.py:974-978 calls it."""
    from oracle.init_utils import seeded_init_
    from tests.golden import cases as C
    from dualdiff_amd.networks.txt_con_fusion import txt_con_XFormersAttn, txt_con_XFormersAttn_plus
    ora = seeded_init_((R.TxtConFusionPlus if plus else R.TxtConFusion)(), C.SEED_SFA)
    ora.load_state_dict({k: bf16_round(v) for k, v in ora.state_dict().items()})
    x = bf16_round(seeded_tensor((m, 320, H, W), 101))
    e = bf16_round(seeded_tensor((m, 77, 768), 102))
    with torch.no_grad():
        ref = ora(x, e)
        with storage_emulation(ora, dtype):
            emul = ora(x, e)
    net = (txt_con_XFormersAttn_plus if plus else txt_con_XFormersAttn)()
    net.load_state_dict(ora.state_dict(), strict=True)
    net = net.to("cuda", dtype).eval()
    with torch.no_grad():
        out = net(attn=None, hidden_states=x.cuda().to(dtype), encoder_hidden_states=e.cuda().to(dtype))
    rec = []
    assert out.shape == ref.shape
    r = report("SFA%s standalone m=%d" % ("+" if plus else "", m), out, ref, dtype, rec, emul)    # floor bound (was a flat 8e-3 in bf16)
    assert r <= 1.0, rec


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_bev_map_embedder(gpu, dtype):
    """VERDICT r2 missing #4: vanilla MagicDrive's BEVControlNetConditioningEmbedding (map_embedder.py:10-77; asymmetric
    (2, 1) paddings, a (2, 1) stride) on the HIP convs vs the oracle restatement pinned to the reference's own module
    (tests/golden/bev_map_embedder.npz): (1, 25, 200, 200) map -> (6, 320, 28, 50), the same embedding for the 6 views."""
    from oracle.init_utils import seeded_init_
    from tests.golden import cases as C
    from dualdiff_amd.networks.map_embedder import BEVControlNetConditioningEmbedding
    ora = seeded_init_(R.BEVControlNetConditioningEmbedding(), C.SEED_BEV_EMB)
    ora.load_state_dict({k: bf16_round(v) for k, v in ora.state_dict().items()})
    bev = C.bev_map()
    with torch.no_grad():
        ref = ora(bev)
        with storage_emulation(ora, dtype):
            emul = ora(bev)
    net = BEVControlNetConditioningEmbedding()
    net.load_state_dict(ora.state_dict(), strict=True)
    net = net.to("cuda", dtype).eval()
    with torch.no_grad():
        out = net(bev.cuda().to(dtype))
    assert tuple(out.shape) == tuple(ref.shape) == (6, 320, 28, 50)
    assert torch.equal(out[0], out[5])
    rec = []
    assert report("BEV map embedder", out, ref, dtype, rec, emul) <= 1.0, rec


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_box_adapter_processor(gpu, dtype):
    """N1: Adapter_XFormersAttnProcessor on the HIP kernels vs the oracle restatement that
    is pinned to the reference's own `_real_call` (tests/golden/adapter_processor.npz), on the golden
    case (3 x 140 tokens, 13 text | 5 box | 5 class tokens, scale 0.7) and at the L0 layer shape."""
    from oracle import diffusers_restated as D
    from oracle.init_utils import seeded_init_
    from tests.golden import cases as C
    from dualdiff_amd.networks.box_adapter import Adapter_XFormersAttnProcessor
    from dualdiff_amd.networks.layers import Attention
    for (b, lq, lt, nt, scale) in ((3, 140, 13, 5, 0.7), (12, 1400, 78, 20, 1.0)):
        oa = seeded_init_(D.Attention(query_dim=320, cross_attention_dim=768, heads=8, dim_head=40), C.SEED_PROC)
        op = seeded_init_(R.AdapterAttnProcessor(320, 768, scale=scale), C.SEED_ADAPTER)
        for m in (oa, op):
            m.load_state_dict({k: bf16_round(v) for k, v in m.state_dict().items()})
        op.num_tokens = nt
        h = bf16_round(seeded_tensor((b, lq, 320), 115))
        ctx = bf16_round(seeded_tensor((b, lt + 2 * nt, 768), 116))
        with torch.no_grad():
            ref = op(oa, h, ctx)
        a = Attention(320, 768, 8, 40)
        a.load_state_dict(oa.state_dict())
        p = Adapter_XFormersAttnProcessor(320, 768, scale=scale)
        p.load_state_dict(op.state_dict())
        a.set_processor(p)
        a = a.to("cuda", dtype).eval()
        p.num_tokens = nt
        with torch.no_grad():
            out = a(h.cuda().to(dtype), encoder_hidden_states=ctx.cuda().to(dtype))
        e = rel_l2(out, ref)
        print("adapter processor b=%d lq=%d nt=%d %s: rel-L2 %.3e" % (b, lq, nt, dtype, e))
        assert out.shape == ref.shape and e <= (2e-3 if dtype == torch.float16 else 8e-3), e


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_controlnet_forward_box_adapter(cnet_case, dtype):
    """N1: ControlNet branch with use_box_adapter + the box_adapter() installer: class tokens reach the
    ControlNet's cross-attentions only; the tokens handed to the UNet stay [cam | text | box]."""
    from dualdiff_amd.networks.box_adapter import Adapter_XFormersAttnProcessor, box_adapter
    inp, refs = cnet_case
    sd = refs[False][0]
    ora = R.BEVControlNetModel(use_occ_3d=False).eval()
    ora.load_state_dict(sd)
    ora.use_box_adapter = True
    R.box_adapter(ora)
    def run():
        return ora(inp["sample"], inp["timestep"], inp["camera_param"], inp["boxes_bg"], inp["text"],
                   inp["cond_bg"], conditioning_scale=0.75)
    with torch.no_grad():
        rdown, rmid, rctx = run()
        with storage_emulation(ora, dtype):
            edown, emid, ectx = run()
    net = _make_cnet(sd, False, dtype)
    net.use_box_adapter = True
    box_adapter(net)
    assert sum(isinstance(p, Adapter_XFormersAttnProcessor) for p in net.attn_processors.values()) == 7
    assert any(k.endswith("attn2.processor.to_k_box.weight") for k in net.state_dict())
    d = _to_dev(inp, dtype)
    with torch.no_grad():
        down, mid, ctx = net(d["sample"], d["timestep"], d["camera_param"], d["boxes_bg"], d["text"],
                             d["cond_bg"], conditioning_scale=0.75, return_dict=False, use_aug_text=False)
    rec = []
    errs = [report("cnet+adapter down[%d]" % i, a, b, dtype, rec, e) for i, (a, b, e) in enumerate(zip(down, rdown, edown))]
    errs.append(report("cnet+adapter mid", mid, rmid, dtype, rec, emid))
    errs.append(report("cnet+adapter ctx tokens", ctx, rctx, dtype, rec, ectx))
    assert ctx.shape[1] == 1 + LTXT + NBOX                          # no class tokens towards the UNet
    assert max(errs) <= 1.0, rec
    # and the adapter is live: the result differs from the plain branch
    assert rel_l2(mid, refs[False][1][1]) > 1e-3


@pytest.mark.parametrize("dtype", [torch.float16])
def test_full_step_dual_branch_graph_vs_oracle(unet_case, cnet_case, dtype):
    """Two DDIM steps of the complete config-2 step (2 ControlNet branches, SFA on, CFG, b = 1 ->
    12 instances) through BEVDenoiser (HIP-graph replay) vs oracle.denoise_step."""
    from dualdiff_amd.networks.unet_2d_condition_multiview import UNet2DConditionModelMultiview
    from dualdiff_amd.pipeline.pipeline_bev_controlnet import BEVDenoiser, ddim_schedule
    usd = unet_case[0]
    inp, refs = cnet_case
    b = 1
    # CFG batch: uncond half first (zero text would do; use the 2 seeded scenes as uncond / cond)
    lat = bf16_round(seeded_tensor((b, 4, H, W), 77))[:, None].expand(-1, NCAM, -1, -1, -1).contiguous()
    prompt = inp["text"]                      # (2, L, 768): row 0 = uncond, row 1 = cond
    cam = inp["camera_param"]
    boxes = [inp["boxes_bg"], inp["boxes_fg"]]
    conds = [inp["cond_bg"], inp["cond_fg"]]
    # ---- oracle: 2 steps
    ounet = R.UNet2DConditionModelMultiview(cross_attention_dim=768, neighboring_view_pair=PAIR).eval()
    ounet.load_state_dict(usd)
    ocn = []
    for occ3d in (False, True):
        o = R.BEVControlNetModel(use_occ_3d=occ3d).eval()
        o.load_state_dict(refs[occ3d][0])
        ocn.append(o)
    ts, coefs = ddim_schedule(50)
    x = lat.clone()
    with torch.no_grad():
        for i in range(2):
            x = R.denoise_step(ounet, ocn, x, int(ts[i]), prompt, cam, boxes, conds, 2.0, coefs[i].tolist())
    # ---- HIP path
    unet = UNet2DConditionModelMultiview(cross_attention_dim=768, neighboring_view_pair=PAIR)
    unet.load_state_dict(usd)
    unet = unet.to("cuda", dtype).eval()
    cns = [_make_cnet(refs[o][0], o, dtype) for o in (False, True)]
    outs = {}
    for graph, hoist in ((True, False), (False, True)):
        den = BEVDenoiser(unet, cns, guidance_scale=2.0, num_inference_steps=50, use_graph=graph,
                          hoist_invariant=hoist)
        with torch.no_grad():
            den.set_inputs(lat.cuda().to(dtype), _to_dev(prompt, dtype), _to_dev(cam, dtype),
                           [_to_dev(bx, dtype) for bx in boxes], [_to_dev(c, dtype) for c in conds])
            den.run(2)
        outs[(graph, hoist)] = den.latents.float().cpu()
    # ---- CFG split: the two halves as separate 6-instance denoisers exchanging their
    # noise predictions every step (here through a local stand-in for parallel.cfg_all_gather)
    box = {}

    def make_exchange(hf):
        def exchange(eps_half):
            box[hf] = eps_half
            if len(box) < 2:
                return None
            return torch.stack([box[0], box[1]])
        return exchange

    halves = []
    for hf in (0, 1):
        d = BEVDenoiser(unet, cns, guidance_scale=2.0, num_inference_steps=50, use_graph=True,
                        cfg_half=hf, cfg_exchange=make_exchange(hf))
        d._combine_halves = lambda: None            # the test drives the exchange itself (one process)
        with torch.no_grad():
            d.set_inputs(lat.cuda().to(dtype), _to_dev(prompt, dtype), _to_dev(cam, dtype),
                         [_to_dev(bx, dtype) for bx in boxes], [_to_dev(c, dtype) for c in conds])
        halves.append(d)
    from dualdiff_amd import ops as O
    with torch.no_grad():
        for i in range(2):
            for d in halves:
                d.step(i)
            eps2 = torch.stack([halves[0]._eps_half, halves[1]._eps_half])
            for d in halves:
                O.cfg_ddim_step(eps2, d.lat2[0], d.coef, d.guidance_scale, x_out=d.lat2[0], x_dup=d.lat2[1])
    assert halves[0].m == 6 and torch.equal(halves[0].latents, halves[1].latents)
    rec = []
    e3 = report("latents after 2 steps (CFG split)", halves[0].latents.float().cpu(), x, dtype, rec)
    e = report("latents after 2 steps (graph)", outs[(True, False)], x, dtype, rec)
    e2 = report("latents after 2 steps (eager+hoist)", outs[(False, True)], x, dtype, rec)
    assert e3 <= 1.0, rec
    # graph replay and hoisting must not change results at all
    assert torch.equal(outs[(True, False)], outs[(False, True)])
    assert max(e, e2) <= 1.0, rec          # no floor given: plain 1e-3 on the latents


@pytest.mark.parametrize("dtype", [torch.bfloat16])
def test_unipc_denoiser_graph_equals_eager_and_restated_scheduler(unet_case, cnet_case, dtype):
    """N4: BEVDenoiser(sampler="unipc") — HIP-graph replay reproduces the eager run bit for bit (the capture
    warm-ups must not advance the multistep history), and the latents follow the restated UniPC scheduler
    applied to the noise predictions the HIP model produced."""
    from dualdiff_amd.networks.unet_2d_condition_multiview import UNet2DConditionModelMultiview
    from dualdiff_amd.pipeline.pipeline_bev_controlnet import BEVDenoiser
    from oracle.unipc import UniPCRestated
    usd = unet_case[0]
    inp, refs = cnet_case
    lat = bf16_round(seeded_tensor((1, 4, H, W), 78))[:, None].expand(-1, NCAM, -1, -1, -1).contiguous()
    unet = UNet2DConditionModelMultiview(cross_attention_dim=768, neighboring_view_pair=PAIR)
    unet.load_state_dict(usd)
    unet = unet.to("cuda", dtype).eval()
    cns = [_make_cnet(refs[False][0], False, dtype)]
    nsteps, run = 20, 4
    outs, eps_log = {}, []
    for graph in (False, True):
        den = BEVDenoiser(unet, cns, guidance_scale=2.0, num_inference_steps=nsteps, use_graph=graph,
                          sampler="unipc")
        with torch.no_grad():
            den.set_inputs(lat.cuda().to(dtype), _to_dev(inp["text"], dtype), _to_dev(inp["camera_param"], dtype),
                           [_to_dev(inp["boxes_bg"], dtype)], [_to_dev(inp["cond_bg"], dtype)])
            if not graph:
                body = den._step_body

                def logged():
                    e = body()
                    eps_log.append(e.float().cpu().clone().reshape(2, NCAM, 4, H, W))
                    return e
                den._step_body = logged
            den.run(run)
        outs[graph] = den.latents.float().cpu()
    assert torch.equal(outs[True], outs[False])
    assert len(eps_log) == run
    sch = UniPCRestated()
    ts = sch.set_timesteps(nsteps)
    assert list(den.timesteps[:run]) == list(ts[:run])
    x = lat.reshape(NCAM, 4, H, W).to(dtype).double()
    for i in range(run):
        e = eps_log[i]
        guided = (e[0] + 2.0 * (e[1] - e[0])).to(dtype).double()
        x = sch.step(guided, int(ts[i]), x).to(dtype).double()      # latents are stored in the model dtype
    rec = []
    err = report("unipc latents after %d steps" % run, outs[True].reshape(NCAM, 4, H, W), x.float(), dtype, rec)
    assert err <= 1.0, rec


@pytest.mark.parametrize("dtype", DTYPES)
def test_multiview_block_standalone(gpu, dtype):
    """One BasicMultiviewTransformerBlock driven directly (no UNet around it: no cross-attention K/V bank,
    no proj_out fold) at the L1 shape — the configuration `__graft_entry__.smoke()` runs."""
    from dualdiff_amd.networks.blocks import BasicMultiviewTransformerBlock
    ora = R.BasicMultiviewTransformerBlock(640, 8, 80, cross_attention_dim=768, neighboring_view_pair=PAIR).eval()
    sd = {k: bf16_round(v) for k, v in seeded_state_dict(ora, 5).items()}
    ora.load_state_dict(sd)
    x = bf16_round(seeded_tensor((6, 350, 640), 1))
    ctx = bf16_round(seeded_tensor((6, 30, 768), 2))
    with torch.no_grad():
        ref = ora(x, encoder_hidden_states=ctx)
        with storage_emulation(ora, dtype):
            emul = ora(x, encoder_hidden_states=ctx)
        blk = BasicMultiviewTransformerBlock(640, 8, 80, cross_attention_dim=768, neighboring_view_pair=PAIR)
        blk.load_state_dict(sd)
        blk = blk.to("cuda", dtype)
        y = blk.run(x.cuda().to(dtype).reshape(-1, 640), 6, 350, ctx.cuda().to(dtype).reshape(-1, 768), 30)
    rec = []
    assert report("multiview block standalone", y.reshape(6, 350, 640), ref, dtype, rec, emul) <= 1.0, rec


def _variant_pair(attn_type, zero_type, seed=7):
    """(oracle block, state dict, inputs) of one neighboring_attn_type / zero_module_type setting at the L1 shape."""
    ora = R.BasicMultiviewTransformerBlock(640, 8, 80, cross_attention_dim=768, neighboring_view_pair=PAIR,
                                           neighboring_attn_type=attn_type, zero_module_type=zero_type).eval()
    sd = {k: bf16_round(v) for k, v in seeded_state_dict(ora, seed).items()}
    ora.load_state_dict(sd)
    return ora, sd, bf16_round(seeded_tensor((12, 350, 640), 3)), bf16_round(seeded_tensor((12, 30, 768), 4))


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("attn_type,zero_type", [("concat", "zero_linear"), ("self", "zero_linear"),
                                                 ("add", "gated"), ("add", "none"), ("concat", "gated")])
def test_multiview_block_variants(gpu, attn_type, zero_type, dtype):
    """The block's other settings (reference blocks.py:81-90 connector kinds, :106-142 neighbour attention kinds;
    the oracle's restatement of them is pinned by tests/golden/multiview_block_variants.npz): 2 scenes x 6 views."""
    from dualdiff_amd.networks.blocks import BasicMultiviewTransformerBlock
    ora, sd, x, ctx = _variant_pair(attn_type, zero_type)
    with torch.no_grad():
        ref = ora(x, encoder_hidden_states=ctx)
        with storage_emulation(ora, dtype):
            emul = ora(x, encoder_hidden_states=ctx)
        blk = BasicMultiviewTransformerBlock(640, 8, 80, cross_attention_dim=768, neighboring_view_pair=PAIR,
                                             neighboring_attn_type=attn_type, zero_module_type=zero_type)
        blk.load_state_dict(sd)
        blk = blk.to("cuda", dtype)
        y = blk.run(x.cuda().to(dtype).reshape(-1, 640), 12, 350, ctx.cuda().to(dtype).reshape(-1, 768), 30)
    rec = []
    name = "multiview block %s/%s" % (attn_type, zero_type)
    assert report(name, y.reshape(12, 350, 640), ref, dtype, rec, emul) <= 1.0, rec
    # the variant matters: the default block on the same weights gives a different answer
    if attn_type != "add":
        base = R.BasicMultiviewTransformerBlock(640, 8, 80, cross_attention_dim=768, neighboring_view_pair=PAIR,
                                                zero_module_type=zero_type).eval()
        base.load_state_dict(sd)
        with torch.no_grad():
            assert rel_l2(base(x, encoder_hidden_states=ctx), ref) > 1e-2


def test_multiview_block_unknown_variant_raises(gpu):
    from dualdiff_amd.networks.blocks import BasicMultiviewTransformerBlock
    with pytest.raises(NotImplementedError):                 # blocks.py:140-142
        BasicMultiviewTransformerBlock(640, 8, 80, cross_attention_dim=768, neighboring_view_pair=PAIR,
                                       neighboring_attn_type="ring")
    with pytest.raises(TypeError):                           # blocks.py:89-90
        BasicMultiviewTransformerBlock(640, 8, 80, cross_attention_dim=768, neighboring_view_pair=PAIR,
                                       zero_module_type="relu")


def test_foreign_checkpoint_import_forward(gpu, tmp_path):
    """a diffusers-layout `unet/` folder written by other code (oracle weights, hand-written 0.17.1
    config.json, safetensors) -> `from_pretrained(..., torch_dtype=fp16)` -> `.to('cuda')` -> `forward()` reproduces the
    oracle that produced the weights (real SD-v1.5 widths, one layer per block)."""
    from dualdiff_amd.networks.unet_2d_condition_multiview import UNet2DConditionModelMultiview
    from tests.test_host_logic import write_foreign_unet_checkpoint
    dtype = torch.float16
    ora, sd = write_foreign_unet_checkpoint(os.path.join(tmp_path, "unet"), "safetensors", seed=23,
                                            block_out_channels=[320, 640, 1280, 1280], cross_attention_dim=768)
    net = UNet2DConditionModelMultiview.from_pretrained(str(tmp_path), subfolder="unet", torch_dtype=dtype).to("cuda")
    x = bf16_round(seeded_tensor((6, 4, 28, 50), 1))
    ctx = bf16_round(seeded_tensor((6, 20, 768), 2))
    t = torch.tensor([601])
    with torch.no_grad():
        y = net(x.cuda().to(dtype), t.cuda(), encoder_hidden_states=ctx.cuda().to(dtype)).sample
    rec = []
    # the checkpoint holds fp32 values and from_pretrained rounds them to fp16: the oracle runs ON THE LOADED VALUES
    ora.load_state_dict({k: v.float().cpu() for k, v in net.state_dict().items()})
    with torch.no_grad():
        ref = ora(x, t, encoder_hidden_states=ctx).sample
        with storage_emulation(ora, dtype):
            emul = ora(x, t, encoder_hidden_states=ctx).sample
    assert report("foreign checkpoint unet eps", y, ref, dtype, rec, emul) <= 1.0, rec


@pytest.mark.parametrize("case", ["no_boxes", "all_boxes_masked", "one_box"])
def test_controlnet_forward_box_edge_cases(cnet_case, case):
    """Edge cases of the box-token path (unet_addon_rawbox.py:852-896, bbox_embedder.py:164-203): `bboxes_3d_data=None` (the
    context is [cam | text] only), every box masked out (all box tokens are the learned null token), a single box — HIP vs
    the oracle on the panorama branch, fp16, same floor-based bound as the main ControlNet case."""
    dtype = torch.float16
    inp, refs = cnet_case
    sd = refs[False][0]
    boxes = inp["boxes_bg"]
    if case == "no_boxes":
        boxes = None
    elif case == "all_boxes_masked":
        boxes = dict(boxes, masks=torch.zeros_like(boxes["masks"]))
    else:
        boxes = {k: v[:, :, :1].contiguous() for k, v in boxes.items()}
    ora = R.BEVControlNetModel(use_occ_3d=False).eval()
    ora.load_state_dict(sd)

    def run():
        return ora(inp["sample"], inp["timestep"], inp["camera_param"], boxes, inp["text"], inp["cond_bg"],
                   conditioning_scale=0.75)
    with torch.no_grad():
        rdown, rmid, rctx = run()
        with storage_emulation(ora, dtype):
            edown, emid, ectx = run()
    net = _make_cnet(sd, False, dtype)
    d = _to_dev(inp, dtype)
    with torch.no_grad():
        down, mid, ctx = net(d["sample"], d["timestep"], d["camera_param"], None if boxes is None else _to_dev(boxes, dtype),
                             d["text"], d["cond_bg"], conditioning_scale=0.75, return_dict=False, use_aug_text=False)
    want_lc = 1 + LTXT + (0 if boxes is None else boxes["bboxes"].shape[2])
    assert ctx.shape == (12, want_lc, 768) == tuple(rctx.shape)
    rec = []
    errs = [report("cnet %s ctx tokens" % case, ctx, rctx, dtype, rec, ectx),
            report("cnet %s down[0]" % case, down[0], rdown[0], dtype, rec, edown[0]),
            report("cnet %s down[11]" % case, down[11], rdown[11], dtype, rec, edown[11]),
            report("cnet %s mid" % case, mid, rmid, dtype, rec, emid)]
    assert max(errs) <= 1.0, rec
    if case == "all_boxes_masked":          # every box token is the same learned null token
        assert torch.equal(ctx[:, 1 + LTXT], ctx[:, -1]) and torch.equal(ctx[0, 1 + LTXT], ctx[7, 1 + LTXT])
