"""Round-2 parity cases the judge asked for (VERDICT.md "Next round" item 1), HIP path vs the CPU oracle:

  * BASELINE configs[0]: one view, plain SD-v1.5 UNet (no attn4, no ControlNet), null text, one DDIM step at
    t = 981 — fp16 and bf16, with the storage-dtype noise floor;
  * the full dual-branch step in bf16 (the bench dtype) and a 50-step fp16 DDIM trajectory against the
    oracle trajectory minted by tests/golden/mint_trajectory.py (drift curve -> parity CSV);
  * a FOREIGN attention processor (the shape of the reference's tools/unet_modify.py:7-57) installed through
    the B3 protocol inside a transformer block on the GPU, and SPLIT_SIZE batch chunking
    (box_adapter.py:41-64) through a whole ControlNet branch;
  * ControlNet surface additions: guess_mode residual scales, add_uncond_to_emb, 3-branch residual sum.

Metric / bound / CSV: tests/parity_util.py.
"""
import os

import numpy as np
import pytest
import torch

from oracle import diffusers_restated as D
from oracle import dualdiff_restated as R
from oracle.init_utils import seeded_init_, seeded_state_dict, seeded_tensor
from oracle.numerics import storage_emulation
from tests.golden import cases as C
from tests.parity_util import bound, log_row, rel_l2, report

pytestmark = pytest.mark.gpu

PAIR = C.VIEW_PAIR
H, W, NCAM = C.H, C.W, C.N_CAM
DTYPES = [torch.float16, torch.bfloat16]
torch.set_num_threads(min(32, os.cpu_count() or 1))
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "trajectory_ddim50.npz")


def _to_dev(x, dtype):
    if isinstance(x, dict):
        return {k: _to_dev(v, dtype) for k, v in x.items()}
    x = x.cuda()
    return x.to(dtype) if x.is_floating_point() else x


# ------------------------------------------------------------------ BASELINE configs[0] ----
@pytest.fixture(scope="module")
def config0_case(gpu):
    """latents N(0,1) (1,4,28,50) seed 0, zeros text (1,77,768), t = 981, plain SD-1.5."""
    ora = D.UNet2DConditionModel(cross_attention_dim=768).eval()
    sd = {k: C.bf16_round(v) for k, v in seeded_state_dict(ora, 41).items()}
    ora.load_state_dict(sd)
    x = C.bf16_round(seeded_tensor((1, 4, H, W), 0))
    txt = torch.zeros((1, 77, 768))
    ts, ratio = R.ddim_timesteps(50)
    assert int(ts[0]) == 981
    coef = R.ddim_coefs(R.ddim_alphas(), 981, ratio)

    def run(dt=None):
        eps = ora(x, torch.tensor(981), encoder_hidden_states=txt).sample
        x0 = (x - coef[1] * eps) / coef[0]
        xn = coef[2] * x0 + coef[3] * eps
        return eps, (xn if dt is None else xn.to(dt).float())       # the latents are stored in the model dtype

    with torch.no_grad():
        ref = run()
        emul = {}
        for dt in DTYPES:
            with storage_emulation(ora, dt):
                emul[dt] = run(dt)
    return sd, x, txt, coef, ref, emul


@pytest.mark.parametrize("dtype", DTYPES)
def test_config0_single_view_sd15_one_ddim_step(config0_case, dtype):
    from dualdiff_amd import ops as O
    from dualdiff_amd.networks.unet_2d_condition_multiview import UNet2DConditionModel
    sd, x, txt, coef, (ref_eps, ref_x), emul = config0_case
    net = UNet2DConditionModel(cross_attention_dim=768)
    net.load_state_dict(sd, strict=True)                    # the stock SD-1.5 key set, nothing extra
    assert not any("attn4" in k or "connector" in k for k in net.state_dict())
    net = net.to("cuda", dtype).eval()
    with torch.no_grad():
        xd = x.cuda().to(dtype)
        eps = net(xd, 981, encoder_hidden_states=txt.cuda().to(dtype)).sample
        # no CFG on this config: feed the same prediction as both halves with guidance 1
        xn = O.cfg_ddim_step(torch.stack([eps, eps]), xd, torch.tensor(coef, dtype=torch.float32, device="cuda"), 1.0)
    rec = []
    r1 = report("config0 sd15 1-view eps", eps, ref_eps, dtype, rec, emul[dtype][0])
    r2 = report("config0 sd15 1-view x after DDIM step", xn, ref_x, dtype, rec, emul[dtype][1])
    assert max(r1, r2) <= 1.0, rec


# ------------------------------------------------------- full step bf16 / 50-step trajectory ----
def _gold():
    if not os.path.exists(GOLD):
        pytest.skip("tests/golden/trajectory_ddim50.npz not minted")
    with np.load(GOLD) as z:
        return {k: torch.from_numpy(z[k]) for k in z.files}


@pytest.fixture(scope="module")
def step_models(gpu):
    """State dicts of the full-width step case (tests/golden/cases.py) — weights only, no oracle forward."""
    ou = R.UNet2DConditionModelMultiview(cross_attention_dim=768, neighboring_view_pair=PAIR)
    usd = {k: C.bf16_round(v) for k, v in seeded_state_dict(ou, C.SEED_STEP_UNET).items()}
    del ou
    csd = []
    for occ3d, seed in ((False, C.SEED_STEP_CNET_BG), (True, C.SEED_STEP_CNET_FG)):
        oc = R.BEVControlNetModel(use_occ_3d=occ3d)
        csd.append({k: C.bf16_round(v) for k, v in seeded_state_dict(oc, seed).items()})
        del oc
    return usd, csd


def _make_cnet(sd, occ3d, dtype):
    from dualdiff_amd.networks.unet_addon_rawbox import BEVControlNetModel
    net = BEVControlNetModel(cross_attention_dim=768)
    net.load_state_dict(sd, strict=False)
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


def _denoiser(step_models, dtype, **kw):
    from dualdiff_amd.networks.unet_2d_condition_multiview import UNet2DConditionModelMultiview
    from dualdiff_amd.pipeline.pipeline_bev_controlnet import BEVDenoiser
    usd, csd = step_models
    unet = UNet2DConditionModelMultiview(cross_attention_dim=768, neighboring_view_pair=PAIR)
    unet.load_state_dict(usd)
    unet = unet.to("cuda", dtype).eval()
    cns = [_make_cnet(csd[0], False, dtype), _make_cnet(csd[1], True, dtype)]
    inp = C.step_inputs(2)
    den = BEVDenoiser(unet, cns, guidance_scale=2.0, num_inference_steps=50, **kw)
    with torch.no_grad():
        den.set_inputs(C.step_latents().cuda().to(dtype), _to_dev(inp["text"], dtype),
                       _to_dev(inp["camera_param"], dtype),
                       [_to_dev(inp["boxes_bg"], dtype), _to_dev(inp["boxes_fg"], dtype)],
                       [_to_dev(inp["cond_bg"], dtype), _to_dev(inp["cond_fg"], dtype)])
    return den


def test_full_step_dual_branch_bf16_vs_oracle(step_models):
    """The bench dtype: two DDIM steps of the complete config-2 step (HIP-graph replay) in bf16 against the
    oracle's latents, with the bf16 storage floor of the same two steps (fixture floor_bf16_1/2)."""
    g = _gold()
    if "floor_bf16_2" not in g:
        pytest.skip("bf16 floor not minted")
    dtype = torch.bfloat16
    den = _denoiser(step_models, dtype, use_graph=True)
    rec, worst = [], 0.0
    with torch.no_grad():
        for k in (1, 2):
            den.step(k - 1)
            worst = max(worst, report("dual-branch latents after %d steps" % k, den.latents[0].float().cpu(),
                                      g["ref_%d" % k], dtype, rec, g["floor_bf16_%d" % k]))
    assert worst <= 1.0, rec


def test_trajectory_50_steps_fp16(step_models):
    """One whole 50-step DDIM sample in fp16 (the reference's dtype) replayed from the HIP graph, compared at
    the checkpoints with the fp32 oracle trajectory; the fp16-storage oracle trajectory is the floor.  The
    drift curve goes to the parity CSV.  Bound: max(1e-3, 1.02 x floor) at EVERY checkpoint (measured: the HIP
    curve stays just below the floor's all the way, 1.46e-3 vs 1.47e-3 after 50 steps)."""
    g = _gold()
    dtype = torch.float16
    den = _denoiser(step_models, dtype, use_graph=True)
    ks = [k for k in C.TRAJ_CHECKPOINTS if "ref_%d" % k in g]
    if not ks:
        pytest.skip("trajectory not minted")
    rec, bad = [], []
    with torch.no_grad():
        for i in range(max(ks)):
            den.step(i)
            k = i + 1
            if k in ks:
                y = den.latents[0].float().cpu()
                e = rel_l2(y, g["ref_%d" % k])
                fl = rel_l2(g["floor_f16_%d" % k], g["ref_%d" % k]) if "floor_f16_%d" % k in g else 0.0
                bnd = bound(fl)
                print("trajectory step %2d: e_hip=%.3e e_floor=%.3e bound=%.3e" % (k, e, fl, bnd))
                log_row("ddim50 trajectory step %d" % k, dtype, e, fl, bnd)
                rec.append((k, e, fl))
                assert torch.isfinite(y).all()
                if e > bnd:
                    bad.append((k, e, fl))
    assert not bad, rec


# ------------------------------------------------------------- foreign processor through B3 ----
class ForeignCrossAttnProcessor:
    """A processor written against the diffusers `Attention` surface only — the shape of the reference's
    tools/unet_modify.py:7-57 (to_q/to_k/to_v as callables, head_to_batch_dim, get_attention_scores kept for
    inspection, bmm, batch_to_head_dim, to_out[0], to_out[1])."""

    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None):
        batch_size, sequence_length, _ = hidden_states.shape
        attention_mask = attn.prepare_attention_mask(attention_mask, sequence_length, batch_size)
        query = attn.to_q(hidden_states)
        encoder_hidden_states = encoder_hidden_states if encoder_hidden_states is not None else hidden_states
        key = attn.to_k(encoder_hidden_states)
        value = attn.to_v(encoder_hidden_states)
        query = attn.head_to_batch_dim(query)
        key = attn.head_to_batch_dim(key)
        value = attn.head_to_batch_dim(value)
        attention_probs = attn.get_attention_scores(query, key, attention_mask)
        attn.attn_probs_original = attention_probs.chunk(2)[1]
        hidden_states = torch.bmm(attention_probs, value)
        hidden_states = attn.batch_to_head_dim(hidden_states)
        hidden_states = attn.to_out[0](hidden_states)
        hidden_states = attn.to_out[1](hidden_states)
        return hidden_states


@pytest.mark.parametrize("dtype", DTYPES)
def test_foreign_processor_in_multiview_block(gpu, dtype):
    """The block's attn2 runs a foreign callable (set_processor), attn1 / attn4 / FF stay on the fused path;
    the result must still match the oracle block (the processor computes the same attention)."""
    from dualdiff_amd.networks.blocks import BasicMultiviewTransformerBlock
    ora = R.BasicMultiviewTransformerBlock(640, 8, 80, cross_attention_dim=768, neighboring_view_pair=PAIR).eval()
    sd = {k: C.bf16_round(v) for k, v in seeded_state_dict(ora, 5).items()}
    ora.load_state_dict(sd)
    x = C.bf16_round(seeded_tensor((6, 350, 640), 1))
    ctx = C.bf16_round(seeded_tensor((6, 30, 768), 2))
    with torch.no_grad():
        ref = ora(x, encoder_hidden_states=ctx)
        with storage_emulation(ora, dtype):
            emul = ora(x, encoder_hidden_states=ctx)
        blk = BasicMultiviewTransformerBlock(640, 8, 80, cross_attention_dim=768, neighboring_view_pair=PAIR)
        blk.load_state_dict(sd)
        blk = blk.to("cuda", dtype)
        blk.attn2.set_processor(ForeignCrossAttnProcessor())
        y = blk.run(x.cuda().to(dtype).reshape(-1, 640), 6, 350, ctx.cuda().to(dtype).reshape(-1, 768), 30)
    assert blk.attn2.attn_probs_original.shape == (3 * 8, 350, 30)          # the foreign code really ran
    rec = []
    assert report("multiview block, foreign attn2 processor", y.reshape(6, 350, 640), ref, dtype, rec, emul) <= 1.0, rec


@pytest.mark.parametrize("dtype", [torch.float16])
def test_split_size_chunking_controlnet(step_models, dtype, monkeypatch):
    """SPLIT_SIZE (box_adapter.py:11,41-64): with the reference's XFormersAttnProcessor installed on every
    attention and SPLIT_SIZE = 5, each attention call is chunked 12 -> 4 + 4 + 4 instances (torch.chunk(3)); residuals must
    equal the unchunked run's within the storage rounding (different GEMM row counts pick other tiles)."""
    from dualdiff_amd.networks import box_adapter as BA
    usd, csd = step_models
    inp = C.step_inputs(2)
    d = _to_dev(inp, dtype)
    net = _make_cnet(csd[1], True, dtype)
    net.set_attn_processor(BA.XFormersAttnProcessor())
    net.graph_forward = False        # the test counts Python-side processor calls: a replayed forward graph makes none
    args = (d["sample"], d["timestep"], d["camera_param"], d["boxes_fg"], d["text"], d["cond_fg"])
    with torch.no_grad():
        base = net(*args, return_dict=False, use_aug_text=False)
        monkeypatch.setattr(BA, "SPLIT_SIZE", 5)
        calls = []
        real = BA.XFormersAttnProcessor._real_call

        def counted(self, attn, hs, *a, **k):
            calls.append(hs.shape[0])
            return real(self, attn, hs, *a, **k)
        monkeypatch.setattr(BA.XFormersAttnProcessor, "_real_call", counted)
        chunked = net(*args, return_dict=False, use_aug_text=False)
    assert calls and set(calls) == {4} and len(calls) == 3 * 14    # 7 blocks x (attn1, attn2) x chunks of 12 -> 4 + 4 + 4
    for i, (a, b) in enumerate(zip(base[0] + [base[1]], chunked[0] + [chunked[1]])):
        e = rel_l2(a, b.float().cpu())
        assert e <= 2e-3, (i, e)


# ------------------------------------------------------------------- ControlNet surface ----
@pytest.mark.parametrize("dtype", [torch.float16])
def test_controlnet_guess_mode_scales(step_models, dtype):
    """guess_mode (unet_addon_rawbox.py:1042-1050): residual i scaled by logspace(-1, 0, 13)[i] * scale."""
    usd, csd = step_models
    inp = C.step_inputs(2)
    ora = R.BEVControlNetModel(use_occ_3d=True).eval()
    ora.load_state_dict(csd[1])
    with torch.no_grad():
        rdown, rmid, _ = ora(inp["sample"], inp["timestep"], inp["camera_param"], inp["boxes_fg"], inp["text"],
                             inp["cond_fg"], conditioning_scale=0.5, guess_mode=True)
        with storage_emulation(ora, dtype):
            edown, emid, _ = ora(inp["sample"], inp["timestep"], inp["camera_param"], inp["boxes_fg"], inp["text"],
                                 inp["cond_fg"], conditioning_scale=0.5, guess_mode=True)
    net = _make_cnet(csd[1], True, dtype)
    d = _to_dev(inp, dtype)
    with torch.no_grad():
        down, mid, _ = net(d["sample"], d["timestep"], d["camera_param"], d["boxes_fg"], d["text"], d["cond_fg"],
                           conditioning_scale=0.5, guess_mode=True, return_dict=False, use_aug_text=False)
    rec = []
    errs = [report("cnet guess_mode down[%d]" % i, a, b, dtype, rec, e) for i, (a, b, e) in enumerate(zip(down, rdown, edown))]
    errs.append(report("cnet guess_mode mid", mid, rmid, dtype, rec, emid))
    assert max(errs) <= 1.0, rec


def test_add_uncond_to_emb(step_models):
    """add_uncond_to_emb (unet_addon_rawbox.py:771-789, by intent — the reference body cannot execute):
    [uncond camera token | text | null box tokens] x N_cam in front of the conditional tokens."""
    dtype = torch.float16
    usd, csd = step_models
    net = _make_cnet(csd[0], False, dtype)
    inp = _to_dev(C.step_inputs(2), dtype)
    with torch.no_grad():
        tok = net.prepare_tokens(inp["camera_param"][1:], {k: v[1:] for k, v in inp["boxes_bg"].items()},
                                 inp["text"][1:], False)["ctx"]                 # (6, 1 + 9 + 5, 768) conditional
        out = net.add_uncond_to_emb(inp["text"][:1], NCAM, tok)
        assert out.shape == (2 * NCAM, tok.shape[1], 768)
        assert torch.equal(out[NCAM:], tok)
        # expected uncond rows, token by token, from the same modules
        cam = net.cam2token.run(net._embed_camera(net.uncond_cam_param([1, 1])).reshape(1, -1).to(dtype).contiguous())
        null_box = net.bbox_embedder.forward_feature(net.bbox_embedder.null_pos_feature[None],
                                                     net.bbox_embedder.null_class_feature[None])
        for v in range(NCAM):
            assert torch.equal(out[v, 0], cam[0])
            assert torch.equal(out[v, 1:1 + C.STEP_LTXT], inp["text"][0])
            for j in range(C.STEP_NBOX):
                assert torch.equal(out[v, 1 + C.STEP_LTXT + j], null_box[0].to(dtype))


def test_three_branch_residual_sum(gpu):
    """decode_nhwc's residual add must sum ANY number of ControlNet branches
    (pipeline_bev_controlnet.py:421-429); parallel_branches hands it tuples of per-branch tensors."""
    from dualdiff_amd import ops as O
    dtype = torch.float16
    from dualdiff_amd.networks.layers import device_init_
    from dualdiff_amd.networks.unet_2d_condition_multiview import UNet2DConditionModel
    with torch.device("cuda"):
        net = UNet2DConditionModel(cross_attention_dim=768).to(dtype)      # SD-1.5 widths (head dims 40 / 80 / 160)
    device_init_(net, 3)
    x = torch.randn((2, 4, 16, 16), device="cuda").to(dtype)
    ctx = torch.randn((2, 7, 768), device="cuda").to(dtype)
    with torch.no_grad():
        x8 = torch.nn.functional.pad(O.nchw_to_nhwc(x), (0, 4))
        tf = torch.full((2,), 500.0, device="cuda")
        st = net.encode_nhwc(x8, 2, 16, 16, tf, ctx.reshape(-1, 768), 7)
        shapes = [s[0].shape for s in st["skips"]]
        for nb in (1, 2, 3, 4):
            branches = [[torch.randn(sh, device="cuda").to(dtype) * 0.1 for sh in shapes] for _ in range(nb)]
            mids = [torch.randn(st["x"].shape, device="cuda").to(dtype) * 0.1 for _ in range(nb)]
            tup = net.decode_nhwc(dict(st), [tuple(b[j] for b in branches) for j in range(len(shapes))], tuple(mids))
            # serial reference: pre-summed residuals in fp32, rounded once
            summed = [sum(b[j].float() for b in branches).to(dtype) for j in range(len(shapes))]
            ser = net.decode_nhwc(dict(st), summed, sum(m.float() for m in mids).to(dtype))
            assert rel_l2(tup, ser.float().cpu()) <= 5e-3, nb
            if nb >= 3:      # dropping the third branch would be a large error
                two = net.decode_nhwc(dict(st), [tuple(b[j] for b in branches[:2]) for j in range(len(shapes))], tuple(mids[:2]))
                assert rel_l2(two, ser.float().cpu()) > 2e-2


# ------------------------------------------------------------------ view split on one GPU ----
class _LocalExchange:
    """In-process stand-in for parallel.HaloExchange: the shards of one half group run as threads of this
    process (one HIP stream each) and copy the neighbour views' K/V out of each other's buffers."""

    def __init__(self, plans):
        import threading
        self.plans = {p.shard: p for p in plans}
        self.barrier = threading.Barrier(len(plans))
        self.bufs = {}

    def bind(self, plan):
        def exchange(kv):
            torch.cuda.current_stream().synchronize()
            self.bufs[plan.shard] = kv
            self.barrier.wait()
            for s, views in plan.recv.items():
                peer = self.plans[s]
                for v in views:
                    kv[plan.slot(v)].copy_(self.bufs[s][peer.slot(v)])
            torch.cuda.current_stream().synchronize()
            self.barrier.wait()
            return kv
        return exchange


@pytest.mark.parametrize("shards", [2, 3])
def test_view_split_denoiser_one_gpu(step_models, shards):
    """SURVEY §8e view split, whole sampler: the 6 views of the scene spread over `shards` virtual ranks
    (threads on one GPU, both CFG halves per rank), each running ControlNet branches + UNet on its own
    view-instances and fetching neighbour K/V per transformer block — two DDIM steps must reproduce the
    unsharded denoiser's latents up to the storage rounding (row counts differ, so GEMM tiles do)."""
    import threading
    from dualdiff_amd.parallel import ViewShard, ViewSplitPlan
    dtype = torch.float16
    full = _denoiser(step_models, dtype, use_graph=False)
    with torch.no_grad():
        full.run(2)
    want = full.latents.float().cpu()                                   # (1, 6, 4, h, w)
    plans = [ViewSplitPlan(shards, r, PAIR, cfg_split=False) for r in range(shards)]
    ex = _LocalExchange(plans)
    dens = [_denoiser(step_models, dtype, use_graph=False, view_shard=ViewShard(p, ex.bind(p))) for p in plans]
    errs = []

    def work(den):
        try:
            with torch.no_grad(), torch.cuda.stream(torch.cuda.Stream()):
                den.run(2)
                torch.cuda.current_stream().synchronize()
        except Exception as e:          # noqa: BLE001  (reported to the main thread)
            errs.append(e)
            ex.barrier.abort()

    torch.cuda.synchronize()                      # inputs were staged on the default stream
    ths = [threading.Thread(target=work, args=(d,)) for d in dens]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    assert not errs, errs
    got = torch.cat([d.latents.float().cpu() for d in dens], dim=1)
    assert got.shape == want.shape
    e = rel_l2(got, want)
    print("view split x%d: latents after 2 steps vs unsharded rel-L2 %.3e" % (shards, e))
    log_row("view split x%d latents after 2 steps vs unsharded" % shards, dtype, e, 0.0, 2e-3)
    assert e <= 2e-3
    # the halo matters: every shard's result depends on views it does not own
    assert all(p.remote for p in plans)


@pytest.mark.parametrize("attn_type", ["add", "concat"])
def test_view_split_block_variants_one_gpu(gpu, attn_type):
    """One multiview block with its 6 views spread over 2 shards (threads, one stream each): the 'add' and the
    'concat' neighbour attention fetch the remote views' K/V through the same slot buffer and reproduce the
    unsharded block; 'self' attends to every view and refuses to shard."""
    import threading
    from dualdiff_amd.networks.blocks import BasicMultiviewTransformerBlock
    from dualdiff_amd.parallel import ViewShard, ViewSplitPlan
    from oracle.init_utils import seeded_state_dict, seeded_tensor
    dtype, nb, l = torch.float16, 2, 350
    kw = dict(cross_attention_dim=768, neighboring_view_pair=PAIR, neighboring_attn_type=attn_type)
    ora = R.BasicMultiviewTransformerBlock(640, 8, 80, **kw)
    sd = seeded_state_dict(ora, 11)
    x = seeded_tensor((nb, 6, l, 640), 5).cuda().to(dtype)
    ctx = seeded_tensor((nb, 6, 30, 768), 6).cuda().to(dtype)

    def make():
        blk = BasicMultiviewTransformerBlock(640, 8, 80, **kw)
        blk.load_state_dict(sd)
        return blk.to("cuda", dtype)

    with torch.no_grad():
        want = make().run(x.reshape(-1, 640), nb * 6, l, ctx.reshape(-1, 768), 30).reshape(nb, 6, l, 640).float().cpu()
    plans = [ViewSplitPlan(2, r, PAIR, cfg_split=False) for r in range(2)]
    ex = _LocalExchange(plans)
    outs, errs = {}, []

    def work(p):
        try:
            with torch.no_grad(), torch.cuda.stream(torch.cuda.Stream()):
                blk = make()
                blk.view_shard = ViewShard(p, ex.bind(p))
                xs, cs = x[:, p.local].contiguous(), ctx[:, p.local].contiguous()
                n = nb * len(p.local)
                outs[p.shard] = blk.run(xs.reshape(-1, 640), n, l, cs.reshape(-1, 768), 30) \
                    .reshape(nb, len(p.local), l, 640).float().cpu()
        except Exception as e:          # noqa: BLE001
            errs.append(e)
            ex.barrier.abort()

    torch.cuda.synchronize()
    ths = [threading.Thread(target=work, args=(p,)) for p in plans]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    assert not errs, errs
    got = torch.cat([outs[p.shard] for p in plans], dim=1)
    e = rel_l2(got, want)
    print("view split block (%s): vs unsharded rel-L2 %.3e" % (attn_type, e))
    log_row("view split x2 block %s vs unsharded" % attn_type, dtype, e, 0.0, 1e-3)
    assert e <= 1e-3
    blk = make()
    blk.neighboring_attn_type = "self"
    blk.view_shard = ViewShard(plans[0], ex.bind(plans[0]))
    with pytest.raises(NotImplementedError):
        blk.run(x[:, plans[0].local].reshape(-1, 640), nb * 3, l, ctx[:, plans[0].local].reshape(-1, 768), 30)
