"""Oracle vs golden vectors minted from the reference's own source (tests/golden/mint.py).

LEVEL 1 fixtures are true reference arithmetic (SFA, SFA+, condition embedder, bbox embedder,
attention-processor protocol).  LEVEL 2 fixtures are the reference's subclasses / forward()s
executed verbatim over the restated diffusers base classes: they pin the reference-owned control
flow (view pairing + per-pair out-projection sum in attn4, token concatenation order, residual
add order, explicit upsample sizes, ControlNet flattening / scaling) — not the leaf arithmetic.
fp32 on CPU; tolerance covers summation-order differences only.
"""
import os

import numpy as np
import pytest
import torch

from oracle import diffusers_restated as D
from oracle import dualdiff_restated as R
from oracle.init_utils import seeded_init_, seeded_state_dict
from tests.golden import cases as C

GOLD = os.path.join(os.path.dirname(__file__), "golden")
RTOL, ATOL = 2e-4, 2e-5


def gold(name):
    return np.load(os.path.join(GOLD, name + ".npz"))


@torch.no_grad()
@pytest.mark.parametrize("name,cls", [("sfa", R.TxtConFusion), ("sfa_plus", R.TxtConFusionPlus)])
def test_sfa(name, cls):
    m = seeded_init_(cls(), C.SEED_SFA)
    x, e = C.sfa_inputs()
    C.compare(gold(name), "out", m(x, e), RTOL, ATOL)


@torch.no_grad()
def test_cond_embedder():
    m = seeded_init_(R.ControlNetConditioningEmbedding(320), C.SEED_EMB)
    C.compare(gold("cond_embedder"), "out", m(C.cond_image()), RTOL, ATOL)


@torch.no_grad()
def test_bev_map_embedder():
    """map_embedder.py:10-77 (asymmetric paddings, (2, 1) stride): oracle vs the reference's own output."""
    m = seeded_init_(R.BEVControlNetConditioningEmbedding(), C.SEED_BEV_EMB)
    out = m(C.bev_map())
    assert tuple(out.shape) == (6, 320, 28, 50)
    C.compare(gold("bev_map_embedder"), "out", out, RTOL, ATOL)


@torch.no_grad()
def test_bbox_embedder():
    m = seeded_init_(R.BBoxEmbedder(), C.SEED_BOX)
    bb, cl, mk = C.box_inputs()
    C.compare(gold("bbox_embedder"), "out", m(bb, cl, mk), RTOL, ATOL)


@torch.no_grad()
def test_attn_processor_protocol():
    g = gold("attn_processor")
    a = seeded_init_(D.Attention(query_dim=320, cross_attention_dim=768, heads=8, dim_head=40), C.SEED_PROC)
    h, ctx = C.proc_inputs()
    C.compare(g, "out", a(h, encoder_hidden_states=ctx), RTOL, ATOL)
    a = seeded_init_(D.Attention(query_dim=320, heads=8, dim_head=40), C.SEED_PROC + 1)
    C.compare(g, "out_self", a(h), RTOL, ATOL)


@torch.no_grad()
def test_adapter_processor():
    """N1: box / class adapter processor against the reference's own `_real_call` (level 1)."""
    a = seeded_init_(D.Attention(query_dim=320, cross_attention_dim=768, heads=8, dim_head=40), C.SEED_PROC)
    proc = seeded_init_(R.AdapterAttnProcessor(320, 768, scale=0.7), C.SEED_ADAPTER)
    h, ctx, nt = C.adapter_inputs()
    proc.num_tokens = nt
    C.compare(gold("adapter_processor"), "out", proc(a, h, ctx), RTOL, ATOL)


@torch.no_grad()
def test_multiview_block():
    m = seeded_init_(R.BasicMultiviewTransformerBlock(**C.block_kwargs(), neighboring_view_pair=C.VIEW_PAIR),
                     C.SEED_BLOCK)
    hs, ctx = C.block_inputs()
    C.compare(gold("multiview_block"), "out", m(hs, encoder_hidden_states=ctx), RTOL, ATOL)


@torch.no_grad()
@pytest.mark.parametrize("attn_type,zero_type", C.BLOCK_VARIANTS)
def test_multiview_block_variants(attn_type, zero_type):
    """The block's other neighboring_attn_type / zero_module_type settings (blocks.py:81-90,106-142) against the
    reference's own block executed on the same seeded weights."""
    m = seeded_init_(R.BasicMultiviewTransformerBlock(**C.block_kwargs(), neighboring_view_pair=C.VIEW_PAIR,
                                                      neighboring_attn_type=attn_type, zero_module_type=zero_type),
                     C.SEED_BLOCK_VAR)
    hs, ctx = C.block_inputs()
    C.compare(gold("multiview_block_variants"), "%s_%s" % (attn_type, zero_type),
              m(hs, encoder_hidden_states=ctx), RTOL, ATOL)


@torch.no_grad()
def test_unet_multiview():
    m = seeded_init_(R.UNet2DConditionModelMultiview(**C.unet_kwargs(), neighboring_view_pair=C.VIEW_PAIR),
                     C.SEED_UNET).eval()
    sample, t, ctx, down, mid = C.unet_inputs()
    g = gold("unet_multiview")
    out = m(sample, t, encoder_hidden_states=ctx, down_block_additional_residuals=down,
            mid_block_additional_residual=mid).sample
    C.compare(g, "out", out, 1e-3, 1e-4)
    C.compare(g, "out_nores", m(sample, t, encoder_hidden_states=ctx).sample, 1e-3, 1e-4)


@torch.no_grad()
@pytest.mark.parametrize("name,occ3d", [("controlnet_bg", False), ("controlnet_fg", True)])
def test_controlnet(name, occ3d):
    m = seeded_init_(R.BEVControlNetModel(**C.controlnet_oracle_kwargs(), use_occ_3d=occ3d),
                     C.SEED_CNET + int(occ3d)).eval()
    inp = C.controlnet_inputs(occ3d)
    down, mid, ctx = m(inp["sample"], inp["timestep"], inp["camera_param"], inp["bboxes_3d_data"],
                       inp["encoder_hidden_states"], inp["controlnet_cond"],
                       conditioning_scale=inp["conditioning_scale"])
    g = gold(name)
    assert len(down) == 12
    for i, d in enumerate(down):
        C.compare(g, "down_%d" % i, d, 1e-3, 1e-4)
    C.compare(g, "mid", mid, 1e-3, 1e-4)
    C.compare(g, "ctx", ctx, RTOL, ATOL)


@torch.no_grad()
def test_controlnet_box_adapter():
    """N1, level 2: ControlNet branch with `use_box_adapter` and the reference's own installer
    (box_adapter.py:414-444): class tokens appended for the ControlNet's cross-attentions only, the
    context handed to the UNet stays [cam | text | box]."""
    m = seeded_init_(R.BEVControlNetModel(**C.controlnet_oracle_kwargs(), use_occ_3d=False), C.SEED_CNET).eval()
    m.use_box_adapter = True
    R.box_adapter(m)
    inp = C.controlnet_inputs(False)
    down, mid, ctx = m(inp["sample"], inp["timestep"], inp["camera_param"], inp["bboxes_3d_data"],
                       inp["encoder_hidden_states"], inp["controlnet_cond"],
                       conditioning_scale=inp["conditioning_scale"])
    g = gold("controlnet_bg_adapter")
    for i, d in enumerate(down):
        C.compare(g, "down_%d" % i, d, 1e-3, 1e-4)
    C.compare(g, "mid", mid, 1e-3, 1e-4)
    C.compare(g, "ctx", ctx, RTOL, ATOL)
    # the adapter must actually change the result (guards against a silently inactive installer)
    g0 = gold("controlnet_bg")
    assert not np.allclose(g["mid" if "mid" in g.files else g.files[0]], g0["mid" if "mid" in g0.files else g0.files[0]])


def test_state_dict_contract_sd15():
    """Structural anchor for the un-pinned diffusers restatement: parameter names, shapes and
    counts of the full-size models equal the SD-v1.5 / DualDiff layout."""
    unet = R.UNet2DConditionModelMultiview(cross_attention_dim=768, neighboring_view_pair=C.VIEW_PAIR)
    sd = unet.state_dict()
    n_unet = sum(v.numel() for k, v in sd.items())
    base = sum(v.numel() for k, v in sd.items() if not any(s in k for s in ("norm4", "attn4", "connector")))
    assert base == 859_520_964, base          # SD-v1.5 UNet parameter count
    assert sd["down_blocks.0.attentions.0.transformer_blocks.0.attn2.to_k.weight"].shape == (320, 768)
    assert sd["up_blocks.1.resnets.2.conv1.weight"].shape == (1280, 1920, 3, 3)
    assert sd["up_blocks.3.resnets.0.conv_shortcut.weight"].shape == (320, 960, 1, 1)
    assert sd["mid_block.attentions.0.transformer_blocks.0.ff.net.0.proj.weight"].shape == (10240, 1280)
    assert sd["down_blocks.2.attentions.1.transformer_blocks.0.attn4.to_out.0.bias"].shape == (1280,)
    assert sd["up_blocks.2.attentions.0.transformer_blocks.0.connector.weight"].shape == (640, 640)
    assert len([k for k in sd if k.endswith("attn4.to_q.weight")]) == 16
    assert n_unet > base
    cn = R.BEVControlNetModel()
    csd = cn.state_dict()
    assert len([k for k in csd if k.startswith("controlnet_down_blocks") and k.endswith("weight")]) == 12
    assert csd["controlnet_down_blocks.11.weight"].shape == (1280, 1280, 1, 1)
    assert csd["cam2token.weight"].shape == (768, 189)
    assert csd["uncond_cam.weight"].shape == (1, 21)
    assert csd["bbox_embedder.bbox_proj.weight"].shape == (768, 216)
    assert csd["txt_con_fusion.to_k.weight"].shape == (320, 768)
    assert csd["controlnet_cond_embedding.blocks.5.weight"].shape == (256, 96, 3, 3)


@torch.no_grad()
def test_ors_projection_bit_exact():
    """N3: the ORS ray-sampling restatement against the reference's own OccupancyRay.project on the
    seeded volume and cameras — integer labels, every one of the 6 x 28 x 50 x 320 samples equal."""
    from oracle import ors_projection as P
    occ, Ks, Rts = C.ors_inputs()
    lab = P.ors_project(occ, Ks, Rts, C.ORS_H, C.ORS_W, C.ORS_RATIO, C.ORS_S, 0.2)
    ref = torch.from_numpy(gold("ors_projection")["labels"].astype(np.int64))
    assert lab.shape == ref.shape == (6, C.ORS_H, C.ORS_W, C.ORS_S)
    assert torch.equal(lab, ref), "%d of %d labels differ" % ((lab != ref).sum().item(), ref.numel())
    # the volume is actually hit: a fair share of samples is not "free / outside"
    assert 0.02 < (ref != 17).float().mean().item() < 0.9
    cond = P.ors_condition(lab, use_fg=True, use_bg=False)
    assert cond.shape == (6, C.ORS_S, C.ORS_H, C.ORS_W) and cond.max().item() <= 1.0
    assert not ((cond * 17).round() >= 11).logical_and((cond * 17).round() < 17).any()     # backgrounds filtered
