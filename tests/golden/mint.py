"""Mints the golden vectors under tests/golden/ by executing the REFERENCE'S OWN SOURCE FILES.

Runs only in the build container (needs /root/reference); nothing here is used at test time.
The reference imports `diffusers` and `xformers`, which are not installed.  This script
registers stand-in modules *in memory* (nothing is written to disk, no reference code is
copied):

  * name-only stubs:  diffusers.utils.import_utils.is_xformers_available -> True,
    diffusers.utils.BaseOutput, diffusers.models.controlnet.zero_module,
    xformers.ops.memory_efficient_attention -> torch softmax(q k^T * scale) v on (B, L, H, D);
  * the diffusers-0.17.1 base classes the reference SUBCLASSES are bound to our CPU
    restatement (oracle/diffusers_restated.py).

Consequences for pinning (DESIGN.md "Oracle"):
  LEVEL 1 (true reference arithmetic; only the SDPA stand-in is ours):
      sfa, sfa_plus, cond_embedder, bev_map_embedder, bbox_embedder, attn_processor, adapter_processor,
      ors_projection (integer labels, bit-exact)
  LEVEL 2 (reference control flow executed verbatim over our restated leaf modules):
      multiview_block, unet_multiview, controlnet_bg, controlnet_fg, controlnet_bg_adapter

Usage:  python tests/golden/mint.py      (rewrites tests/golden/*.npz)
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference/MD_txt_con_fusion"
sys.path.insert(0, ROOT)

from oracle import diffusers_restated as D          # noqa: E402
from oracle.init_utils import seeded_state_dict, seeded_tensor   # noqa: E402
from tests.golden import cases as C                  # noqa: E402


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def install_stubs():
    def mea(q, k, v, attn_bias=None, op=None, scale=None):
        assert attn_bias is None
        s = (q.permute(0, 2, 1, 3).float() @ k.permute(0, 2, 3, 1).float()) * scale
        o = torch.softmax(s, dim=-1) @ v.permute(0, 2, 1, 3).float()
        return o.permute(0, 2, 1, 3).to(q.dtype)

    _mod("xformers", ops=_mod("xformers.ops", memory_efficient_attention=mea))

    class BaseOutput:
        pass

    class XFormersAttnProcessor:      # name imported (and shadowed) by box_adapter.py:13,17
        pass

    diffusers = _mod("diffusers", UNet2DConditionModel=D.UNet2DConditionModel)
    diffusers.__path__ = []
    _mod("diffusers.utils", BaseOutput=BaseOutput).__path__ = []
    _mod("diffusers.utils.import_utils", is_xformers_available=lambda: True)
    _mod("diffusers.configuration_utils", register_to_config=D.register_to_config, ConfigMixin=D.ConfigMixin)
    _mod("diffusers.models").__path__ = []
    _mod("diffusers.models.attention", Attention=D.Attention, BasicTransformerBlock=D.BasicTransformerBlock,
         AdaLayerNorm=D.AdaLayerNorm)
    _mod("diffusers.models.attention_processor", Attention=D.Attention, AttentionProcessor=object,
         AttnProcessor=D.AttnProcessor, XFormersAttnProcessor=XFormersAttnProcessor)
    _mod("diffusers.models.controlnet", zero_module=D.zero_module)
    _mod("diffusers.models.embeddings", TimestepEmbedding=D.TimestepEmbedding, Timesteps=D.Timesteps)
    _mod("diffusers.models.modeling_utils", ModelMixin=D.ModelMixin)
    _mod("diffusers.models.unet_2d_blocks", CrossAttnDownBlock2D=D.CrossAttnDownBlock2D,
         CrossAttnUpBlock2D=D.CrossAttnUpBlock2D, DownBlock2D=D.DownBlock2D, UpBlock2D=D.UpBlock2D,
         UNetMidBlock2DCrossAttn=D.UNetMidBlock2DCrossAttn, get_down_block=D.get_down_block)
    _mod("diffusers.models.unet_2d_condition", UNet2DConditionModel=D.UNet2DConditionModel,
         UNet2DConditionOutput=D.UNet2DConditionOutput)
    sys.path.insert(0, REF)


def save(name, **arrays):
    """Fixtures stay small: tensors above 16k elements are stored as a strided subsample plus
    float64 checksums (sum, sum|x|) — see tests/golden/cases.py:compact()."""
    path = os.path.join(HERE, name + ".npz")
    out = {}
    for k, v in arrays.items():
        out.update(C.compact(k, v.detach().float()))
    np.savez_compressed(path, **out)
    print("wrote %s (%.1f KB)" % (path, os.path.getsize(path) / 1024))


def load_from(ref_module, oracle_module, seed):
    """Seeded weights are defined on the ORACLE module's parameter names; loading them strictly
    into the reference module also checks the state-dict name contract."""
    sd = seeded_state_dict(oracle_module, seed)
    missing, unexpected = ref_module.load_state_dict(sd, strict=False)
    assert not missing and not unexpected, (missing, unexpected)


@torch.no_grad()
def mint_block_variants():
    """`python tests/golden/mint.py variants`: the reference's BasicMultiviewTransformerBlock with the other
    neighboring_attn_type / zero_module_type settings it defines (level 2: its control flow over restated leaves)."""
    install_stubs()
    from oracle import dualdiff_restated as R
    from magicdrive.networks import blocks as ref_blocks
    kw = C.block_kwargs()
    hs, ctx = C.block_inputs()
    out = {}
    for attn_type, zero_type in C.BLOCK_VARIANTS:
        ref = ref_blocks.BasicMultiviewTransformerBlock(**kw, neighboring_view_pair=C.VIEW_PAIR,
                                                        neighboring_attn_type=attn_type, zero_module_type=zero_type)
        load_from(ref, R.BasicMultiviewTransformerBlock(**kw, neighboring_view_pair=C.VIEW_PAIR,
                                                        neighboring_attn_type=attn_type, zero_module_type=zero_type),
                  C.SEED_BLOCK_VAR)
        out["%s_%s" % (attn_type, zero_type)] = ref(hs, encoder_hidden_states=ctx)
    save("multiview_block_variants", **out)


@torch.no_grad()
def main():
    install_stubs()
    from oracle import dualdiff_restated as R
    from magicdrive.networks import txt_con_fusion as ref_sfa
    from magicdrive.networks import map_embedder as ref_map
    from magicdrive.networks import bbox_embedder as ref_bbox
    from magicdrive.networks import box_adapter as ref_proc
    from magicdrive.networks import blocks as ref_blocks
    from magicdrive.networks import unet_2d_condition_multiview as ref_unet
    from magicdrive.networks import unet_addon_rawbox as ref_cnet

    # ---- L1: SFA / SFA+ --------------------------------------------------------------------
    x, e = C.sfa_inputs()
    for name, ref_cls, ora_cls in (("sfa", ref_sfa.txt_con_XFormersAttn, R.TxtConFusion),
                                   ("sfa_plus", ref_sfa.txt_con_XFormersAttn_plus, R.TxtConFusionPlus)):
        ref = ref_cls()
        load_from(ref, ora_cls(), C.SEED_SFA)
        save(name, out=ref(attn=None, hidden_states=x, encoder_hidden_states=e))

    # ---- L1: condition embedder ------------------------------------------------------------
    ref = ref_map.ControlNetConditioningEmbedding(320, block_out_channels=(16, 32, 96, 256))
    load_from(ref, R.ControlNetConditioningEmbedding(320), C.SEED_EMB)
    save("cond_embedder", out=ref(C.cond_image()))

    # ---- L1: bbox embedder -----------------------------------------------------------------
    ref = ref_bbox.ContinuousBBoxWithTextEmbedding(n_classes=10, mode="all-xyz", minmax_normalize=False,
                                                   use_text_encoder_init=False)
    load_from(ref, R.BBoxEmbedder(), C.SEED_BOX)
    bb, cl, mk = C.box_inputs()
    save("bbox_embedder", out=ref(bb, cl, mk))

    # ---- L1: attention-processor protocol --------------------------------------------------
    attn = D.Attention(query_dim=320, cross_attention_dim=768, heads=8, dim_head=40)
    attn.load_state_dict(seeded_state_dict(attn, C.SEED_PROC))
    h, ctx = C.proc_inputs()
    save("attn_processor", out=ref_proc.XFormersAttnProcessor()._real_call(attn, h, ctx),
         out_self=ref_proc.XFormersAttnProcessor()._real_call(_self_attn(), h))

    # ---- L1: box / class adapter processor (N1, box_adapter.py:177-411) --------------------
    attn = D.Attention(query_dim=320, cross_attention_dim=768, heads=8, dim_head=40)
    attn.load_state_dict(seeded_state_dict(attn, C.SEED_PROC))
    ref = ref_proc.Adapter_XFormersAttnProcessor(hidden_size=320, cross_attention_dim=768, scale=0.7)
    load_from(ref, R.AdapterAttnProcessor(320, 768), C.SEED_ADAPTER)
    h, ctx, nt = C.adapter_inputs()
    ref.num_tokens = nt
    save("adapter_processor", out=ref._real_call(attn, h, ctx))

    # ---- L2: multiview transformer block ---------------------------------------------------
    kw = C.block_kwargs()
    ref = ref_blocks.BasicMultiviewTransformerBlock(**kw, neighboring_view_pair=C.VIEW_PAIR)
    load_from(ref, R.BasicMultiviewTransformerBlock(**kw, neighboring_view_pair=C.VIEW_PAIR), C.SEED_BLOCK)
    hs, ctx = C.block_inputs()
    save("multiview_block", out=ref(hs, encoder_hidden_states=ctx))

    # ---- L2: multiview UNet (reduced widths, true 28x50 latents -> explicit upsample sizes) --
    ukw = C.unet_kwargs()
    ref = ref_unet.UNet2DConditionModelMultiview(**ukw, neighboring_view_pair=C.VIEW_PAIR).eval()
    load_from(ref, R.UNet2DConditionModelMultiview(**ukw, neighboring_view_pair=C.VIEW_PAIR), C.SEED_UNET)
    sample, t, ctx, down, mid = C.unet_inputs()
    save("unet_multiview",
         out=ref(sample, t, encoder_hidden_states=ctx, down_block_additional_residuals=down,
                 mid_block_additional_residual=mid).sample,
         out_nores=ref(sample, t, encoder_hidden_states=ctx).sample)

    # ---- L2: ControlNet branches (bg: panorama embedder; fg: ORS-3D input), SFA on -----------
    for name, occ3d in (("controlnet_bg", False), ("controlnet_fg", True)):
        ckw = C.controlnet_kwargs()
        ref = ref_cnet.BEVControlNetModel(
            **ckw["diffusers"],
            map_embedder_cls="magicdrive.networks.map_embedder.ControlNetConditioningEmbedding",
            map_embedder_param={"block_out_channels": ckw["cond_channels"]},
            cam_embedder_param={"input_dims": 3, "num_freqs": 4, "include_input": True, "log_sampling": True},
            bbox_embedder_cls="magicdrive.networks.bbox_embedder.ContinuousBBoxWithTextEmbedding",
            bbox_embedder_param={"n_classes": 10, "class_token_dim": 768, "trainable_class_token": False,
                                 "use_text_encoder_init": False, "embedder_num_freq": 4,
                                 "proj_dims": [768, 512, 512, ckw["diffusers"]["cross_attention_dim"]],
                                 "mode": "all-xyz", "minmax_normalize": False},
        ).eval()
        # flags poked after construction exactly as misc/test_utils.py:123-136 does
        ref.use_cam_in_temb = False
        ref.use_box_adapter = False
        ref.adm_proj = None
        ref.use_txt_con_fusion = True
        ref.use_txt_con_fusionp = False
        ref.txt_con_fusionp = None
        ref.use_occ_3d = occ3d
        ora = R.BEVControlNetModel(**C.controlnet_oracle_kwargs(), use_occ_3d=occ3d)
        sd = seeded_state_dict(ora, C.SEED_CNET + int(occ3d))
        missing, unexpected = ref.load_state_dict(sd, strict=False)
        unused = ("txt_con_fusionp", "adm_proj") + (("controlnet_cond_embedding",) if occ3d else ())
        assert not unexpected and all(m.startswith(unused) for m in missing), (missing, unexpected)
        if occ3d:
            ref.controlnet_cond_embedding = None
        inp = C.controlnet_inputs(occ3d)
        down, mid, ctx = ref(inp["sample"], inp["timestep"], inp["camera_param"], inp["bboxes_3d_data"],
                             inp["encoder_hidden_states"], inp["controlnet_cond"],
                             conditioning_scale=inp["conditioning_scale"], return_dict=False,
                             use_aug_text=False)
        arrays = {"down_%d" % i: d for i, d in enumerate(down)}
        save(name, mid=mid, ctx=ctx, **arrays)
        if not occ3d:
            # same branch with the box / class adapter installed by the reference's own installer
            # (box_adapter.py:414-444) and `use_box_adapter` poked like multiview_runner.py:191,241-242
            ref.use_box_adapter = True
            ref_proc.box_adapter(ref)
            down, mid, ctx = ref(inp["sample"], inp["timestep"], inp["camera_param"], inp["bboxes_3d_data"],
                                 inp["encoder_hidden_states"], inp["controlnet_cond"],
                                 conditioning_scale=inp["conditioning_scale"], return_dict=False,
                                 use_aug_text=False)
            arrays = {"down_%d" % i: d for i, d in enumerate(down)}
            save(name + "_adapter", mid=mid, ctx=ctx, **arrays)


def mint_ors():
    """LEVEL 1: the reference's own `OccupancyRay.project` (networks/occ3d_proj.py:49-113) on seeded
    stand-ins for its data files.  `cv2` is a name-only stub; the stub `Quaternion` returns the rotation
    MATRIX stored in the fake camera table, so no quaternion arithmetic of ours enters."""
    import tempfile
    _mod("cv2")

    class Quaternion:
        def __init__(self, rot):
            self.rotation_matrix = np.asarray(rot, dtype=np.float64)
    _mod("pyquaternion", Quaternion=Quaternion)
    from magicdrive.networks import occ3d_proj as ref_ors
    occ, Ks, Rts = C.ors_inputs()
    tok = "seeded-sample"
    with tempfile.TemporaryDirectory() as root:
        os.makedirs(os.path.join(root, "scene"))
        np.savez(os.path.join(root, "scene", "labels.npz"), semantics=occ.numpy().astype(np.uint8))
        proj = object.__new__(ref_ors.OccupancyRay)               # skip __init__: it unpickles data files
        proj.camera_data = {tok: {v: {"translation": Rts[i][:3, 3].tolist(), "rotation": Rts[i][:3, :3].numpy(),
                                      "intrinsic": Ks[i].tolist()} for i, v in enumerate(C.ORS_VIEWS)}}
        proj.occ3d_idx = {tok: "scene"}
        proj.device = "cpu"
        proj.dataroot = root
        proj.image_shape = (896, 1600)
        proj.sample_point = C.ORS_S
        proj.sample_step = 0.2
        proj.compress_ratio = C.ORS_RATIO
        proj.image_shape_compress = [int(896 * C.ORS_RATIO), int(1600 * C.ORS_RATIO)]
        labels = proj.project(tok)                                 # (6, 28, 50, 320) int64
    assert tuple(labels.shape) == (6, C.ORS_H, C.ORS_W, C.ORS_S), labels.shape
    path = os.path.join(HERE, "ors_projection.npz")
    np.savez_compressed(path, labels=labels.numpy().astype(np.uint8))
    print("wrote %s (%.1f KB); classes seen: %s" % (path, os.path.getsize(path) / 1024,
                                                      sorted(set(labels.unique().tolist()))))


@torch.no_grad()
def mint_bev_embedder():
    """`python tests/golden/mint.py bev`: the reference's BEVControlNetConditioningEmbedding (level 1: its own arithmetic)."""
    install_stubs()
    from oracle import dualdiff_restated as R
    from magicdrive.networks import map_embedder as ref_map
    ref = ref_map.BEVControlNetConditioningEmbedding()
    load_from(ref, R.BEVControlNetConditioningEmbedding(), C.SEED_BEV_EMB)
    save("bev_map_embedder", out=ref(C.bev_map()))


def _self_attn():
    a = D.Attention(query_dim=320, heads=8, dim_head=40)
    a.load_state_dict(seeded_state_dict(a, C.SEED_PROC + 1))
    return a


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "ors":
        install_stubs()
        mint_ors()
    elif len(sys.argv) > 1 and sys.argv[1] == "variants":
        mint_block_variants()
    elif len(sys.argv) > 1 and sys.argv[1] == "bev":
        mint_bev_embedder()
    else:
        main()
        mint_ors()
        mint_block_variants()
        mint_bev_embedder()
