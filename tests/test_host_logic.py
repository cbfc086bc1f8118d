"""Host-side logic of the drop-in surface (no GPU): schedules, token/CFG helpers, checkpoints,
processor registry, neighbour maps — checked against the oracle / the reference's documented
behaviour."""
import json
import os

import pytest
import torch

from oracle import dualdiff_restated as R


def test_ddim_schedule_matches_oracle():
    from dualdiff_amd.pipeline.pipeline_bev_controlnet import ddim_schedule
    ts, coefs = ddim_schedule(50)
    ots, ratio = R.ddim_timesteps(50)
    assert torch.equal(ts, ots) and ts[0] == 981 and ts[-1] == 1
    acp = R.ddim_alphas()
    for i in (0, 1, 25, 49):
        ref = R.ddim_coefs(acp, int(ts[i]), ratio)
        assert torch.allclose(coefs[i], torch.tensor(ref), rtol=1e-5, atol=1e-6)
    # last step uses alphas_cumprod[0] (set_alpha_to_one = False in the SD-v1.5 scheduler config)
    assert abs(coefs[-1][2].item() - acp[0].sqrt().item()) < 1e-6


def small_unet():
    from dualdiff_amd.networks.unet_2d_condition_multiview import UNet2DConditionModelMultiview
    return UNet2DConditionModelMultiview(block_out_channels=(256, 256, 256, 256), cross_attention_dim=64,
                                         layers_per_block=1,
                                         neighboring_view_pair={"0": [5, 1], "1": [0, 2], "2": [1, 3],
                                                                "3": [2, 4], "4": [3, 5], "5": [4, 0]})


def test_unet_surface_and_checkpoint_roundtrip(tmp_path):
    from dualdiff_amd.networks.layers import seeded_init_
    from dualdiff_amd.networks.unet_2d_condition_multiview import UNet2DConditionModelMultiview
    net = seeded_init_(small_unet(), 3)
    assert net.config.in_channels == 4 and net.config["neighboring_attn_type"] == "add"
    blk = net.down_blocks[0].attentions[0].transformer_blocks[0]
    assert blk.neighboring_view_pair[0] == [5, 1]               # JSON string keys are int-cast (blocks.py:14-21)
    assert blk.n_cam == 6 and set(blk.new_module) == {"norm4", "attn4", "connector"}
    net.save_pretrained(tmp_path)
    cfg = json.load(open(os.path.join(tmp_path, "config.json")))
    assert cfg["_class_name"] == "UNet2DConditionModelMultiview" and cfg["_diffusers_version"] == "0.17.1"
    back = UNet2DConditionModelMultiview.from_pretrained(str(tmp_path), torch_dtype=torch.float16)
    assert back.dtype == torch.float16 and not back.training
    for k, v in net.state_dict().items():
        assert torch.equal(back.state_dict()[k].float(), v.half().float()), k
    # accepted-but-no-op knobs the reference flips
    back.enable_xformers_memory_efficient_attention()
    back.enable_gradient_checkpointing([True] * 8)
    assert back.trainable_parameters == []                       # the release trains no UNet params


def test_neighbour_maps_follow_view_pairs():
    net = small_unet()
    blk = net.mid_block.attentions[0].transformer_blocks[0]
    left, right = blk.neighbour_maps(12, torch.device("cpu"))
    assert left.tolist() == [5, 0, 1, 2, 3, 4, 11, 6, 7, 8, 9, 10]
    assert right.tolist() == [1, 2, 3, 4, 5, 0, 7, 8, 9, 10, 11, 6]


def test_attn_processor_registry():
    from dualdiff_amd.networks.box_adapter import XFormersAttnProcessor
    from dualdiff_amd.networks.layers import HIPAttnProcessor
    net = small_unet()
    procs = net.attn_processors
    assert len(procs) == 3 * sum(1 for _ in net.modules() if hasattr(_, "neighbour_maps"))
    assert all(k.endswith(".processor") for k in procs)
    net.set_attn_processor(XFormersAttnProcessor())
    assert all(isinstance(p, XFormersAttnProcessor) for p in net.attn_processors.values())
    with pytest.raises(ValueError):
        net.set_attn_processor({"x.processor": HIPAttnProcessor()})
    net.set_default_attn_processor()
    assert all(type(p) is HIPAttnProcessor for p in net.attn_processors.values())


def small_cnet():
    from dualdiff_amd.networks.unet_addon_rawbox import BEVControlNetModel
    return BEVControlNetModel(block_out_channels=(320, 256, 256, 256), cross_attention_dim=768, layers_per_block=1,
                              map_embedder_cls="magicdrive.networks.map_embedder.ControlNetConditioningEmbedding",
                              map_embedder_param={"block_out_channels": [16, 32, 96, 256]},
                              bbox_embedder_cls="magicdrive.networks.bbox_embedder.ContinuousBBoxWithTextEmbedding",
                              bbox_embedder_param={"n_classes": 10, "class_token_dim": 768, "embedder_num_freq": 4,
                                                   "proj_dims": [768, 512, 512, 768], "mode": "all-xyz",
                                                   "minmax_normalize": False, "use_text_encoder_init": False})


def test_controlnet_reference_config_names_and_cfg_helpers():
    """A config written for the reference (magicdrive.* dotted paths) builds our classes, and the CFG
    helpers behave like unet_addon_rawbox.py:327-335,671-769."""
    from dualdiff_amd.networks.bbox_embedder import ContinuousBBoxWithTextEmbedding
    from dualdiff_amd.networks.layers import seeded_init_
    from dualdiff_amd.networks.map_embedder import ControlNetConditioningEmbedding
    cn = seeded_init_(small_cnet(), 5)
    assert isinstance(cn.controlnet_cond_embedding, ControlNetConditioningEmbedding)
    assert isinstance(cn.bbox_embedder, ContinuousBBoxWithTextEmbedding)
    assert len(cn.controlnet_down_blocks) == 1 + 3 * 2 + 1
    assert cn.uncond_cam_param([2, 6]).shape == (2, 6, 3, 7)
    assert cn.uncond_cam_param(4).shape == (1, 4, 3, 7)
    cam = torch.randn(2, 6, 3, 7)
    boxes = {"bboxes": torch.randn(2, 6, 3, 8, 3), "classes": torch.ones(2, 6, 3, dtype=torch.long),
             "masks": torch.ones(2, 6, 3, dtype=torch.bool)}
    out = cn.add_uncond_to_kwargs(camera_param=cam, bboxes_3d_data=boxes, image=None, max_len=5, use_aug_text=False)
    assert out["camera_param"].shape == (4, 6, 3, 7)
    assert torch.equal(out["camera_param"][2:], cam)
    assert torch.equal(out["camera_param"][0, 0], cn.uncond_cam.weight.detach()[0].reshape(3, 7))
    b = out["bboxes_3d_data"]
    assert b["bboxes"].shape == (4, 6, 5, 8, 3) and b["masks"].shape == (4, 6, 5)
    assert not b["masks"][:2].any() and b["masks"][2:, :, :3].all() and not b["masks"][2:, :, 3:].any()
    assert out["use_aug_text"] is False and out["image"] is None
    lst = cn.add_uncond_to_kwargs(camera_param=cam, bboxes_3d_data=[boxes, None], image=None, max_len=4)
    assert lst["bboxes_3d_data"][0]["classes"].shape == (4, 6, 4) and lst["bboxes_3d_data"][1]["bboxes"].shape == (4, 6, 4, 8, 3)
    # attribute protocol of misc/test_utils.py:123-136 must be assignable
    cn.adm_proj = None
    cn.txt_con_fusionp = None
    cn.controlnet_cond_embedding = None
    assert "adm_proj.0.weight" not in cn.state_dict()
    with pytest.raises(KeyError):                                  # use_aug_text is a required kwarg (:812)
        cn.forward(torch.zeros(1, 6, 4, 28, 50), 1, cam[:1], None, torch.zeros(1, 7, 768), torch.zeros(6, 320, 28, 50))


def test_fourier_embedder_matches_oracle():
    from dualdiff_amd.networks.embedder import get_embedder
    x = torch.randn(5, 7, 3)
    e = get_embedder(3, 4)
    assert e.out_dim == 27
    assert torch.allclose(e(x), R.fourier_embed(x, 4))


def test_load_module_and_move_to():
    from dualdiff_amd.misc.common import load_module, move_to
    cls = load_module("dualdiff_amd.networks.txt_con_fusion.txt_con_XFormersAttn")
    assert cls.__name__ == "txt_con_XFormersAttn"
    d = move_to({"a": torch.zeros(2), "b": [None, True]}, "cpu")
    assert d["b"] == [None, True]
    with pytest.raises(TypeError):
        move_to(3, "cpu")


def test_models_refuse_cpu_inputs():
    net = small_unet()
    with pytest.raises((RuntimeError, NotImplementedError)):
        net(torch.zeros(6, 4, 8, 8), 1, torch.zeros(6, 3, 64))
