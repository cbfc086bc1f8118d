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


def test_box_adapter_use_box_token_installer():
    """box_adapter.py:441-442: with use_box_token the adapter projections take the 128-wide boxworld feature tokens (no
    copy of to_k / to_v); the token-LIST context itself raises NotImplementedError in the reference's processor
    (:269-270) and here; attn1 / attn4 keep the plain processor."""
    from dualdiff_amd.networks.box_adapter import Adapter_XFormersAttnProcessor, XFormersAttnProcessor, box_adapter
    net = small_unet()
    box_adapter(net, use_box_token=True)
    procs = net.attn_processors
    ad = {k: p for k, p in procs.items() if isinstance(p, Adapter_XFormersAttnProcessor)}
    assert ad and all(k.endswith("attn2.processor") for k in ad)
    assert all(type(p) is XFormersAttnProcessor for k, p in procs.items() if k not in ad)
    for p in ad.values():
        assert p.to_k_box.weight.shape[1] == 128 and p.to_v_cls.weight.shape[1] == 128
        assert torch.isfinite(p.to_k_box.weight).all() and p.to_k_box.weight.abs().max() > 0
    name, p = next(iter(ad.items()))
    attn = dict(net.named_modules())[name[: -len(".processor")]]
    with pytest.raises(NotImplementedError):
        p(attn, torch.zeros(1, 4, attn.to_q.in_features), [torch.zeros(1, 3, 768), torch.zeros(1, 2, 128, 1)])


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
    from dualdiff_amd.misc.common import load_module
    from dualdiff_amd.networks.map_embedder import BEVControlNetConditioningEmbedding
    from dualdiff_amd.networks.unet_addon_rawbox import _own_path
    bev = load_module(_own_path("magicdrive.networks.map_embedder.BEVControlNetConditioningEmbedding"))()   # vanilla MagicDrive's
    assert isinstance(bev, BEVControlNetConditioningEmbedding) and len(bev.blocks) == 6 and bev._geom[-1] == (1, 2)
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


# the diffusers-0.17.1 config key set of an SD-v1.5 `unet/config.json`, plus the keys the MagicDrive / DualDiff release
# adds (unet_2d_condition_multiview.py:173-179); widths shrunk so the CPU suite stays fast
FOREIGN_UNET_CONFIG = {
    "_class_name": "UNet2DConditionModelMultiview", "_diffusers_version": "0.17.1", "_name_or_path": "pretrained/sd-v1-5/unet",
    "act_fn": "silu", "addition_embed_type": None, "addition_embed_type_num_heads": 64, "attention_head_dim": 8,
    "block_out_channels": [64, 64, 128, 128], "center_input_sample": False, "class_embed_type": None,
    "class_embeddings_concat": False, "conv_in_kernel": 3, "conv_out_kernel": 3, "cross_attention_dim": 64,
    "cross_attention_norm": None,
    "down_block_types": ["CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "DownBlock2D"],
    "downsample_padding": 1, "dual_cross_attention": False, "encoder_hid_dim": None, "flip_sin_to_cos": True,
    "freq_shift": 0, "in_channels": 4, "layers_per_block": 1, "mid_block_only_cross_attention": None,
    "mid_block_scale_factor": 1, "mid_block_type": "UNetMidBlock2DCrossAttn", "norm_eps": 1e-05, "norm_num_groups": 32,
    "num_class_embeds": None, "only_cross_attention": False, "out_channels": 4, "projection_class_embeddings_input_dim": None,
    "resnet_out_scale_factor": 1.0, "resnet_skip_time_act": False, "resnet_time_scale_shift": "default", "sample_size": 64,
    "time_cond_proj_dim": None, "time_embedding_act_fn": None, "time_embedding_type": "positional",
    "timestep_post_act": None, "upcast_attention": False,
    "up_block_types": ["UpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D"],
    "use_linear_projection": False,
    "trainable_state": "only_new", "neighboring_attn_type": "add", "zero_module_type": "zero_linear",
    "crossview_attn_type": "basic", "img_size": [224, 400],
    "neighboring_view_pair": {"0": [5, 1], "1": [0, 2], "2": [1, 3], "3": [2, 4], "4": [3, 5], "5": [4, 0]},
}


def write_foreign_unet_checkpoint(folder, fmt, seed=21, **override):
    """A checkpoint folder NOT produced by this package's save_pretrained: config.json as diffusers 0.17.1 writes it and
    the weights of the oracle's independent restatement of the network (its own module tree and parameter names, which
    follow diffusers'), as a torch pickle (`.bin`, what the MagicDrive release ships) or safetensors."""
    from oracle import dualdiff_restated as R
    from oracle.init_utils import seeded_state_dict
    cfg = dict(FOREIGN_UNET_CONFIG, **override)
    ora = R.UNet2DConditionModelMultiview(
        block_out_channels=tuple(cfg["block_out_channels"]), cross_attention_dim=cfg["cross_attention_dim"],
        layers_per_block=cfg["layers_per_block"], attention_head_dim=cfg["attention_head_dim"],
        neighboring_view_pair=cfg["neighboring_view_pair"]).eval()
    sd = seeded_state_dict(ora, seed)
    ora.load_state_dict(sd)
    os.makedirs(folder, exist_ok=True)
    with open(os.path.join(folder, "config.json"), "w") as f:
        json.dump(cfg, f, indent=2)
    if fmt == "bin":
        torch.save({k: v.clone() for k, v in sd.items()}, os.path.join(folder, "diffusion_pytorch_model.bin"))
    else:
        from safetensors.torch import save_file
        save_file({k: v.contiguous() for k, v in sd.items()}, os.path.join(folder, "diffusion_pytorch_model.safetensors"))
    return ora, sd


@pytest.mark.parametrize("fmt", ["bin", "safetensors"])
def test_unet_imports_a_foreign_diffusers_checkpoint(tmp_path, fmt):
    """SURVEY §8f N4 (checkpoint import): a diffusers-layout folder written by OTHER code loads with every key matched —
    nothing missing, nothing unexpected, values intact — through `from_pretrained(dir, subfolder=..., torch_dtype=...)`
    exactly as the reference calls it (misc/test_utils.py:111-113, runner/multiview_runner.py:118-132)."""
    from dualdiff_amd.networks.unet_2d_condition_multiview import UNet2DConditionModelMultiview
    ora, sd = write_foreign_unet_checkpoint(os.path.join(tmp_path, "unet"), fmt)
    net = UNet2DConditionModelMultiview.from_pretrained(str(tmp_path), subfolder="unet", torch_dtype=torch.float16,
                                                        low_cpu_mem_usage=False, device_map=None)
    own = net.state_dict()
    assert set(own) == set(sd), (sorted(set(own) ^ set(sd))[:8])
    for k, v in sd.items():
        assert own[k].dtype == torch.float16 and torch.equal(own[k].float(), v.half().float()), k
    assert net.config["img_size"] == [224, 400] and net.config["sample_size"] == 64
    assert net.config["time_embedding_type"] == "positional"          # unknown-to-us diffusers keys are kept in .config
    # a key the checkpoint lacks is an error, not a silent random init
    broken = {k: v for k, v in sd.items() if not k.startswith("mid_block.attentions.0.transformer_blocks.0.attn4")}
    torch.save(broken, os.path.join(tmp_path, "unet", "diffusion_pytorch_model.bin"))
    st = os.path.join(tmp_path, "unet", "diffusion_pytorch_model.safetensors")
    if os.path.exists(st):
        os.remove(st)
    with pytest.raises(RuntimeError, match="missing keys"):
        UNet2DConditionModelMultiview.from_pretrained(str(tmp_path), subfolder="unet")


def test_fold_lora_key_validation_and_alpha():
    """fold_lora_ must reject malformed / incomplete LoRA dicts with a clear error and honour
    network_alpha (delta = scale * alpha / rank * up @ down)."""
    from dualdiff_amd.lora import fold_lora_, lora_keys
    from dualdiff_amd.networks.layers import Attention, seeded_init_

    class Holder(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.attn1 = Attention(64, None, heads=2, dim_head=32)

    net = seeded_init_(Holder(), 5)
    w0 = net.attn1.to_q.weight.detach().clone()
    g = torch.Generator().manual_seed(1)
    lora = {k: torch.randn(shape, generator=g) * 0.1 for k, shape in lora_keys(net, rank=4).items()}
    assert len(lora) == 8
    down, up = lora["attn1.processor.to_q_lora.down.weight"], lora["attn1.processor.to_q_lora.up.weight"]
    assert fold_lora_(net, lora, 0.5) == 4
    assert torch.allclose(net.attn1.to_q.weight, w0 + 0.5 * up @ down, atol=1e-6)
    # network_alpha: alpha / rank scales the delta
    net2 = seeded_init_(Holder(), 5)
    lora_a = dict(lora)
    lora_a["attn1.processor.to_q_lora.alpha"] = torch.tensor(8.0)
    fold_lora_(net2, lora_a, 0.5)
    assert torch.allclose(net2.attn1.to_q.weight, w0 + 0.5 * (8.0 / 4) * up @ down, atol=1e-6)
    assert torch.allclose(net2.attn1.to_k.weight, net.attn1.to_k.weight)        # other projections: alpha absent -> 1
    # incomplete / malformed dictionaries fail loudly and touch nothing
    net3 = seeded_init_(Holder(), 5)
    for bad, exc in (({k: v for k, v in lora.items() if not k.endswith("to_q_lora.up.weight")}, KeyError),
                     ({k: v for k, v in lora.items() if not k.endswith("to_q_lora.down.weight")}, KeyError),
                     ({**lora, "attn1.to_q.weight": w0}, KeyError),
                     ({"attn9.processor.to_q_lora.down.weight": down, "attn9.processor.to_q_lora.up.weight": up}, KeyError),
                     ({"attn1.processor.to_x_lora.down.weight": down, "attn1.processor.to_x_lora.up.weight": up}, KeyError),
                     ({"attn1.processor.to_q_lora.down.weight": down[:, :32],
                       "attn1.processor.to_q_lora.up.weight": up}, ValueError)):
        with pytest.raises(exc):
            fold_lora_(net3, bad)
    assert torch.equal(net3.attn1.to_q.weight, w0)
    # network_alphas mapping in the key forms diffusers' loaders produce: all four forms name the same pair;
    # an entry that matches nothing is an error, not a silently wrong scale
    for key in ("attn1.processor.to_q", "unet.attn1.processor.to_q_lora.down.weight.alpha", "attn1.to_q.alpha",
                "unet.attn1.processor.to_q_lora.alpha"):
        net4 = seeded_init_(Holder(), 5)
        fold_lora_(net4, lora, 0.5, network_alphas={key: 8.0})
        assert torch.allclose(net4.attn1.to_q.weight, w0 + 0.5 * (8.0 / 4) * up @ down, atol=1e-6), key
        assert torch.allclose(net4.attn1.to_k.weight, net.attn1.to_k.weight), key
    net5 = seeded_init_(Holder(), 5)
    with pytest.raises(KeyError):
        fold_lora_(net5, lora, 0.5, network_alphas={"unet.attn7.processor.to_q_lora.down.weight.alpha": 8.0})
    assert torch.equal(net5.attn1.to_q.weight, w0)
    # the loader hands over the WHOLE mapping — text-encoder alphas are someone else's and are skipped, and
    # 'to_out.0[.lora...]' stems (the Linear inside the ModuleList) name the out-projection
    net6 = seeded_init_(Holder(), 5)
    wo0 = net6.attn1.to_out[0].weight.detach().clone()
    dn_o, up_o = lora["attn1.processor.to_out_lora.down.weight"], lora["attn1.processor.to_out_lora.up.weight"]
    fold_lora_(net6, lora, 0.5, network_alphas={"text_encoder.text_model.encoder.layers.0.self_attn.q_proj.alpha": 3.0,
                                                "unet.attn1.to_out.0.lora.down.weight.alpha": 8.0,
                                                "unet.attn1.to_q.alpha": 8.0})
    assert torch.allclose(net6.attn1.to_q.weight, w0 + 0.5 * (8.0 / 4) * up @ down, atol=1e-6)
    assert torch.allclose(net6.attn1.to_out[0].weight, wo0 + 0.5 * (8.0 / 4) * up_o @ dn_o, atol=1e-6)


def _synthetic_full_report(n_classes=40):
    """A bench report as bench.main() builds it, with a 40-class roofline table (the size that broke the driver's
    parser in round 3)."""
    rows = []
    for i in range(n_classes):
        rows.append({"kernel": "dd_gemm2_kernel<_Float16, 2, %d, 5, 2, 3, %s, false>" % (i, "true" if i % 2 else "false"),
                     "launches_per_step": 131 - i, "avg_us": 12.54 + i, "ms_per_step": 1.6422 / (i + 1),
                     "bound": "hbm" if i % 3 else "mfma", "achieved": 1091.2, "unit": "GB/s", "frac": 0.1364,
                     "algorithmic_bytes_per_launch": 13679374.04580153, "algorithmic_flops_per_launch": 4203561585.3,
                     "traffic": 28661875.96 if i % 4 else None,
                     "l2_stage": {"staged_bytes_per_launch": 1.2e8, "rate_GBps": 9000.1, "frac_of_guide_lower_bound": 0.53}})
    top = dict(rows[0], peak=8000.0, event_overhead_us_subtracted=3.3, event_overhead_method="x" * 200,
               share_of_timed_kernels=0.131, timed_kernels_ms_per_step=12.534, pmc_table="r04_pmc_traffic.json",
               tuned_table="dualdiff_amd/tuned/gfx950.json", classes=rows)
    return {"metric": "denoising-steps/sec, 6-view 224×400 fp16, 50-step DDIM @ 1/2/4/8 MI355X", "value": 88.04557868,
            "unit": "steps/s", "n_gpus": 8, "steps": 20, "warmup": 5, "ms_per_step": 11.3577, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "fp16", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: " + "w" * 180, "scenes_per_gpu": 1,
                       "parallelism": "scene-sharded x8 (no data-path collective)",
                       "extensions": {"frames_per_scene": 1, "fp8": None, "lora_rank_folded": 0}, "hip_graph": True,
                       "streams": 3, "invariant_conditioning": "recomputed every step", "algorithmic_tflop_per_step": 5.922,
                       "algorithmic_tflop_counting": "c" * 260},
            "model_tflops": 521.4, "executed_tflops": 502.3, "outputs_finite": True, "roofline": top,
            "cpu_baseline": {"value": 0.0758, "unit": "steps/s", "cores": 32, "kind": "port", "sample": "s" * 342,
                             "config2_step_seconds_all": [13.1, 12.7], "bf16": {"value": 0.0917, "sample": "b" * 60},
                             "seconds_spent": 58.4},
            "other_dtype": {"dtype": "bf16", "value": 89.18, "unit": "steps/s", "ms_per_step": 11.21, "outputs_finite": True},
            "speedup_vs_cpu": 1161.5,
            "strong_scaling": {"mode": "view-split", "value": 123.4, "ms_per_step": 8.1, "speedup_vs_n1_ms": 1.4,
                               "message_bytes": 7950000, "verified_on_multi_gpu_hardware": True, "graph": "piecewise"},
            # round 5 legs
            "batched": {"scenes_per_gpu": 4, "value": 116.08, "unit": "scene-steps/s", "ms_per_scene_step": 8.61,
                        "outputs_finite": True,
                        "roofline": {"kernel": "dd_conv3s<f16,4,2,6,2,5,1,0>", "bound": "mfma", "achieved": 901.2, "peak": 2500.0,
                                     "unit": "TFLOP/s", "frac": 0.3605, "avg_us": 151.2, "launches_per_step": 33,
                                     "next": [{"k": "dd_attn5<f16,D40>", "n": 14, "us": 240.1, "b": "mfma", "f": 0.27}] * 3},
                        "roofline_classes": rows},
            "unipc20": {"value": 88.1, "unit": "steps/s", "ms_per_step": 11.35, "ms_per_20_step_sample": 227.0,
                        "outputs_finite": True, "sampler": "u" * 120},
            "dropin": {"value": 77.2, "unit": "steps/s", "ms_per_step": 12.95, "forward_graphs": [1, 1, 1], "vs_fused": 0.87,
                       "outputs_finite": True, "loop": "l" * 200},
            # round 6
            "dropin_varlen": {"value": 71.3, "unit": "steps/s", "ms_per_step": 14.02, "box_counts": [20, 7, 13, 20, 31, 7],
                              "steps_per_sample": 20, "captures": [1, 1, 1], "vs_dropin": 0.924, "outputs_finite": True,
                              "loop": "v" * 200},
            "env": ["DD_PERSIST=0"],
            "collective": {"backend": "nccl", "world_size": 8, "rank_sum": 28, "rank_sum_check": True}}


def test_bench_line_stays_parseable_and_short():
    """BENCH_r03.json.parsed was null because the line had grown to 20 KB.  The printed line must be
    one short JSON object carrying every contract key, whatever the size of the class table."""
    import bench
    full = _synthetic_full_report(40)
    assert len(json.dumps(full)) > 8000                       # the full report is the big one
    line = json.dumps(bench.compact_line(full, "gpurun_out/bench_full_fp16_n8_scenes.json"))
    assert len(line) < 4096 and "\n" not in line
    back = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in back, k
    assert back["value"] == full["value"] and back["config"]["workload"].startswith("BASELINE configs[1]")
    r = back["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["frac"] == 0.1364 and len(r["next"]) == 5 and "classes" not in r and "l2_stage" not in r
    c = back["cpu_baseline"]
    assert set(("value", "unit", "cores", "kind", "sample")) <= set(c)
    assert "1e-3" in back["tolerance"] and back["full_report"].endswith(".json")
    assert back["strong_scaling"]["mode"] == "view-split"
    # round 5: the three new legs ride in the same line, without their free-text and without the batched class table
    assert back["dropin"]["vs_fused"] == 0.87 and "loop" not in back["dropin"]
    assert back["unipc20"]["ms_per_20_step_sample"] == 227.0 and "sampler" not in back["unipc20"]
    assert back["batched"]["roofline"]["frac"] == 0.3605 and "roofline_classes" not in back["batched"]
    # round 6: the changing-box-count leg, the DD_* switches that were set, and the N > 1 collective evidence
    assert back["dropin_varlen"]["vs_dropin"] == 0.924 and back["dropin_varlen"]["captures"] == [1, 1, 1] and "loop" not in back["dropin_varlen"]
    assert back["env"] == ["DD_PERSIST=0"] and back["collective"]["rank_sum_check"] is True
    # degenerate inputs: no roofline / no CPU leg (N > 1 ranks), and an absurdly long free-text field still fits
    bare = dict(full, roofline=None, cpu_baseline=None)
    assert len(json.dumps(bench.compact_line(bare))) < 4096
    assert bench.compact_line({k: v for k, v in full.items() if k != "env"})["env"] == []
    fat = _synthetic_full_report(40)
    fat["strong_scaling"]["error"] = "e" * 6000
    fat["config"]["workload"] = "w" * 3000
    fat_line = json.dumps(bench.compact_line(fat, "x.json"))
    assert len(fat_line) < 4096 and json.loads(fat_line)["value"] == fat["value"]


def test_pmc_summary_counts_only_the_timed_steps(tmp_path):
    """HBM traffic per launch must come from the dispatches of the timed steps, per launch shape —
    not from every dispatch of a symbol in the process (tuner launches behind a 320 MB flush were charged the flush's
    write-back: the '23 MB floor')."""
    import csv
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = ["Dispatch_Id", "Kernel_Name", "Grid_Size", "Workgroup_Size", "Counter_Name", "Counter_Value"]

    def write(d, counter, scale):
        os.makedirs(d)
        rows, did = [], 0
        def add(name, grid, val):
            nonlocal did
            did += 1
            rows.append([did, name, grid, 256, counter, val * scale])
        for _ in range(50):                       # tuner: flush + a tiny GEMM charged with the flush's write-back
            add("FillFunctor", 1 << 20, 327680.0)
            add("dd_gemm2_kernel_tiny", 256, 23000.0)
        for step in range(3):                     # warm-up + 2 timed steps
            add("dd_gemm2_kernel_tiny", 256, 1000.0)
            add("dd_gemm2_kernel_tiny", 256, 1000.0)
            add("dd_gemm2_kernel_tiny", 512, 3000.0)          # same symbol, another shape
            add("dd_cfg_ddim_kernel", 64, 10.0)
        with open(os.path.join(d, "x_counter_collection.csv"), "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(hdr)
            w.writerows(rows)
    write(str(tmp_path / "f"), "FETCH_SIZE", 1.0)
    write(str(tmp_path / "w"), "WRITE_SIZE", 0.5)
    out = str(tmp_path / "o.json")
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "pmc_summary.py"), str(tmp_path / "f"), str(tmp_path / "w"),
                        out, str(tmp_path / "o.csv")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    k = json.load(open(out))["kernels"]
    assert "FillFunctor" not in k                                           # outside the timed steps
    tiny = k["dd_gemm2_kernel_tiny"]
    assert tiny["launches_per_step"] == 3 and len(tiny["shapes"]) == 2
    # (2 x 1000 + 3000) / 3 KB fetched (x2 corrected) + half of that written
    want = ((2 * 1000 + 3000) / 3.0) * 1024 * 2 + ((2 * 1000 + 3000) / 3.0) * 0.5 * 1024
    assert abs(tiny["hbm_bytes_per_launch"] - want) < 1.0
    # the bench line's lookup reads the same structure
    import bench
    assert abs(bench._pmc_traffic("dd_gemm2_kernel_tiny", k) - want) < 1.0


def test_inline_asm_vector_memory_loads_carry_their_own_wait_states():
    """Round 6: a buffer descriptor restored from a spilled scalar (v_readlane) or made uniform (v_readfirstlane) right in
    front of an inline-asm buffer load is read 5 wait states too early — the compiler's hazard recogniser does not look
    inside asm.  Every inline-asm vector-memory load of the kernels therefore starts with its own `s_nop 4`."""
    import glob
    import re
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "dualdiff_amd", "csrc")
    seen = 0
    for path in glob.glob(os.path.join(root, "*.hip")) + glob.glob(os.path.join(root, "*.h")):
        for m in re.finditer(r'asm volatile\("([^"]*(?:buffer_load|global_load|buffer_store|global_store)[^"]*)"', open(path).read()):
            seen += 1
            assert m.group(1).startswith("s_nop 4"), (path, m.group(1))
    assert seen >= 1
