"""Seeds, reduced configurations and seeded inputs of the golden-vector cases.

Shared by tests/golden/mint.py (which runs the reference source on them) and
tests/test_oracle_golden.py (which runs the oracle on them).  Inputs are regenerated from seeds;
only the reference OUTPUTS are stored (tests/golden/*.npz).  No reference code here.
"""
import torch

from oracle.init_utils import seeded_tensor

SEED_SFA, SEED_EMB, SEED_BOX, SEED_PROC, SEED_BLOCK, SEED_UNET, SEED_CNET = 11, 12, 13, 14, 15, 16, 17

# configs/dataset/Nuscenes.yaml:27-33
VIEW_PAIR = {0: [5, 1], 1: [0, 2], 2: [1, 3], 3: [2, 4], 4: [3, 5], 5: [4, 0]}

H, W = 28, 50           # 224x400 / 8
N_CAM = 6


def sfa_inputs():
    return seeded_tensor((6, 320, H, W), 101), seeded_tensor((6, 77, 768), 102)


def cond_image():
    g = torch.Generator().manual_seed(103)
    return torch.rand((1, 3, 224, 2400), generator=g)


SEED_BEV_EMB = 27


def bev_map():
    """(1, 25, 200, 200) binary-ish BEV map layers (map_embedder.py:23 conditioning_size)."""
    g = torch.Generator().manual_seed(107)
    return (torch.rand((1, 25, 200, 200), generator=g) > 0.7).float()


def box_inputs():
    g = torch.Generator().manual_seed(104)
    bb = (torch.rand((12, 5, 8, 3), generator=g) - 0.5) * 100.0
    cl = torch.randint(0, 10, (12, 5), generator=g)
    mk = torch.rand((12, 5), generator=g) > 0.3
    return bb, cl, mk


def proc_inputs():
    return seeded_tensor((3, 140, 320), 105), seeded_tensor((3, 13, 768), 106)


def adapter_inputs():
    """hidden (3, 140, 320); context = 13 text(+cam) tokens | 5 box tokens | 5 class tokens."""
    return seeded_tensor((3, 140, 320), 115), seeded_tensor((3, 13 + 5 + 5, 768), 116), 5


SEED_ADAPTER = 21


def block_kwargs():
    return dict(dim=64, num_attention_heads=8, attention_head_dim=8, cross_attention_dim=48)


def block_inputs():
    return seeded_tensor((12, 35, 64), 107), seeded_tensor((12, 9, 48), 108)


def unet_kwargs():
    return dict(block_out_channels=(32, 64, 128, 128), cross_attention_dim=64, attention_head_dim=8)


def unet_inputs():
    m = N_CAM
    sample = seeded_tensor((m, 4, H, W), 109)
    t = torch.tensor(481)
    ctx = seeded_tensor((m, 11, 64), 110)
    shapes = [(32, 28, 50)] * 3 + [(32, 14, 25)] + [(64, 14, 25)] * 2 + [(64, 7, 13)] + \
             [(128, 7, 13)] * 2 + [(128, 4, 7)] * 3
    down = [seeded_tensor((m,) + s, 200 + i, 0.5) for i, s in enumerate(shapes)]
    mid = seeded_tensor((m, 128, 4, 7), 230, 0.5)
    return sample, t, ctx, down, mid


def controlnet_kwargs():
    """Constructor kwargs for the REFERENCE class (diffusers-style names)."""
    return {
        "diffusers": dict(in_channels=4, block_out_channels=(320, 64, 128, 128), layers_per_block=2,
                          cross_attention_dim=768, attention_head_dim=8, norm_num_groups=32,
                          camera_in_dim=189, camera_out_dim=768, uncond_cam_in_dim=(3, 7),
                          conditioning_embedding_out_channels=(16, 32, 96, 256), drop_cond_ratio=0.0),
        "cond_channels": (16, 32, 96, 256),
    }


def controlnet_oracle_kwargs():
    return dict(in_channels=4, block_out_channels=(320, 64, 128, 128), layers_per_block=2,
                cross_attention_dim=768, attention_head_dim=8, use_txt_con_fusion=True)


def controlnet_inputs(occ3d):
    """b = 2 scenes (so the (b n) flattening order matters), 6 views, per-scene timesteps."""
    b = 2
    g = torch.Generator().manual_seed(300 + int(occ3d))
    inp = {
        "sample": seeded_tensor((b, N_CAM, 4, H, W), 301),
        "timestep": torch.tensor([981, 41]),
        "camera_param": seeded_tensor((b, N_CAM, 3, 7), 302),
        "encoder_hidden_states": seeded_tensor((b, 9, 768), 303),
        "conditioning_scale": 0.75,
    }
    n_box_views = 1 if occ3d else N_CAM       # fg branch shares boxes over views (bbox_view_shared)
    inp["bboxes_3d_data"] = {
        "bboxes": (torch.rand((b, n_box_views, 4, 8, 3), generator=g) - 0.5) * 100.0,
        "classes": torch.randint(0, 10, (b, n_box_views, 4), generator=g),
        "masks": torch.rand((b, n_box_views, 4), generator=g) > 0.3,
    }
    if occ3d:   # ORS-3D class ids / 17 (dataset/utils.py:417-420)
        inp["controlnet_cond"] = torch.randint(0, 18, (b * N_CAM, 320, H, W), generator=g).float() / 17.0
    else:       # ORS panorama image
        inp["controlnet_cond"] = torch.rand((b, 3, 224, 2400), generator=g)
    return inp


MAX_STORED = 16384


def compact(key, t):
    """Golden-fixture encoding of one fp32 tensor: full tensor when small, else an evenly
    strided subsample of the flattened tensor + float64 checksums of the whole tensor."""
    import numpy as np
    flat = t.reshape(-1)
    n = flat.numel()
    if n <= MAX_STORED:
        return {key: flat.numpy().reshape(tuple(t.shape))}
    stride = (n + MAX_STORED - 1) // MAX_STORED
    stride += 1 - stride % 2            # odd stride: walks across rows / channels
    return {key + "__sub": flat[::stride].numpy(), key + "__stride": np.int64(stride),
            key + "__shape": np.asarray(t.shape, dtype=np.int64),
            key + "__sum": np.float64(flat.double().sum().item()),
            key + "__abssum": np.float64(flat.double().abs().sum().item())}


def compare(npz, key, t, rtol, atol):
    """Checks tensor `t` against the fixture entry `key`; returns the max abs error seen."""
    import numpy as np
    t = t.detach().float()
    if key in npz.files:
        ref = torch.from_numpy(npz[key])
        assert tuple(ref.shape) == tuple(t.shape), (key, ref.shape, t.shape)
        err = (t - ref).abs().max().item()
        assert torch.allclose(t, ref, rtol=rtol, atol=atol), "%s: max err %g" % (key, err)
        return err
    assert tuple(npz[key + "__shape"]) == tuple(t.shape), (key, npz[key + "__shape"], t.shape)
    stride = int(npz[key + "__stride"])
    flat = t.reshape(-1)
    ref = torch.from_numpy(npz[key + "__sub"])
    sub = flat[::stride]
    err = (sub - ref).abs().max().item()
    assert torch.allclose(sub, ref, rtol=rtol, atol=atol), "%s: max err %g" % (key, err)
    n = flat.numel()
    s, a = flat.double().sum().item(), flat.double().abs().sum().item()
    tol = atol * n ** 0.5 * 4 + rtol * float(npz[key + "__abssum"])
    assert abs(s - float(npz[key + "__sum"])) <= tol, (key, "sum", s, float(npz[key + "__sum"]))
    assert abs(a - float(npz[key + "__abssum"])) <= tol, (key, "abssum")
    return err


# ------------------------------------------------------------------- ORS projection (N3) ----
ORS_H, ORS_W, ORS_S = 28, 50, 320
ORS_RATIO = 400 / 8 / 1600            # misc/test_utils.py:215 for 224x400 images
ORS_VIEWS = ["CAM_FRONT_LEFT", "CAM_FRONT", "CAM_FRONT_RIGHT", "CAM_BACK_RIGHT", "CAM_BACK", "CAM_BACK_LEFT"]


def ors_inputs():
    """Seeded stand-ins for the reference's data files: a 200 x 200 x 16 class volume (0..17) with
    structure (ground plane, boxes), and 6 cameras on a ring looking outwards (nuScenes-like
    intrinsics for a 1600 x 900 image, camera ~1.5 m above the ground)."""
    import math
    g = torch.Generator().manual_seed(131)
    occ = torch.full((200, 200, 16), 17, dtype=torch.int64)
    occ[:, :, 2] = 11                                           # drivable surface layer
    occ[:, :, :2] = 15
    for _ in range(60):                                         # boxes of random classes
        x0, y0 = int(torch.randint(0, 190, (1,), generator=g)), int(torch.randint(0, 190, (1,), generator=g))
        sx, sy, sz = [int(torch.randint(2, 10, (1,), generator=g)) for _ in range(3)]
        occ[x0:x0 + sx, y0:y0 + sy, 3:3 + sz] = int(torch.randint(0, 17, (1,), generator=g))
    Ks, Rts = [], []
    for i in range(6):
        yaw = math.radians(55.0 - 60.0 * i) + float(torch.rand(1, generator=g) - 0.5) * 0.1
        # camera axes in the ego frame: z forward (along yaw), x right, y down
        fwd = torch.tensor([math.cos(yaw), math.sin(yaw), 0.0])
        down = torch.tensor([0.0, 0.0, -1.0])
        right = torch.linalg.cross(down, fwd)
        R = torch.stack([right, down, fwd], dim=1)               # columns = camera axes
        t = torch.tensor([1.5 * math.cos(yaw), 0.5 * math.sin(yaw), 1.5])
        Rt = torch.eye(4)
        Rt[:3, :3], Rt[:3, 3] = R, t
        f = 1260.0 + 10.0 * i
        K = torch.tensor([[f, 0.0, 800.0 + i], [0.0, f, 450.0 - i], [0.0, 0.0, 1.0]])
        Ks.append(K)
        Rts.append(Rt)
    return occ, Ks, Rts


# ---------------------------------------------------------------- full dual-branch step / trajectory ----
# The config-2 step at full SD-v1.5 widths (b = 1 scene: 12 view-instances, 2 ControlNet branches, SFA on,
# CFG 2.0) with short token lists (9 text tokens, 5 boxes) so that the CPU oracle stays affordable.  Same
# seeds as tests/test_model_gpu.py's dual-branch test.  Used by tests/golden/mint_trajectory.py (oracle
# trajectory -> trajectory_ddim50.npz) and tests/test_parity_r02_gpu.py (HIP path vs that fixture).
STEP_NBOX, STEP_LTXT = 5, 9
SEED_STEP_UNET, SEED_STEP_CNET_BG, SEED_STEP_CNET_FG, SEED_STEP_LAT = 21, 31, 32, 77
TRAJ_CHECKPOINTS = (1, 2, 3, 5, 10, 20, 30, 40, 50)


def bf16_round(t):
    return t.to(torch.bfloat16).to(torch.float32) if t.is_floating_point() else t


def step_inputs(b=2, nbox=None, ltxt=None):
    """(uncond, cond) rows of every conditioning input, rounded to bf16-representable values.  nbox / ltxt default
    to the short token lists of the minted trajectory (5 boxes, 9 text tokens); the bench-context case passes the
    bench workload's 20 boxes / 77 text tokens (Lc = 1 + 77 + 20 = 98)."""
    STEP_NBOX, STEP_LTXT = (nbox or globals()["STEP_NBOX"]), (ltxt or globals()["STEP_LTXT"])
    g = torch.Generator().manual_seed(7)
    return {
        "sample": bf16_round(seeded_tensor((b, N_CAM, 4, H, W), 11)),
        "timestep": torch.tensor([981, 41][:b]),          # int64 like the scheduler hands them over (pipeline_bev_controlnet.py:381); a float tensor would be cast to the storage dtype with the other inputs (bf16: 981 -> 980)
        "camera_param": bf16_round(seeded_tensor((b, N_CAM, 3, 7), 12)),
        "text": bf16_round(seeded_tensor((b, STEP_LTXT, 768), 13)),
        "boxes_bg": {"bboxes": bf16_round((torch.rand((b, N_CAM, STEP_NBOX, 8, 3), generator=g) - 0.5) * 20.0),
                     "classes": torch.randint(0, 10, (b, N_CAM, STEP_NBOX), generator=g),
                     "masks": torch.rand((b, N_CAM, STEP_NBOX), generator=g) > 0.3},
        "boxes_fg": {"bboxes": bf16_round((torch.rand((b, 1, STEP_NBOX, 8, 3), generator=g) - 0.5) * 20.0),
                     "classes": torch.randint(0, 10, (b, 1, STEP_NBOX), generator=g),
                     "masks": torch.rand((b, 1, STEP_NBOX), generator=g) > 0.3},
        "cond_bg": bf16_round(torch.rand((b, 3, 224, 2400), generator=g)),
        "cond_fg": bf16_round(torch.randint(0, 18, (b * N_CAM, 320, H, W), generator=g).float() / 17.0),
    }


def step_latents():
    """(1, 6, 4, 28, 50): one draw replicated over the 6 views (pipeline_bev_controlnet.py:345)."""
    return bf16_round(seeded_tensor((1, 4, H, W), SEED_STEP_LAT))[:, None].expand(-1, N_CAM, -1, -1, -1).contiguous()


# variants of the multiview block the reference also defines (blocks.py:81-90,106-142): golden fixture
# multiview_block_variants.npz (tests/golden/mint.py variants)
BLOCK_VARIANTS = (("concat", "zero_linear"), ("self", "zero_linear"), ("add", "gated"), ("add", "none"))
SEED_BLOCK_VAR = 25
