"""EXTENSION (BASELINE configs[3], no reference semantics): the video transformer block — ST-Attn + cross-attn +
cross-view attn + temporal attn — on the HIP kernels vs its CPU definition oracle/video_restated.py.
Same metric / bound / CSV as the image-path parity tests (tests/parity_util.py)."""
import os

import pytest
import torch

from oracle import video_restated as V
from oracle.init_utils import seeded_state_dict, seeded_tensor
from oracle.numerics import storage_emulation
from tests.golden import cases as C
from tests.parity_util import report

pytestmark = pytest.mark.gpu
PAIR = C.VIEW_PAIR
DTYPES = [torch.float16, torch.bfloat16]
torch.set_num_threads(min(32, os.cpu_count() or 1))


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("dim,n,frames,scenes", [(320, 1400, 3, 1), (640, 350, 4, 2), (1280, 91, 8, 1), (1280, 28, 2, 1)])
def test_video_block(gpu, dtype, dim, n, frames, scenes):
    from dualdiff_amd.networks.video_blocks import VideoMultiviewTransformerBlock
    hd = dim // 8
    ora = V.VideoMultiviewTransformerBlock(dim, 8, hd, cross_attention_dim=768, neighboring_view_pair=PAIR,
                                           n_frames=frames).eval()
    sd = {k: C.bf16_round(v) for k, v in seeded_state_dict(ora, 51).items()}      # attn_temp.to_out non-zero
    ora.load_state_dict(sd)
    m = scenes * frames * 6
    x = C.bf16_round(seeded_tensor((m, n, dim), 1))
    ctx = C.bf16_round(seeded_tensor((m, 20, 768), 2))
    with torch.no_grad():
        ref = ora(x, encoder_hidden_states=ctx)
        with storage_emulation(ora, dtype):
            emul = ora(x, encoder_hidden_states=ctx)
        blk = VideoMultiviewTransformerBlock(dim, 8, hd, cross_attention_dim=768, neighboring_view_pair=PAIR,
                                             n_frames=frames)
        blk.load_state_dict(sd, strict=True)
        blk = blk.to("cuda", dtype)
        y = blk.run(x.cuda().to(dtype).reshape(-1, dim), m, n, ctx.cuda().to(dtype).reshape(-1, 768), 20)
    rec = []
    r = report("video block C=%d n=%d T=%d" % (dim, n, frames), y.reshape(m, n, dim), ref, dtype, rec, emul)
    assert r <= 1.0, rec


def test_video_block_single_frame_equals_image_block(gpu):
    """T = 1: ST-Attn sees [frame 0 ; frame 0] (duplicated keys do not change a softmax average) and temporal
    attention over one frame is V itself -> with attn_temp.to_out = 0 the video block IS the image block."""
    from dualdiff_amd.networks.blocks import BasicMultiviewTransformerBlock
    from dualdiff_amd.networks.video_blocks import VideoMultiviewTransformerBlock
    from oracle import dualdiff_restated as R
    dtype = torch.float16
    ora = R.BasicMultiviewTransformerBlock(640, 8, 80, cross_attention_dim=768, neighboring_view_pair=PAIR)
    sd = {k: C.bf16_round(v) for k, v in seeded_state_dict(ora, 5).items()}
    img = BasicMultiviewTransformerBlock(640, 8, 80, cross_attention_dim=768, neighboring_view_pair=PAIR)
    img.load_state_dict(sd)
    vid = VideoMultiviewTransformerBlock(640, 8, 80, cross_attention_dim=768, neighboring_view_pair=PAIR, n_frames=1)
    missing, unexpected = vid.load_state_dict(sd, strict=False)
    assert not unexpected and all(k.startswith(("norm_temp", "attn_temp")) for k in missing)
    with torch.no_grad():
        for p_ in list(vid.norm_temp.parameters()) + list(vid.attn_temp.parameters()):
            p_.normal_(0, 0.02)
        vid.attn_temp.to_out[0].weight.zero_()
        vid.attn_temp.to_out[0].bias.zero_()
        img, vid = img.to("cuda", dtype), vid.to("cuda", dtype)
        x = C.bf16_round(seeded_tensor((6, 350, 640), 1)).cuda().to(dtype).reshape(-1, 640)
        ctx = C.bf16_round(seeded_tensor((6, 30, 768), 2)).cuda().to(dtype).reshape(-1, 768)
        a = img.run(x, 6, 350, ctx, 30).float()
        b = vid.run(x, 6, 350, ctx, 30).float()
    e = ((a - b).norm() / a.norm()).item()
    print("video(T=1) vs image block rel-L2 %.3e" % e)
    assert e <= 1e-3


@pytest.mark.parametrize("dtype", [torch.float16])
def test_video_unet_forward(gpu, dtype):
    """Whole video UNet (2 frames x 6 views, full SD-v1.5 widths, ControlNet residuals) vs the oracle."""
    from dualdiff_amd.networks.unet_2d_condition_multiview import UNet2DConditionModelMultiviewVideo
    frames = 2
    ora = V.UNet2DConditionModelMultiviewVideo(cross_attention_dim=768, neighboring_view_pair=PAIR, n_frames=frames).eval()
    sd = {k: C.bf16_round(v) for k, v in seeded_state_dict(ora, 61).items()}
    ora.load_state_dict(sd)
    m = frames * 6
    sample = C.bf16_round(seeded_tensor((m, 4, C.H, C.W), 1))
    ctx = C.bf16_round(seeded_tensor((m, 15, 768), 2))
    shapes = [(320, 28, 50)] * 3 + [(320, 14, 25)] + [(640, 14, 25)] * 2 + [(640, 7, 13)] + \
             [(1280, 7, 13)] * 2 + [(1280, 4, 7)] * 3
    down = [C.bf16_round(seeded_tensor((m,) + s, 100 + i, 0.3)) for i, s in enumerate(shapes)]
    mid = C.bf16_round(seeded_tensor((m, 1280, 4, 7), 130, 0.3))

    def run():
        return ora(sample, torch.tensor(481), encoder_hidden_states=ctx, down_block_additional_residuals=down,
                   mid_block_additional_residual=mid).sample
    with torch.no_grad():
        ref = run()
        with storage_emulation(ora, dtype):
            emul = run()
    net = UNet2DConditionModelMultiviewVideo(cross_attention_dim=768, neighboring_view_pair=PAIR, n_frames=frames)
    net.load_state_dict(sd, strict=True)
    net = net.to("cuda", dtype).eval()
    with torch.no_grad():
        out = net(sample.cuda().to(dtype), 481, encoder_hidden_states=ctx.cuda().to(dtype),
                  down_block_additional_residuals=[d.cuda().to(dtype) for d in down],
                  mid_block_additional_residual=mid.cuda().to(dtype)).sample
    rec = []
    assert report("video unet eps (T=2, 6 views)", out, ref, dtype, rec, emul) <= 1.0, rec
