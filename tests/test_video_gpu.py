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
    def oracle():
        with torch.no_grad():
            ref = run()
            with storage_emulation(ora, dtype):
                emul = run()
        return {"ref": ref, "emul": emul}
    from tests.parity_util import oracle_cache
    o = oracle_cache("video_unet_T2_%s" % str(dtype).split(".")[-1], oracle)
    ref, emul = o["ref"], o["emul"]
    net = UNet2DConditionModelMultiviewVideo(cross_attention_dim=768, neighboring_view_pair=PAIR, n_frames=frames)
    net.load_state_dict(sd, strict=True)
    net = net.to("cuda", dtype).eval()
    with torch.no_grad():
        out = net(sample.cuda().to(dtype), 481, encoder_hidden_states=ctx.cuda().to(dtype),
                  down_block_additional_residuals=[d.cuda().to(dtype) for d in down],
                  mid_block_additional_residual=mid.cuda().to(dtype)).sample
    rec = []
    assert report("video unet eps (T=2, 6 views)", out, ref, dtype, rec, emul) <= 1.0, rec


# ------------------------------------------------------------------ frame split on one GPU ----
class _LocalFrameExchange:
    """In-process stand-in for parallel.FrameExchange: the frame shards run as threads of this process (one HIP
    stream each) and read each other's buffers between two barriers."""

    def __init__(self, plans):
        import threading
        self.plans = plans
        self.barrier = threading.Barrier(len(plans))
        self.slots = {}

    def bind(self, plan):
        outer = self

        class Bound:
            def _swap(self, key, value):
                torch.cuda.current_stream().synchronize()
                outer.slots[(key, plan.rank)] = value
                outer.barrier.wait()

            def _done(self):
                torch.cuda.current_stream().synchronize()
                outer.barrier.wait()

            def st_sources(self, first_local, last_local):
                self._swap("st", (first_local, last_local))
                first = outer.slots[("st", 0)][0].clone()
                prev = outer.slots[("st", max(plan.rank - 1, 0))][1 if plan.rank else 0].clone()
                self._done()
                return first, prev

            def gather_frames(self, local):
                self._swap("tmp", local)
                out = torch.cat([outer.slots[("tmp", p.rank)] for p in outer.plans], dim=0)
                self._done()
                return out
        return Bound()


def _run_threads(work, items, ex):
    import threading
    errs = []

    def guarded(it):
        try:
            with torch.no_grad(), torch.cuda.stream(torch.cuda.Stream()):
                work(it)
                torch.cuda.current_stream().synchronize()
        except Exception as e:          # noqa: BLE001  (reported to the main thread)
            errs.append(e)
            ex.barrier.abort()
    torch.cuda.synchronize()
    ths = [threading.Thread(target=guarded, args=(it,)) for it in items]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    assert not errs, errs


@pytest.mark.parametrize("dim,n,frames,scenes,shards", [(640, 350, 4, 2, 2), (1280, 91, 8, 1, 3), (320, 1400, 3, 1, 3)])
def test_frame_split_video_block_one_gpu(gpu, dim, n, frames, scenes, shards):
    """One video block with its T frames spread over `shards` virtual ranks (threads on one GPU): ST-Attn fetches
    frame 0 and the previous frame across the cut, temporal attention runs local queries against the gathered K|V of
    all frames — the concatenated result reproduces the unsharded block up to the storage rounding (row counts per
    launch differ, so GEMM tiles do).  Balanced and ragged (8 frames over 3) splits."""
    from dualdiff_amd.networks.video_blocks import VideoMultiviewTransformerBlock
    from dualdiff_amd.parallel import FrameShard, FrameSplitPlan
    from tests.parity_util import log_row, rel_l2
    dtype, hd = torch.float16, dim // 8
    ora = V.VideoMultiviewTransformerBlock(dim, 8, hd, cross_attention_dim=768, neighboring_view_pair=PAIR, n_frames=frames)
    sd = {k: C.bf16_round(v) for k, v in seeded_state_dict(ora, 51).items()}
    x = C.bf16_round(seeded_tensor((scenes, frames, 6, n, dim), 1)).cuda().to(dtype)
    ctx = C.bf16_round(seeded_tensor((scenes, frames, 6, 20, 768), 2)).cuda().to(dtype)

    def make():
        blk = VideoMultiviewTransformerBlock(dim, 8, hd, cross_attention_dim=768, neighboring_view_pair=PAIR, n_frames=frames)
        blk.load_state_dict(sd, strict=True)
        return blk.to("cuda", dtype)
    with torch.no_grad():
        want = make().run(x.reshape(-1, dim), scenes * frames * 6, n, ctx.reshape(-1, 768), 20) \
            .reshape(scenes, frames, 6, n, dim).float().cpu()
    plans = [FrameSplitPlan(shards, r, frames) for r in range(shards)]
    ex = _LocalFrameExchange(plans)
    outs = {}

    def work(p):
        blk = make()
        blk.frame_shard = FrameShard(p, ex.bind(p))
        xs, cs = x[:, p.lo:p.hi].contiguous(), ctx[:, p.lo:p.hi].contiguous()
        m = scenes * p.n_local * 6
        outs[p.rank] = blk.run(xs.reshape(-1, dim), m, n, cs.reshape(-1, 768), 20) \
            .reshape(scenes, p.n_local, 6, n, dim).float().cpu()
    _run_threads(work, plans, ex)
    got = torch.cat([outs[p.rank] for p in plans], dim=1)
    e = rel_l2(got, want)
    print("frame split x%d block C=%d T=%d: vs unsharded rel-L2 %.3e" % (shards, dim, frames, e))
    log_row("frame split x%d video block C=%d T=%d vs unsharded" % (shards, dim, frames), dtype, e, 0.0, 1e-3)
    assert e <= 1e-3
    # the exchange matters: without the other ranks' frames the last shard's result is far off
    blk = make()
    blk.n_frames = plans[-1].n_local
    p = plans[-1]
    with torch.no_grad():
        alone = blk.run(x[:, p.lo:p.hi].reshape(-1, dim), scenes * p.n_local * 6, n,
                        ctx[:, p.lo:p.hi].reshape(-1, 768), 20).reshape(scenes, p.n_local, 6, n, dim).float().cpu()
    assert rel_l2(alone, want[:, p.lo:p.hi]) > 20 * max(e, 1e-4)


def test_frame_split_video_unet_one_gpu(gpu):
    """Whole video UNet (4 frames x 6 views, full widths, ControlNet residuals) with its frames over 2 virtual ranks
    vs the unsharded network: every video block exchanges through FrameShard; eps agrees to the storage rounding."""
    from dualdiff_amd.networks.unet_2d_condition_multiview import UNet2DConditionModelMultiviewVideo
    from dualdiff_amd.parallel import FrameShard, FrameSplitPlan
    from tests.parity_util import log_row, rel_l2
    dtype, frames, shards = torch.float16, 4, 2
    ora = V.UNet2DConditionModelMultiviewVideo(cross_attention_dim=768, neighboring_view_pair=PAIR, n_frames=frames)
    sd = {k: C.bf16_round(v) for k, v in seeded_state_dict(ora, 61).items()}
    del ora
    sample = C.bf16_round(seeded_tensor((frames, 6, 4, C.H, C.W), 1)).cuda().to(dtype)
    ctx = C.bf16_round(seeded_tensor((frames, 6, 15, 768), 2)).cuda().to(dtype)
    shapes = [(320, 28, 50)] * 3 + [(320, 14, 25)] + [(640, 14, 25)] * 2 + [(640, 7, 13)] + \
             [(1280, 7, 13)] * 2 + [(1280, 4, 7)] * 3
    down = [C.bf16_round(seeded_tensor((frames, 6) + s, 100 + i, 0.3)).cuda().to(dtype) for i, s in enumerate(shapes)]
    mid = C.bf16_round(seeded_tensor((frames, 6, 1280, 4, 7), 130, 0.3)).cuda().to(dtype)

    def make():
        net = UNet2DConditionModelMultiviewVideo(cross_attention_dim=768, neighboring_view_pair=PAIR, n_frames=frames)
        net.load_state_dict(sd, strict=True)
        return net.to("cuda", dtype).eval()

    def fwd(net, lo, hi):
        f = lambda t: t[lo:hi].reshape((-1,) + tuple(t.shape[2:]))          # noqa: E731
        return net(f(sample), 481, encoder_hidden_states=f(ctx), down_block_additional_residuals=[f(d) for d in down],
                   mid_block_additional_residual=f(mid)).sample.float().cpu()
    with torch.no_grad():
        want = fwd(make(), 0, frames)
    plans = [FrameSplitPlan(shards, r, frames) for r in range(shards)]
    ex = _LocalFrameExchange(plans)
    nets = [make() for _ in plans]
    for net, p in zip(nets, plans):
        net.set_frame_shard(FrameShard(p, ex.bind(p)))
    outs = {}

    def work(i):
        outs[i] = fwd(nets[i], plans[i].lo, plans[i].hi)
    _run_threads(work, list(range(shards)), ex)
    got = torch.cat([outs[i] for i in range(shards)], dim=0)
    e = rel_l2(got, want)
    print("frame split x2 video UNet T=4: eps vs unsharded rel-L2 %.3e" % e)
    log_row("frame split x2 video unet eps (T=4) vs unsharded", dtype, e, 0.0, 2e-3)
    assert e <= 2e-3
    with pytest.raises(ValueError):
        nets[0].set_frame_shard(FrameShard(FrameSplitPlan(2, 0, 6)))
    nets[0].set_frame_shard(None)
    assert all(getattr(m, "frame_shard", None) is None for m in nets[0].modules())
