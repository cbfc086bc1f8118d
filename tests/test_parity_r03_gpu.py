"""Round-3 parity cases (VERDICT r2 "Next round" item 1 and ADVICE r2), HIP path vs the CPU oracle:

  * the complete dual-branch step at the BENCH context — 77 text tokens + 20 boxes (+ camera token: Lc = 98), the
    shapes bench.py times — two DDIM steps in fp16 and bf16 with the storage floor computed on the same inputs
    (rounds 1-2 ran the end-to-end cases with 9 text tokens + 5 boxes only);
  * configs[3] at its stated size: the whole video UNet at T = 8 frames x 6 views vs oracle/video_restated.py
    (EXTENSION: the semantics are this build's own, see dualdiff_amd/networks/video_blocks.py);
  * configs[4] at its stated size: 16 frames, fp8 (e4m3fn) attention-projection weights + folded LoRA — whole
    video UNet finite + equal to the same network with the DEQUANTISED weights in 16 bit, and one video block
    at T = 16 against the CPU oracle with dequantised weights;
  * `python bench.py --gpus 2` (no torchrun environment) on the one GPU of the box (gloo plumbing mode) prints
    n_gpus == 2;
  * a 2-rank RCCL run of the view split, skipped unless the box has >= 2 GPUs.

Metric / bound / CSV: tests/parity_util.py (1.02 x floor; why not 1.00: its docstring).
"""
import contextlib
import json
import os
import subprocess
import sys

import pytest
import torch

from oracle import dualdiff_restated as R
from oracle import video_restated as V
from oracle.init_utils import seeded_state_dict, seeded_tensor
from oracle.numerics import storage_emulation
from tests.golden import cases as C
from tests.parity_util import oracle_cache, rel_l2, report

pytestmark = pytest.mark.gpu

PAIR = C.VIEW_PAIR
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
torch.set_num_threads(min(32, os.cpu_count() or 1))
BENCH_NBOX, BENCH_LTXT = 20, 77


def _to_dev(x, dtype):
    if isinstance(x, dict):
        return {k: _to_dev(v, dtype) for k, v in x.items()}
    x = x.cuda()
    return x.to(dtype) if x.is_floating_point() else x


# ------------------------------------------------------------------ bench-context full step ----
@pytest.fixture(scope="module")
def bench_context_case(gpu):
    """Oracle models + 2-step reference / floors of the dual-branch step at the bench context."""
    unet = R.UNet2DConditionModelMultiview(cross_attention_dim=768, neighboring_view_pair=PAIR).eval()
    usd = {k: C.bf16_round(v) for k, v in seeded_state_dict(unet, C.SEED_STEP_UNET).items()}
    unet.load_state_dict(usd)
    cns, csd = [], []
    for occ3d, seed in ((False, C.SEED_STEP_CNET_BG), (True, C.SEED_STEP_CNET_FG)):
        cn = R.BEVControlNetModel(use_occ_3d=occ3d).eval()
        sd = {k: C.bf16_round(v) for k, v in seeded_state_dict(cn, seed).items()}
        cn.load_state_dict(sd)
        cns.append(cn)
        csd.append(sd)
    inp = C.step_inputs(2, nbox=BENCH_NBOX, ltxt=BENCH_LTXT)
    assert inp["text"].shape[1] == 77 and inp["boxes_bg"]["bboxes"].shape[2] == 20
    boxes, conds = [inp["boxes_bg"], inp["boxes_fg"]], [inp["cond_bg"], inp["cond_fg"]]
    ts, ratio = R.ddim_timesteps(50)
    acp = R.ddim_alphas()

    def run(dt):
        x, outs = C.step_latents(), []
        with contextlib.ExitStack() as st, torch.no_grad():
            if dt is not None:
                for m in [unet] + cns:
                    st.enter_context(storage_emulation(m, dt))
            for i in range(2):
                t = int(ts[i])
                x = R.denoise_step(unet, cns, x, t, inp["text"], inp["camera_param"], boxes, conds, 2.0,
                                   R.ddim_coefs(acp, t, ratio))
                if dt is not None:
                    x = x.to(dt).float()
                outs.append(x[0].clone())
        return outs

    def oracle():
        out = {}
        for tag, dt in (("ref", None), ("f16", torch.float16), ("bf16", torch.bfloat16)):
            for i, x in enumerate(run(dt)):
                out["%s_%d" % (tag, i)] = x
        return out
    o = oracle_cache("bench_context_two_steps", oracle)
    ref = [o["ref_0"], o["ref_1"]]
    floors = {torch.float16: [o["f16_0"], o["f16_1"]], torch.bfloat16: [o["bf16_0"], o["bf16_1"]]}
    return usd, csd, inp, ref, floors


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_full_step_bench_context(bench_context_case, dtype):
    """2 ControlNet branches (SFA on) + multiview UNet + CFG + DDIM on 12 view-instances with 98 context tokens —
    bench.py's shapes — through the HIP-graph replay, against the fp32 oracle; floor = the same two steps with the
    reference's storage numerics."""
    from tests.test_parity_r02_gpu import _make_cnet
    from dualdiff_amd.networks.unet_2d_condition_multiview import UNet2DConditionModelMultiview
    from dualdiff_amd.pipeline.pipeline_bev_controlnet import BEVDenoiser
    usd, csd, inp, ref, floors = bench_context_case
    unet = UNet2DConditionModelMultiview(cross_attention_dim=768, neighboring_view_pair=PAIR)
    unet.load_state_dict(usd)
    unet = unet.to("cuda", dtype).eval()
    cns = [_make_cnet(csd[0], False, dtype), _make_cnet(csd[1], True, dtype)]
    den = BEVDenoiser(unet, cns, guidance_scale=2.0, num_inference_steps=50, use_graph=True)
    rec, worst = [], 0.0
    with torch.no_grad():
        den.set_inputs(C.step_latents().cuda().to(dtype), _to_dev(inp["text"], dtype), _to_dev(inp["camera_param"], dtype),
                       [_to_dev(inp["boxes_bg"], dtype), _to_dev(inp["boxes_fg"], dtype)],
                       [_to_dev(inp["cond_bg"], dtype), _to_dev(inp["cond_fg"], dtype)])
        for k in (1, 2):
            den.step(k - 1)
            worst = max(worst, report("bench-context (77 txt + 20 boxes) latents after %d steps" % k,
                                      den.latents[0].float().cpu(), ref[k - 1], dtype, rec, floors[dtype][k - 1]))
    assert worst <= 1.0, rec


# ------------------------------------------------------------------ configs[3]: video UNet at T = 8 ----
def _unet_residual_shapes():
    return [(320, 28, 50)] * 3 + [(320, 14, 25)] + [(640, 14, 25)] * 2 + [(640, 7, 13)] + \
           [(1280, 7, 13)] * 2 + [(1280, 4, 7)] * 3


@pytest.mark.parametrize("dtype", [torch.float16])
def test_video_unet_forward_T8(gpu, dtype):
    """EXTENSION, configs[3] at its stated size: whole video UNet, 8 frames x 6 views = 48 instances, full
    SD-v1.5 widths, ControlNet residuals, vs the CPU definition (slow: ~1.5 min of oracle on the box's host)."""
    from dualdiff_amd.networks.unet_2d_condition_multiview import UNet2DConditionModelMultiviewVideo
    frames = 8
    ora = V.UNet2DConditionModelMultiviewVideo(cross_attention_dim=768, neighboring_view_pair=PAIR, n_frames=frames).eval()
    sd = {k: C.bf16_round(v) for k, v in seeded_state_dict(ora, 61).items()}
    ora.load_state_dict(sd)
    m = frames * 6
    sample = C.bf16_round(seeded_tensor((m, 4, C.H, C.W), 1))
    ctx = C.bf16_round(seeded_tensor((m, 15, 768), 2))
    down = [C.bf16_round(seeded_tensor((m,) + s, 100 + i, 0.3)) for i, s in enumerate(_unet_residual_shapes())]
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
    o = oracle_cache("video_unet_T8_f16", oracle)
    ref, emul = o["ref"], o["emul"]
    net = UNet2DConditionModelMultiviewVideo(cross_attention_dim=768, neighboring_view_pair=PAIR, n_frames=frames)
    net.load_state_dict(sd, strict=True)
    net = net.to("cuda", dtype).eval()
    with torch.no_grad():
        out = net(sample.cuda().to(dtype), 481, encoder_hidden_states=ctx.cuda().to(dtype),
                  down_block_additional_residuals=[d.cuda().to(dtype) for d in down],
                  mid_block_additional_residual=mid.cuda().to(dtype)).sample
    rec = []
    assert out.shape == (m, 4, C.H, C.W)
    assert report("video unet eps (T=8, 6 views)", out, ref, dtype, rec, emul) <= 1.0, rec


# ------------------------------------------------------------------ configs[4]: 16 frames, fp8 + LoRA ----
def _dequantised(sd_weight, dtype):
    from dualdiff_amd import ops as O
    q8, sc = O.quantize_fp8(sd_weight.to(dtype).cuda())
    return (q8.float() * sc[:, None]).cpu()


# ------------------------------------------------------------------ bench.py --gpus N ----
def test_bench_self_launch_two_ranks_on_one_gpu():
    """`python bench.py --gpus 2 ...` WITHOUT a torchrun environment (the shape of the driver's command): the parent
    starts 2 child ranks before touching the GPU; here both share cuda:0 and the bookkeeping collectives run over
    gloo (DD_BENCH_SHARE_GPU / DD_BENCH_BACKEND: plumbing, never a measurement).  n_gpus must be 2."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(DD_BENCH_SHARE_GPU="1", DD_BENCH_BACKEND="gloo")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--single-dtype", "--no-roofline", "--no-cpu-baseline"], capture_output=True, text=True,
                       env=env, cwd=ROOT, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["outputs_finite"] and out["value"] > 0 and out["scaling"] == "weak"


def test_bench_frame_split_two_ranks_on_one_gpu():
    """SURVEY §8e frame split through bench.py: a 4-frame video over 2 self-launched ranks that share cuda:0; every
    video block's ST-Attn sources and temporal K|V cross the ranks through parallel.FrameExchange (gloo, host-staged:
    plumbing, never a measurement).  The line must describe the split and carry the unverified-on-hardware label."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(DD_BENCH_SHARE_GPU="1", DD_BENCH_BACKEND="gloo")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--frames", "4", "--parallelism", "frame-split", "--single-dtype", "--no-roofline",
                        "--no-cpu-baseline"], capture_output=True, text=True, env=env, cwd=ROOT, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["n_gpus"] == 2 and out["outputs_finite"] and out["value"] > 0 and out["scaling"] == "strong"
    fs = out["config"]["frame_split"]
    assert fs["frames_per_shard"] == [2, 2] and fs["verified_on_multi_gpu_hardware"] is False and not fs["cfg_halves_split"]
    assert fs["rank0_temporal_gathered_bytes_per_forward"] > 0 and out["config"]["hip_graph"] is False


def test_bench_frame_split_with_cfg_halves_four_ranks_on_one_gpu():
    """Frame split x CFG split: 4 self-launched ranks on cuda:0 = 2 frame shards x 2 CFG halves of a 4-frame video;
    the frame exchange runs inside the half groups, the rank pairs all-gather the noise prediction (gloo plumbing)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(DD_BENCH_SHARE_GPU="1", DD_BENCH_BACKEND="gloo")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "2", "--warmup", "1",
                        "--frames", "4", "--parallelism", "frame-split", "--single-dtype", "--no-roofline",
                        "--no-cpu-baseline"], capture_output=True, text=True, env=env, cwd=ROOT, timeout=2400)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    fs = out["config"]["frame_split"]
    assert out["n_gpus"] == 4 and out["outputs_finite"] and fs["frames_per_shard"] == [2, 2] and fs["cfg_halves_split"]


def test_bench_challenge_tiles_writes_a_loadable_table(tmp_path):
    """`bench.py --challenge-tiles` (how the 96x64 / 32x64 tiles entered the tracked table): every entry's incumbent is
    timed against the challengers once and the merged table is written to --tune-cache; the file loads back and still
    covers the step's shapes.  (Challenging with a tile that is already the incumbent must change nothing.)"""
    import ast
    cache = str(tmp_path / "table.json")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--challenge-tiles", "52", "--tune-cache", cache,
                        "--steps", "1", "--warmup", "1", "--single-dtype", "--no-roofline", "--no-cpu-baseline"],
                       capture_output=True, text=True, env=env, cwd=ROOT, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    blob = json.load(open(cache))
    assert blob["arch"] == "gfx950" and len(blob["entries"]) >= 100
    keys = {ast.literal_eval(k) for k, _ in blob["entries"]}
    assert ("g", 1092, 1280, 1280, 0, 0, False, False) in keys
    tracked = {k: v for k, v in json.load(open(os.path.join(ROOT, "dualdiff_amd", "tuned", "gfx950.json")))["entries"]}
    changed = [k for k, v in blob["entries"] if k in tracked and tracked[k][0] == 52 and v[0] != 52]
    assert not changed, changed          # an incumbent cannot lose to itself


# ------------------------------------------------------------------ RCCL view split (>= 2 GPUs) ----
@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs >= 2 GPUs (RCCL device path of HaloExchange)")
def test_view_split_two_ranks_rccl():
    """the RCCL device path of the view split — batch_isend_irecv on device tensors inside the
    half group, stream ordering against the attn4 kernels — compared with the unsharded run.  tools/view_split_rccl.py
    runs both and prints the relative error; it is launched under torchrun as a child process."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29631",
                        os.path.join(ROOT, "tools", "view_split_rccl.py")], capture_output=True, text=True, env=env,
                       cwd=ROOT, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("VIEW_SPLIT_RCCL")][-1]
    err = float(line.split("rel_l2=")[1].split()[0])
    assert err <= 2e-3, line


@pytest.mark.parametrize("dtype", [torch.float16])
def test_fp8_mfma_lora_video_unet_16_frames(gpu, dtype):
    """EXTENSION, configs[4] at its stated size on the fp8 MATRIX path (round 3: W8A8, dd_gemm8): the whole 16-frame video
    UNet (one CFG half = 96 instances) with a folded rank-4 LoRA and enable_fp8_weights(mfma=True) — finite, and within the
    quantisation noise of two e4m3 operands of the same network in 16 bit (the op-level tests of tests/test_fp8_mfma_gpu.py
    pin the arithmetic exactly; here the bound only guards against a broken path at full size)."""
    from dualdiff_amd.lora import fold_lora_, lora_keys
    from dualdiff_amd.networks.layers import device_init_, enable_fp8_weights
    from dualdiff_amd.networks.unet_2d_condition_multiview import UNet2DConditionModelMultiviewVideo
    frames, m = 16, 96
    with torch.device("cuda"):
        net = UNet2DConditionModelMultiviewVideo(cross_attention_dim=768, neighboring_view_pair=PAIR, n_frames=frames).to(dtype)
    device_init_(net, 3)
    g = torch.Generator(device="cuda").manual_seed(77)
    lora = {k: torch.randn(shape, generator=g, device="cuda") * 0.02 for k, shape in sorted(lora_keys(net, 4).items())}
    fold_lora_(net, lora, 1.0)
    net.eval()
    x = torch.randn((m, 4, C.H, C.W), generator=g, device="cuda").to(dtype)
    ctx = torch.randn((m, 98, 768), generator=g, device="cuda").to(dtype)
    with torch.no_grad():
        y16 = net(x, 481, encoder_hidden_states=ctx).sample.float().cpu()
        enable_fp8_weights(net, mfma=True)
        y8 = net(x, 481, encoder_hidden_states=ctx).sample.float().cpu()
    e = rel_l2(y8, y16)
    print("fp8 MFMA (W8A8) + LoRA video UNet, 16 frames: vs 16-bit rel-L2 %.3e" % e)
    from tests.parity_util import log_row
    log_row("fp8 MFMA W8A8 + LoRA video unet T=16 vs 16-bit", dtype, e, 0.0, 2.5e-1)
    # a random-init UNet amplifies any perturbation ~10x from a block to its output (two valid fp16 roundings of this
    # network already differ by 1.6e-3 there); e4m3 operands (2^-4 relative) in 31 projections land at ~1e-1.  What
    # the quantisation does to SAMPLES is what tools/frechet_ext.py reports.
    assert torch.isfinite(y8).all() and 1e-4 < e < 2.5e-1
