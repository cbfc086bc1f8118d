"""N2: VAE decode (`decode_latents`, pipeline_bev_controlnet.py:101-113) on the HIP path against the fp32
CPU restatement of diffusers' AutoencoderKL decoder (oracle/vae_decoder.py — parity unpinned: diffusers
and the SD-v1.5 VAE weights are not available here; weights are seeded).  Same metric and bound as
tests/test_model_gpu.py:  e(HIP) <= max(1e-3, 1.5 * e_floor)  with e_floor the error of the oracle
run with every leaf output rounded to the storage dtype."""
import os

import pytest
import torch

from oracle import vae_decoder as V
from oracle.init_utils import seeded_state_dict, seeded_tensor
from oracle.numerics import storage_emulation

pytestmark = pytest.mark.gpu
torch.set_num_threads(min(32, os.cpu_count() or 1))


def bf16_round(t):
    return t.to(torch.bfloat16).to(torch.float32)


def rel_l2(y, ref):
    y, ref = y.detach().float().cpu(), ref.float()
    return ((y - ref).norm() / (ref.norm() + 1e-20)).item()


@pytest.fixture(scope="module")
def vae_case(gpu):
    ora = V.AutoencoderKLDecoder().eval()
    sd = {k: bf16_round(v) for k, v in seeded_state_dict(ora, 31).items()}
    ora.load_state_dict(sd)
    return ora, sd


def _hip(sd, dtype):
    from dualdiff_amd.networks.vae_decoder import AutoencoderKLDecoder
    net = AutoencoderKLDecoder()
    missing, unexpected = net.load_state_dict(sd, strict=True)
    return net.to("cuda", dtype).eval()


def _check(name, y, ref, emul, dtype):
    e, fl = rel_l2(y, ref), rel_l2(emul, ref)
    bound = max(1e-3, 1.5 * fl)
    print("%-30s %-8s e_hip=%.3e e_floor=%.3e bound=%.3e" % (name, str(dtype).split(".")[-1], e, fl, bound))
    assert torch.isfinite(y.float()).all()
    assert e <= bound, (name, e, fl)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_vae_decode_small(vae_case, dtype):
    """2 views of 12 x 20 latents -> 96 x 160 images: every layer type of the decoder incl. the 512-wide
    single-head attention (240 tokens, not a multiple of the GEMM tiles) and the three fused upsamplings."""
    ora, sd = vae_case
    z = bf16_round(seeded_tensor((2, 4, 12, 20), 5, 3.0))
    with torch.no_grad():
        ref = ora.decode(z)
        with storage_emulation(ora, dtype):
            emul = ora.decode(z)
    y = _hip(sd, dtype).decode(z.cuda())
    assert y.shape == (2, 3, 96, 160) and y.dtype == dtype
    _check("vae decode 2x12x20", y, ref, emul, dtype)


def test_vae_decode_full_size_view(vae_case):
    """One view at the workload's size (28 x 50 latents -> 224 x 400 image, 1400 attention tokens)."""
    ora, sd = vae_case
    dtype = torch.bfloat16
    z = bf16_round(seeded_tensor((1, 4, 28, 50), 6, 3.0))
    with torch.no_grad():
        ref = ora.decode(z)
        with storage_emulation(ora, dtype):
            emul = ora.decode(z)
    y = _hip(sd, dtype).decode(z.cuda())
    assert y.shape == (1, 3, 224, 400)
    _check("vae decode 1x28x50", y, ref, emul, dtype)


def test_decode_latents_six_views(vae_case):
    """`decode_latents` on a whole scene (1, 6, 4, 28, 50): scaling, per-view decode, [0, 1] range; views
    are decoded independently, so a view decoded alone must agree with its slice of the batch."""
    from dualdiff_amd.networks.vae_decoder import decode_latents
    ora, sd = vae_case
    dtype = torch.bfloat16
    net = _hip(sd, dtype)
    lat = bf16_round(seeded_tensor((1, 6, 4, 28, 50), 7, 0.5)).cuda()
    img = decode_latents(net, lat)
    assert img.shape == (1, 6, 3, 224, 400) and img.dtype == torch.float32
    assert torch.isfinite(img).all() and img.min().item() >= 0.0 and img.max().item() <= 1.0
    assert img.std().item() > 1e-3
    one = decode_latents(net, lat[:, 2:3])
    assert rel_l2(one[0, 0], img[0, 2].cpu()) < 2e-2
    # and against the oracle's decode_latents for that view
    with torch.no_grad():
        ref = V.decode_latents(ora, lat[:, 2:3].float().cpu())
        with storage_emulation(ora, dtype):
            emul = V.decode_latents(ora, lat[:, 2:3].float().cpu())
    _check("decode_latents view 2", one, ref, emul, dtype)


def test_softmax_rows_and_f32_gemm(gpu):
    from dualdiff_amd import ops as O
    g = torch.Generator().manual_seed(1)
    q = torch.randn((203, 512), generator=g).to(torch.bfloat16).cuda()
    k = torch.randn((208, 512), generator=g).to(torch.bfloat16).cuda()       # N must be a multiple of 8
    s = O.gemm(q, k, alpha=512 ** -0.5, out_f32=True)
    ref = (q.float().cpu() @ k.float().cpu().T) * 512 ** -0.5
    assert s.dtype == torch.float32 and s.shape == (203, 208) and rel_l2(s, ref) < 1e-5
    s = s[:, :203]                                                           # ragged row length, ld = 208
    p = O.softmax_rows(s, torch.bfloat16)
    assert p.shape == (203, 208) and (p[:, 203:] == 0).all()
    pref = torch.softmax(s.cpu(), dim=-1)
    assert (p[:, :203].float().cpu() - pref).abs().max().item() <= 2.0 ** -8 * pref.max().item() + 1e-6
