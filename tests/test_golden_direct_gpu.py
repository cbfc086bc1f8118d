"""HIP modules against the GOLDEN FIXTURES directly — no oracle in the proof (VERDICT r3 missing #3).

The level-1 fixtures under tests/golden/ are outputs of the reference's OWN source files
(tests/golden/mint.py executes txt_con_fusion.py:74-181,242-337, map_embedder.py:10-77,114-138,
bbox_embedder.py:164-203, box_adapter.py:66-175,177-411 on the seeded inputs of tests/golden/cases.py).
Here the HIP modules get the SAME name-keyed seeded weights (`seeded_state_dict` is keyed by the
state-dict name, which the drop-in surface shares with the reference) and the SAME inputs, and their
output is compared with the stored reference output: the full tensor where the fixture stores it,
else the strided sub-sample plus the float64 checksums of the whole tensor.  The only helpers taken
from `oracle/` are the seeded generators (`init_utils`); no oracle forward runs in this file.

Bound.  The fixtures are fp32 results of fp32 weights; the HIP path stores weights, inputs and every
inter-kernel tensor in 16 bit.  One rounding to a p-bit significand has relative rms error
2^-p / sqrt(3) (fp16: 2.8e-4, bf16: 2.3e-3); `depth` independent roundings on the path to an output
element add in quadrature, so   rel-L2 <= SLACK * 2^-p / sqrt(3) * sqrt(depth)   with the depth
counted per case below (weights + inputs count as one rounding per GEMM operand).  SLACK = 1.5
covers the spread of one realisation of the noise; measured values are printed and logged to the
parity CSV.
"""
import math
import os

import numpy as np
import pytest
import torch

from oracle.init_utils import seeded_state_dict
from tests.golden import cases as C
from tests.parity_util import log_row

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden")
DTYPES = [torch.float16, torch.bfloat16]
SLACK = 1.5


def gold(name):
    return np.load(os.path.join(GOLD, name + ".npz"))


def unit_noise(dtype):
    return 2.0 ** -(11 if dtype == torch.float16 else 8) / math.sqrt(3.0)


def check(name, npz, key, out, dtype, depth):
    """rel-L2 of `out` against fixture entry `key` over the stored elements (+ whole-tensor checksums)."""
    t = out.detach().float().cpu()
    assert torch.isfinite(t).all(), name
    bnd = SLACK * unit_noise(dtype) * math.sqrt(depth)
    if key in npz.files:
        ref = torch.from_numpy(npz[key])
        assert tuple(ref.shape) == tuple(t.shape), (name, ref.shape, t.shape)
        got = t
    else:
        assert tuple(npz[key + "__shape"]) == tuple(t.shape), (name, npz[key + "__shape"], t.shape)
        ref = torch.from_numpy(npz[key + "__sub"])
        got = t.reshape(-1)[::int(npz[key + "__stride"])]
        # checksums of the WHOLE tensor: the elements the sub-sample skips are covered too
        asum, rsum = float(npz[key + "__abssum"]), float(npz[key + "__sum"])
        assert abs(t.double().abs().sum().item() - asum) <= bnd * asum, (name, "abssum")
        assert abs(t.double().sum().item() - rsum) <= bnd * asum, (name, "sum")
    e = ((got - ref).norm() / ref.norm()).item()
    print("%-44s %-8s rel-L2 vs reference fixture %.3e  bound %.3e (depth %d)"
          % (name, str(dtype).split(".")[-1], e, bnd, depth))
    log_row("golden-direct " + name, dtype, e, float("nan"), bnd)
    assert e <= bnd, (name, e, bnd)
    return e


def seeded(module, seed, dtype):
    module.load_state_dict(seeded_state_dict(module, seed), strict=True)
    return module.to("cuda", dtype).eval()


@torch.no_grad()
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("fused", [True, False], ids=["xattn320", "three_launches"])
@pytest.mark.parametrize("name", ["sfa", "sfa_plus"])
def test_sfa_vs_reference_fixture(gpu, name, fused, dtype, monkeypatch):
    """txt_con_fusion.py:74-181 (SFA) and :242-337 (SFA+) on (6, 320, 28, 50) x (6, 77, 768).
    depth: x, W_q | e, W_kv (2) -> q, k, v stored (1) -> attention out (1) -> W_o (1) -> + x, out stored (1);
    SFA+ adds a second attention stage (2)."""
    from dualdiff_amd.networks import layers
    from dualdiff_amd.networks.txt_con_fusion import txt_con_XFormersAttn, txt_con_XFormersAttn_plus
    if name == "sfa_plus" and not fused:
        pytest.skip("SFA+ has one form")
    monkeypatch.setattr(layers, "XATTN_FUSED", fused)
    net = seeded((txt_con_XFormersAttn_plus if name == "sfa_plus" else txt_con_XFormersAttn)(), C.SEED_SFA, dtype)
    x, e = C.sfa_inputs()
    out = net(attn=None, hidden_states=x.cuda().to(dtype), encoder_hidden_states=e.cuda().to(dtype))
    check("%s (%s)" % (name, "fused" if fused else "3 launches"), gold(name), "out", out, dtype,
          depth=8 if name == "sfa_plus" else 6)


@torch.no_grad()
@pytest.mark.parametrize("dtype", DTYPES)
def test_cond_embedder_vs_reference_fixture(gpu, dtype):
    """map_embedder.py:79-138: (1, 3, 224, 2400) panorama -> (6, 320, 28, 50); 8 convs, SiLU after each but the last.
    depth: input + 8 x (weights, stored output) = 17."""
    from dualdiff_amd.networks.map_embedder import ControlNetConditioningEmbedding
    net = seeded(ControlNetConditioningEmbedding(320, block_out_channels=(16, 32, 96, 256)), C.SEED_EMB, dtype)
    out = net(C.cond_image().cuda().to(dtype))
    check("cond_embedder", gold("cond_embedder"), "out", out, dtype, depth=17)


@torch.no_grad()
@pytest.mark.parametrize("dtype", DTYPES)
def test_bev_map_embedder_vs_reference_fixture(gpu, dtype):
    """map_embedder.py:10-77: (1, 25, 200, 200) BEV map -> (6, 320, 28, 50), 8 convs (binary input: exact)."""
    from dualdiff_amd.networks.map_embedder import BEVControlNetConditioningEmbedding
    net = seeded(BEVControlNetConditioningEmbedding(), C.SEED_BEV_EMB, dtype)
    out = net(C.bev_map().cuda().to(dtype))
    check("bev_map_embedder", gold("bev_map_embedder"), "out", out, dtype, depth=16)


@torch.no_grad()
@pytest.mark.parametrize("dtype", DTYPES)
def test_bbox_embedder_vs_reference_fixture(gpu, dtype):
    """bbox_embedder.py:164-203: Fourier(8 x 3 corners) -> Linear -> SiLU -> cat class token -> 3-layer MLP with
    null-masking.  The corner coordinates stay fp32 (|x| <= 50 times frequencies up to 8: the PHASE must not be rounded
    to 16 bit; the module computes the Fourier features in the coordinates' dtype and casts the features).
    depth: features, 4 x (weights, stored output), class token = 10."""
    from dualdiff_amd.networks.bbox_embedder import ContinuousBBoxWithTextEmbedding
    net = seeded(ContinuousBBoxWithTextEmbedding(n_classes=10, mode="all-xyz", minmax_normalize=False,
                                                 use_text_encoder_init=False), C.SEED_BOX, dtype)
    bb, cl, mk = C.box_inputs()
    out = net(bb.cuda(), cl.cuda(), mk.cuda())
    assert out.dtype == dtype
    check("bbox_embedder", gold("bbox_embedder"), "out", out, dtype, depth=10)


@torch.no_grad()
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("proc", ["builtin", "reference_named"])
def test_attn_processor_vs_reference_fixture(gpu, proc, dtype):
    """box_adapter.py:66-175 `XFormersAttnProcessor._real_call` on a (3, 140, 320) x (3, 13, 768) cross-attention and on
    the self-attention of the same width — through the built-in processor and through the class that carries the
    reference's name.  depth: x, W_q | ctx, W_kv (2) -> q/k/v (1) -> attention out (1) -> W_o (1) -> out (1)."""
    from dualdiff_amd.networks.box_adapter import XFormersAttnProcessor
    from dualdiff_amd.networks.layers import Attention
    g = gold("attn_processor")
    h, ctx = C.proc_inputs()
    a = seeded(Attention(320, 768, 8, 40), C.SEED_PROC, dtype)
    s = seeded(Attention(320, None, 8, 40), C.SEED_PROC + 1, dtype)
    if proc == "reference_named":
        a.set_processor(XFormersAttnProcessor())
        s.set_processor(XFormersAttnProcessor())
    hd = h.cuda().to(dtype)
    check("attn_processor cross (%s)" % proc, g, "out", a(hd, encoder_hidden_states=ctx.cuda().to(dtype)), dtype, depth=6)
    check("attn_processor self (%s)" % proc, g, "out_self", s(hd), dtype, depth=6)


@torch.no_grad()
@pytest.mark.parametrize("dtype", DTYPES)
def test_adapter_processor_vs_reference_fixture(gpu, dtype):
    """box_adapter.py:177-411 `Adapter_XFormersAttnProcessor._real_call`: text + box + class K/V banks, scale 0.7,
    context = 13 text(+cam) | 5 box | 5 class tokens.  depth: as the plain processor + the two extra attention sums (2)."""
    from dualdiff_amd.networks.box_adapter import Adapter_XFormersAttnProcessor
    from dualdiff_amd.networks.layers import Attention
    a = Attention(320, 768, 8, 40)
    a.load_state_dict(seeded_state_dict(a, C.SEED_PROC), strict=True)
    p = Adapter_XFormersAttnProcessor(320, 768, scale=0.7)
    p.load_state_dict(seeded_state_dict(p, C.SEED_ADAPTER), strict=True)
    a.set_processor(p)
    a = a.to("cuda", dtype).eval()
    h, ctx, nt = C.adapter_inputs()
    p.num_tokens = nt
    out = a(h.cuda().to(dtype), encoder_hidden_states=ctx.cuda().to(dtype))
    check("adapter_processor", gold("adapter_processor"), "out", out, dtype, depth=8)
