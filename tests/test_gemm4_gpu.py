"""dd_gemm4_kernel — the pipelined dense family as a PERSISTENT walk over tiles (round 6).

Launched instead of dd_gemm3_kernel when a pipelined tile's grid exceeds one residency generation.  Its K loop performs
the same multiply-accumulates in the same order per accumulator as the LDS-DMA family (dd_gemm2_kernel) and its epilogues
the same arithmetic as store_tile / store_tile_ln, so every result must equal, BIT FOR BIT, the one of the same-shaped
dd_gemm2 tile — with two exceptions of ONE ULP on < 0.1 % of the elements, where the compiler contracts the last multiply /
multiply-add of the epilogue with the conversion to the storage type differently in the two kernels (fp16: a mixed-precision
fma rounds once where mul + convert round twice): the softmax-scaled head-major planes and the LayerNorm second output: 96x64 (72 / 73 vs 52), 192x128 (75 vs 44, plain and GEGLU), 160x160 (78 vs 28), 80x320 with the
LayerNorm-emitting epilogue (74 vs 40).  Shapes: several generations of tiles with ragged row / column tails, K = 320 /
640 (the short loops the persistent walk is for) and a two-source operand (the up path's concat, seam inside the loop);
epilogue operands in every combination the fixed-count epilogues cover (bias, alpha, residual, SiLU, accumulate,
head-major planes with the softmax scale).  Against torch in fp32 as well (tolerance: one rounding of the output)."""
import ctypes

import pytest
import torch

from dualdiff_amd import _native
from dualdiff_amd import ops as O

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def _same_but_for_rare_ulps(a, b, dtype, frac=1e-3):
    """Equal, or different on at most `frac` of the elements by at most one unit in the last place AT THE MAGNITUDE OF THE
    TERMS involved (gamma * x_hat + beta may cancel: an fma and a mul + add then differ by an ulp of the product, not of
    the small result), bounded here by one ulp of max(1, |value|) doubled."""
    if torch.equal(a, b):
        return True
    ne = a != b
    ulp = 2.0 ** (-10 if dtype == torch.float16 else -7)
    d = (a.float() - b.float()).abs()[ne]
    scale = torch.maximum(a.float().abs(), b.float().abs())[ne].clamp_min(1.0)
    return ne.float().mean().item() <= frac and (d / scale).max().item() <= 2.0 * ulp


def _name(rows, n, k, tile, dtype, geglu=False, ln_out=False):
    d = _native.GemmDesc()
    d.a = d.w = d.out = 4096
    d.rows, d.n, d.k, d.k1 = rows, n, k, k
    d.lda, d.ldc = k, n
    d.alpha, d.dtype, d.tile = 1.0, 0 if dtype == torch.float16 else 1, tile
    d.epilogue = 1 if geglu else 0
    if ln_out:
        d.ln_out = d.lno_gamma = d.lno_beta = 4096
        d.ld_ln_out = n
    return _native.load().dd_gemm_kernel_name(ctypes.byref(d)).decode()


def _mk(rows, n, k, dtype, seed, wrows=None):
    g = torch.Generator(device=DEV).manual_seed(seed)
    x = torch.randn(rows, k, device=DEV, generator=g).to(dtype)
    w = (torch.randn(wrows or n, k, device=DEV, generator=g) * k ** -0.5).to(dtype)
    return x, w, g


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("tile,twin,rows,n,k", [
    (72, 52, 67200, 320, 320),        # 700 x 5 tiles of 96 x 64 on 512 resident workgroups
    (72, 52, 50001, 328, 640),        # ragged rows and columns
    (73, 52, 67200, 320, 320),        # 5-slot ring: one workgroup per CU, D = 4
    (75, 44, 67200, 960, 320),        # fused Q|K|V shape at 48 instances
    (75, 44, 50001, 328, 640),
    (78, 28, 67200, 320, 320),        # 10 waves: operand loads inside the epilogue
    (78, 28, 90003, 328, 320),
])
def test_persistent_tiles_equal_their_lds_dma_twins_bit_for_bit(gpu, dtype, tile, twin, rows, n, k):
    assert _name(rows, n, k, tile, dtype).startswith("dd_gemm4_kernel<")        # the persistent form takes these grids
    assert _name(rows, n, k, twin, dtype).startswith("dd_gemm2_kernel<")
    x, w, g = _mk(rows, n, k, dtype, 3)
    bias = torch.randn(n, device=DEV, generator=g).to(dtype)
    res = torch.randn(rows, n, device=DEV, generator=g).to(dtype)
    ref32 = x.float() @ w.float().t()
    for kw in ({}, {"bias": bias}, {"bias": bias, "res": res, "alpha": 0.5}, {"res": res, "epilogue": O.DD_EPI_SILU}):
        b = kw.pop("bias", None)
        got = O.gemm(x, w, b, tile=tile, split_k=1, **kw)
        want = O.gemm(x, w, b, tile=twin, split_k=1, **kw)
        assert torch.equal(got, want), (tile, kw.keys())
    got = O.gemm(x, w, bias, tile=tile, split_k=1)
    tol = 2.0 ** (-10 if dtype == torch.float16 else -7)
    err = (got.float() - (ref32 + bias.float())).abs().max().item() / (ref32.abs().max().item() + 1.0)
    assert err < tol, err
    # accumulate (preloaded target: tiles with at most 4 output vectors per lane) — else the launcher keeps dd_gemm3
    base = torch.randn(rows, n, device=DEV, generator=g).to(dtype)
    o1, o2 = base.clone(), base.clone()
    O.gemm(x, w, bias, tile=tile, split_k=1, out=o1, accumulate=True)
    O.gemm(x, w, bias, tile=twin, split_k=1, out=o2, accumulate=True)
    assert torch.equal(o1, o2)
    # head-major planes with the softmax scale on the first planes (fused Q|K|V projection)
    if n % 40 == 0:
        hm = (40, n // 40 // 3 if n // 40 >= 3 else 1, 0.158 * 1.4426950408889634)
        assert _same_but_for_rare_ulps(O.gemm(x, w, None, tile=tile, split_k=1, head_major=hm),
                                       O.gemm(x, w, None, tile=twin, split_k=1, head_major=hm), dtype)
        hm0 = (40, 0, 1.0)                                      # unscaled planes: bit for bit
        assert torch.equal(O.gemm(x, w, bias, tile=tile, split_k=1, head_major=hm0), O.gemm(x, w, bias, tile=twin, split_k=1, head_major=hm0))


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("rows,n,k", [(67200, 2560, 320), (33001, 1296, 640)])
def test_persistent_geglu_tile_equals_its_twin(gpu, dtype, rows, n, k):
    """GEGLU (n = weight rows = 2 x outputs): tile 75 on the persistent walk vs tile 44 of dd_gemm2."""
    assert _name(rows, n // 2, k, 75, dtype, geglu=True).startswith("dd_gemm4_kernel<")
    x, w, g = _mk(rows, n // 2, k, dtype, 5, wrows=n)
    bias = torch.randn(n, device=DEV, generator=g).to(dtype)
    for b in (None, bias):
        got = O.gemm(x, w, b, tile=75, epilogue=O.DD_EPI_GEGLU)
        want = O.gemm(x, w, b, tile=44, epilogue=O.DD_EPI_GEGLU)
        assert got.shape == (rows, n // 2) and torch.equal(got, want)
    h = x.float() @ w.float().t() + bias.float()
    ref = h[:, :n // 2] * torch.nn.functional.gelu(h[:, n // 2:])
    err = (got.float() - ref).abs().max().item() / (ref.abs().max().item() + 1e-6)
    assert err < 2.0 ** (-9 if dtype == torch.float16 else -6), err


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("rows,k,two_source", [(67200, 320, False), (67200, 1280, False), (40003, 640, True)])
def test_persistent_layernorm_tile_equals_its_twin(gpu, dtype, rows, k, two_source):
    """80 x 320 tile: out AND LayerNorm(out); tile 74 (pipelined; persistent beyond 256 row tiles) vs tile 40 (dd_gemm2)."""
    n = 320
    assert _name(rows, n, k, 74, dtype, ln_out=True).startswith("dd_gemm4_kernel<")
    assert _name(20000, n, k, 74, dtype, ln_out=True).startswith("dd_gemm3_kernel<")      # one generation: one tile per workgroup
    x, w, g = _mk(rows, n, k, dtype, 7)
    bias = torch.randn(n, device=DEV, generator=g).to(dtype)
    res = torch.randn(rows, n, device=DEV, generator=g).to(dtype)
    gamma = (1 + 0.1 * torch.randn(n, device=DEV, generator=g)).to(dtype)
    beta = (0.1 * torch.randn(n, device=DEV, generator=g)).to(dtype)
    kw = {}
    if two_source:                                              # [g | h] operand of the folded feed-forward output projection
        kw["a2"] = x[:, k // 2:].contiguous()
        x = x[:, :k // 2].contiguous()
    outs = []
    for tile in (74, 40):
        o = O.gemm(x, w, bias, res=res, ln_out=(gamma, beta, 1e-5), tile=tile, **kw)
        outs.append((o.clone(), o._ln_out.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and _same_but_for_rare_ulps(outs[0][1], outs[1][1], dtype)
    ln = torch.nn.functional.layer_norm(outs[0][0].float(), (n,), gamma.float(), beta.float(), 1e-5)
    assert (outs[0][1].float() - ln).abs().max().item() < (4e-3 if dtype == torch.float16 else 3e-2)


def test_persistent_tiles_random_shapes_and_operands(gpu):
    """Sixteen seeded random problems (rows, columns with ragged tails, K in the short-loop range, either dtype, bias /
    residual / alpha drawn at random) on every persistent tile against its dd_gemm2 twin, bit for bit, each launched twice
    with a large unrelated launch in between (different register / LDS leftovers): the counted waits and the untracked
    operand loads of dd_gemm4_kernel must not depend on what ran before."""
    rng = torch.Generator().manual_seed(606)
    pick = lambda seq: seq[int(torch.randint(len(seq), (1,), generator=rng))]
    junk_a = torch.randn(8192, 2048, device=DEV, dtype=torch.float16)
    for case in range(16):
        dtype = pick([torch.float16, torch.bfloat16])
        tile, twin = pick([(72, 52), (73, 52), (75, 44), (78, 28)])
        rows = int(torch.randint(30000, 90000, (1,), generator=rng))
        n = 8 * int(torch.randint(8, 170, (1,), generator=rng))
        k = pick([320, 640, 960])
        x, w, g = _mk(rows, n, k, dtype, 1000 + case)
        kw = {}
        bias = torch.randn(n, device=DEV, generator=g).to(dtype) if pick([0, 1]) else None
        if pick([0, 1]):
            kw["res"] = torch.randn(rows, n, device=DEV, generator=g).to(dtype)
        if pick([0, 1]):
            kw["alpha"] = 0.5
        want = O.gemm(x, w, bias, tile=twin, split_k=1, **kw)
        for rep in range(2):
            got = O.gemm(x, w, bias, tile=tile, split_k=1, **kw)
            assert torch.equal(got, want), (case, rep, tile, rows, n, k, dtype, sorted(kw), bias is not None)
            (junk_a @ junk_a[:2048]).sum().item()
        ref = x.float() @ w.float().t()
        if bias is not None:
            ref = ref + bias.float()
        ref = ref * kw.get("alpha", 1.0) + (kw["res"].float() if "res" in kw else 0.0)
        err = (got.float() - ref).abs().max().item() / (ref.abs().max().item() + 1.0)
        assert err < 2.0 ** (-10 if dtype == torch.float16 else -7), (case, err)
