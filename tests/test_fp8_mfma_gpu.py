"""EXTENSION (BASELINE configs[4]: "fp8 weights (CDNA4 fp8 MFMA)"; no reference semantics): the W8A8 projection on the fp8
matrix path (csrc/gemm8.hip) — row quantisation behind the LayerNorm, the fp8 x fp8 GEMM, and the attention
projections of a transformer block running on it.  Oracle: torch float8_e4m3fn arithmetic on the CPU (exact products of
the SAME quantised operands, fp32 accumulation)."""
import pytest
import torch

from oracle import leaf_ops as L

pytestmark = pytest.mark.gpu
DTYPES = [torch.float16, torch.bfloat16]


def rnd(shape, dtype, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dtype).cuda()


@pytest.fixture(scope="module")
def ops(gpu):
    from dualdiff_amd import ops as O
    return O


def _rowquant_ref(y):
    """y: fp32 [rows, c] (values already rounded to the storage type) -> (q float8 [rows, c], scale [rows])."""
    amax = y.abs().amax(dim=1)
    scale = torch.where(amax > 0, amax / 448.0, torch.ones_like(amax))
    q = (y * (1.0 / scale)[:, None]).clamp(-448.0, 448.0).to(torch.float8_e4m3fn)
    return q, scale


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("rows,c,ln", [(1400 * 2 + 3, 320, True), (701, 640, True), (91, 1280, True), (350, 640, False),
                                       (5, 320, False), (64, 1280, False)])
def test_rowquant_fp8(ops, dtype, rows, c, ln):
    x = rnd((rows, c), dtype, 1, 3.0)
    x[rows // 2] = 0                                   # an all-zero row: scale 1, zeros
    g_, b_ = (1.0 + 0.1 * rnd((c,), torch.float32, 2)).to(dtype), rnd((c,), dtype, 3, 0.1)
    q, s = ops.rowquant_fp8(x, (g_, b_, 1e-5) if ln else None)
    kp = (c + 127) // 128 * 128
    assert q.shape == (rows, kp) and s.shape == (rows,)
    y = ops.layernorm(x, g_, b_, 1e-5).float().cpu() if ln else x.float().cpu()      # the storage-rounded LayerNorm output
    qr, sr = _rowquant_ref(y)
    assert torch.equal(s.cpu(), sr), "row scales differ"
    got = q.cpu().view(torch.uint8)
    assert torch.equal(got[:, c:], torch.zeros_like(got[:, c:])), "padding columns must be zero"
    # the hardware conversion and torch's cast are both round-to-nearest-even; y / scale is computed as y * (1 / scale)
    # on both sides — identical bytes expected, a last-bit tie-break difference on at most a handful of elements tolerated
    diff = (got[:, :c] != qr.view(torch.uint8)).sum().item()
    assert diff <= max(2, rows * c // 100000), "%d of %d quantised bytes differ" % (diff, rows * c)
    back = q.cpu().float()[:, :c] * s.cpu()[:, None]
    assert (back - y).abs().max().item() <= 0.0725 * y.abs().amax().item() + 1e-6       # e4m3: 3 mantissa bits -> 2^-4 + slack


G8_CASES = [(1092, 3840, 1280), (336, 1280, 1280), (4200, 1920, 640), (4200, 640, 640), (16800, 960, 320), (77, 72, 200),
            (129, 132, 136), (1003, 328, 2048)]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("rows,n,k", G8_CASES)
@pytest.mark.parametrize("epi", ["plain", "bias+res", "head_major"])
def test_gemm8_exact_products(ops, dtype, rows, n, k, epi):
    """fp8 x fp8 products are exact in fp32 and the kernel accumulates in fp32: against the fp32 matmul of the SAME
    quantised operands only the summation order and the final rounding to the storage type differ."""
    if epi == "head_major" and (n % 40 or rows * n > 20e6):
        pytest.skip("head-major planes need n % 40 == 0")
    a = rnd((rows, k), dtype, 1)
    w = rnd((n, k), dtype, 2, k ** -0.5)
    if k <= 1536:
        a8, sa = ops.rowquant_fp8(a.contiguous())
    else:                                   # wider than the row quantiser covers: quantise with torch (same arithmetic)
        q_, sa = _rowquant_ref(a.float().cpu())
        a8, sa = q_.cuda(), sa.cuda()
    w8, sw = ops.quantize_fp8_padded(w)
    bias = rnd((n,), dtype, 3) if epi == "bias+res" else None
    res = rnd((rows, n), dtype, 4) if epi == "bias+res" else None
    hm = (40, n // 40 // 3, 0.25) if epi == "head_major" else None
    y = ops.gemm8(a8, sa, w8, sw, bias, res=res, head_major=hm, dtype=dtype)
    ref = (a8.cpu().float() @ w8.cpu().float().t()) * sa.cpu()[:, None] * sw.cpu()[None, :]
    if bias is not None:
        ref = ref + bias.float().cpu()[None, :] + res.float().cpu()
    if hm is not None:
        ref = ref.reshape(rows, n // 40, 40).permute(1, 0, 2).clone()
        ref[: hm[1]] *= hm[2]
    tol = (2.0 ** -10 if dtype == torch.float16 else 2.0 ** -7) * 1.5
    err = (y.float().cpu() - ref).abs().max().item() / (ref.abs().max().item() + 1e-12)
    print("gemm8 %dx%dx%d %s %s: rel-to-max %.3e" % (rows, n, k, epi, dtype, err))
    assert y.shape == ref.shape and err <= tol


@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm8_vs_16bit_projection(ops, dtype):
    """What the quantisation costs: LayerNorm -> fused Q|K|V projection, W8A8 against the 16-bit path."""
    rows, c = 1092, 1280
    x = rnd((rows, c), dtype, 1)
    g_, b_ = (1.0 + 0.1 * rnd((c,), torch.float32, 2)).to(dtype), rnd((c,), dtype, 3, 0.1)
    w = rnd((3 * c, c), dtype, 4, c ** -0.5)
    y16 = ops.gemm(ops.layernorm(x, g_, b_, 1e-5), w).float().cpu()
    a8, sa = ops.rowquant_fp8(x, (g_, b_, 1e-5))
    w8, sw = ops.quantize_fp8_padded(w)
    y8 = ops.gemm8(a8, sa, w8, sw, dtype=dtype).float().cpu()
    e = ((y8 - y16).norm() / y16.norm()).item()
    print("W8A8 vs 16-bit Q|K|V projection: rel-L2 %.3e" % e)
    assert 1e-3 < e < 6e-2            # two e4m3 operands: ~2^-4 / sqrt(3) per product, averaged over K


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("rows,c", [(4200, 640), (1092, 1280), (333, 640)])
def test_gemm8_geglu(ops, dtype, rows, c):
    """The GEGLU projection on the fp8 matrix path: W8 holds the 8C rows [hidden | gate]; out = h * gelu_erf(g)."""
    x = rnd((rows, c), dtype, 1)
    w = rnd((8 * c, c), dtype, 2, c ** -0.5)
    b = rnd((8 * c,), dtype, 3)
    a8, sa = ops.rowquant_fp8(x)
    w8, sw = ops.quantize_fp8_padded(w)
    y = ops.gemm8(a8, sa, w8, sw, b, dtype=dtype, geglu=True)
    a_deq = a8.cpu().float()[:, :c] * sa.cpu()[:, None]
    w_deq = w8.cpu().float()[:, :c] * sw.cpu()[:, None]
    ref = L.linear_ref(a_deq, w_deq, b, geglu=True)
    err = (y.float().cpu() - ref).abs().max().item() / (ref.abs().max().item() + 1e-12)
    print("gemm8 GEGLU %dx%dx%d %s: rel-to-max %.3e" % (rows, 8 * c, c, dtype, err))
    assert y.shape == (rows, 4 * c) and err <= (2.0 ** -10 if dtype == torch.float16 else 2.0 ** -7) * 2.0


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("dim,n", [(640, 350), (1280, 91)])
def test_block_on_fp8_matrix_path(gpu, dtype, dim, n):
    """A multiview transformer block with enable_fp8_weights(mfma=True): the fused Q|K|V projections of attn1 / attn4 and
    the GEGLU projection run W8A8 on the fp8 MFMA.  Against the same block in 16 bit the output moves by the quantisation
    noise of two e4m3 operands (reported), stays finite, and the kernels really are the fp8 ones."""
    from oracle import dualdiff_restated as R
    from oracle.init_utils import seeded_state_dict, seeded_tensor
    from tests.golden import cases as C
    from dualdiff_amd import ops as O
    from dualdiff_amd.networks.blocks import BasicMultiviewTransformerBlock
    from dualdiff_amd.networks.layers import enable_fp8_weights
    ora = R.BasicMultiviewTransformerBlock(dim, 8, dim // 8, cross_attention_dim=768, neighboring_view_pair=C.VIEW_PAIR)
    sd = {k: C.bf16_round(v) for k, v in seeded_state_dict(ora, 5).items()}
    x = C.bf16_round(seeded_tensor((6, n, dim), 1)).cuda().to(dtype).reshape(-1, dim)
    ctx = C.bf16_round(seeded_tensor((6, 30, 768), 2)).cuda().to(dtype).reshape(-1, 768)
    outs = {}
    for mfma in (False, True):
        blk = BasicMultiviewTransformerBlock(dim, 8, dim // 8, cross_attention_dim=768, neighboring_view_pair=C.VIEW_PAIR)
        blk.load_state_dict(sd)
        blk = blk.to("cuda", dtype)
        if mfma:
            assert enable_fp8_weights(blk, mfma=True) == 3
        timer = O.KernelTimer()
        O.set_timer(timer)
        with torch.no_grad():
            outs[mfma] = blk.run(x, 6, n, ctx, 30).float().cpu()
        O.set_timer(None)
        names = set(timer.summary())
        assert ("dd_gemm8_kernel" in names) == mfma and ("dd_rowquant_fp8_kernel" in names) == mfma
    e = ((outs[True] - outs[False]).norm() / outs[False].norm()).item()
    print("block C=%d on the fp8 matrix path vs 16-bit: rel-L2 %.3e" % (dim, e))
    assert torch.isfinite(outs[True]).all() and 1e-4 < e < 5e-2
