"""Kernel-level parity: every HIP kernel (through the C-ABI) vs the CPU leaf-op oracle.

Tolerances: outputs are compared in fp32 against the fp32 oracle evaluated on the same
fp16/bf16-rounded inputs.  The only rounding the kernel may add is the final store, so
|y - ref| <= TOL[dtype] * max(|ref|) elementwise with TOL = 2^-10 (fp16) / 2^-7 (bf16)
(one ulp of the storage type at the tensor's scale, plus fp32 summation-order noise).
"""
import pytest
import torch

from oracle import leaf_ops as L

pytestmark = pytest.mark.gpu

TOL = {torch.float16: 2.0 ** -10, torch.bfloat16: 2.0 ** -7}
DTYPES = [torch.float16, torch.bfloat16]


def rnd(shape, dtype, seed, scale=1.0, device="cuda"):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dtype).to(device)


def check(y, ref, dtype, what, tol_mult=1.0):
    y = y.detach().float().cpu()
    ref = ref.float()
    assert y.shape == ref.shape, (what, y.shape, ref.shape)
    assert torch.isfinite(y).all(), what + ": non-finite output"
    err = (y - ref).abs().max().item()
    scale = ref.abs().max().item() + 1e-12
    rel = err / scale
    print("%-40s max|err|=%.3e  rel-to-max=%.3e" % (what, err, rel))
    assert rel <= TOL[dtype] * tol_mult, "%s: rel err %.3e > %.3e" % (what, rel, TOL[dtype] * tol_mult)


@pytest.fixture(scope="module")
def ops(gpu):
    from dualdiff_amd import ops as O
    return O


# ------------------------------------------------------------------ MFMA layout sanity ----
def test_gemm_identity_asymmetric(ops):
    """A = I with an asymmetric W catches any row/col swap in the fragment maps."""
    n = 128
    a = torch.eye(n, dtype=torch.float16, device="cuda")
    w = (torch.arange(n * n, dtype=torch.float32).reshape(n, n) % 61 - 30).to(torch.float16).cuda()
    y = ops.gemm(a, w, tile=1)
    check(y, w.float().cpu().t(), torch.float16, "gemm A=I")


ALL_TILES = [1, 2, 3, 4, 5, 11, 12, 13, 14, 15, 16, 20, 23, 24, 27, 28, 44, 46, 52, 59, 60]
DMA_TILES = [11, 12, 13, 14, 15, 16, 20, 23, 24, 27, 28, 44, 46, 52]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("tile", ALL_TILES)
@pytest.mark.parametrize("rows,n,k", [(256, 256, 256), (200, 320, 320), (77, 72, 200), (1400, 320, 2880)])
def test_gemm_tiles(ops, dtype, tile, rows, n, k):
    a = rnd((rows, k), dtype, 1)
    w = rnd((n, k), dtype, 2, 0.05)
    b = rnd((n,), dtype, 3)
    y = ops.gemm(a, w, b, tile=tile)
    check(y, L.linear_ref(a, w, b), dtype, "gemm tile%d %dx%dx%d" % (tile, rows, n, k))


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("rows,k", [(16800, 320), (1400 * 2 + 37, 320), (80, 1600), (7, 64)])
def test_gemm_layernorm_emitting_epilogue(ops, dtype, rows, k):
    """Tile 40 (80 whole rows x 320 columns per workgroup): the epilogue stores out = A W^T + bias + res AND
    LayerNorm(out) (statistics over the stored, rounded values) — both must match GEMM followed by LayerNorm."""
    a = rnd((rows, k), dtype, 1)
    w = rnd((320, k), dtype, 2, k ** -0.5)
    b = rnd((320,), dtype, 3)
    res = rnd((rows, 320), dtype, 4) * 3
    g = rnd((320,), dtype, 5) + 1.0
    be = rnd((320,), dtype, 6)
    y = ops.gemm(a, w, b, res=res, ln_out=(g, be, 1e-5))
    ref = L.linear_ref(a, w, b, res=res)
    check(y, ref, dtype, "ln-emitting gemm: out %dx320x%d" % (rows, k))
    # LayerNorm of the kernel's OWN stored output (isolates the fused normalisation from the GEMM rounding)
    want = L.layernorm_ref(y, g, be)
    check(y._ln_out, want, dtype, "ln-emitting gemm: LayerNorm(out)", 2.0)
    y2 = ops.gemm(a, w, None, ln_out=(g, be, 1e-5))               # proj_in form: no residual, no bias
    check(y2._ln_out, L.layernorm_ref(y2, g, be), dtype, "ln-emitting gemm (plain): LayerNorm(out)", 2.0)
    # the tile is also a plain tile
    check(ops.gemm(a, w, b, res=res, tile=40), ref, dtype, "tile 40 plain")
    # default = tile 74 (pipelined K loop, round 5); the dd_gemm2 form of the same tile gives the same bits
    y40 = ops.gemm(a, w, b, res=res, ln_out=(g, be, 1e-5), tile=40)
    assert torch.equal(y, y40) and torch.equal(y._ln_out, y40._ln_out)


@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_epilogue_full(ops, dtype):
    rows, n, k = 12 * 91, 640, 1280
    a = rnd((rows, 768), dtype, 1)
    a2 = rnd((rows, k - 768), dtype, 11)
    w = rnd((n, k), dtype, 2, 0.03)
    b = rnd((n,), dtype, 3)
    res = rnd((rows, n), dtype, 4)
    rv = rnd((12, n), dtype, 5)
    y = ops.gemm(a, w, b, a2=a2, res=res, rowvec=rv, rows_per_inst=91, alpha=0.5)
    ref = L.linear_ref(a, w, b, a2=a2, res=res, rowvec=rv, rows_per_inst=91, alpha=0.5)
    check(y, ref, dtype, "gemm concat+bias+rowvec+alpha+res")
    # accumulate into an existing output, strided output (column slice of a wider buffer)
    buf = rnd((rows, 2 * n), dtype, 6)
    prev = buf[:, n:].clone()
    ops.gemm(a, w, b, a2=a2, out=buf[:, n:], accumulate=True)
    # accumulate adds the storage-rounded previous value; one extra rounding -> 2x tol
    check(buf[:, n:], L.linear_ref(a, w, b, a2=a2) + prev.float().cpu(), dtype, "gemm accumulate strided", 2.0)
    y = ops.gemm(a, w[:, :768].contiguous(), b, epilogue=ops.DD_EPI_SILU)
    check(y, L.linear_ref(a, w[:, :768], b, silu=True), dtype, "gemm silu")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("rows,c", [(1400, 320), (350, 640), (91 * 3, 1280)])
def test_gemm_geglu(ops, dtype, rows, c):
    a = rnd((rows, c), dtype, 1)
    w = rnd((8 * c, c), dtype, 2, 0.05)
    b = rnd((8 * c,), dtype, 3)
    y = ops.gemm(a, w, b, epilogue=ops.DD_EPI_GEGLU)
    check(y, L.linear_ref(a, w, b, geglu=True), dtype, "geglu %dx%d" % (rows, c), 2.0)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("split", [0, 2, 5])
def test_gemm_split_k(ops, dtype, split):
    rows, n, k = 336, 1280, 2560
    a = rnd((rows, k), dtype, 1)
    w = rnd((n, k), dtype, 2, 0.02)
    b = rnd((n,), dtype, 3)
    res = rnd((rows, n), dtype, 4)
    y = ops.gemm(a, w, b, res=res, split_k=split)
    check(y, L.linear_ref(a, w, b, res=res), dtype, "gemm split_k=%d" % split)


# ------------------------------------------------------------------------------ conv ----
CONV_CASES = [
    # m, hin, win, cin, cout, stride, up_size
    (2, 28, 50, 320, 320, 1, None),
    (2, 28, 50, 320, 320, 2, None),
    (3, 14, 25, 640, 640, 2, None),
    (3, 7, 13, 1280, 1280, 2, None),
    (2, 4, 7, 1280, 1280, 1, (7, 13)),
    (2, 7, 13, 1280, 1280, 1, (14, 25)),
    (1, 14, 25, 640, 640, 1, (28, 50)),
    (2, 28, 50, 960, 320, 1, None),
    (12, 4, 7, 2560, 1280, 1, None),        # deep level -> split-K
    (2, 28, 50, 8, 320, 1, None),           # conv_in, latent channels padded 4 -> 8
    (1, 56, 100, 16, 32, 2, None),          # condition embedder style convs
    (1, 28, 50, 96, 256, 2, None),
]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", CONV_CASES, ids=[str(c) for c in CONV_CASES])
def test_conv3x3(ops, dtype, case):
    m, hin, win, cin, cout, stride, up = case
    x = rnd((m * hin * win, cin), dtype, 1)
    w = rnd((cout, cin, 3, 3), dtype, 2, (9 * cin) ** -0.5)
    b = rnd((cout,), dtype, 3)
    y = ops.conv3x3(x, L.pack_conv_weight(w), b, m, hin, win, stride=stride, up_size=up)
    ref = L.conv3x3_ref(x, w, b, m, hin, win, stride=stride, up_size=up)
    check(y, ref, dtype, "conv %s" % (case,))


@pytest.mark.parametrize("tile", DMA_TILES)
@pytest.mark.parametrize("case", [CONV_CASES[1], CONV_CASES[4], CONV_CASES[7], CONV_CASES[9], CONV_CASES[11]],
                         ids=lambda c: str(c))
def test_conv3x3_dma_tiles(ops, tile, case):
    """LDS-DMA kernel family on the awkward cases: stride 2, folded upsample, wide Cin, padded Cin=8,
    K tail (Cin=96 -> K=864), for every tile / stage configuration, incl. split-K."""
    dtype = torch.float16
    m, hin, win, cin, cout, stride, up = case
    x = rnd((m * hin * win, cin), dtype, 1)
    w = rnd((cout, cin, 3, 3), dtype, 2, (9 * cin) ** -0.5)
    b = rnd((cout,), dtype, 3)
    ref = L.conv3x3_ref(x, w, b, m, hin, win, stride=stride, up_size=up)
    for split in (1, 3):
        if split > 1 and 9 * cin < 64 * 4 * split:
            continue
        y = ops.conv3x3(x, L.pack_conv_weight(w), b, m, hin, win, stride=stride, up_size=up, tile=tile, split_k=split)
        check(y, ref, dtype, "conv dma tile%d split%d %s" % (tile, split, case))


CONV3S_TILES = [31, 34, 35, 37]
# (m, h, w, cin, cout): 4x7 / 7x13 / 14x25 levels, ragged instance counts (partial last tile), one
# instance per tile, Cout not a multiple of the tile
CONV3S_CASES = [(12, 4, 7, 1280, 1280), (5, 4, 7, 128, 192), (12, 7, 13, 640, 1280), (7, 7, 13, 192, 64),
                (3, 14, 25, 640, 640), (1, 4, 7, 64, 72), (13, 2, 3, 64, 64)]


@pytest.mark.parametrize("tile", CONV3S_TILES)
@pytest.mark.parametrize("case", CONV3S_CASES, ids=lambda c: str(c))
@pytest.mark.parametrize("dtype", DTYPES)
def test_conv3x3_small_image_direct(ops, tile, case, dtype):
    """Direct small-image conv family (dd_conv3s_kernel): whole instances per workgroup, taps as LDS
    row gathers; with the ResnetBlock2D epilogue and split-K over channel chunks."""
    m, h, w_, cin, cout = case
    if h * w_ > {31: 384, 35: 128}.get(tile, 192):
        pytest.skip("image larger than the tile")
    x = rnd((m * h * w_, cin), dtype, 1)
    w = rnd((cout, cin, 3, 3), dtype, 2, (9 * cin) ** -0.5)
    b = rnd((cout,), dtype, 3)
    temb = rnd((m, cout), dtype, 4)
    res = rnd((m * h * w_, cout), dtype, 5)
    ref = L.conv3x3_ref(x, w, b, m, h, w_) + temb.float().cpu().repeat_interleave(h * w_, 0) + res.float().cpu()
    for split in (1, 2, 5):
        if split > cin // 64:
            continue
        y = ops.conv3x3(x, L.pack_conv_weight(w), b, m, h, w_, rowvec=temb, res=res, tile=tile, split_k=split)
        check(y, ref, dtype, "conv3s tile%d split%d %s" % (tile, split, case))


def test_conv3x3_small_image_direct_rejects(ops):
    """Shapes outside the family's contract are refused (the autotuner skips them), never mis-run."""
    dtype = torch.bfloat16
    x = rnd((2 * 28 * 50, 320), dtype, 1)
    w = rnd((320, 320, 3, 3), dtype, 2, 0.02)
    with pytest.raises(RuntimeError):
        ops.conv3x3(x, L.pack_conv_weight(w), None, 2, 28, 50, tile=31)          # 1400 pixels > 384
    x = rnd((2 * 14 * 25, 640), dtype, 1)
    w = rnd((640, 640, 3, 3), dtype, 2, 0.02)
    with pytest.raises(RuntimeError):
        ops.conv3x3(x, L.pack_conv_weight(w), None, 2, 14, 25, stride=2, tile=31)  # stride 2


def _ln_fold(w, b, gamma, beta, dtype):
    """W' = W * gamma (rounded to the storage dtype), colsum of the rounded W', b' = W beta + b."""
    wf = w.float()
    wp = (wf * gamma.float()[None, :]).to(dtype).contiguous()
    lnb = wf @ beta.float() + (b.float() if b is not None else 0)
    return wp, wp.float().sum(1).contiguous(), lnb.contiguous()


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("rows,c,n,tile", [(700, 320, 960, 0), (16800, 320, 320, 13), (350, 640, 1920, 11),
                                          (91, 1280, 3840, 0), (1092, 1280, 1280, 15), (37, 640, 640, 15),
                                          (4200, 640, 640, 23), (336, 1280, 1280, 24), (129, 320, 64, 13)])
def test_gemm_layernorm_fold(ops, dtype, rows, c, n, tile):
    """LayerNorm folded into the consumer GEMM (row statistics in the kernel prologue) against
    LayerNorm -> Linear of the reference; input with a non-zero row mean and per-row scale."""
    x = (rnd((rows, c), dtype, 1).float() * (0.5 + rnd((rows, 1), torch.float32, 7).abs()) + 0.7).to(dtype)
    w = rnd((n, c), dtype, 2, c ** -0.5)
    b = rnd((n,), dtype, 3)
    gamma = (1.0 + 0.2 * rnd((c,), torch.float32, 4)).to(dtype)
    beta = (0.1 * rnd((c,), torch.float32, 5)).to(dtype)
    res = rnd((rows, n), dtype, 6)
    wp, colsum, lnb = _ln_fold(w, b, gamma, beta, dtype)
    y = ops.gemm(x, wp, None, res=res, ln=(colsum, lnb, 1e-5), tile=tile)
    ref = L.linear_ref(L.layernorm_ref(x, gamma, beta, 1e-5), w, b, res=res)
    check(y, ref, dtype, "gemm + LN fold %dx%dx%d tile%d" % (rows, n, c, tile), 2.0)


@pytest.mark.parametrize("tile", [11, 14, 20])
def test_gemm_layernorm_fold_geglu(ops, tile):
    dtype = torch.bfloat16
    rows, c = 700, 640
    x = (rnd((rows, c), dtype, 1).float() * 1.5 - 0.4).to(dtype)
    w = rnd((8 * c, c), dtype, 2, c ** -0.5)
    b = rnd((8 * c,), dtype, 3)
    gamma = (1.0 + 0.2 * rnd((c,), torch.float32, 4)).to(dtype)
    beta = (0.1 * rnd((c,), torch.float32, 5)).to(dtype)
    wp, colsum, lnb = _ln_fold(w, b, gamma, beta, dtype)
    y = ops.gemm(x, wp, None, ln=(colsum, lnb, 1e-5), epilogue=ops.DD_EPI_GEGLU, tile=tile)
    ref = L.linear_ref(L.layernorm_ref(x, gamma, beta, 1e-5), w, b, geglu=True)
    check(y, ref, dtype, "geglu + LN fold tile%d" % tile, 3.0)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("rows,c,tile", [(16800, 320, 0), (4200, 640, 13), (1092, 1280, 15), (336, 1280, 11),
                                         (700, 320, 26), (129, 640, 14), (37, 320, 1), (350, 640, 4)])
def test_gemm_ln_stats_producer_consumer(ops, dtype, rows, c, tile):
    """The epilogue of a C x C projection (+ residual) leaves per-row partial sums of what it stores; the
    next GEMM evaluates LayerNorm(h) W^T + b from them (`ln_stats` -> `_ln_stats` -> `ln=`) — against
    Linear -> + res -> LayerNorm -> Linear of the reference."""
    a = rnd((rows, c), dtype, 1)
    w1 = rnd((c, c), dtype, 2, c ** -0.5)
    b1 = rnd((c,), dtype, 3)
    res = (rnd((rows, c), dtype, 4).float() * (0.5 + rnd((rows, 1), torch.float32, 7).abs()) + 0.6).to(dtype)
    h = ops.gemm(a, w1, b1, res=res, ln_stats=True, tile=tile)
    st = h._ln_stats
    assert st.shape == (rows, c // 32, 2) and st.dtype == torch.float32
    hf = h.float().cpu().reshape(rows, c // 32, 32)
    # statistics are taken from the fp32 values before the store rounding: compare at storage precision
    tol = {torch.float16: 2e-3, torch.bfloat16: 1.6e-2}[dtype]
    s_ref, q_ref = hf.sum(-1), (hf * hf).sum(-1)
    assert (st[..., 0].cpu() - s_ref).abs().max().item() <= tol * hf.abs().sum(-1).max().item()
    assert (st[..., 1].cpu() - q_ref).abs().max().item() <= 2 * tol * q_ref.max().item()
    # consumer
    n = 3 * c
    w2 = rnd((n, c), dtype, 5, c ** -0.5)
    b2 = rnd((n,), dtype, 6)
    gamma = (1.0 + 0.2 * rnd((c,), torch.float32, 8)).to(dtype)
    beta = (0.1 * rnd((c,), torch.float32, 9)).to(dtype)
    wp, colsum, lnb = _ln_fold(w2, b2, gamma, beta, dtype)
    y = ops.gemm(h, wp, None, ln=(colsum, lnb, 1e-5))
    ref = L.linear_ref(L.layernorm_ref(h, gamma, beta, 1e-5), w2, b2)
    check(y, ref, dtype, "LN stats producer->consumer %dx%d tile%d" % (rows, c, tile), 2.0)
    # the in-kernel statistics of the same input give the same result up to the statistics' rounding
    h2 = h.clone()
    y2 = ops.gemm(h2, wp, None, ln=(colsum, lnb, 1e-5))
    check(y2, ref, dtype, "LN fold in-kernel stats", 2.0)


def test_gemm_ln_stats_rejects(ops):
    dtype = torch.bfloat16
    a = rnd((64, 320), dtype, 1)
    with pytest.raises(ValueError):
        ops.gemm(a, rnd((48, 320), dtype, 2), None, ln_stats=True)            # n % 32 != 0
    with pytest.raises(ValueError):
        ops.gemm(a, rnd((640, 320), dtype, 2), None, ln_stats=True, epilogue=ops.DD_EPI_GEGLU)


def test_gemm_layernorm_fold_rejects(ops):
    dtype = torch.bfloat16
    x = rnd((64, 512), dtype, 1)
    w = rnd((64, 512), dtype, 2)
    z = torch.zeros(64, device="cuda")
    with pytest.raises(RuntimeError):
        ops.gemm(x, w, None, ln=(z, z, 1e-5))                 # K outside {320, 640, 1280}
    x = rnd((64, 320), dtype, 1)
    w = rnd((64, 320), dtype, 2)
    with pytest.raises(RuntimeError):
        ops.gemm(x, w, None, ln=(z, z, 1e-5), tile=1)         # register-staged family has no fold


@pytest.mark.parametrize("tile", [11, 12, 14, 16, 20, 24, 44, 46])
def test_gemm_geglu_dma_tiles(ops, tile):
    dtype = torch.bfloat16
    rows, c = 700, 640
    a = rnd((rows, c), dtype, 1)
    w = rnd((8 * c, c), dtype, 2, 0.05)
    b = rnd((8 * c,), dtype, 3)
    y = ops.gemm(a, w, b, epilogue=ops.DD_EPI_GEGLU, tile=tile)
    check(y, L.linear_ref(a, w, b, geglu=True), dtype, "geglu dma tile%d" % tile, 2.0)


@pytest.mark.parametrize("tile", DMA_TILES + [59, 60])          # 59 / 60 (32-row tiles) have no conv form
def test_gemm_concat_dma_tiles(ops, tile):
    """Two-source A operand (up-path shortcut on [h, skip]) + full epilogue on the DMA family."""
    dtype = torch.float16
    rows, n, k = 12 * 91, 640, 1280
    a = rnd((rows, 768), dtype, 1)
    a2 = rnd((rows, k - 768), dtype, 11)
    w = rnd((n, k), dtype, 2, 0.03)
    b = rnd((n,), dtype, 3)
    res = rnd((rows, n), dtype, 4)
    rv = rnd((12, n), dtype, 5)
    y = ops.gemm(a, w, b, a2=a2, res=res, rowvec=rv, rows_per_inst=91, alpha=0.5, tile=tile)
    ref = L.linear_ref(a, w, b, a2=a2, res=res, rowvec=rv, rows_per_inst=91, alpha=0.5)
    check(y, ref, dtype, "gemm concat dma tile%d" % tile)


@pytest.mark.parametrize("dtype", DTYPES)
def test_conv3x3_resnet_epilogue(ops, dtype):
    """conv + bias + per-instance time-embedding vector + residual (ResnetBlock2D conv1/conv2)."""
    m, h, w_, cin, cout = 3, 14, 25, 640, 640
    x = rnd((m * h * w_, cin), dtype, 1)
    w = rnd((cout, cin, 3, 3), dtype, 2, (9 * cin) ** -0.5)
    b = rnd((cout,), dtype, 3)
    temb = rnd((m, cout), dtype, 4)
    res = rnd((m * h * w_, cout), dtype, 5)
    y = ops.conv3x3(x, L.pack_conv_weight(w), b, m, h, w_, rowvec=temb, res=res)
    ref = L.conv3x3_ref(x, w, b, m, h, w_) + temb.float().cpu().repeat_interleave(h * w_, 0) + res.float().cpu()
    check(y, ref, dtype, "conv resnet epilogue")


@pytest.mark.parametrize("dtype", DTYPES)
def test_conv_out_small(ops, dtype):
    m, h, w_, cin, cout = 2, 28, 50, 320, 4
    x = rnd((m * h * w_, cin), dtype, 1)
    w = rnd((cout, cin, 3, 3), dtype, 2, (9 * cin) ** -0.5)
    b = rnd((cout,), dtype, 3)
    y = ops.conv3x3_small_cout(x, L.pack_conv_weight(w), b, m, h, w_)
    ref = L.conv3x3_ref(x, w, b, m, h, w_).reshape(m, h, w_, cout).permute(0, 3, 1, 2)
    check(y, ref, dtype, "conv_out 320->4 (NCHW out)")


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.bfloat16])
@pytest.mark.parametrize("shape,nf,inc", [((12, 6, 7, 3), 4, True), ((60, 8, 3), 4, True), ((5, 3), 8, False)])
def test_fourier_embed(ops, dtype, shape, nf, inc):
    """dd_fourier_embed vs the Embedder of the reference (networks/embedder.py:18-67): block order
    [x | sin f0 | cos f0 | ...], arithmetic in fp32, camera / box coordinate magnitudes."""
    x = (rnd(shape, torch.float32, 1) * 30.0).to(dtype)
    freqs = [2.0 ** i for i in range(nf)]
    y = ops.fourier_embed(x, freqs, inc)
    xf = x.float().cpu()
    outs = [xf] if inc else []
    for f in freqs:
        outs += [torch.sin(xf * f), torch.cos(xf * f)]
    ref = torch.cat(outs, dim=-1)
    assert y.shape == ref.shape and y.dtype == dtype
    tol = {torch.float32: 2e-5, torch.float16: 1e-3, torch.bfloat16: 8e-3}[dtype]
    scale = ref.abs().max().item()
    assert (y.float().cpu() - ref).abs().max().item() <= tol * max(scale, 1.0)


# ----------------------------------------------------------------------------- norms ----
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("m,hw,c1,c2,silu,eps", [
    (3, 1400, 320, 0, True, 1e-5), (2, 350, 640, 0, False, 1e-6), (2, 91, 1280, 0, True, 1e-5),
    (2, 1400, 640, 320, True, 1e-5), (2, 350, 1280, 640, True, 1e-5), (3, 28, 1280, 1280, True, 1e-5),
    (2, 91, 1280, 640, True, 1e-5), (1, 1400, 320, 320, True, 1e-5),
    # VAE decoder sizes: 4 channels per group (a 16-B vector holds two whole groups), large images
    (2, 240, 128, 0, True, 1e-6), (1, 89600, 128, 0, True, 1e-6), (2, 22400, 256, 0, True, 1e-6),
    (2, 5600, 512, 0, False, 1e-6)])
def test_groupnorm(ops, dtype, m, hw, c1, c2, silu, eps):
    x1 = rnd((m * hw, c1), dtype, 1) + 0.5
    x2 = rnd((m * hw, c2), dtype, 2, 2.0) if c2 else None
    g = rnd((c1 + c2,), dtype, 3) + 1.0
    b = rnd((c1 + c2,), dtype, 4)
    y = ops.groupnorm(x1, g, b, m, hw, 32, eps, silu, x2=x2)
    ref = L.groupnorm_ref(x1, g, b, m, hw, 32, eps, silu, x2=x2)
    check(y, ref, dtype, "groupnorm m%d hw%d c%d+%d silu=%s" % (m, hw, c1, c2, silu), 2.0)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("m,hw,c", [(2, 1400, 320), (2, 350, 640), (3, 28, 1280), (1, 5600, 128)])
def test_groupnorm_large_mean(ops, dtype, m, hw, c):
    """Groups whose |mean| is ~100x their std (real SD checkpoints have them): a one-pass
    E[x^2] - mean^2 variance loses ~13 bits of the fp32 statistics there; the kernels accumulate around a
    per-group pivot instead.  The values are exactly representable inputs, so the reference (two-pass,
    fp32) sees the same numbers."""
    gen = torch.Generator().manual_seed(5)
    base = (torch.randn((m, 1, 32, 1), generator=gen) * 60.0 + 100.0).expand(m, hw, 32, c // 32)
    x = (base + torch.randn((m, hw, 32, c // 32), generator=gen)).reshape(m * hw, c).to(dtype).cuda()
    g = rnd((c,), dtype, 3) + 1.0
    b = rnd((c,), dtype, 4)
    y = ops.groupnorm(x, g, b, m, hw, 32, 1e-5, False)
    ref = L.groupnorm_ref(x, g, b, m, hw, 32, 1e-5, False)
    check(y, ref, dtype, "groupnorm large-mean m%d hw%d c%d" % (m, hw, c), 2.0)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("rows,c", [(1400 * 2, 320), (701, 640), (91, 1280)])
def test_layernorm(ops, dtype, rows, c):
    x = rnd((rows, c), dtype, 1) * 3 + 1
    g = rnd((c,), dtype, 2) + 1.0
    b = rnd((c,), dtype, 3)
    check(ops.layernorm(x, g, b), L.layernorm_ref(x, g, b), dtype, "layernorm %dx%d" % (rows, c), 2.0)


# ------------------------------------------------------------------------- attention ----
ATTN_CASES = [
    # batch, lq, lk, heads, d
    (2, 1400, 1400, 8, 40), (3, 350, 350, 8, 80), (3, 91, 91, 8, 160), (6, 28, 28, 8, 160),
    (2, 1400, 98, 8, 40), (2, 350, 78, 8, 80), (2, 91, 98, 8, 160), (2, 1400, 77, 8, 40),
    (1, 17, 1, 8, 40), (1, 130, 65, 2, 80),
]


@pytest.mark.parametrize("variant", [0])
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", ATTN_CASES, ids=[str(c) for c in ATTN_CASES])
def test_attention(ops, dtype, variant, case):
    b, lq, lk, h, d = case
    c = h * d
    q = rnd((b * lq, c), dtype, 1)
    k = rnd((b * lk, c), dtype, 2)
    v = rnd((b * lk, c), dtype, 3)
    y = ops.attention(q, k, v, b, lq, lk, h, d, variant=variant)
    # P is rounded to the storage type before the PV product (as flash kernels do): 4x tol
    check(y, L.attention_ref(q, k, v, b, lq, lk, h, d), dtype, "attention %s v%d" % (case, variant), 4.0)


@pytest.mark.parametrize("dtype", DTYPES)
def test_attention_fused_qkv_and_neighbours(ops, dtype):
    """Strided q/k/v views of one fused [rows, 3C] projection + kv_batch_map + accumulate
    (the attn4 pattern: out = Attn(q_v, kv_left) + Attn(q_v, kv_right))."""
    b, l, h, d = 6, 350, 8, 80
    c = h * d
    qkv = rnd((b * l, 3 * c), dtype, 1)
    q, k, v = qkv[:, :c], qkv[:, c:2 * c], qkv[:, 2 * c:]
    left = torch.tensor([5, 0, 1, 2, 3, 4], dtype=torch.int32, device="cuda")
    right = torch.tensor([1, 2, 3, 4, 5, 0], dtype=torch.int32, device="cuda")
    out = ops.attention(q, k, v, b, l, l, h, d, kv_batch_map=left)
    ops.attention(q, k, v, b, l, l, h, d, kv_batch_map=right, out=out, accumulate=True)
    ref = (L.attention_ref(q, k, v, b, l, l, h, d, kv_batch_map=left)
           + L.attention_ref(q, k, v, b, l, l, h, d, kv_batch_map=right))
    check(out, ref, dtype, "attn4 neighbour sum", 6.0)


def test_attention_softmax_spike(ops):
    """Force the online-softmax rescale branch: one key dominates late in the sequence."""
    b, lq, lk, h, d = 1, 64, 300, 8, 40
    dtype = torch.float16
    q = rnd((b * lq, h * d), dtype, 1)
    k = rnd((b * lk, h * d), dtype, 2)
    v = rnd((b * lk, h * d), dtype, 3)
    k[250] = q[7] * 4.0
    y = ops.attention(q, k, v, b, lq, lk, h, d)
    check(y, L.attention_ref(q, k, v, b, lq, lk, h, d), dtype, "attention spike", 4.0)


@pytest.mark.parametrize("variant", [0])
@pytest.mark.parametrize("lk", [1, 31, 33, 100, 129, 200])
def test_attention_ragged_tail_all_scores_negative(ops, variant, lk):
    """Keys past lk are zero rows in LDS and score exactly 0.  When every real score of a row is far
    below 0 the padded keys would own the running max (and, without the zeroed ones column, the
    denominator): the result must still be the softmax over the lk real keys only."""
    b, lq, h, d = 2, 48, 8, 40
    dtype = torch.bfloat16
    q = rnd((b * lq, h * d), dtype, 1) + 1.5
    k = -(rnd((b * lk, h * d), dtype, 2).abs() + 1.0) * 2.0       # q . k strongly negative everywhere
    v = rnd((b * lk, h * d), dtype, 3)
    y = ops.attention(q, k, v, b, lq, lk, h, d, variant=variant)
    ref = L.attention_ref(q, k, v, b, lq, lk, h, d)
    assert torch.isfinite(y.float()).all()
    check(y, ref, dtype, "attention ragged tail lk=%d v%d" % (lk, variant), 4.0)


@pytest.mark.parametrize("variant", [0])
@pytest.mark.parametrize("dtype", DTYPES)
def test_attention_v5_fused_qkv_and_neighbours(ops, dtype, variant):
    """The buffer-load staging with strided q/k/v views of one fused projection, kv_batch_map and
    accumulate (attn4) — at d = 40 with a ragged last key tile."""
    b, l, h, d = 6, 350, 8, 40
    c = h * d
    qkv = rnd((b * l, 3 * c), dtype, 1)
    q, k, v = qkv[:, :c], qkv[:, c:2 * c], qkv[:, 2 * c:]
    left = torch.tensor([5, 0, 1, 2, 3, 4], dtype=torch.int32, device="cuda")
    right = torch.tensor([1, 2, 3, 4, 5, 0], dtype=torch.int32, device="cuda")
    out = ops.attention(q, k, v, b, l, l, h, d, kv_batch_map=left, variant=variant)
    ops.attention(q, k, v, b, l, l, h, d, kv_batch_map=right, out=out, accumulate=True, variant=variant)
    ref = (L.attention_ref(q, k, v, b, l, l, h, d, kv_batch_map=left)
           + L.attention_ref(q, k, v, b, l, l, h, d, kv_batch_map=right))
    check(out, ref, dtype, "attn4 neighbour sum v%d" % variant, 6.0)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", [(6, 350, 8, 40), (3, 1400, 8, 40), (4, 91, 8, 160), (2, 350, 8, 80)])
def test_attention_head_major_qkv(ops, dtype, case):
    """Fused QKV projection written HEAD-MAJOR by the GEMM epilogue (one [rows][D] plane per head, the Q
    planes pre-multiplied by scale * log2 e) -> attention on those planes, incl. the neighbour-view pattern
    (kv_batch_map + accumulate).  Against the row-major reference: projection, then two attentions summed."""
    b, l, h, d = case
    c = h * d
    x = rnd((b * l, c), dtype, 1)
    w = rnd((3 * c, c), dtype, 2, c ** -0.5)
    scale = d ** -0.5
    qkv_hm = ops.gemm(x, w, None, head_major=(d, h, scale * 1.4426950408889634))
    assert qkv_hm.shape == (3 * h, b * l, d)
    qkv = ops.gemm(x, w, None)                                   # row-major twin
    # the planes hold the same projection (Q planes scaled)
    hm = qkv_hm.float().cpu().permute(1, 0, 2).reshape(b * l, 3 * c)
    rm = qkv.float().cpu()
    tol = {torch.float16: 2e-3, torch.bfloat16: 1.6e-2}[dtype]
    assert (hm[:, c:] - rm[:, c:]).abs().max().item() == 0.0
    assert (hm[:, :c] - rm[:, :c] * scale * 1.4426950408889634).abs().max().item() <= tol * rm.abs().max().item()
    q, k, v = qkv_hm[:h], qkv_hm[h:2 * h], qkv_hm[2 * h:]
    left = torch.tensor([(i - 1) % b for i in range(b)], dtype=torch.int32, device="cuda")
    right = torch.tensor([(i + 1) % b for i in range(b)], dtype=torch.int32, device="cuda")
    out = ops.attention(q, k, v, b, l, l, h, d, kv_batch_map=left, q_prescaled=True)
    ops.attention(q, k, v, b, l, l, h, d, kv_batch_map=right, out=out, accumulate=True, q_prescaled=True)
    ref = (L.attention_ref(qkv[:, :c], qkv[:, c:2 * c], qkv[:, 2 * c:], b, l, l, h, d, kv_batch_map=left)
           + L.attention_ref(qkv[:, :c], qkv[:, c:2 * c], qkv[:, 2 * c:], b, l, l, h, d, kv_batch_map=right))
    check(out, ref, dtype, "head-major attn4 %s" % (case,), 6.0)
    own = ops.attention(q, k, v, b, l, l, h, d, q_prescaled=True)
    check(own, L.attention_ref(qkv[:, :c], qkv[:, c:2 * c], qkv[:, 2 * c:], b, l, l, h, d), dtype,
          "head-major self %s" % (case,), 4.0)


LN2 = 0.6931471805599453


@pytest.mark.parametrize("variant", [0])
@pytest.mark.parametrize("kind", ["spike", "negative", "plain"])
@pytest.mark.parametrize("lk", [1, 33, 200, 300])
def test_attention_prescaled_q(ops, variant, kind, lk):
    """q_prescaled: q carries scale * log2(e), the running max rides in the QK^T MFMA's C operand.  Late
    spikes (rescale branch, also inside the ragged last chunk), rows whose every score is far below zero
    (the first chunk must LOWER the initial max of 0; padded keys score above every real one), plain data."""
    b, lq, h, d = 2, 80, 8, 40
    dtype = torch.bfloat16
    q = rnd((b * lq, h * d), dtype, 1)
    k = rnd((b * lk, h * d), dtype, 2)
    v = rnd((b * lk, h * d), dtype, 3)
    if kind == "spike":
        k[lk - 1] = q[7] * 6.0
        if lk > 40:
            k[lk // 2] = q[9] * 4.0
    elif kind == "negative":
        q = q + 1.5
        k = -(k.abs() + 1.0) * 3.0
    # q is what the projection epilogue would have stored: already in log2 units
    y = ops.attention(q, k, v, b, lq, lk, h, d, variant=variant, q_prescaled=True)
    ref = L.attention_ref(q, k, v, b, lq, lk, h, d, scale=LN2)
    assert torch.isfinite(y.float()).all()
    check(y, ref, dtype, "prescaled attention %s lk=%d v%d" % (kind, lk, variant), 4.0)


@pytest.mark.parametrize("variant", [0])
def test_attention_v5_softmax_spike(ops, variant):
    b, lq, lk, h, d = 1, 64, 300, 8, 40
    dtype = torch.float16
    q = rnd((b * lq, h * d), dtype, 1)
    k = rnd((b * lk, h * d), dtype, 2)
    v = rnd((b * lk, h * d), dtype, 3)
    k[250] = q[7] * 4.0
    k[299] = q[9] * 4.0                     # spike inside the ragged last chunk
    y = ops.attention(q, k, v, b, lq, lk, h, d, variant=variant)
    check(y, L.attention_ref(q, k, v, b, lq, lk, h, d), dtype, "attention spike v%d" % variant, 4.0)


# ----------------------------------------------------------------------- elementwise ----
@pytest.mark.parametrize("dtype", DTYPES)
def test_elementwise(ops, dtype):
    a, b, c = rnd((1400 * 320,), dtype, 1), rnd((1400 * 320,), dtype, 2), rnd((1400 * 320,), dtype, 3)
    check(ops.add(a, b), a.float().cpu() + b.float().cpu(), dtype, "add2")
    check(ops.add(a, b, c), a.float().cpu() + b.float().cpu() + c.float().cpu(), dtype, "add3")
    check(ops.scale(a, 0.37), a.float().cpu() * 0.37, dtype, "scale")
    check(ops.silu(a), torch.nn.functional.silu(a.float().cpu()), dtype, "silu")
    x = rnd((3, 4, 28, 50), dtype, 4)
    y = ops.nchw_to_nhwc(x, 8)
    ref = torch.zeros(3 * 1400, 8)
    ref[:, :4] = x.float().cpu().permute(0, 2, 3, 1).reshape(-1, 4)
    check(y, ref, dtype, "nchw->nhwc pad8")
    z = ops.nhwc_to_nchw(y, 3, 4, 28, 50)
    check(z, x.float().cpu(), dtype, "nhwc->nchw")


@pytest.mark.parametrize("dtype", DTYPES)
def test_timestep_embedding(ops, dtype):
    t = torch.tensor([981.0, 1.0, 500.0, 0.0], device="cuda")
    y = ops.timestep_embedding(t, 320, dtype)
    check(y, L.timestep_embedding_ref(t, 320), dtype, "timestep embedding", 2.0)


@pytest.mark.parametrize("dtype", DTYPES)
def test_cfg_ddim(ops, dtype):
    n = 6 * 4 * 28 * 50
    eps = rnd((2, n), dtype, 1)
    x = rnd((n,), dtype, 2)
    coef = torch.tensor([0.9, 0.4359, 0.95, 0.3122], device="cuda")
    dup = torch.empty((2, n), dtype=dtype, device="cuda")
    y = ops.cfg_ddim_step(eps, x, coef, 2.0, x_dup=dup[1])
    ref = L.cfg_ddim_ref(eps, x, coef.cpu(), 2.0)
    check(y, ref, dtype, "cfg+ddim", 3.0)
    assert torch.equal(dup[1], y)


def test_fails_loudly_on_cpu_tensor(ops):
    with pytest.raises(RuntimeError):
        ops.add(torch.zeros(8, dtype=torch.float16), torch.zeros(8, dtype=torch.float16))


@pytest.mark.parametrize("dtype", DTYPES)
def test_cfg_unipc_sequence(ops, dtype):
    """dd_cfg_unipc_step over a whole 8-step UniPC run (warm-up, order-2 steps, lower-order final step)
    against the step-by-step restatement fed with the same guided noise."""
    from dualdiff_amd.pipeline.schedulers import unipc_schedule
    from oracle.unipc import UniPCRestated
    steps, n, g = 8, 6 * 4 * 28 * 50, 2.0
    sch = UniPCRestated()
    ts = sch.set_timesteps(steps)
    _, tab = unipc_schedule(steps)
    x = rnd((n,), dtype, 1)
    hist = torch.zeros((3, n), dtype=torch.float32, device="cuda")
    ref = x.float().cpu().double()
    dup = torch.empty_like(x)
    for i, t in enumerate(ts.tolist()):
        eps = rnd((2, n), dtype, 10 + i)
        e32 = eps.float().cpu()
        guided = (e32[0] + g * (e32[1] - e32[0])).to(dtype).double()       # rounded like the kernel / reference
        ref = sch.step(guided, t, ref)
        x = ops.cfg_unipc_step(eps, x, hist, tab[i].cuda(), g, x_dup=dup)
        assert torch.equal(x, dup)
        # the kernel's sample is rounded to the storage type every step; follow it so errors do not compound
        check(x, ref.float(), dtype, "cfg+unipc step %d" % i, 3.0)
        ref = x.float().cpu().double()
        sch.last_sample = hist[0].cpu().double()
        sch.model_outputs = [hist[2].cpu().double(), hist[1].cpu().double()]


# ------------------------------------------------------------------ ORS projection (N3) ----
@torch.no_grad()
def test_ors_projection_bit_exact_gpu():
    """dd_ors_project through dualdiff_amd.networks.occ3d_proj.OccupancyRay against (a) the labels the
    reference's own project() produced (golden) and (b) the oracle — integer work, every label equal."""
    import os
    import numpy as np
    from oracle import ors_projection as P
    from tests.golden import cases as C
    from dualdiff_amd.networks.occ3d_proj import OccupancyRay
    occ, Ks, Rts = C.ors_inputs()
    proj = OccupancyRay(image_shape=(900, 1600), sample_point=C.ORS_S, sample_step=0.2,
                        compress_ratio=C.ORS_RATIO, device="cuda")
    assert proj.image_shape_compress == [C.ORS_H, C.ORS_W]
    lab = proj.project_volume(occ, Ks, Rts).cpu()
    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "ors_projection.npz"))["labels"]
    ref = torch.from_numpy(gold.astype(np.int64))
    assert lab.shape == ref.shape
    assert torch.equal(lab, ref), "%d of %d labels differ from the reference" % ((lab != ref).sum().item(), ref.numel())
    assert torch.equal(lab, P.ors_project(occ, Ks, Rts, C.ORS_H, C.ORS_W, C.ORS_RATIO, C.ORS_S, 0.2))
    # fused condition output: filtering + /17 + channels-first, in the branch's storage dtype
    for fg, bg in ((True, True), (True, False), (False, True)):
        for dt in (torch.float16, torch.bfloat16):
            cond = proj.condition_volume(occ, Ks, Rts, dtype=dt, use_fg=fg, use_bg=bg).cpu()
            want = P.ors_condition(ref, use_fg=fg, use_bg=bg).to(dt)
            assert torch.equal(cond, want), (fg, bg, dt)


@torch.no_grad()
@pytest.mark.parametrize("seed", [0, 1, 2])
def test_ors_projection_random_rigs(seed):
    """Random volumes and rigs (cameras inside and outside the volume, rays leaving through every face,
    other latent sizes / sample counts): bit-exact against the oracle."""
    from oracle import ors_projection as P
    from dualdiff_amd.networks.occ3d_proj import OccupancyRay
    g = torch.Generator().manual_seed(900 + seed)
    occ = torch.randint(0, 18, (200, 200, 16), generator=g)
    h, w, s = [(7, 13, 64), (28, 50, 200), (16, 9, 511)][seed]
    Ks, Rts = [], []
    for i in range(3 + seed):
        q, _ = torch.linalg.qr(torch.randn(3, 3, generator=g, dtype=torch.float64))
        Rt = torch.eye(4, dtype=torch.float64)
        Rt[:3, :3] = q
        Rt[:3, 3] = (torch.rand(3, generator=g, dtype=torch.float64) - 0.5) * torch.tensor([90.0, 90.0, 8.0])
        Rt[2, 3] += 2.0
        K = torch.tensor([[1260.0 + 10 * i, 0.0, 800.0], [0.0, 1255.0, 450.0 + 5 * i], [0.0, 0.0, 1.0]])
        Ks.append(K)
        Rts.append(Rt.float())
    ratio = w / 1600
    proj = OccupancyRay(image_shape=(h / ratio + 0.5, 1600), sample_point=s, sample_step=0.25 if seed == 2 else 0.2,
                        compress_ratio=ratio, device="cuda")
    assert proj.image_shape_compress == [h, w]
    lab = proj.project_volume(occ, Ks, Rts).cpu()
    want = P.ors_project(occ, Ks, Rts, h, w, ratio, s, proj.sample_step)
    assert torch.equal(lab, want), "%d of %d labels differ" % ((lab != want).sum().item(), want.numel())
    assert (lab == 17).any() and (lab != 17).any()


def test_ors_projection_rejects_bad_input():
    from dualdiff_amd import ops as O
    occ = torch.zeros((200, 200, 8), dtype=torch.uint8, device="cuda")
    with pytest.raises(ValueError):
        O.ors_project(occ, torch.zeros(1, 3, device="cuda"), torch.zeros(1, 4, 3, device="cuda"), 8)
    occ = torch.zeros((200, 200, 16), dtype=torch.uint8, device="cuda")
    with pytest.raises(TypeError):
        O.ors_project(occ, torch.zeros(1, 3, device="cuda"), torch.zeros(1, 4, 3, device="cuda"), 8,
                      cond_dtype=torch.float32)


def test_ors_projection_edge_cases():
    """Rays that never enter the volume (camera far outside, looking away) give class 17 everywhere; an
    empty volume (all 17) gives 17 everywhere; a camera inside a uniform volume reads that class until its
    rays leave the 80 m x 80 m x 6.4 m box, and 17 after."""
    from oracle import ors_projection as P
    from dualdiff_amd.networks.occ3d_proj import OccupancyRay
    K = torch.tensor([[1260.0, 0.0, 800.0], [0.0, 1260.0, 450.0], [0.0, 0.0, 1.0]])
    proj = OccupancyRay(image_shape=(900, 1600), sample_point=64, sample_step=0.5, compress_ratio=16 / 1600,
                        device="cuda")
    h, w = proj.image_shape_compress
    # camera axes: z forward = +x of the ego frame, x right = -y, y down = -z
    R = torch.tensor([[0.0, 0.0, 1.0], [-1.0, 0.0, 0.0], [0.0, -1.0, 0.0]])
    away = torch.eye(4); away[:3, :3] = R; away[:3, 3] = torch.tensor([100.0, 0.0, 1.0])       # 60 m outside, looking out
    inside = torch.eye(4); inside[:3, :3] = R; inside[:3, 3] = torch.tensor([0.0, 0.0, 1.0])
    full = torch.full((200, 200, 16), 5)
    lab = proj.project_volume(full, [K, K], [away, inside]).cpu()
    assert torch.equal(lab, P.ors_project(full, [K, K], [away, inside], h, w, 16 / 1600, 64, 0.5))
    assert (lab[0] == 17).all()
    assert (lab[1, :, :, 0] == 5).all() and (lab[1, h // 2, w // 2, :60] == 5).all()
    assert (lab[1] == 17).any()                               # steep rays leave through the top / bottom
    empty = torch.full((200, 200, 16), 17)
    assert (proj.project_volume(empty, [K], [inside]) == 17).all()
    cond = proj.condition_volume(empty, [K], [inside], dtype=torch.bfloat16)
    assert cond.shape == (1, 64, h, w) and (cond.float() == 1.0).all()


# ------------------------------------------------------------------ persistent tile walk ----
@pytest.mark.parametrize("tile", DMA_TILES + [59])
def test_gemm_persistent_walk_dense(ops, tile):
    """More tiles than resident workgroups and a short K loop (the QKV / to_out shapes of the 28x50 level): the
    LDS-DMA family walks several tiles per workgroup with the ring running ahead across the tile boundary
    (dd_gemm2_kernel, persistent mode).  Two-source A, bias, time-embedding vector, residual, ragged last row tile;
    every tile configuration must give the SAME bits (same K order per output element) and match the fp32 reference."""
    dtype = torch.float16
    rows, n, k = 16800 - 37, 960, 320
    a = rnd((rows, 192), dtype, 1)
    a2 = rnd((rows, k - 192), dtype, 11)
    w = rnd((n, k), dtype, 2, 0.05)
    b = rnd((n,), dtype, 3)
    res = rnd((rows, n), dtype, 4)
    y = ops.gemm(a, w, b, a2=a2, res=res, tile=tile, split_k=1)
    check(y, L.linear_ref(a, w, b, a2=a2, res=res), dtype, "persistent dense tile%d" % tile)
    y0 = ops.gemm(a, w, b, a2=a2, res=res, tile=12, split_k=1)
    assert torch.equal(y, y0), "tile %d and tile 12 disagree bitwise" % tile


@pytest.mark.parametrize("tile", [11, 12, 14, 16, 20, 24, 44, 46])
def test_gemm_persistent_walk_geglu_headmajor(ops, tile):
    dtype = torch.bfloat16
    rows, c = 16800, 320
    a = rnd((rows, c), dtype, 1)
    w = rnd((8 * c, c), dtype, 2, 0.05)
    b = rnd((8 * c,), dtype, 3)
    y = ops.gemm(a, w, b, epilogue=ops.DD_EPI_GEGLU, tile=tile)
    check(y, L.linear_ref(a, w, b, geglu=True), dtype, "persistent geglu tile%d" % tile, 2.0)
    w3 = rnd((3 * c, c), dtype, 5, 0.05)
    hm = ops.gemm(a, w3, None, head_major=(40, 8, 0.5), tile=tile)        # (24 planes, rows, 40)
    ref = L.linear_ref(a, w3, None).reshape(rows, 24, 40).permute(1, 0, 2).clone()
    ref[:8] *= 0.5
    check(hm, ref, dtype, "persistent head-major tile%d" % tile, 2.0)


# ------------------------------------------------------------------ pipelined dense family (round 5) ----
P_TILES = [72, 73, 74, 75, 76, 77, 78]
P_TWIN = {72: 52, 73: 52, 74: 40, 75: 44, 76: 59, 77: 59, 78: 28}     # same tile shape, dd_gemm2_kernel
P_GEGLU = [75]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("tile", P_TILES)
def test_gemm_pipelined_tiles(ops, dtype, tile):
    """dd_gemm3_kernel (fragment reads of K-step c+1 under the MFMAs of K-step c): every K-loop length around the ring
    depth (1 .. 7 steps and long), ragged row / column tails, and BIT-identity with the dd_gemm2_kernel tile of the same
    shape (same K order per accumulator)."""
    for (rows, n, k) in [(256, 256, 64), (200, 320, 128), (333, 192, 192), (96, 64, 256), (1092, 1280, 320),
                         (77, 72, 384), (129, 136, 448), (336, 1280, 1280), (1400, 320, 2880)]:
        a = rnd((rows, k), dtype, 1)
        w = rnd((n, k), dtype, 2, 0.05)
        b = rnd((n,), dtype, 3)
        y = ops.gemm(a, w, b, tile=tile, split_k=1)
        check(y, L.linear_ref(a, w, b), dtype, "gemm3 tile%d %dx%dx%d" % (tile, rows, n, k))
        assert torch.equal(y, ops.gemm(a, w, b, tile=P_TWIN[tile], split_k=1)), "tile %d vs its gemm2 twin" % tile


@pytest.mark.parametrize("tile", P_TILES)
def test_gemm_pipelined_epilogues(ops, tile):
    """Two-source A (seam inside the K range, incl. a seam that falls into the prologue stages), bias + time vector +
    residual + alpha, accumulate into a strided view, SiLU, head-major planes, split-K (slabs + reduce launch)."""
    dtype = torch.float16
    rows, n, k = 12 * 91, 640, 1280
    for k1 in (768, 64, 1216):
        a = rnd((rows, k1), dtype, 1)
        a2 = rnd((rows, k - k1), dtype, 11)
        w = rnd((n, k), dtype, 2, 0.03)
        b = rnd((n,), dtype, 3)
        res = rnd((rows, n), dtype, 4)
        rv = rnd((12, n), dtype, 5)
        y = ops.gemm(a, w, b, a2=a2, res=res, rowvec=rv, rows_per_inst=91, alpha=0.5, tile=tile, split_k=1)
        ref = L.linear_ref(a, w, b, a2=a2, res=res, rowvec=rv, rows_per_inst=91, alpha=0.5)
        check(y, ref, dtype, "gemm3 concat tile%d k1=%d" % (tile, k1))
        for split in (2, 3, 5):
            ys = ops.gemm(a, w, b, a2=a2, res=res, rowvec=rv, rows_per_inst=91, alpha=0.5, tile=tile, split_k=split)
            check(ys, ref, dtype, "gemm3 concat tile%d k1=%d split%d" % (tile, k1, split))
    a = rnd((rows, k), dtype, 1)
    buf = rnd((rows, n + 64), dtype, 6)
    want = buf.float().cpu().clone()
    want[:, 32:32 + n] += L.linear_ref(a, w, b)
    ops.gemm(a, w, b, out=buf[:, 32:32 + n], accumulate=True, tile=tile, split_k=1)
    check(buf, want, dtype, "gemm3 accumulate strided tile%d" % tile, 2.0)
    y = ops.gemm(a, w, b, epilogue=ops.DD_EPI_SILU, tile=tile, split_k=1)
    check(y, torch.nn.functional.silu(L.linear_ref(a, w, b)), dtype, "gemm3 silu tile%d" % tile, 2.0)
    w3 = rnd((3 * 640, k), dtype, 7, 0.03)
    hm = ops.gemm(a, w3, None, head_major=(80, 8, 0.25), tile=tile, split_k=1)
    want = L.linear_ref(a, w3, None).reshape(rows, 24, 80).permute(1, 0, 2).clone()
    want[:8] *= 0.25
    check(hm, want, dtype, "gemm3 head-major tile%d" % tile, 2.0)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("tile", P_GEGLU)
def test_gemm_pipelined_geglu(ops, dtype, tile):
    for rows, c in ((700, 640), (1092, 1280), (173, 320)):
        a = rnd((rows, c), dtype, 1)
        w = rnd((8 * c, c), dtype, 2, 0.05)
        b = rnd((8 * c,), dtype, 3)
        y = ops.gemm(a, w, b, epilogue=ops.DD_EPI_GEGLU, tile=tile)
        check(y, L.linear_ref(a, w, b, geglu=True), dtype, "gemm3 geglu tile%d %dx%d" % (tile, rows, c), 2.0)
        assert torch.equal(y, ops.gemm(a, w, b, epilogue=ops.DD_EPI_GEGLU, tile=P_TWIN[tile]))


def test_gemm_pipelined_rejects_conv_and_ln(ops):
    x = rnd((2 * 14 * 25, 64), torch.float16, 1)
    w = rnd((64, 64, 3, 3), torch.float16, 2, 0.05)
    with pytest.raises(RuntimeError):
        ops.conv3x3(x, L.pack_conv_weight(w), None, 2, 14, 25, tile=72)


# ------------------------------------------------------------------ thin conv (condition embedder) ----
THIN_CASES = [(8, 16, 1), (16, 16, 1), (16, 32, 2), (32, 32, 1), (8, 32, 1), (16, 32, 1), (16, 16, 2), (8, 16, 2),
              (32, 16, 1)]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cin,cout,stride", THIN_CASES)
@pytest.mark.parametrize("m,h,w_", [(3, 37, 53), (2, 8, 64), (1, 5, 200), (2, 56, 100)])
def test_conv3x3_thin_channels(ops, dtype, cin, cout, stride, m, h, w_):
    """dd_conv3x3_thin (first layers of ControlNetConditioningEmbedding, map_embedder.py:79-113): odd image sizes
    (partial tiles in both directions, borders inside a tile), both strides, bias + SiLU, against the fp32 reference."""
    assert ops.thin_conv_ok(cin, cout, stride, m)
    real_cin = 3 if cin == 8 else cin                          # 3 condition channels arrive zero-padded to 8
    x = rnd((m * h * w_, cin), dtype, 1)
    if real_cin != cin:
        x[:, real_cin:] = 0
    w = rnd((cout, cin, 3, 3), dtype, 2, (9 * real_cin) ** -0.5)
    b = rnd((cout,), dtype, 3)
    for silu in (False, True):
        y = ops.conv3x3(x, L.pack_conv_weight(w), b, m, h, w_, stride=stride,
                        epilogue=ops.DD_EPI_SILU if silu else ops.DD_EPI_NONE)
        ref = L.conv3x3_ref(x, w, b, m, h, w_, stride=stride)
        if silu:
            ref = torch.nn.functional.silu(ref)
        check(y, ref, dtype, "thin conv %d->%d s%d %dx%d silu=%d" % (cin, cout, stride, h, w_, silu))


# ------------------------------------------------------------------ direct conv on row bands (28x50 level) ----
@pytest.mark.parametrize("case", [(12, 28, 50, 320, 320), (3, 28, 50, 640, 320), (2, 30, 41, 64, 72), (1, 20, 20, 128, 64),
                                  (2, 70, 6, 64, 64), (5, 28, 50, 960, 320)], ids=lambda c: str(c))
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("btile", [39])
def test_conv3x3_band_direct(ops, case, dtype, btile):
    """BAND form of dd_conv3s_kernel (tile 39): images larger than the 384-row tile are cut into bands of whole image rows
    with a W + 1 pixel halo on either side; first / last band (image border inside the halo), a ragged last band, widths
    that leave 1..8 image rows per band, ResnetBlock2D epilogue, split-K over channel chunks."""
    m, h, w_, cin, cout = case
    x = rnd((m * h * w_, cin), dtype, 1)
    w = rnd((cout, cin, 3, 3), dtype, 2, (9 * cin) ** -0.5)
    b = rnd((cout,), dtype, 3)
    temb = rnd((m, cout), dtype, 4)
    res = rnd((m * h * w_, cout), dtype, 5)
    ref = L.conv3x3_ref(x, w, b, m, h, w_) + temb.float().cpu().repeat_interleave(h * w_, 0) + res.float().cpu()
    for split in (1, 2, 5):
        if split > cin // 64:
            continue
        y = ops.conv3x3(x, L.pack_conv_weight(w), b, m, h, w_, rowvec=temb, res=res, tile=btile, split_k=split)
        check(y, ref, dtype, "conv3s band t%d split%d %s" % (btile, split, case))
        if btile != 39:              # same K order per accumulator as the eight-wave form: same bits
            assert torch.equal(y, ops.conv3x3(x, L.pack_conv_weight(w), b, m, h, w_, rowvec=temb, res=res, tile=39, split_k=split))
    # same bits as the implicit-GEMM family's 160x160 tile?  No: different K order per output (tap-major vs chunk-major);
    # but two band launches agree with each other bit for bit
    y1 = ops.conv3x3(x, L.pack_conv_weight(w), b, m, h, w_, tile=btile, split_k=1)
    y2 = ops.conv3x3(x, L.pack_conv_weight(w), b, m, h, w_, tile=btile, split_k=1)
    assert torch.equal(y1, y2)


# ------------------------------------------------------------------ neighbour PAIR in one attention launch ----
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("prescaled", [False, True])
@pytest.mark.parametrize("b,l,h,d", [(12, 1400, 8, 40), (6, 350, 8, 80), (12, 91, 8, 160), (6, 28, 8, 160), (6, 130, 2, 40),
                                     (6, 65, 8, 80)])
def test_attention_neighbour_pair_one_launch(ops, dtype, prescaled, b, l, h, d):
    """dd_attn_desc.kv_batch_map2 (round 3): Attn(q, kv[left]) + Attn(q, kv[right]) in ONE launch — two softmax passes
    with their own state, summed in fp32 and rounded once — against the oracle's sum and against the two-launch
    accumulate form it replaces (which rounds twice), on head-major planes as the model passes them."""
    c = h * d
    scale = d ** -0.5
    qkv = rnd((b * l, 3 * c), dtype, 1)
    q, k, v = qkv[:, :c], qkv[:, c:2 * c], qkv[:, 2 * c:]
    n = 6
    left = torch.tensor([(i // n) * n + (i % n + n - 1) % n for i in range(b)], dtype=torch.int32, device="cuda")
    right = torch.tensor([(i // n) * n + (i % n + 1) % n for i in range(b)], dtype=torch.int32, device="cuda")
    ref = (L.attention_ref(q, k, v, b, l, l, h, d, kv_batch_map=left)
           + L.attention_ref(q, k, v, b, l, l, h, d, kv_batch_map=right))
    if prescaled:       # head-major planes, Q carrying scale * log2(e) (what the projection epilogue writes)
        def planes(t, s=1.0):
            return (t.float() * s).to(dtype).reshape(b * l, h, d).permute(1, 0, 2).contiguous()
        qh, kh, vh = planes(q, scale * 1.4426950408889634), planes(k), planes(v)
        one = ops.attention(qh, kh, vh, b, l, l, h, d, kv_batch_map=left, kv_batch_map2=right, q_prescaled=True)
        two = ops.attention(qh, kh, vh, b, l, l, h, d, kv_batch_map=left, q_prescaled=True)
        ops.attention(qh, kh, vh, b, l, l, h, d, kv_batch_map=right, out=two, accumulate=True, q_prescaled=True)
    else:
        one = ops.attention(q, k, v, b, l, l, h, d, kv_batch_map=left, kv_batch_map2=right)
        two = ops.attention(q, k, v, b, l, l, h, d, kv_batch_map=left)
        ops.attention(q, k, v, b, l, l, h, d, kv_batch_map=right, out=two, accumulate=True)
    if prescaled:
        # Q was rounded AFTER the scale went in (as the projection epilogue does), so the oracle on the unscaled
        # operands is not the yardstick here: the two-launch form on the SAME planes is (it rounds twice -> 2 ulp)
        check(one, two.float().cpu(), dtype, "neighbour pair (head-major, prescaled) vs two launches b=%d l=%d d=%d" % (b, l, d), 3.0)
    else:
        check(one, ref, dtype, "neighbour pair one launch b=%d l=%d d=%d" % (b, l, d), 4.0)
        # one rounding instead of two: never worse than the two-launch form against the oracle
        e1 = (one.float().cpu() - ref).abs().max().item()
        e2 = (two.float().cpu() - ref).abs().max().item()
        assert e1 <= e2 * 1.05 + 1e-6, (e1, e2)
    # and a third neighbour still accumulates on top (odd neighbour counts)
    three = ops.attention(q, k, v, b, l, l, h, d, kv_batch_map=left, kv_batch_map2=right)
    ops.attention(q, k, v, b, l, l, h, d, kv_batch_map=left, out=three, accumulate=True)
    ref3 = ref + L.attention_ref(q, k, v, b, l, l, h, d, kv_batch_map=left)
    if not prescaled:
        check(three, ref3, dtype, "pair + accumulate", 6.0)


def test_attention_pair_rejects(ops):
    q = rnd((6 * 64, 320), torch.float16, 1)
    mp = torch.arange(6, dtype=torch.int32, device="cuda")
    with pytest.raises(ValueError):
        ops.attention(q, q, q, 6, 64, 64, 8, 40, kv_batch_map2=mp)                    # needs kv_batch_map
    with pytest.raises(RuntimeError):
        ops.attention(q, q, q, 6, 64, 64, 8, 40, kv_batch_map=mp, kv_batch_map2=mp, variant=7)   # the variants are gone (ABI 3)


# ------------------------------------------------------------------ fused cross-attention (dd_xattn320) ----
XATTN_CASES = [
    # instances, rows per instance, keys, residual, bias, ln_out, K/V as column slices of a wider projection
    (12, 1400, 77, True, True, False, False),      # SFA (txt_con_fusion.py), one scene
    (6, 1400, 98, True, True, True, True),         # attn2 of a 28x50 block: K/V bank slices, emits the next LayerNorm
    (3, 100, 98, True, True, True, True),          # ragged last tile (80 + 20 rows)
    (2, 80, 128, False, False, False, False),      # exactly one tile, the maximum key count, no epilogue extras
    (5, 37, 1, True, False, False, False),         # one key: softmax = 1
    (4, 161, 17, False, True, True, False),
    (2, 1400, 78, True, True, False, True),        # no boxes: 1 + 77 tokens
    (6, 1400, 15, True, True, True, True),         # the short token lists of the end-to-end parity cases
    (12, 1400, 9, True, True, False, False),
]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", XATTN_CASES, ids=lambda c: str(c))
def test_xattn320_fused(ops, dtype, case):
    """q-projection -> SDPA over the context keys -> out-projection + bias + residual (+ LayerNorm) in ONE launch vs
    the fp32 leaf-op oracle and vs the three-launch path it replaces (same roundings: q, P and the attention output
    in the storage type)."""
    inst, n, lk, has_res, has_bias, has_ln, sliced = case
    c, h, d = 320, 8, 40
    rows = inst * n
    x = rnd((rows, c), dtype, 1)
    res = rnd((rows, c), dtype, 2) if has_res else None
    wq = rnd((c, c), dtype, 3, c ** -0.5)
    wo = rnd((c, c), dtype, 4, c ** -0.5)
    bo = rnd((c,), dtype, 5) if has_bias else None
    if sliced:                                       # K | V as columns 640.. of a [inst*lk, 3*640] bank
        bank = rnd((inst * lk, 1920), dtype, 6)
        k, v = bank[:, 640:960], bank[:, 960:1280]
    else:
        k, v = rnd((inst * lk, c), dtype, 6), rnd((inst * lk, c), dtype, 7)
    g_, b_ = (1.0 + 0.1 * rnd((c,), torch.float32, 8)).to(dtype), rnd((c,), dtype, 9, 0.1)
    scale = d ** -0.5
    y = ops.xattn320(x, wq, wo, bo, k, v, inst, n, lk, scale, res=res, ln_out=(g_, b_, 1e-5) if has_ln else None)
    # fp32 oracle
    q_ref = L.linear_ref(x, wq, None)
    o_ref = L.attention_ref(q_ref, k.float().cpu(), v.float().cpu(), inst, n, lk, h, d, scale)
    ref = L.linear_ref(o_ref, wo, bo, res=res)
    check(y, ref, dtype, "xattn320 %s" % (case,), 8.0)
    # the three launches it replaces
    q = ops.gemm(x, wq)
    o = ops.attention(q, k, v, inst, n, lk, h, d, scale)
    y3 = ops.gemm(o, wo, bo, res=res)
    check(y, y3.float().cpu(), dtype, "xattn320 vs 3 launches %s" % (case,), 4.0)
    # the same K / V handed over as head-major planes (8, instances * lk, 40): identical arithmetic -> identical bits
    kh = k.reshape(inst * lk, h, d).permute(1, 0, 2).contiguous()
    vh = v.reshape(inst * lk, h, d).permute(1, 0, 2).contiguous()
    yh = ops.xattn320(x, wq, wo, bo, kh, vh, inst, n, lk, scale, res=res)
    assert torch.equal(yh, y), "head-major K / V operands changed the result"
    if has_ln:
        ln_ref = L.layernorm_ref(y, g_, b_)          # LayerNorm of the values as stored
        check(y._ln_out, ln_ref, dtype, "xattn320 ln_out %s" % (case,), 3.0)


def test_xattn320_rejects(ops):
    x = rnd((160, 320), torch.float16, 1)
    w = rnd((320, 320), torch.float16, 2)
    kv = rnd((2 * 129, 320), torch.float16, 3)
    with pytest.raises(RuntimeError):
        ops.xattn320(x, w, w, None, kv, kv, 2, 80, 129, 0.158)          # more than 128 keys
    with pytest.raises(ValueError):
        ops.xattn320(x[:, :312], w, w, None, kv[:256], kv[:256], 2, 80, 128, 0.158)


def test_xattn320_concurrent_launches_bitwise(ops):
    """The fused kernel moves every operand by LDS-DMA behind COUNTED vmcnt waits; a count that included instructions
    whose lanes are all out of range (they retire at once, out of order) let a wait pass with an older weight slab still in
    flight — invisible on an idle chip, wrong bits in 2 of 3 launches once other streams kept the memory system busy
    (the dual-branch step runs three such launches side by side).  Three streams, different key counts, background
    traffic: every result must equal the idle-chip result bit for bit."""
    dt, n, c = torch.float16, 1400, 320
    cases = []
    for i, (inst, lk) in enumerate(((6, 15), (6, 9), (6, 98))):
        x, res = rnd((inst * n, c), dt, 10 + i), rnd((inst * n, c), dt, 20 + i)
        wq, wo, b = rnd((c, c), dt, 30 + i, c ** -0.5), rnd((c, c), dt, 40 + i, c ** -0.5), rnd((c,), dt, 50 + i)
        bank = rnd((inst * lk, 1920), dt, 60 + i)
        g_, b_ = rnd((c,), dt, 70 + i), rnd((c,), dt, 80 + i)
        args = (x, wq, wo, b, bank[:, 640:960], bank[:, 960:1280], inst, n, lk, 40 ** -0.5)
        ref = ops.xattn320(*args, res=res, ln_out=(g_, b_, 1e-5))
        cases.append((args, res, (g_, b_, 1e-5), ref, ref._ln_out))
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream() for _ in cases]
    big = torch.empty(1 << 27, dtype=torch.float16, device="cuda")
    outs = []
    for it in range(40):
        for (args, res, ln, ref, lnref), st in zip(cases, streams):
            with torch.cuda.stream(st):
                y = ops.xattn320(*args, res=res, ln_out=ln)
                outs.append((y, y._ln_out, ref, lnref))
        if it % 3 == 0:
            big.add_(1)
    torch.cuda.synchronize()
    bad = sum(1 for y, yl, ref, lnref in outs if not torch.equal(y, ref) or not torch.equal(yl, lnref))
    assert bad == 0, "%d of %d concurrent launches differ from the idle-chip result" % (bad, len(outs))


def test_dma_gemm_conv_concurrent_launches_bitwise(ops):
    """The same hazard class checked for the LDS-DMA GEMM / conv families (counted vmcnt waits over rings whose padding
    taps and tile tails are out-of-range lanes): three streams + background traffic, idle-chip bits required."""
    dt = torch.float16
    cases = []
    for i, (m, h, w, cin, cout, tile, split) in enumerate(((12, 28, 50, 320, 320, 28, 1), (12, 28, 50, 320, 320, 39, 1),
                                                           (12, 14, 25, 640, 640, 12, 1), (12, 7, 13, 1280, 1280, 31, 4),
                                                           (12, 4, 7, 1280, 1280, 37, 5), (12, 4, 7, 640, 1280, 13, 3))):
        x, wt, b = rnd((m * h * w, cin), dt, i), rnd((cout, 9 * cin), dt, 10 + i, (9 * cin) ** -0.5), rnd((cout,), dt, 20 + i)
        fn = (lambda x=x, wt=wt, b=b, m=m, h=h, w=w, tile=tile, split=split: ops.conv3x3(x, wt, b, m, h, w, tile=tile, split_k=split))
        cases.append((fn, fn()))
    for i, (rows, n, k, tile, split) in enumerate(((1092, 1280, 1280, 13, 1), (16800, 320, 320, 27, 1), (1003, 328, 2048, 12, 2))):
        a, wt = rnd((rows, k), dt, 30 + i), rnd((n, k), dt, 40 + i, k ** -0.5)
        fn = (lambda a=a, wt=wt, tile=tile, split=split: ops.gemm(a, wt, tile=tile, split_k=split))
        cases.append((fn, fn()))
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream() for _ in range(3)]
    big = torch.empty(1 << 27, dtype=torch.float16, device="cuda")
    outs = []
    for it in range(15):
        for ci, (fn, ref) in enumerate(cases):
            with torch.cuda.stream(streams[(ci + it) % 3]):
                outs.append((ci, fn(), ref))
        if it % 2 == 0:
            big.add_(1)
    torch.cuda.synchronize()
    bad = sorted({ci for ci, y, ref in outs if not torch.equal(y, ref)})
    assert not bad, "cases with launches that differ from the idle-chip result: %s" % bad


# ------------------------------------------------------------------ split-K reduce folded into the consuming GroupNorm ----
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", [(12, 14, 25, 640, 640, 31, 2), (12, 7, 13, 1280, 1280, 31, 4), (12, 4, 7, 1280, 1280, 37, 5),
                                  (12, 4, 7, 2560, 1280, 31, 12), (5, 7, 13, 640, 1280, 12, 3), (3, 14, 25, 320, 640, 31, 2)],
                         ids=lambda c: str(c))
@pytest.mark.parametrize("silu,want_x", [(True, False), (False, True)])
def test_conv_splitk_reduce_folded_into_groupnorm(ops, dtype, case, silu, want_x, monkeypatch):
    """dd_groupnorm_splitk (round 3): the GroupNorm that reads a split-K conv's output adds the slabs, applies the conv's
    epilogue (bias + time vector + residual) and normalises in ONE launch — in place of the reduce launch + GroupNorm.
    Same arithmetic in the same order: bit-identical to the two-step form (both for y and, when kept, for x).  The path
    is OFF by default (0.5 % slower on the step, ops.GN_SPLITK) and switched on here."""
    from dualdiff_amd.networks.layers import GroupNorm
    monkeypatch.setattr(ops, "GN_SPLITK", True)
    m, h, w_, cin, cout, tile, split = case
    x = rnd((m * h * w_, cin), dtype, 1)
    w = L.pack_conv_weight(rnd((cout, cin, 3, 3), dtype, 2, (9 * cin) ** -0.5))
    b = rnd((cout,), dtype, 3)
    temb = rnd((m, cout), dtype, 4)
    res = rnd((m * h * w_, cout), dtype, 5) if want_x else None
    gn = GroupNorm(32, cout, 1e-6 if want_x else 1e-5).to("cuda", dtype)
    with torch.no_grad():
        gn.weight.copy_(1.0 + 0.1 * rnd((cout,), torch.float32, 6))
        gn.bias.copy_(rnd((cout,), torch.float32, 7, 0.1))
    ref_x = ops.conv3x3(x, w, b, m, h, w_, rowvec=temb, res=res, tile=tile, split_k=split)
    ref_y = gn.run(ref_x, m, h * w_, silu)
    out = ops.conv3x3(x, w, b, m, h, w_, rowvec=temb, res=res, tile=tile, split_k=split, gn_next=(gn, silu, want_x))
    assert getattr(out, "_gn_cache", None) is not None, "the fused reduce + GroupNorm path was not taken"
    y = gn.run(out, m, h * w_, silu)
    assert torch.equal(y, ref_y)
    if want_x:
        assert torch.equal(out, ref_x)
    else:
        with pytest.raises(RuntimeError):
            gn.run(out, m, h * w_, silu)              # the cache is consumed once and x was never written


