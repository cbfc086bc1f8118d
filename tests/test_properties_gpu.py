"""Size-independent properties of the hot kernels at BASELINE's FULL sizes (12 and 48 view-instances of 28x50 latents), where
the CPU oracle would take minutes: exact ones where the arithmetic allows (scaling by a power of two commutes with every
rounding), tolerance-bound ones otherwise.  Complements the oracle comparisons at sizes the oracle finishes in seconds."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DTYPES = [torch.float16, torch.bfloat16]


def r(*shape, seed=0, scale=1.0, dtype=torch.float16):
    g = torch.Generator(device="cuda").manual_seed(seed)
    return (torch.randn(*shape, generator=g, device="cuda") * scale).to(dtype)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("b,l,lk,h,d", [(12, 1400, 1400, 8, 40), (48, 1400, 77, 8, 40), (12, 350, 350, 8, 80), (12, 91, 98, 8, 160)])
def test_attention_constant_values_come_back_exactly(gpu, dtype, b, l, lk, h, d):
    """softmax rows sum to one: if every key of a head carries the SAME value vector, the output is that vector for every
    query whatever Q and K are — exact up to one rounding of the normalisation (covers ragged last key tiles: 1400 = 21 x 64
    + 56, 77, 98)."""
    from dualdiff_amd import ops as O
    c = h * d
    q, k = r(b * l, c, seed=1, dtype=dtype), r(b * lk, c, seed=2, dtype=dtype)
    vrow = r(b, c, seed=3, dtype=dtype)
    v = vrow[:, None, :].expand(b, lk, c).reshape(b * lk, c).contiguous()
    o = O.attention(q, k, v, b, l, lk, h, d, d ** -0.5)
    want = vrow[:, None, :].expand(b, l, c).reshape(b * l, c).float()
    err = (o.float() - want).abs().max().item()
    ulp = 2.0 ** (-10 if dtype == torch.float16 else -7)
    assert torch.isfinite(o.float()).all() and err <= 2 * ulp * want.abs().max().item(), err


@pytest.mark.parametrize("dtype", DTYPES)
def test_attention_is_invariant_to_key_order(gpu, dtype):
    """Permuting the (key, value) pairs of every instance changes only the summation order."""
    from dualdiff_amd import ops as O
    b, l, h, d = 12, 1400, 8, 40
    c = h * d
    q, k, v = (r(b * l, c, seed=s, dtype=dtype) for s in (4, 5, 6))
    perm = torch.randperm(l, generator=torch.Generator().manual_seed(7)).cuda()
    idx = (torch.arange(b, device="cuda")[:, None] * l + perm[None]).reshape(-1)
    o1 = O.attention(q, k, v, b, l, l, h, d, d ** -0.5).float()
    o2 = O.attention(q, k[idx].contiguous(), v[idx].contiguous(), b, l, l, h, d, d ** -0.5).float()
    e = ((o1 - o2).norm() / o1.norm()).item()
    assert e <= (1e-3 if dtype == torch.float16 else 8e-3), e


def _same_up_to_scale(ys, y, s):
    """ys == s * y bit for bit wherever y is a NORMAL number of the storage type with headroom (|y| >= 2^-10: results that
    round into fp16's denormal range keep fewer bits than their scaled twins), and to one denormal step elsewhere."""
    ysf, yf = ys.float(), y.float() * s
    big = y.float().abs() >= 2.0 ** -10
    assert big.float().mean().item() > 0.99
    assert torch.equal(ysf[big], yf[big])
    assert (ysf[~big] - yf[~big]).abs().max().item() <= s * 2.0 ** -24 if (~big).any() else True


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("rows,n,k", [(16800, 320, 320), (16800, 1280, 320), (4200, 640, 640), (1092, 1280, 1280), (336, 1280, 5120)])
def test_gemm_commutes_with_power_of_two_scaling(gpu, dtype, rows, n, k):
    """(4 a) W^T == 4 (a W^T) BIT FOR BIT: scaling by a power of two is exact in the 16-bit storage types and in the fp32
    accumulator, so every product, every partial sum and the final rounding scale with it (no bias, values far from the
    overflow / denormal ranges).  Catches any path that mixes in an unscaled term or rounds somewhere else."""
    from dualdiff_amd import ops as O
    a, w = r(rows, k, seed=8, dtype=dtype), r(n, k, seed=9, scale=k ** -0.5, dtype=dtype)
    y1 = O.gemm(a, w, None)
    y4 = O.gemm((a.float() * 4).to(dtype), w, None)
    _same_up_to_scale(y4, y1, 4.0)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("m,hh,ww,cin,cout", [(12, 28, 50, 320, 320), (12, 14, 25, 640, 640), (12, 4, 7, 1280, 1280), (48, 28, 50, 320, 320)])
def test_conv3x3_commutes_with_power_of_two_scaling_and_is_linear(gpu, dtype, m, hh, ww, cin, cout):
    """conv(2 x) == 2 conv(x) bit for bit (as above), and conv(x1 + x2) ~ conv(x1) + conv(x2) within the rounding of the
    three 16-bit outputs — at the full 12- and 48-instance sizes of the 28x50 level (band form) and the deep levels
    (direct form + split-K)."""
    from dualdiff_amd import ops as O
    x1, x2 = r(m * hh * ww, cin, seed=10, dtype=dtype), r(m * hh * ww, cin, seed=11, dtype=dtype)
    w = r(cout, 9 * cin, seed=12, scale=(9 * cin) ** -0.5, dtype=dtype)
    y1 = O.conv3x3(x1, w, None, m, hh, ww)
    y2 = O.conv3x3(x2, w, None, m, hh, ww)
    _same_up_to_scale(O.conv3x3((x1.float() * 2).to(dtype), w, None, m, hh, ww), y1, 2.0)
    xs = (x1.float() + x2.float()).to(dtype)                      # rounded once; its own rounding error enters the bound
    ys = O.conv3x3(xs, w, None, m, hh, ww).float()
    e = ((ys - (y1.float() + y2.float())).norm() / ys.norm()).item()
    assert e <= (1.5e-3 if dtype == torch.float16 else 1.2e-2), e


@pytest.mark.parametrize("dtype", DTYPES)
def test_cfg_ddim_step_fixed_points(gpu, dtype):
    """CFG + DDIM in one kernel: equal halves make the guidance a no-op whatever the scale, and a zero noise prediction
    scales the latents by sqrt(a_prev / a_t) — on the full 12-instance latent batch."""
    from dualdiff_amd import ops as O
    n = 6 * 4 * 28 * 50
    x = r(6, 4, 28, 50, seed=13, dtype=dtype)
    e = r(6, 4, 28, 50, seed=14, dtype=dtype)
    coef = torch.tensor([0.6, 0.8, 0.7, 0.7141428], device="cuda")
    outs = []
    for gscale in (1.0, 2.0, 7.5):
        xo, xd = torch.empty_like(x), torch.empty_like(x)
        O.cfg_ddim_step(torch.stack([e, e]), x, coef, gscale, x_out=xo, x_dup=xd)
        assert torch.equal(xo, xd)
        outs.append(xo.clone())
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    want = (coef[2] * (x.float() - coef[1] * e.float()) / coef[0] + coef[3] * e.float())
    assert ((outs[0].float() - want).norm() / want.norm()).item() <= (6e-4 if dtype == torch.float16 else 5e-3)
    xo = torch.empty_like(x)
    O.cfg_ddim_step(torch.zeros(2, 6, 4, 28, 50, device="cuda", dtype=dtype), x, coef, 2.0, x_out=xo, x_dup=torch.empty_like(x))
    assert ((xo.float() - x.float() * (0.7 / 0.6)).norm() / x.float().norm()).item() <= (6e-4 if dtype == torch.float16 else 5e-3)
    assert n == x.numel()
