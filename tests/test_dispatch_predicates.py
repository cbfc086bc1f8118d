"""Dispatch predicates at the reference's OTHER resolutions, without a GPU.

The tuned table and several kernel preconditions were derived on the 28x50 pyramid (configs/exp/*: 224x400).  The
reference also ships 256x704 (32x88 latents, configs/exp-hd/256x704.yaml:11), 432x768 (54x96, exp-hd/432x768.yaml:11) and
192x384 (24x48, exp-drive-wm/192x384.yaml).  Every launcher validates and plans on the host before its first HIP call:
dd_gemm_kernel_name / dd_attention_kernel_name report the plan without launching, and a rejected dd_xattn320 /
dd_attention call returns before touching the device.  Here: the plans at those pyramids are either a valid kernel or an
explicit "unsupported" (never a mis-launch), and the Python dispatchers fall back accordingly."""
import ctypes

import pytest
import torch

from dualdiff_amd import _native

LEVELS = {"224x400": [(28, 50), (14, 25), (7, 13), (4, 7)], "256x704": [(32, 88), (16, 44), (8, 22), (4, 11)],
          "432x768": [(54, 96), (27, 48), (14, 24), (7, 12)], "192x384": [(24, 48), (12, 24), (6, 12), (3, 6)]}
CH = [320, 640, 1280, 1280]


def _conv_desc(m, h, w, cin, cout, tile):
    d = _native.GemmDesc()
    d.a = d.w = d.out = 4096                      # aligned dummies: nothing is dereferenced by the planner
    d.rows, d.n, d.k, d.k1 = m * h * w, cout, 9 * cin, 9 * cin
    d.lda, d.ldc = cin, cout
    d.alpha = 1.0
    d.dtype = 0
    d.conv = 1
    d.cin, d.hin, d.hv, d.hout, d.win, d.wv, d.wout, d.stride = cin, h, h, h, w, w, w, 1
    d.tile = tile
    return d


@pytest.mark.parametrize("res", sorted(LEVELS))
def test_direct_and_band_conv_plans_reject_what_does_not_fit(res):
    """Direct conv (tile 31: whole instances, H*W <= 384 rows) and its band form (tile 39: whole image rows + a halo of
    W + 1 pixels on either side inside 472 LDS rows) at every level of every shipped resolution, 6 view-instances."""
    lib = _native.load()
    for (h, w), c in zip(LEVELS[res], CH):
        direct = lib.dd_gemm_kernel_name(ctypes.byref(_conv_desc(6, h, w, c, c, 31))).decode()
        band = lib.dd_gemm_kernel_name(ctypes.byref(_conv_desc(6, h, w, c, c, 39))).decode()
        fits_direct = h * w <= 384
        rows_per_band = min(384, 472 - 16 - 2 * (w + 1)) // w
        fits_band = h * w > 384 and rows_per_band >= 1
        assert direct.startswith("dd_conv3s_kernel") == fits_direct and (fits_direct or direct == "unsupported"), (res, h, w, direct)
        assert band.startswith("dd_conv3s_kernel") == fits_band and (fits_band or band == "unsupported"), (res, h, w, band)
        if fits_band:                                            # whole rows per band, every pixel covered once
            grid = band.split("grid=")[1].split(" ")[0]
            tiles_m = int(grid.split("x")[0])
            assert tiles_m == 6 * -(-h // rows_per_band), (res, h, w, band)
        # the implicit-GEMM family takes every shape: the dispatcher's fallback always exists
        generic = lib.dd_gemm_kernel_name(ctypes.byref(_conv_desc(6, h, w, c, c, 12))).decode()
        assert generic.startswith("dd_gemm2_kernel"), generic
    # an image WIDER than the band buffer (56 x 100 map level of the condition embedder: wider than the slab)
    wide = lib.dd_gemm_kernel_name(ctypes.byref(_conv_desc(6, 112, 220, 64, 64, 39))).decode()
    assert wide == "unsupported"
    # pipelined dense tiles never take a convolution
    assert lib.dd_gemm_kernel_name(ctypes.byref(_conv_desc(6, 14, 25, 640, 640, 72))).decode() == "unsupported"


def _attn_desc(batch, lq, lk, heads, d, ldk=None, variant=0, prescaled=0):
    a = _native.AttnDesc()
    a.q = a.k = a.v = a.o = 4096
    c = heads * d
    a.ldq = a.ldo = c
    a.ldk = a.ldv = ldk or c
    a.q_batch_stride, a.o_batch_stride = lq * c, lq * c
    a.k_batch_stride = a.v_batch_stride = lk * (ldk or c)
    a.batch, a.heads, a.head_dim, a.lq, a.lk = batch, heads, d, lq, lk
    a.scale, a.dtype, a.variant, a.q_prescaled = d ** -0.5, 0, variant, prescaled
    return a


def test_attention_plan_follows_the_shape_at_every_resolution():
    lib = _native.load()
    name = lambda *a, **k: lib.dd_attention_kernel_name(ctypes.byref(_attn_desc(*a, **k))).decode()
    # the bench shape: 12 x 8 heads x 1400 rows -> 48 rows per wave on 768 slots, 64-key tiles when q is prescaled
    assert name(12, 1400, 1400, 8, 40, prescaled=1).startswith("dd_attn5_kernel<_Float16, 40, 3, 64, 1, 3, true, false> grid=768")
    assert "40, 3, 128, 1, 3, false" in name(12, 1400, 1400, 8, 40)
    for res, levels in LEVELS.items():
        for (h, w), c in zip(levels, CH):
            for m in (6, 12):
                n = name(m, h * w, h * w, 8, c // 8)
                assert n.startswith("dd_attn5_kernel<_Float16, %d, " % (c // 8)), (res, h, w, n)
                qt = int(n.split(", ")[2])
                grid = int(n.split("grid=")[1])
                assert qt in (1, 2, 3) and grid == -(-h * w // (64 * qt)) * m * 8, (res, h, w, n)
                if h * w < 256:
                    assert qt == 1, n                             # short sequences: 16 rows per wave
    # text cross-attention: 77 + 1 + boxes keys
    assert "80, 1, 64" in name(12, 350, 98, 8, 80)
    # rejections instead of mis-launches: retired variants, K/V planes beyond the 32-bit buffer offsets, bad head dim
    assert name(12, 1400, 1400, 8, 40, variant=7) == "unsupported"
    assert name(1, 64, 70000, 8, 40, ldk=16384) == "unsupported"
    bad = _attn_desc(12, 350, 350, 8, 80)
    bad.head_dim = 64
    assert lib.dd_attention_kernel_name(ctypes.byref(bad)).decode() == "unsupported"
    bad = _attn_desc(12, 350, 350, 8, 80)
    bad.ldk = 636                                                # not a multiple of 8
    assert lib.dd_attention_kernel_name(ctypes.byref(bad)).decode() == "invalid"
    assert lib.dd_attention(ctypes.byref(bad), None) == -1       # rejected before any device call


def test_xattn320_predicate_and_validation():
    from dualdiff_amd import ops as O
    assert O.xattn320_ok(320, 8, 98, 12 * 1400)                 # 210 workgroups of 80 rows: one generation
    assert not O.xattn320_ok(320, 8, 98, 48 * 1400)             # 840: the three-launch form
    assert O.xattn320_ok(320, 8, 98, 6 * 54 * 96) == ((6 * 54 * 96 + 79) // 80 <= O.XATTN_MAX_WGS)
    assert not O.xattn320_ok(320, 8, 129) and not O.xattn320_ok(640, 8, 77) and not O.xattn320_ok(320, 5, 77)
    lib = _native.load()
    d = _native.XAttnDesc()
    d.x = d.wq = d.wo = d.bo = d.k = d.v = d.out = 4096
    d.instances, d.rows_per_inst, d.lk, d.channels, d.heads = 6, 54 * 96, 200, 320, 8
    d.ldx = d.ldo = 320
    d.ldk = d.ldv = 640
    assert lib.dd_xattn320(ctypes.byref(d), None) == -2         # 200 keys: unsupported, nothing launched
    d.lk, d.channels = 98, 640
    assert lib.dd_xattn320(ctypes.byref(d), None) == -2


def test_gemm_plans_exist_for_every_level_of_every_resolution():
    """Whatever the tuner finds in the table, the C-side heuristic (tile 0) must yield a launchable plan for the dense
    shapes of these pyramids, incl. K tails that rule out the LDS-DMA family."""
    lib = _native.load()
    for res, levels in LEVELS.items():
        for (h, w), c in zip(levels, CH):
            for n, k in ((c, c), (3 * c, c), (c, 4 * c), (2 * c, 768)):
                d = _native.GemmDesc()
                d.a = d.w = d.out = 4096
                d.rows, d.n, d.k, d.k1 = 6 * h * w, n, k, k
                d.lda, d.ldc, d.alpha, d.dtype = k, n, 1.0, 0
                nm = lib.dd_gemm_kernel_name(ctypes.byref(d)).decode()
                assert nm.startswith("dd_gemm"), (res, h, w, n, k, nm)
                for tile in (72, 73, 74, 75, 76, 77, 78):
                    d.tile = tile
                    nm = lib.dd_gemm_kernel_name(ctypes.byref(d)).decode()
                    # one tile per workgroup, or (round 6) the persistent walk when the grid exceeds one generation
                    # of resident workgroups and the K loop is at least as long as the ring
                    tiles = [int(v) for v in nm.split("grid=")[1].split(" ")[0].split("x")]
                    depth = {72: 3, 73: 5, 75: 3, 78: 3}.get(tile)
                    many = depth is not None and tiles[0] * tiles[1] > 256 * (2 if tile == 72 else 1) and k // 64 >= depth
                    assert nm.startswith("dd_gemm4_kernel" if many else "dd_gemm3_kernel"), (tile, nm)
                d.tile = 0


def test_layernorm_fold_plans_without_a_tuned_tile():
    """with `ln_colsum` set and no tile named (DD_AUTOTUNE=0, or an untuned shape first met inside a capture)
    the planner maps the heuristic's register-staged pick onto its LDS-DMA twin BY SHAPE; the fixed id table it used
    pointed at two tiles round 5 had removed, so 64x64 / 128x64 picks returned DD_ERR_UNSUPPORTED."""
    lib = _native.load()
    for rows in (96, 336, 1092, 4200, 16800, 67200):
        for k in (320, 640, 1280):
            for n in (k, 3 * k):
                d = _native.GemmDesc()
                d.a = d.w = d.out = d.ln_colsum = d.ln_bias = 4096
                d.rows, d.n, d.k, d.k1 = rows, n, k, k
                d.lda, d.ldc = k, n
                d.alpha, d.ln_eps = 1.0, 1e-5
                d.dtype, d.tile = 0, 0
                name = lib.dd_gemm_kernel_name(ctypes.byref(d)).decode()
                assert name.startswith("dd_gemm2_kernel<"), (rows, n, k, name)
            g = _native.GemmDesc()                               # the GEGLU projection (8C wide, gate in the epilogue)
            g.a = g.w = g.out = g.ln_colsum = g.ln_bias = 4096
            g.rows, g.n, g.k, g.k1 = rows, 4 * k, k, k
            g.lda, g.ldc = k, 4 * k
            g.alpha, g.ln_eps, g.epilogue = 1.0, 1e-5, 1
            name = lib.dd_gemm_kernel_name(ctypes.byref(g)).decode()
            assert name.startswith("dd_gemm2_kernel<") and name.split(">")[0].endswith("true"), (rows, k, name)
