"""Token / condition preparation kernels (csrc/tokens.hip) against the tensor-op chains they replace — BIT-exact: these
kernels only move, select and round values (reference: unet_addon_rawbox.py:308-361,832-896,1007; bbox_embedder.py:164-203;
map_embedder.py:116-125)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DTYPES = [torch.float16, torch.bfloat16]


def r(*shape, seed=0, scale=1.0, dtype=torch.float32):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dtype).cuda()


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("m,c,h,w,views,cpad", [(2, 3, 224, 400, 6, 8), (12, 4, 28, 50, 1, 8), (12, 320, 28, 50, 1, None),
                                               (3, 5, 7, 13, 2, 16), (2, 72, 9, 11, 1, 80), (1, 130, 4, 7, 3, 136), (2, 4, 16, 16, 1, None), (2, 5, 7, 9, 1, 6)])
def test_nchw_to_nhwc_views(gpu, dtype, m, c, h, w, views, cpad):
    from dualdiff_amd import ops as O
    x = r(m, c, h, views * w, seed=1, dtype=dtype)
    y = O.nchw_to_nhwc(x, cpad, views=views)
    cp = (cpad if cpad is not None else c)
    ref = x.reshape(m, c, h, views, w).permute(0, 3, 2, 4, 1).reshape(m * views * h * w, c)     # b, view, h, w, c
    assert y.shape == (m * views * h * w, cp)
    assert torch.equal(y[:, :c], ref) and (cp == c or not y[:, c:].any())


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("in_dtype", [torch.float16, torch.bfloat16, torch.float32])
def test_camera_features_match_the_embedder_chain(gpu, dtype, in_dtype):
    from dualdiff_amd import ops as O
    from dualdiff_amd.networks.embedder import get_embedder
    fe = get_embedder(3, 4)
    cam = r(2, 6, 3, 7, seed=2, dtype=in_dtype)
    ref = fe(cam.permute(0, 1, 3, 2)).reshape(12, -1).to(dtype)                                    # (12, 189)
    out = O.camera_features(cam, fe.freq_bands, fe.include_input, dtype, 192)
    assert out.shape == (12, 192) and torch.equal(out[:, :189], ref) and not out[:, 189:].any()


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("pts_dtype", [torch.float32, None])
@pytest.mark.parametrize("normalize,with_mask,want_cls", [(False, True, False), (True, True, True), (False, False, True)])
def test_bbox_embedder_fused_equals_tensor_ops(gpu, dtype, pts_dtype, normalize, with_mask, want_cls):
    """ContinuousBBoxWithTextEmbedding.forward: the one-launch operand preparation against the tensor-op chain of the
    same module, bit for bit (tokens and, for the box adapter, the class embeddings)."""
    from dualdiff_amd.networks import bbox_embedder as B
    from oracle.init_utils import seeded_state_dict
    net = B.ContinuousBBoxWithTextEmbedding(n_classes=10, mode="all-xyz", minmax_normalize=normalize, use_text_encoder_init=False)
    net.load_state_dict(seeded_state_dict(net, 13))
    net = net.to("cuda", dtype).eval()
    g = torch.Generator().manual_seed(5)
    bb = ((torch.rand((12, 20, 8, 3), generator=g) - 0.5) * 100.0).cuda().to(pts_dtype or dtype)
    cl = torch.randint(0, 10, (12, 20), generator=g).cuda()
    mk = (torch.rand((12, 20), generator=g) > 0.3).cuda() if with_mask else None
    outs = {}
    with torch.no_grad():
        for fused in (True, False):
            B.FUSED_BOX_TOKENS = fused
            try:
                outs[fused] = net(bb, cl, mk, return_cls_emb=want_cls)
            finally:
                B.FUSED_BOX_TOKENS = True
    a, b_ = outs[True], outs[False]
    if want_cls:
        assert torch.equal(a[1], b_[1])
        a, b_ = a[0], b_[0]
    assert a.shape == (12, 20, 768) and a.dtype == dtype and torch.equal(a, b_)


def test_box_tokens_class_index_is_bounds_checked(gpu):
    """a kept box whose class index is the dataset's padding value (-1) or >= n_classes must not read foreign
    memory.  -1 wraps to the last class like torch indexing; what stays out of range gives NaN class tokens (torch would
    raise a device assert); masked-out rows ignore their class index altogether."""
    from dualdiff_amd import ops as O
    dtype, rows, npts, nf, ctd, ncls = torch.float16, 6, 8, 4, 768, 10
    g = torch.Generator().manual_seed(3)
    pts = torch.rand((rows, npts, 3), generator=g).cuda().to(dtype)
    table = torch.randn((ncls, ctd), generator=g).cuda().to(dtype)
    null_pos = torch.randn((npts * 3 * (1 + 2 * nf),), generator=g).cuda().to(dtype)
    null_cls = torch.randn((ctd,), generator=g).cuda().to(dtype)
    classes = torch.tensor([3, -1, 10, 9, 12345, -11], device="cuda")
    masks = torch.tensor([True, True, True, True, False, False], device="cuda")
    pos = torch.empty((rows, npts * 3 * (1 + 2 * nf)), dtype=dtype, device="cuda")
    cat = torch.zeros((rows, 2 * ctd), dtype=dtype, device="cuda")
    freqs = [2.0 ** i for i in range(nf)]
    O.box_tokens(pts, classes, masks, table, null_pos, null_cls, freqs, True, pos, cat, ctd)
    torch.cuda.synchronize()
    got = cat[:, ctd:]
    assert torch.equal(got[0], table[3]) and torch.equal(got[1], table[9]) and torch.equal(got[3], table[9])
    assert torch.isnan(got[2].float()).all()                     # 10 is one past the table
    assert torch.equal(got[4], null_cls) and torch.equal(got[5], null_cls)
    assert torch.isfinite(pos.float()).all()


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("box_views,text_per_view,nbox", [(6, False, 20), (1, False, 5), (6, True, 3), (6, False, 0)])
def test_ctx_assemble(gpu, dtype, box_views, text_per_view, nbox):
    from dualdiff_amd import ops as O
    scenes, n_cam, lt, dim = 2, 6, 77, 768
    m = scenes * n_cam
    cam = r(m, dim, seed=3, dtype=dtype)
    text = r(m if text_per_view else scenes, lt, dim, seed=4, dtype=dtype)
    box = r(scenes * box_views, nbox, dim, seed=5, dtype=dtype) if nbox else None
    full, txt = O.ctx_assemble(cam, text, box, n_cam, text_per_view=text_per_view)
    e = text if text_per_view else text[:, None].expand(-1, n_cam, -1, -1).reshape(m, lt, dim)
    parts = [cam[:, None], e]
    if nbox:
        parts.append(box.reshape(scenes, box_views, nbox, dim).expand(-1, n_cam, -1, -1).reshape(m, nbox, dim))
    assert torch.equal(full, torch.cat(parts, dim=1)) and torch.equal(txt, e)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shared_boxes,adapter", [(False, False), (True, False), (False, True)])
def test_prepare_tokens_fused_equals_tensor_ops(gpu, dtype, shared_boxes, adapter):
    """BEVControlNetModel.prepare_tokens: four launches against the tensor-op chain — ctx, ctx2d, txt (and the adapter's
    class-token context) bit for bit."""
    from dualdiff_amd.networks import unet_addon_rawbox as U
    from dualdiff_amd.networks.layers import device_init_
    with torch.device("cuda"):
        net = U.BEVControlNetModel(cross_attention_dim=768).to(dtype)
    device_init_(net, 3)
    net.use_box_adapter = adapter
    if adapter:
        from dualdiff_amd.networks.box_adapter import box_adapter
        box_adapter(net)
        net = net.to("cuda", dtype)
    g = torch.Generator().manual_seed(9)
    b, n_cam, nb = 2, 6, 1 if shared_boxes else 6
    cam = r(b, n_cam, 3, 7, seed=6, dtype=dtype)
    text = r(b, 77, 768, seed=7, dtype=dtype)
    boxes = {"bboxes": ((torch.rand((b, nb, 20, 8, 3), generator=g) - 0.5) * 100.0).cuda().to(dtype),
             "classes": torch.randint(0, 10, (b, nb, 20), generator=g).cuda(),
             "masks": (torch.rand((b, nb, 20), generator=g) > 0.3).cuda()}
    outs = {}
    with torch.no_grad():
        for fused in (True, False):
            U.FUSED_TOKENS = fused
            try:
                outs[fused] = net.prepare_tokens(cam, boxes, text)
            finally:
                U.FUSED_TOKENS = True
    a, b_ = outs[True], outs[False]
    assert a["lc"] == b_["lc"] == 98 and a["m"] == b_["m"] == 12
    for k in ("ctx", "ctx2d", "txt") + (("ctx2d_cn",) if adapter else ()):
        assert a[k].is_contiguous() and torch.equal(a[k], b_[k]), k
