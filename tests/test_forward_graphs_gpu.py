"""HIP graphs behind the PUBLIC forward() surfaces (model_base.ForwardGraphs).

The reference's callers reach the models only through `controlnet(...)` / `unet(...)` (pipeline_bev_controlnet.py:405-446,
476-484); those calls now replay a cached graph per (shapes, scalars, attribute flags).  Checked here, on the full-size
models of the bench workload: a replay equals the eager launch of the same call BIT FOR BIT (same kernels, same tile
table); inputs are refreshed on every call (timestep, latents, residuals); the attribute-poke protocol of
misc/test_utils.py:123-136 (`use_txt_con_fusion` flipped between calls) selects another graph and flipping back reuses
the first; `load_state_dict` / `_invalidate` / a processor swap drop the graphs; a foreign processor runs eagerly.  A key
is recorded the SECOND time it is seen (a loop whose context length changes every batch must not capture per call) and at
most ForwardGraphs.MAX_ENTRIES graphs stay alive per model."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _models(dtype=torch.float16):
    import bench
    dev = torch.device("cuda:0")
    unet, cns = bench.build_models(dtype, dev)
    inputs = bench.synthetic_inputs(1, dtype, dev, seed=11)
    return unet, cns, inputs, dev


def _cn_call(cn, inputs, j, t, lat=None):
    lat0, prompt, cam, boxes, conds = inputs
    lmi = torch.cat([lat0 if lat is None else lat] * 2)
    down, mid, ctx = cn(lmi, t.expand(2), cam, boxes[j], prompt, conds[j], conditioning_scale=1.0, guess_mode=False,
                        return_dict=False, use_aug_text=False)
    return [d.clone() for d in down] + [mid.clone(), ctx.clone()]


def _same(a, b):
    return all(torch.equal(x, y) for x, y in zip(a, b))


def test_controlnet_forward_graph_equals_eager_and_follows_attribute_pokes(gpu):
    unet, cns, inputs, dev = _models()
    cn = cns[0]
    t1, t2 = torch.tensor(981, device=dev), torch.tensor(401, device=dev)
    with torch.no_grad():
        cn.graph_forward = False
        e_t1 = _cn_call(cn, inputs, 0, t1)
        e_t2 = _cn_call(cn, inputs, 0, t2)
        assert cn.__dict__.get("_fwd_graphs") is None
        cn.graph_forward = True
        g_seen = _cn_call(cn, inputs, 0, t1)                    # first sight of the key: eager, nothing recorded
        graphs = cn.__dict__["_fwd_graphs"]
        assert len(graphs.entries) == 0 and _same(g_seen, e_t1)
        g_first = _cn_call(cn, inputs, 0, t1)                   # second sight: records the graph, returns its first replay
        g_again = _cn_call(cn, inputs, 0, t1)                   # pure replay
        g_t2 = _cn_call(cn, inputs, 0, t2)                      # same graph, another timestep
        assert len(graphs.entries) == 1
        assert _same(g_first, e_t1) and _same(g_again, e_t1) and _same(g_t2, e_t2)
        assert not _same(e_t1[:13], e_t2[:13])                   # the timestep does reach the residuals
        # another latent -> same graph, refreshed input
        lat2 = inputs[0] * 0.5
        cn.graph_forward = False
        e_lat = _cn_call(cn, inputs, 0, t1, lat2)
        cn.graph_forward = True
        assert _same(_cn_call(cn, inputs, 0, t1, lat2), e_lat) and len(graphs.entries) == 1
        # attribute poke (misc/test_utils.py:123-136): SFA off -> another key, another graph; back -> the first one again
        cn.use_txt_con_fusion = False
        cn.graph_forward = False
        e_nosfa = _cn_call(cn, inputs, 0, t1)
        cn.graph_forward = True
        _cn_call(cn, inputs, 0, t1)                             # (first sight of the new key)
        g_nosfa = _cn_call(cn, inputs, 0, t1)
        assert _same(g_nosfa, e_nosfa) and not _same(e_nosfa[:13], e_t1[:13]) and len(graphs.entries) == 2
        cn.use_txt_con_fusion = True
        assert _same(_cn_call(cn, inputs, 0, t1), e_t1) and len(graphs.entries) == 2
        # invalidation: new weights -> the old graphs are gone, the result follows the weights
        sd = {k: (v * 1.01 if v.is_floating_point() and v.dim() >= 2 else v) for k, v in cn.state_dict().items()}
        cn.load_state_dict(sd)
        assert len(graphs.entries) == 0
        g_new = _cn_call(cn, inputs, 0, t1)
        cn.graph_forward = False
        assert _same(g_new, _cn_call(cn, inputs, 0, t1)) and not _same(g_new[:13], e_t1[:13])


def test_unet_forward_graph_equals_eager_zero_copy_residuals_and_foreign_processor(gpu):
    from dualdiff_amd.networks.layers import HIPAttnProcessor
    unet, cns, inputs, dev = _models()
    lat0, prompt, cam, boxes, conds = inputs
    cn = cns[0]
    t = torch.tensor(981, device=dev)
    lmi = torch.cat([lat0] * 2)
    x = lmi.reshape(12, *lmi.shape[2:])

    def unet_call(down, mid, ctx, tt=t):
        return unet(x, tt, encoder_hidden_states=ctx, down_block_additional_residuals=down,
                    mid_block_additional_residual=mid).sample.clone()

    with torch.no_grad():
        for _ in range(2):                                      # (the second call records the ControlNet's graph)
            down, mid, ctx = cn(lmi, t.expand(2), cam, boxes[0], prompt, conds[0], conditioning_scale=1.0, guess_mode=False,
                                return_dict=False, use_aug_text=False)      # graph outputs: views of static buffers
        unet.graph_forward = False
        e = unet_call(down, mid, ctx)
        e_plain = unet(x, t, encoder_hidden_states=ctx).sample.clone()
        unet.graph_forward = True
        g0 = unet_call(down, mid, ctx)                          # first sight: eager
        g1 = unet_call(down, mid, ctx)                          # one-branch flow: residuals / tokens read in place
        g2 = unet_call(down, mid, ctx)
        assert torch.equal(g0, e)
        graphs = unet.__dict__["_fwd_graphs"]
        ent = next(iter(graphs.entries.values()))
        assert any(ent["alias"]) and torch.equal(g1, e) and torch.equal(g2, e)
        # the same call with COPIES of the residuals (what a two-branch sum produces): the aliased graph is replaced by
        # one that owns its input buffers, results unchanged
        g3 = unet_call([d.clone() for d in down], mid.clone(), ctx.clone())
        ent = next(iter(graphs.entries.values()))
        assert torch.equal(g3, e) and not any(ent["alias"])
        g4 = unet_call([d * 0.5 for d in down], mid * 0.5, ctx.clone())
        unet.graph_forward = False
        assert torch.equal(g4, unet_call([d * 0.5 for d in down], mid * 0.5, ctx.clone()))
        unet.graph_forward = True
        # no residuals: another key
        assert torch.equal(unet(x, t, encoder_hidden_states=ctx).sample, e_plain) and len(graphs.entries) == 1
        assert torch.equal(unet(x, t, encoder_hidden_states=ctx).sample, e_plain) and len(graphs.entries) == 2
        # the cap: the least recently used graph goes when a new key would exceed it
        cap = type(graphs).MAX_ENTRIES
        type(graphs).MAX_ENTRIES = 2
        try:
            ctx_short = ctx[:, :40].contiguous()                 # another context length: a third key
            for _ in range(2):
                y3 = unet(x, t, encoder_hidden_states=ctx_short).sample
            assert bool(torch.isfinite(y3.float()).all()) and len(graphs.entries) == 2
        finally:
            type(graphs).MAX_ENTRIES = cap
        # python-number timestep
        assert torch.equal(unet(x, 981, encoder_hidden_states=ctx).sample, e_plain)

        class Foreign:                                          # not a built-in type: forward() must run eagerly
            calls = 0

            def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, temb=None, **kw):
                Foreign.calls += 1
                return HIPAttnProcessor()(attn, hidden_states, encoder_hidden_states, attention_mask, temb, **kw)
        blk = unet.mid_block.attentions[0].transformer_blocks[0]
        blk.attn1.set_processor(Foreign())
        assert unet._graphs() is None
        y = unet(x, t, encoder_hidden_states=ctx).sample
        n1 = Foreign.calls
        y2 = unet(x, t, encoder_hidden_states=ctx).sample
        # through the processor protocol the block fuses differently (no bit equality with the built-in path)
        assert n1 >= 1 and Foreign.calls == 2 * n1 and torch.equal(y, y2)
        assert ((y.float() - e_plain.float()).norm() / e_plain.float().norm()).item() < 5e-3
        blk.attn1.set_processor(HIPAttnProcessor())
        assert unet._graphs() is not None
        assert torch.equal(unet(x, t, encoder_hidden_states=ctx).sample, e_plain) and Foreign.calls == 2 * n1
