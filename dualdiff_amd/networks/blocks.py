"""`BasicMultiviewTransformerBlock` on the HIP kernels.

Mirrors /root/reference/MD_txt_con_fusion/magicdrive/networks/blocks.py:35-238
(neighboring_attn_type="add", zero_module_type="zero_linear"): self-attention, text/box
cross-attention, neighbour-view attention `attn4` + `connector`, GEGLU feed-forward.

attn4 (blocks.py:106-142,190-222) is computed without the reference's per-pair recomputation:
Q/K/V of every view are projected ONCE by a fused GEMM; the left-neighbour and right-neighbour
attentions read the neighbours' K/V in place through `kv_batch_map` and the second call
accumulates into the first (`out = Attn(q_v, kv_left) + Attn(q_v, kv_right)`); because `to_out`
is linear, `sum_u to_out(o_u) = W_o (o_L + o_R) + 2 b_o` (the bias enters once per pair,
blocks.py:203-217) — one out-projection GEMM with the bias pre-scaled by the neighbour count.
"""
import torch
import torch.nn as nn

from .. import ops as O
from .._native import Unsupported
from . import layers
from .layers import Attention, BasicTransformerBlock, LayerNorm, Linear, want_ln_stats


def _ensure_kv_is_int(view_pair):
    """JSON configs carry string keys (blocks.py:14-21)."""
    return {int(k): [int(x) for x in v] for k, v in view_pair.items()}


# Both neighbours of a view in ONE attention launch (dd_attn_desc.kv_batch_map2): two softmax passes inside the kernel,
# summed in fp32, instead of two launches of which the second reads O back to accumulate.  DD_ATTN4_PAIR=0: two launches.
ATTN4_PAIR = __import__("os").environ.get("DD_ATTN4_PAIR", "1") != "0"


def _neighbour_sum(q, k, v, batch, l, heads, dim_head, scale, maps, prescaled):
    """sum_j Attn(q, kv[maps[j]]) (blocks.py:203-217): pairs of neighbours per launch, a single one accumulates.  The
    library refuses the pair form (DD_ERR_UNSUPPORTED, nothing launched) where its 32-bit buffer offsets do not fit or
    another kernel variant is forced; those calls take the two-launch accumulate form."""
    o, j = None, 0
    while j < len(maps):
        pair = ATTN4_PAIR and j + 1 < len(maps)
        if pair:
            try:
                o = O.attention(q, k, v, batch, l, l, heads, dim_head, scale, kv_batch_map=maps[j],
                                kv_batch_map2=maps[j + 1], out=o, accumulate=j > 0, q_prescaled=prescaled)
                j += 2
                continue
            except Unsupported:
                pass
        o = O.attention(q, k, v, batch, l, l, heads, dim_head, scale, kv_batch_map=maps[j], out=o, accumulate=j > 0,
                        q_prescaled=prescaled)
        j += 1
    return o


class GatedConnector(nn.Module):
    """tanh(alpha) * x with a zero-initialised per-channel alpha (blocks.py:24-32); on the HIP path the gate is
    folded into the out-projection weights like the Linear connector."""

    def __init__(self, dim):
        super().__init__()
        self.alpha = nn.Parameter(torch.zeros(dim))


class BasicMultiviewTransformerBlock(BasicTransformerBlock):
    def __init__(self, dim, num_attention_heads, attention_head_dim, cross_attention_dim=None,
                 neighboring_view_pair=None, neighboring_attn_type="add", zero_module_type="zero_linear",
                 **unused):
        super().__init__(dim, num_attention_heads, attention_head_dim, cross_attention_dim=cross_attention_dim)
        if neighboring_attn_type not in ("add", "concat", "self"):
            raise NotImplementedError("Unknown type: %s" % neighboring_attn_type)          # blocks.py:140-142
        self.neighboring_view_pair = _ensure_kv_is_int(neighboring_view_pair)
        self.neighboring_attn_type = neighboring_attn_type
        self.zero_module_type = zero_module_type
        self.norm4 = LayerNorm(dim)
        self.attn4 = Attention(dim, dim, num_attention_heads, attention_head_dim)
        if zero_module_type == "zero_linear":                 # blocks.py:81-90
            self.connector = Linear(dim, dim)
        elif zero_module_type == "gated":
            self.connector = GatedConnector(dim)
        elif zero_module_type == "none":
            self.connector = None
        else:
            raise TypeError("Unknown zero module type: %s" % zero_module_type)
        self._maps = {}
        # dualdiff_amd.parallel.ViewShard when the views of a scene are spread over several GPUs
        # (UNet2DConditionModelMultiview.set_view_shard); None = all n_cam views are local
        self.view_shard = None

    # `connector(to_out(o))` is two linear maps with nothing in between (blocks.py:203-222): they always run as
    # ONE GEMM with folded weights (_folded_out).
    fold_connector = True

    def _folded_out(self, nb):
        """(W, b) of `connector(sum of nb to_out applications)` as ONE Linear on the summed attention outputs:
        zero_linear: W = Wc Wo, b = Wc (nb b_o) + b_c; gated: W = diag(tanh a) Wo, b = tanh(a) * nb b_o;
        none: W = Wo, b = nb b_o.  Folded in fp32 whenever a parameter changes."""
        wo, bo = self.attn4.to_out[0].weight, self.attn4.to_out[0].bias
        cp = [] if self.connector is None else list(self.connector.parameters())
        key = (nb, wo._version, bo._version, wo.data_ptr()) + tuple((t._version, t.data_ptr()) for t in cp)
        holder = self.attn4.to_out[0].__dict__
        hit = holder.get("_pk_fold")
        if hit is None or hit[0] != key:
            with torch.no_grad():
                wof, bof = wo.detach().float(), bo.detach().float() * nb
                if self.zero_module_type == "zero_linear":
                    wc, bc = self.connector.weight.detach().float(), self.connector.bias.detach().float()
                    w, b = wc @ wof, wc @ bof + bc
                elif self.zero_module_type == "gated":
                    g = torch.tanh(self.connector.alpha.detach().float())
                    w, b = g[:, None] * wof, g * bof
                else:
                    w, b = wof, bof
                hit = holder["_pk_fold"] = (key, w.to(wo.dtype).contiguous(), b.to(wo.dtype).contiguous())
        return hit[1], hit[2]

    @property
    def n_cam(self):
        return len(self.neighboring_view_pair)

    @property
    def new_module(self):
        ret = {"norm4": self.norm4, "attn4": self.attn4}
        if self.connector is not None:
            ret["connector"] = self.connector
        return ret

    def neighbour_maps(self, batch, device):
        """int32 [batch] maps: instance b*n_cam+v -> instance of its k-th neighbour."""
        key = (batch, str(device))
        if key not in self._maps:
            n = self.n_cam
            depth = max(len(v) for v in self.neighboring_view_pair.values())
            maps = []
            for j in range(depth):
                idx = [(i // n) * n + self.neighboring_view_pair[i % n][j] for i in range(batch)]
                maps.append(torch.tensor(idx, dtype=torch.int32, device=device))
            self._maps[key] = maps
        return self._maps[key]

    def _attn4_sharded(self, h, batch, l):
        """attn4 when this rank holds only `n_local` of the n_cam views (SURVEY §8e view split): Q/K/V of
        the local views are projected as usual; their K/V go into the first slots of a slot-major buffer
        (slot, half/scene, [K heads | V heads], l, d) whose remaining slots the exchange fills with the
        neighbours' K/V from other ranks — a view's K/V of one level is one contiguous message — and the
        two neighbour attentions read the buffer through kv_batch_map exactly as in the unsharded case
        (same kernel, same per-(instance, head) K/V values -> the same bits)."""
        a, sh = self.attn4, self.view_shard
        hd, d = a.heads, a.dim_head
        if not (layers.HEAD_MAJOR and a.to_q.bias is None):
            raise NotImplementedError("view split needs the head-major bias-free attn4 projection")
        nloc = sh.n_local
        if batch % nloc:
            raise ValueError("%d instances are not a multiple of the %d local views" % (batch, nloc))
        nbat = batch // nloc
        qkv = a.project_qkv(h, self.norm4, head_major=True)                       # (3 * hd, batch * l, d)
        kv = torch.empty((sh.plan.n_slots, nbat, 2 * hd, l, d), dtype=h.dtype, device=h.device)
        kv[:nloc].copy_(qkv[hd:].reshape(2 * hd, nbat, nloc, l, d).permute(2, 1, 0, 3, 4))
        sh.exchange(kv)
        flat = kv.reshape(sh.plan.n_slots * nbat, 2 * hd, l, d)
        k4, v4 = flat[:, :hd], flat[:, hd:]
        maps = sh.maps(nbat, h.device)
        o = _neighbour_sum(qkv[:hd], k4, v4, batch, l, hd, d, a.scale, maps, True)
        return o, len(maps)

    def _cross_view(self, h, batch, l):
        """norm4 -> attn4 over the neighbour views -> connector -> + h (blocks.py:106-142,190-222)."""
        a = self.attn4
        c, hd, d = a.inner_dim, a.heads, a.dim_head
        kind = self.neighboring_attn_type
        if kind != "add" and not (layers.HEAD_MAJOR and a.to_q.bias is None):
            raise NotImplementedError("neighboring_attn_type='%s' needs the head-major bias-free projection" % kind)
        if kind == "self":                            # :135-139: ONE self-attention over all views of a scene
            if self.view_shard is not None:
                raise NotImplementedError("neighboring_attn_type='self' attends to every view: not view-shardable")
            qkv = a.project_qkv(h, self.norm4, head_major=True)
            o = O.attention(qkv[:hd], qkv[hd:2 * hd], qkv[2 * hd:], batch // self.n_cam, self.n_cam * l,
                            self.n_cam * l, hd, d, q_prescaled=True)
            nb = 1
        elif kind == "concat":                        # :122-134: the neighbours' tokens as ONE key sequence
            qkv = a.project_qkv(h, self.norm4, head_major=True)
            if self.view_shard is not None:
                sh = self.view_shard
                nloc = sh.n_local
                nbat = batch // nloc
                kv = torch.empty((sh.plan.n_slots, nbat, 2 * hd, l, d), dtype=h.dtype, device=h.device)
                kv[:nloc].copy_(qkv[hd:].reshape(2 * hd, nbat, nloc, l, d).permute(2, 1, 0, 3, 4))
                sh.exchange(kv)
                src = kv.reshape(sh.plan.n_slots * nbat, 2 * hd, l, d)
                maps = sh.maps(nbat, h.device)
            else:
                src = qkv[hd:].reshape(2 * hd, batch, l, d).permute(1, 0, 2, 3)      # (instance, 2 hd, l, d) view
                maps = self.neighbour_maps(batch, h.device)
            cat = torch.empty((batch, 2 * hd, len(maps), l, d), dtype=h.dtype, device=h.device)
            for j, mp in enumerate(maps):
                cat[:, :, j] = src.index_select(0, mp.long())
            cat = cat.reshape(batch, 2 * hd, len(maps) * l, d)
            o = O.attention(qkv[:hd], cat[:, :hd], cat[:, hd:], batch, l, len(maps) * l, hd, d, q_prescaled=True)
            nb = 1
        elif self.view_shard is not None:
            o, nb = self._attn4_sharded(h, batch, l)
        else:
            hm = layers.HEAD_MAJOR and a.to_q.bias is None
            qkv = a.project_qkv(h, self.norm4, head_major=hm)
            q, k, v = ((qkv[:a.heads], qkv[a.heads:2 * a.heads], qkv[2 * a.heads:]) if hm
                       else (qkv[:, :c], qkv[:, c:2 * c], qkv[:, 2 * c:]))
            maps = self.neighbour_maps(batch, h.device)
            o = _neighbour_sum(q, k, v, batch, l, a.heads, a.dim_head, a.scale, maps, hm)
            nb = len(maps)
        # connector(to_out(...)) as one GEMM with the residual add (folded weights, see _folded_out)
        w, b = self._folded_out(nb)
        if layers.LN_PRODUCER and w.shape[0] == 320 and layers.LN_FOLD == "0":   # emits norm3(out) as well
            out = O.gemm(o, w, b, res=h, ln_out=(self.norm3.weight, self.norm3.bias, self.norm3.eps))
            out._ln_cache = (self.norm3, out._ln_out)
            return out
        return O.gemm(o, w, b, res=h, ln_stats=want_ln_stats())        # feeds norm3

    def run(self, h, batch, l, ctx2d, lc, defer_ff_out=False):
        h = self._attn(self.attn1, self.norm1, h, batch, l, next_norm=self.norm2)
        h = self._attn(self.attn2, self.norm2, h, batch, l, ctx2d, lc, next_norm=self.norm4)
        h = self._cross_view(h, batch, l)
        # ---- feed-forward ------------------------------------------------------------------
        if defer_ff_out:
            return self.ff.run(h, norm=self.norm3, defer_out=True), h
        return self.ff.run(h, res=h, norm=self.norm3)
