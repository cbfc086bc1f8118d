"""Leaf-op oracles (torch fp32 on CPU) for the HIP kernels — TEST INFRASTRUCTURE ONLY.

Each function states the op exactly as the reference reaches it through torch.nn /
diffusers-0.17.1 / xformers (file:line relative to /root/reference/MD_txt_con_fusion/).
Inputs are taken in the storage dtype (fp16 / bf16), upcast to fp32 and evaluated in fp32,
which is the "fp32 accumulate inside the library kernel" behaviour of the reference's eval
path (SURVEY.md §8c "Numerics of the reference path").
"""
import math

import torch
import torch.nn.functional as F


def f32(t):
    return None if t is None else t.detach().to("cpu", torch.float32)


def linear_ref(a, w, bias=None, a2=None, res=None, rowvec=None, rows_per_inst=1, alpha=1.0,
               geglu=False, silu=False):
    """nn.Linear / 1x1 conv with the fused epilogue of dd_gemm.
    geglu: diffusers GEGLU, `h, gate = proj(x).chunk(2, -1); h * gelu(gate)` (erf gelu)."""
    a = f32(a)
    if a2 is not None:
        a = torch.cat([a, f32(a2)], dim=1)
    y = a @ f32(w).t()
    if bias is not None:
        y = y + f32(bias)
    if geglu:
        h, g = y.chunk(2, dim=-1)
        return h * F.gelu(g)
    if rowvec is not None:
        y = y + f32(rowvec).repeat_interleave(rows_per_inst, dim=0)
    y = alpha * y
    if res is not None:
        y = y + f32(res)
    if silu:
        y = F.silu(y)
    return y


def pack_conv_weight(w):
    """[cout, cin, 3, 3] (torch) -> [cout, 9*cin] with k = (ky*3+kx)*cin + ci (kernel layout)."""
    cout, cin = w.shape[:2]
    return w.permute(0, 2, 3, 1).reshape(cout, 9 * cin).contiguous()


def conv3x3_ref(x_nhwc, w_oihw, bias, m, hin, win, stride=1, up_size=None):
    """nn.Conv2d(k=3, pad=1, stride) on an NHWC batch, optional F.interpolate(nearest, size)
    first (diffusers Upsample2D with explicit output_size,
    networks/unet_2d_condition_multiview.py:369-374,500-501).  Returns NHWC rows."""
    cin = x_nhwc.shape[1]
    x = f32(x_nhwc).reshape(m, hin, win, cin).permute(0, 3, 1, 2)
    if up_size is not None:
        x = F.interpolate(x, size=tuple(up_size), mode="nearest")
    y = F.conv2d(x, f32(w_oihw), f32(bias), stride=stride, padding=1)
    return y.permute(0, 2, 3, 1).reshape(-1, y.shape[1])


def groupnorm_ref(x_nhwc, gamma, beta, m, hw, groups, eps, silu, x2=None):
    """torch.nn.GroupNorm(groups, C, eps) [+ SiLU] on an NHWC batch (ResnetBlock2D norm1/2,
    Transformer2DModel.norm, conv_norm_out)."""
    x = f32(x_nhwc)
    if x2 is not None:
        x = torch.cat([x, f32(x2)], dim=1)
    c = x.shape[1]
    xn = x.reshape(m, hw, c).permute(0, 2, 1)
    y = F.group_norm(xn, groups, f32(gamma), f32(beta), eps)
    if silu:
        y = F.silu(y)
    return y.permute(0, 2, 1).reshape(m * hw, c)


def layernorm_ref(x, gamma, beta, eps=1e-5):
    x = f32(x)
    return F.layer_norm(x, (x.shape[1],), f32(gamma), f32(beta), eps)


def attention_ref(q, k, v, batch, lq, lk, heads, head_dim, scale=None, kv_batch_map=None):
    """xformers.ops.memory_efficient_attention(q, k, v, attn_bias=None, scale) on
    (B, L, H, D) tensors (networks/box_adapter.py:115-156): softmax(q k^T * scale) v."""
    scale = head_dim ** -0.5 if scale is None else scale
    c = heads * head_dim
    qf = f32(q)[:, :c].reshape(batch, lq, heads, head_dim).permute(0, 2, 1, 3)
    kf = f32(k)[:, :c].reshape(-1, lk, heads, head_dim).permute(0, 2, 1, 3)
    vf = f32(v)[:, :c].reshape(-1, lk, heads, head_dim).permute(0, 2, 1, 3)
    if kv_batch_map is not None:
        idx = kv_batch_map.to("cpu", torch.long)
        kf, vf = kf[idx], vf[idx]
    s = (qf @ kf.transpose(-1, -2)) * scale
    o = torch.softmax(s, dim=-1) @ vf
    return o.permute(0, 2, 1, 3).reshape(batch * lq, c)


def timestep_embedding_ref(t, dim, flip_sin_to_cos=True, freq_shift=0.0, max_period=10000):
    """diffusers get_timestep_embedding (Timesteps(320, True, 0),
    networks/unet_addon_rawbox.py:142-144)."""
    half = dim // 2
    exponent = -math.log(max_period) * torch.arange(half, dtype=torch.float32) / (half - freq_shift)
    emb = f32(t)[:, None] * torch.exp(exponent)[None, :]
    emb = torch.cat([torch.sin(emb), torch.cos(emb)], dim=-1)
    if flip_sin_to_cos:
        emb = torch.cat([emb[:, half:], emb[:, :half]], dim=-1)
    return emb


def cfg_ddim_ref(eps, x, coef, guidance):
    """pipeline/pipeline_bev_controlnet.py:487-499 with DDIMScheduler.step (eta=0):
    eps = eps_u + g (eps_c - eps_u); x0 = (x - sqrt(1-a_t) eps)/sqrt(a_t);
    x' = sqrt(a_prev) x0 + sqrt(1 - a_prev) eps."""
    e = f32(eps)
    e = e[0] + guidance * (e[1] - e[0])
    sa_t, s1a_t, sa_p, s1a_p = [float(c) for c in coef]
    x0 = (f32(x) - s1a_t * e) / sa_t
    return sa_p * x0 + s1a_p * e
