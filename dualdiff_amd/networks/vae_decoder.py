"""VAE decode on the HIP path — the step after the denoising loop.

Mirrors what the reference's pipeline calls at pipeline/pipeline_bev_controlnet.py:101-113,532:
`decode_latents(latents)` = `vae.decode(latents / scaling_factor).sample` per view, then `/ 2 + 0.5` and
clamp.  `AutoencoderKLDecoder` carries diffusers' parameter names for the decode half of `AutoencoderKL`
(`post_quant_conv`, `decoder.conv_in`, `decoder.mid_block.{resnets,attentions}`, `decoder.up_blocks.*`,
`decoder.conv_norm_out`, `decoder.conv_out`), so the stock SD-v1.5 `vae/diffusion_pytorch_model.bin` loads
with `load_state_dict(strict=False)` (the encoder keys are simply unused).

Everything runs on the kernels of the denoising path (NHWC implicit-GEMM 3x3 convs with the nearest-x2
upsampling fused into the gather, fused GroupNorm+SiLU, GEMM epilogues for bias / residual) at up to
6 x 224 x 400 pixels.  The one new piece is the mid block's single 512-wide attention head over the
h*w = 1400 latent tokens: the flash kernel is specialised for the UNet's head dims (40 / 80 / 160), and
this attention is 25 GFLOP once per sample, so it is three launches per view — logits GEMM with fp32
output (`out_f32`), row softmax (`dd_softmax_rows`), P @ V GEMM against V^T produced directly by a GEMM
with swapped operands; the value bias is added after P @ V (rows of P sum to 1).
"""
import torch
import torch.nn as nn

from .. import ops as O
from .layers import Conv3x3, GroupNorm, Linear, Upsample2D, _Cached, _Dropout, to_nhwc

SCALING_FACTOR = 0.18215


class VaeResnetBlock2D(nn.Module):
    """ResnetBlock2D without time embedding (temb_channels=None), eps 1e-6."""

    def __init__(self, in_channels, out_channels, groups=32, eps=1e-6):
        super().__init__()
        self.norm1 = GroupNorm(groups, in_channels, eps)
        self.conv1 = Conv3x3(in_channels, out_channels)
        self.norm2 = GroupNorm(groups, out_channels, eps)
        self.conv2 = Conv3x3(out_channels, out_channels)
        self.conv_shortcut = Linear(in_channels, out_channels, conv=True) if in_channels != out_channels else None

    def run(self, x, m, h, w):
        hw = h * w
        hid = self.conv1.run(self.norm1.run(x, m, hw, True), m, h, w)
        a = self.norm2.run(hid, m, hw, True)
        sc = x if self.conv_shortcut is None else self.conv_shortcut.run(x)
        return self.conv2.run(a, m, h, w, res=sc)


class VaeAttention(_Cached):
    def __init__(self, channels, groups=32, eps=1e-6):
        super().__init__()
        self.channels = channels
        self.group_norm = GroupNorm(groups, channels, eps)
        self.to_q = Linear(channels, channels)
        self.to_k = Linear(channels, channels)
        self.to_v = Linear(channels, channels)
        self.to_out = nn.ModuleList([Linear(channels, channels), _Dropout()])
        self.scale = channels ** -0.5

    def _qk(self):
        if "_pk_qk" not in self.__dict__:
            self.__dict__["_pk_qk"] = (torch.cat([self.to_q.weight, self.to_k.weight]).detach().contiguous(),
                                       torch.cat([self.to_q.bias, self.to_k.bias]).detach().contiguous())
        return self.__dict__["_pk_qk"]

    def run(self, x, m, hw):
        c = self.channels
        t = self.group_norm.run(x, m, hw, False)
        wqk, bqk = self._qk()
        qk = O.gemm(t, wqk, bqk)                                     # (m*hw, 2C): q | k
        out = torch.empty_like(x)
        wv = self.to_v.weight.detach()
        for i in range(m):                                           # one view at a time: 1 head, hw tokens
            rows = slice(i * hw, (i + 1) * hw)
            q, k = qk[rows, :c], qk[rows, c:]
            s = O.gemm(q, k.contiguous(), alpha=self.scale, out_f32=True)           # (hw, hw) fp32 logits
            p = O.softmax_rows(s, x.dtype)                                          # (hw, hw) probabilities
            vt = O.gemm(wv, t[rows])                                                # V^T (C, hw), bias added below
            o = O.gemm(p, vt, self.to_v.bias)                                       # P V + b_v
            O.gemm(o, self.to_out[0].w2d, self.to_out[0].bias, res=x[rows], out=out[rows])
        return out


class VaeMidBlock(nn.Module):
    def __init__(self, c, eps):
        super().__init__()
        self.resnets = nn.ModuleList([VaeResnetBlock2D(c, c, eps=eps) for _ in range(2)])
        self.attentions = nn.ModuleList([VaeAttention(c, eps=eps)])

    def run(self, x, m, h, w):
        x = self.resnets[0].run(x, m, h, w)
        x = self.attentions[0].run(x, m, h * w)
        return self.resnets[1].run(x, m, h, w)


class UpDecoderBlock2D(nn.Module):
    def __init__(self, cin, cout, add_upsample, eps, layers=3):
        super().__init__()
        self.resnets = nn.ModuleList([VaeResnetBlock2D(cin if i == 0 else cout, cout, eps=eps) for i in range(layers)])
        self.upsamplers = nn.ModuleList([Upsample2D(cout)]) if add_upsample else None

    def run(self, x, m, h, w):
        for r in self.resnets:
            x = r.run(x, m, h, w)
        if self.upsamplers is not None:
            x = self.upsamplers[0].conv.run(x, m, h, w, up_size=(2 * h, 2 * w))     # nearest x2 fused in the gather
            h, w = 2 * h, 2 * w
        return x, h, w


class Decoder(nn.Module):
    def __init__(self, block_out_channels=(128, 256, 512, 512), latent_channels=4, out_channels=3, eps=1e-6):
        super().__init__()
        rev = list(reversed(block_out_channels))
        self.conv_in = Conv3x3(latent_channels, rev[0])
        self.mid_block = VaeMidBlock(rev[0], eps)
        blocks, prev = [], rev[0]
        for i, c in enumerate(rev):
            blocks.append(UpDecoderBlock2D(prev, c, add_upsample=i != len(rev) - 1, eps=eps))
            prev = c
        self.up_blocks = nn.ModuleList(blocks)
        self.conv_norm_out = GroupNorm(32, rev[-1], eps)
        self.conv_out = Conv3x3(rev[-1], out_channels)


class AutoencoderKLDecoder(nn.Module):
    def __init__(self, block_out_channels=(128, 256, 512, 512), latent_channels=4, scaling_factor=SCALING_FACTOR):
        super().__init__()
        self.post_quant_conv = Linear(latent_channels, latent_channels, conv=True)
        self.decoder = Decoder(block_out_channels, latent_channels)
        self.scaling_factor = scaling_factor

    @property
    def dtype(self):
        return self.decoder.conv_in.weight.dtype

    def _pq(self):
        """post_quant_conv packed for the 8-channel padded latent layout: (8, 8), zero padding."""
        d = self.post_quant_conv.__dict__
        if "_pk_pq" not in d:
            w = self.post_quant_conv.weight.detach().reshape(4, 4)
            wp = w.new_zeros((8, 8))
            wp[:4, :4] = w
            bp = w.new_zeros((8,))
            bp[:4] = self.post_quant_conv.bias.detach()
            d["_pk_pq"] = (wp.contiguous(), bp.contiguous())
        return d["_pk_pq"]

    @torch.no_grad()
    def decode(self, z, pre_scale=1.0):
        """z: (m, 4, h, w) latents on the GPU -> (m, 3, 8h, 8w) in the model dtype."""
        if not z.is_cuda:
            raise RuntimeError("the VAE decoder runs on the GPU only")
        m, _, h, w = z.shape
        x8 = O.nchw_to_nhwc(z.to(self.dtype).contiguous(), 8)          # (m*h*w, 8), channels 4..7 zero
        wp, bp = self._pq()
        x8 = O.gemm(x8, wp, bp) if pre_scale == 1.0 else O.gemm(O.scale(x8, pre_scale), wp, bp)
        dec = self.decoder
        x = dec.conv_in.run(x8, m, h, w)
        x = dec.mid_block.run(x, m, h, w)
        for blk in dec.up_blocks:
            x, h, w = blk.run(x, m, h, w)
        x = dec.conv_norm_out.run(x, m, h * w, True)
        return O.conv3x3_small_cout(x, dec.conv_out.packed, dec.conv_out.bias, m, h, w)


@torch.no_grad()
def decode_latents(vae: AutoencoderKLDecoder, latents):
    """pipeline_bev_controlnet.py:101-113 without the host copy: latents (b, n_cam, 4, h, w) ->
    images (b, n_cam, 3, 8h, 8w) in [0, 1], on the GPU, fp32."""
    b = latents.shape[0]
    img = vae.decode(latents.flatten(0, 1), pre_scale=1.0 / vae.scaling_factor)
    img = img.view(b, -1, *img.shape[1:]).float()
    return (img / 2 + 0.5).clamp_(0, 1)
