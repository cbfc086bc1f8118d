"""CPU restatement of the SD-v1.5 VAE decoder (`AutoencoderKL.decode`) — TEST INFRASTRUCTURE ONLY.

SURVEY.md §8f N2: the step after the denoising loop,
/root/reference/MD_txt_con_fusion/magicdrive/pipeline/pipeline_bev_controlnet.py:101-113 (`decode_latents`:
latents / scaling_factor -> `vae.decode(...).sample` per view -> `/ 2 + 0.5`, clamp) called at :532.

PARITY UNPINNED: the VAE is diffusers' `AutoencoderKL` (diffusers 0.17.1 pinned by the reference's
requirements; the weights are the stock `runwayml/stable-diffusion-v1-5` `vae/`), neither of which is in
/root/reference or in this image.  This file restates the published architecture in fp32 torch with
diffusers' parameter names (so a real `vae/diffusion_pytorch_model.bin` would load):
  post_quant_conv 1x1 (4->4); decoder.conv_in 3x3 (4->512); decoder.mid_block = resnet, single-head
  attention over h*w tokens (GroupNorm 32, q/k/v/out Linear 512, residual), resnet; four up blocks of three
  resnets each at channels (512, 512, 256, 128) with nearest-x2 + conv3x3 upsamplers after the first three;
  GroupNorm(32, 128) -> SiLU -> conv_out 3x3 (128->3).  Resnets have no time embedding; every GroupNorm
  uses eps 1e-6.
The resnet / upsampler classes are the ones of oracle/diffusers_restated.py (same arithmetic as the UNet's).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import diffusers_restated as D

SCALING_FACTOR = 0.18215      # vae.config.scaling_factor of SD-v1.5


class VaeAttention(nn.Module):
    """diffusers `Attention(..., heads=1, dim_head=C, norm_num_groups=32, residual_connection=True, bias=True)`
    as the VAE mid block builds it: softmax in fp32 (upcast_softmax), scale C^-0.5."""

    def __init__(self, channels, groups=32, eps=1e-6):
        super().__init__()
        self.group_norm = nn.GroupNorm(groups, channels, eps=eps, affine=True)
        self.to_q = nn.Linear(channels, channels)
        self.to_k = nn.Linear(channels, channels)
        self.to_v = nn.Linear(channels, channels)
        self.to_out = nn.ModuleList([nn.Linear(channels, channels), nn.Dropout(0.0)])
        self.scale = channels ** -0.5

    def forward(self, x):
        b, c, h, w = x.shape
        t = self.group_norm(x).view(b, c, h * w).transpose(1, 2)
        q, k, v = self.to_q(t), self.to_k(t), self.to_v(t)
        p = torch.softmax(torch.bmm(q, k.transpose(1, 2)) * self.scale, dim=-1)
        o = self.to_out[0](torch.bmm(p, v))
        return o.transpose(1, 2).reshape(b, c, h, w) + x


class MidBlock(nn.Module):
    def __init__(self, c, eps):
        super().__init__()
        self.resnets = nn.ModuleList([D.ResnetBlock2D(in_channels=c, out_channels=c, temb_channels=None, eps=eps)
                                      for _ in range(2)])
        self.attentions = nn.ModuleList([VaeAttention(c, eps=eps)])

    def forward(self, x):
        x = self.resnets[0](x, None)
        x = self.attentions[0](x)
        return self.resnets[1](x, None)


class UpDecoderBlock(nn.Module):
    def __init__(self, cin, cout, add_upsample, eps, layers=3):
        super().__init__()
        self.resnets = nn.ModuleList([D.ResnetBlock2D(in_channels=cin if i == 0 else cout, out_channels=cout,
                                                      temb_channels=None, eps=eps) for i in range(layers)])
        self.upsamplers = nn.ModuleList([D.Upsample2D(cout, out_channels=cout)]) if add_upsample else None

    def forward(self, x):
        for r in self.resnets:
            x = r(x, None)
        if self.upsamplers is not None:
            x = self.upsamplers[0](x)
        return x


class Decoder(nn.Module):
    def __init__(self, block_out_channels=(128, 256, 512, 512), latent_channels=4, out_channels=3, eps=1e-6):
        super().__init__()
        rev = list(reversed(block_out_channels))
        self.conv_in = nn.Conv2d(latent_channels, rev[0], 3, padding=1)
        self.mid_block = MidBlock(rev[0], eps)
        blocks, prev = [], rev[0]
        for i, c in enumerate(rev):
            blocks.append(UpDecoderBlock(prev, c, add_upsample=i != len(rev) - 1, eps=eps))
            prev = c
        self.up_blocks = nn.ModuleList(blocks)
        self.conv_norm_out = nn.GroupNorm(32, rev[-1], eps=eps)
        self.conv_out = nn.Conv2d(rev[-1], out_channels, 3, padding=1)

    def forward(self, z):
        x = self.mid_block(self.conv_in(z))
        for b in self.up_blocks:
            x = b(x)
        return self.conv_out(F.silu(self.conv_norm_out(x)))


class AutoencoderKLDecoder(nn.Module):
    """The decode half of AutoencoderKL: `decode(z) = decoder(post_quant_conv(z))`."""

    def __init__(self, block_out_channels=(128, 256, 512, 512), latent_channels=4):
        super().__init__()
        self.post_quant_conv = nn.Conv2d(latent_channels, latent_channels, 1)
        self.decoder = Decoder(block_out_channels, latent_channels)

    def decode(self, z):
        return self.decoder(self.post_quant_conv(z))


def decode_latents(vae, latents):
    """pipeline_bev_controlnet.py:101-113 up to the host copy: latents (b, n_cam, 4, h, w) ->
    images (b, n_cam, 3, 8h, 8w) in [0, 1]."""
    b = latents.shape[0]
    z = (1.0 / SCALING_FACTOR * latents).flatten(0, 1)
    img = vae.decode(z)
    img = img.view(b, -1, *img.shape[1:])
    return (img / 2 + 0.5).clamp(0, 1)
