"""Oracle: the SD VAE decoder (AutoencoderKL.decode) on CPU in plain torch ops.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restates, for the stock SD-1.5 configuration (group norm, SiLU, one single-head attention in the mid block):
  sd/diffusers/src/diffusers/models/autoencoders/autoencoder_kl.py:287-320   decode / _decode (post_quant_conv, decoder)
  sd/diffusers/src/diffusers/models/autoencoders/vae.py:204-323              Decoder (conv_in, mid block, up blocks, norm/act/conv_out)
  sd/diffusers/src/diffusers/models/resnet.py  ResnetBlock2D.forward         norm1-silu-conv1-norm2-silu-conv2 (+1x1 shortcut), temb None
  sd/diffusers/src/diffusers/models/unets/unet_2d_blocks.py  UNetMidBlock2D / UpDecoderBlock2D (3 resnets + nearest-2x + conv)
  sd/diffusers/src/diffusers/models/attention_processor.py:3242-3335  AttnProcessor2_0 (group_norm, q/k/v linears, SDPA, to_out, +residual)
Parameters come as a state dict with diffusers' own key names.  Pinned by tests/golden/make_golden_vae.py against the reference's
vendored AutoencoderKL (tests/test_oracle_vae_golden.py)."""
import math

import torch
import torch.nn.functional as F


def _gn(x, sd, key, groups, eps=1e-6):
    return F.group_norm(x, groups, sd[key + '.weight'], sd[key + '.bias'], eps)


def _conv(x, sd, key, pad):
    return F.conv2d(x, sd[key + '.weight'], sd[key + '.bias'], padding=pad)


def resnet(x, sd, key, groups):
    h = _conv(F.silu(_gn(x, sd, key + '.norm1', groups)), sd, key + '.conv1', 1)
    h = _conv(F.silu(_gn(h, sd, key + '.norm2', groups)), sd, key + '.conv2', 1)
    if key + '.conv_shortcut.weight' in sd:
        x = _conv(x, sd, key + '.conv_shortcut', 0)
    return x + h                                              # output_scale_factor = 1


def attention(x, sd, key, groups):
    b, c, hh, ww = x.shape
    h = _gn(x.view(b, c, hh * ww), sd, key + '.group_norm', groups).transpose(1, 2)          # [b, t, c]
    q = F.linear(h, sd[key + '.to_q.weight'], sd[key + '.to_q.bias'])
    k = F.linear(h, sd[key + '.to_k.weight'], sd[key + '.to_k.bias'])
    v = F.linear(h, sd[key + '.to_v.weight'], sd[key + '.to_v.bias'])
    w = torch.softmax((q @ k.transpose(1, 2)).float() / math.sqrt(c), dim=-1).to(v.dtype)   # one head of dim c
    o = F.linear(w @ v, sd[key + '.to_out.0.weight'], sd[key + '.to_out.0.bias'])
    return o.transpose(1, 2).reshape(b, c, hh, ww) + x


def decode(sd, z, block_out_channels=(128, 256, 512, 512), layers_per_block=2, groups=32):
    """z: latents already divided by the scaling factor, [n, latent_channels, h, w] -> image [n, 3, 8h, 8w]."""
    if 'post_quant_conv.weight' in sd:
        z = _conv(z, sd, 'post_quant_conv', 0)
    x = _conv(z, sd, 'decoder.conv_in', 1)
    x = resnet(x, sd, 'decoder.mid_block.resnets.0', groups)
    x = attention(x, sd, 'decoder.mid_block.attentions.0', groups)
    x = resnet(x, sd, 'decoder.mid_block.resnets.1', groups)
    nb = len(block_out_channels)
    for i in range(nb):
        for j in range(layers_per_block + 1):
            x = resnet(x, sd, f'decoder.up_blocks.{i}.resnets.{j}', groups)
        if i != nb - 1:
            x = F.interpolate(x, scale_factor=2.0, mode='nearest')
            x = _conv(x, sd, f'decoder.up_blocks.{i}.upsamplers.0.conv', 1)
    x = F.silu(_gn(x, sd, 'decoder.conv_norm_out', groups))
    return _conv(x, sd, 'decoder.conv_out', 1)
