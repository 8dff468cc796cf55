"""Oracle: the ImageNet-64 noisy classifier used by ImageNetScorer, as pure functions (torch CPU fp32).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restates the inference path of
  edm/unet.py:701-912   EncoderUNetModel (pool='attention', resblock_updown, scale-shift norm)
  edm/unet.py:161-274   ResBlock._forward
  edm/unet.py:277-323   AttentionBlock._forward (legacy head order)
  edm/unet.py:346-372   QKVAttentionLegacy.forward
  edm/unet.py:379-407   QKVAttention.forward (used by the attention pool)
  edm/unet.py:40-69     AttentionPool2d.forward
  edm/nn_utils.py:17-19 GroupNorm32, :103-121 timestep_embedding
State-dict keys are the reference module's own ('input_blocks.1.0.in_layers.2.weight', ...).
"""
import math
from dataclasses import dataclass, field
from typing import Dict, List, Tuple

import torch
import torch.nn.functional as F


@dataclass
class ClsCfg:
    image_size: int = 64
    in_channels: int = 3
    model_channels: int = 128
    out_channels: int = 1000
    num_res_blocks: int = 4
    attention_ds: Tuple[int, ...] = (2, 4, 8)       # image_size // {32,16,8}  (scorers.py:121-123)
    channel_mult: Tuple[int, ...] = (1, 2, 3, 4)
    num_head_channels: int = 64


@dataclass
class ClsLayer:
    kind: str            # 'conv_in' | 'res' | 'attn'
    prefix: str
    cin: int = 0
    cout: int = 0
    down: bool = False


def encoder_layout(cfg: ClsCfg):
    """Enumerates input_blocks / middle_block exactly as EncoderUNetModel.__init__ (unet.py:755-840)."""
    mc = cfg.model_channels
    ch = int(cfg.channel_mult[0] * mc)
    blocks: List[List[ClsLayer]] = [[ClsLayer('conv_in', 'input_blocks.0.0', cfg.in_channels, ch)]]
    ds = 1
    for level, mult in enumerate(cfg.channel_mult):
        for _ in range(cfg.num_res_blocks):
            i = len(blocks)
            layers = [ClsLayer('res', f'input_blocks.{i}.0', ch, int(mult * mc))]
            ch = int(mult * mc)
            if ds in cfg.attention_ds:
                layers.append(ClsLayer('attn', f'input_blocks.{i}.1', ch, ch))
            blocks.append(layers)
        if level != len(cfg.channel_mult) - 1:
            i = len(blocks)
            blocks.append([ClsLayer('res', f'input_blocks.{i}.0', ch, ch, down=True)])
            ds *= 2
    middle = [ClsLayer('res', 'middle_block.0', ch, ch), ClsLayer('attn', 'middle_block.1', ch, ch),
              ClsLayer('res', 'middle_block.2', ch, ch)]
    return blocks, middle, ch, cfg.image_size // ds


def timestep_embedding(timesteps, dim, max_period=10000):
    """nn_utils.py:103-121."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(0, half, dtype=torch.float32) / half)
    args = timesteps[:, None].float() * freqs[None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if dim % 2:
        emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
    return emb


def _gn32(x, w, b):
    return F.group_norm(x.float(), 32, w, b, eps=1e-5)


def res_block(sd, p, L: ClsLayer, x, emb):
    """ResBlock._forward (unet.py:254-274) with use_scale_shift_norm=True; down = 2x2 avg-pool of both
    the activated branch and the skip input (unet.py:213-215, 255-260)."""
    h = F.silu(_gn32(x, sd[f'{p}.in_layers.0.weight'], sd[f'{p}.in_layers.0.bias']))
    if L.down:
        h = F.avg_pool2d(h, 2, 2)
        x = F.avg_pool2d(x, 2, 2)
    h = F.conv2d(h, sd[f'{p}.in_layers.2.weight'], sd[f'{p}.in_layers.2.bias'], padding=1)
    e = F.linear(F.silu(emb), sd[f'{p}.emb_layers.1.weight'], sd[f'{p}.emb_layers.1.bias'])[:, :, None, None]
    scale, shift = torch.chunk(e, 2, dim=1)
    h = _gn32(h, sd[f'{p}.out_layers.0.weight'], sd[f'{p}.out_layers.0.bias']) * (1 + scale) + shift
    h = F.conv2d(F.silu(h), sd[f'{p}.out_layers.3.weight'], sd[f'{p}.out_layers.3.bias'], padding=1)
    if L.cin != L.cout:
        x = F.conv2d(x, sd[f'{p}.skip_connection.weight'], sd[f'{p}.skip_connection.bias'])
    return x + h


def qkv_attention_legacy(qkv, n_heads):
    """QKVAttentionLegacy.forward (unet.py:355-372): heads split first, then q|k|v."""
    bs, width, length = qkv.shape
    ch = width // (3 * n_heads)
    q, k, v = qkv.reshape(bs * n_heads, ch * 3, length).split(ch, dim=1)
    scale = 1 / math.sqrt(math.sqrt(ch))
    w = torch.einsum('bct,bcs->bts', q * scale, k * scale)
    w = torch.softmax(w.float(), dim=-1)
    a = torch.einsum('bts,bcs->bct', w, v)
    return a.reshape(bs, -1, length)


def qkv_attention_new(qkv, n_heads):
    """QKVAttention.forward (unet.py:388-407): q|k|v split first, then heads."""
    bs, width, length = qkv.shape
    ch = width // (3 * n_heads)
    q, k, v = qkv.chunk(3, dim=1)
    scale = 1 / math.sqrt(math.sqrt(ch))
    w = torch.einsum('bct,bcs->bts', (q * scale).reshape(bs * n_heads, ch, length),
                     (k * scale).reshape(bs * n_heads, ch, length))
    w = torch.softmax(w.float(), dim=-1)
    a = torch.einsum('bts,bcs->bct', w, v.reshape(bs * n_heads, ch, length))
    return a.reshape(bs, -1, length)


def attention_block(sd, p, x, head_ch):
    """AttentionBlock._forward (unet.py:317-323)."""
    b, c, hh, ww = x.shape
    xf = x.reshape(b, c, -1)
    qkv = F.conv1d(_gn32(xf, sd[f'{p}.norm.weight'], sd[f'{p}.norm.bias']), sd[f'{p}.qkv.weight'], sd[f'{p}.qkv.bias'])
    h = qkv_attention_legacy(qkv, c // head_ch)
    h = F.conv1d(h, sd[f'{p}.proj_out.weight'], sd[f'{p}.proj_out.bias'])
    return (xf + h).reshape(b, c, hh, ww)


def attention_pool(sd, p, x, head_ch):
    """AttentionPool2d.forward (unet.py:61-69)."""
    b, c = x.shape[:2]
    x = x.reshape(b, c, -1)
    x = torch.cat([x.mean(dim=-1, keepdim=True), x], dim=-1)
    x = x + sd[f'{p}.positional_embedding'][None]
    x = F.conv1d(x, sd[f'{p}.qkv_proj.weight'], sd[f'{p}.qkv_proj.bias'])
    x = qkv_attention_new(x, c // head_ch)
    x = F.conv1d(x, sd[f'{p}.c_proj.weight'], sd[f'{p}.c_proj.bias'])
    return x[:, :, 0]


@torch.no_grad()
def encoder_unet(sd: Dict[str, torch.Tensor], cfg: ClsCfg, x, timesteps):
    """EncoderUNetModel.forward (unet.py:889-912), pool='attention'. Returns logits [N, out_channels]."""
    mc = cfg.model_channels
    emb = timestep_embedding(timesteps, mc)
    emb = F.linear(emb, sd['time_embed.0.weight'], sd['time_embed.0.bias'])
    emb = F.linear(F.silu(emb), sd['time_embed.2.weight'], sd['time_embed.2.bias'])
    blocks, middle, ch, res = encoder_layout(cfg)
    h = x.float()
    for layers in blocks:
        for L in layers:
            if L.kind == 'conv_in':
                h = F.conv2d(h, sd[f'{L.prefix}.weight'], sd[f'{L.prefix}.bias'], padding=1)
            elif L.kind == 'res':
                h = res_block(sd, L.prefix, L, h, emb)
            else:
                h = attention_block(sd, L.prefix, h, cfg.num_head_channels)
    for L in middle:
        h = res_block(sd, L.prefix, L, h, emb) if L.kind == 'res' else attention_block(sd, L.prefix, h, cfg.num_head_channels)
    h = F.silu(_gn32(h, sd['out.0.weight'], sd['out.0.bias']))
    return attention_pool(sd, 'out.2', h, cfg.num_head_channels)


def count_flops(cfg: ClsCfg):
    """2 FLOP/MAC over conv, attention bmm and linear layers (38.16 GF at defaults)."""
    blocks, middle, ch, res_f = encoder_layout(cfg)
    emb = cfg.model_channels * 4
    conv = attn = lin = 0.0
    res = cfg.image_size
    lin += 2.0 * cfg.model_channels * emb + 2.0 * emb * emb
    for layers in blocks + [middle]:
        for L in layers:
            if L.kind == 'conv_in':
                conv += 2.0 * res * res * L.cin * L.cout * 9
            elif L.kind == 'res':
                if L.down:
                    res //= 2
                conv += 2.0 * res * res * (L.cin * L.cout + L.cout * L.cout) * 9
                if L.cin != L.cout:
                    conv += 2.0 * res * res * L.cin * L.cout
                lin += 2.0 * emb * 2 * L.cout
            else:
                t = res * res
                conv += 2.0 * t * L.cin * 3 * L.cin + 2.0 * t * L.cin * L.cin
                attn += 2 * 2.0 * t * t * L.cin
    t = res * res + 1
    conv += 2.0 * t * ch * 3 * ch + 2.0 * t * ch * cfg.out_channels
    attn += 2 * 2.0 * t * t * ch
    return dict(conv=conv, attention=attn, linear=lin, total=conv + attn + lin)
