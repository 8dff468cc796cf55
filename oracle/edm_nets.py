"""Oracle: EDM denoiser networks as pure functions over a flat state dict (torch CPU, fp32).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restates, op for op, the inference behaviour of
  edm/training/networks.py:49-90    Conv2d (3x3 / 1x1 / resample-only, [1,1] filter)
  edm/training/networks.py:96-106   GroupNorm
  edm/training/networks.py:113-118  AttentionOp (fp32 softmax(q^T k / sqrt(d)))
  edm/training/networks.py:134-187  UNetBlock
  edm/training/networks.py:193-206  PositionalEmbedding
  edm/training/networks.py:229-363  SongUNet   (DDPM++ options only)
  edm/training/networks.py:372-461  DhariwalUNet (ADM)
  edm/training/networks.py:632-671  EDMPrecond
State-dict keys are the reference's own (`model.enc.64x64_conv.weight`, ...), so a state dict
taken from a reference module evaluates here unchanged.
"""
import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import numpy as np
import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------------------
# configuration + layout enumeration (mirrors the constructors' loops)

@dataclass
class NetCfg:
    arch: str = 'adm'                 # 'adm' (DhariwalUNet) | 'ddpmpp' (SongUNet, DDPM++ options)
    img_resolution: int = 64
    img_channels: int = 3
    label_dim: int = 0
    model_channels: int = 192
    channel_mult: List[int] = field(default_factory=lambda: [1, 2, 3, 4])
    channel_mult_emb: int = 4
    num_blocks: int = 3
    attn_resolutions: List[int] = field(default_factory=lambda: [32, 16, 8])
    augment_dim: int = 0
    sigma_data: float = 0.5
    sigma_min: float = 0.0
    sigma_max: float = float('inf')


@dataclass
class BlockSpec:
    name: str          # state-dict prefix below 'model.', e.g. 'enc.64x64_block0'
    kind: str          # 'conv' | 'block' | 'norm'
    cin: int
    cout: int
    res_in: int
    up: bool = False
    down: bool = False
    attention: bool = False
    heads: int = 0


def adm_layout(cfg: NetCfg):
    """Block list of DhariwalUNet.__init__ (networks.py:402-433)."""
    mc = cfg.model_channels
    enc, dec = [], []
    cout = cfg.img_channels
    for level, mult in enumerate(cfg.channel_mult):
        res = cfg.img_resolution >> level
        if level == 0:
            cin, cout = cout, mc * mult
            enc.append(BlockSpec(f'enc.{res}x{res}_conv', 'conv', cin, cout, res))
        else:
            enc.append(BlockSpec(f'enc.{res}x{res}_down', 'block', cout, cout, res * 2, down=True))
        for idx in range(cfg.num_blocks):
            cin, cout = cout, mc * mult
            att = res in cfg.attn_resolutions
            enc.append(BlockSpec(f'enc.{res}x{res}_block{idx}', 'block', cin, cout, res,
                                 attention=att, heads=(cout // 64 if att else 0)))
    skips = [b.cout for b in enc]
    for level, mult in reversed(list(enumerate(cfg.channel_mult))):
        res = cfg.img_resolution >> level
        if level == len(cfg.channel_mult) - 1:
            dec.append(BlockSpec(f'dec.{res}x{res}_in0', 'block', cout, cout, res, attention=True, heads=cout // 64))
            dec.append(BlockSpec(f'dec.{res}x{res}_in1', 'block', cout, cout, res))
        else:
            dec.append(BlockSpec(f'dec.{res}x{res}_up', 'block', cout, cout, res // 2, up=True))
        for idx in range(cfg.num_blocks + 1):
            cin = cout + skips.pop()
            cout = mc * mult
            att = res in cfg.attn_resolutions
            dec.append(BlockSpec(f'dec.{res}x{res}_block{idx}', 'block', cin, cout, res,
                                 attention=att, heads=(cout // 64 if att else 0)))
    return enc, dec, cout


def ddpmpp_layout(cfg: NetCfg):
    """Block list of SongUNet.__init__ with encoder/decoder 'standard' (networks.py:275-318)."""
    mc = cfg.model_channels
    enc, dec = [], []
    cout = cfg.img_channels
    for level, mult in enumerate(cfg.channel_mult):
        res = cfg.img_resolution >> level
        if level == 0:
            cin, cout = cout, mc
            enc.append(BlockSpec(f'enc.{res}x{res}_conv', 'conv', cin, cout, res))
        else:
            enc.append(BlockSpec(f'enc.{res}x{res}_down', 'block', cout, cout, res * 2, down=True))
        for idx in range(cfg.num_blocks):
            cin, cout = cout, mc * mult
            att = res in cfg.attn_resolutions
            enc.append(BlockSpec(f'enc.{res}x{res}_block{idx}', 'block', cin, cout, res,
                                 attention=att, heads=(1 if att else 0)))
    skips = [b.cout for b in enc]
    for level, mult in reversed(list(enumerate(cfg.channel_mult))):
        res = cfg.img_resolution >> level
        if level == len(cfg.channel_mult) - 1:
            dec.append(BlockSpec(f'dec.{res}x{res}_in0', 'block', cout, cout, res, attention=True, heads=1))
            dec.append(BlockSpec(f'dec.{res}x{res}_in1', 'block', cout, cout, res))
        else:
            dec.append(BlockSpec(f'dec.{res}x{res}_up', 'block', cout, cout, res // 2, up=True))
        for idx in range(cfg.num_blocks + 1):
            cin = cout + skips.pop()
            cout = mc * mult
            att = (idx == cfg.num_blocks) and (res in cfg.attn_resolutions)
            dec.append(BlockSpec(f'dec.{res}x{res}_block{idx}', 'block', cin, cout, res,
                                 attention=att, heads=(1 if att else 0)))
    return enc, dec, cout


# --------------------------------------------------------------------------------------
# primitives

def silu(x):
    return x * torch.sigmoid(x)


def resample_down(x):
    """networks.py:84-85 with f = outer([1,1],[1,1])/4: depthwise 2x2 stride-2 conv."""
    c = x.shape[1]
    f = torch.full([c, 1, 2, 2], 0.25, dtype=x.dtype)
    return F.conv2d(x, f, groups=c, stride=2)


def resample_up(x):
    """networks.py:82-83 with 4f = ones: depthwise transposed conv stride 2 (= nearest 2x)."""
    c = x.shape[1]
    f = torch.ones([c, 1, 2, 2], dtype=x.dtype)
    return F.conv_transpose2d(x, f, groups=c, stride=2)


def conv2d(x, w, b, up=False, down=False):
    """Conv2d.forward, non-fused branch (networks.py:81-89)."""
    if up:
        x = resample_up(x)
    if down:
        x = resample_down(x)
    if w is not None:
        x = F.conv2d(x, w, padding=w.shape[-1] // 2)
    if b is not None:
        x = x + b.reshape(1, -1, 1, 1)
    return x


def group_norm(x, w, b, eps):
    """GroupNorm.forward (networks.py:99,104-106): min(32, C//4) groups."""
    c = x.shape[1]
    return F.group_norm(x, num_groups=min(32, c // 4), weight=w, bias=b, eps=eps)


def attention_weights(q, k):
    """AttentionOp.forward (networks.py:116)."""
    return torch.einsum('ncq,nck->nqk', q.float(), (k / math.sqrt(k.shape[1])).float()).softmax(dim=2)


def positional_embedding(x, num_channels, endpoint=False, max_positions=10000):
    """PositionalEmbedding.forward (networks.py:200-206)."""
    half = num_channels // 2
    freqs = torch.arange(0, half, dtype=torch.float32)
    freqs = freqs / (half - (1 if endpoint else 0))
    freqs = (1 / max_positions) ** freqs
    x = torch.outer(x, freqs.to(x.dtype))
    return torch.cat([x.cos(), x.sin()], dim=1)


def unet_block(sd: Dict[str, torch.Tensor], p: str, spec: BlockSpec, x, emb, *, adaptive, skip_scale, eps,
               resample_proj):
    """UNetBlock.forward (networks.py:166-187); dropout is identity at inference."""
    g = lambda k: sd.get(f'{p}.{k}')
    orig = x
    x = conv2d(silu(group_norm(x, g('norm0.weight'), g('norm0.bias'), eps)),
               g('conv0.weight'), g('conv0.bias'), up=spec.up, down=spec.down)
    params = F.linear(emb, g('affine.weight'), g('affine.bias'))[:, :, None, None]
    if adaptive:
        scale, shift = params.chunk(2, dim=1)
        x = silu(torch.addcmul(shift, group_norm(x, g('norm1.weight'), g('norm1.bias'), eps), scale + 1))
    else:
        x = silu(group_norm(x + params, g('norm1.weight'), g('norm1.bias'), eps))
    x = conv2d(x, g('conv1.weight'), g('conv1.bias'))
    if spec.cin != spec.cout or spec.up or spec.down:
        has_w = resample_proj or spec.cin != spec.cout          # networks.py:158
        sk = conv2d(orig, g('skip.weight') if has_w else None, g('skip.bias') if has_w else None,
                    up=spec.up, down=spec.down)
    else:
        sk = orig
    x = (x + sk) * skip_scale
    if spec.heads:
        n, c, h, w_ = x.shape
        qkv = conv2d(group_norm(x, g('norm2.weight'), g('norm2.bias'), eps), g('qkv.weight'), g('qkv.bias'))
        q, k, v = qkv.reshape(n * spec.heads, c // spec.heads, 3, -1).unbind(2)
        w = attention_weights(q, k)
        a = torch.einsum('nqk,nck->ncq', w, v)
        x = conv2d(a.reshape(n, c, h, w_), g('proj.weight'), g('proj.bias')) + x
        x = x * skip_scale
    return x


# --------------------------------------------------------------------------------------
# U-Nets

def dhariwal_unet(sd, cfg: NetCfg, x, noise_labels, class_labels):
    """DhariwalUNet.forward (networks.py:435-461). `sd` keys are relative to the U-Net ('map_layer0.weight')."""
    emb = positional_embedding(noise_labels, cfg.model_channels)
    emb = silu(F.linear(emb, sd['map_layer0.weight'], sd['map_layer0.bias']))
    emb = F.linear(emb, sd['map_layer1.weight'], sd['map_layer1.bias'])
    if cfg.label_dim:
        emb = emb + F.linear(class_labels, sd['map_label.weight'])
    emb = silu(emb)
    enc, dec, cout = adm_layout(cfg)
    kw = dict(adaptive=True, skip_scale=1.0, eps=1e-5, resample_proj=False)
    skips = []
    for b in enc:
        if b.kind == 'conv':
            x = conv2d(x, sd[f'{b.name}.weight'], sd[f'{b.name}.bias'])
        else:
            x = unet_block(sd, b.name, b, x, emb, **kw)
        skips.append(x)
    for b in dec:
        if x.shape[1] != b.cin:
            x = torch.cat([x, skips.pop()], dim=1)
        x = unet_block(sd, b.name, b, x, emb, **kw)
    x = silu(group_norm(x, sd['out_norm.weight'], sd['out_norm.bias'], 1e-5))
    return conv2d(x, sd['out_conv.weight'], sd['out_conv.bias'])


def song_unet(sd, cfg: NetCfg, x, noise_labels, class_labels):
    """SongUNet.forward, DDPM++ configuration (networks.py:320-363): positional embedding with
    endpoint=True and sin/cos swap (:323), labels scaled by sqrt(label_dim) (:328), augment labels None."""
    mc = cfg.model_channels
    emb = positional_embedding(noise_labels, mc, endpoint=True)
    emb = emb.reshape(emb.shape[0], 2, -1).flip(1).reshape(*emb.shape)
    if cfg.label_dim:
        emb = emb + F.linear(class_labels * math.sqrt(cfg.label_dim), sd['map_label.weight'], sd['map_label.bias'])
    emb = silu(F.linear(emb, sd['map_layer0.weight'], sd['map_layer0.bias']))
    emb = silu(F.linear(emb, sd['map_layer1.weight'], sd['map_layer1.bias']))
    enc, dec, cout = ddpmpp_layout(cfg)
    kw = dict(adaptive=False, skip_scale=math.sqrt(0.5), eps=1e-6, resample_proj=True)
    skips = []
    for b in enc:
        if b.kind == 'conv':
            x = conv2d(x, sd[f'{b.name}.weight'], sd[f'{b.name}.bias'])
        else:
            x = unet_block(sd, b.name, b, x, emb, **kw)
        skips.append(x)
    for b in dec:
        if x.shape[1] != b.cin:
            x = torch.cat([x, skips.pop()], dim=1)
        x = unet_block(sd, b.name, b, x, emb, **kw)
    r = cfg.img_resolution
    x = silu(group_norm(x, sd[f'dec.{r}x{r}_aux_norm.weight'], sd[f'dec.{r}x{r}_aux_norm.bias'], 1e-6))
    return conv2d(x, sd[f'dec.{r}x{r}_aux_conv.weight'], sd[f'dec.{r}x{r}_aux_conv.bias'])


class EDMPrecondOracle:
    """EDMPrecond (networks.py:632-671) on CPU: callable `net(x, sigma, class_labels)` with the
    attributes the sampler touches (`round_sigma`, `img_resolution`, `img_channels`, `label_dim`,
    `sigma_min`, `sigma_max`)."""

    def __init__(self, cfg: NetCfg, state_dict: Dict[str, torch.Tensor]):
        self.cfg = cfg
        self.img_resolution = cfg.img_resolution
        self.img_channels = cfg.img_channels
        self.label_dim = cfg.label_dim
        self.sigma_min, self.sigma_max, self.sigma_data = cfg.sigma_min, cfg.sigma_max, cfg.sigma_data
        pre = 'model.'
        self.sd = {k[len(pre):]: v.detach().to(torch.float32) for k, v in state_dict.items() if k.startswith(pre)}
        self.evals = 0      # rows pushed through the denoiser (the BASELINE metric's unit)

    def round_sigma(self, sigma):
        return torch.as_tensor(sigma)

    @torch.no_grad()
    def __call__(self, x, sigma, class_labels=None):
        x = x.to(torch.float32)
        sigma = torch.as_tensor(sigma).to(torch.float32).reshape(-1, 1, 1, 1)
        if self.label_dim == 0:
            class_labels = None
        elif class_labels is None:
            class_labels = torch.zeros([1, self.label_dim])
        else:
            class_labels = class_labels.to(torch.float32).reshape(-1, self.label_dim)
        sd2 = self.sigma_data ** 2
        c_skip = sd2 / (sigma ** 2 + sd2)
        c_out = sigma * self.sigma_data / (sigma ** 2 + sd2).sqrt()
        c_in = 1 / (sd2 + sigma ** 2).sqrt()
        c_noise = sigma.log() / 4
        fn = dhariwal_unet if self.cfg.arch == 'adm' else song_unet
        F_x = fn(self.sd, self.cfg, c_in * x, c_noise.flatten(), class_labels)
        self.evals += x.shape[0]
        return c_skip * x + c_out * F_x.to(torch.float32)


# --------------------------------------------------------------------------------------
# analytic FLOP counter (2 FLOP / MAC; conv + attention bmm + linear), SURVEY section 6 / 8(d)

def count_flops(cfg: NetCfg) -> Dict[str, float]:
    enc, dec, cout = (adm_layout if cfg.arch == 'adm' else ddpmpp_layout)(cfg)
    mc = cfg.model_channels
    emb_ch = mc * cfg.channel_mult_emb
    conv = attn = lin = 0.0
    adaptive = cfg.arch == 'adm'
    for b in enc + dec:
        if b.kind == 'conv':
            conv += 2.0 * b.res_in ** 2 * b.cout * b.cin * 9
            continue
        r_out = b.res_in * 2 if b.up else (b.res_in // 2 if b.down else b.res_in)
        px = r_out ** 2
        conv += 2.0 * px * b.cout * b.cin * 9 + 2.0 * px * b.cout * b.cout * 9
        lin += 2.0 * emb_ch * b.cout * (2 if adaptive else 1)
        has_skip_w = (b.cin != b.cout) or ((b.up or b.down) and not adaptive)
        if has_skip_w:
            conv += 2.0 * px * b.cout * b.cin
        if b.heads:
            conv += 2.0 * px * b.cout * 3 * b.cout + 2.0 * px * b.cout * b.cout
            attn += 2 * 2.0 * px * px * b.cout
    conv += 2.0 * cfg.img_resolution ** 2 * cfg.img_channels * cout * 9
    nch = mc
    lin += 2.0 * nch * emb_ch + 2.0 * emb_ch * emb_ch
    if cfg.label_dim:
        lin += 2.0 * cfg.label_dim * (emb_ch if adaptive else nch)
    return dict(conv=conv, attention=attn, linear=lin, total=conv + attn + lin)
