"""Random-init weights for the hot-path networks, as flat state dicts with the reference's key names.

BASELINE prescribes random-init weights (no checkpoints travel).  The draws follow the reference
constructors' order on the torch global CPU generator, so `edm_state_dict(cfg, seed)` equals
`EDMPrecond(...).state_dict()` built under `torch.manual_seed(seed)` parameter for parameter
(edm/training/networks.py:19-24 weight_init, :31-37 Linear, :50-66 Conv2d, :135-164 UNetBlock,
:230-318 SongUNet, :373-433 DhariwalUNet), and `classifier_state_dict` equals the random-init
`EncoderUNetModel` of edm/unet.py:708-870 (torch.nn default initialisers).

`refill_degenerate` implements the weight rule of SURVEY.md section 8(d): a freshly constructed ADM
denoiser has every residual-branch output conv zero-initialised (networks.py:392), so F_x == 0 and the
U-Net would be numerically invisible; every parameter whose max-abs is < 1e-4 is re-drawn.
"""
import math
from collections import OrderedDict

import numpy as np
import torch

from .config import EDMConfig, ClassifierConfig, edm_blocks, classifier_layers


def _winit(shape, mode, fan_in, fan_out):
    if mode == 'xavier_uniform':
        return np.sqrt(6 / (fan_in + fan_out)) * (torch.rand(*shape) * 2 - 1)
    if mode == 'kaiming_uniform':
        return np.sqrt(3 / fan_in) * (torch.rand(*shape) * 2 - 1)
    if mode == 'kaiming_normal':
        return np.sqrt(1 / fan_in) * torch.randn(*shape)
    raise ValueError(mode)


class _Builder:
    def __init__(self):
        self.sd = OrderedDict()

    def linear(self, name, fin, fout, bias=True, mode='kaiming_normal', w=1, b=0):
        self.sd[f'{name}.weight'] = _winit([fout, fin], mode, fin, fout) * w
        if bias:
            self.sd[f'{name}.bias'] = _winit([fout], mode, fin, fout) * b

    def conv(self, name, cin, cout, k, mode='kaiming_normal', w=1, b=0):
        if not k:
            return
        self.sd[f'{name}.weight'] = _winit([cout, cin, k, k], mode, cin * k * k, cout * k * k) * w
        self.sd[f'{name}.bias'] = _winit([cout], mode, cin * k * k, cout * k * k) * b

    def norm(self, name, c):
        self.sd[f'{name}.weight'] = torch.ones(c)
        self.sd[f'{name}.bias'] = torch.zeros(c)


def edm_state_dict(cfg: EDMConfig, seed: int = 0, prefix: str = 'model.'):
    torch.manual_seed(seed)
    adm = cfg.arch == 'adm'
    mc, emb = cfg.model_channels, cfg.emb_channels
    B = _Builder()
    if adm:
        init = dict(mode='kaiming_uniform', w=np.sqrt(1 / 3), b=np.sqrt(1 / 3))
        zero = dict(mode='kaiming_uniform', w=0, b=0)
        attn = init
        if cfg.augment_dim:
            B.linear('map_augment', cfg.augment_dim, mc, bias=False, **zero)
        B.linear('map_layer0', mc, emb, **init)
        B.linear('map_layer1', emb, emb, **init)
        if cfg.label_dim:
            B.linear('map_label', cfg.label_dim, emb, bias=False, mode='kaiming_normal', w=np.sqrt(cfg.label_dim))
    else:
        init = dict(mode='xavier_uniform', w=1, b=0)
        zero = dict(mode='xavier_uniform', w=1e-5, b=0)
        attn = dict(mode='xavier_uniform', w=np.sqrt(0.2), b=0)
        if cfg.label_dim:
            B.linear('map_label', cfg.label_dim, mc, **init)
        if cfg.augment_dim:
            B.linear('map_augment', cfg.augment_dim, mc, bias=False, **init)
        B.linear('map_layer0', mc, emb, **init)
        B.linear('map_layer1', emb, emb, **init)
    enc, dec, cfin = edm_blocks(cfg)
    for blk in enc + dec:
        n = blk.name
        if blk.kind == 'conv':
            B.conv(n, blk.cin, blk.cout, 3, **init)
            continue
        B.norm(f'{n}.norm0', blk.cin)
        B.conv(f'{n}.conv0', blk.cin, blk.cout, 3, **init)
        B.linear(f'{n}.affine', emb, blk.cout * (2 if adm else 1), **init)
        B.norm(f'{n}.norm1', blk.cout)
        B.conv(f'{n}.conv1', blk.cout, blk.cout, 3, **zero)
        if blk.cin != blk.cout or blk.up or blk.down:
            k = 1 if ((not adm) or blk.cin != blk.cout) else 0      # resample_proj=True only for SongUNet
            B.conv(f'{n}.skip', blk.cin, blk.cout, k, **init)
        if blk.heads:
            B.norm(f'{n}.norm2', blk.cout)
            B.conv(f'{n}.qkv', blk.cout, blk.cout * 3, 1, **attn)
            B.conv(f'{n}.proj', blk.cout, blk.cout, 1, **zero)
    r = cfg.img_resolution
    if adm:
        B.norm('out_norm', cfin)
        B.conv('out_conv', cfin, cfg.img_channels, 3, **zero)
    else:
        B.norm(f'dec.{r}x{r}_aux_norm', cfin)
        B.conv(f'dec.{r}x{r}_aux_conv', cfin, cfg.img_channels, 3, **zero)
    return OrderedDict((prefix + k, v.to(torch.float32)) for k, v in B.sd.items())


def classifier_state_dict(cfg: ClassifierConfig, seed: int = 1):
    """Random-init EncoderUNetModel(pool='attention', use_scale_shift_norm, resblock_updown) parameters.
    torch.nn modules are instantiated (and discarded) in the reference's construction order so the
    default initialisers consume the generator identically."""
    nn = torch.nn
    torch.manual_seed(seed)
    sd = OrderedDict()

    def take(prefix, mod, zero=False):
        for k, v in mod.named_parameters():
            sd[f'{prefix}.{k}'] = (torch.zeros_like(v) if zero else v.detach().clone())

    mc = cfg.model_channels
    emb = mc * 4
    take('time_embed.0', nn.Linear(mc, emb))
    take('time_embed.2', nn.Linear(emb, emb))
    layers, ch, res = classifier_layers(cfg)
    for L in layers:
        p = L.prefix
        if L.kind == 'conv_in':
            take(p, nn.Conv2d(L.cin, L.cout, 3, padding=1))
        elif L.kind == 'res':
            take(f'{p}.in_layers.0', nn.GroupNorm(32, L.cin))
            take(f'{p}.in_layers.2', nn.Conv2d(L.cin, L.cout, 3, padding=1))
            take(f'{p}.emb_layers.1', nn.Linear(emb, 2 * L.cout))
            take(f'{p}.out_layers.0', nn.GroupNorm(32, L.cout))
            take(f'{p}.out_layers.3', nn.Conv2d(L.cout, L.cout, 3, padding=1), zero=True)
            if L.cin != L.cout:
                take(f'{p}.skip_connection', nn.Conv2d(L.cin, L.cout, 1))
        else:
            take(f'{p}.norm', nn.GroupNorm(32, L.cin))
            take(f'{p}.qkv', nn.Conv1d(L.cin, 3 * L.cin, 1))
            take(f'{p}.proj_out', nn.Conv1d(L.cin, L.cin, 1), zero=True)
    take('out.0', nn.GroupNorm(32, ch))
    sd['out.2.positional_embedding'] = torch.randn(ch, res ** 2 + 1) / ch ** 0.5
    take('out.2.qkv_proj', nn.Conv1d(ch, 3 * ch, 1))
    take('out.2.c_proj', nn.Conv1d(ch, cfg.out_channels, 1))
    return sd


HEAD_SCALE_DECIDABLE = 20.0


def scale_classifier_head(sd, head_scale: float = HEAD_SCALE_DECIDABLE):
    """The second, documented classifier fixture: the same constructor draws and seeds, with the output head
    (AttentionPool2d.c_proj, edm/unet.py:61-69) multiplied by `head_scale`, so that the 1000 logits of the random-init classifier have
    a standard deviation of ~3.4 (0.17 x 20) like a trained classifier's instead of 0.17, and the target-class probability of two
    candidates differs by per cent instead of by 1e-4 relative.  Used alike by the oracle and the build (it is a state dict)."""
    out = OrderedDict(sd)
    for k in ('out.2.c_proj.weight', 'out.2.c_proj.bias'):
        out[k] = sd[k] * head_scale
    return out


def refill_degenerate(sd, seed: int, threshold: float = 1e-4):
    """SURVEY.md 8(d) weight rule.  Parameters with max|.| < threshold are re-drawn, in dict order, from
    torch.Generator().manual_seed(seed + 1): tensors with >= 2 dims as U(-b, b), b = sqrt(3 / fan_in),
    fan_in = prod(shape[1:]); 1-D tensors (biases) as U(-0.1, 0.1).  Returns (new dict, refilled names)."""
    g = torch.Generator().manual_seed(seed + 1)
    out, names = OrderedDict(), []
    for k, v in sd.items():
        if v.numel() and float(v.abs().max()) < threshold:
            u = torch.rand(v.shape, generator=g, dtype=torch.float32) * 2 - 1
            if v.dim() >= 2:
                fan_in = int(np.prod(v.shape[1:]))
                v = u * math.sqrt(3.0 / fan_in)
            else:
                v = u * 0.1
            names.append(k)
        out[k] = v
    return out, names


def checksum(sd):
    """Order-sensitive fingerprint of a state dict: (sum, sum|.|, numel) in float64."""
    s = a = 0.0
    n = 0
    for v in sd.values():
        v64 = v.double()
        s += float(v64.sum())
        a += float(v64.abs().sum())
        n += v.numel()
    return dict(sum=s, abs_sum=a, numel=n)


def vae_decoder_state_dict(block_out_channels=(128, 256, 512, 512), layers_per_block=2, latent_channels=4, seed=0):
    """Random-init parameters of the SD VAE *decoder* (+ post_quant_conv) under diffusers' key names
    (sd/diffusers/src/diffusers/models/autoencoders/vae.py:204-279, autoencoder_kl.py:105-110): BASELINE config 4 prescribes
    random-init weights (SD-1.5's cannot be fetched).  Own seeded initialiser, fan-in scaled so activations stay O(1) through the
    17 residual blocks; norms near identity.  Keys in construction order."""
    g = torch.Generator().manual_seed(seed)
    sd = OrderedDict()

    def conv(key, cout, cin, k, gain=1.0):
        sd[key + '.weight'] = torch.randn(cout, cin, k, k, generator=g) * (gain / math.sqrt(cin * k * k))
        sd[key + '.bias'] = torch.randn(cout, generator=g) * 0.05

    def lin(key, cout, cin, gain=1.0):
        sd[key + '.weight'] = torch.randn(cout, cin, generator=g) * (gain / math.sqrt(cin))
        sd[key + '.bias'] = torch.randn(cout, generator=g) * 0.05

    def norm(key, c):
        sd[key + '.weight'] = 1.0 + 0.1 * torch.randn(c, generator=g)
        sd[key + '.bias'] = 0.1 * torch.randn(c, generator=g)

    def resnet(key, cin, cout):
        norm(key + '.norm1', cin); conv(key + '.conv1', cout, cin, 3)
        norm(key + '.norm2', cout); conv(key + '.conv2', cout, cout, 3, gain=0.5)
        if cin != cout:
            conv(key + '.conv_shortcut', cout, cin, 1)

    top = block_out_channels[-1]
    conv('post_quant_conv', latent_channels, latent_channels, 1)
    conv('decoder.conv_in', top, latent_channels, 3)
    resnet('decoder.mid_block.resnets.0', top, top)
    a = 'decoder.mid_block.attentions.0'
    norm(a + '.group_norm', top)
    for n in ('to_q', 'to_k', 'to_v'):
        lin(a + '.' + n, top, top)
    lin(a + '.to_out.0', top, top, gain=0.5)
    resnet('decoder.mid_block.resnets.1', top, top)
    rev = list(reversed(block_out_channels))
    prev = rev[0]
    for i, c in enumerate(rev):
        for j in range(layers_per_block + 1):
            resnet(f'decoder.up_blocks.{i}.resnets.{j}', prev if j == 0 else c, c)
        if i != len(rev) - 1:
            conv(f'decoder.up_blocks.{i}.upsamplers.0.conv', c, c, 3)
        prev = c
    norm('decoder.conv_norm_out', block_out_channels[0])
    conv('decoder.conv_out', 3, block_out_channels[0], 3)
    return sd
