"""The SD VAE decoder on the HIP kernels: drop-in for the `vae` object of the SD search loop (`vae.decode(z, return_dict=False)[0]`,
`vae.config.scaling_factor`), i.e. for `AutoencoderKL.decode` of the vendored diffusers
(sd/diffusers/src/diffusers/models/autoencoders/autoencoder_kl.py:287-320, vae.py:204-323 `Decoder`).

Decoding is the largest single cost of an SD candidate (2.48 TFLOP per 512x512 image) and the reference decodes
the candidates of an iteration one at a time; here all N go through as one batch, on the same kernels as the EDM path:
  3x3 / 1x1 convs           -> dts_conv2d (implicit GEMM on MFMA; nearest-2x upsample fused into the gather; residual add in the epilogue)
  GroupNorm(32, eps 1e-6)+SiLU -> dts_gn_* (statistics fused into the producing conv's epilogue where possible)
  mid-block attention       -> dts_attention (one head of dim 512: two 256-wide value slices)
  conv_out (128 -> 3)       -> dts_conv_out3
Activations are NHWC in `dtype` (float16 like the reference pipeline, or bfloat16); the 4 latent channels are zero-padded to 64 for
the MFMA conv (cin % 64 == 0).  Parameters: a state dict with diffusers' key names (`AutoencoderKL.state_dict()` / the safetensors
file of SD-1.5's vae/).  Stock configuration only: group norm, SiLU, one attention in the mid block, UpDecoderBlock2D.
"""
import math
import types

import torch

from . import ops


class VAEDecoder:
    def __init__(self, state_dict, block_out_channels=(128, 256, 512, 512), layers_per_block=2, latent_channels=4, norm_num_groups=32,
                 scaling_factor=0.18215, device='cuda', dtype=torch.float16):
        if not torch.cuda.is_available():
            raise RuntimeError('VAEDecoder (HIP) needs a GPU: there is no CPU fallback in this package')
        if dtype not in (torch.float16, torch.bfloat16):
            raise ValueError('VAEDecoder: activations are float16 or bfloat16 (head dim 512 attention is 16-bit only)')
        self.device, self.dtype = torch.device(device), dtype
        self.boc, self.lpb, self.groups, self.lat = tuple(block_out_channels), layers_per_block, norm_num_groups, latent_channels
        self.config = types.SimpleNamespace(scaling_factor=scaling_factor, block_out_channels=list(block_out_channels),
                                            latent_channels=latent_channels, force_upcast=False)
        self.decodes = 0
        self._load(state_dict)

    @classmethod
    def from_pretrained(cls, path, device='cuda', dtype=torch.float16):
        """Reads a diffusers `vae/` directory (or a single `.safetensors` file): `config.json` for the shape and
        `diffusion_pytorch_model[.fp16].safetensors` for the parameters -- the on-disk format of SD-1.5's VAE.
        Only the decoder half (`decoder.*`, `post_quant_conv.*`) is read; tensors are loaded lazily, so the encoder never leaves disk."""
        import json
        import os
        from safetensors import safe_open
        cfg = {}
        if os.path.isdir(path):
            cfg_file = os.path.join(path, 'config.json')
            if os.path.exists(cfg_file):
                with open(cfg_file) as f:
                    cfg = json.load(f)
            names = [n for n in ('diffusion_pytorch_model.safetensors', 'diffusion_pytorch_model.fp16.safetensors') if
                     os.path.exists(os.path.join(path, n))]
            if not names:
                raise FileNotFoundError(f'{path}: no diffusion_pytorch_model[.fp16].safetensors (a .bin pickle is not read: convert it to safetensors)')
            path = os.path.join(path, names[0])
        for key, want in (('act_fn', 'silu'), ('_class_name', 'AutoencoderKL')):
            if key in cfg and cfg[key] != want:
                raise ValueError(f'VAEDecoder: {key}={cfg[key]!r} is not the stock SD VAE ({want!r})')
        for t in cfg.get('up_block_types', []):
            if t != 'UpDecoderBlock2D':
                raise ValueError(f'VAEDecoder: up block {t!r} unsupported')
        sd = {}
        with safe_open(path, framework='pt', device='cpu') as f:
            for k in f.keys():
                if k.startswith('decoder.') or k.startswith('post_quant_conv.'):
                    sd[k] = f.get_tensor(k)
        if not sd:
            raise ValueError(f'{path}: no decoder.* tensors')
        kw = {k: cfg[k] for k in ('layers_per_block', 'latent_channels', 'norm_num_groups', 'scaling_factor') if k in cfg}
        if 'block_out_channels' in cfg:
            kw['block_out_channels'] = tuple(cfg['block_out_channels'])
        return cls(sd, device=device, dtype=dtype, **kw)

    # ------------------------------------------------------------------------------------------
    def _f(self, t):
        return t.detach().to(self.device, torch.float32).contiguous()

    def _conv_params(self, sd, key, pad_in=None, pad_out=None):
        w, b = self._f(sd[key + '.weight']), self._f(sd[key + '.bias'])
        if w.dim() == 2:                                           # Linear -> 1x1 conv
            w = w[:, :, None, None].contiguous()
        if pad_in is not None and w.shape[1] < pad_in:             # zero input channels up to the MFMA granule
            w = torch.cat([w, torch.zeros(w.shape[0], pad_in - w.shape[1], *w.shape[2:], device=w.device)], 1).contiguous()
        if pad_out is not None and w.shape[0] < pad_out:
            w = torch.cat([w, torch.zeros(pad_out - w.shape[0], *w.shape[1:], device=w.device)], 0).contiguous()
            b = torch.cat([b, torch.zeros(pad_out - b.shape[0], device=b.device)]).contiguous()
        return ops.pack_conv_weight(w, self.dtype), b

    def _resnet_params(self, sd, key):
        P = types.SimpleNamespace()
        P.g1, P.b1 = self._f(sd[key + '.norm1.weight']), self._f(sd[key + '.norm1.bias'])
        P.w1, P.c1 = self._conv_params(sd, key + '.conv1')
        P.g2, P.b2 = self._f(sd[key + '.norm2.weight']), self._f(sd[key + '.norm2.bias'])
        P.w2, P.c2 = self._conv_params(sd, key + '.conv2')
        P.ws = P.cs = None
        if key + '.conv_shortcut.weight' in sd:
            P.ws, P.cs = self._conv_params(sd, key + '.conv_shortcut')
        return P

    def _load(self, sd):
        top = self.boc[-1]
        self.pq = self._conv_params(sd, 'post_quant_conv', pad_in=64, pad_out=64) if 'post_quant_conv.weight' in sd else None
        self.conv_in = self._conv_params(sd, 'decoder.conv_in', pad_in=64)
        self.mid0 = self._resnet_params(sd, 'decoder.mid_block.resnets.0')
        self.mid1 = self._resnet_params(sd, 'decoder.mid_block.resnets.1')
        a = 'decoder.mid_block.attentions.0'
        A = types.SimpleNamespace()
        A.g, A.b = self._f(sd[a + '.group_norm.weight']), self._f(sd[a + '.group_norm.bias'])
        wq = torch.cat([self._f(sd[a + f'.to_{n}.weight']) for n in 'qkv'], 0)[:, :, None, None].contiguous()     # q | k | v blocks
        A.wqkv = ops.pack_conv_weight(wq, self.dtype)
        A.bqkv = torch.cat([self._f(sd[a + f'.to_{n}.bias']) for n in 'qkv']).contiguous()
        A.wo, A.bo = self._conv_params(sd, a + '.to_out.0')
        A.dim = top
        self.attn = A
        self.up = []
        for i in range(len(self.boc)):
            res = [self._resnet_params(sd, f'decoder.up_blocks.{i}.resnets.{j}') for j in range(self.lpb + 1)]
            ups = self._conv_params(sd, f'decoder.up_blocks.{i}.upsamplers.0.conv') if i != len(self.boc) - 1 else None
            self.up.append((res, ups))
        self.out_g, self.out_b = self._f(sd['decoder.conv_norm_out.weight']), self._f(sd['decoder.conv_norm_out.bias'])
        self.out_w = self._f(sd['decoder.conv_out.weight']).permute(0, 2, 3, 1).contiguous()       # [3][kh][kw][c]
        self.out_cb = self._f(sd['decoder.conv_out.bias'])
        torch.cuda.synchronize(self.device)

    # ------------------------------------------------------------------------------------------
    def _resnet(self, x, P):
        """ResnetBlock2D.forward with temb None: norm1-silu-conv1-norm2-silu-conv2, + (1x1 shortcut of) the input."""
        G = self.groups
        h = ops.group_norm(x, G, 1e-6, P.g1, P.b1, silu=True)
        h = ops.conv2d(h, P.w1, P.c1, gn_stats=True)
        h = ops.group_norm(h, G, 1e-6, P.g2, P.b2, silu=True)
        sk = x if P.ws is None else ops.conv2d(x, P.ws, P.cs)
        return ops.conv2d(h, P.w2, P.c2, residual=sk, gn_stats=True)

    def _attention(self, x):
        """AttnProcessor2_0 on [n, hw, c]: group norm, fused q|k|v projection, one head of dim c, output projection + residual."""
        A = self.attn
        n, hh, ww, c = x.shape
        hn = ops.group_norm(x, self.groups, 1e-6, A.g, A.b, silu=False)
        qkv = ops.conv2d(hn, A.wqkv, A.bqkv)
        o = ops.attention(qkv.view(n, hh * ww, 3 * c), 1, 1.0 / math.sqrt(c))
        return ops.conv2d(o.view(n, hh, ww, c), A.wo, A.bo, residual=x, gn_stats=True)

    @torch.no_grad()
    def decode(self, z, return_dict=False, generator=None):
        """z [n, latent_channels, h, w] (already divided by the scaling factor by the caller, pipeline...:1112) -> ([n, 3, 8h, 8w],)."""
        z = z.to(self.device)
        n = z.shape[0]
        x = ops.nchw_to_nhwc_pad(z.float().contiguous(), self.dtype, 64)
        if self.pq is not None:
            x = ops.conv2d(x, self.pq[0], self.pq[1])                           # 1x1 on the 4 (of 64) live channels
        x = ops.conv2d(x, self.conv_in[0], self.conv_in[1], gn_stats=True)
        x = self._resnet(x, self.mid0)
        x = self._attention(x)
        x = self._resnet(x, self.mid1)
        for res, ups in self.up:
            for P in res:
                x = self._resnet(x, P)
            if ups is not None:
                x = ops.conv2d(x, ups[0], ups[1], up=True, gn_stats=True)       # nearest-2x fused into the conv's gather
        h = ops.group_norm(x, self.groups, 1e-6, self.out_g, self.out_b, silu=True)
        img = ops.conv_out3(h, self.out_w, self.out_cb).to(self.dtype)
        self.decodes += n
        if return_dict:
            return types.SimpleNamespace(sample=img)
        return (img,)
