"""EDM denoisers on the HIP kernels: the host-side mirror of the reference's network interface.

Same call surface as the reference objects the sampler touches (edm/training/networks.py:632-671 and its
use at edm/main.py:80,84,87,882):  `net(x, sigma, class_labels) -> float32 [n, C, R, R]`, `net.round_sigma`,
`net.img_resolution`, `net.img_channels`, `net.label_dim`, `net.sigma_min/max/data`, `net.to(device)`.
Parameters are taken from a flat state dict with the reference's own key names
('model.enc.64x64_block0.conv0.weight', ...), so a state dict exported from a reference module loads as is.

Every tensor op below is a libdts_hip kernel (ops.py); PyTorch only owns the memory and the stream.
Activations are NHWC in `dtype`.  Compute modes: ops.F16X3 (DEFAULT: split precision on the 16-bit matrix cores, the reference's fp32
results -- same selected candidates as the reference's fp32 run, networks.py:658 on a CPU), float32 (parity mode on the f32 matrix
instruction), bfloat16 / float16 (throughput modes: another sample after the first near-tied pick).

Per UNetBlock (networks.py:166-187) the launch sequence is
  gn_coef+gn_apply(SiLU, [2x2 pool]) -> conv0 [fused nearest-up gather] -> gn_coef(+scale/shift)+gn_apply(SiLU)
  -> [1x1 skip conv | resample] -> conv1 (+residual, *skip_scale)
  -> [gn -> qkv 1x1 -> fused attention -> proj 1x1 (+residual, *skip_scale)]
and the channel concat of the decoder (networks.py:458) is never materialised: the norm and the 1x1 skip
conv read both sources.
"""
import math
import os
from typing import Dict, Optional

import numpy as np
import torch

from . import ops
from .config import EDMConfig, Block, edm_blocks


def _qkv_perm_edm(heads: int, d: int) -> torch.Tensor:
    """networks.py:182: qkv.reshape(n*heads, d, 3, hw).unbind(2) => source channel head*3d + ch*3 + s.
    Destination layout: s*C + head*d + ch."""
    s, h, c = torch.meshgrid(torch.arange(3), torch.arange(heads), torch.arange(d), indexing='ij')
    return (h * 3 * d + c * 3 + s).reshape(-1).to(torch.int32)


class _BlockParams:
    pass


class EDMPrecond:
    """EDMPrecond + DhariwalUNet / SongUNet(DDPM++) forward on MI355X."""

    def __init__(self, cfg: EDMConfig, state_dict: Dict[str, torch.Tensor], device='cuda', dtype=ops.F16X3):
        if not torch.cuda.is_available():
            raise RuntimeError('EDMPrecond (HIP) needs a GPU: there is no CPU fallback in this package')
        self.cfg = cfg
        self.device = torch.device(device)
        self.dtype = dtype                               # compute mode: float32 (parity) | bfloat16 | float16 | ops.F16X3 (split precision)
        self.act_dtype = ops.act_dtype(dtype)            # storage type of the activations
        self.x3 = dtype == ops.F16X3                     # GroupNorm results that only feed a convolution leave as its split operand image
        self.img_resolution, self.img_channels, self.label_dim = cfg.img_resolution, cfg.img_channels, cfg.label_dim
        self.sigma_min, self.sigma_max, self.sigma_data = cfg.sigma_min, cfg.sigma_max, cfg.sigma_data
        self.use_fp16 = dtype == torch.float16
        self.adm = cfg.arch == 'adm'
        self.eps = 1e-5 if self.adm else 1e-6
        self.skip_scale = 1.0 if self.adm else math.sqrt(0.5)
        self.evals = 0                               # rows pushed through the denoiser (BASELINE metric unit)
        # GroupNorm apply inside the consuming 3x3 conv (built, bit-identical, tested) -- OFF by default: measured slower than the
        # separate apply pass (profiles/r02_gn_fusion.txt); DTS_GN_FUSE=1 turns it on
        self.fuse_gn = os.environ.get('DTS_GN_FUSE', '0') == '1'
        from .graphs import GraphCache
        self._graphs = GraphCache(self._device_forward)     # HIP-graph replay of the fixed-shape forward (graphs.py)
        sd = {k[len('model.'):]: v for k, v in state_dict.items() if k.startswith('model.')}
        self._load(sd)

    # ------------------------------------------------------------------------------------------
    def _dev(self, t):
        return t.detach().to(self.device, torch.float32).contiguous()

    def _load(self, sd):
        cfg, dev, dt = self.cfg, self.device, self.dtype
        f = self._dev
        mc = cfg.model_channels
        # embedding MLP (f32, tiny)
        self.map0_w, self.map0_b = f(sd['map_layer0.weight']), f(sd['map_layer0.bias'])
        self.map1_w, self.map1_b = f(sd['map_layer1.weight']), f(sd['map_layer1.bias'])
        if cfg.label_dim:
            self.label_w = f(sd['map_label.weight'])
            self.label_b = f(sd['map_label.bias']) if 'map_label.bias' in sd else None
        half = mc // 2
        freqs = torch.arange(0, half, dtype=torch.float32)
        freqs = freqs / (half - (0 if self.adm else 1))           # endpoint=True for DDPM++ (networks.py:269)
        self.freqs = ((1 / 10000) ** freqs).to(dev)
        enc, dec, cfin = edm_blocks(cfg)
        self.enc, self.dec = enc, dec
        first = enc[0]
        self.conv_in_w, self.conv_in_b = f(sd[f'{first.name}.weight']), f(sd[f'{first.name}.bias'])
        self.blocks: Dict[str, _BlockParams] = {}
        aff_w, aff_b, off = [], [], 0
        for b in enc[1:] + dec:
            P = _BlockParams()
            n = b.name
            g = lambda k: sd[f'{n}.{k}']
            P.g0, P.b0 = f(g('norm0.weight')), f(g('norm0.bias'))
            P.w0, P.cb0 = ops.pack_conv_weight(f(g('conv0.weight')), dt), f(g('conv0.bias'))
            caff = b.cout * (2 if self.adm else 1)
            aff_w.append(g('affine.weight'))
            aff_b.append(g('affine.bias'))
            P.aff_off, P.caff = off, caff
            off += caff
            P.g1, P.b1 = f(g('norm1.weight')), f(g('norm1.bias'))
            P.w1, P.cb1 = ops.pack_conv_weight(f(g('conv1.weight')), dt), f(g('conv1.bias'))
            P.skip_w = P.skip_b = None
            if f'{n}.skip.weight' in sd:
                P.skip_w, P.skip_b = ops.pack_conv_weight(f(g('skip.weight')), dt), f(g('skip.bias'))
                P.cb1_skip = (P.cb1 + P.skip_b).contiguous()         # the bias of conv1 with the skip convolution folded in (ops.conv_folds_skip)
            if b.heads:
                d = b.cout // b.heads
                perm = _qkv_perm_edm(b.heads, d).to(dev)
                P.g2, P.b2 = f(g('norm2.weight')), f(g('norm2.bias'))
                P.wqkv = ops.pack_conv_weight(f(g('qkv.weight')), dt, out_perm=perm)
                P.bqkv = f(g('qkv.bias'))[perm.long()].contiguous()
                P.wproj, P.bproj = ops.pack_conv_weight(f(g('proj.weight')), dt), f(g('proj.bias'))
            self.blocks[n] = P
        # all blocks' affine layers as ONE GEMM over the shared embedding (networks.py:152,170)
        self.aff_w = ops.pack_conv_weight(f(torch.cat(aff_w, 0))[:, :, None, None].contiguous(), dt)
        self.aff_b = f(torch.cat(aff_b, 0))
        self.aff_total = off
        r = cfg.img_resolution
        on, oc = ('out_norm', 'out_conv') if self.adm else (f'dec.{r}x{r}_aux_norm', f'dec.{r}x{r}_aux_conv')
        self.out_g, self.out_b = f(sd[f'{on}.weight']), f(sd[f'{on}.bias'])
        self.out_w = f(sd[f'{oc}.weight']).permute(0, 2, 3, 1).contiguous()          # [3][kh][kw][c]
        self.out_cb = f(sd[f'{oc}.bias'])
        self.cfin = cfin
        torch.cuda.synchronize(dev)

    # ------------------------------------------------------------------------------------------
    def to(self, device):
        return self

    def eval(self):
        return self

    def round_sigma(self, sigma):
        return torch.as_tensor(sigma)

    @staticmethod
    def _groups(c):
        return min(32, c // 4)                                                     # networks.py:99

    def _embedding(self, c_noise, class_labels, n):
        """Mapping network: networks.py:437-447 (ADM) / :322-332 (DDPM++).  f32 [n, emb_channels]."""
        cfg = self.cfg
        emb = ops.pos_embedding(c_noise, self.freqs, swap=not self.adm)           # :437 / :322-323
        if self.adm:
            h = ops.linear(emb, self.map0_w, self.map0_b, act_out=True)
            if not cfg.label_dim:
                return ops.linear(h, self.map1_w, self.map1_b, act_out=True)
            h = ops.linear(h, self.map1_w, self.map1_b)
            lab = self._labels(class_labels, n)
            return ops.linear(lab, self.label_w, None, out=h, accumulate=True, act_out=True)   # silu(emb + map_label)
        if cfg.label_dim:
            lab = self._labels(class_labels, n) * math.sqrt(cfg.label_dim)        # :328
            ops.linear(lab.contiguous(), self.label_w, self.label_b, out=emb, accumulate=True)
        h = ops.linear(emb, self.map0_w, self.map0_b, act_out=True)
        return ops.linear(h, self.map1_w, self.map1_b, act_out=True)

    def _labels(self, class_labels, n):
        L = self.cfg.label_dim
        if class_labels is None:
            lab = torch.zeros([1, L], device=self.device)                           # networks.py:657
        else:
            lab = class_labels.to(self.device, torch.float32).reshape(-1, L)
        if lab.shape[0] == 1 and n > 1:
            lab = lab.expand(n, L)
        if lab.shape[0] != n:
            raise ValueError(f'class_labels rows {lab.shape[0]} != batch {n}')
        return lab.contiguous()

    def _block(self, b: Block, x1, x2, aff):
        P = self.blocks[b.name]
        G = self._groups
        ss = aff[:, P.aff_off:P.aff_off + P.caff]
        bnc = None if self.adm else ss
        skip_src = None
        if self.fuse_gn and not b.down and ops.conv_fuses_gn(x1, P.w0, x2=x2, up=b.up):
            # norm0 + SiLU applied inside conv0 on its staged input tile: the normalised (and, in the decoder, concatenated) tensor
            # is never written (networks.py:168)
            coef = ops.gn_coefficients(x1, G(b.cin), self.eps, P.g0, P.b0, x2=x2)
            h = ops.conv2d(x1, P.w0, P.cb0, x2=x2, up=b.up, bias_nc=bnc, gn_stats=True, gn_coef=coef, gn_silu=True)
        else:
            # split-precision mode, blocks with a 1x1 skip convolution: the norm0 pass also writes the operand image of the RAW input
            # (2x2-averaged in a down block), which the skip convolution reads -- no dts_split3_f16 pass over the same tensor
            raw = self.x3 and P.skip_w is not None and os.environ.get('DTS_X3_RAW_SPLIT', '1') != '0'
            h = ops.group_norm(x1, G(b.cin), self.eps, P.g0, P.b0, x2=x2, silu=True, pool=b.down, split_out=self.x3, raw_split=raw)
            if raw:
                h, skip_src = h
            h = ops.conv2d(h, P.w0, P.cb0, up=b.up, bias_nc=bnc, gn_stats=True)
        fuse1 = self.fuse_gn and ops.conv_fuses_gn(h, P.w1)
        if fuse1:        # norm1 (+ adaptive scale/shift) + SiLU inside conv1 (networks.py:173-175)
            coef1 = ops.gn_coefficients(h, G(b.cout), self.eps, P.g1, P.b1, scale_shift=ss if self.adm else None)
        else:
            h = ops.group_norm(h, G(b.cout), self.eps, P.g1, P.b1, scale_shift=ss if self.adm else None, silu=True, split_out=self.x3)
        if P.skip_w is not None and skip_src is not None and ops.conv_folds_skip(h, P.w1, (skip_src, P.skip_w, b.up)):
            # split-precision mode: the 1x1 skip convolution is a second K loop of conv1's launch (no f32 skip tensor written and read back)
            sk = None
        elif P.skip_w is not None and skip_src is not None:
            sk = ops.conv2d(skip_src, P.skip_w, P.skip_b, up=b.up)
        elif P.skip_w is not None:
            src1, src2 = (ops.resample2x(x1, up=False), None) if b.down else (x1, x2)
            sk = ops.conv2d(src1, P.skip_w, P.skip_b, x2=src2, up=b.up)
        elif b.up or b.down:
            sk = ops.resample2x(x1, up=b.up)
        else:
            sk = x1
        if sk is None:
            x = ops.conv2d(h, P.w1, P.cb1_skip, skip=(skip_src, P.skip_w, b.up), out_scale=self.skip_scale, gn_stats=True)
        else:
            x = ops.conv2d(h, P.w1, P.cb1, residual=sk, out_scale=self.skip_scale, gn_stats=True,
                           gn_coef=coef1 if fuse1 else None, gn_silu=True)
        if b.heads:
            n, hh, ww, c = x.shape
            hn = ops.group_norm(x, G(c), self.eps, P.g2, P.b2, silu=False, split_out=self.x3)
            # split-precision mode: qkv leaves the projection as the attention's operand image and the attention's result as the proj
            # convolution's (no f32 tensors and no split passes in between) wherever the split-precision attention kernel runs
            fx3 = self.x3 and ops.attention_x3_ok(hh * ww, c // b.heads)
            qkv = ops.conv2d(hn, P.wqkv, P.bqkv, out_split2=fx3)
            a = ops.attention(qkv.view(n, hh * ww, 3 * c), b.heads, 1.0 / math.sqrt(c // b.heads), x3=self.x3, split_out=fx3)
            a = ops.SplitAct(a.data.view(n, hh, ww, 2 * c), c) if fx3 else a.view(n, hh, ww, c)
            x = ops.conv2d(a, P.wproj, P.bproj, residual=x, out_scale=self.skip_scale, gn_stats=True)
        return x

    @torch.no_grad()
    def unet(self, xin, c_noise, class_labels):
        """F_x = model(c_in*x, c_noise, labels): xin f32 NCHW -> f32 NCHW."""
        n = xin.shape[0]
        emb = self._embedding(c_noise, class_labels, n)
        e = ops.cast_from_f32(emb, self.act_dtype)
        aff = ops.conv2d(e.view(1, n, 1, -1), self.aff_w, self.aff_b).view(n, self.aff_total)
        x = ops.conv_in3(xin, self.conv_in_w, self.conv_in_b, self.enc[0].cout, self.act_dtype)
        skips = [x]
        for b in self.enc[1:]:
            x = self._block(b, x, None, aff)
            skips.append(x)
        for b in self.dec:
            x2 = skips.pop() if x.shape[-1] != b.cin else None
            x = self._block(b, x, x2, aff)
        h = ops.group_norm(x, self._groups(self.cfin), 1e-5 if self.adm else 1e-6, self.out_g, self.out_b, silu=True)
        return ops.conv_out3(h, self.out_w, self.out_cb)

    @torch.no_grad()
    def _device_forward(self, x, sigma, class_labels):
        """The fixed-shape device part of the forward (all inputs already on the GPU): what graphs.GraphCache captures."""
        xin, coef = ops.edm_precond_in(x, sigma, self.sigma_data)
        F = self.unet(xin, coef[:, 3].contiguous(), class_labels)
        return ops.edm_precond_out(x, F, coef)

    @torch.no_grad()
    def __call__(self, x, sigma, class_labels=None, force_fp32=False, **model_kwargs):
        """EDMPrecond.forward (networks.py:654-668), same signature.  `force_fp32` only matters to the reference when the checkpoint says
        use_fp16 and the device is a GPU (networks.py:658); here the compute mode is fixed at construction (`dtype`), and every mode but
        float16 already computes at least at the reference's fp32 precision, so the flag is honoured where it can be (a float16 network
        refuses it instead of silently answering in half precision).  `model_kwargs` go to the U-Net in the reference (`augment_labels`
        is the only one its models take, None at inference): anything else is an error, as it would be there."""
        if force_fp32 and self.dtype == torch.float16:
            raise ValueError('force_fp32=True: this network was built with dtype=float16; build it with dtype=ops.F16X3 (default) or float32')
        aug = model_kwargs.pop('augment_labels', None)
        if aug is not None:
            raise NotImplementedError('augment_labels: training-time augmentation conditioning is outside the sampling path (networks.py:330-331)')
        if model_kwargs:
            raise TypeError(f'unexpected keyword arguments {sorted(model_kwargs)} (the reference U-Nets would raise too)')
        x = x.to(self.device, torch.float64).contiguous()
        n = x.shape[0]
        sigma = torch.as_tensor(sigma).to(self.device, torch.float64).reshape(-1).contiguous()
        if sigma.numel() not in (1, n):
            raise ValueError(f'sigma has {sigma.numel()} entries for batch {n}')
        if class_labels is not None:
            class_labels = class_labels.to(self.device, torch.float32).contiguous()
        self.evals += n
        return self._graphs(x, sigma, class_labels)
