"""The ImageNet-64 noisy classifier (scorer backbone) on the HIP kernels.

Mirrors the inference path of `EncoderUNetModel(pool='attention', use_scale_shift_norm=True,
resblock_updown=True)` (edm/unet.py:701-912) with parameters under the reference's state-dict keys
('input_blocks.1.0.in_layers.2.weight', ...).  In the reference this network runs on the CPU
(main.py:69 never moves the scorer to the device); here it runs on the GPU next to the
denoiser, reusing the same conv / GroupNorm / attention kernels:
  ResBlock._forward      edm/unet.py:254-274  (GroupNorm32+SiLU [+avg-pool], conv, scale-shift norm, conv, skip)
  AttentionBlock._forward edm/unet.py:317-323 (legacy head order, regrouped at weight-pack time)
  AttentionPool2d.forward edm/unet.py:61-69
"""
import math
from typing import Dict

import torch

from . import ops
from .config import ClassifierConfig, ClsLayer, classifier_layers


def _qkv_perm_legacy(heads: int, d: int) -> torch.Tensor:
    """QKVAttentionLegacy (edm/unet.py:363-365): qkv.reshape(bs*heads, 3d, T).split(d) => source channel
    head*3d + s*d + ch.  Destination: s*C + head*d + ch."""
    s, h, c = torch.meshgrid(torch.arange(3), torch.arange(heads), torch.arange(d), indexing='ij')
    return (h * 3 * d + s * d + c).reshape(-1).to(torch.int32)


class _P:
    pass


class EncoderUNetModel:
    def __init__(self, cfg: ClassifierConfig, state_dict: Dict[str, torch.Tensor], device='cuda', dtype=ops.F16X3):
        if not torch.cuda.is_available():
            raise RuntimeError('EncoderUNetModel (HIP) needs a GPU: there is no CPU fallback in this package')
        self.cfg, self.device, self.dtype = cfg, torch.device(device), dtype
        self.act_dtype = ops.act_dtype(dtype)            # float32 activations in the split-precision mode (ops.F16X3)
        self.x3 = dtype == ops.F16X3
        from .graphs import GraphCache
        self._graphs = GraphCache(self._device_forward)
        f = lambda t: t.detach().to(self.device, torch.float32).contiguous()
        sd, dt = state_dict, dtype
        mc = cfg.model_channels
        half = mc // 2
        self.freqs = torch.exp(-math.log(10000) * torch.arange(0, half, dtype=torch.float32) / half).to(self.device)
        self.te0_w, self.te0_b = f(sd['time_embed.0.weight']), f(sd['time_embed.0.bias'])
        self.te2_w, self.te2_b = f(sd['time_embed.2.weight']), f(sd['time_embed.2.bias'])
        self.layers, self.ch, self.res = classifier_layers(cfg)
        self.params: Dict[str, _P] = {}
        emb_w, emb_b, off = [], [], 0
        hc = cfg.num_head_channels
        for L in self.layers:
            P, p = _P(), L.prefix
            if L.kind == 'conv_in':
                P.w, P.b = f(sd[f'{p}.weight']), f(sd[f'{p}.bias'])
            elif L.kind == 'res':
                P.g0, P.b0 = f(sd[f'{p}.in_layers.0.weight']), f(sd[f'{p}.in_layers.0.bias'])
                P.w0, P.cb0 = ops.pack_conv_weight(f(sd[f'{p}.in_layers.2.weight']), dt), f(sd[f'{p}.in_layers.2.bias'])
                emb_w.append(sd[f'{p}.emb_layers.1.weight'])
                emb_b.append(sd[f'{p}.emb_layers.1.bias'])
                P.off, off = off, off + 2 * L.cout
                P.g1, P.b1 = f(sd[f'{p}.out_layers.0.weight']), f(sd[f'{p}.out_layers.0.bias'])
                P.w1, P.cb1 = ops.pack_conv_weight(f(sd[f'{p}.out_layers.3.weight']), dt), f(sd[f'{p}.out_layers.3.bias'])
                P.sw = P.sb = None
                if L.cin != L.cout:
                    P.sw, P.sb = ops.pack_conv_weight(f(sd[f'{p}.skip_connection.weight']), dt), f(sd[f'{p}.skip_connection.bias'])
                    P.cb1_skip = (P.cb1 + P.sb).contiguous()         # conv1's bias with the skip convolution folded in (ops.conv_folds_skip)
            else:
                heads = L.cin // hc
                perm = _qkv_perm_legacy(heads, hc).to(self.device)
                P.g, P.b = f(sd[f'{p}.norm.weight']), f(sd[f'{p}.norm.bias'])
                P.wqkv = ops.pack_conv_weight(f(sd[f'{p}.qkv.weight']), dt, out_perm=perm)
                P.bqkv = f(sd[f'{p}.qkv.bias'])[perm.long()].contiguous()
                P.wproj, P.bproj = ops.pack_conv_weight(f(sd[f'{p}.proj_out.weight']), dt), f(sd[f'{p}.proj_out.bias'])
                P.heads = heads
            self.params[p] = P
        self.emb_w = ops.pack_conv_weight(f(torch.cat(emb_w, 0))[:, :, None, None].contiguous(), dt)
        self.emb_b = f(torch.cat(emb_b, 0))
        self.emb_total = off
        self.out_g, self.out_b = f(sd['out.0.weight']), f(sd['out.0.bias'])
        self.pos = f(sd['out.2.positional_embedding'])
        self.pool_wqkv = ops.pack_conv_weight(f(sd['out.2.qkv_proj.weight']), dt)
        self.pool_bqkv = f(sd['out.2.qkv_proj.bias'])
        self.cproj_w = f(sd['out.2.c_proj.weight']).reshape(cfg.out_channels, self.ch).contiguous()
        self.cproj_b = f(sd['out.2.c_proj.bias'])
        torch.cuda.synchronize(self.device)

    def _res(self, L: ClsLayer, x, emb_all):
        P = self.params[L.prefix]
        # (split-precision mode, blocks with a 1x1 skip convolution: the norm pass also writes the skip convolution's operand image of the raw
        # [2x2-averaged] input: networks.EDMPrecond._block)
        raw = self.x3 and P.sw is not None
        h = ops.group_norm(x, 32, 1e-5, P.g0, P.b0, silu=True, pool=L.down, split_out=self.x3, raw_split=raw)
        if raw:
            h, x = h
        elif L.down:
            x = ops.resample2x(x, up=False)
        h = ops.conv2d(h, P.w0, P.cb0, gn_stats=True)
        ss = emb_all[:, P.off:P.off + 2 * L.cout]
        h = ops.group_norm(h, 32, 1e-5, P.g1, P.b1, scale_shift=ss, silu=True, split_out=self.x3)
        if raw and ops.conv_folds_skip(h, P.w1, (x, P.sw, False)):          # the 1x1 skip convolution as a second K loop of conv1's launch
            return ops.conv2d(h, P.w1, P.cb1_skip, skip=(x, P.sw, False), gn_stats=True)
        sk = x if P.sw is None else ops.conv2d(x, P.sw, P.sb)
        return ops.conv2d(h, P.w1, P.cb1, residual=sk, gn_stats=True)

    def _attn(self, L: ClsLayer, x):
        P = self.params[L.prefix]
        n, hh, ww, c = x.shape
        hn = ops.group_norm(x, 32, 1e-5, P.g, P.b, silu=False, split_out=self.x3)
        fx3 = self.x3 and ops.attention_x3_ok(hh * ww, c // P.heads)        # (see networks.EDMPrecond._block)
        qkv = ops.conv2d(hn, P.wqkv, P.bqkv, out_split2=fx3)
        a = ops.attention(qkv.view(n, hh * ww, 3 * c), P.heads, 1.0 / math.sqrt(c // P.heads), x3=self.x3, split_out=fx3)
        a = ops.SplitAct(a.data.view(n, hh, ww, 2 * c), c) if fx3 else a.view(n, hh, ww, c)
        return ops.conv2d(a, P.wproj, P.bproj, residual=x, gn_stats=True)

    @torch.no_grad()
    def __call__(self, x, timesteps):
        """x f32 NCHW [n,3,R,R] on the GPU (values in [0,1], scorers.py:153); timesteps [n] -> logits f32 [n, K]."""
        t = timesteps.to(self.device, torch.float32).contiguous()
        return self._graphs(x.to(self.device, torch.float32).contiguous(), t)    # HIP-graph replay of the fixed-shape forward

    @torch.no_grad()
    def _device_forward(self, x, t):
        n = x.shape[0]
        emb = ops.pos_embedding(t, self.freqs)                                   # nn_utils.py:103-121
        emb = ops.linear(emb, self.te0_w, self.te0_b, act_out=True)
        emb = ops.linear(emb, self.te2_w, self.te2_b, act_out=True)              # SiLU of emb_layers[0], shared by all blocks
        e = ops.cast_from_f32(emb, self.act_dtype)
        emb_all = ops.conv2d(e.view(1, n, 1, -1), self.emb_w, self.emb_b).view(n, self.emb_total)
        h = None
        for L in self.layers:
            if L.kind == 'conv_in':
                P = self.params[L.prefix]
                h = ops.conv_in3(x.contiguous(), P.w, P.b, L.cout, self.act_dtype)
            elif L.kind == 'res':
                h = self._res(L, h, emb_all)
            else:
                h = self._attn(L, h)
        h = ops.group_norm(h, 32, 1e-5, self.out_g, self.out_b, silu=True)
        tok = ops.attnpool_tokens(h, self.pos)                                   # [n, hw+1, c]
        nt = tok.shape[1]
        qkv = ops.conv2d(tok.view(n, nt, 1, self.ch), self.pool_wqkv, self.pool_bqkv)
        heads = self.ch // self.cfg.num_head_channels
        a = ops.attention(qkv.view(n, nt, 3 * self.ch), heads, 1.0 / math.sqrt(self.cfg.num_head_channels), x3=self.x3)
        return ops.linear(ops.take_token(a, 0), self.cproj_w, self.cproj_b)
