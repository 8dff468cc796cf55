#!/usr/bin/env python3
"""Do the 4-wave and the ping-pong conv kernels agree bit for bit when a residual tile is added (conv_bench reported DIFFER)?
Compares both against the f32-parity kernel on the same (16-bit-rounded) inputs."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from diffusion_tts_amd import ops, _lib

torch.manual_seed(0)
for dt in (torch.bfloat16, torch.float16):
    for name, r, cin, cout, n in [('L0 192->192', 64, 192, 192, 32), ('L0 384->192', 64, 384, 192, 16), ('L1 384->384', 32, 384, 384, 64), ('L2 576->576', 16, 576, 576, 64)]:
        x = torch.randn(n, r, r, cin, device='cuda').to(dt)
        w = (torch.randn(cout, 3, 3, cin, device='cuda') / (cin * 9) ** 0.5).to(dt)
        b = torch.randn(cout, device='cuda')
        res = torch.randn(n, r, r, cout, device='cuda').to(dt)
        outs = {}
        for v in (0, 1):
            _lib.set_tuning('conv_variant', v)
            for rep in range(3):
                o = ops.conv2d(x, w, b, residual=res, gn_stats=True)
            outs[v] = o.clone()
            _lib.set_tuning('conv_variant', -1)
        ref = ops.conv2d(x.float(), w.float(), b, residual=res.float()).to(dt)
        nores = {}
        for v in (0, 1):
            _lib.set_tuning('conv_variant', v)
            nores[v] = ops.conv2d(x, w, b).clone()
            _lib.set_tuning('conv_variant', -1)
        d = (outs[0].float() - outs[1].float()).abs()
        print(f'{str(dt):16s} {name}: v1 vs pp differing {int((d > 0).sum())} of {d.numel()}, max {float(d.max()):.4g}; '
              f'v1 vs f32-ref differing {int((outs[0] != ref).sum())} max {float((outs[0].float() - ref.float()).abs().max()):.4g}; '
              f'pp vs f32-ref differing {int((outs[1] != ref).sum())} max {float((outs[1].float() - ref.float()).abs().max()):.4g}; '
              f'no-residual v1 vs pp differing {int((nores[0] != nores[1]).sum())}', flush=True)
