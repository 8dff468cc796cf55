#!/usr/bin/env python3
"""GroupNorm path comparison (fused single launch vs partial+coef+apply) on the ADM level shapes (GPU box only)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusion_tts_amd import ops

def timeit(fn, iters=10):
    for _ in range(2): fn()
    ts = []
    for _ in range(iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda._sleep(200_000); e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort(); return ts[len(ts)//2] * 1e3

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
for name, r, c in [('L0', 64, 192), ('L0cat', 64, 384), ('L1', 32, 384), ('L1cat', 32, 768), ('L2', 16, 576), ('L2cat', 16, 1152), ('L3', 8, 768), ('L3cat', 8, 1536)]:
    x = torch.randn(n, r, r, c, device='cuda').to(torch.bfloat16)
    g, b = torch.randn(c, device='cuda'), torch.randn(c, device='cuda')
    ss = torch.randn(n, 2 * c, device='cuda').to(torch.bfloat16)
    t_split = timeit(lambda: ops.group_norm(x, 32, 1e-5, g, b, scale_shift=ss, path='split'))
    try:
        t_fused = timeit(lambda: ops.group_norm(x, 32, 1e-5, g, b, scale_shift=ss, path='fused'))
    except Exception as e:
        t_fused = float('nan')
    coef = torch.randn(n, c, 2, device='cuda')
    t_apply = timeit(lambda: ops.gn_apply(x, coef, silu=True))
    mb = x.numel() * 2 / 1e6
    print(f'{name:6s} {mb:7.1f} MB  split {t_split:8.1f} us   fused {t_fused:8.1f} us   apply only {t_apply:8.1f} us = {2 * mb / t_apply:6.2f} TB/s', flush=True)
