#!/usr/bin/env python3
"""Runs one conv shape a few times (for rocprofv3 --pmc passes)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusion_tts_amd import ops
n, r, cin, cout, k = [int(v) for v in sys.argv[1:6]]
dt = torch.bfloat16
x = torch.randn(n, r, r, cin, device='cuda').to(dt)
w = (torch.randn(cout, k, k, cin, device='cuda') / (cin * k * k) ** 0.5).to(dt)
b = torch.randn(cout, device='cuda')
out = torch.empty(n, r, r, cout, device='cuda', dtype=dt)
for _ in range(5):
    ops.conv2d(x, w, b, out=out)
torch.cuda.synchronize()
