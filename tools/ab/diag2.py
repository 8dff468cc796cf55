#!/usr/bin/env python3
"""Epilogue sub-phases from a DIAGNOSTIC conv build (see diag.py)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from diffusion_tts_amd import ops

def run(name, r, cin, cout, k, n=64, bias=True):
    x = torch.randn(n, r, r, cin, device='cuda').to(torch.bfloat16)
    w = (torch.randn(cout, k, k, cin, device='cuda') / (cin * k * k) ** 0.5).to(torch.bfloat16)
    b = torch.randn(cout, device='cuda') if bias else None
    for _ in range(3):
        ops.conv2d(x, w, b)
    torch.cuda.synchronize()
    ws = ops._conv_workspace(x.device)
    ws.zero_()
    torch.cuda._sleep(2_000_000)
    ops.conv2d(x, w, b)
    torch.cuda.synchronize()
    st = ws.view(torch.int64).cpu()
    nblk = (cout // 192) * ((n * r * r + 127) // 128)
    d = st[:nblk * 8].view(nblk, 8).double()
    # order: tsA start, tsB after prologue, tsC after K loop, est0 values packed + LDS written, est1 after barrier, tsD stores issued, tsE drained
    seq = torch.stack([d[:, 0], d[:, 1], d[:, 2], d[:, 5], d[:, 6], d[:, 3], d[:, 4]], dim=1)
    names = ['prologue', 'K loop', 'epi: bias/pack/LDS write', 'epi: barrier', 'epi: LDS read + store issue', 'store drain']
    print(name)
    for i in range(6):
        seg = seq[:, i + 1] - seq[:, i]
        print(f'   {names[i]:30s} mean {seg.mean():9.0f} cyc   p10 {seg.quantile(0.1):9.0f}   p90 {seg.quantile(0.9):9.0f}')

run('L0 3x3 192->192', 64, 192, 192, 3)
run('L1 1x1 384->1152', 32, 384, 1152, 1)
run('L0 3x3 192->192 no bias', 64, 192, 192, 3, bias=False)
