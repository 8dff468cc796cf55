#!/usr/bin/env python3
"""The SD VAE's mid-block attention (one head of dim 512, T = 4096) at N rows: LDS-DMA staging (default) vs the register-staged form
(DTS_ATT_DB=2), interleaved.    python tools/att512_bench.py --n 16"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusion_tts_amd import ops, _lib

ap = argparse.ArgumentParser()
ap.add_argument('--n', type=int, default=16)
ap.add_argument('--t', type=int, default=4096)
a = ap.parse_args()
qkv = torch.randn(a.n, a.t, 3 * 512, device='cuda').half()
outs = {}
ts = {'dma': [], 'regs': []}
for rnd in range(8):
    for name, knob in (('dma', -1), ('regs', 2)):
        _lib.set_tuning('att_db', knob)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda._sleep(200_000)
        e0.record()
        outs[name] = ops.attention(qkv, 1, 512 ** -0.5)
        e1.record()
        torch.cuda.synchronize()
        if rnd:
            ts[name].append(e0.elapsed_time(e1))
        _lib.set_tuning('att_db', -1)
fl = a.n * a.t * a.t * 512 * 4
for k, v in ts.items():
    ms = sorted(v)[len(v) // 2]
    print(f'{k:5s}: {ms * 1e3:8.1f} us   {fl / ms / 1e9:7.1f} TFLOP/s (algorithmic: Q.K^T once)')
print('outputs identical:', torch.equal(outs['dma'], outs['regs']))
