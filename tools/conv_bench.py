#!/usr/bin/env python3
"""Per-shape throughput of dts_conv2d on the ADM-64 / classifier layer shapes (tuning harness, GPU box only).
Times each shape with HIP events (GPU kept busy ahead of the measured launch) and prints TFLOP/s."""
import argparse
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusion_tts_amd import ops

SHAPES = [  # (name, res, cin, cout, k)
    ('L0 3x3 192->192', 64, 192, 192, 3), ('L0 3x3 384->192', 64, 384, 192, 3), ('L0 1x1 384->192', 64, 384, 192, 1),
    ('L1 3x3 384->384', 32, 384, 384, 3), ('L1 3x3 192->384', 32, 192, 384, 3), ('L1 3x3 768->384', 32, 768, 384, 3),
    ('L1 1x1 384->1152', 32, 384, 1152, 1), ('L1 1x1 384->384', 32, 384, 384, 1),
    ('L2 3x3 576->576', 16, 576, 576, 3), ('L2 3x3 1152->576', 16, 1152, 576, 3), ('L2 1x1 576->1728', 16, 576, 1728, 1),
    ('L3 3x3 768->768', 8, 768, 768, 3), ('L3 3x3 1536->768', 8, 1536, 768, 3), ('L3 1x1 768->2304', 8, 768, 2304, 1),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--n', type=int, default=64)
    ap.add_argument('--dtype', default='bf16')
    ap.add_argument('--iters', type=int, default=10)
    ap.add_argument('--stats', action='store_true', help='also emit the fused GroupNorm strip statistics (as the network does)')
    a = ap.parse_args()
    dt = {'bf16': torch.bfloat16, 'f16': torch.float16, 'f32': torch.float32}[a.dtype]
    dev = 'cuda'
    tot_f = tot_t = 0.0
    for name, r, cin, cout, k in SHAPES:
        x = torch.randn(a.n, r, r, cin, device=dev).to(dt)
        w = (torch.randn(cout, k, k, cin, device=dev) / (cin * k * k) ** 0.5).to(dt)
        b = torch.randn(cout, device=dev)
        res = torch.randn(a.n, r, r, cout, device=dev).to(dt)
        out = torch.empty(a.n, r, r, cout, device=dev, dtype=dt)
        for _ in range(2):
            ops.conv2d(x, w, b, residual=res, out=out, gn_stats=a.stats)
        ts = []
        for _ in range(a.iters):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda._sleep(200_000)
            e0.record()
            ops.conv2d(x, w, b, residual=res, out=out, gn_stats=a.stats)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        ts.sort()
        ms = ts[len(ts) // 2]
        fl = 2.0 * a.n * r * r * cout * cin * k * k
        tot_f += fl
        tot_t += ms
        print(f'{name:22s} P={a.n*r*r:7d} K={cin*k*k:6d}  {ms*1e3:8.1f} us  {fl/ms/1e9:8.1f} TFLOP/s', flush=True)
    print(f'unweighted total: {tot_f/tot_t/1e9:.1f} TFLOP/s')


if __name__ == '__main__':
    main()
