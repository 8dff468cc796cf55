#!/usr/bin/env python3
"""Per-shape throughput of dts_conv2d on the ADM-64 / classifier layer shapes (tuning harness, GPU box only).
Times each shape with HIP events (GPU kept busy ahead of the measured launch) and prints TFLOP/s.  `--variants` times kernel
variants (tuning knobs, dts_set_tuning) INTERLEAVED in one process, round by round, on random data, and checks that every variant
produces bit-identical outputs:   python tools/conv_bench.py --stats --variants conv_variant=0 conv_variant=1"""
import argparse
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusion_tts_amd import ops, _lib

SHAPES = [  # (name, res, cin, cout, k)
    ('L0 3x3 192->192', 64, 192, 192, 3), ('L0 3x3 384->192', 64, 384, 192, 3), ('L0 1x1 384->192', 64, 384, 192, 1),
    ('L1 3x3 384->384', 32, 384, 384, 3), ('L1 3x3 192->384', 32, 192, 384, 3), ('L1 3x3 768->384', 32, 768, 384, 3),
    ('L1 1x1 384->1152', 32, 384, 1152, 1), ('L1 1x1 384->384', 32, 384, 384, 1),
    ('L2 3x3 576->576', 16, 576, 576, 3), ('L2 3x3 1152->576', 16, 1152, 576, 3), ('L2 1x1 576->1728', 16, 576, 1728, 1),
    ('L3 3x3 768->768', 8, 768, 768, 3), ('L3 3x3 1536->768', 8, 1536, 768, 3), ('L3 1x1 768->2304', 8, 768, 2304, 1),
]

SHAPES_128 = [  # widths that are no multiple of 192: the ImageNet classifier (n = batch) and the SD VAE decoder (use --n 4)
    ('CLS 64x64 128->128', 64, 128, 128, 3), ('CLS 32x32 128->256', 32, 128, 256, 3), ('CLS 32x32 256->256', 32, 256, 256, 3),
    ('CLS 8x8 512->512', 8, 512, 512, 3),
    ('VAE 64x64 512->512', 64, 512, 512, 3), ('VAE 128x128 512->512', 128, 512, 512, 3), ('VAE 256x256 512->256', 256, 512, 256, 3),
    ('VAE 256x256 256->256', 256, 256, 256, 3), ('VAE 512x512 256->128', 512, 256, 128, 3), ('VAE 512x512 128->128', 512, 128, 128, 3),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--set', default='adm', choices=['adm', '128'], help="layer shapes: the ADM U-Net's, or the 128-multiple widths")
    ap.add_argument('--n', type=int, default=64)
    ap.add_argument('--dtype', default='bf16')
    ap.add_argument('--iters', type=int, default=10)
    ap.add_argument('--stats', action='store_true', help='also emit the fused GroupNorm strip statistics (as the network does)')
    ap.add_argument('--no-res', action='store_true', help='no residual input')
    ap.add_argument('--up', action='store_true', help='fused nearest-2x upsample of the input (the listed resolution is the INPUT; implies --no-res)')
    ap.add_argument('--variants', nargs='*', default=None, help='e.g. conv_variant=0 conv_variant=1 (interleaved A/B)')
    a = ap.parse_args()
    if a.set == '128':
        SHAPES[:] = SHAPES_128
    if a.variants:
        return ab(a)
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


def ab(a):
    x3 = a.dtype == 'f16x3'          # split precision: f32 tensors, the conv reads a split image and packed (hi | lo) weights
    dt = {'bf16': torch.bfloat16, 'f16': torch.float16, 'f32': torch.float32, 'f16x3': torch.float32}[a.dtype]
    variants = [dict((kv.split('=')[0], int(kv.split('=')[1])) for kv in v.split(',')) for v in a.variants]
    tot = [[0.0, 0.0] for _ in variants]
    for name, r, cin, cout, k in SHAPES:
        x = torch.randn(a.n, r, r, cin, device='cuda').to(dt)
        w = (torch.randn(cout, k, k, cin, device='cuda') / (cin * k * k) ** 0.5).to(dt)
        if x3:
            w = ops.pack_conv_weight(torch.randn(cout, cin, k, k, device='cuda') / (cin * k * k) ** 0.5, ops.F16X3)
            x = ops.SplitAct(ops.split3_f16(x), cin)
        b = torch.randn(cout, device='cuda')
        ro = 2 * r if a.up else r
        res = None if (a.no_res or a.up) else torch.randn(a.n, r, r, cout, device='cuda').to(dt)
        outs = [torch.empty(a.n, ro, ro, cout, device='cuda', dtype=dt) for _ in variants]
        ts = [[] for _ in variants]
        stats = [None] * len(variants)
        for rnd in range(a.iters + 1):
            for vi, v in enumerate(variants):
                for kk, val in v.items():
                    _lib.set_tuning(kk, val)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda._sleep(200_000)
                e0.record()
                o = ops.conv2d(x, w, b, residual=res, out=outs[vi], gn_stats=a.stats, up=a.up)
                e1.record()
                torch.cuda.synchronize()
                stats[vi] = o._gn_stats
                if rnd:
                    ts[vi].append(e0.elapsed_time(e1))
                for kk in v:
                    _lib.set_tuning(kk, -1)
        fl = 2.0 * a.n * ro * ro * cout * cin * k * k
        same = all(torch.equal(outs[0], o) for o in outs[1:])
        same_st = all((stats[0] is None) == (s_ is None) and (s_ is None or torch.allclose(stats[0], s_, rtol=1e-5, atol=1e-3)) for s_ in stats[1:])
        row = f'{name:22s} P={a.n * ro * ro:7d} K={cin * k * k:6d}'
        for vi in range(len(variants)):
            ms = sorted(ts[vi])[len(ts[vi]) // 2]
            tot[vi][0] += fl; tot[vi][1] += ms
            row += f' | {a.variants[vi]}: {ms * 1e3:7.1f} us {fl / ms / 1e9:7.1f} TF/s'
        print(row + f' | outputs {"identical" if same else "DIFFER"}, stats {"ok" if same_st else "DIFFER"}', flush=True)
    print('unweighted totals: ' + ' | '.join(f'{a.variants[vi]}: {tot[vi][0] / tot[vi][1] / 1e9:.1f} TF/s' for vi in range(len(variants))))


if __name__ == '__main__':
    main()
