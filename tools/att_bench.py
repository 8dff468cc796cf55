#!/usr/bin/env python3
"""Per-shape throughput of dts_attention on the ADM-64 / DDPM++ / classifier attention shapes (tuning harness, GPU box only).

Variants (tuning knobs, include/dts.h dts_set_tuning) are timed INTERLEAVED in one process, round by round, on random data
(cdna_hip_programming.md section 5.4 rules 24/25); prints the median per variant and the algorithmic bytes per launch.

    python tools/att_bench.py --n 64 --variants att_xcd=0 att_xcd=1
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusion_tts_amd import ops, _lib

SHAPES = [  # (name, tokens, heads, head_dim)
    ('ADM L1 32x32 h6', 1024, 6, 64), ('ADM L2 16x16 h9', 256, 9, 64), ('ADM L3 8x8 h12', 64, 12, 64),
    ('CLS 32x32 h4', 1024, 4, 64), ('CLS 16x16 h6', 256, 6, 64), ('CLS 8x8 h8', 64, 8, 64), ('DDPM++ 16x16 d256', 256, 1, 256),
]


def x3_table(a):
    print('f32 qkv, per launch: attention32_kernel (f32 MFMA) | split-precision path = dts_split2_f16 + attention_x3_kernel | the f16 kernel for scale')
    for name, t, heads, d in SHAPES:
        if d != 64:
            continue
        c = heads * d
        qkv = torch.randn(a.n, t, 3 * c, device='cuda')
        q16 = qkv.half()
        fl = 4.0 * a.n * heads * t * t * d
        res = []
        for fn in (lambda: ops.attention(qkv, heads, d ** -0.5), lambda: ops.attention(qkv, heads, d ** -0.5, x3=True), lambda: ops.attention(q16, heads, d ** -0.5)):
            for _ in range(2):
                fn()
            ts = []
            for r in range(a.rounds):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda._sleep(200_000)
                e0.record()
                for _ in range(4):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) / 4)
            res.append(sorted(ts)[len(ts) // 2])
        print(f'{name:20s} T={t:5d} | f32 {res[0] * 1e3:7.1f} us {fl / res[0] / 1e9:6.1f} TF/s | f16x3 {res[1] * 1e3:7.1f} us {fl / res[1] / 1e9:6.1f} TF/s (algorithmic) | f16 {res[2] * 1e3:7.1f} us', flush=True)


def x3_kernel_table(a):
    """attention_x3_kernel alone, as the network calls it inside an attention block: the qkv projection's split image in, the proj
    convolution's split image out; DTS_ATT_QT = 1 | 2 (query tiles per wave) interleaved against the launcher's rule, outputs compared."""
    print('attention_x3_kernel on a split qkv image (split image out), us per launch: launcher rule | att_qt=1 | att_qt=2' + ''.join(f' | variant {d_}' for d_ in a.variants_x3))
    for name, t, heads, d in SHAPES:
        if d != 64 or t < 128:
            continue
        c = heads * d
        sp = ops.SplitQKV(torch.empty(0), (a.n, t, 1, 3 * c))
        sp.data = torch.empty((a.n, t, 6 * c), dtype=torch.float16, device='cuda')
        ops._call('dts_split2_f16', ops._ptr(torch.randn(a.n, t, 3 * c, device='cuda'), 'qkv', torch.float32), 3 * c, ops._ptr(sp.data), a.n * t)
        fl = 4.0 * a.n * heads * t * t * d
        outs, res = [], []
        variants = [(-1, -1), (1, -1), (2, -1)] + [(-1, 16 + d_) for d_ in a.variants_x3]
        ts = [[] for _ in variants]
        for r in range(a.rounds + 1):
            for vi, (q, db) in enumerate(variants):
                _lib.set_tuning('att_qt', q)
                _lib.set_tuning('att_db', db)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda._sleep(200_000)
                e0.record()
                for _ in range(4):
                    o = ops.attention(sp.view(a.n, t, 3 * c), heads, d ** -0.5, x3=True, split_out=True)
                e1.record()
                torch.cuda.synchronize()
                if r:
                    ts[vi].append(e0.elapsed_time(e1) / 4)
                else:
                    outs.append(o.data.clone())
        _lib.set_tuning('att_qt', -1)
        _lib.set_tuning('att_db', -1)
        med = [sorted(t_)[len(t_) // 2] for t_ in ts]
        same = all(torch.equal(outs[0], o_) for o_ in outs[1:])
        print(f'{name:20s} T={t:5d} | ' + ' | '.join(f'{m * 1e3:7.1f} us {fl / m / 1e9:6.1f} TF/s' for m in med) + f' | outputs {"identical" if same else "DIFFER"}', flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--n', type=int, default=64)
    ap.add_argument('--dtype', default='bf16')
    ap.add_argument('--rounds', type=int, default=9)
    ap.add_argument('--variants', nargs='*', default=['att_xcd=0', 'att_xcd=1'])
    ap.add_argument('--variants-x3', type=int, nargs='*', default=[], help='(--x3-kernel) A/B: 0 = the kernel without the three latency changes of round 6 (K fragment prefetch, one-round max exchange, V prefetch)')
    ap.add_argument('--x3-kernel', action='store_true', help='attention_x3_kernel alone on split images, att_qt variants')
    ap.add_argument('--x3', action='store_true', help='f32 qkv: the f32-MFMA kernel vs the split-precision kernel (split pass included)')
    a = ap.parse_args()
    dt = {'bf16': torch.bfloat16, 'f16': torch.float16, 'f32': torch.float32}[a.dtype]
    if a.x3_kernel:
        return x3_kernel_table(a)
    if a.x3:
        return x3_table(a)
    variants = [dict((kv.split('=')[0], int(kv.split('=')[1])) for kv in v.split(',')) for v in a.variants]
    print('variants:', variants)
    for name, t, heads, d in SHAPES:
        c = heads * d
        qkv = torch.randn(a.n, t, 3 * c, device='cuda').to(dt)
        fl = 4.0 * a.n * heads * t * t * d
        alg = a.n * t * 4 * c * qkv.element_size()                     # read q,k,v once, write out once
        ts = [[] for _ in variants]
        for _ in range(2):
            ops.attention(qkv, heads, d ** -0.5)
        for r in range(a.rounds):
            for vi, v in enumerate(variants):
                for k, val in v.items():
                    _lib.set_tuning(k, val)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda._sleep(200_000)
                e0.record()
                for _ in range(4):
                    ops.attention(qkv, heads, d ** -0.5)
                e1.record()
                torch.cuda.synchronize()
                ts[vi].append(e0.elapsed_time(e1) / 4)
                for k in v:
                    _lib.set_tuning(k, -1)
        row = f'{name:20s} T={t:5d}  alg {alg / 1e6:7.1f} MB '
        for vi, v in enumerate(variants):
            ms = sorted(ts[vi])[len(ts[vi]) // 2]
            row += f' | {a.variants[vi]}: {ms * 1e3:7.1f} us {fl / ms / 1e9:7.1f} TF/s {alg / ms / 1e9:6.2f} TB/s'
        print(row, flush=True)


if __name__ == '__main__':
    main()
