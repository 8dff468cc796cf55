#!/usr/bin/env python3
"""Timing of the folded skip convolution's second K loop (conv_pp_kernel<.., SK>) on the ADM-64 shapes: the 3x3 launch with the skip operand
against the same launch without it, in alternating rounds (clock drift hits both alike).  (Round 6 also ran it per issue-position variant of the
loop -- profiles/r06_experiments.txt item 5; the variants are gone from the kernel.)"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusion_tts_amd import ops, _lib

SHAPES = [(64, 64, 192, 384, 192, False), (64, 32, 384, 768, 384, False), (64, 16, 576, 1152, 576, False), (64, 32, 384, 192, 384, True)]


def timeit(fn, iters=20):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


def main():
    variants = [1]
    for n, hw, c, cs, cout, up in SHAPES:
        hs = hw // 2 if up else hw
        h = ops.SplitAct(ops.split3_f16(torch.randn(n, hw, hw, c, device='cuda')), c)
        src = ops.SplitAct(ops.split3_f16(torch.randn(n, hs, hs, cs, device='cuda')), cs)
        w1 = ops.pack_conv_weight(torch.randn(cout, c, 3, 3, device='cuda') / (c * 9) ** 0.5, ops.F16X3)
        ws = ops.pack_conv_weight(torch.randn(cout, cs, 1, 1, device='cuda') / cs ** 0.5, ops.F16X3)
        b = torch.randn(cout, device='cuda')
        out = torch.empty(n, hw, hw, cout, device='cuda')
        gf = 2.0 * n * hw * hw * cout * cs / 1e9
        ref, acc = None, {}
        for rnd in range(5):                      # rounds of (plain, every variant): clock drift hits all alike
            _lib.set_tuning('conv_skip_fold', -1)
            acc.setdefault('plain', []).append(timeit(lambda: ops.conv2d(h, w1, b, out=out, gn_stats=True), 10))
            for v in variants:
                _lib.set_tuning('conv_skip_fold', 1)
                acc.setdefault(v, []).append(timeit(lambda: ops.conv2d(h, w1, b, out=out, gn_stats=True, skip=(src, ws, up)), 10))
                if ref is None:
                    ref = out.clone()
                assert torch.equal(out, ref), v
        med = {k: sorted(v)[len(v) // 2] for k, v in acc.items()}
        line = f'{(n, hw, c, cs, cout, up)}: plain {med["plain"]:7.1f} us |'
        for v in variants:
            d = med[v] - med['plain']
            line += f' skip loop {d:6.1f} us ({gf / d * 1e3:4.0f} TFLOP/s algorithmic, operand read at {n * hs * hs * cs * 4 / d / 1e6:4.2f} TB/s)'
        _lib.set_tuning('conv_skip_fold', -1)
        print(line, flush=True)


if __name__ == '__main__':
    main()
