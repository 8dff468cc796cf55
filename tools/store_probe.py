#!/usr/bin/env python3
"""A/B of the 16-byte paired split-image stores (round 6) against the former two 8-byte stores per lane, interleaved in one process, outputs
compared bit for bit:  (a) dts_gn_apply_x3 (GroupNorm apply -> split image; DTS_GN_FUSE=2 selects the 8-byte form) on the ADM level
shapes, with and without the raw image;  (b) the qkv 1x1 projection with the attention's split image as output (DTS_CONV_EPI32=2)."""
import argparse
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusion_tts_amd import ops, _lib


def ab(fn, knob, old_value, iters):
    ts = [[], []]
    outs = [None, None]
    for rnd in range(iters + 1):
        for vi, val in enumerate((-1, old_value)):
            _lib.set_tuning(knob, val)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda._sleep(200_000)
            e0.record()
            o = fn()
            e1.record()
            torch.cuda.synchronize()
            outs[vi] = o
            if rnd:
                ts[vi].append(e0.elapsed_time(e1))
    _lib.set_tuning(knob, -1)
    med = [sorted(t)[len(t) // 2] * 1e3 for t in ts]
    return med, outs


def datas(o):
    o = o if isinstance(o, tuple) else (o,)
    return [t.data for t in o]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--n', type=int, default=64)
    ap.add_argument('--iters', type=int, default=12)
    a = ap.parse_args()
    tot = [0.0, 0.0]
    for name, r, c, cat, raw in [('L0 64x64x192', 64, 192, 0, False), ('L0 cat 192+192 +raw', 64, 192, 192, True), ('L1 32x32x384', 32, 384, 0, False),
                                 ('L1 cat 384+384 +raw', 32, 384, 384, True), ('L2 16x16x576', 16, 576, 0, False), ('L2 cat 576+576 +raw', 16, 576, 576, True),
                                 ('L3 8x8x768', 8, 768, 0, False)]:
        x1 = torch.randn(a.n, r, r, c, device='cuda')
        x2 = torch.randn(a.n, r, r, cat, device='cuda') if cat else None
        coef = torch.randn(a.n, c + cat, 2, device='cuda')
        med, outs = ab(lambda: ops.gn_apply(x1, coef, x2=x2, silu=True, split_out=True, raw_split=raw), 'gn_fuse', 2, a.iters)
        same = all(torch.equal(p, q) for p, q in zip(datas(outs[0]), datas(outs[1])))
        mb = (a.n * r * r * (c + cat) * 4) * (3 if raw else 2) / 1e6
        tot[0] += med[0]; tot[1] += med[1]
        print(f'gn_apply_x3 {name:22s} {mb:7.1f} MB | 16-byte stores {med[0]:7.1f} us {mb / med[0]:5.2f} TB/s | 8-byte {med[1]:7.1f} us {mb / med[1]:5.2f} TB/s | '
              f'{"identical" if same else "DIFFER"}', flush=True)
    print(f'gn_apply_x3 sum: {tot[0]:.1f} vs {tot[1]:.1f} us')
    for name, r, cin, cout in [('qkv 32x32 384->1152', 32, 384, 1152), ('qkv 16x16 576->1728', 16, 576, 1728), ('qkv 8x8 768->2304', 8, 768, 2304)]:
        x = ops.SplitAct(ops.split3_f16(torch.randn(a.n, r, r, cin, device='cuda')), cin)
        w = ops.pack_conv_weight(torch.randn(cout, cin, 1, 1, device='cuda') / cin ** 0.5, ops.F16X3)
        b = torch.randn(cout, device='cuda')
        med, outs = ab(lambda: ops.conv2d(x, w, b, out_split2=True), 'conv_epi32', 2, a.iters)
        same = torch.equal(outs[0].data, outs[1].data)
        print(f'{name:24s} | 16-byte stores {med[0]:7.1f} us | 8-byte {med[1]:7.1f} us | {"identical" if same else "DIFFER"}', flush=True)


if __name__ == '__main__':
    main()
