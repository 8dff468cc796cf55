#!/usr/bin/env python3
"""FEASIBILITY PROBE (round 6): two HIP streams with COMPLEMENTARY CU masks (hipExtStreamCreateWithCUMask) -- can a matrix-bound kernel
(the 3x3 ping-pong conv) on one half of the chip run beside an HBM-bound kernel (the GroupNorm apply) on the other half, so that the HBM
phases of one half-batch chain hide under the matrix phases of the other?  Measures, per partition scheme (by XCD: user-mask bit k is CU
k // 8 of XCD k % 8 on a multi-XCC device; or half of every XCD's CUs):
   conv alone on the whole chip / on half the chip; GroupNorm apply alone on the whole / half chip; both at once on complementary halves."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusion_tts_amd import ops

hip = C.CDLL('libamdhip64.so')
hip.hipExtStreamCreateWithCUMask.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(C.c_uint32)]
hip.hipExtStreamCreateWithCUMask.restype = C.c_int


def masked_stream(bits):
    words = (C.c_uint32 * 8)(*[sum(1 << b for b in range(32) if bits[32 * w + b]) for w in range(8)])
    st = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(st), 8, words)
    assert rc == 0, f'hipExtStreamCreateWithCUMask -> {rc}'
    return torch.cuda.ExternalStream(st.value)


def timeit(work, iters=8):
    """work: list of (stream, fn); all launched back to back, wall time until every stream has drained"""
    for st, fn in work:
        with torch.cuda.stream(st):
            fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        evs = []
        torch.cuda.synchronize()
        t0 = torch.cuda.Event(enable_timing=True)
        t0.record()
        for st, fn in work:
            st.wait_event(t0)
            with torch.cuda.stream(st):
                for _ in range(4):
                    fn()
                e = torch.cuda.Event(enable_timing=True)
                e.record(st)
                evs.append(e)
        torch.cuda.synchronize()
        ts.append([t0.elapsed_time(e) / 4 * 1e3 for e in evs])
    ts.sort(key=lambda v: max(v))
    return ts[len(ts) // 2]


def main():
    n = 32                                                          # a half batch
    x = ops.SplitAct(ops.split3_f16(torch.randn(n, 64, 64, 192, device='cuda')), 192)
    w = ops.pack_conv_weight(torch.randn(192, 192, 3, 3, device='cuda') / (192 * 9) ** 0.5, ops.F16X3)
    b = torch.randn(192, device='cuda')
    res = torch.randn(n, 64, 64, 192, device='cuda')
    g1 = torch.randn(n, 64, 64, 192, device='cuda')
    coef = torch.randn(n, 192, 2, device='cuda')
    conv = lambda: ops.conv2d(x, w, b, residual=res, gn_stats=True)
    gn = lambda: ops.gn_apply(g1, coef, silu=True, split_out=True)
    full = torch.cuda.current_stream()
    print(f'whole chip, one at a time (us per launch, {n} rows at 64x64x192): conv {timeit([(full, conv)])[0]:.1f}   gn_apply {timeit([(full, gn)])[0]:.1f}')
    both_full = [(torch.cuda.Stream(), conv), (torch.cuda.Stream(), gn)]
    print('whole chip, two unmasked streams at once: conv %.1f  gn_apply %.1f' % tuple(timeit(both_full)))
    for name, bits_a in (('by XCD (XCDs 0-3 | 4-7)', [(k % 8) < 4 for k in range(256)]), ('half of every XCD', [(k // 8) < 16 for k in range(256)])):
        sa, sb = masked_stream(bits_a), masked_stream([not v for v in bits_a])
        ca = timeit([(sa, conv)])[0]
        ga = timeit([(sb, gn)])[0]
        tb = timeit([(sa, conv), (sb, gn)])
        tcc = timeit([(sa, conv), (sb, conv)])
        tgg = timeit([(sa, gn), (sb, gn)])
        print(f'{name}: half chip alone: conv {ca:.1f}  gn_apply {ga:.1f} | conv on A beside gn_apply on B: {tb[0]:.1f} / {tb[1]:.1f} | conv beside conv: {tcc[0]:.1f} / {tcc[1]:.1f} | '
              f'gn beside gn: {tgg[0]:.1f} / {tgg[1]:.1f}', flush=True)


if __name__ == '__main__':
    main()
