#!/usr/bin/env python3
"""Reads the per-block phase stamps a DIAGNOSTIC build of the conv kernel (tools/ab/make_diag.py, loaded through DTS_LIB_PATH;
never the shipped library) leaves in the split-K workspace and prints where a block's life goes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from diffusion_tts_amd import ops

NAMES = ['prologue', 'K loop', 'epilogue until LDS tile written', 'barrier', 'LDS read + store issue', 'store drain (vmcnt 0)']


def run(name, r, cin, cout, k, n=64, bias=True, residual=False):
    x = torch.randn(n, r, r, cin, device='cuda').to(torch.bfloat16)
    w = (torch.randn(cout, k, k, cin, device='cuda') / (cin * k * k) ** 0.5).to(torch.bfloat16)
    b = torch.randn(cout, device='cuda') if bias else None
    res = torch.randn(n, r, r, cout, device='cuda').to(torch.bfloat16) if residual else None
    for _ in range(3):
        ops.conv2d(x, w, b, residual=res)
    torch.cuda.synchronize()
    ws = ops._conv_workspace(x.device)
    ws.zero_()
    torch.cuda._sleep(2_000_000)
    ops.conv2d(x, w, b, residual=res)
    torch.cuda.synchronize()
    st = ws.view(torch.int64).cpu()
    tile = 192 if cout % 192 == 0 else 128
    nblk = (cout // tile) * ((n * r * r + 127) // 128)
    d = st[:nblk * 12].view(nblk, 12).double()
    rt0 = (d[:, 7] - d[:, 7].min()) / 100.0      # us
    rt1 = (d[:, 8] - d[:, 7].min()) / 100.0
    print(f'{name}: {nblk} blocks; kernel span {rt1.max():.1f} us; block life {((rt1 - rt0).mean()):.1f} us')
    for i in range(6):
        seg = d[:, i + 1] - d[:, i]
        print(f'   {NAMES[i]:34s} mean {seg.mean():9.0f} cyc   p10 {seg.quantile(0.1):9.0f}   p90 {seg.quantile(0.9):9.0f}')
    print(f'   epilogue detail: entry->bias issued {float((d[:, 9] - d[:, 2]).mean()):6.0f}   loads returned {float((d[:, 10] - d[:, 9]).mean()):6.0f}   compute + LDS writes done {float((d[:, 3] - d[:, 10]).mean()):6.0f}')
    clk = ((d[:, 6] - d[:, 0]) / ((rt1 - rt0) * 1e-6)).median() / 1e9
    print(f'   in-kernel clock ~{clk:.2f} GHz')
    h = torch.histc(rt0.float(), bins=20, min=0, max=float(rt1.max()))
    print('   block starts per 5% of span:', [int(v) for v in h])
    order = torch.argsort(rt0)
    for g0 in range(0, nblk, 512):
        idx = order[g0:g0 + 512]
        print(f'   generation {g0 // 512}: prologue {float((d[idx, 1] - d[idx, 0]).mean()):7.0f}  K loop {float((d[idx, 2] - d[idx, 1]).mean()):8.0f}  epilogue {float((d[idx, 5] - d[idx, 2]).mean()):7.0f} cyc')


if __name__ == '__main__':
    run('L0 3x3 192->192', 64, 192, 192, 3)
    run('L0 3x3 192->192 +residual', 64, 192, 192, 3, residual=True)
    run('L1 3x3 384->384 +residual', 32, 384, 384, 3, residual=True)
    run('L1 1x1 384->1152', 32, 384, 1152, 1)
