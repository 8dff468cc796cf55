#!/usr/bin/env python3
"""Reads the per-block phase stamps a DIAGNOSTIC build of the conv kernel leaves in the split-K workspace
(tools/ab/libdts_diag.so via DTS_LIB_PATH; never the shipped library) and prints where a block's life goes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from diffusion_tts_amd import ops

def run(name, r, cin, cout, k, n=64):
    x = torch.randn(n, r, r, cin, device='cuda').to(torch.bfloat16)
    w = (torch.randn(cout, k, k, cin, device='cuda') / (cin * k * k) ** 0.5).to(torch.bfloat16)
    b = torch.randn(cout, device='cuda')
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
    t = d[:, :5] - d[:, :1]
    rt0 = (d[:, 5] - d[:, 5].min()) / 100.0      # us
    rt1 = (d[:, 6] - d[:, 5].min()) / 100.0
    print(f'{name}: {nblk} blocks; kernel span {rt1.max():.1f} us; block life {((rt1 - rt0).mean()):.1f} us')
    names = ['prologue', 'K loop', 'epilogue issue', 'store drain (vmcnt 0)']
    for i in range(4):
        seg = t[:, i + 1] - t[:, i]
        print(f'   {names[i]:24s} mean {seg.mean():9.0f} cyc   p10 {seg.quantile(0.1):9.0f}   p90 {seg.quantile(0.9):9.0f}')
    clk = (t[:, 4] / ((rt1 - rt0) * 1e-6)).median() / 1e9
    print(f'   in-kernel clock ~{clk:.2f} GHz')
    # generations: histogram of start times
    h = torch.histc(rt0.float(), bins=20, min=0, max=float(rt1.max()))
    print('   block starts per 5% of span:', [int(v) for v in h])
    h = torch.histc(rt1.float(), bins=20, min=0, max=float(rt1.max()))
    print('   block ends   per 5% of span:', [int(v) for v in h])
    la = st[:nblk * 8].view(nblk, 8)[:, 7]
    print('   LDS_ALLOC values (first 512 blocks):', sorted(set(int(v) & 0xff for v in la[:512])), ' base!=0 count', int(((la[:512] & 0xff) != 0).sum()))

run('L0 3x3 192->192', 64, 192, 192, 3)
run('L1 1x1 384->1152', 32, 384, 1152, 1)
