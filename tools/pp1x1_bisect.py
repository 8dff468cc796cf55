import os, sys, math
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from diffusion_tts_amd import ops, _lib
torch.manual_seed(0)
dt = torch.bfloat16
for (n, res, cin, cout, ks, use_res, use_stats) in [(1, 32, 192, 384, 1, True, True), (1, 32, 192, 384, 1, False, True), (1, 32, 192, 384, 1, True, False),
                                                   (1, 32, 192, 384, 1, False, False), (1, 32, 384, 384, 1, True, True), (1, 32, 128, 384, 1, False, False),
                                                   (1, 32, 64, 384, 1, False, False), (4, 32, 192, 384, 1, False, False), (1, 16, 192, 384, 1, False, False),
                                                   (1, 32, 192, 192, 1, False, False), (1, 32, 192, 384, 3, True, True)]:
    x = torch.randn(n, res, res, cin, device='cuda').to(dt)
    w = (torch.randn(cout, ks, ks, cin, device='cuda') / math.sqrt(cin * ks * ks)).to(dt)
    b = torch.randn(cout, device='cuda')
    r = torch.randn(n, res, res, cout, device='cuda').to(dt) if use_res else None
    ref = ops.conv2d(x.float(), w.float(), b, residual=None if r is None else r.float())
    _lib.set_tuning('conv_variant', 1)
    outs = [ops.conv2d(x, w, b, residual=r, gn_stats=use_stats).float() for _ in range(3)]
    _lib.set_tuning('conv_variant', -1)
    errs = [float((o - ref).abs().max()) for o in outs]
    bad = (outs[0] - ref).abs() > 0.1
    where = bad.nonzero()
    print((n, res, cin, cout, ks, use_res, use_stats), 'max err per run', [f'{e:.3g}' for e in errs], 'bad', int(bad.sum()),
          'first bad idx', where[:3].tolist() if len(where) else None, 'last', where[-2:].tolist() if len(where) else None, flush=True)
