#!/usr/bin/env python3
"""One attention_x3_kernel shape launched a few times (for rocprofv3 --pmc passes: tools/att_pmc.sh): ADM 32x32, T = 1024, 6 heads, 64 rows."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusion_tts_amd import ops

n, t, heads, d = int(sys.argv[1]) if len(sys.argv) > 1 else 64, 1024, 6, 64
c = heads * d
sp = ops.SplitQKV(torch.empty((n, t, 6 * c), dtype=torch.float16, device='cuda'), (n, t, 1, 3 * c))
ops._call('dts_split2_f16', ops._ptr(torch.randn(n, t, 3 * c, device='cuda'), 'qkv', torch.float32), 3 * c, ops._ptr(sp.data), n * t)
for _ in range(6):
    o = ops.attention(sp.view(n, t, 3 * c), heads, d ** -0.5, x3=True, split_out=True)
torch.cuda.synchronize()
print('ok', float(o.data.float().abs().mean()))
