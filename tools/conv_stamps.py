#!/usr/bin/env python3
"""Where does a conv launch spend its time?  Builds a SECOND library with -DDTS_STAMPS (in-kernel s_memtime stamps around the prologue,
the K loop and the epilogue of conv_igemm_kernel / conv_pp_kernel; the product library has none), runs the ADM layer shapes at a given
batch and prints, per shape: launch time (HIP events), and per block the median cycles of prologue / K loop / epilogue, the in-kernel
clock, and the spread of block start / end times (how long the launch ramp and the tail are).
    python tools/conv_stamps.py --n 8 [--variant 0|1]"""
import argparse, ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIB = os.path.join(ROOT, 'diffusion_tts_amd', 'libdts_hip_stamps.so')


def build():
    csrc = os.path.join(ROOT, 'diffusion_tts_amd', 'csrc')
    objs = []
    for src in ['conv_igemm.hip', 'conv_small.hip', 'groupnorm.hip', 'attention.hip', 'elementwise.hip']:
        o = os.path.join('/tmp', 'stamps_' + src.replace('.hip', '.o'))
        if src == 'conv_igemm.hip' or not os.path.exists(o) or os.path.getmtime(os.path.join(csrc, src)) > os.path.getmtime(o):
            extra = ['-mllvm', '-amdgpu-mfma-vgpr-form=1'] if src == 'attention.hip' else []
            subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-DDTS_STAMPS', '-DDTS_DIAG_KERNELS', *extra, '-c',
                                   os.path.join(csrc, src), '-o', o], stderr=subprocess.DEVNULL)
        objs.append(o)
    subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB, *objs])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--n', type=int, default=8)
    ap.add_argument('--variant', type=int, default=-1)
    ap.add_argument('--stages', type=int, default=-1)
    ap.add_argument('--splits', type=int, default=-1)
    ap.add_argument('--build-only', action='store_true')
    ap.add_argument('--dtype', default='bf16', choices=['bf16', 'f16', 'f16x3'])
    ap.add_argument('--res', action='store_true', help='with a residual input')
    ap.add_argument('--only', default='', help='substring filter on the shape names')
    a = ap.parse_args()
    if not os.path.exists(LIB) or a.build_only:
        build()
    if a.build_only:
        return
    os.environ['DTS_LIB_PATH'] = LIB
    import torch
    import numpy as np
    from diffusion_tts_amd import ops, _lib
    from tools.conv_bench import SHAPES
    lib = _lib.load()
    lib.dts_debug_read_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
    lib.dts_debug_read_segments.argtypes = [ctypes.c_void_p, ctypes.c_int]
    for k_, v_ in (('conv_variant', a.variant), ('conv_stages', a.stages), ('conv_splits', a.splits)):
        if v_ >= 0:
            _lib.set_tuning(k_, v_)
    x3 = a.dtype == 'f16x3'
    dt = {'bf16': torch.bfloat16, 'f16': torch.float16, 'f16x3': torch.float32}[a.dtype]
    for name, r, cin, cout, k in SHAPES:
        if a.only not in name:
            continue
        x = torch.randn(a.n, r, r, cin, device='cuda').to(dt)
        if x3:      # the conv reads a split image (as inside the network, where the GroupNorm apply wrote it): no split pass in the timing
            w = ops.pack_conv_weight((torch.randn(cout, cin, k, k, device='cuda') / (cin * k * k) ** 0.5), ops.F16X3)
            x = ops.SplitAct(ops.split3_f16(x), cin)
        else:
            w = (torch.randn(cout, k, k, cin, device='cuda') / (cin * k * k) ** 0.5).to(dt)
        b = torch.randn(cout, device='cuda')
        out = torch.empty(a.n, r, r, cout, device='cuda', dtype=dt)
        res_ = torch.randn(a.n, r, r, cout, device='cuda').to(dt) if a.res else None
        _conv = ops.conv2d
        ops_conv2d = lambda *aa, **kk: _conv(*aa, residual=res_, **kk)
        for _ in range(3):
            ops_conv2d(x, w, b, out=out, gn_stats=True)
        torch.cuda.synchronize()
        assert lib.dts_debug_clear_stamps() == 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda._sleep(200_000)
        e0.record()
        ops_conv2d(x, w, b, out=out, gn_stats=True)
        e1.record()
        torch.cuda.synchronize()
        buf = np.zeros(8192 * 8, dtype=np.uint64)
        assert lib.dts_debug_read_stamps(buf.ctypes.data, buf.size) == 0
        st = buf.reshape(8192, 8)
        kern = ops.conv_kernel(x, w)
        live = st[:, 3] != 0                                        # the stamps were cleared before the launch
        s = st[live].astype(np.int64)
        nb = int(live.sum())
        pro, loop, epi = s[:, 1] - s[:, 0], s[:, 2] - s[:, 1], s[:, 3] - s[:, 2]
        rt = (s[:, 5] - s[:, 4])                                    # 100 MHz ticks
        clk = np.median((s[:, 3] - s[:, 0]) / np.maximum(rt, 1)) * 100e6 / 1e9
        first, last_start, last_end = s[:, 4].min(), s[:, 4].max(), s[:, 5].max()
        print(f'{name:20s} n={a.n:3d} kernel={kern} blocks={nb:4d} launch+reduce {e0.elapsed_time(e1)*1e3:6.1f} us | per block (median cycles): prologue {int(np.median(pro)):6d} '
              f'loop {int(np.median(loop)):6d} epilogue {int(np.median(epi)):6d} | clock {clk:4.2f} GHz | kernel span {(last_end-first)/100:5.1f} us, '
              f'last block starts +{(last_start-first)/100:4.1f} us, median block {np.median(rt)/100:5.1f} us', flush=True)
        if (s[:, 6] != 0).all():
            e1_, e2_, e3_ = s[:, 6] - s[:, 2], s[:, 7] - s[:, 6], s[:, 3] - s[:, 7]
            print(f'    epilogue (fast path) median cycles: accumulators -> staged tile + statistics {int(np.median(e1_))}, barrier {int(np.median(e2_))}, copy-out {int(np.median(e3_))}', flush=True)
        if kern in (4, 6):
            sb = np.zeros(4096 * 2 * 8, dtype=np.uint64)
            assert lib.dts_debug_read_segments(sb.ctypes.data, sb.size) == 0
            sg = sb.reshape(4096, 2, 8).astype(np.int64)
            ok = sg[:, 0, 3] != 0
            tiles = k * k * ((2 * cin if x3 else cin) // 64) / max(1, round(int(ok.sum()) / ((a.n * r * r // 256) * (cout // (32 * kern)))))
            names = ['DMA issue', 'frag reads', 'LOAD barrier', 'MFMA', 'vmcnt', 'COMPUTE barrier']
            for g_ in (0, 1):
                med = np.median(sg[ok][:, g_, :6], axis=0) / tiles
                print(f'    group {g_} cycles per K tile: ' + '  '.join(f'{nm} {v:6.0f}' for nm, v in zip(names, med)) + f'   sum {med.sum():6.0f}', flush=True)


main()
