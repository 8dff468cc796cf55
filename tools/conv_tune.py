#!/usr/bin/env python3
"""Which launch form would be fastest for every convolution the networks actually issue at a given batch?  Records the distinct
dts_conv2d calls of one ADM-64 denoiser forward and one classifier forward at `--n` rows (shapes, two-source concat, fused upsample,
residual, statistics), then times each with the dispatcher's own choice and with every forced form -- ping-pong kernel (where eligible)
and implicit-GEMM kernel, K split 1..8 -- interleaved round by round, and prints per call: launches per forward, the dispatcher's time,
the best forced form and what switching would save per forward.  A tuning aid for the cost models in conv_igemm.hip (GPU box only).
    python tools/conv_tune.py --n 8"""
import argparse
import os
import sys
from collections import OrderedDict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from diffusion_tts_amd import ops, _lib


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--n', type=int, default=8)
    ap.add_argument('--iters', type=int, default=7)
    ap.add_argument('--min-us', type=float, default=0.0, help='only print calls whose possible saving per forward is at least this')
    ap.add_argument('--net', default='adm64', choices=['adm64', 'ddpmpp32'], help='adm64 (+ the classifier) or the DDPM++ CIFAR-32 denoiser of BASELINE configs[0..1]')
    a = ap.parse_args()
    sys.argv = [sys.argv[0]]
    job = bench.Job(bench.parse())
    calls = OrderedDict()
    real = ops.conv2d

    def rec(x1, w, bias=None, *, x2=None, bias_nc=None, residual=None, up=False, out_scale=1.0, out=None, gn_stats=False, **kw):
        key = (tuple(x1.shape), tuple(w.shape), None if x2 is None else tuple(x2.shape), bool(up), residual is not None, bool(gn_stats),
               bias_nc is not None, str(x1.dtype), 'gn_coef' in kw and kw['gn_coef'] is not None)
        calls[key] = calls.get(key, 0) + 1
        return real(x1, w, bias, x2=x2, bias_nc=bias_nc, residual=residual, up=up, out_scale=out_scale, out=out, gn_stats=gn_stats, **kw)

    if a.net == 'adm64':
        net, scorer, _ = bench.build_adm(job, torch.bfloat16, scorer_name='imagenet')
        mods, res_, nlab = [net, scorer.model], 64, 1000
    else:
        from diffusion_tts_amd import init as dinit
        from diffusion_tts_amd.config import ddpmpp_cifar10
        from diffusion_tts_amd.networks import EDMPrecond
        cfg = ddpmpp_cifar10()
        sd, _ = dinit.refill_degenerate(dinit.edm_state_dict(cfg, 0), 0)
        net, scorer = EDMPrecond(cfg, sd, device=job.dev, dtype=torch.bfloat16), None
        mods, res_, nlab = [net], 32, 10
    ops.conv2d = rec
    was = [m._graphs.enabled for m in mods]
    for m in mods:
        m._graphs.enabled = False
    x = torch.randn(a.n, 3, res_, res_, dtype=torch.float64, device=job.dev)
    lab = torch.eye(nlab, device=job.dev)[torch.arange(a.n) % nlab]
    d = net(x, torch.tensor([1.5], dtype=torch.float64), lab)
    if scorer is not None:
        scorer(ops.quantize_u8(d.to(torch.float64)), lab, torch.zeros(a.n, device=job.dev))
    ops.conv2d = real
    for m, w_ in zip(mods, was):
        m._graphs.enabled = w_
    print(f'{len(calls)} distinct conv calls, {sum(calls.values())} launches in one {a.net} denoiser forward' + (' + one classifier forward' if scorer is not None else '') + f' at {a.n} rows', flush=True)

    forms = [('auto', {})]
    for s in (1, 2, 3, 4, 5, 6, 8):
        forms.append((f'pp/{s}', {'conv_variant': 1, 'conv_splits': s}))
    for s in (1, 2, 3, 4, 5, 6, 8):
        forms.append((f'ig/{s}', {'conv_variant': 0, 'conv_splits': s}))
    total_auto = total_best = 0.0
    rows = []
    for key, cnt in calls.items():
        xs, ws_, x2s, up, has_res, st, bnc, dts, gnc = key
        if gnc:                      # GroupNorm-fusing calls (not the default path) are not swept
            continue
        dt = {'torch.bfloat16': torch.bfloat16, 'torch.float16': torch.float16, 'torch.float32': torch.float32}[dts]
        n, h, w_, c1 = xs
        cout, kh, kw_, cin = ws_
        x1 = torch.randn(*xs, device=job.dev).to(dt)
        x2 = None if x2s is None else torch.randn(*x2s, device=job.dev).to(dt)
        wt = (torch.randn(*ws_, device=job.dev) / (cin * kh * kw_) ** 0.5).to(dt)
        b = torch.randn(cout, device=job.dev)
        ho, wo = (2 * h, 2 * w_) if up else (h, w_)
        res = torch.randn(n, ho, wo, cout, device=job.dev).to(dt) if has_res else None
        nc = torch.randn(n, cout, device=job.dev).to(dt) if bnc else None
        out = torch.empty(n, ho, wo, cout, device=job.dev, dtype=dt)
        ts = {name: [] for name, _ in forms}
        kern = {}
        for rnd in range(a.iters + 1):
            for name, knobs in forms:
                for kk, val in knobs.items():
                    _lib.set_tuning(kk, val)
                try:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    torch.cuda._sleep(100_000)
                    e0.record()
                    ops.conv2d(x1, wt, b, x2=x2, bias_nc=nc, residual=res, up=up, out=out, gn_stats=st)
                    e1.record()
                    torch.cuda.synchronize()
                    if rnd:
                        ts[name].append(e0.elapsed_time(e1) * 1e3)
                    kern[name] = ops.conv_kernel(x1, wt, x2=x2, up=up) if hasattr(ops, 'conv_kernel') else -1
                except Exception:
                    ts[name] = None
                finally:
                    for kk in knobs:
                        _lib.set_tuning(kk, -1)
                if ts[name] is None:
                    ts[name] = []
        med = {k_: (sorted(v)[len(v) // 2] if v else float('inf')) for k_, v in ts.items()}
        # a forced pp on an ineligible shape falls back to the implicit-GEMM kernel: keep only forms that ran the kernel they name
        cand = {k_: v for k_, v in med.items() if k_ != 'auto' and not (k_.startswith('pp/') and kern.get(k_, 0) not in (4, 6))}
        best = min(cand, key=cand.get)
        save = (med['auto'] - cand[best]) * cnt
        total_auto += med['auto'] * cnt
        total_best += min(med['auto'], cand[best]) * cnt
        rows.append((save, f'{str(xs):22s} w{str(ws_):20s} x2={"y" if x2s else "-"} up={int(up)} res={int(has_res)} st={int(st)} {dts[6:]:8s} x{cnt:3d} | auto {med["auto"]:6.1f} us (kernel {kern.get("auto")}) | '
                           f'best {best:5s} {cand[best]:6.1f} us | saves {save:7.1f} us per forward | ' + ' '.join(f'{k_}={v:.0f}' for k_, v in med.items() if k_ != 'auto' and v < 1e9)))
    for save, line in sorted(rows, key=lambda r: -r[0]):
        if save >= a.min_us:
            print(line, flush=True)
    print(f'sum over the recorded launches: dispatcher {total_auto / 1e3:.3f} ms, best forced form per call {total_best / 1e3:.3f} ms')


main()
