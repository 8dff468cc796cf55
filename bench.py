#!/usr/bin/env python3
"""bench.py -- candidate U-Net steps/sec of the noise-trajectory-search hot path on MI355X.

Workload (BASELINE.json configs[2], the configuration the metric is quoted on): EDM ImageNet-64 (ADM, 295.9 M
parameters, random init + the documented weight rule), epsilon-greedy search, N = 64 candidates, ImageNet scorer
(65.4 M-parameter classifier, random init), sigma step i = 5 of the 18-step schedule with S_churn = 40.
One "step" = ONE search iteration over the candidate batch: build the 64 candidate noises (K14), one Heun step =
two denoiser forwards over the 64 candidates (K1-K9), quantise (K10), score the 64 predicted images (K12), gather
rewards, pick the survivor and rebuild the pivot.  That is 2*64 = 128 candidate U-Net evaluations ("candidate
U-Net steps") per GPU per step.  Synthetic inputs (N(0,1) latents and noises) are resident in HBM before the timed
region; the host-RNG + upload inclusive rate is reported separately in DESIGN.md.

Multi-GPU (`--gpus N`, launched by torch.distributed.run): candidates are sharded across ranks, 64 per GPU (weak
scaling, default) or 64 in total (`--scaling strong`); one RCCL all-gather of the rewards per step.

Prints ONE JSON line on rank 0 (see the field list in the task contract), including `roofline` for the dominant
kernel (the implicit-GEMM conv: algorithmic conv FLOPs / summed launch durations measured with HIP events on the
launch stream) and `cpu_baseline` (the CPU oracle timed on this box's host cores on a bounded sample).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist

PEAK_TFLOPS = {'bf16': 2500.0, 'f16': 2500.0, 'f32': 157.3}      # dense MFMA peaks, MI355X_MICROARCH.md
ADM_GFLOP_PER_EVAL = 219.33
CLS_GFLOP_PER_IMG = 38.16


_T0 = time.perf_counter()


def log(msg):
    if int(os.environ.get('RANK', '0')) == 0:
        print(f'[bench +{time.perf_counter() - _T0:7.1f}s] {msg}', file=sys.stderr, flush=True)


class HipEvents:
    """hipEvent pairs ATTACHED TO A KERNEL'S OWN DISPATCH (hipExtLaunchKernelGGL start/stop events, passed down through
    dts_conv_args.ev_start/ev_stop): the timestamps are those of the kernel's packet, which is what `rocprofv3 --kernel-trace`
    reports, with no barrier packet or fence between launches.  Events recorded around a launch from the host (even without the
    system fence, even behind a GPU spin) read 30-50 % long on some boxes of the pool and right on others."""

    def __init__(self):
        import ctypes
        self.c = ctypes
        self.hip = ctypes.CDLL('libamdhip64.so')
        self.hip.hipEventCreate.argtypes = [ctypes.POINTER(ctypes.c_void_p)]
        self.hip.hipEventElapsedTime.argtypes = [ctypes.POINTER(ctypes.c_float), ctypes.c_void_p, ctypes.c_void_p]
        self.hip.hipEventDestroy.argtypes = [ctypes.c_void_p]
        self.live = []

    def pair(self):
        evs = []
        for _ in range(2):
            ev = self.c.c_void_p()
            assert self.hip.hipEventCreate(self.c.byref(ev)) == 0
            self.live.append(ev)
            evs.append(ev)
        return evs[0], evs[1]

    def elapsed_ms(self, a, b):
        ms = self.c.c_float()
        rc = self.hip.hipEventElapsedTime(self.c.byref(ms), a, b)
        assert rc == 0, rc
        return ms.value

    def close(self):
        for ev in self.live:
            self.hip.hipEventDestroy(ev)
        self.live = []


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=6)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--dtype', default='bf16', choices=['bf16', 'f16', 'f32'])
    ap.add_argument('--candidates', type=int, default=64, help='candidates per GPU (weak) or in total (strong)')
    ap.add_argument('--scaling', default='weak', choices=['weak', 'strong'])
    ap.add_argument('--scorer', default='imagenet', choices=['imagenet', 'brightness'])
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-kernel-timing', action='store_true')
    ap.add_argument('--conv-table', action='store_true', help='log the per-shape conv launch table of the instrumented steps')
    ap.add_argument('--cpu-sample', type=int, default=32, help='candidates in the CPU-baseline sample')
    return ap.parse_args()


def cpu_baseline(sample_n, seed=0):
    """The oracle (CPU restatement of the reference, fp32 torch-CPU ops) on this box's host cores: one epsilon-greedy
    iteration over `sample_n` candidates = 2*sample_n denoiser rows + sample_n classifier images."""
    from diffusion_tts_amd import init as dinit
    from diffusion_tts_amd.config import adm_imagenet64, ClassifierConfig
    from oracle.edm_nets import NetCfg, EDMPrecondOracle
    from oracle.classifier import ClsCfg
    from oracle import sampler as osamp, scorers as oscore
    # the GPU box gives one GPU's share of the host (16 cores); asking torch for every core the machine has
    # oversubscribes the cgroup quota and runs ~100x slower
    cores = min(16, len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1))
    torch.set_num_threads(cores)
    cfg = adm_imagenet64()
    sd, _ = dinit.refill_degenerate(dinit.edm_state_dict(cfg, seed), seed)
    net = EDMPrecondOracle(NetCfg('adm', 64, 3, 1000, 192, [1, 2, 3, 4], 4, 3, [32, 16, 8]), sd)
    csd, _ = dinit.refill_degenerate(dinit.classifier_state_dict(ClassifierConfig(), 1), 1)
    scorer = oscore.ImageNetOracle(ClsCfg(), csd)
    t_steps = osamp.sigma_schedule(net, 18)
    ctx = osamp._Ctx(net, 18, 40, 0.05, 50, 1.003)
    g = torch.Generator().manual_seed(7)
    x = torch.randn(1, 3, 64, 64, generator=g, dtype=torch.float64) * t_steps[5]
    eps = torch.randn(sample_n, 3, 64, 64, generator=g, dtype=torch.float64)
    lab = torch.eye(1000)[torch.tensor([7])].repeat(sample_n, 1)
    t0 = time.perf_counter()
    _, x0 = ctx.heun_step(x.repeat(sample_n, 1, 1, 1), t_steps[5], t_steps[6], 5, eps, lab)
    sc = scorer(osamp.to_uint8(x0), lab, torch.zeros(sample_n))
    int(sc.argmax())
    dt = time.perf_counter() - t0
    return {'value': round(2 * sample_n / dt, 3), 'unit': 'candidate U-Net steps/sec', 'cores': torch.get_num_threads(),
            'kind': 'port', 'seconds': round(dt, 2),
            'sample': f'1 eps-greedy iteration over {sample_n} candidates of the same workload: {2 * sample_n} ADM-64 '
                      f'denoiser rows + {sample_n} classifier images, fp32 torch-CPU oracle'}


def main():
    a = parse()
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    ndev = max(1, torch.cuda.device_count())
    dev = torch.device('cuda', (local % ndev) if world > 1 else 0)
    if world > 1:
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        torch.cuda.set_device(dev)
        # RCCL over xGMI ("nccl" IS RCCL on ROCm).  DTS_DIST_BACKEND=gloo exists only to rehearse the multi-rank control
        # flow on a one-GPU box (ranks share the card; collectives staged through the host).
        backend = os.environ.get('DTS_DIST_BACKEND', 'nccl')
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev)
        else:
            dist.init_process_group(backend)
    if a.gpus != world and rank == 0 and world > 1:
        print(f'warning: --gpus {a.gpus} but WORLD_SIZE {world}', file=sys.stderr)
    torch.cuda.set_device(dev)

    from diffusion_tts_amd import init as dinit, ops
    from diffusion_tts_amd.config import adm_imagenet64
    from diffusion_tts_amd.networks import EDMPrecond
    from diffusion_tts_amd.parallel import CandidateShards
    from diffusion_tts_amd.sampler import _Loop
    from diffusion_tts_amd.scorers import ImageNetScorer, BrightnessScorer

    dtype = {'bf16': torch.bfloat16, 'f16': torch.float16, 'f32': torch.float32}[a.dtype]
    cfg = adm_imagenet64()
    sd, _ = dinit.refill_degenerate(dinit.edm_state_dict(cfg, 0), 0)
    log('weights initialised on host')
    net = EDMPrecond(cfg, sd, device=dev, dtype=dtype)
    del sd
    log('denoiser packed on device')
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        scorer = ImageNetScorer(device=dev, compute_dtype=dtype, seed=1) if a.scorer == 'imagenet' else BrightnessScorer()
    shards = CandidateShards()
    n_total = a.candidates * world if a.scaling == 'weak' else a.candidates
    lo, hi = shards.span(n_total)
    nl = hi - lo

    # ---- synthetic state, resident in HBM before timing
    g = torch.Generator().manual_seed(1234)                       # same stream on every rank
    L = _Loop(net, dev, 18, 40, 0.05, 50, 1.003, None, shards)
    step_indices = torch.arange(18, dtype=torch.float64)
    t_steps = (80 ** (1 / 7) + step_indices / 17 * (0.002 ** (1 / 7) - 80 ** (1 / 7))) ** 7
    t_steps = torch.cat([t_steps, torch.zeros(1, dtype=torch.float64)])
    i_step = 5
    x_cur = (torch.randn(1, 3, 64, 64, generator=g, dtype=torch.float64) * t_steps[i_step]).to(dev)
    pivot = torch.randn(1, 3, 64, 64, generator=g, dtype=torch.float64).to(dev)
    labels = torch.eye(1000)[torch.tensor([7])].to(dev)
    lab_l = labels.repeat(nl, 1).contiguous()
    total_steps = a.warmup + a.steps
    noise, modes, scales = [], [], []
    lam = 0.15 * np.sqrt(3 * 64 * 64)
    for s in range(min(total_steps, 4)):                          # 4 distinct noise sets, cycled
        gfull = torch.randn(n_total, 3, 64, 64, generator=g, dtype=torch.float64)
        noise.append(gfull[lo:hi].to(dev).contiguous())
        m = (torch.rand(n_total, generator=g) < 0.6).to(torch.int32)
        sc = (torch.rand(n_total, generator=g) * lam).float()
        modes.append((m, m[lo:hi].to(dev).contiguous()))
        scales.append((sc, sc[lo:hi].to(dev).contiguous()))
    state = {'pivot': pivot}

    def one_step(s):
        q = s % len(noise)
        cand = ops.candidate_noise(state['pivot'], noise[q], modes[q][1], scales[q][1])
        _, x0 = L.step(x_cur, t_steps[i_step], t_steps[i_step + 1], i_step, cand, lab_l, nb=nl)
        loc = L.score(scorer, x0, lab_l).to(dev, torch.float32)
        scores = shards.gather_rewards(loc, n_total, 1).cpu()
        best = int(scores.argmax())
        if lo <= best < hi:                                       # survivor rebuilt locally; replicated via host noise in the real loop
            j = best - lo
            state['pivot'] = ops.candidate_noise(state['pivot'], noise[q][j:j + 1].contiguous(), modes[q][1][j:j + 1].contiguous(),
                                                 scales[q][1][j:j + 1].contiguous())
        return best

    def barrier():
        if world > 1:
            dist.barrier(device_ids=[dev.index]) if dist.get_backend() == 'nccl' else dist.barrier()
        torch.cuda.synchronize(dev)

    # setup, not warm-up: the forwards are captured as HIP graphs on the third call of a shape (graphs.py); two untimed iterations
    # here keep that one-off capture (~0.5 s) out of the W warm-up steps and the K timed steps whatever W is
    for s in range(2):
        one_step(s)
    torch.cuda.synchronize(dev)
    log(f'state resident; {nl} candidates on this rank; forwards captured ({net._graphs.captures} graphs); warmup')
    for s in range(a.warmup):
        one_step(s)
        torch.cuda.synchronize(dev)
        log(f'warmup step {s} done')
    evals0 = net.evals
    barrier()
    t0 = time.perf_counter()
    for s in range(a.steps):
        one_step(a.warmup + s)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev if dist.get_backend() == 'nccl' else 'cpu')
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax)
    log(f'timed region: {dt:.3f}s for {a.steps} steps')
    rows_local = net.evals - evals0
    rows_total = 2 * n_total * a.steps
    value = rows_total / dt

    # ---- per-kernel timing of the dominant kernel (implicit-GEMM conv) with HIP events on the launch stream
    roof = None
    if not a.no_kernel_timing and rank != 0:
        for s in range(min(2, a.steps)):           # the steps contain the reward all-gather: every rank must take part
            one_step(a.warmup + s)
        torch.cuda.synchronize(dev)
    if not a.no_kernel_timing and rank == 0:
        rec = []
        orig = ops.conv2d
        hev = HipEvents()

        def timed_conv(x1, w, bias=None, **kw):
            e0, e1 = hev.pair()
            out = orig(x1, w, bias, timing_events=(e0, e1), **kw)
            n_, ho, wo, co = out.shape
            rec.append((2.0 * n_ * ho * wo * co * w.shape[1] * w.shape[2] * w.shape[3], e0, e1,
                        (tuple(x1.shape), tuple(w.shape), 'x2' if kw.get('x2') is not None else '', 'up' if kw.get('up') else '',
                         'res' if kw.get('residual') is not None else '', 'bnc' if kw.get('bias_nc') is not None else '',
                         'stats' if kw.get('gn_stats') else '')))
            return out
        ops.conv2d = timed_conv
        caches = [net._graphs] + ([scorer.model._graphs] if hasattr(scorer, 'model') else [])
        was = [c.enabled for c in caches]
        for c in caches:                      # the timed region replays HIP graphs; per-launch events need the eager sequence
            c.enabled = False                 # (the same kernels with the same arguments, launched one by one)
        try:
            for s in range(min(2, a.steps)):
                one_step(a.warmup + s)
            torch.cuda.synchronize(dev)
        finally:
            ops.conv2d = orig
            for c, w_ in zip(caches, was):
                c.enabled = w_
        fl = sum(r[0] for r in rec)
        ms = sum(hev.elapsed_ms(r[1], r[2]) for r in rec)
        if a.conv_table:                      # per-shape view of the conv launches inside the network (stderr)
            agg = {}
            for r in rec:
                t = agg.setdefault(r[3], [0, 0.0, 0.0])
                t[0] += 1; t[1] += r[0]; t[2] += hev.elapsed_ms(r[1], r[2])
            for k_, (c_, f_, m_) in sorted(agg.items(), key=lambda kv: -kv[1][2]):
                log(f'conv {str(k_):100s} x{c_:4d}  {m_ / min(2, a.steps):7.3f} ms/step  {f_ / m_ / 1e9:7.1f} TFLOP/s')
        hev.close()
        ach = fl / (ms * 1e-3) / 1e12
        peak = PEAK_TFLOPS[a.dtype]
        traffic, tsrc = None, None
        try:        # HBM bytes per launch of the dominant kernel from the committed PMC passes (cannot be collected in-process)
            with open(os.path.join(ROOT, 'profiles', 'r01_hbm_traffic_pmc.json')) as f:
                ks = json.load(f)['kernels']
                k = next(v for name, v in ks.items() if name.startswith('conv_igemm_kernel<bf16_t, 6, 4, 2, 2, true>'))   # dominant variant
            if a.dtype == 'bf16' and a.candidates == 64:
                traffic, tsrc = round(k['hbm_bytes_per_launch']), 'profiles/r01_hbm_traffic_pmc.json (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, separate passes)'
        except Exception:
            pass
        roof = {'bound': 'mfma', 'achieved': round(ach, 1), 'peak': peak, 'unit': 'TFLOP/s', 'frac': round(ach / peak, 4),
                'traffic': traffic, 'traffic_source': tsrc, 'kernel': 'conv_igemm_kernel', 'launches': len(rec),
                'avg_launch_us': round(ms * 1e3 / len(rec), 2), 'avg_launch_gflop': round(fl / len(rec) / 1e9, 3),
                'conv_ms_per_step': round(ms / min(2, a.steps), 2),
                'whole_step_frac': round((value / world) * ADM_GFLOP_PER_EVAL * 1e9 / (peak * 1e12), 4)}
    barrier()

    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        log('kernel timing done; CPU baseline (oracle) running')
        cpu = cpu_baseline(a.cpu_sample)
        log('CPU baseline done')

    if rank == 0:
        out = {
            'metric': 'candidate U-Net steps/sec, EDM ImageNet-64 eps-greedy N=64', 'value': round(value, 2),
            'unit': 'candidate U-Net steps/sec', 'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup,
            'ms_per_step': round(dt / a.steps * 1e3, 3), 'higher_is_better': True, 'scaling': a.scaling,
            'vs_baseline': None, 'dtype': a.dtype, 'data': 'synthetic (N(0,1) latents/noises, random-init weights + weight rule)',
            'config': {'workload': 'EDM ImageNet-64 (ADM 295.9M) eps-greedy search iteration, imagenet scorer, sigma step 5/18',
                       'candidates_total': n_total, 'candidates_per_gpu': nl, 'rows_per_step_total': 2 * n_total,
                       'scorer': a.scorer, 'parallelism': f'candidates sharded x{world}, 1 all-gather of rewards per step'},
            'roofline': roof, 'cpu_baseline': cpu,
            'scorer_images_per_sec': round(n_total * a.steps / dt, 2), 'rows_local': rows_local,
        }
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
