#!/usr/bin/env python3
"""bench.py -- candidate U-Net steps/sec of the noise-trajectory-search hot path on MI355X.

Default workload `adm64_eps_greedy` (BASELINE.json configs[2], the configuration the metric is quoted on): EDM ImageNet-64
(ADM, 295.9 M parameters, random init + the documented weight rule), epsilon-greedy search, N = 64 candidates IN TOTAL,
ImageNet scorer (65.4 M-parameter classifier, random init), sigma step i = 5 of the 18-step schedule with S_churn = 40.
One "step" = ONE search iteration over the candidate batch: build the candidate noises (K14), one Heun step = two denoiser
forwards over the candidates (K1-K9), quantise (K10), score the predicted images (K12), gather rewards, pick the survivor and
rebuild the pivot on every rank.  That is 2*64 = 128 candidate U-Net evaluations ("candidate U-Net steps") per step.
Synthetic inputs (N(0,1) latents and noises) are resident in HBM before the timed region.

Multi-GPU: `python bench.py --gpus N` STARTS N ranks itself (python -m torch.distributed.run, one process per GPU, RCCL) before
anything touches the GPU; launched under torch.distributed.run it joins the existing job.  Default `--scaling strong`: the 64
candidates are sharded over the ranks (8 per GPU at 8 GPUs = BASELINE config 3), one RCCL all-gather of the rewards per step;
the weak-scaling rate (64 candidates per GPU) is measured in the same run and reported as `weak_value`.

Other workloads (their own metric strings; not the headline): `ddpmpp32_rejection` (configs[1]: DDPM++ CIFAR-32, rejection N=16,
brightness) and `adm64_mcts` (configs[4]: ADM-64 MCTS, S rollouts per timestep, imagenet scorer; one step = one whole search).

Prints ONE JSON line on rank 0 with, besides the contract fields: `roofline` (dominant kernel = the implicit-GEMM conv:
algorithmic conv FLOPs / summed launch durations from HIP events attached to the kernel's own dispatch), `cpu_baseline` (the CPU
oracle on this box's host cores, one iteration over the same 64 candidates), `parity` (GPU f32 / f16 / bf16 rewards and argmax
against that oracle iteration on the SAME inputs, plus index agreement of the 16-bit modes over 8 more iterations) and
`e2e_evals_per_s` (config 3 end to end through generate_image_grid, host RNG and uploads included).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_TFLOPS = {'bf16': 2500.0, 'f16': 2500.0, 'f32': 157.3,       # dense MFMA peaks, MI355X_MICROARCH.md
               'f16x3': 2500.0}   # split precision runs on the f16 matrix instruction: ALGORITHMIC FLOPs (one per MAC pair, not its three passes) over that peak
DTYPE_NAMES = ['bf16', 'f16', 'f32', 'f16x3']


def torch_dtype(name):
    import torch
    from diffusion_tts_amd import ops
    return {'bf16': torch.bfloat16, 'f16': torch.float16, 'f32': torch.float32, 'f16x3': ops.F16X3}[name]
GFLOP_PER_EVAL = {'adm64': 219.33, 'ddpmpp32': 42.38}             # BASELINE.md section 2 / SURVEY.md 8(d)
CLS_GFLOP_PER_IMG = 38.16
# written by tools/final_run.sh at HEAD: tools/pmc_traffic.py from the rocprofv3 --pmc passes, tools/rocprof_dominant.py from the --kernel-trace --stats pass
def csrc_digest():
    """sha256 over the kernel sources the library is built from (csrc/*.hip, dts_common.h, include/dts.h): the static profile files quoted
    in the line carry the digest of the tree they were collected from (tools/collect_profiles.sh), so a kernel edit without re-running the
    profile passes shows up as `stale: true` instead of riding along silently."""
    import glob
    import hashlib
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(ROOT, 'diffusion_tts_amd', 'csrc', '*.hip'))) + [os.path.join(ROOT, 'diffusion_tts_amd', 'csrc', 'dts_common.h'),
                                                                                          os.path.join(ROOT, 'include', 'dts.h')]
    for f_ in files:
        with open(f_, 'rb') as fh:
            h.update(os.path.basename(f_).encode() + b'\0' + fh.read())
    return h.hexdigest()[:16]


def _stale(doc):
    """True / False when the profile records the kernel-source digest it was collected from, None when it predates the field"""
    d = doc.get('csrc_sha256')
    return None if d is None else d != csrc_digest()


TRAFFIC_PROFILE = {'f16x3': 'profiles/r06_hbm_traffic_pmc_f16x3.json', 'bf16': 'profiles/r06_hbm_traffic_pmc_bf16.json'}
ROCPROF_PROFILE = {'f16x3': 'profiles/r06_rocprof_dominant_f16x3.json', 'bf16': 'profiles/r06_rocprof_dominant_bf16.json'}

_T0 = time.perf_counter()


def log(msg):
    if int(os.environ.get('RANK', '0')) == 0:
        print(f'[bench +{time.perf_counter() - _T0:7.1f}s] {msg}', file=sys.stderr, flush=True)


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=None)
    ap.add_argument('--warmup', type=int, default=None)
    ap.add_argument('--workload', default='adm64_eps_greedy', choices=['adm64_eps_greedy', 'ddpmpp32_rejection', 'adm64_mcts'])
    ap.add_argument('--dtype', default='f16x3', choices=DTYPE_NAMES, help='f16x3 (default) = split precision on the 16-bit MFMA: the mode that reproduces the reference\'s fp32 selections; '
                    'f32 = parity mode on the f32 MFMA; bf16 / f16 = throughput modes (another sample after the first near-tied pick)')
    ap.add_argument('--candidates', type=int, default=None, help='candidates in total (strong, default 64) or per GPU (weak)')
    ap.add_argument('--scaling', default='strong', choices=['weak', 'strong'])
    ap.add_argument('--scorer', default='imagenet', choices=['imagenet', 'brightness'])
    ap.add_argument('--S', type=int, default=256, help='adm64_mcts: rollouts per timestep (BASELINE config 5: 256)')
    ap.add_argument('--mcts-children', type=int, default=4, help='adm64_mcts: children per node (SamplingParams.N)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-kernel-timing', action='store_true')
    ap.add_argument('--no-parity', action='store_true')
    ap.add_argument('--no-e2e', action='store_true')
    ap.add_argument('--no-weak', action='store_true', help='skip the weak-scaling leg of a multi-rank run')
    ap.add_argument('--no-subrecords', action='store_true', help='default workload only: skip the sub-records (8 candidates per GPU, 32x32 rejection, MCTS slice, f32 scorer, SD beam, whole-search index agreement)')
    ap.add_argument('--mcts-slice', type=int, default=64, help='S of the MCTS sub-record of the default line')
    ap.add_argument('--conv-sequence', default=None, help='write the per-launch conv shape sequence of one step (JSON) for tools/pmc_traffic.py')
    ap.add_argument('--conv-table', action='store_true', help='log the per-shape conv launch table of the instrumented steps')
    ap.add_argument('--cpu-sample', type=int, default=64, help='candidates in the CPU-baseline / parity iteration')
    a = ap.parse_args(argv)
    d_steps, d_warm = {'adm64_eps_greedy': (6, 2), 'ddpmpp32_rejection': (20, 3), 'adm64_mcts': (1, 0)}[a.workload]
    a.steps = d_steps if a.steps is None else a.steps
    a.warmup = d_warm if a.warmup is None else a.warmup
    if a.candidates is None:
        a.candidates = 16 if a.workload == 'ddpmpp32_rejection' else 64
    return a


def launch_ranks(n):
    """`python bench.py --gpus N` outside a distributed job: start N ranks (one process per GPU) as CHILD processes of this one,
    which has not touched the GPU (no HIP call before this point), relay their output and exit with their status.  Never an
    exec of a process that initialised the GPU."""
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n}', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    log(f'starting {n} ranks: {" ".join(cmd[1:8])} ...')
    return subprocess.call(cmd, env=env)


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


# ----------------------------------------------------------------------------------------------------------------------
def sigma_steps():
    import torch
    idx = torch.arange(18, dtype=torch.float64)
    t = (80 ** (1 / 7) + idx / 17 * (0.002 ** (1 / 7) - 80 ** (1 / 7))) ** 7
    return torch.cat([t, torch.zeros(1, dtype=torch.float64)])


def cpu_model():
    try:
        with open('/proc/cpuinfo') as f:
            for ln in f:
                if ln.lower().startswith('model name'):
                    return ln.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def oracle_iteration(sample_n, seed=0):
    """The oracle (CPU restatement of the reference, fp32 torch-CPU ops) on this box's host cores: ONE epsilon-greedy iteration
    (sigma step 5) over `sample_n` candidates = 2*sample_n ADM-64 denoiser rows + sample_n classifier images.  Returns the timing
    record for `cpu_baseline` and the inputs / rewards the `parity` leg replays on the GPU."""
    import torch
    from diffusion_tts_amd import init as dinit
    from diffusion_tts_amd.config import adm_imagenet64, ClassifierConfig
    from oracle.edm_nets import NetCfg, EDMPrecondOracle
    from oracle.classifier import ClsCfg
    from oracle import sampler as osamp, scorers as oscore
    # the GPU box gives one GPU's share of the host (16 cores); asking torch for every core the machine has oversubscribes the
    # cgroup quota and runs ~100x slower
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
    best = int(sc.argmax())
    dt = time.perf_counter() - t0
    rec = {'value': round(2 * sample_n / dt, 3), 'unit': 'candidate U-Net steps/sec', 'cores': torch.get_num_threads(),
           'kind': 'port', 'seconds': round(dt, 2), 'cpu_model': cpu_model(),
           'sample': f'1 eps-greedy iteration over {sample_n} candidates of the same workload: {2 * sample_n} ADM-64 '
                     f'denoiser rows + {sample_n} classifier images, fp32 torch-CPU oracle'}
    return rec, dict(x=x, eps=eps, lab=lab, rewards=sc.float(), best=best)


def conv_roofline(a, run_once, reps, dtype_name, value_per_gpu, gflop_per_eval, extra_caches=()):
    """Per-launch timing of the dominant kernel over `reps` eager repetitions of `run_once` (HIP events attached to the conv
    kernel's own dispatch)."""
    import torch
    from diffusion_tts_amd import ops
    rec = []
    orig = ops.conv2d
    hev = HipEvents()

    def timed_conv(x1, w, bias=None, **kw):
        e0, e1 = hev.pair()
        out = orig(x1, w, bias, timing_events=(e0, e1), **kw)
        n_, ho, wo, co = out.shape
        es = out.element_size()
        cin = w.shape[3]
        # algorithmic bytes of the launch: every input / weight / residual byte read once, every output byte written once
        # (split precision: the operand image is 4 bytes per logical input element like the f32 tensor it stands for, the packed weight 2 x f16)
        wbytes = w.packed.numel() * 2 if isinstance(w, ops.X3Weight) else w.numel() * es
        alg = es * (x1.numel() + (kw['x2'].numel() if kw.get('x2') is not None else 0) + out.numel() +
                    (out.numel() if kw.get('residual') is not None else 0)) + wbytes
        # which kernel the launcher picks (csrc/conv_igemm.hip conv_pick_pp via dts_conv_kernel): the names a kernel trace shows
        kern = ops.conv_kernel(x1, w, x2=kw.get('x2'), up=bool(kw.get('up')), residual=kw.get('residual'), gn_coef=kw.get('gn_coef'))
        fam = {6: 'conv_pp_kernel', 4: 'conv_pp_kernel/128'}.get(kern) or ('conv_igemm_kernel/3x3' if w.shape[1] == 3 else 'conv_igemm_kernel/1x1')
        kdim = w.shape[1] * w.shape[2] * w.shape[3]
        sk = kw.get('skip')
        if sk is not None:                  # the block's 1x1 skip convolution folded into this launch: its K steps and operands count here
            kdim += sk[0].shape[3]
            alg += es * sk[0].numel() + sk[1].packed.numel() * 2
        rec.append((2.0 * n_ * ho * wo * co * kdim, e0, e1,
                    (tuple(x1.shape), tuple(w.shape), 'x2' if kw.get('x2') is not None else '', 'up' if kw.get('up') else '',
                     'res' if kw.get('residual') is not None else '', 'bnc' if kw.get('bias_nc') is not None else '',
                     'stats' if kw.get('gn_stats') else '') + ((f'skip{sk[0].shape[3]}' + ('up' if sk[2] else ''),) if sk is not None else ()), alg, fam))
        return out
    ops.conv2d = timed_conv
    was = [c.enabled for c in extra_caches]
    for c in extra_caches:                  # the timed region replays HIP graphs; per-launch events need the eager sequence
        c.enabled = False                   # (the same kernels with the same arguments, launched one by one)
    try:
        for s in range(reps):
            run_once(s)
        torch.cuda.synchronize()
    finally:
        ops.conv2d = orig
        for c, w_ in zip(extra_caches, was):
            c.enabled = w_
    fl = sum(r[0] for r in rec)
    ms = sum(hev.elapsed_ms(r[1], r[2]) for r in rec)
    if a.conv_table:                        # per-shape view of the conv launches inside the network (stderr)
        agg = {}
        for r in rec:
            t = agg.setdefault(r[3], [0, 0.0, 0.0])
            t[0] += 1; t[1] += r[0]; t[2] += hev.elapsed_ms(r[1], r[2])
        for k_, (c_, f_, m_) in sorted(agg.items(), key=lambda kv: -kv[1][2]):
            log(f'conv {str(k_):100s} x{c_:4d}  {m_ / reps:7.3f} ms/step  {f_ / m_ / 1e9:7.1f} TFLOP/s')
    if a.conv_sequence:                     # the conv launches of ONE step in launch order: tools/pmc_traffic.py lines the PMC
        per_step = len(rec) // reps         # dispatches of the same command up against it (per-shape HBM bytes vs algorithmic)
        with open(a.conv_sequence, 'w') as f:
            json.dump({'launches_per_step': per_step, 'workload': a.workload, 'dtype': dtype_name, 'candidates': a.candidates,
                       'sequence': [{'shape': str(r[3]), 'kernel': r[5], 'alg_bytes': r[4], 'flop': r[0],
                                     'us': round(hev.elapsed_ms(r[1], r[2]) * 1e3, 2)} for r in rec[:per_step]]}, f)
    ach = fl / (ms * 1e-3) / 1e12
    peak = PEAK_TFLOPS[dtype_name]
    per_kernel = {}
    for r in rec:                           # per conv kernel: launches, time, algorithmic FLOPs and bytes (the PMC bytes are per kernel name)
        k_ = per_kernel.setdefault(r[5], dict(launches=0, ms=0.0, gflop=0.0, alg_mb=0.0))
        k_['launches'] += 1; k_['ms'] += hev.elapsed_ms(r[1], r[2]); k_['gflop'] += r[0] / 1e9; k_['alg_mb'] += r[4] / 1e6
    hev.close()
    kernels = {k_: dict(launches=v['launches'], tflops=round(v['gflop'] / max(v['ms'], 1e-9), 1), avg_launch_us=round(v['ms'] * 1e3 / v['launches'], 2),
                        alg_mb_per_launch=round(v['alg_mb'] / v['launches'], 1)) for k_, v in per_kernel.items()}
    dom_family = max(per_kernel, key=lambda k_: per_kernel[k_]['ms'])
    traffic, tsrc, rp, traffic_stale = None, None, None, None
    headline_shape = a.workload == 'adm64_eps_greedy' and a.candidates == 64 and a.gpus == 1
    try:        # HBM bytes per launch of the dominant kernel: a STATIC figure from committed rocprofv3 --pmc passes of this very
        # command in this dtype (counters cannot be collected from inside the process), valid for the default workload only
        if headline_shape and dtype_name in TRAFFIC_PROFILE:
            with open(os.path.join(ROOT, TRAFFIC_PROFILE[dtype_name])) as f:
                doc = json.load(f)
            traffic = round(doc['kernels_by_family'][dom_family]['hbm_bytes_per_launch'])
            traffic_stale = _stale(doc)
            tsrc = (f'static, from {TRAFFIC_PROFILE[dtype_name]} (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, separate passes of this command, collected at commit '
                    f'{doc.get("collected_at_commit", "unknown")}; counters cannot be read from inside the process, so NOT measured in this run)')
    except Exception:
        pass
    try:        # the same kernel's average duration in the rocprofv3 --kernel-trace --stats run of this command (static, like the counters):
        # a profiled process runs several per cent slower (MI355X_MICROARCH.md, DVFS give-back item 2), so both figures are carried
        if headline_shape and dtype_name in ROCPROF_PROFILE:
            with open(os.path.join(ROOT, ROCPROF_PROFILE[dtype_name])) as f:
                rp = json.load(f)
    except Exception:
        rp = None
    # the DOMINANT kernel = the conv kernel with the most time in the step (conv_pp_kernel on the ADM workload: the 3x3 layers); its
    # numbers are the top-level ones (what `rocprofv3 --kernel-trace --stats` reports for that kernel name must agree with avg_launch_us);
    # the aggregate over every implicit-GEMM conv launch of the step is kept beside it
    dom = max(per_kernel, key=lambda k_: per_kernel[k_]['ms'])
    d = per_kernel[dom]
    dach = d['gflop'] / d['ms']            # GFLOP / ms = TFLOP/s
    x3 = dtype_name == 'f16x3'
    out_extra = {}
    if x3:      # split precision: three 16-bit matrix products per algorithmic multiply-add
        out_extra['matrix_work_frac'] = round(3 * dach / peak, 4)
        out_extra['how'] = ('achieved / frac count ALGORITHMIC FLOPs (one per multiply-add pair); matrix_work_frac counts the three f16 products each costs. '
                            'The events bracket the conv kernel only: where a layer\'s input is not already a split image (skip / concat inputs) a dts_split3_f16 '
                            'pass precedes it and is NOT in avg_launch_us; whole_step_frac includes everything')
    if rp is not None and rp.get('kernel_family') == dom:
        rus = float(rp['avg_launch_us'])
        out_extra['rocprof_avg_launch_us'] = round(rus, 2)
        out_extra['rocprof_frac'] = round(d['gflop'] / d['launches'] / (rus * 1e-3) / peak, 4)      # GFLOP / ms = TFLOP/s
        out_extra['rocprof_stale'] = _stale(rp)
        out_extra['rocprof_source'] = f"static, {ROCPROF_PROFILE[dtype_name]} (rocprofv3 --kernel-trace --stats of this command at commit {rp.get('collected_at_commit', 'unknown')}, {rp.get('calls')} calls)"
    return {'bound': 'mfma', 'achieved': round(dach, 1), 'peak': peak, 'unit': 'TFLOP/s', 'frac': round(dach / peak, 4), **out_extra,
            'traffic': traffic, 'traffic_source': tsrc, 'traffic_stale': traffic_stale, 'kernel': dom, 'launches': d['launches'],
            'avg_launch_us': round(d['ms'] * 1e3 / d['launches'], 2), 'avg_launch_gflop': round(d['gflop'] / d['launches'], 3),
            'alg_mb_per_launch': round(d['alg_mb'] / d['launches'], 1), 'ms_per_step': round(d['ms'] / reps, 2),
            'all_conv': {'achieved': round(ach, 1), 'frac': round(ach / peak, 4), 'launches': len(rec),
                         'avg_launch_us': round(ms * 1e3 / max(1, len(rec)), 2), 'conv_ms_per_step': round(ms / reps, 2), 'kernels': kernels},
            'whole_step_frac': round(value_per_gpu * gflop_per_eval * 1e9 / (peak * 1e12), 4)}


# ----------------------------------------------------------------------------------------------------------------------
class Job:
    """Process-group plumbing shared by the workloads."""

    def __init__(self, a):
        import torch
        import torch.distributed as dist
        self.a, self.torch, self.dist = a, torch, dist
        self.world = int(os.environ.get('WORLD_SIZE', '1'))
        self.rank = int(os.environ.get('RANK', '0'))
        local = int(os.environ.get('LOCAL_RANK', '0'))
        if a.gpus != self.world:
            # under torch.distributed.run the job size is WORLD_SIZE; a different --gpus is a launch mistake, not something to
            # paper over (a silent n_gpus=1 line was what round 1 printed for `--gpus 8`)
            print(f'bench.py: --gpus {a.gpus} but WORLD_SIZE={self.world}; launch with `python bench.py --gpus N` (it starts the '
                  f'ranks itself) or with torch.distributed.run --nproc-per-node N bench.py --gpus N', file=sys.stderr)
            sys.exit(2)
        ndev = max(1, torch.cuda.device_count())
        self.dev = torch.device('cuda', (local % ndev) if self.world > 1 else 0)
        self.backend = None
        # DTS_SHARD_ALWAYS_COLLECT=1 with one rank: a one-rank RCCL group whose collectives are all issued (parallel.py) -- the way to
        # run the RCCL code path of the sharded iteration on a one-GPU box
        self.dist_on = self.world > 1 or os.environ.get('DTS_SHARD_ALWAYS_COLLECT', '0') == '1'
        if self.dist_on:
            if self.world == 1:
                import socket
                with socket.socket() as so:
                    so.bind(('127.0.0.1', 0))
                    port = so.getsockname()[1]
                for k_, v_ in (('MASTER_ADDR', '127.0.0.1'), ('MASTER_PORT', str(port)), ('RANK', '0'), ('WORLD_SIZE', '1')):
                    os.environ.setdefault(k_, v_)
            os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
            torch.cuda.set_device(self.dev)
            # RCCL over xGMI ("nccl" IS RCCL on ROCm).  DTS_DIST_BACKEND=gloo exists only to rehearse the multi-rank control flow on
            # a one-GPU box (ranks share the card; collectives staged through the host).
            self.backend = os.environ.get('DTS_DIST_BACKEND', 'nccl')
            if self.backend == 'nccl':
                dist.init_process_group('nccl', device_id=self.dev)
            else:
                dist.init_process_group(self.backend)
            assert dist.get_world_size() == a.gpus, (dist.get_world_size(), a.gpus)
        torch.cuda.set_device(self.dev)

    def barrier(self):
        if self.dist_on:
            self.dist.barrier(device_ids=[self.dev.index]) if self.backend == 'nccl' else self.dist.barrier()
        self.torch.cuda.synchronize(self.dev)

    def max_over_ranks(self, dt):
        if self.dist_on:
            t = self.torch.tensor([dt], dtype=self.torch.float64, device=self.dev if self.backend == 'nccl' else 'cpu')
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
            dt = float(t)
        return dt

    def timed(self, one_step, steps, warmup):
        """W untimed steps, then exactly K steps between barrier + synchronize on both sides; MAX over ranks."""
        torch = self.torch
        for s in range(warmup):
            one_step(s)
            torch.cuda.synchronize(self.dev)
        self.barrier()
        t0 = time.perf_counter()
        for s in range(steps):
            one_step(warmup + s)
        self.barrier()
        return self.max_over_ranks(time.perf_counter() - t0)

    def finish(self):
        if self.dist_on:
            self.dist.destroy_process_group()


class EpsGreedyIteration:
    """One search iteration of BASELINE config 3 on `n_total` candidates sharded over the job's ranks, inputs resident in HBM."""

    def __init__(self, job, net, scorer, n_total, i_step=5, sets=4, seed=1234):
        import numpy as np
        import torch
        from diffusion_tts_amd import ops
        from diffusion_tts_amd.parallel import CandidateShards
        from diffusion_tts_amd.sampler import _Loop
        self.ops, self.torch = ops, torch
        self.job, self.net, self.scorer, self.n_total, self.i = job, net, scorer, n_total, i_step
        dev = job.dev
        self.shards = CandidateShards()
        self.shards.require_candidates(n_total, 'bench')
        self.lo, self.hi = self.shards.span(n_total)
        self.nl = self.hi - self.lo
        self.L = _Loop(net, dev, 18, 40, 0.05, 50, 1.003, None, self.shards)
        self.t_steps = sigma_steps()
        g = torch.Generator().manual_seed(seed)                       # same stream on every rank
        self.x_cur = (torch.randn(1, 3, 64, 64, generator=g, dtype=torch.float64) * self.t_steps[i_step]).to(dev)
        self.pivot0 = torch.randn(1, 3, 64, 64, generator=g, dtype=torch.float64).to(dev)
        self.pivot = self.pivot0.clone()
        lab = torch.eye(1000)[torch.tensor([7])].to(dev)
        self.lab_l = lab.repeat(self.nl, 1).contiguous()
        lam = 0.15 * np.sqrt(3 * 64 * 64)
        # the full candidate set is replicated on every rank (as the real loop replicates the host RNG stream): the survivor
        # is rebuilt everywhere from its index alone
        self.noise, self.mode, self.scale = [], [], []
        for _ in range(sets):
            self.noise.append(torch.randn(n_total, 3, 64, 64, generator=g, dtype=torch.float64).to(dev))
            self.mode.append((torch.rand(n_total, generator=g) < 0.6).to(torch.int32).to(dev))
            self.scale.append((torch.rand(n_total, generator=g) * lam).float().to(dev))

    def candidates(self, q, lo, hi):
        return self.ops.candidate_noise(self.pivot, self.noise[q][lo:hi].contiguous(), self.mode[q][lo:hi].contiguous(),
                                        self.scale[q][lo:hi].contiguous())

    def __call__(self, s):
        q = s % len(self.noise)
        cand = self.candidates(q, self.lo, self.hi)
        _, x0 = self.L.step(self.x_cur, self.t_steps[self.i], self.t_steps[self.i + 1], self.i, cand, self.lab_l, nb=self.nl)
        loc = self.L.score(self.scorer, x0, self.lab_l).to(self.job.dev, self.torch.float32)
        scores = self.shards.gather_rewards(loc, self.n_total, 1).cpu()
        best = int(scores.argmax())                                   # first max; identical on every rank
        self.pivot = self.candidates(q, best, best + 1)               # every rank rebuilds the survivor (edm/main.py:848-857)
        self.last_scores = scores
        return best


def scorer_dtype(dtype):
    """The classifier runs in float16 beside a bfloat16 denoiser (scorers.ImageNetScorer: the reward error is the classifier's)."""
    import torch
    return torch.float16 if dtype == torch.bfloat16 else dtype


def build_adm(job, dtype, with_scorer=True, scorer_name='imagenet', sd=None, head_scale=None, with_net=True):
    """with_net=False: the scorer alone (net and state dict come back as None): the decidable-fixture legs need a second classifier head,
    not a second 296 M-parameter denoiser."""
    import warnings
    from diffusion_tts_amd import init as dinit
    from diffusion_tts_amd.config import adm_imagenet64
    from diffusion_tts_amd.networks import EDMPrecond
    from diffusion_tts_amd.scorers import ImageNetScorer, BrightnessScorer
    cfg = adm_imagenet64()
    net = None
    if with_net:
        if sd is None:
            sd, _ = dinit.refill_degenerate(dinit.edm_state_dict(cfg, 0), 0)
        net = EDMPrecond(cfg, sd, device=job.dev, dtype=dtype)
    scorer = None
    if with_scorer:
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            if scorer_name != 'imagenet':
                scorer = BrightnessScorer()
            elif head_scale is None:
                scorer = ImageNetScorer(device=job.dev, compute_dtype=scorer_dtype(dtype), seed=1)
            else:       # the decidable fixture: same draws, output head x head_scale (init.scale_classifier_head)
                from diffusion_tts_amd.config import ClassifierConfig
                csd, _ = dinit.refill_degenerate(dinit.classifier_state_dict(ClassifierConfig(), 1), 1)
                scorer = ImageNetScorer(weights=dinit.scale_classifier_head(csd, head_scale), device=job.dev, compute_dtype=scorer_dtype(dtype))
    return net, scorer, sd


def parity_leg(job, orc, nets, short_agreement=True):
    """GPU f32 / f16 / bf16 against the oracle iteration `orc` on the SAME 64 inputs, then index agreement of the 16-bit modes with
    the f32 mode over 8 more search iterations (K14-built candidates at eight sigma steps with churn)."""
    import torch
    from diffusion_tts_amd.sampler import _Loop
    from diffusion_tts_amd.parallel import CandidateShards
    dev = job.dev
    t_steps = sigma_steps()
    out = {'inputs': f'the cpu_baseline iteration: {orc["eps"].shape[0]} candidates, sigma step 5/18, seed 7'}
    srt = torch.sort(orc['rewards'], descending=True).values
    out['oracle_argmax'] = orc['best']
    out['oracle_top2_gap'] = float(srt[0] - srt[1])
    n = orc['eps'].shape[0]
    for name, (net, scorer) in nets.items():
        L = _Loop(net, dev, 18, 40, 0.05, 50, 1.003, None, CandidateShards(enabled=False))
        lab = orc['lab'].to(dev)
        _, x0 = L.step(orc['x'].to(dev), t_steps[5], t_steps[6], 5, orc['eps'].to(dev), lab, nb=n)
        sc = L.score(scorer, x0, lab).float().cpu()
        out[name] = {'max_reward_err': float((sc - orc['rewards']).abs().max()), 'argmax': int(sc.argmax()),
                     'index_equal': bool(int(sc.argmax()) == orc['best'])}
    if not short_agreement:                # the whole-search leg (teacher_forced_agreement) replaces the 8-iteration sample
        return out
    steps_i = [2, 3, 5, 7, 9, 11, 13, 14]
    agree = {name: 0 for name in nets if name != 'f32'}
    dev_max = {name: 0.0 for name in agree}                   # largest |reward - f32 reward| over all candidates and iterations
    regret = {name: [] for name in agree}                     # f32 reward given up by each differing pick (0 = same candidate)
    gaps = []
    for r, i_step in enumerate(steps_i):
        best, scores = {}, {}
        for name, (net, scorer) in nets.items():               # same seed => the same state, pivot and candidate set for every dtype
            it = EpsGreedyIteration(job, net, scorer, n, i_step=i_step, sets=1, seed=4321 + r)
            best[name] = it(0)
            scores[name] = it.last_scores.flatten().double().cpu()
            if name == 'f32':
                s = torch.sort(scores[name], descending=True).values
                gaps.append(float(s[0] - s[1]))
        for name in agree:
            agree[name] += int(best[name] == best['f32'])
            dev_max[name] = max(dev_max[name], float((scores[name] - scores['f32']).abs().max()))
            regret[name].append(float(f"{float(scores['f32'][best['f32']] - scores['f32'][best[name]]):.3e}"))
    # a pick can only be expected to survive a change of arithmetic when the f32 top-2 gap exceeds the mode's own reward noise:
    # `decidable` counts those iterations (gap > 2 x the mode's largest reward deviation) and how many of them agree
    decidable = {}
    for name in agree:
        idx = [k for k, g_ in enumerate(gaps) if g_ > 2 * dev_max[name]]
        decidable[name] = f"{sum(1 for k in idx if regret[name][k] == 0.0)}/{len(idx)}"
    out['index_agreement'] = {'reference': 'GPU f32 parity mode (checked against the oracle above)', 'iterations': len(steps_i),
                              'sigma_steps': steps_i, 'f32_top2_gaps': [float(f'{g_:.3e}') for g_ in gaps],
                              **{name: f'{v}/{len(steps_i)}' for name, v in agree.items()},
                              'max_reward_dev_vs_f32': {k_: float(f'{v:.3e}') for k_, v in dev_max.items()},
                              'f32_reward_given_up': regret, 'agree_where_gap_exceeds_2x_dev': decidable}
    return out


def teacher_forced_agreement(job, nets, n=64, K=4, num_steps=18, seed=2024, lambda_param=0.15, eps_p=0.4):
    """Index agreement of the 16-bit modes over ONE WHOLE config-3 search: 18 sigma steps x K = 4 local-search iterations = 72
    iterations of N = 64 candidates (edm/main.py:730-860).  The search is driven by the f32 parity mode (its argmax picks the pivot, its
    Heun step advances the state, edm/main.py:842-860) and every dtype evaluates the SAME state, pivot and candidate set in every iteration
    ("teacher forcing"), so all 72 decisions stay comparable instead of diverging after the first differing pick.
    Per 16-bit dtype: picks equal to f32's, the same count over the DECIDABLE iterations (f32 top-2 gap > 2 x the dtype's largest reward
    deviation from f32 anywhere in the search), the f32 reward given up by each differing pick (regret) and the largest reward deviation."""
    import numpy as np
    import torch
    from diffusion_tts_amd import ops
    from diffusion_tts_amd.sampler import _Loop
    from diffusion_tts_amd.parallel import CandidateShards
    dev = job.dev
    t_steps = sigma_steps()
    loops = {name: _Loop(net, dev, num_steps, 40, 0.05, 50, 1.003, None, CandidateShards(enabled=False)) for name, (net, _) in nets.items()}
    g = torch.Generator().manual_seed(seed)
    x_cur = (torch.randn(1, 3, 64, 64, generator=g, dtype=torch.float64) * t_steps[0]).to(dev)
    lab1 = torch.eye(1000)[torch.tensor([7])].to(dev)
    lab = lab1.repeat(n, 1).contiguous()
    lam = lambda_param * np.sqrt(3 * 64 * 64)
    others = [k_ for k_ in nets if k_ != 'f32']
    picks = {k_: [] for k_ in nets}
    devs = {k_: [] for k_ in others}
    regret = {k_: [] for k_ in others}
    gaps = []
    for i in range(num_steps):
        pivot = torch.randn(1, 3, 64, 64, generator=g, dtype=torch.float64).to(dev)           # edm/main.py:737
        for k in range(K):
            noise = torch.randn(n, 3, 64, 64, generator=g, dtype=torch.float64).to(dev)
            mode = (torch.rand(n, generator=g) < (1 - eps_p)).to(torch.int32).to(dev)           # :751 perturb the pivot | fresh noise
            scale = (torch.rand(n, generator=g) * lam).float().to(dev)
            cand = ops.candidate_noise(pivot, noise, mode, scale)
            sc = {}
            for name, (net, scorer) in nets.items():
                _, x0 = loops[name].step(x_cur, t_steps[i], t_steps[i + 1], i, cand, lab, nb=n)
                sc[name] = loops[name].score(scorer, x0, lab).double().cpu()
                picks[name].append(int(sc[name].argmax()))
            srt = torch.sort(sc['f32'], descending=True).values
            gaps.append(float(srt[0] - srt[1]))
            b = picks['f32'][-1]
            for name in others:
                devs[name].append(float((sc[name] - sc['f32']).abs().max()))
                regret[name].append(float(sc['f32'][b] - sc['f32'][picks[name][-1]]))
            pivot = ops.candidate_noise(pivot, noise[b:b + 1].contiguous(), mode[b:b + 1].contiguous(), scale[b:b + 1].contiguous())
        x_cur, _ = loops['f32'].step(x_cur, t_steps[i], t_steps[i + 1], i, pivot, lab1)          # edm/main.py:860
    iters = num_steps * K
    out = {'iterations': iters, 'candidates': n, 'driver': 'GPU f32 parity mode (teacher forcing: every dtype sees f32\'s state, pivot and candidates)',
           'f32_top2_gap': {'min': float(f'{min(gaps):.3e}'), 'median': float(f'{sorted(gaps)[len(gaps) // 2]:.3e}'), 'max': float(f'{max(gaps):.3e}')}}
    for name in others:
        dmax = max(devs[name])
        same = [int(a_ == b_) for a_, b_ in zip(picks[name], picks['f32'])]
        dec = [j for j, g_ in enumerate(gaps) if g_ > 2 * dmax]
        out[name] = {'agree': f'{sum(same)}/{iters}', 'agree_decidable': f'{sum(same[j] for j in dec)}/{len(dec)}',
                     'max_reward_dev_vs_f32': float(f'{dmax:.3e}'), 'max_regret': float(f'{max(regret[name]):.3e}'),
                     'sum_regret': float(f'{sum(regret[name]):.3e}')}
    return out


def e2e_leg(job, net, scorer, dtype):
    """BASELINE config 3 end to end: generate_image_grid, eps-greedy N=64 K=4, 18 sigma steps, host RNG + uploads included."""
    import torch
    from diffusion_tts_amd.sampler import SamplingMethod, generate_image_grid
    lat = torch.randn(1, 3, 64, 64, generator=torch.Generator().manual_seed(3))
    lab = torch.eye(1000)[torch.tensor([5])]
    best = None
    for rep in range(2):
        torch.cuda.synchronize(job.dev)
        t0 = time.perf_counter()
        res = generate_image_grid(net, None, lat, lab, seed=0, gridw=1, gridh=1, device=job.dev, num_steps=18, S_churn=40,
                                  S_min=0.05, S_max=50, S_noise=1.003, sampling_method=SamplingMethod.EPS_GREEDY,
                                  sampling_params=dict(scorer=scorer, N=64, K=4, lambda_param=0.15, eps=0.4),
                                  compute_dtype=dtype, verbose=False)
        torch.cuda.synchronize(job.dev)
        dt = time.perf_counter() - t0
        rate = res['net_rows'] / dt
        best = rate if best is None else max(best, rate)
    return {'e2e_evals_per_s': round(best, 1), 'e2e_rows': res['net_rows'], 'e2e_seconds_per_image': round(res['net_rows'] / best, 3)}


# the reward deviation between two correct fp32 implementations of this fixture (measured: GPU f32 mode vs the reference's CPU run 1.3e-8, f16x3 1.8e-8,
# f32 mode vs f16x3 1.5e-8): a first differing selection counts as "below fp32 noise" only against THIS scale -- a 16-bit mode whose own noise
# (3e-7 .. 5e-7) swamps a 1.6e-7 gap differs at a decision fp32 arithmetic decides
FP32_REWARD_NOISE = 2e-8


def reference_run_golden():
    """tests/golden/config3_golden.npz + manifest: the REFERENCE's own end-to-end run of BASELINE configs[2] (edm/main.py generate_image_grid on
    the CPU of the build container, tests/golden/make_golden_config3.py) -- arrays and scalars only; None when the fixture is not in the tree."""
    import numpy as np
    gd = os.path.join(ROOT, 'tests', 'golden')
    try:
        with open(os.path.join(gd, 'config3_manifest.json')) as f:
            man = json.load(f)
        return np.load(os.path.join(gd, 'config3_golden.npz')), man, np.load(os.path.join(gd, 'fullsize_golden.npz'))['eg64_latents']
    except OSError:
        return None


def free_running_vs_f32(job, nets):
    """One FREE-RUNNING config-3 search (generate_image_grid, eps-greedy N = 64 K = 4, 18 sigma steps, the same host RNG) per compute
    mode, compared (a) with the f32 parity mode's and (b) with THE REFERENCE'S OWN RUN of the same search (reference_run_golden: same seed,
    latents, label and weights): how many of the 72 selections coincide, where the first differing selection is, and max |x_final - x_final(ref)|
    (north_star: selected indices bit-exact, final images within 1e-3 abs).  A 16-bit search that picks another near-tied candidate
    once follows another trajectory from there on, so its final image is a different sample: the figure says how different."""
    import numpy as np
    import torch
    from diffusion_tts_amd.sampler import SamplingMethod, generate_image_grid
    from diffusion_tts_amd.hashing import seed0_scale
    lat = torch.randn(1, 3, 64, 64, generator=torch.Generator().manual_seed(3))
    lab = torch.eye(1000)[torch.tensor([5])]
    gold = reference_run_golden()
    seed = 0
    if gold is not None:
        g, gm, glat = gold
        assert np.array_equal(lat.numpy(), glat) and gm['params'] == dict(N=64, K=4, lambda_param=0.15, eps=0.4) and gm['num_steps'] == 18, 'fixture changed'
        seed = int(gm['seed'])
    res = {}
    for name, (net, scorer) in nets.items():
        r = generate_image_grid(net, None, lat, lab, seed=seed, gridw=1, gridh=1, device=job.dev, num_steps=18, S_churn=40, S_min=0.05,
                                S_max=50, S_noise=1.003, sampling_method=SamplingMethod.EPS_GREEDY,
                                sampling_params=dict(scorer=scorer, N=64, K=4, lambda_param=0.15, eps=0.4), scale_fn=seed0_scale,
                                compute_dtype=torch_dtype(name), reuse_winner=False, verbose=False)
        res[name] = (r['x'].double().cpu(), [int(s_[0]) for s_ in r['selected']], float(r['final_scores'][0]), [w_.reshape(-1).double() for w_ in r['rewards']],
                     r['image'][0].permute(1, 2, 0).numpy())

    def vs_reference(name):
        x, sel, sc, rew, img = res[name]
        same = [int(a_ == int(b_)) for a_, b_ in zip(sel, g['selected'])]
        first = same.index(0) if 0 in same else None
        upto = len(sel) if first is None else first + 1
        err = max(float((rew[j] - torch.from_numpy(g['rewards'][j]).double()).abs().max()) for j in range(upto))
        rec = {'same_selections': f'{sum(same)}/{len(same)}', 'first_differing_selection': first, 'max_reward_err_while_states_equal': float(f'{err:.3e}')}
        if first is None:
            rec['max_abs_x_final'] = float(f'{float((x - torch.from_numpy(g["last_D"]).double()).abs().max()):.3e}')
            rec['png_pixels_differing'] = int((img.astype(np.int32) != g['image'].astype(np.int32)).sum())
            rec['final_score_err'] = float(f'{abs(sc - float(g["final_score"][0])):.1e}')
        else:
            rec['reference_top2_gap_at_first_difference'] = float(f'{gm["top2_gaps"][first]:.3e}')
            rec['first_difference_is_below_fp32_noise'] = bool(gm['top2_gaps'][first] <= 4 * min(err, FP32_REWARD_NOISE))
        return rec
    out = {'search': f'config 3 end to end (eps-greedy N=64 K=4, 18 sigma steps, seed {seed}), every mode from the same host RNG; reference = f32 parity mode',
           'f32_final_score': res['f32'][2]}
    if gold is not None:
        out['reference_run'] = {'source': 'tests/golden/config3_golden.npz: the reference\'s own generate_image_grid run of this search (CPU, tests/golden/'
                                          f'make_golden_config3.py): {gm["net_rows"]} rows, {gm["exact_ties"]} exact ties, smallest other top-2 gap {gm["min_nonzero_gap"]:.2e}',
                                'f32': vs_reference('f32')}
    for name in nets:
        if name == 'f32':
            continue
        x, sel, sc, rew, _ = res[name]
        same = [int(a_ == b_) for a_, b_ in zip(sel, res['f32'][1])]
        first = same.index(0) if 0 in same else None
        out[name] = {'max_abs_x_final_vs_f32': float(f'{float((x - res["f32"][0]).abs().max()):.3e}'), 'same_selections': f'{sum(same)}/{len(same)}',
                     'first_differing_selection': first, 'final_score': sc}
        # up to (and including) the first differing selection both searches evaluated the SAME candidates: there the reward deviation is the
        # mode's noise and the f32 top-2 gap says whether that decision was decidable at all in fp32 arithmetic
        upto = len(sel) if first is None else first + 1
        dev = max(float((rew[j] - res['f32'][3][j]).abs().max()) for j in range(upto))
        out[name]['max_reward_dev_while_states_equal'] = float(f'{dev:.3e}')
        if first is not None:
            srt = torch.sort(res['f32'][3][first], descending=True).values
            gap = float(srt[0] - srt[1])
            out[name]['f32_top2_gap_at_first_difference'] = float(f'{gap:.3e}')
            out[name]['first_difference_is_below_fp32_noise'] = bool(gap <= 4 * min(dev, FP32_REWARD_NOISE))      # two fp32 summation orders disagree there too
        if gold is not None:
            out[name]['vs_reference_run'] = vs_reference(name)
    return out


def sharded_forms_vs_reference(job, net, scorer, dtype_name, chunk=8):
    """The launch forms of a SHARDED search (BASELINE configs[2]: "candidates sharded 8 x") against the reference's own run: at 8 candidates
    per rank every convolution picks other split-K factors / kernel forms than at 64 (another fixed f32 summation order).
    generate_image_grid(candidate_chunk=8) issues exactly the launches rank r of 8 issues for its share, eight times per iteration, walked
    along the reference's recorded selections so that all 72 decisions see the reference's candidates: reward errors, the build's own
    argmax against the reference's wherever the reference's top-2 gap exceeds 4x the error there, final state and PNG."""
    import numpy as np
    import torch
    from diffusion_tts_amd.sampler import SamplingMethod, generate_image_grid
    from diffusion_tts_amd.hashing import seed0_scale
    gold = reference_run_golden()
    if gold is None:
        return None
    g, gm, glat = gold
    lab = torch.eye(1000)[torch.tensor([5])]
    sel_ref = [int(v) for v in g['selected']]
    r = generate_image_grid(net, None, torch.from_numpy(glat), lab, seed=int(gm['seed']), gridw=1, gridh=1, device=job.dev, num_steps=18, S_churn=40,
                            S_min=0.05, S_max=50, S_noise=1.003, sampling_method=SamplingMethod.EPS_GREEDY,
                            sampling_params=dict(scorer=scorer, N=64, K=4, lambda_param=0.15, eps=0.4), scale_fn=seed0_scale,
                            compute_dtype=torch_dtype(dtype_name), reuse_winner=False, verbose=False, forced_selections=sel_ref, candidate_chunk=chunk)
    rew = np.stack([w_.reshape(-1).numpy() for w_ in r['rewards']]).astype(np.float64)
    own = [int(s_[0]) for s_ in r['selected']]
    errs = np.abs(rew - g['rewards'].astype(np.float64)).max(axis=1)
    gaps = gm['top2_gaps']
    dec = [d for d in range(72) if gaps[d] > 0 and gaps[d] > 4 * errs[d]]
    img = r['image'][0].permute(1, 2, 0).numpy()
    return {'what': f'config 3 walked along the reference run\'s selections with the candidates evaluated in pieces of {chunk} (the kernels of one rank of {64 // chunk})',
            'dtype': dtype_name, 'max_reward_err': float(f'{errs.max():.3e}'), 'own_argmax_equals_reference': f'{sum(int(a_ == b_) for a_, b_ in zip(own, sel_ref))}/72',
            'decidable': f'{sum(int(own[d] == sel_ref[d]) for d in dec)}/{len(dec)} (reference top-2 gap > 4x the reward error there; {sum(1 for g_ in gaps if g_ == 0)} exact ties besides)',
            'net_rows': int(r['net_rows']), 'max_abs_x_final': float(f'{float((r["x"].double().cpu() - torch.from_numpy(g["last_D"]).double()).abs().max()):.3e}'),
            'png_pixels_differing': int((img.astype(np.int32) != g['image'].astype(np.int32)).sum())}


def _meets_vs_reference(rec):
    """north star against the reference's own run: every selection equal and the final image within 1e-3 (or the first difference at a decision
    the reference itself decided by less than the fp32 reward noise); True when the fixture is absent (then only the f32-mode comparison speaks)"""
    if rec is None:
        return True
    if rec['first_differing_selection'] is None:
        return rec['max_abs_x_final'] <= 1e-3
    return bool(rec.get('first_difference_is_below_fp32_noise', False))


def parity_mode_records(a, job, nets):
    """The timed region of the headline (one eps-greedy iteration over N = 64 candidates) in the modes that meet the north star's tolerance:
    the f32 parity mode (v_mfma_f32_16x16x4_f32, peak 157.3 TFLOP/s) and the split-precision mode f16x3 (16-bit MFMA, three passes)."""
    out = {}
    for name in ('f32', 'f16x3'):
        if name == a.dtype:         # the headline IS this mode: its record is the line itself
            out[name] = 'the headline of this line (value / ms_per_step / roofline above)'
            continue
        net, scorer = nets[name]
        it = EpsGreedyIteration(job, net, scorer, 64)
        for s_ in range(3):
            it(s_)
        steps = 4 if name == 'f32' else 10
        dt = job.timed(it, steps, 1)
        v = 2 * 64 * steps / dt
        out[name] = {'value': round(v, 2), 'unit': 'candidate U-Net steps/sec', 'ms_per_step': round(dt / steps * 1e3, 3), 'steps': steps,
                     'config': f'{name} denoiser + {name} classifier, N=64, graph replay',
                     'roofline': {'bound': 'mfma', 'achieved': round(v * GFLOP_PER_EVAL['adm64'] / 1e3, 1), 'peak': PEAK_TFLOPS[name], 'unit': 'TFLOP/s',
                                  'frac': round(v * GFLOP_PER_EVAL['adm64'] * 1e9 / (PEAK_TFLOPS[name] * 1e12), 4), 'traffic': None,
                                  'kernel': 'whole step (denoiser FLOPs only, algorithmic: 219.33 GFLOP per evaluation)'}}
        del it
        log(f"parity mode {name}: {out[name]['value']} evals/s, {out[name]['ms_per_step']} ms/step")
    return out


# ----------------------------------------------------------------------------------------------------------------------
def sub_records(a, job, net, scorer, dtype, sd, bf16_pair=None):
    """The other numbers DESIGN.md quotes, measured in the SAME driver-run process as the headline (each with its own ms_per_step and
    roofline): the per-GPU share of an 8-GPU run (8 candidates), BASELINE configs[1] at 32x32 with its own CPU baseline, an MCTS slice
    (configs[4] at S = --mcts-slice) and the iteration with the reference's fp32 scorer arithmetic (main.py:69)."""
    import copy
    import torch
    from diffusion_tts_amd.scorers import ImageNetScorer
    out = {}
    # (1) 8 candidates per GPU: what each rank of `--gpus 8` runs (N = 64 sharded 8 ways), minus the all-gather
    it8 = EpsGreedyIteration(job, net, scorer, 8)
    for s in range(3):
        it8(s)
    dt8 = job.timed(it8, 40, 3)
    v8 = 2 * 8 * 40 / dt8
    a8 = copy.copy(a)
    a8.candidates, a8.conv_table, a8.conv_sequence = 8, False, None
    caches = [net._graphs] + ([scorer.model._graphs] if hasattr(scorer, 'model') else [])
    roof8 = None if a.no_kernel_timing else conv_roofline(a8, lambda s: it8(s), 2, a.dtype, v8, GFLOP_PER_EVAL['adm64'], caches)
    out['share_8_per_gpu'] = {'value': round(v8, 2), 'unit': 'candidate U-Net steps/sec per GPU', 'ms_per_step': round(dt8 / 40 * 1e3, 3), 'steps': 40,
                              'candidates_per_gpu': 8, 'projected_8gpu_value': round(8 * v8, 1),
                              'note': 'the per-GPU work of --gpus 8 at N=64 (strong scaling), without the reward all-gather', 'roofline': roof8}
    out['share_8_per_gpu']['dtype'] = a.dtype
    out['share_8_per_gpu']['projected_8gpu_speedup_vs_this_line'] = 'projected_8gpu_value / value of this line (same dtype)'
    del it8
    log(f"sub-record share_8_per_gpu: {out['share_8_per_gpu']['ms_per_step']} ms/step")
    if bf16_pair is not None:       # the same 8-row share in the bf16 throughput mode (rounds 1-4 quoted this one)
        itb = EpsGreedyIteration(job, bf16_pair[0], bf16_pair[1], 8)
        for s in range(3):
            itb(s)
        dtb = job.timed(itb, 40, 3)
        out['share_8_per_gpu_bf16'] = {'value': round(2 * 8 * 40 / dtb, 2), 'unit': 'candidate U-Net steps/sec per GPU', 'ms_per_step': round(dtb / 40 * 1e3, 3),
                                       'steps': 40, 'candidates_per_gpu': 8, 'dtype': 'bf16', 'projected_8gpu_value': round(8 * 2 * 8 * 40 / dtb, 1)}
        del itb
        log(f"sub-record share_8_per_gpu_bf16: {out['share_8_per_gpu_bf16']['ms_per_step']} ms/step")
    if a.dtype in ('bf16', 'f16'):
        # (2) a 16-bit denoiser with an fp32 classifier (the reference scores in fp32, main.py:69): what exact-precision scoring costs beside it
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            sc32 = ImageNetScorer(device=job.dev, compute_dtype=torch.float32, seed=1)
        it32 = EpsGreedyIteration(job, net, sc32, 64)
        for s in range(3):
            it32(s)
        dt32 = job.timed(it32, 10, 2)
        out['f32_scorer'] = {'value': round(2 * 64 * 10 / dt32, 2), 'unit': 'candidate U-Net steps/sec', 'ms_per_step': round(dt32 / 10 * 1e3, 3), 'steps': 10,
                             'config': f'{a.dtype} denoiser + float32 classifier (parity-mode scorer kernels), N=64',
                             'whole_step_frac': round(2 * 64 * 10 / dt32 * GFLOP_PER_EVAL['adm64'] * 1e9 / (PEAK_TFLOPS[a.dtype] * 1e12), 4)}
        del it32, sc32
        log(f"sub-record f32_scorer: {out['f32_scorer']['ms_per_step']} ms/step")
    # (3) MCTS slice (BASELINE configs[4] shape at a smaller S)
    rec = mcts_record(a, job, a.mcts_slice, 1, 0, built=(net, scorer))
    out['mcts_s'] = {k_: rec[k_] for k_ in ('metric', 'value', 'unit', 'ms_per_step', 'steps', 'dtype', 'config', 'roofline')}
    log(f"sub-record mcts_s (S={a.mcts_slice}): {rec['value']} evals/s")
    # (4) BASELINE configs[1]: DDPM++ CIFAR-32 rejection N=16 with its CPU baseline
    a32 = copy.copy(a)
    a32.conv_table, a32.conv_sequence, a32.workload, a32.candidates = False, None, 'ddpmpp32_rejection', 16
    rec = rejection32_record(a32, job, steps=20 if a.dtype in ('f16x3', 'f32') else 40, warmup=3)
    out['ddpmpp32_rejection'] = {k_: rec[k_] for k_ in ('metric', 'value', 'unit', 'ms_per_step', 'steps', 'dtype', 'config', 'roofline', 'cpu_baseline')}
    log(f"sub-record ddpmpp32_rejection: {rec['value']} evals/s, cpu {rec['cpu_baseline']}")
    # (5) BASELINE configs[3] with this build's parts: SD beam B=4 N=16 over [N,4,64,64] fp16 latents -> [N,3,512,512] decodes
    try:
        out['sd_beam_config4'] = sd_beam_record(job)
        log(f"sub-record sd_beam_config4: {out['sd_beam_config4']['value']} decodes/s")
    except Exception as e:          # the stand-in U-Net lives with the tests; a tree without them still prints the headline
        out['sd_beam_config4'] = {'error': f'{type(e).__name__}: {e}'}
    return out


def sd_beam_record(job, steps=4, B=4, N=16):
    """SD beam search at BASELINE configs[3] size through SDSearchPipeline: candidate-batched U-Net calls (a shape-faithful stand-in: the
    diffusers U-Net is an opaque module on this path and is not available offline), the fused DDIM candidate step, 64-row decodes through
    the HIP VAE decoder (SD-1.5 width, random init), CLIP scorer (random-init CLIP) with device-side pre-processing, one scorer call per
    decoded batch.  Unit: candidate decodes per second (the VAE decode, 2.48 TFLOP per candidate, is the part of config 4 this build owns)."""
    import torch
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from sd_standins import shape_unet, TinyTextEncoder, TinyTokenizer, tiny_clip
    from diffusion_tts_amd import init as dinit
    from diffusion_tts_amd.sd_pipeline import SDSearchPipeline
    from diffusion_tts_amd.scorers import CLIPScorer, ByteTokenizer
    from diffusion_tts_amd.vae import VAEDecoder
    dev = job.dev
    dec = VAEDecoder(dinit.vae_decoder_state_dict(seed=5), device=dev, dtype=torch.float16)
    unet, te = shape_unet().half().to(dev), TinyTextEncoder().half().to(dev)
    pipe = SDSearchPipeline(unet, dec, device=dev, text_encoder=te, tokenizer=TinyTokenizer())
    scorer = CLIPScorer(model=tiny_clip(0), tokenizer=ByteTokenizer(1000, 998, 999), device=dev)
    best, nd = None, 0
    for rep in range(3):
        torch.manual_seed(7)
        lat = torch.randn(1, 4, 64, 64).half()
        d0 = dec.decodes
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        out, _ = pipe(prompt='a photo of a cat', latents=lat, num_inference_steps=steps, score_function=scorer, method='beam',
                      params={'N': N, 'B': B, 'K': 20, 'lambda': 0.15, 'eps': 0.4, 'S': 8}, output_type='pt')
        torch.cuda.synchronize(dev)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
        nd = dec.decodes - d0
    return {'metric': f'candidate VAE decodes/sec, SD beam B={B} N={N}, CLIP scorer', 'value': round(nd / best, 1), 'unit': 'candidate decodes/sec',
            'seconds_per_search': round(best, 3), 'ddim_steps': steps, 'decodes': nd, 'unet_rows': out.unet_rows, 'scorer_calls': out.scorer_calls,
            'device_preprocessed_images': scorer.device_preprocessed,
            'roofline': {'bound': 'mfma', 'achieved': round(nd * 2.48 / best, 1), 'peak': PEAK_TFLOPS['f16'], 'unit': 'TFLOP/s',
                         'frac': round(nd * 2.48 / best / PEAK_TFLOPS['f16'], 4), 'traffic': None,
                         'kernel': 'whole search (VAE decoder FLOPs only; stand-in U-Net, CLIP and the loop included in the time)'},
            'config': {'workload': 'SD-1.5-shaped beam search: [N,4,64,64] fp16 latents, stand-in U-Net, HIP VAE decoder (random init), random-init CLIP scorer'}}


def run_eps_greedy(a, job):
    import torch
    dtype = torch_dtype(a.dtype)
    world, rank = job.world, job.rank
    net, scorer, sd = build_adm(job, dtype, scorer_name=a.scorer)
    log('denoiser + scorer packed on device')
    n_total = a.candidates * world if a.scaling == 'weak' else a.candidates
    it = EpsGreedyIteration(job, net, scorer, n_total)
    # setup, not warm-up: the forwards are captured as HIP graphs on the third call of a shape (graphs.py); untimed iterations here
    # keep that one-off capture (~0.5 s) out of the W warm-up steps and the K timed steps whatever W is
    for s in range(3):
        it(s)
    torch.cuda.synchronize(job.dev)
    log(f'state resident; {it.nl} of {n_total} candidates on this rank; forwards captured ({net._graphs.captures} graphs)')
    evals0, replays0 = net.evals, net._graphs.replays + 2 * a.warmup
    dt = job.timed(it, a.steps, a.warmup)
    rows_local = net.evals - evals0
    graph_replays = net._graphs.replays - replays0
    value = 2 * n_total * a.steps / dt
    log(f'timed region: {dt:.3f}s for {a.steps} steps -> {value:.1f} evals/s')

    weak = None
    if world > 1 and a.scaling == 'strong' and not a.no_weak:        # weak-scaling rate in the same run (64 candidates PER GPU)
        itw = EpsGreedyIteration(job, net, scorer, a.candidates * world)
        for s in range(3):
            itw(s)
        dtw = job.timed(itw, a.steps, 1)
        weak = 2 * a.candidates * world * a.steps / dtw
        del itw
        log(f'weak-scaling leg: {weak:.1f} evals/s')

    roof = None
    if not a.no_kernel_timing:
        caches = [net._graphs] + ([scorer.model._graphs] if hasattr(scorer, 'model') else [])
        reps = min(2, a.steps)
        if rank == 0:
            roof = conv_roofline(a, lambda s: it(a.warmup + s), reps, a.dtype, value / world, GFLOP_PER_EVAL['adm64'], caches)
        else:                                   # the steps contain the reward all-gather: every rank must take part
            for s in range(reps):
                it(a.warmup + s)
            torch.cuda.synchronize(job.dev)
    job.barrier()

    extra = {}
    if rank == 0 and world == 1:
        if not a.no_e2e and a.candidates == 64 and a.scorer == 'imagenet':
            extra.update(e2e_leg(job, net, scorer, dtype))
            log(f'end-to-end config 3: {extra["e2e_evals_per_s"]} evals/s')
        cpu = orc = None
        if not a.no_cpu_baseline:
            log('CPU baseline (oracle) running')
            cpu, orc = oracle_iteration(a.cpu_sample)
            log('CPU baseline done')
        extra['cpu_baseline'] = cpu
        if orc is not None and not a.no_parity and a.scorer == 'imagenet':
            nets = {a.dtype: (net, scorer)}
            modes = ('f32', 'f16x3', 'f16', 'bf16')
            for name in modes:
                if name not in nets:
                    n_, s_, _ = build_adm(job, torch_dtype(name), sd=sd)
                    nets[name] = (n_, s_)
            want_sub = not a.no_subrecords and a.candidates == 64
            extra['parity'] = parity_leg(job, orc, {k: nets[k] for k in modes}, short_agreement=not want_sub)
            log('parity leg done')
            if want_sub:       # index agreement as a RATE: one whole config-3 search, teacher-forced on the f32 pivots
                extra['parity']['index_agreement'] = teacher_forced_agreement(job, {k: nets[k] for k in modes}, n=64)
                log(f"whole-search index agreement: {extra['parity']['index_agreement']}")
                # the same on the second fixture: classifier output head x 20 (logit std ~3.4, init.scale_classifier_head)
                from diffusion_tts_amd import init as dinit
                hs = dinit.HEAD_SCALE_DECIDABLE
                nets_h = {k: (nets[k][0], build_adm(job, torch_dtype(k), head_scale=hs, with_net=False)[1]) for k in modes}
                extra['parity']['index_agreement_scaled_head'] = dict(teacher_forced_agreement(job, nets_h, n=64), head_scale=hs,
                    fixture='same constructors and seeds, AttentionPool2d.c_proj x head_scale in oracle and build alike')
                log(f"whole-search index agreement, scaled head: {extra['parity']['index_agreement_scaled_head']}")
                del nets_h
                extra['parity']['free_running_vs_f32'] = free_running_vs_f32(job, {k: nets[k] for k in modes})
                log(f"free-running searches vs f32: {extra['parity']['free_running_vs_f32']}")
                fr = extra['parity']['free_running_vs_f32'].get(a.dtype)
                # north star: selected indices bit-exact, final images within 1e-3 abs -- for the mode `value` is measured in
                extra['parity']['headline'] = ({'dtype': a.dtype, 'reference': 'GPU f32 parity mode (equal to the CPU oracle in tests/, which the reference\'s own goldens pin)',
                                                'same_selections': fr['same_selections'], 'max_abs_x_final': fr['max_abs_x_final_vs_f32'],
                                                'first_differing_selection': fr['first_differing_selection'],
                                                'f32_top2_gap_at_first_difference': fr.get('f32_top2_gap_at_first_difference'),
                                                'max_reward_dev_while_states_equal': fr['max_reward_dev_while_states_equal'],
                                                # the same search against THE REFERENCE'S OWN RUN of it (tests/golden/config3_golden.npz), when the fixture is in the tree
                                                'vs_reference_run': fr.get('vs_reference_run'),
                                                'fixture_seed': ('the search seed of this fixture (71) was picked on the GPU beforehand for wide top-2 margins (tools/seed_scan.py, '
                                                                 'profiles/r05_seed_scan.txt: on most seeds some decision is decided by less than fp32 noise, i.e. a coin flip between '
                                                                 'two correct fp32 summation orders); a second reference run at an UNSCANNED seed (0) is walked decision by decision in '
                                                                 'tests/test_gpu_fullsize.py::test_config3_second_seed_walk_against_the_reference_run'),
                                                # every selection equal, or the only differences sit where the reference's own top-2 gap is below fp32 noise
                                                # (a decision no fp32 implementation reproduces: the f32 mode's own picks move there with its summation order)
                                                'meets_north_star': bool((fr['same_selections'] == '72/72' or fr.get('first_difference_is_below_fp32_noise', False))
                                                                         and fr['max_abs_x_final_vs_f32'] <= 1e-3 and _meets_vs_reference(fr.get('vs_reference_run')))}
                                               if fr is not None else {'dtype': a.dtype, 'reference': 'this IS the f32 parity mode'})
                try:
                    extra['parity']['sharded_launch_forms'] = sharded_forms_vs_reference(job, nets[a.dtype][0], nets[a.dtype][1], a.dtype)
                    log(f"sharded launch forms vs the reference run: {extra['parity']['sharded_launch_forms']}")
                except Exception as e:          # (a record, not the measurement: say so instead of losing the line)
                    extra['parity']['sharded_launch_forms'] = {'error': f'{type(e).__name__}: {e}'}
                extra['parity_modes'] = parity_mode_records(a, job, nets)
            # the same timed region in the 16-bit THROUGHPUT modes (f16 is the reference's own CUDA dtype, networks.py:658): every throughput and
            # index-agreement figure then sits in ONE driver-run record.  They do not reproduce the reference's selections (parity.free_running_vs_f32).
            others = [m for m in ('bf16', 'f16') if m != a.dtype]
            extra['other_dtype'] = {}
            for other in others:
                ito = EpsGreedyIteration(job, nets[other][0], nets[other][1], n_total)
                for s in range(3):
                    ito(s)
                st_o = max(a.steps, 10)
                dto = job.timed(ito, st_o, a.warmup)
                extra['other_dtype'][other] = {'value': round(2 * n_total * st_o / dto, 2), 'ms_per_step': round(dto / st_o * 1e3, 3), 'steps': st_o,
                                               'scorer_dtype': str(scorer_dtype(torch_dtype(other))).split('.')[-1],
                                               'whole_step_frac': round(2 * n_total * st_o / dto * GFLOP_PER_EVAL['adm64'] * 1e9 / (PEAK_TFLOPS[other] * 1e12), 4),
                                               'note': 'throughput mode: NOT the reference\'s selections on the same seed (see parity.free_running_vs_f32)'}
                del ito
                log(f"{other}: {extra['other_dtype'][other]['value']} evals/s")
            nets_tp = nets.get('bf16')
            del nets
        else:
            nets_tp = None
        if not a.no_subrecords and a.candidates == 64 and a.scorer == 'imagenet':
            extra['sub_records'] = sub_records(a, job, net, scorer, dtype, sd, bf16_pair=nets_tp if a.dtype != 'bf16' else None)
    if rank == 0:
        out = {
            'metric': 'candidate U-Net steps/sec, EDM ImageNet-64 eps-greedy N=64', 'value': round(value, 2),
            'unit': 'candidate U-Net steps/sec', 'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup,
            'ms_per_step': round(dt / a.steps * 1e3, 3), 'higher_is_better': True, 'scaling': a.scaling if world > 1 else 'strong',
            'vs_baseline': None, 'dtype': a.dtype, 'data': 'synthetic (N(0,1) latents/noises, random-init weights + weight rule)',
            'config': {'workload': 'EDM ImageNet-64 (ADM 295.9M) eps-greedy search iteration, imagenet scorer, sigma step 5/18',
                       'candidates_total': n_total, 'candidates_per_gpu': it.nl, 'rows_per_step_total': 2 * n_total,
                       'scorer': a.scorer, 'scorer_dtype': str(scorer_dtype(dtype)).split('.')[-1] if a.scorer == 'imagenet' else 'f64',
                       'denoiser_dtype': a.dtype,
                       # every forward of the timed region was a replayed HIP graph (a refused capture is an ERROR under bench.py: DTS_GRAPHS_STRICT)
                       'graph_replay': bool(net._graphs.enabled and graph_replays > 0), 'graph_replays_in_timed_region': graph_replays,
                       'reuse_winner': 'n/a: a bench step evaluates all N candidates.  generate_image_grid (e2e_rows): f16x3 / f32 recompute the final pivot step at batch 1 like '
                                       'edm/main.py:860 (8 995 rows per config-3 search = the reference\'s count); bf16 / f16 reuse the winner\'s row (8 960 rows, same values)',
                       'parity_grade': a.dtype in ('f16x3', 'f32'),
                       'parallelism': f'candidates sharded x{world}, 1 all-gather of rewards per step'},
            'rccl_ranks': world if (world > 1 and job.backend == 'nccl') else (0 if world > 1 else 1),
            'dist_backend': job.backend, 'weak_value': None if weak is None else round(weak, 2),
            'roofline': roof, 'cpu_baseline': extra.pop('cpu_baseline', None), 'parity': extra.pop('parity', None),
            'scorer_images_per_sec': round(n_total * a.steps / dt, 2), 'rows_local': rows_local, **extra,
        }
        print(json.dumps(out), flush=True)


def oracle_rejection32_step(n_total, seed=0):
    """CPU baseline of the 32x32 workload: the oracle's DDPM++ CIFAR-32 Heun step (sigma step 5, S_churn 40) over the same N trajectories
    + the brightness score of the predicted images, fp32 torch-CPU on this box's host cores."""
    import torch
    from diffusion_tts_amd import init as dinit
    from diffusion_tts_amd.config import ddpmpp_cifar10
    from oracle.edm_nets import NetCfg, EDMPrecondOracle
    from oracle import sampler as osamp, scorers as oscore
    cores = min(16, len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1))
    torch.set_num_threads(cores)
    cfg = ddpmpp_cifar10()
    sd, _ = dinit.refill_degenerate(dinit.edm_state_dict(cfg, seed), seed)
    net = EDMPrecondOracle(NetCfg('ddpmpp', 32, 3, 10, 128, [2, 2, 2], 4, 4, [16], 9), sd)
    t_steps = osamp.sigma_schedule(net, 18)
    ctx = osamp._Ctx(net, 18, 40, 0.05, 50, 1.003)
    g = torch.Generator().manual_seed(99)
    x = torch.randn(n_total, 3, 32, 32, generator=g, dtype=torch.float64) * t_steps[5]
    eps = torch.randn(n_total, 3, 32, 32, generator=g, dtype=torch.float64)
    lab = torch.eye(10)[torch.tensor([3])].repeat(n_total, 1)
    scorer = oscore.BrightnessOracle()
    reps, t0 = 0, time.perf_counter()
    while reps < 2 or time.perf_counter() - t0 < 4.0:          # a bounded sample: a few seconds of CPU work
        xn, _ = ctx.heun_step(x, t_steps[5], t_steps[6], 5, eps, lab)
        sc = scorer(osamp.to_uint8(xn), lab, torch.zeros(n_total))
        reps += 1
    dt = time.perf_counter() - t0
    return {'value': round(2 * n_total * reps / dt, 2), 'unit': 'candidate U-Net steps/sec', 'cores': torch.get_num_threads(), 'kind': 'port',
            'seconds': round(dt, 2), 'cpu_model': cpu_model(), 'sample': f'{reps} Heun steps of the same workload ({n_total} trajectories = {2 * n_total} DDPM++-32 denoiser rows each, '
                                               f'+ brightness score), fp32 torch-CPU oracle'}, float(sc.max())


def rejection32_record(a, job, steps=None, warmup=None, cpu=True):
    """BASELINE configs[1]: DDPM++ CIFAR-32, rejection sampling over N=16 trajectories, brightness scorer.  One step = one Heun step
    of the N trajectories (2*N candidate U-Net evaluations) + the score of the predicted images; sigma step 5, S_churn = 40."""
    import torch
    from diffusion_tts_amd import init as dinit, ops
    from diffusion_tts_amd.config import ddpmpp_cifar10
    from diffusion_tts_amd.networks import EDMPrecond
    from diffusion_tts_amd.parallel import CandidateShards
    from diffusion_tts_amd.sampler import _Loop
    from diffusion_tts_amd.scorers import BrightnessScorer
    dtype = torch_dtype(a.dtype)
    world, rank, dev = job.world, job.rank, job.dev
    steps = a.steps if steps is None else steps
    warmup = a.warmup if warmup is None else warmup
    cfg = ddpmpp_cifar10()
    sd, _ = dinit.refill_degenerate(dinit.edm_state_dict(cfg, 0), 0)
    net = EDMPrecond(cfg, sd, device=dev, dtype=dtype)
    scorer = BrightnessScorer()
    shards = CandidateShards()
    cand = 16 if a.workload != 'ddpmpp32_rejection' else a.candidates
    n_total = cand * world if a.scaling == 'weak' else cand
    shards.require_candidates(n_total, 'bench')
    lo, hi = shards.span(n_total)
    nl = hi - lo
    L = _Loop(net, dev, 18, 40, 0.05, 50, 1.003, None, shards)
    t_steps = sigma_steps()
    g = torch.Generator().manual_seed(99)
    x = (torch.randn(n_total, 3, 32, 32, generator=g, dtype=torch.float64) * t_steps[5])[lo:hi].to(dev).contiguous()
    eps = [torch.randn(n_total, 3, 32, 32, generator=g, dtype=torch.float64)[lo:hi].to(dev).contiguous() for _ in range(4)]
    lab = torch.eye(10)[torch.tensor([3])].repeat(nl, 1).to(dev).contiguous()

    def one_step(s):
        xn, _ = L.step(x, t_steps[5], t_steps[6], 5, eps[s % 4], lab, nb=nl)
        loc = L.score(scorer, xn, lab).to(dev, torch.float32)
        return int(shards.gather_rewards(loc, n_total, 1).cpu().argmax())

    for s in range(3):
        one_step(s)
    torch.cuda.synchronize(dev)
    dt = job.timed(one_step, steps, warmup)
    value = 2 * n_total * steps / dt
    roof = None
    if not a.no_kernel_timing:
        if rank == 0:
            roof = conv_roofline(a, lambda s: one_step(s), min(2, steps), a.dtype, value / world, GFLOP_PER_EVAL['ddpmpp32'], [net._graphs])
        else:
            for s in range(min(2, steps)):
                one_step(s)
    job.barrier()
    cpu_rec = None
    if cpu and rank == 0 and world == 1 and not a.no_cpu_baseline:
        cpu_rec, _ = oracle_rejection32_step(n_total)
    return {
        'metric': 'candidate U-Net steps/sec, EDM CIFAR-10 32x32 rejection N=16', 'value': round(value, 2),
        'unit': 'candidate U-Net steps/sec', 'n_gpus': world, 'steps': steps, 'warmup': warmup,
        'ms_per_step': round(dt / steps * 1e3, 3), 'higher_is_better': True, 'scaling': a.scaling if world > 1 else 'strong',
        'vs_baseline': None, 'dtype': a.dtype, 'data': 'synthetic (N(0,1) latents/noises, random-init weights + weight rule)',
        'config': {'workload': 'EDM CIFAR-10 (DDPM++ 55.7M) rejection-sampling Heun step over the N trajectories, brightness scorer, sigma step 5/18',
                   'candidates_total': n_total, 'candidates_per_gpu': nl, 'rows_per_step_total': 2 * n_total,
                   'gflop_per_eval': GFLOP_PER_EVAL['ddpmpp32'], 'parallelism': f'trajectories sharded x{world}'},
        'roofline': roof, 'cpu_baseline': cpu_rec}


def run_rejection32(a, job):
    rec = rejection32_record(a, job)
    if job.rank == 0:
        print(json.dumps(rec), flush=True)


def mcts_record(a, job, S, steps, warmup, built=None):
    """BASELINE configs[4]: ADM-64 MCTS, S rollouts per timestep, imagenet scorer.  One step = one whole image search through
    generate_image_grid (18 sigma steps; node expansions batched, ragged rollouts batched and sharded over the ranks)."""
    import numpy as np
    import torch
    from diffusion_tts_amd.sampler import SamplingMethod, generate_image_grid
    dtype = torch_dtype(a.dtype)
    net, scorer = built if built is not None else build_adm(job, dtype, scorer_name=a.scorer)[:2]
    lat = torch.randn(1, 3, 64, 64, generator=torch.Generator().manual_seed(3))
    lab = torch.eye(1000)[torch.tensor([5])]

    def search(S_, seed):
        np.random.seed(seed)
        return generate_image_grid(net, None, lat, lab, seed=seed, gridw=1, gridh=1, device=job.dev, num_steps=18, S_churn=40,
                                   S_min=0.05, S_max=50, S_noise=1.003, sampling_method=SamplingMethod.MCTS,
                                   sampling_params=dict(scorer=scorer, N=a.mcts_children, S=S_), compute_dtype=dtype, verbose=False)
    search(min(S, 16), 0)                       # setup: kernel attributes, graph captures of the common batch sizes
    rows = [0]

    def one_step(s):
        rows[0] += search(S, 1 + s)['net_rows']
    for s in range(warmup):
        one_step(s)
    rows[0] = 0
    dt = job.timed(one_step, steps, 0)
    rows_total = rows[0]                        # rows of THIS rank; rollouts are sharded, expansions replicated
    if job.world > 1:
        t = torch.tensor([rows_total], dtype=torch.float64, device=job.dev if job.backend == 'nccl' else 'cpu')
        job.dist.all_reduce(t)
        rows_total = int(t)
    value = rows_total / dt
    return {
        'metric': f'candidate U-Net steps/sec, EDM ImageNet-64 MCTS S={S}', 'value': round(value, 2),
        'unit': 'candidate U-Net steps/sec', 'n_gpus': job.world, 'steps': steps, 'warmup': warmup,
        'ms_per_step': round(dt / steps * 1e3, 1), 'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None,
        'dtype': a.dtype, 'data': 'synthetic (N(0,1) latents/noises, random-init weights + weight rule)',
        'config': {'workload': f'EDM ImageNet-64 (ADM 295.9M) MCTS search of one image: 18 sigma steps, {a.mcts_children} children per node, '
                               f'S={S} rollouts per timestep, imagenet scorer', 'denoiser_rows_per_search': rows_total // max(1, steps),
                   'parallelism': f'rollouts of each group of 16 sharded x{job.world}'},
        'roofline': {'bound': 'mfma', 'achieved': round(value / job.world * GFLOP_PER_EVAL['adm64'] / 1e3, 1), 'peak': PEAK_TFLOPS[a.dtype],
                     'unit': 'TFLOP/s', 'frac': round(value / job.world * GFLOP_PER_EVAL['adm64'] * 1e9 / (PEAK_TFLOPS[a.dtype] * 1e12), 4),
                     'traffic': None, 'kernel': 'whole search (denoiser FLOPs only)'},
        'cpu_baseline': None}


def run_mcts(a, job):
    rec = mcts_record(a, job, a.S, a.steps, a.warmup)
    if job.rank == 0:
        print(json.dumps(rec), flush=True)


def main():
    os.environ.setdefault('DTS_GRAPHS_STRICT', '1')      # a refused HIP-graph capture must fail the measurement, not fall back to eager launches
    a = parse()
    if a.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(launch_ranks(a.gpus))          # before any GPU / HIP call in this process
    job = Job(a)
    {'adm64_eps_greedy': run_eps_greedy, 'ddpmpp32_rejection': run_rejection32, 'adm64_mcts': run_mcts}[a.workload](a, job)
    job.finish()


if __name__ == '__main__':
    main()
