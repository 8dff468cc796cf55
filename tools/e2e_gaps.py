#!/usr/bin/env python3
"""Where the host spends the time between the reward sync of one eps-greedy iteration and the first launches of the next
(BASELINE config 3 through generate_image_grid).  GPU box only."""
import sys, os, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusion_tts_amd import sampler as sm, ops
from diffusion_tts_amd.scorers import ImageNetScorer
dev = torch.device('cuda')
net = sm.load_network('random:adm_imagenet64', device=dev, dtype=torch.bfloat16)
with warnings.catch_warnings():
    warnings.simplefilter('ignore')
    sc = ImageNetScorer(device=dev)
lat = torch.randn(1, 3, 64, 64); lab = torch.eye(1000)[torch.tensor([5])]
T = {'sync_wait': 0.0, 'turnaround': 0.0, 'enqueue': 0.0, 'prefetch': 0.0, 'n': 0}
mark = {'t': None}
orig_gather = sm.CandidateShards.gather_rewards
orig_step = sm._Loop.step
orig_prefetch = sm._Lookahead.prefetch

def gather(self, local, n, rows):
    t0 = time.perf_counter()
    torch.cuda.synchronize()                 # what scores.cpu() would wait for
    T['sync_wait'] += time.perf_counter() - t0
    mark['t'] = time.perf_counter()
    return orig_gather(self, local, n, rows)

def step(self, *a, **k):
    if mark['t'] is not None:
        T['turnaround'] += time.perf_counter() - mark['t']; T['n'] += 1; mark['t'] = None
    t0 = time.perf_counter()
    r = orig_step(self, *a, **k)
    T['enqueue'] += time.perf_counter() - t0
    return r

def prefetch(self):
    t0 = time.perf_counter()
    orig_prefetch(self)
    T['prefetch'] += time.perf_counter() - t0

sm.CandidateShards.gather_rewards = gather
sm._Loop.step = step
sm._Lookahead.prefetch = prefetch
for rep in range(2):
    for k in T: T[k] = 0
    torch.cuda.synchronize(); t0 = time.perf_counter()
    res = sm.generate_image_grid(net, None, lat, lab, seed=0, gridw=1, gridh=1, device=dev, num_steps=18, S_churn=40, S_min=0.05,
                                 S_max=50, S_noise=1.003, sampling_method=sm.SamplingMethod.EPS_GREEDY,
                                 sampling_params=dict(scorer=sc, N=64, K=4, lambda_param=0.15, eps=0.4), verbose=False)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    n = max(T['n'], 1)
    print(f'rep {rep}: {dt:.2f} s; per iteration: host waits for the GPU {T["sync_wait"]/n*1e3:.2f} ms, sync->next step() call '
          f'{T["turnaround"]/n*1e3:.2f} ms, step() enqueue {T["enqueue"]/n*1e3:.2f} ms, host draws (prefetch) {T["prefetch"]/n*1e3:.2f} ms', flush=True)
