#!/usr/bin/env python3
"""Where an MCTS search (BASELINE configs[4] shape, S given) spends its time: every _Loop.step / _Loop.score call timed with a device
synchronisation around it, grouped by batch rows.    python tools/mcts_profile.py --S 64"""
import argparse, os, sys, time
from collections import defaultdict
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from diffusion_tts_amd import sampler as sm

ap = argparse.ArgumentParser()
ap.add_argument('--S', type=int, default=64)
a = ap.parse_args()
sys.argv = [sys.argv[0]]
ba = bench.parse()
job = bench.Job(ba)
net, scorer, _ = bench.build_adm(job, torch.bfloat16, scorer_name='imagenet')
acc = defaultdict(lambda: [0, 0.0])
orig_step, orig_score = sm._Loop.step, sm._Loop.score


def step(self, x_cur, t_cur, t_next, i, eps, labels, nb=None, interleave=False):
    rows = eps.shape[0] if nb is None else nb
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = orig_step(self, x_cur, t_cur, t_next, i, eps, labels, nb, interleave)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    k = ('step', rows, 2 if i < self.num_steps - 1 else 1)
    acc[k][0] += 1; acc[k][1] += dt
    return out


def score(self, scorer_, x, labels):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = orig_score(self, scorer_, x, labels)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    k = ('score', x.shape[0], 1)
    acc[k][0] += 1; acc[k][1] += dt
    return out


lat = torch.randn(1, 3, 64, 64, generator=torch.Generator().manual_seed(3))
lab = torch.eye(1000)[torch.tensor([5])]


def search(S, seed):
    np.random.seed(seed)
    return sm.generate_image_grid(net, None, lat, lab, seed=seed, gridw=1, gridh=1, device=job.dev, num_steps=18, S_churn=40, S_min=0.05, S_max=50,
                                  S_noise=1.003, sampling_method=sm.SamplingMethod.MCTS, sampling_params=dict(scorer=scorer, N=4, S=S),
                                  compute_dtype=torch.bfloat16, verbose=False)


search(16, 0)
search(a.S, 1)
sm._Loop.step, sm._Loop.score = step, score
torch.cuda.synchronize(); t0 = time.perf_counter()
res = search(a.S, 2)
torch.cuda.synchronize(); tot = time.perf_counter() - t0
print(f'S={a.S}: {tot:.2f} s (with a sync around every call), {res["net_rows"]} denoiser rows; graphs: denoiser {net._graphs.captures} captures {net._graphs.replays} replays')
tt = 0.0
for k in sorted(acc):
    n, t = acc[k]
    tt += t
    print(f'  {k[0]:5s} rows {k[1]:3d} forwards/call {k[2]}: {n:5d} calls, {t:7.2f} s, {t / n * 1e3:7.2f} ms per call')
print(f'  device calls {tt:.2f} s, host-side remainder {tot - tt:.2f} s')
