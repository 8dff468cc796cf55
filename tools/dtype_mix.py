#!/usr/bin/env python3
"""Which network's precision decides the reward error of the 16-bit modes: the denoiser's or the classifier's?  Eight eps-greedy
iterations of the bench workload (64 candidates, random-init weights) in every (denoiser dtype, scorer dtype) combination against the
all-f32 run: largest reward deviation, index agreement, reward given up by differing picks."""
import argparse, os, sys, types
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
import bench

a = types.SimpleNamespace(gpus=1)
job = bench.Job(a)
DT = {'f32': torch.float32, 'f16': torch.float16, 'bf16': torch.bfloat16}
nets, scorers, sd = {}, {}, None
for k, dt in DT.items():
    n_, s_, sd = bench.build_adm(job, dt, sd=sd)
    nets[k], scorers[k] = n_, s_
steps_i = [2, 3, 5, 7, 9, 11, 13, 14]
ref = []
for r, i_step in enumerate(steps_i):
    it = bench.EpsGreedyIteration(job, nets['f32'], scorers['f32'], 64, i_step=i_step, sets=1, seed=4321 + r)
    b = it(0)
    ref.append((b, it.last_scores.flatten().double().cpu()))
gaps = [float((lambda s: s[0] - s[1])(torch.sort(sc, descending=True).values)) for _, sc in ref]
print('f32 top-2 gaps', [f'{g:.2e}' for g in gaps])
for dn in ('bf16', 'f16', 'f32'):
    for sn in ('bf16', 'f16', 'f32'):
        if dn == sn == 'f32':
            continue
        dev, agree, regret = 0.0, 0, []
        for r, i_step in enumerate(steps_i):
            it = bench.EpsGreedyIteration(job, nets[dn], scorers[sn], 64, i_step=i_step, sets=1, seed=4321 + r)
            b = it(0)
            sc = it.last_scores.flatten().double().cpu()
            dev = max(dev, float((sc - ref[r][1]).abs().max()))
            agree += int(b == ref[r][0])
            regret.append(float(ref[r][1][ref[r][0]] - ref[r][1][b]))
        print(f'denoiser {dn:5s} scorer {sn:5s}: max reward deviation {dev:.2e}  agreement {agree}/8  reward given up {[f"{x:.1e}" for x in regret if x > 0]}', flush=True)
