#!/usr/bin/env python3
"""Pick the seed of the full config-3 reference golden (tests/golden/make_golden_config3.py).  A free-running reference search costs ~2 h of
CPU, so its seed is chosen here first, on the GPU (6 s per search): for each seed, run BASELINE configs[2] end to end (eps-greedy N = 64,
K = 4, 18 sigma steps, the fixture's latents / label / weights) in the split-precision mode and list the top-2 reward gap of every decision.
A seed is usable when every decision is either an exact tie (no churn noise: all candidates identical, first-max rule) or has a gap well
above the fp32 reward noise (~1e-8) -- a gap below that is a coin flip between two correct fp32 implementations, after which a free-running
comparison compares two different trajectories.  The chosen seed is then confirmed in the f32 parity mode.
    python tools/seed_scan.py --seeds 1-16 [--confirm SEED]"""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--seeds', default='1-16')
    ap.add_argument('--confirm', type=int, default=None, help='also run this seed in the f32 parity mode and compare selections')
    ap.add_argument('--out', default=None)
    a = ap.parse_args()
    os.environ.setdefault('DTS_GRAPHS_STRICT', '1')
    import numpy as np
    import torch
    from helpers import full_weights
    from diffusion_tts_amd import ops, sampler as sm, scorers as S
    from diffusion_tts_amd.hashing import seed0_scale
    from diffusion_tts_amd.networks import EDMPrecond
    gold = np.load(os.path.join(ROOT, 'tests', 'golden', 'fullsize_golden.npz'))
    with open(os.path.join(ROOT, 'tests', 'golden', 'fullsize_manifest.json')) as f:
        man = json.load(f)
    cfg, sd = full_weights(man, 'adm_imagenet64')
    ccfg, csd = full_weights(man, 'cls_imagenet64')
    lat = torch.from_numpy(gold['eg64_latents'])
    lab = torch.eye(1000)[torch.from_numpy(gold['eg64_label_idx']).long()]
    params = dict(N=64, K=4, lambda_param=0.15, eps=0.4)

    def run(dtype, seed):
        net = run.nets.get(dtype)
        if net is None:
            net = run.nets[dtype] = (EDMPrecond(cfg, sd, device='cuda', dtype=dtype),
                                     S.ImageNetScorer(weights=csd, cfg=ccfg, device='cuda', compute_dtype=dtype))
        h = sm.generate_image_grid(net[0], None, lat, lab, seed=seed, gridw=1, gridh=1, device=torch.device('cuda'), num_steps=18, S_churn=40,
                                   S_min=0.05, S_max=50, S_noise=1.003, sampling_method=sm.SamplingMethod.EPS_GREEDY,
                                   sampling_params=dict(scorer=net[1], **params), scale_fn=seed0_scale, compute_dtype=dtype, verbose=False)
        rew = [r.reshape(-1).double().numpy() for r in h['rewards']]
        gaps = [float(np.sort(r)[::-1][0] - np.sort(r)[::-1][1]) for r in rew]
        return [int(s_[0]) for s_ in h['selected']], gaps, rew, h
    run.nets = {}
    lo, hi = (int(v) for v in a.seeds.split('-'))
    table = {}
    for seed in range(lo, hi + 1):
        sel, gaps, _, h = run(ops.F16X3, seed)
        nz = [g for g in gaps if g > 0]
        table[seed] = dict(min_nonzero_gap=min(nz), ties=len(gaps) - len(nz), decisions=len(gaps), below_5e8=sum(g < 5e-8 for g in nz), selected=sel,
                           net_rows=int(h['net_rows']))
        print(f'seed {seed:3d}: {len(gaps)} decisions, {len(gaps) - len(nz)} exact ties, min non-zero top-2 gap {min(nz):.3e}, '
              f'{sum(g < 5e-8 for g in nz)} below 5e-8, {sum(g < 2e-7 for g in nz)} below 2e-7', flush=True)
    best = max(table, key=lambda s_: table[s_]['min_nonzero_gap'])
    print(f'best seed {best}: {table[best]}', flush=True)
    if a.confirm is not None or True:
        c = best if a.confirm is None else a.confirm
        sel3, gaps3, rew3, _ = run(ops.F16X3, c)
        sel32, gaps32, rew32, _ = run(torch.float32, c)
        dev = max(float(np.abs(x - y).max()) for x, y in zip(rew3, rew32))
        print(f'seed {c}: f32 parity mode makes the same {len(sel32)} selections: {sel3 == sel32}; max reward deviation {dev:.2e}; '
              f'f32 min non-zero gap {min(g for g in gaps32 if g > 0):.3e}', flush=True)
        table['confirmed'] = dict(seed=c, same_selections=sel3 == sel32, max_reward_dev=dev, gaps_f32=gaps32, selected_f32=sel32)
    if a.out:
        with open(a.out, 'w') as f:
            json.dump(table, f, indent=1)


main()
