#!/usr/bin/env python3
"""Where does an MCTS search (BASELINE configs[4]) spend its time?  Wraps the sampler's device step and scorer with synchronising timers
and prints, per batch size, the number of Heun steps, their total and mean time, the scorer's share and what is left for the host (tree
walk, UCB, tensor concatenation, reward read-back).  The timers serialise host and device, so the sum is an upper bound of the search.
    python tools/mcts_breakdown.py [--S 64] [--dtype f16x3]"""
import argparse, collections, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--S', type=int, default=64)
    ap.add_argument('--dtype', default='f16x3')
    a = ap.parse_args()
    os.environ.setdefault('DTS_GRAPHS_STRICT', '1')
    import numpy as np
    import torch
    import bench
    from diffusion_tts_amd import sampler
    from diffusion_tts_amd.sampler import SamplingMethod, generate_image_grid
    ba = bench.parse(['--workload', 'adm64_mcts', '--dtype', a.dtype])
    job = bench.Job(ba)
    dtype = bench.torch_dtype(a.dtype)
    net, scorer = bench.build_adm(job, dtype, scorer_name=ba.scorer)[:2]
    lat = torch.randn(1, 3, 64, 64, generator=torch.Generator().manual_seed(3))
    lab = torch.eye(1000)[torch.tensor([5])]

    def search(S_, seed):
        np.random.seed(seed)
        return generate_image_grid(net, None, lat, lab, seed=seed, gridw=1, gridh=1, device=job.dev, num_steps=18, S_churn=40, S_min=0.05, S_max=50,
                                   S_noise=1.003, sampling_method=SamplingMethod.MCTS, sampling_params=dict(scorer=scorer, N=4, S=S_),
                                   compute_dtype=dtype, verbose=False)
    search(min(a.S, 16), 0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = search(a.S, 1)
    torch.cuda.synchronize()
    free = time.perf_counter() - t0
    print(f'free-running search: {free:.2f} s, {r["net_rows"]} rows -> {r["net_rows"] / free:.1f} evals/s', flush=True)

    tally = collections.defaultdict(lambda: [0, 0.0])
    step0, score0 = sampler._Loop.step, sampler._Loop.score

    def step(self, x_cur, t_cur, t_next, i, eps, labels, nb=None, interleave=False, live=None):
        rows = eps.shape[0] if nb is None else nb
        torch.cuda.synchronize()
        t = time.perf_counter()
        out = step0(self, x_cur, t_cur, t_next, i, eps, labels, nb=nb, interleave=interleave, live=live)
        torch.cuda.synchronize()
        k = ('step', rows, 2 if i < self.num_steps - 1 else 1)
        tally[k][0] += 1
        tally[k][1] += time.perf_counter() - t
        return out

    def score(self, scorer_, x, labels):
        torch.cuda.synchronize()
        t = time.perf_counter()
        out = score0(self, scorer_, x, labels)
        torch.cuda.synchronize()
        k = ('score', x.shape[0], 0)
        tally[k][0] += 1
        tally[k][1] += time.perf_counter() - t
        return out
    sampler._Loop.step, sampler._Loop.score = step, score
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = search(a.S, 1)
    torch.cuda.synchronize()
    total = time.perf_counter() - t0
    dev = 0.0
    for k in sorted(tally):
        n, t = tally[k]
        dev += t
        print(f'{k[0]:5s} rows {k[1]:3d} forwards/step {k[2]}: {n:5d} calls, {t:7.3f} s total, {t / n * 1e3:7.2f} ms each', flush=True)
    print(f'timed search {total:.2f} s: device sections {dev:.2f} s, host remainder {total - dev:.2f} s', flush=True)


main()
