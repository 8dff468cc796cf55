#!/usr/bin/env python3
"""BASELINE.json configs[0] and configs[1] AT FULL SIZE through THE REFERENCE ITSELF (edm/main.py generate_image_grid on the CPU of the build
container, the 128-wide DDPM++ CIFAR-32 denoiser under the weight rule):
  configs[0]  NAIVE sampler, 18 Heun steps, S_churn 40           (edm/main.py:862-866): 35 denoiser rows
  configs[1]  REJECTION_SAMPLING, N = 16, brightness scorer      (edm/main.py:101-137): 560 denoiser rows, one scorer call of 16 (+ the final one)

Run:  PYTHONHASHSEED=0 python tests/golden/make_golden_configs01.py      (needs /root/reference; about a minute)

Writes tests/golden/configs01_golden.npz + configs01_manifest.json: the final states (= the last denoiser call's output: the last step is an Euler
step to sigma 0), the uint8 images, the 16 rewards of the rejection step and the trajectory the reference kept -- read off its own final state
(which row of the last 16-row denoiser output its `x_next` is), not re-derived from the rewards.  Inputs are the latents / labels of
tests/test_gpu_fullsize.py::test_baseline_config1 / config2 (torch.Generator seeds 0 / 1, labels 3 / 7); weights re-created by diffusion_tts_amd.init
and pinned by the checksum in fullsize_manifest.json.  Nothing of the reference's text is stored."""
import json
import os
import sys
import time

assert os.environ.get('PYTHONHASHSEED') == '0', 'run with PYTHONHASHSEED=0'

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg                       # noqa: E402
import make_golden_fullsize as mf              # noqa: E402

import numpy as np                             # noqa: E402
import torch                                   # noqa: E402

from diffusion_tts_amd.config import ddpmpp_cifar10  # noqa: E402

KW = dict(num_steps=18, seed=0)


def main():
    torch.set_num_threads(max(1, min(8, os.cpu_count() or 1)))
    t00 = time.time()
    with open(os.path.join(HERE, 'fullsize_manifest.json')) as f:
        fman = json.load(f)
    net, ck = mf.ref_full(ddpmpp_cifar10(), mg.NET_SEED)
    a_, b_ = ck['checksum'], fman['ddpmpp_cifar10']['checksum']
    assert a_['numel'] == b_['numel'] and abs(a_['abs_sum'] - b_['abs_sum']) <= 1e-12 * b_['abs_sum'], (a_, b_)
    out, man = {}, dict(ddpmpp_cifar10=ck, torch=torch.__version__, numpy=np.__version__, threads=torch.get_num_threads(),
                        S=dict(S_churn=40, S_min=0.05, S_max=50, S_noise=1.003), **KW)
    bright = mg.ref_scorers.BrightnessScorer()
    # ---- configs[0]: naive
    lat = torch.randn(1, 3, 32, 32, generator=torch.Generator().manual_seed(0))
    lab = torch.eye(10)[torch.tensor([3])]
    with torch.no_grad():
        lg, sl, png, err = mg.run_ref_search(net, bright, lat, lab, 'NAIVE', {}, KW['num_steps'], seed=KW['seed'])
    assert err is None, err
    rows = int(sum(c[0].shape[0] for c in lg.calls))
    assert rows == 35 and len(sl.calls) == 1
    out.update(naive_latents=lat.numpy(), naive_x_final=lg.calls[-1][2].double().numpy(), naive_image=png, naive_final_score=sl.calls[-1][1].numpy())
    man['naive'] = dict(label=3, latent_seed=0, net_rows=rows, scorer_calls=len(sl.calls))
    print(f'[{time.time() - t00:6.1f}s] configs[0] naive: {rows} rows, final score {float(sl.calls[-1][1][0]):.6f}', flush=True)
    # ---- configs[1]: rejection N = 16, brightness
    lat = torch.randn(1, 3, 32, 32, generator=torch.Generator().manual_seed(1))
    lab = torch.eye(10)[torch.tensor([7])]
    with torch.no_grad():
        lg, sl, png, err = mg.run_ref_search(net, bright, lat, lab, 'REJECTION_SAMPLING', dict(N=16), KW['num_steps'], seed=KW['seed'])
    assert err is None, err
    rows = int(sum(c[0].shape[0] for c in lg.calls))
    assert rows == 16 * 35 and len(sl.calls) == 2 and sl.calls[0][1].shape[0] == 16
    rew = sl.calls[0][1].numpy()
    last = lg.calls[-1][2].double()                                         # [16, 3, 32, 32]: the 16 trajectories' final states
    # the trajectory the reference kept: its final image is the quantisation of ONE of these rows (edm/main.py:136-137, 869)
    q = (last * 127.5 + 128).clip(0, 255).to(torch.uint8).permute(0, 2, 3, 1).numpy()
    kept = [j for j in range(16) if np.array_equal(q[j], png)]
    assert len(kept) >= 1, 'the final image is none of the 16 trajectories'
    srt = np.sort(rew)[::-1]
    out.update(rej_latents=lat.numpy(), rej_rewards=rew, rej_kept=np.array(kept[:1], dtype=np.int64), rej_x_final=last[kept[0]:kept[0] + 1].numpy(),
               rej_image=png, rej_final_score=sl.calls[-1][1].numpy())
    man['rejection'] = dict(label=7, latent_seed=1, params=dict(N=16), net_rows=rows, scorer_calls=len(sl.calls), kept=int(kept[0]), rows_with_that_image=len(kept),
                            argmax_of_rewards=int(rew.argmax()), top2_gap=float(srt[0] - srt[1]),
                            kept_source='the trajectory whose quantised final state is the reference run\'s PNG')
    print(f'[{time.time() - t00:6.1f}s] configs[1] rejection: {rows} rows, kept trajectory {kept[0]} (argmax of the rewards {int(rew.argmax())}, top-2 gap {srt[0] - srt[1]:.3e})', flush=True)
    np.savez_compressed(os.path.join(HERE, 'configs01_golden.npz'), **out)
    with open(os.path.join(HERE, 'configs01_manifest.json'), 'w') as f:
        json.dump(man, f, indent=1)
    print(f'[{time.time() - t00:6.1f}s] wrote configs01_golden.npz')


if __name__ == '__main__':
    main()
