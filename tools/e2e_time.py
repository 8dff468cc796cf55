#!/usr/bin/env python3
"""End-to-end wall time of BASELINE config 3 through generate_image_grid (host RNG + uploads included)."""
import sys, os, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusion_tts_amd.sampler import SamplingMethod, generate_image_grid, load_network
from diffusion_tts_amd.scorers import ImageNetScorer
dev = torch.device('cuda')
net = load_network('random:adm_imagenet64', device=dev, dtype=torch.bfloat16)
with warnings.catch_warnings():
    warnings.simplefilter('ignore')
    sc = ImageNetScorer(device=dev)
lat = torch.randn(1, 3, 64, 64); lab = torch.eye(1000)[torch.tensor([5])]
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 18
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    res = generate_image_grid(net, None, lat, lab, seed=0, gridw=1, gridh=1, device=dev, num_steps=steps, S_churn=40, S_min=0.05,
                              S_max=50, S_noise=1.003, sampling_method=SamplingMethod.EPS_GREEDY,
                              sampling_params=dict(scorer=sc, N=64, K=4, lambda_param=0.15, eps=0.4), verbose=False)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f'rep {rep}: {dt:.2f} s for {res["net_rows"]} denoiser rows -> {res["net_rows"]/dt:.0f} candidate evals/s (host RNG + uploads included)', flush=True)
