#!/usr/bin/env python3
"""Throughput of the HIP SD-VAE decoder at config-4 sizes: N latents [N,4,64,64] -> images [N,3,512,512] (random-init weights of
SD-1.5's decoder shape).  2.48 TFLOP per decode.  GPU box only."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusion_tts_amd import init as dinit
from diffusion_tts_amd.vae import VAEDecoder

ap = argparse.ArgumentParser()
ap.add_argument('--n', type=int, default=16)
ap.add_argument('--dtype', default='f16')
ap.add_argument('--iters', type=int, default=5)
a = ap.parse_args()
dt = {'f16': torch.float16, 'bf16': torch.bfloat16}[a.dtype]
dec = VAEDecoder(dinit.vae_decoder_state_dict(seed=0), device='cuda', dtype=dt)
z = torch.randn(a.n, 4, 64, 64, device='cuda')
for _ in range(2):
    dec.decode(z)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.iters):
    img = dec.decode(z)[0]
torch.cuda.synchronize()
dt_s = (time.perf_counter() - t0) / a.iters
print(f'VAE decode N={a.n} {a.dtype}: {dt_s * 1e3:.1f} ms per batch = {a.n / dt_s:.1f} decodes/s = {a.n * 2.48 / dt_s:.0f} TFLOP/s '
      f'({a.n * 2.48 / dt_s / 2500 * 100:.1f} % of the 2.5 PFLOP/s dense MFMA peak); output {tuple(img.shape)} {img.dtype}')
