#!/usr/bin/env python3
"""BASELINE config 4 on one MI355X with this build's parts: SD beam search B=4, N=16 over [N,4,64,64] fp16 latents, candidate-batched 2N-row
U-Net calls (stand-in U-Net: the diffusers U-Net stays an opaque module in this path and is not available offline), the fused DDIM candidate
step, N-row decodes through the HIP VAE decoder (SD-1.5 width, random init) and a random-init CLIP ViT-L/14-shaped scorer.  Reports decodes/s:
the VAE decode (2.48 TFLOP per candidate) is the largest cost of an SD candidate and the part of config 4 this build owns."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import torch
from diffusion_tts_amd import init as dinit
from diffusion_tts_amd.sd_pipeline import SDSearchPipeline
from diffusion_tts_amd.scorers import CLIPScorer, BrightnessScorer, ByteTokenizer
from diffusion_tts_amd.vae import VAEDecoder
from sd_standins import shape_unet, TinyTextEncoder, TinyTokenizer, tiny_clip

ap = argparse.ArgumentParser()
ap.add_argument('--steps', type=int, default=4)
ap.add_argument('--B', type=int, default=4)
ap.add_argument('--N', type=int, default=16)
ap.add_argument('--scorer', default='clip', choices=['clip', 'brightness'])
a = ap.parse_args()
dev = 'cuda'
dec = VAEDecoder(dinit.vae_decoder_state_dict(seed=5), device=dev, dtype=torch.float16)
unet, te = shape_unet().half().to(dev), TinyTextEncoder().half().to(dev)
pipe = SDSearchPipeline(unet, dec, device=dev, text_encoder=te, tokenizer=TinyTokenizer())
scorer = CLIPScorer(model=tiny_clip(0), tokenizer=ByteTokenizer(1000, 998, 999), device=dev) if a.scorer == 'clip' else BrightnessScorer()
best = None
for rep in range(3):
    torch.manual_seed(7)
    lat = torch.randn(1, 4, 64, 64).half()
    d0 = dec.decodes
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out, score = pipe(prompt='a photo of a cat', latents=lat, num_inference_steps=a.steps, score_function=scorer, method='beam',
                      params={'N': a.N, 'B': a.B, 'K': 20, 'lambda': 0.15, 'eps': 0.4, 'S': 8}, output_type='pt')
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    best = dt if best is None else min(best, dt)
    nd = dec.decodes - d0
print(f'SD beam B={a.B} N={a.N}, {a.steps} DDIM steps, {a.scorer} scorer: {best:.3f} s = {nd / best:.1f} candidate decodes/s '
      f'({nd * 2.48 / best:.0f} TFLOP/s of VAE work incl. the loop, the stand-in U-Net and the scorer); {out.unet_rows} U-Net rows, {len(out.scores)} scorer calls')
