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
ap.add_argument('--host-preprocess', action='store_true', help="CLIP scorer: the reference's host path (images.cpu() + PIL) instead of the device kernels")
ap.add_argument('--gpus', type=int, default=1, help='ranks (candidates of every timestep sharded; DTS_DIST_BACKEND=gloo lets ranks share one GPU)')
a = ap.parse_args()
if a.gpus > 1 and 'WORLD_SIZE' not in os.environ:          # start the ranks as children before anything touches the GPU
    import socket, subprocess
    with socket.socket() as so:
        so.bind(('127.0.0.1', 0))
        port = so.getsockname()[1]
    sys.exit(subprocess.call([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={a.gpus}', '--master-addr', '127.0.0.1',
                              '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]))
world, rank = int(os.environ.get('WORLD_SIZE', '1')), int(os.environ.get('RANK', '0'))
if world > 1:
    import torch.distributed as dist
    backend = os.environ.get('DTS_DIST_BACKEND', 'nccl')
    ndev = max(1, torch.cuda.device_count())
    torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', '0')) % ndev)
    if backend == 'nccl':
        dist.init_process_group('nccl', device_id=torch.device('cuda', torch.cuda.current_device()))
    else:
        dist.init_process_group(backend)
dev = torch.device('cuda', torch.cuda.current_device())
dec = VAEDecoder(dinit.vae_decoder_state_dict(seed=5), device=dev, dtype=torch.float16)
unet, te = shape_unet().half().to(dev), TinyTextEncoder().half().to(dev)
pipe = SDSearchPipeline(unet, dec, device=dev, text_encoder=te, tokenizer=TinyTokenizer())
scorer = (CLIPScorer(model=tiny_clip(0), tokenizer=ByteTokenizer(1000, 998, 999), device=dev, device_preprocess=not a.host_preprocess)
          if a.scorer == 'clip' else BrightnessScorer())
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
if world > 1:
    t = torch.tensor([float(nd), best], dtype=torch.float64)
    both = [torch.zeros(2, dtype=torch.float64) for _ in range(world)]
    dist.all_gather(both, t) if backend != 'nccl' else dist.all_gather(both := [b.to(dev) for b in both], t.to(dev))
    nd, best = int(sum(float(b[0]) for b in both)), max(float(b[1]) for b in both)
    dist.destroy_process_group()
if rank == 0:
    pre = '' if a.scorer != 'clip' else (', host PIL pre-processing' if a.host_preprocess else f', device pre-processing ({scorer.device_preprocessed} images)')
    print(f'SD beam B={a.B} N={a.N}, {a.steps} DDIM steps, {a.scorer} scorer{pre}, {world} rank(s): {best:.3f} s = {nd / best:.1f} candidate decodes/s '
          f'({nd * 2.48 / best:.0f} TFLOP/s of VAE work incl. the loop, the stand-in U-Net and the scorer); {out.unet_rows} U-Net rows per rank, '
          f'{len(out.scores)} scores, {out.scorer_calls} scorer calls, {out.collectives} reward collectives')
