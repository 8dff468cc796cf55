"""Unified CLI of the MI355X-native noise-trajectory-search path -- same flags as the reference's main.py:84-97.

    python main.py --backend edm --scorer imagenet --method eps_greedy --N 64 --K 4
    python main.py --backend edm --scorer brightness --method rejection --N 16 --network random:ddpmpp_cifar10

Arguments (verbatim from the reference):
    --backend   : 'sd' or 'edm' (required)
    --scorer    : 'brightness', 'compressibility', 'clip', or 'imagenet' (required)
    --method    : naive | rejection | beam | mcts | zero_order | eps_greedy (default naive)
    --prompt, --output, --N, --lambda_, --eps, --K, --B, --S, --seed, --device
Additions (the reference hard-codes a checkpoint URL, main.py:157-158; there is no network here):
    --network   : 'random:adm_imagenet64[:seed]' (default), 'random:ddpmpp_cifar10[:seed]', the local path of an NVIDIA EDM
                  network pickle (*.pkl, read without executing its embedded source), or a .pt bundle
    --dtype     : f16x3 (default: split precision on the 16-bit matrix cores -- the reference's fp32 rewards and therefore its selected
                  candidates on the same seed, ~2.6x the f32 mode's speed) | f32 (parity mode, f32 matrix instruction) | bf16 | f16
                  (throughput modes: ~2.8x faster again, but a near-tied pick differs after ~10 decisions and the result is another sample)
    --seeds LIST --outdir DIR [--subdirs] [--class N]: bulk mode (flags of the reference's edm/generate.py): one search per
                  seed, <outdir>/<seed:06d>.png; with torch.distributed.run the SEEDS are split over the ranks (no collective)
Multi-GPU: launch with `python -m torch.distributed.run --nproc-per-node N main.py ...`; the N candidates of every
search iteration are sharded across the ranks (diffusion_tts_amd/parallel.py).
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def get_scorer(backend, scorer_name, device, compute_dtype=None):
    """main.py:60-71 of the reference.  compute_dtype: the search's --dtype; the ImageNet classifier runs in the search's own mode for the two parity
    modes (float32, f16x3) and in float16 otherwise (also beside a bfloat16 denoiser: scorers.ImageNetScorer)."""
    from diffusion_tts_amd import scorers as S
    if scorer_name == 'brightness':
        return S.BrightnessScorer(dtype=torch.float32)
    if scorer_name == 'compressibility':                               # sd/scorers.py:79 normalises by 150000 bytes, edm/scorers.py:177 by 3000
        return S.CompressibilityScorer(dtype=torch.float32, max_size=150000 if backend == 'sd' else 3000)
    if scorer_name == 'imagenet' and backend == 'edm':
        return S.ImageNetScorer(dtype=torch.float32, device=device, compute_dtype=compute_dtype if compute_dtype in (torch.float32, 'f16x3') else torch.float16)
    if scorer_name == 'clip' and backend == 'sd':
        return S.CLIPScorer(dtype=torch.float32, device=device)        # local HF cache only; raises with instructions otherwise
    raise ValueError(f"Unknown or invalid scorer '{scorer_name}' for backend '{backend}'")


def load_sd_vae(model_id, dev, kind='hip'):
    """The SD VAE of the search loop.  kind='hip' (default): this build's HIP decoder (vae.VAEDecoder, drop-in for `vae.decode`) read from
    the safetensors `vae/` directory of the locally cached SD-1.5 snapshot (or $DTS_SD_VAE_DIR) -- and an error when that directory cannot
    be found or read: there is no silent fallback.  kind='diffusers' (`--vae diffusers`): the stock AutoencoderKL module, chosen explicitly."""
    if kind == 'diffusers':
        from diffusers import AutoencoderKL
        return AutoencoderKL.from_pretrained(model_id, subfolder='vae', torch_dtype=torch.float16, local_files_only=True).to(dev)
    if kind != 'hip':
        raise ValueError(f"--vae must be 'hip' or 'diffusers', got {kind!r}")
    from diffusion_tts_amd.vae import VAEDecoder
    path = os.environ.get('DTS_SD_VAE_DIR')
    if path is None:
        from huggingface_hub import snapshot_download
        path = os.path.join(snapshot_download(model_id, local_files_only=True, allow_patterns=['vae/*']), 'vae')
    if not os.path.isdir(path):
        raise FileNotFoundError(f'SD VAE directory {path!r} not found (set DTS_SD_VAE_DIR, or pass --vae diffusers for the stock module)')
    return VAEDecoder.from_pretrained(path, device=dev, dtype=torch.float16)


def main_sd(args):
    """SD backend (reference main.py:111-147).  The search loop, the fused DDIM candidate step and candidate batching are
    this build's (diffusion_tts_amd/sd_pipeline.py); the U-Net / VAE / text encoder are the stock diffusers / transformers
    modules on PyTorch-ROCm, which must be importable and have SD-1.5 weights in the local HF cache (no network here)."""
    try:
        from diffusers import AutoencoderKL, UNet2DConditionModel          # noqa: F401
        from transformers import CLIPTextModel, CLIPTokenizer              # noqa: F401
    except Exception as e:      # pragma: no cover
        raise RuntimeError('--backend sd needs `diffusers` (U-Net/VAE) and SD-1.5 weights in the local cache; neither is '
                           'available in this image.  Drive diffusion_tts_amd.sd_pipeline.SDSearchPipeline(unet, vae) directly '
                           '(tests/test_gpu_sd.py shows the call).') from e
    from diffusion_tts_amd.sd_pipeline import SDSearchPipeline
    import torch.distributed as dist
    model_id = 'runwayml/stable-diffusion-v1-5'
    world, local = int(os.environ.get('WORLD_SIZE', '1')), int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1:                                 # one process per GPU: the candidates of every search decision are sharded (sd_pipeline.py)
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        torch.cuda.set_device(local)
        dist.init_process_group('nccl', device_id=torch.device('cuda', local))
    dev = torch.device('cuda', local) if world > 1 else torch.device(args.device)
    unet = UNet2DConditionModel.from_pretrained(model_id, subfolder='unet', torch_dtype=torch.float16, local_files_only=True).to(dev)
    vae = load_sd_vae(model_id, dev, getattr(args, 'vae', 'hip'))
    tok = CLIPTokenizer.from_pretrained(model_id, subfolder='tokenizer', local_files_only=True)
    te = CLIPTextModel.from_pretrained(model_id, subfolder='text_encoder', torch_dtype=torch.float16, local_files_only=True).to(dev)

    scorer = get_scorer('sd', args.scorer, dev)
    pipe = SDSearchPipeline(unet, vae, device=dev, text_encoder=te, tokenizer=tok)     # encodes the prompt itself (pipeline...:976-992)
    params = {'N': args.N, 'lambda': args.lambda_, 'eps': args.eps, 'K': args.K, 'B': args.B, 'S': args.S}
    torch.manual_seed(args.seed)                                                       # same host RNG stream on every rank
    best, best_score = None, float('-inf')
    for _ in range(params['N'] if args.method == 'rejection' else 1):          # reference main.py:134
        lat = torch.randn(1, unet.config.in_channels, unet.config.sample_size, unet.config.sample_size)
        out, score = pipe(prompt=args.prompt, latents=lat, num_inference_steps=50, score_function=scorer, method='naive' if args.method == 'rejection' else args.method,
                          params=params, output_type='pil')
        score = float(score.item() if torch.is_tensor(score) else score)
        if score > best_score:
            best, best_score = out, score
    outname = args.output or f'sd_{args.method}_{args.scorer}.png'
    if int(os.environ.get('RANK', '0')) == 0:
        best.images[0].save(outname)
        print(f'\n[SD] Saved: {outname}\nBest score: {best_score}  (VAE: {type(vae).__name__}, reward collectives: {best.collectives})\n')
    if world > 1:
        dist.destroy_process_group()
    return best


def main(argv=None):
    parser = argparse.ArgumentParser(description='Unified Diffusion Image Generator (EDM/SD), MI355X-native hot path',
                                     formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    parser.add_argument('--backend', type=str, choices=['edm', 'sd'], required=True, help='Backend: edm or sd')
    parser.add_argument('--scorer', type=str, choices=['brightness', 'compressibility', 'clip', 'imagenet'], required=True,
                        help='Scorer name')
    parser.add_argument('--method', type=str, default='naive',
                        help='Sampling method (naive, rejection, beam, mcts, zero_order, eps_greedy)')
    parser.add_argument('--prompt', type=str, default='YOUR PROMPT HERE', help='Prompt for SD')
    parser.add_argument('--output', type=str, default=None, help='Output filename (default: auto)')
    parser.add_argument('--N', type=int, default=4, help='Master param N')
    parser.add_argument('--lambda_', type=float, default=0.15, help='Master param lambda')
    parser.add_argument('--eps', type=float, default=0.4, help='Master param eps')
    parser.add_argument('--K', type=int, default=20, help='Master param K')
    parser.add_argument('--B', type=int, default=2, help='Master param B')
    parser.add_argument('--S', type=int, default=8, help='Master param S')
    parser.add_argument('--seed', type=int, default=0, help='Random seed')
    parser.add_argument('--device', type=str, default='cuda', help='Device')
    parser.add_argument('--network', type=str, default='random:adm_imagenet64', help='EDM network spec (see module docstring)')
    parser.add_argument('--dtype', type=str, default='f16x3', choices=['bf16', 'f16', 'f32', 'f16x3'], help='compute mode (see the module docstring)')
    parser.add_argument('--vae', type=str, default='hip', choices=['hip', 'diffusers'],
                        help="SD backend: 'hip' = this build's VAE decoder (an error if its safetensors cannot be read), 'diffusers' = the stock module")
    parser.add_argument('--seeds', type=str, default=None, help='bulk mode: seeds, e.g. 0-63 or 1,2,5-10 (one image per seed)')
    parser.add_argument('--outdir', type=str, default='out', help='bulk mode: output directory')
    parser.add_argument('--subdirs', action='store_true', help='bulk mode: one subdirectory per 1000 seeds')
    parser.add_argument('--class', dest='class_idx', type=int, default=None, help='bulk mode: class label (default: per-seed random)')
    args = parser.parse_args(argv)

    if args.backend == 'sd' and args.scorer == 'imagenet':
        raise ValueError('imagenet scorer is only available for edm backend')
    if args.backend == 'edm' and args.scorer == 'clip':
        raise ValueError('clip scorer is only available for sd backend')
    if args.backend == 'sd':
        return main_sd(args)
    if not str(args.device).startswith('cuda'):
        raise RuntimeError(f"--device {args.device}: this build is the GPU path; the CPU path is the reference itself "
                           f"(its restatement lives in oracle/ as test infrastructure)")

    import torch.distributed as dist
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1:
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        torch.cuda.set_device(local)
        dist.init_process_group('nccl', device_id=torch.device('cuda', local))
    device = torch.device('cuda', local if world > 1 else (torch.device(args.device).index or 0))
    torch.cuda.set_device(device)

    from diffusion_tts_amd.sampler import SamplingMethod, generate_image_grid, load_network
    from diffusion_tts_amd.ops import F16X3
    dtype = {'bf16': torch.bfloat16, 'f16': torch.float16, 'f32': torch.float32, 'f16x3': F16X3}[args.dtype]
    scorer = get_scorer('edm', args.scorer, device, compute_dtype=dtype)
    net = load_network(args.network, device=device, dtype=dtype)
    if args.seeds is not None:                                                        # bulk mode: seeds sharded over the ranks
        from diffusion_tts_amd.bulk import generate_seeds
        mm = {'naive': SamplingMethod.NAIVE, 'rejection': SamplingMethod.REJECTION_SAMPLING, 'beam': SamplingMethod.BEAM_SEARCH,
              'mcts': SamplingMethod.MCTS, 'zero_order': SamplingMethod.ZERO_ORDER, 'eps_greedy': SamplingMethod.EPS_GREEDY}
        sp = {'scorer': scorer}
        if args.method != 'naive':
            sp.update(N=args.N, K=args.K, lambda_param=args.lambda_, eps=args.eps, B=args.B, S=args.S)
        done = generate_seeds(net, args.seeds, args.outdir, sampling_method=mm[args.method], sampling_params=sp,
                              class_idx=args.class_idx, subdirs=args.subdirs, device=device, compute_dtype=dtype,
                              num_steps=18, S_churn=40, S_min=0.05, S_max=50, S_noise=1.003)
        print(f'[EDM] rank {os.environ.get("RANK", "0")}: {len(done)} images -> {args.outdir}')
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return done
    num_images = 1
    r = net.img_resolution
    latents = torch.randn([num_images, net.img_channels, r, r])                       # drawn before seeding, as main.py:161
    class_labels = torch.eye(net.label_dim)[torch.randint(net.label_dim, size=[num_images])] if net.label_dim else None
    if world > 1:                                                                      # replicate the unseeded draws
        latents, class_labels = latents.to(device), None if class_labels is None else class_labels.to(device)
        dist.broadcast(latents, 0)
        if class_labels is not None:
            dist.broadcast(class_labels, 0)
        latents, class_labels = latents.cpu(), None if class_labels is None else class_labels.cpu()
    method_map = {'naive': SamplingMethod.NAIVE, 'rejection': SamplingMethod.REJECTION_SAMPLING,
                  'beam': SamplingMethod.BEAM_SEARCH, 'mcts': SamplingMethod.MCTS,
                  'zero_order': SamplingMethod.ZERO_ORDER, 'eps_greedy': SamplingMethod.EPS_GREEDY}
    if args.method not in method_map:
        raise ValueError(f'Unknown method: {args.method}')
    sampling_params = {'scorer': scorer}
    if args.method in ['rejection', 'zero_order', 'eps_greedy', 'beam', 'mcts']:
        sampling_params.update(N=args.N, K=args.K, lambda_param=args.lambda_, eps=args.eps, B=args.B, S=args.S)
    outname = args.output or f'edm_{args.method}_{args.scorer}.png'
    res = generate_image_grid(net, outname, latents, class_labels, seed=args.seed, gridw=1, gridh=1, device=device,
                              num_steps=18, S_churn=40, S_min=0.05, S_max=50, S_noise=1.003,
                              sampling_method=method_map[args.method], sampling_params=sampling_params, compute_dtype=dtype)
    if int(os.environ.get('RANK', '0')) == 0:
        print(f'\n[EDM] Saved: {outname}  (denoiser rows: {res["net_rows"]}, reward collectives: {res["collectives"]})')
        print(f'[EDM] compute mode {args.dtype}; denoiser forwards: {net._graphs.path_report()}\n')
    if world > 1:
        dist.destroy_process_group()
    return res


if __name__ == '__main__':
    main()
