"""Seed-sharded bulk generation: many independent searches, one image per seed, seeds split over the ranks with NO data-path
collective (the second way the path shards; the candidates of ONE search are sharded by
parallel.CandidateShards instead).

Mirrors the batch bookkeeping of the reference's bulk generator (edm/generate.py:254-309): seeds are split into
`ceil(len(seeds) / (batch * world)) * world` batches by `tensor_split` and rank r takes batches r, r+world, ...; images are
written as `<outdir>/<seed:06d>.png` (`--subdirs`: one directory per 1000 seeds).  Each image runs the search loop of
`generate_image_grid` with `seed` as its RNG seed, so a seed's result does not depend on the number of ranks.  Latents and
labels come from a per-seed CPU generator (the reference's StackedRandomGenerator draws on the CUDA device, a stream no
other device reproduces; its per-seed independence is what is kept)."""
import os
from typing import Any, Dict, Iterable, List, Optional

import torch
import torch.distributed as dist

from . import ops
from .sampler import SamplingMethod, generate_image_grid, load_network


def parse_int_list(s):
    """'1,2,5-10' -> [1, 2, 5, 6, 7, 8, 9, 10]  (edm/generate.py:197-207)."""
    if isinstance(s, (list, tuple)):
        return [int(v) for v in s]
    out: List[int] = []
    for part in str(s).split(','):
        if '-' in part:
            lo, hi = part.split('-')
            out.extend(range(int(lo), int(hi) + 1))
        else:
            out.append(int(part))
    return out


def rank_batches(seeds: Iterable[int], max_batch_size: int, rank: int, world: int):
    """The reference's split (edm/generate.py:259-261)."""
    seeds = list(seeds)
    num_batches = ((len(seeds) - 1) // (max_batch_size * world) + 1) * world
    return torch.as_tensor(seeds, dtype=torch.int64).tensor_split(num_batches)[rank::world]


def seed_inputs(seed: int, net, class_idx: Optional[int] = None):
    g = torch.Generator().manual_seed(int(seed))
    latents = torch.randn([1, net.img_channels, net.img_resolution, net.img_resolution], generator=g)
    labels = None
    if net.label_dim:
        idx = int(torch.randint(net.label_dim, size=[1], generator=g)) if class_idx is None else int(class_idx)
        labels = torch.zeros(1, net.label_dim)
        labels[0, idx] = 1
    return latents, labels


@torch.no_grad()
def generate_seeds(network, seeds, outdir, *, sampling_method=SamplingMethod.NAIVE, sampling_params: Optional[Dict[str, Any]] = None,
                   class_idx: Optional[int] = None, max_batch_size: int = 64, subdirs: bool = False, device='cuda',
                   compute_dtype=ops.F16X3, verbose=False, **sampler_kwargs):
    """Returns {seed: result dict of generate_image_grid} for the seeds this rank generated."""
    dist_on = dist.is_available() and dist.is_initialized()
    rank, world = (dist.get_rank(), dist.get_world_size()) if dist_on else (0, 1)
    net = load_network(network, device=device, dtype=compute_dtype)
    done = {}
    for batch in rank_batches(parse_int_list(seeds), max_batch_size, rank, world):
        for seed in batch.tolist():
            latents, labels = seed_inputs(seed, net, class_idx)
            image_dir = os.path.join(outdir, f'{seed - seed % 1000:06d}') if subdirs else outdir
            os.makedirs(image_dir, exist_ok=True)
            done[seed] = generate_image_grid(net, os.path.join(image_dir, f'{seed:06d}.png'), latents, labels, seed=seed, gridw=1,
                                             gridh=1, device=device, sampling_method=sampling_method,
                                             sampling_params=sampling_params, compute_dtype=compute_dtype, verbose=verbose,
                                             shard_candidates=False, **sampler_kwargs)
    return done
