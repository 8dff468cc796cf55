"""Candidate sharding across the GPUs of one node (one process per GPU, torch.distributed over RCCL/xGMI).

The reference search loop is single-GPU (edm/main.py:12 pins CUDA_VISIBLE_DEVICES=0);
sharding the N candidates of one search iteration is this build's addition:
  - rank r owns a contiguous block of candidate indices n in [lo, hi);
  - every rank replays the same host RNG stream, so noise is never communicated;
  - the only data-path collective is ONE all-gather of the per-candidate rewards (N*B floats, <= 1 KB) per
    search iteration; every rank then takes the same first-max argmax, so the survivor is known everywhere
    from its index alone and is rebuilt locally from the replicated host noise.
"""
from typing import Tuple

import torch
import torch.distributed as dist


class CandidateShards:
    def __init__(self, group=None, enabled=True):
        """enabled=False: a world-of-one view even inside an initialised process group (seed-sharded bulk generation runs
        whole searches per rank, bulk.py)."""
        self.group = group
        self.enabled = enabled and dist.is_available() and dist.is_initialized()
        self.rank = dist.get_rank(group) if self.enabled else 0
        self.world = dist.get_world_size(group) if self.enabled else 1
        self.collectives = 0
        # RCCL ("nccl") moves device tensors directly over xGMI; with the gloo backend (CPU tests, or several ranks
        # sharing one GPU) the few bytes exchanged are staged through the host
        self.host_staged = self.enabled and dist.get_backend(group) == 'gloo'

    def span(self, n_candidates: int, rank: int = None) -> Tuple[int, int]:
        """Contiguous, near-even split of [0, N) (the first N % world ranks get one extra)."""
        r = self.rank if rank is None else rank
        q, rem = divmod(n_candidates, self.world)
        lo = r * q + min(r, rem)
        return lo, lo + q + (1 if r < rem else 0)

    def gather_rewards(self, local: torch.Tensor, n_candidates: int, rows_per_candidate: int) -> torch.Tensor:
        """local: this rank's rewards for candidates [lo,hi) in candidate-major order, shape [(hi-lo)*rows].
        Returns all N*rows rewards (candidate-major), identical on every rank, on `local`'s device."""
        if self.world == 1:
            return local
        q, rem = divmod(n_candidates, self.world)
        cap = (q + (1 if rem else 0)) * rows_per_candidate
        dev = torch.device('cpu') if self.host_staged else local.device
        send = torch.zeros(cap, dtype=local.dtype, device=dev)
        send[:local.numel()] = local.reshape(-1).to(dev)
        recv = torch.empty(cap * self.world, dtype=local.dtype, device=dev)
        dist.all_gather_into_tensor(recv, send, group=self.group)
        recv = recv.to(local.device)
        self.collectives += 1
        parts = []
        for r in range(self.world):
            lo, hi = self.span(n_candidates, r)
            parts.append(recv[r * cap:r * cap + (hi - lo) * rows_per_candidate])
        return torch.cat(parts)

    def broadcast_from_owner(self, tensor: torch.Tensor, candidate: int, n_candidates: int) -> torch.Tensor:
        """Rejection sampling's survivor image travels once from the rank that owns `candidate`."""
        if self.world == 1:
            return tensor
        owner = next(r for r in range(self.world) if self.span(n_candidates, r)[0] <= candidate < self.span(n_candidates, r)[1])
        src = owner if self.group is None else dist.get_global_rank(self.group, owner)
        if self.host_staged and tensor.is_cuda:
            host = tensor.cpu()
            dist.broadcast(host, src=src, group=self.group)
            tensor.copy_(host)
        else:
            dist.broadcast(tensor, src=src, group=self.group)
        self.collectives += 1
        return tensor

    def sync_numpy_rng(self):
        """Every rank adopts rank 0's numpy global-generator state: MCTS picks rollout children with numpy's RNG
        (edm/main.py:593), which the reference never seeds; replicas must make the same picks, and rank 0's stream is
        exactly the single-process one."""
        import numpy as np
        if self.world == 1:
            return
        name, keys, pos, has_gauss, cached = np.random.get_state()
        t = torch.cat([torch.from_numpy(keys.astype(np.int64)), torch.tensor([pos, has_gauss], dtype=torch.int64)])
        g = torch.tensor([cached], dtype=torch.float64)
        src = 0 if self.group is None else dist.get_global_rank(self.group, 0)
        if self.host_staged:
            dist.broadcast(t, src=src, group=self.group)
            dist.broadcast(g, src=src, group=self.group)
        else:
            dev = torch.device('cuda', torch.cuda.current_device())
            td, gd = t.to(dev), g.to(dev)
            dist.broadcast(td, src=src, group=self.group)
            dist.broadcast(gd, src=src, group=self.group)
            t, g = td.cpu(), gd.cpu()
        np.random.set_state((name, t[:-2].numpy().astype(np.uint32), int(t[-2]), int(t[-1]), float(g[0])))
