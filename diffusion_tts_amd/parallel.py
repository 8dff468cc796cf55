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
    def __init__(self, group=None):
        self.group = group
        self.enabled = dist.is_available() and dist.is_initialized()
        self.rank = dist.get_rank(group) if self.enabled else 0
        self.world = dist.get_world_size(group) if self.enabled else 1
        self.collectives = 0

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
        send = torch.zeros(cap, dtype=local.dtype, device=local.device)
        send[:local.numel()] = local.reshape(-1)
        recv = torch.empty(cap * self.world, dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(recv, send, group=self.group)
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
        dist.broadcast(tensor, src=owner if self.group is None else dist.get_global_rank(self.group, owner), group=self.group)
        self.collectives += 1
        return tensor
