"""Candidate sharding across the GPUs of one node (one process per GPU, torch.distributed over RCCL/xGMI).

The reference search loop is single-GPU (edm/main.py:12 pins CUDA_VISIBLE_DEVICES=0);
sharding the N candidates of one search iteration is this build's addition:
  - rank r owns a contiguous block of candidate indices n in [lo, hi);
  - every rank replays the same host RNG stream, so noise is never communicated;
  - the only data-path collective is ONE all-gather of the per-candidate rewards (N*B floats, <= 1 KB) per
    search iteration; every rank then takes the same first-max argmax, so the survivor is known everywhere
    from its index alone and is rebuilt locally from the replicated host noise.
"""
import os
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
        # DTS_SHARD_ALWAYS_COLLECT=1: a world of ONE still issues every collective (a one-rank RCCL all-gather / broadcast is a real
        # communicator and a real kernel): the only way to run the RCCL branch of this file on a one-GPU box (tests/test_gpu_sharded.py)
        self.solo = self.world == 1 and not (self.enabled and os.environ.get('DTS_SHARD_ALWAYS_COLLECT', '0') == '1')
        self.collectives = 0              # data-path collectives (reward all-gathers, survivor broadcasts)
        self.setup_collectives = 0        # one-off replication of host state (hash scale table, numpy RNG state)
        # RCCL ("nccl") moves device tensors directly over xGMI; gloo moves host tensors.  A tensor on the side the process
        # group has no backend for is staged through the other side (composite groups such as "cpu:gloo,cuda:nccl" serve both).
        backend = str(dist.get_backend(group)).lower() if self.enabled else ''
        self.dev_direct = 'nccl' in backend          # device tensors go out as they are
        self.host_direct = 'gloo' in backend or not self.enabled
        self.host_staged = self.enabled and not self.dev_direct    # device payloads take the host route (gloo-only groups)

    def _device(self):
        return torch.device('cuda', torch.cuda.current_device())

    def require_candidates(self, n_candidates: int, what: str):
        """Every rank must own at least one candidate: an empty shard would skip the collectives its peers then wait in forever.
        Raised identically on every rank (the arguments are replicated)."""
        if self.world > n_candidates:
            raise ValueError(f'{what}: N={n_candidates} candidates cannot be sharded over {self.world} ranks (need N >= ranks); '
                             f'run with fewer ranks or shard_candidates=False')

    def broadcast_host(self, t: torch.Tensor, src_rank: int = 0) -> torch.Tensor:
        """Replicates a HOST tensor from `src_rank` (group rank) in place and returns it."""
        if self.solo:
            return t
        src = src_rank if self.group is None else dist.get_global_rank(self.group, src_rank)
        if self.host_direct:
            dist.broadcast(t, src=src, group=self.group)
        else:
            d = t.to(self._device())
            dist.broadcast(d, src=src, group=self.group)
            t.copy_(d.cpu())
        self.setup_collectives += 1
        return t

    def replicate_scale_table(self, scale_fn, steps: int, K: int, N: int):
        """The eps-greedy step-size table (edm/main.py:776) comes from Python's per-process salted `hash()`: under
        torch.distributed.run every rank would build different candidates and rebuild a different pivot.  Rank 0's table is the
        single-process one; it is computed there once and broadcast ([steps,K,N] f64, a few KB), and every rank looks it up."""
        tab = torch.zeros(steps, K, N, dtype=torch.float64)
        if self.rank == 0 or self.world == 1:
            for i in range(steps):
                for k in range(K):
                    for n in range(N):
                        tab[i, k, n] = scale_fn(i, k, n)
        self.broadcast_host(tab, 0)
        return lambda i, k, n: float(tab[i, k, n])

    def span(self, n_candidates: int, rank: int = None) -> Tuple[int, int]:
        """Contiguous, near-even split of [0, N) (the first N % world ranks get one extra)."""
        r = self.rank if rank is None else rank
        q, rem = divmod(n_candidates, self.world)
        lo = r * q + min(r, rem)
        return lo, lo + q + (1 if r < rem else 0)

    def gather_rewards(self, local: torch.Tensor, n_candidates: int, rows_per_candidate: int) -> torch.Tensor:
        """local: this rank's rewards for candidates [lo,hi) in candidate-major order, shape [(hi-lo)*rows].
        Returns all N*rows rewards (candidate-major), identical on every rank, on `local`'s device."""
        if self.solo:
            return local
        q, rem = divmod(n_candidates, self.world)
        cap = (q + (1 if rem else 0)) * rows_per_candidate
        if local.is_cuda:
            dev = local.device if self.dev_direct else torch.device('cpu')
        else:
            dev = local.device if self.host_direct else self._device()
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
        if self.solo:
            return tensor
        owner = next(r for r in range(self.world) if self.span(n_candidates, r)[0] <= candidate < self.span(n_candidates, r)[1])
        src = owner if self.group is None else dist.get_global_rank(self.group, owner)
        if tensor.is_cuda and not self.dev_direct:
            host = tensor.cpu()
            dist.broadcast(host, src=src, group=self.group)
            tensor.copy_(host)
        elif not tensor.is_cuda and not self.host_direct:
            d = tensor.to(self._device())
            dist.broadcast(d, src=src, group=self.group)
            tensor.copy_(d.cpu())
        else:
            dist.broadcast(tensor, src=src, group=self.group)
        self.collectives += 1
        return tensor

    def sync_numpy_rng(self):
        """Every rank adopts rank 0's numpy global-generator state: MCTS picks rollout children with numpy's RNG
        (edm/main.py:593), which the reference never seeds; replicas must make the same picks, and rank 0's stream is
        exactly the single-process one."""
        import numpy as np
        if self.solo:
            return
        name, keys, pos, has_gauss, cached = np.random.get_state()
        t = torch.cat([torch.from_numpy(keys.astype(np.int64)), torch.tensor([pos, has_gauss], dtype=torch.int64)])
        g = torch.tensor([cached], dtype=torch.float64)
        self.broadcast_host(t, 0)
        self.broadcast_host(g, 0)
        np.random.set_state((name, t[:-2].numpy().astype(np.uint32), int(t[-2]), int(t[-1]), float(g[0])))
