"""The EDM noise-trajectory-search sampling loop on MI355X: drop-in for `generate_image_grid`
(edm/main.py:47-886) with the same `SamplingMethod` / `SamplingParams` plugin surface (edm/main.py:27-43).

What stays on the host (as in the reference): the sigma schedule, the search control flow, the RNG.  All search
randomness is drawn from the torch *CPU* global generator (and numpy's for MCTS) in exactly the reference's call
order -- including draws whose values the reference throws away -- and uploaded, because the parity oracle is the
reference's `--device cpu` run and a device Philox stream could not reproduce it (hard part 2).
What runs on the GPU: every tensor operation -- candidate construction (K14), the fp64 Heun/Euler step (K9), the
denoiser, quantisation (K10) and the scorer.

With torch.distributed initialised (one process per GPU) the N candidates of each search iteration are sharded
across ranks (parallel.CandidateShards): one all-gather of N*B rewards per iteration, identical argmax on every rank.
"""
import math
import os
from dataclasses import dataclass, field
from enum import Enum, auto
from typing import Any, Callable, Dict, Optional, Sequence

import numpy as np
import torch

from . import ops
from .hashing import builtin_scale
from .parallel import CandidateShards
from .scorers import Scorer, CompressibilityScorer


class SamplingMethod(Enum):          # edm/main.py:27-33
    MCTS = auto()
    BEAM_SEARCH = auto()
    ZERO_ORDER = auto()
    NAIVE = auto()
    REJECTION_SAMPLING = auto()
    EPS_GREEDY = auto()


@dataclass
class SamplingParams:                # edm/main.py:35-43
    B: int = 2
    N: int = 4
    K: int = 20
    lambda_param: float = 0.15
    eps: float = 0.4
    S: int = 8
    scorer: Scorer = field(default_factory=lambda: CompressibilityScorer(dtype=torch.float32))


def load_network(spec, device='cuda', dtype=ops.F16X3):
    """`network_pkl` argument of generate_image_grid.  The reference unpickles an NVIDIA EDM checkpoint from a URL
    (edm/main.py:69-70); URLs cannot be fetched here, so accepted forms are: a ready network object; the local path of
    an EDM network pickle (`*.pkl`, read by checkpoint.load_edm_pickle without executing its embedded source); a
    torch-saved dict {'cfg': EDMConfig kwargs, 'state_dict': {...reference keys...}}; or 'random:<preset>[:seed]'
    with preset in {adm_imagenet64, ddpmpp_cifar10} (random init + the documented weight rule)."""
    from . import init as dinit
    from .config import EDMConfig, adm_imagenet64, ddpmpp_cifar10
    from .networks import EDMPrecond
    if callable(spec) and hasattr(spec, 'round_sigma'):
        return spec
    if isinstance(spec, str) and spec.startswith('random:'):
        parts = spec.split(':')
        preset = {'adm_imagenet64': adm_imagenet64, 'ddpmpp_cifar10': ddpmpp_cifar10}[parts[1]]()
        seed = int(parts[2]) if len(parts) > 2 else 0
        sd, _ = dinit.refill_degenerate(dinit.edm_state_dict(preset, seed), seed)
        return EDMPrecond(preset, sd, device=device, dtype=dtype)
    if isinstance(spec, str) and spec.endswith(('.pt', '.pth')):
        blob = torch.load(spec, map_location='cpu')
        return EDMPrecond(EDMConfig(**blob['cfg']), blob['state_dict'], device=device, dtype=dtype)
    if isinstance(spec, str) and spec.endswith('.pkl'):
        # an NVIDIA EDM network pickle, e.g. a downloaded edm-imagenet-64x64-cond-adm.pkl (the reference's `network_pkl`,
        # main.py:157-158): weights and constructor arguments are read without executing the source embedded in the file
        from .checkpoint import load_edm_pickle
        cfg, sd = load_edm_pickle(spec)
        return EDMPrecond(cfg, sd, device=device, dtype=dtype)
    raise ValueError(f'cannot load network from {spec!r}: pass a network object, an EDM network .pkl (local path), a .pt '
                     f'state-dict bundle or "random:<preset>[:seed]" (URLs cannot be fetched here)')


class _Loop:
    """State shared by the search methods: schedule, device step, bookkeeping."""

    def __init__(self, net, device, num_steps, S_churn, S_min, S_max, S_noise, scale_fn, shards):
        self.net, self.dev, self.num_steps = net, device, num_steps
        self.S_churn, self.S_min, self.S_max, self.S_noise = S_churn, S_min, S_max, S_noise
        self.scale_fn, self.shards = scale_fn, shards
        self.rewards, self.selected = [], []
        self.record_noises, self.best_noises = False, {}
        self.reuse_winner = False
        self.forced = None                    # generate_image_grid(forced_selections=...)
        self.chunk = None                     # generate_image_grid(candidate_chunk=...)

    def up(self, t, dtype=None):
        return t.to(self.dev, dtype).contiguous() if dtype is not None else t.to(self.dev).contiguous()

    def churn(self, t_cur):
        gamma = min(self.S_churn / self.num_steps, np.sqrt(2) - 1) if self.S_min <= t_cur <= self.S_max else 0
        t_hat = self.net.round_sigma(t_cur + gamma * t_cur)
        coef = (t_hat ** 2 - t_cur ** 2).sqrt() * self.S_noise
        return float(t_hat), float(coef)

    def step(self, x_cur, t_cur, t_next, i, eps, labels, nb=None, interleave=False, live=None):
        """edm/main.py:82-96 on the device.  x_cur [xb,...] f64 is broadcast to nb rows; eps [nb,...] f64|f32.
        live: rows that are real candidates (the rest is batch-shape padding): only those count as candidate evaluations."""
        nb = eps.shape[0] if nb is None else nb
        evals0 = getattr(self.net, 'evals', None)
        t_hat, coef = self.churn(t_cur)
        t_hat_t = torch.tensor([t_hat], dtype=torch.float64)
        x_hat = ops.heun_xhat(x_cur, eps, coef, nb, interleave)
        D = self.net(x_hat, t_hat_t, labels)
        d_cur, x_next = ops.heun_euler(x_hat, D, t_hat, float(t_next))
        if i < self.num_steps - 1:
            D = self.net(x_next, torch.as_tensor(t_next, dtype=torch.float64).reshape(1), labels)
            ops.heun_correct(x_hat, D, d_cur, t_hat, float(t_next), x_next)
        if live is not None and live != nb and evals0 is not None:       # padding rows went through the denoiser but are not counted
            self.net.evals = evals0 + (self.net.evals - evals0) // nb * live
        return x_next, D

    def score(self, scorer, x, labels):
        img = ops.quantize_u8(x)
        ts = torch.zeros(img.shape[0], device=self.dev)
        return scorer(img, labels, ts)


# ---------------------------------------------------------------------------------------------------------
def _naive(L: _Loop, t_steps, x_next, labels, p, pre):
    for i in range(L.num_steps):
        eps = torch.randn(x_next.shape, dtype=torch.float64)                         # edm/main.py:865
        x_next, _ = L.step(x_next, t_steps[i], t_steps[i + 1], i, L.up(eps), labels)
    return x_next


def _rejection(L: _Loop, t_steps, x_next, labels, p, pre):
    """edm/main.py:101-137; rows are b-major (repeat_interleave).  Sharded: rank r carries candidates [lo,hi) of
    every sample through all steps; rewards all-gathered once; the winner's image broadcast from its owner."""
    N, B = p.N, x_next.shape[0]
    L.shards.require_candidates(N, 'rejection sampling')
    lo, hi = L.shards.span(N)
    nl = hi - lo
    shape1 = tuple(x_next.shape[1:])
    lab = None if labels is None else labels.repeat_interleave(nl, dim=0).contiguous()
    x = None
    for i in range(L.num_steps):
        if pre is not None and i in pre:
            eps = pre[i][:, :N].reshape(B * N, *shape1).to(torch.float64)
        else:
            eps = torch.randn((B * N,) + shape1, dtype=torch.float64)                # edm/main.py:120
        eps_l = L.up(eps.view(B, N, *shape1)[:, lo:hi].reshape(B * nl, *shape1))
        if x is None:
            x, _ = L.step(x_next, t_steps[i], t_steps[i + 1], i, eps_l, lab, interleave=True)
        else:
            x, _ = L.step(x, t_steps[i], t_steps[i + 1], i, eps_l, lab)
    loc = L.score(p.scorer, x, lab).to(L.dev, torch.float32).view(B, nl).t().contiguous().reshape(-1)   # candidate-major
    allr = L.shards.gather_rewards(loc, N, B).view(N, B).t().contiguous().cpu()                # [B, N]
    best = allr.argmax(dim=1)                                                                  # edm/main.py:136
    L.rewards.append(allr)
    L.selected.append(best)
    out = []
    xr = x.view(B, nl, *shape1)
    for b, j in enumerate(best.tolist()):
        row = xr[b, j - lo].clone() if lo <= j < hi else torch.empty(shape1, dtype=torch.float64, device=L.dev)
        out.append(L.shards.broadcast_from_owner(row, j, N))
    return torch.stack(out)


def _beam(L: _Loop, t_steps, x_next, labels, p, pre):
    # The reference branch is dead code: it reads method_params.b / .k, which SamplingParams does not define
    # (edm/main.py:140), so `--method beam` raises AttributeError on entry.  Kept verbatim for drop-in behaviour.
    b, k = p.b, p.k
    raise RuntimeError('unreachable')


class _Lookahead:
    """Pulls items of a generator up to `depth` ahead of their use.  The search randomness never depends on search
    results (only the pivot does), so the host can draw iteration k+1's numbers -- in the reference's order -- while the
    GPU is still busy with iteration k.  `stage(item)` (optional) starts the item's host->device copy right away."""

    def __init__(self, gen, depth=2, stage=None):
        self.gen, self.depth, self.buf, self.stage = gen, depth, [], stage

    def prefetch(self):
        while len(self.buf) < self.depth:
            try:
                item = next(self.gen)
            except StopIteration:
                break
            self.buf.append(item if self.stage is None else self.stage(item))

    def get(self):
        if not self.buf:
            self.prefetch()
        return self.buf.pop(0)


def _eps_greedy_draws(L, p, pre, shape, lam):
    """The host-RNG stream of edm/main.py:724-797 in call order: per timestep one pivot, then per local-search
    iteration the N coins / Gaussians / hash-derived scales."""
    N, K, eps_p = p.N, p.K, p.eps
    if not (pre is not None and 'pivot' in pre):
        torch.randn(shape, dtype=torch.float64)                                       # :727, overwritten at :737
    for i in range(L.num_steps):
        yield pre[f'pivot_{i}'] if (pre is not None and f'pivot_{i}' in pre) else torch.randn(shape, dtype=torch.float64)
        for k in range(K):
            g_h, mode, scale = [], [], []
            for n in range(N):
                if torch.rand(1) < (1 - eps_p):                                       # :751
                    if pre is not None and i in pre and k < pre[i].shape[1] and n < pre[i].shape[2]:
                        g = pre[i][:, k, n].reshape(shape).to(torch.float64)
                    else:
                        g = torch.randn(shape, dtype=torch.float64)                   # :767
                    s = torch.ones([shape[0], 1, 1, 1]) * L.scale_fn(i, k, n) * lam   # float32, :779
                    mode.append(1)
                    scale.append(float(s.flatten()[0]))
                else:
                    key = f'fresh_{i}_{k}_{n}'
                    g = pre[key].to(torch.float64) if (pre is not None and key in pre) else torch.randn(shape, dtype=torch.float64)
                    mode.append(0)
                    scale.append(0.0)
                g_h.append(g)
            yield g_h, torch.tensor(mode, dtype=torch.int32), torch.tensor(scale, dtype=torch.float32)


def _eps_greedy(L: _Loop, t_steps, x_next, labels, p, pre):
    """edm/main.py:714-860 (ZERO_ORDER and EPS_GREEDY share this branch in the reference)."""
    lam = p.lambda_param * np.sqrt(3 * 64 * 64)            # scaled by 3*64*64 whatever the resolution (:716)
    N, K, B = p.N, p.K, x_next.shape[0]
    shape = tuple(x_next.shape)
    L.shards.require_candidates(N, 'eps-greedy / zero-order search')
    if not L.shards.solo:
        # hash()-derived step sizes are salted per process (PYTHONHASHSEED): ranks must share ONE table or their candidates and
        # rebuilt pivots diverge.  Rank 0's table == the single-process run's.
        L.scale_fn = L.shards.replicate_scale_table(L.scale_fn, L.num_steps, K, N)
    lo, hi = L.shards.span(N)
    nl = hi - lo
    lab_l = None if labels is None else labels.repeat(nl, 1).contiguous()
    copy_stream = torch.cuda.Stream(device=L.dev)

    def stage(item):
        # candidate directions of this rank (6.3 MB of fp64 at N=64): pinned and sent on a side stream while the GPU runs the
        # previous iteration -- an in-line pageable upload left the GPU idle for ~3 ms of every 41 ms iteration
        if not isinstance(item, tuple):
            return item                                                                # a pivot draw
        g_h, mode_t, scale_t = item
        host = pinned[stage.n % len(pinned)]                                           # ring: depth + 2 buffers, reused
        stage.n += 1
        if hi > lo:
            torch.cat(g_h[lo:hi], dim=0, out=host)
        with torch.cuda.stream(copy_stream):
            g_dev = host.to(L.dev, non_blocking=True)
            ready = torch.cuda.Event()
            ready.record(copy_stream)
        return g_h, mode_t, scale_t, g_dev, ready, host                                # `host` stays alive until the copy has run

    stage.n = 0
    pinned = [torch.empty((nl * B,) + shape[1:], dtype=torch.float64).pin_memory() for _ in range(4)]
    draws = _Lookahead(_eps_greedy_draws(L, p, pre, shape, lam), stage=stage)
    for i in range(L.num_steps):
        t_cur, t_next = t_steps[i], t_steps[i + 1]
        x_cur = x_next
        pivot = L.up(draws.get(), torch.float64)
        x_win = None
        for k in range(K):
            g_h, mode_t, scale_t, g_l, ready, _pinned = draws.get()                    # n-major rows (:800)
            torch.cuda.current_stream(L.dev).wait_event(ready)
            g_l.record_stream(torch.cuda.current_stream(L.dev))
            cand = ops.candidate_noise(pivot, g_l, L.up(mode_t[lo:hi]), L.up(scale_t[lo:hi]))
            if L.chunk is None or L.chunk * B >= nl * B:
                x_cand, x0 = L.step(x_cur, t_cur, t_next, i, cand, lab_l, nb=nl * B)
                loc = L.score(p.scorer, x0, lab_l).to(L.dev, torch.float32)
            else:
                # candidate_chunk: this rank's candidates go through the denoiser and the scorer in sequential pieces of `chunk` candidates
                # (n-major rows, so a piece is chunk * B consecutive rows) -- exactly the launches rank r of N / chunk ranks issues for
                # its share, on one GPU
                xs, locs = [], []
                for a in range(0, nl * B, L.chunk * B):
                    z = min(nl * B, a + L.chunk * B)
                    lab_c = None if lab_l is None else lab_l[a:z].contiguous()
                    xc_, x0_ = L.step(x_cur, t_cur, t_next, i, cand[a:z].contiguous(), lab_c, nb=z - a)
                    xs.append(xc_.clone())
                    locs.append(L.score(p.scorer, x0_, lab_c).to(L.dev, torch.float32).clone())
                x_cand, loc = torch.cat(xs, dim=0), torch.cat(locs, dim=0)
            draws.prefetch()                                                           # host draws overlap the queued GPU work
            scores = L.shards.gather_rewards(loc, N, B).reshape(N, B).cpu()
            best = scores.argmax(dim=0)                                                # first max (:842)
            L.rewards.append(scores)
            L.selected.append(best)
            if L.forced is not None:                                                   # follow another run's trajectory (B == 1)
                best = torch.tensor([L.forced[len(L.selected) - 1]])
            # survivor rebuilt from the replicated host noise: no second collective
            bl = best.tolist()
            g_w = torch.stack([g_h[j][b] for b, j in enumerate(bl)])
            if len(set(bl)) == 1:
                pivot = ops.candidate_noise(pivot, L.up(g_w), L.up(mode_t[bl[0]:bl[0] + 1]), L.up(scale_t[bl[0]:bl[0] + 1]))
            else:
                rows = [ops.candidate_noise(pivot[b:b + 1].contiguous(), L.up(g_w[b:b + 1]), L.up(mode_t[j:j + 1]),
                                            L.up(scale_t[j:j + 1])) for b, j in enumerate(bl)]
                pivot = torch.cat(rows, dim=0)
            if L.record_noises:                                                        # best_noises_this_timestep (:741, :854)
                L.best_noises.setdefault(i, []).append(pivot.cpu())
            if L.reuse_winner and k == K - 1:
                # the step the reference recomputes at :860 for the final pivot IS the winner's row of this iteration
                rows = []
                for b, j in enumerate(bl):
                    own = lo <= j < hi
                    r = x_cand[(j - lo) * B + b].clone() if own else torch.empty(shape[1:], dtype=torch.float64, device=L.dev)
                    rows.append(L.shards.broadcast_from_owner(r, j, N))
                x_win = torch.stack(rows)
        if x_win is not None:
            x_next = x_win
        else:
            x_next, _ = L.step(x_cur, t_cur, t_next, i, pivot, labels)                 # :860
    return x_next


class _Node:
    __slots__ = ('x', 'children', 'reward', 'visit')

    def __init__(self, x, visit=0):
        self.x, self.children, self.reward, self.visit = x, [], 0, visit


# granularity the ragged rollout batches are padded to (copies of row 0, not counted as evaluations).  1 = no padding: every batch size
# 1 .. 16 gets its own captured HIP graph (graphs.py SMALL budget); 4 = the former rule (four batch shapes, up to 3 wasted rows per forward)
# DTS_MCTS_PAD_ROWS sets it.  DTS_MCTS_PAD keeps the boolean meaning it had through round 4 ('1' = pad to multiples of four, '0' = no padding).
def _mcts_pad_rows():
    v = os.environ.get('DTS_MCTS_PAD_ROWS')
    if v is None:
        legacy = os.environ.get('DTS_MCTS_PAD')
        return 4 if legacy is not None and legacy.strip() not in ('', '0') else 1
    try:
        return max(1, min(16, int(v)))
    except ValueError:
        import warnings
        warnings.warn(f'DTS_MCTS_PAD_ROWS={v!r} is not an integer: rollout batches are not padded')
        return 1


MCTS_PAD_ROLLOUTS = _mcts_pad_rows()


def _mcts(L: _Loop, t_steps, x_next, labels, p, pre):
    """edm/main.py:405-713.  Tree statistics, UCB1 selection and the numpy child draw are the reference's; the
    tensor work is re-batched without changing any value: a node's b expansions are one batched step (same x,
    b noises) and the group's deterministic zero-noise rollouts advance together, one batched step per timestep
    over the simulations that have reached it (the reference runs them one by one, batch 1)."""
    b, S, B, ns = p.N, p.S, x_next.shape[0], L.num_steps
    shape1 = tuple(x_next.shape[1:])
    results = []
    L.shards.sync_numpy_rng()                         # replicas must draw the same rollout children (edm/main.py:593)
    mbs = min(2, B)
    for mb0 in range(0, B, mbs):
        xb = x_next[mb0:mb0 + mbs]
        lb = None if labels is None else labels[mb0:mb0 + mbs]
        m = xb.shape[0]
        noise = {}
        for i in range(ns):
            if pre is not None and i in pre:
                noise[i] = L.up(pre[i].repeat(m, 1, 1, 1, 1))
            else:
                noise[i] = L.up(torch.randn(m, b, *shape1))                           # float32 (:446)
        roots = [_Node(xb[s:s + 1].clone(), visit=1) for s in range(m)]
        lab1 = lambda s: None if lb is None else lb[s:s + 1]
        for i in range(ns):
            t_cur, t_next = t_steps[i], t_steps[i + 1]
            todo = [(s, j) for s in range(m) if not roots[s].children for j in range(b)]
            if todo:
                xe = torch.cat([roots[s].x for s, j in todo], dim=0)
                ne = torch.cat([noise[i][s:s + 1, j] for s, j in todo], dim=0).contiguous()
                le = None if lb is None else torch.cat([lb[s:s + 1] for s, j in todo], dim=0).contiguous()
                xn, _ = L.step(xe, t_cur, t_next, i, ne, le)
                for q, (s, j) in enumerate(todo):
                    roots[s].children.append(_Node(xn[q:q + 1]))
            group = min(16, S * m)
            for g0 in range(0, S * m, group):
                paths, starts = [], []
                for sim in range(g0, min(g0 + group, S * m)):
                    s = sim % m
                    node, it = roots[s], i
                    tc, tn = t_cur, t_next
                    path = [node]
                    while node.children:
                        ucb = [float('inf') if c.visit == 0 else
                               c.reward / c.visit + np.sqrt(2 * np.log(node.visit) / c.visit) for c in node.children]
                        node = node.children[int(np.argmax(ucb))]
                        it += 1
                        if it < ns:
                            tc, tn = t_steps[it], t_steps[it + 1]
                        path.append(node)
                    if it < ns - 1:
                        for j in range(b):
                            torch.randn(1, *shape1)            # the reference's eager .get() default (:578-579): drawn, unused
                        eb = noise[it][s].contiguous()                                  # [b, ...] f32
                        lbb = None if lb is None else lb[s:s + 1].repeat(b, 1).contiguous()
                        xc, _ = L.step(node.x, tc, tn, it, eb, lbb, nb=b)
                        for j in range(b):
                            node.children.append(_Node(xc[j:j + 1]))
                        node = node.children[np.random.randint(0, len(node.children))]   # numpy global RNG (:593)
                        it += 1
                        path.append(node)
                    paths.append(path)
                    starts.append((node.x, it, s))
                # batched ragged rollouts; sharded: rank r advances the simulations [lo, hi) of this group and the
                # group's rewards are all-gathered (the tree itself is replicated: SURVEY.md section 8e)
                nsim = len(starts)
                slo, shi = L.shards.span(nsim)
                mine = list(range(slo, shi))
                cur = {q: starts[q][0].clone() for q in mine}
                if mine:
                    for j in range(min(starts[q][1] for q in mine), ns):
                        act = [q for q in mine if starts[q][1] <= j]
                        xa = torch.cat([cur[q] for q in act], dim=0)
                        la = None if lb is None else torch.cat([lb[starts[q][2]:starts[q][2] + 1] for q in act], dim=0).contiguous()
                        # the number of live rollouts changes from step to step (1 .. 16).  Every size replays its own captured HIP graph
                        # (graphs.py keeps up to 16 small shapes per module) instead of launching ~600 kernels one by one from the host.
                        # DTS_MCTS_PAD_ROWS=4 restores the former rule -- sizes rounded up to a multiple of 4 with copies of row 0, four shapes --
                        # which cost up to 3 wasted rows per forward at ~0.5 ms per row (profiles/r05_experiments.txt item 13).  Rows are
                        # independent of each other's VALUES; they can depend on the batch SIZE in the last bits, because the conv launchers
                        # choose the split-K factor / launch form from the row count (another fixed f32 summation order -- as is every
                        # batch size against the reference's batch-1 rollouts, edm/main.py:640-660).  Padding rows are not counted as evaluations.
                        ka = len(act)
                        g_ = MCTS_PAD_ROLLOUTS if (ka <= 16 and getattr(L.net, 'dtype', None) != torch.float32) else 1
                        kp = -(-ka // g_) * g_
                        if kp > ka:
                            xa = torch.cat([xa, xa[:1].expand(kp - ka, *xa.shape[1:])], dim=0)
                            if la is not None:
                                la = torch.cat([la, la[:1].expand(kp - ka, la.shape[1])], dim=0).contiguous()
                        xo, _ = L.step(xa, t_steps[j], t_steps[j + 1], j, torch.zeros_like(xa), la, live=ka)
                        for r_, q in enumerate(act):
                            cur[q] = xo[r_:r_ + 1]
                    den = torch.cat([cur[q] for q in mine], dim=0)
                    sl = None if lb is None else torch.cat([lb[starts[q][2]:starts[q][2] + 1] for q in mine], dim=0).contiguous()
                    loc = L.score(p.scorer, den, sl).to(L.dev, torch.float32)
                else:
                    loc = torch.empty(0, dtype=torch.float32, device=L.dev)
                rew = L.shards.gather_rewards(loc, nsim, 1).cpu()
                L.rewards.append(rew)
                for path, r in zip(paths, rew):
                    for nd in path:
                        nd.reward += r.item()
                        nd.visit += 1
            for s in range(m):
                best, best_r, best_j = None, -float('inf'), -1
                for j, c in enumerate(roots[s].children):
                    if c.visit > 0 and c.reward / c.visit > best_r:
                        best, best_r, best_j = c, c.reward / c.visit, j
                assert best is not None
                L.selected.append(torch.tensor([best_j]))
                roots[s] = best
        results += [r.x for r in roots]
    return torch.cat(results, dim=0)


_METHODS = {
    SamplingMethod.NAIVE: _naive, SamplingMethod.REJECTION_SAMPLING: _rejection, SamplingMethod.BEAM_SEARCH: _beam,
    SamplingMethod.MCTS: _mcts, SamplingMethod.ZERO_ORDER: _eps_greedy, SamplingMethod.EPS_GREEDY: _eps_greedy,
}


@torch.no_grad()
def generate_image_grid(
    network_pkl, dest_path, latents, class_labels,
    seed=0, gridw=8, gridh=8, device=torch.device('cuda'),
    num_steps=18, sigma_min=0.002, sigma_max=80, rho=7,
    S_churn=0, S_min=0, S_max=float('inf'), S_noise=1,
    sampling_method: SamplingMethod = SamplingMethod.NAIVE,
    sampling_params: Optional[Dict[str, Any]] = None,
    precomputed_noise: Optional[Dict[Any, torch.Tensor]] = None,
    *, scale_fn: Callable[[int, int, int], float] = builtin_scale, compute_dtype=ops.F16X3, verbose=True,
    reuse_winner: Optional[bool] = None, record_noises: bool = False, shard_candidates: bool = True,
    forced_selections: Optional[Sequence[int]] = None, candidate_chunk: Optional[int] = None,
):
    """Same positional/keyword surface as edm/main.py:47-55.  Keyword-only extras: `scale_fn` (the hash-derived step
    table, edm/main.py:776), `compute_dtype` (default ops.F16X3: split precision, the reference's fp32 selections at a third of the 16-bit
    rate; float32 = parity mode on the f32 matrix instruction; bfloat16 / float16 = throughput modes), `verbose`, `reuse_winner` (see below), `record_noises`
    (keep the per-iteration winning noises for `dump_noise_trajectory`; costs one D2H copy per iteration), `forced_selections` (eps-greedy /
    zero-order, one image: decision d continues from candidate forced_selections[d] instead of its own argmax, which is still what
    `selected` records -- how a search is walked along ANOTHER run's trajectory so that every decision of the two stays comparable; the
    parity tests follow the reference's recorded run this way; it must hold num_steps * K indices in [0, N), and in a sharded run every
    rank must pass the same list: the override is applied after the reward all-gather, on every rank alike), `candidate_chunk` (eps-greedy /
    zero-order: evaluate the candidates of an iteration in sequential pieces of that many candidates instead of one batch -- the kernels,
    split-K factors and launch forms a rank of a sharded run uses for a share of that size, on one GPU; values are the batched run's up to
    the f32 summation order those launch forms imply).  Writes the PNG grid like the
    reference when `dest_path` is not None and additionally returns a dict with the final state and the search trace."""
    device = torch.device(device)
    if device.type != 'cuda':
        raise RuntimeError('generate_image_grid (HIP): device must be a GPU; the CPU path is the reference itself')
    torch.manual_seed(seed)                                                           # edm/main.py:58
    p = SamplingParams(**(sampling_params or {}))
    if verbose:
        print(f'Using sampling method: {sampling_method.name}')
    net = load_network(network_pkl, device=device, dtype=compute_dtype)
    shards = CandidateShards(enabled=shard_candidates)     # False: this rank runs the whole search (bulk.py shards seeds instead)
    step_indices = torch.arange(num_steps, dtype=torch.float64)                       # edm/main.py:78-80 (host)
    t_steps = (sigma_max ** (1 / rho) + step_indices / (num_steps - 1) * (sigma_min ** (1 / rho) - sigma_max ** (1 / rho))) ** rho
    t_steps = torch.cat([net.round_sigma(t_steps), torch.zeros_like(t_steps[:1])])
    L = _Loop(net, device, num_steps, S_churn, S_min, S_max, S_noise, scale_fn, shards)
    # eps-greedy: the reference re-runs step() at batch 1 for the final pivot of each timestep (edm/main.py:860) although
    # that row was just computed in the last candidate batch.  The default mode (f16x3) and float32 recompute it like the reference, so
    # `net_rows` equals the reference's count (config 3: 8 995); the 16-bit throughput modes reuse the row (8 960 rows, same values).
    L.reuse_winner = (compute_dtype not in (torch.float32, ops.F16X3)) if reuse_winner is None else bool(reuse_winner)
    L.record_noises = bool(record_noises)
    if forced_selections is not None:
        if sampling_method not in (SamplingMethod.EPS_GREEDY, SamplingMethod.ZERO_ORDER) or latents.shape[0] != 1:
            raise ValueError('forced_selections: eps-greedy / zero-order search of one image only')
        L.forced = [int(v) for v in forced_selections]
        if len(L.forced) != num_steps * p.K or any(not 0 <= v < p.N for v in L.forced):
            raise ValueError(f'forced_selections: need num_steps * K = {num_steps * p.K} indices in [0, {p.N}), got {len(L.forced)} '
                             f'(range {min(L.forced, default=None)} .. {max(L.forced, default=None)})')
    if candidate_chunk is not None:
        if sampling_method not in (SamplingMethod.EPS_GREEDY, SamplingMethod.ZERO_ORDER) or int(candidate_chunk) < 1:
            raise ValueError('candidate_chunk: a positive candidate count, eps-greedy / zero-order search only')
        L.chunk = int(candidate_chunk)
    x0 = (latents.to(torch.float64).cpu() * t_steps[0]).to(device).contiguous()       # edm/main.py:99
    labels = None if class_labels is None else class_labels.to(device, torch.float32).contiguous()
    evals0 = getattr(net, 'evals', 0)
    x_next = _METHODS[sampling_method](L, t_steps, x0, labels, p, precomputed_noise)
    image = ops.quantize_u8(x_next)                                                   # edm/main.py:869
    scores = p.scorer(image.clone(), labels, torch.zeros(image.shape[0], device=device))
    avg_score = float(scores.float().mean())
    if verbose:
        print(f'Average score: {avg_score}')
    img_cpu = image.cpu()
    if dest_path is not None and (shards.rank == 0 or not shard_candidates):
        import PIL.Image
        if verbose:
            print(f'Saving image grid to "{dest_path}"...')
        r, c = net.img_resolution, net.img_channels
        grid = img_cpu.reshape(gridh, gridw, *img_cpu.shape[1:]).permute(0, 3, 1, 4, 2).reshape(gridh * r, gridw * r, c)
        PIL.Image.fromarray(grid.numpy(), 'RGB').save(dest_path)
    if verbose:
        print('Done.')
    return dict(x=x_next, image=img_cpu, final_scores=scores.cpu(), avg_score=avg_score, t_steps=t_steps,
                rewards=L.rewards, selected=L.selected, net_rows=getattr(net, 'evals', 0) - evals0,
                collectives=shards.collectives, best_noises={i: torch.stack(v) for i, v in L.best_noises.items()})


def dump_noise_trajectory(result, directory='.'):
    """Write the winning noise of every local-search iteration in the format the reference's diffusion-map tool loads
    (edm/dmap.py:16-24): `all_timestep_noises.pkl` = {timestep index: Tensor[K, B, C, H, W]} and `t_steps.pkl`.  The
    reference's loop collects exactly this list (edm/main.py:739-741, 853-854) but never writes it; `result` is the
    dict returned by `generate_image_grid(..., record_noises=True)`."""
    import os
    import pickle
    if not result.get('best_noises'):
        raise ValueError('no noise trajectory recorded: call generate_image_grid(..., record_noises=True) with a local-search method')
    os.makedirs(directory, exist_ok=True)
    with open(os.path.join(directory, 'all_timestep_noises.pkl'), 'wb') as f:
        pickle.dump({int(i): v.cpu() for i, v in result['best_noises'].items()}, f)
    with open(os.path.join(directory, 't_steps.pkl'), 'wb') as f:
        pickle.dump(result['t_steps'].cpu(), f)
