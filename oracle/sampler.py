"""Oracle: the EDM noise-trajectory-search sampling loop on CPU (torch CPU, fp64 state).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restates `generate_image_grid` of the reference with the network and scorer injected instead of
being unpickled / downloaded, and with a trace (rewards, selected indices) returned instead of a PNG:
  edm/main.py:78-80    sigma schedule                    -> sigma_schedule
  edm/main.py:82-96    step() (churn + Euler + Heun)     -> heun_step
  edm/main.py:101-137  REJECTION_SAMPLING                -> _rejection
  edm/main.py:138-140  BEAM_SEARCH (dead: AttributeError)-> _beam
  edm/main.py:405-713  MCTS                              -> _mcts
  edm/main.py:714-860  ZERO_ORDER | EPS_GREEDY           -> _eps_greedy
  edm/main.py:862-866  NAIVE                             -> _naive
  edm/main.py:868-877  final uint8 image + mean score    -> search() tail
The host-RNG call order (torch global CPU generator, numpy global generator) is the reference's,
including draws whose values are discarded (edm/main.py:578-579).
"""
from dataclasses import dataclass
from typing import Any, Callable, Dict, Optional

import numpy as np
import torch


@dataclass
class SearchParams:           # SamplingParams, edm/main.py:35-43
    B: int = 2
    N: int = 4
    K: int = 20
    lambda_param: float = 0.15
    eps: float = 0.4
    S: int = 8
    scorer: Any = None


def sigma_schedule(net, num_steps=18, sigma_min=0.002, sigma_max=80, rho=7):
    idx = torch.arange(num_steps, dtype=torch.float64)
    t = (sigma_max ** (1 / rho) + idx / (num_steps - 1) * (sigma_min ** (1 / rho) - sigma_max ** (1 / rho))) ** rho
    return torch.cat([net.round_sigma(t), torch.zeros_like(t[:1])])


def to_uint8(x):
    """edm/main.py:126,661,827,869: scale, clip, truncating cast."""
    return (x * 127.5 + 128).clip(0, 255).to(torch.uint8)


class _Ctx:
    def __init__(self, net, num_steps, S_churn, S_min, S_max, S_noise):
        self.net, self.num_steps = net, num_steps
        self.S_churn, self.S_min, self.S_max, self.S_noise = S_churn, S_min, S_max, S_noise

    def heun_step(self, x_cur, t_cur, t_next, i, eps_i, labels):
        gamma = min(self.S_churn / self.num_steps, np.sqrt(2) - 1) if self.S_min <= t_cur <= self.S_max else 0
        t_hat = self.net.round_sigma(t_cur + gamma * t_cur)
        x_hat = x_cur + (t_hat ** 2 - t_cur ** 2).sqrt() * self.S_noise * eps_i
        denoised = self.net(x_hat, t_hat, labels).to(torch.float64)
        d_cur = (x_hat - denoised) / t_hat
        x_next = x_hat + (t_next - t_hat) * d_cur
        if i < self.num_steps - 1:
            denoised = self.net(x_next, t_next, labels).to(torch.float64)
            d_prime = (x_next - denoised) / t_next
            x_next = x_hat + (t_next - t_hat) * (0.5 * d_cur + 0.5 * d_prime)
        return x_next, denoised


def _zero_ts(n):
    return torch.zeros(n)


def _naive(ctx, t_steps, x_next, labels, p, pre, trace, scale_fn):
    for i in range(ctx.num_steps):
        eps_i = torch.randn_like(x_next)
        x_next, _ = ctx.heun_step(x_next, t_steps[i], t_steps[i + 1], i, eps_i, labels)
    return x_next


def _rejection(ctx, t_steps, x_next, labels, p, pre, trace, scale_fn):
    N, B = p.N, x_next.shape[0]
    x = x_next.repeat_interleave(N, dim=0)                      # b-major layout
    lab = labels.repeat_interleave(N, dim=0)
    for i in range(ctx.num_steps):
        if pre is not None and i in pre:
            eps_i = pre[i][:, :N].reshape(B * N, *x.shape[1:])
        else:
            eps_i = torch.randn_like(x)
        x, _ = ctx.heun_step(x, t_steps[i], t_steps[i + 1], i, eps_i, lab)
    scores = p.scorer(to_uint8(x), lab, _zero_ts(x.shape[0])).view(B, N)
    best = scores.argmax(dim=1)
    trace['rewards'].append(scores.clone())
    trace['selected'].append(best.clone())
    xr = x.view(B, N, *x.shape[1:])
    return torch.stack([xr[b, j] for b, j in enumerate(best)])


def _beam(ctx, t_steps, x_next, labels, p, pre, trace, scale_fn):
    b, k = p.b, p.k            # edm/main.py:140 -- SamplingParams has no such fields: AttributeError
    raise RuntimeError('unreachable')


def _eps_greedy(ctx, t_steps, x_next, labels, p, pre, trace, scale_fn):
    lam = p.lambda_param * np.sqrt(3 * 64 * 64)                 # scaled by 3*64*64 at every resolution (:716)
    N, K, eps, B = p.N, p.K, p.eps, x_next.shape[0]
    pivot = pre['pivot'] if (pre is not None and 'pivot' in pre) else torch.randn_like(x_next)
    for i in range(ctx.num_steps):
        t_cur, t_next = t_steps[i], t_steps[i + 1]
        x_cur = x_next
        pivot = pre[f'pivot_{i}'] if (pre is not None and f'pivot_{i}' in pre) else torch.randn_like(x_cur)
        for k in range(K):
            cands = []
            for n in range(N):
                if torch.rand(1) < (1 - eps):
                    if pre is not None and i in pre and k < pre[i].shape[1] and n < pre[i].shape[2]:
                        u = pre[i][:, k, n].reshape(pivot.shape)
                    else:
                        u = torch.randn_like(pivot)
                    dims = tuple(range(1, u.dim()))
                    u = u / torch.norm(u, p=2, dim=dims, keepdim=True)
                    shape = [u.shape[0]] + [1] * (u.dim() - 1)
                    scale = torch.ones(shape) * scale_fn(i, k, n) * lam        # float32 scale (:779)
                    cands.append(pivot + scale * u)
                else:
                    key = f'fresh_{i}_{k}_{n}'
                    cands.append(pre[key] if (pre is not None and key in pre) else torch.randn_like(x_cur))
            all_noises = torch.cat(cands, dim=0)                               # n-major layout
            x_exp = x_cur.repeat(N, 1, 1, 1)
            lab_exp = None if labels is None else labels.repeat(N, 1)
            _, x0 = ctx.heun_step(x_exp, t_cur, t_next, i, all_noises, lab_exp)
            scores = p.scorer(to_uint8(x0), lab_exp, _zero_ts(x0.shape[0])).reshape(N, B)
            best = scores.argmax(dim=0)
            trace['rewards'].append(scores.clone())
            trace['selected'].append(best.clone())
            nb = all_noises.reshape(N, B, *all_noises.shape[1:])
            pivot = torch.stack([nb[j, b] for b, j in enumerate(best)])
            trace['best_noises'].setdefault(i, []).append(pivot.clone())       # best_noises_this_timestep (:741, :854)
        x_next, _ = ctx.heun_step(x_cur, t_cur, t_next, i, pivot, labels)
    return x_next


class _Node:
    __slots__ = ('x', 'children', 'reward', 'visit')

    def __init__(self, x, visit=0):
        self.x, self.children, self.reward, self.visit = x, [], 0, visit


def _mcts(ctx, t_steps, x_next, labels, p, pre, trace, scale_fn):
    b, S, B, ns = p.N, p.S, x_next.shape[0], ctx.num_steps
    results = []
    mbs = min(2, B)
    for mb0 in range(0, B, mbs):
        xb = x_next[mb0:mb0 + mbs]
        lb = None if labels is None else labels[mb0:mb0 + mbs]
        m = xb.shape[0]
        noise = {}
        for i in range(ns):
            if pre is not None and i in pre:
                noise[i] = pre[i].repeat(m, 1, 1, 1, 1)
            else:
                noise[i] = torch.randn(m, b, *xb.shape[1:])          # float32 (edm/main.py:446)
        roots = [_Node(xb[s:s + 1].clone(), visit=1) for s in range(m)]
        lab1 = lambda s: None if lb is None else lb[s:s + 1]
        for i in range(ns):
            t_cur, t_next = t_steps[i], t_steps[i + 1]
            todo = [(s, j) for s in range(m) if not roots[s].children for j in range(b)]
            if todo:
                xe = torch.cat([roots[s].x for s, j in todo], dim=0)
                ne = torch.cat([noise[i][s:s + 1, j] for s, j in todo], dim=0)
                le = None if lb is None else torch.cat([lb[s:s + 1] for s, j in todo], dim=0)
                xn, _ = ctx.heun_step(xe, t_cur, t_next, i, ne, le)
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
                            _wasted = torch.randn(1, *xb.shape[1:])          # eager .get() default (:578-579)
                            e = noise[it][s, j:j + 1]
                            xc, _ = ctx.heun_step(node.x, tc, tn, it, e, lab1(s))
                            node.children.append(_Node(xc))
                        node = node.children[np.random.randint(0, len(node.children))]
                        it += 1
                        path.append(node)
                    paths.append(path)
                    starts.append((node.x.clone(), it, s))
                outs = []
                for x1, it, s in starts:
                    for j in range(it, ns):
                        x1, _ = ctx.heun_step(x1, t_steps[j], t_steps[j + 1], j, torch.zeros_like(x1), lab1(s))
                    outs.append(x1)
                den = torch.cat(outs, dim=0)
                sl = None if lb is None else torch.cat([lb[s:s + 1] for _, _, s in starts], dim=0)
                rew = p.scorer(to_uint8(den), sl, _zero_ts(den.shape[0]))
                trace['rewards'].append(rew.clone())
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
                trace['selected'].append(torch.tensor([best_j]))
                roots[s] = best                     # tree below the chosen child is kept (:702-703)
        results += [r.x for r in roots]
    return torch.cat(results, dim=0)


_METHODS: Dict[str, Callable] = {
    'naive': _naive, 'rejection': _rejection, 'beam': _beam, 'mcts': _mcts,
    'zero_order': _eps_greedy, 'eps_greedy': _eps_greedy,      # one shared branch (edm/main.py:714)
}


def builtin_hash_scale(i, k, n):
    """edm/main.py:776 -- depends on PYTHONHASHSEED."""
    return hash(f"{i}_{k}_{n}") % 1000 / 1000.0


@torch.no_grad()
def search(net, latents, class_labels, *, method='naive', params: Optional[Dict[str, Any]] = None,
           seed=0, num_steps=18, sigma_min=0.002, sigma_max=80, rho=7,
           S_churn=0, S_min=0, S_max=float('inf'), S_noise=1,
           precomputed_noise=None, scale_fn: Callable = builtin_hash_scale):
    """CPU counterpart of generate_image_grid (edm/main.py:47-886) minus pickle loading and PNG writing."""
    torch.manual_seed(seed)
    p = SearchParams(**(params or {}))
    t_steps = sigma_schedule(net, num_steps, sigma_min, sigma_max, rho)
    ctx = _Ctx(net, num_steps, S_churn, S_min, S_max, S_noise)
    x0 = latents.to(torch.float64) * t_steps[0]
    trace = dict(rewards=[], selected=[], best_noises={})
    x_next = _METHODS[method](ctx, t_steps, x0, class_labels, p, precomputed_noise, trace, scale_fn)
    image = to_uint8(x_next)
    scores = p.scorer(image.clone(), class_labels, _zero_ts(image.shape[0]))
    return dict(x=x_next, image=image, final_scores=scores, avg_score=scores.mean().item(),
                t_steps=t_steps, rewards=trace['rewards'], selected=trace['selected'],
                best_noises={i: torch.stack(v) for i, v in trace['best_noises'].items()})
