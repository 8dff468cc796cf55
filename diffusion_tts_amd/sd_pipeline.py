"""The SD backend's noise-trajectory-search loop (drop-in for the reference's modified
`StableDiffusionPipeline.__call__`, sd/diffusers/.../pipeline_stable_diffusion.py:785-1485, from step 4 on) with
candidate-BATCHED evaluation.

BASELINE config 4 keeps the denoiser and the decoder as the stock diffusers modules ("diffusers U-Net path"): here
`unet` and `vae` are opaque callables with the diffusers call surface
    unet(sample, t, encoder_hidden_states=..., return_dict=False)[0]      vae.decode(z, return_dict=False)[0]
What this build replaces is everything around them: the DDIM scheduler step fused over the N candidates that share one
(x, eps) (`dts_ddim_candidates`, K13), classifier-free-guidance combine (`dts_cfg_combine`), quantisation (K10), and
the loop structure -- the reference evaluates the N candidates of an iteration ONE AT A TIME (batch 2 through the
U-Net, batch 1 through the VAE: pipeline...:1082-1129, 1383-1429); here all N go through the U-Net as one 2N-row call
and through the VAE as one N-row call.  Values are unchanged: rows are independent.

Host RNG: torch's CPU generator in the reference's call order (the pipeline's `--device cpu` run is the parity
oracle), including the variance-noise draws `scheduler.step` makes and drops when called without `variance_noise`
(scheduling_ddim.py:457-460).  Candidate construction (normalise + axpy, pipeline...:1371-1379) is a few KB of f32
arithmetic on the host, bit-identical to the reference, then uploaded.

Reference quirks kept: SD MCTS never scores or back-propagates (pipeline...:1201-1313), so every timestep takes the
FIRST expanded child; its rollouts only consume RNG.  The draws are reproduced; the dead U-Net work is skipped unless
`mcts_dead_compute=True`.  `mcts_backprop=True` (off by default: the default stays reference-faithful) runs the search the
reference's code sets out to do -- UCB1 selection, one expansion per simulation, a stochastic DDIM rollout to the last
timestep, decode + score, visit/reward back-propagation, best-mean child.

Multi-GPU (one process per GPU, torch.distributed over RCCL; parallel.CandidateShards): every rank replays the same
host RNG, so the candidate noises are known everywhere.  eps-greedy / zero-order shard the N candidates of an iteration, beam shards the
B*N candidates of a timestep; each rank runs the U-Net, the VAE decode and the scorer on its own candidates only and ONE all-gather of
the rewards per iteration (eps-greedy) / per timestep (beam) follows.  Survivors never travel: the eps-greedy survivor is a host noise
tensor, and a beam survivor's latent is one fused DDIM step of replicated inputs (the beam, its noise prediction, the candidate noise),
which every rank computes for all B*N candidates anyway (32 KB each) -- so no broadcast is needed and the collective count is exactly one
per search decision.  SD MCTS (reference-faithful default) never scores, so there is nothing to shard; `mcts_backprop` runs replicated.

Prompt handling (pipeline...:812-814, 976-992, `encode_prompt` :330-460): with a `text_encoder` / `tokenizer` pair the
pipeline encodes `prompt` and `negative_prompt` itself exactly as `encode_prompt` does; `prompt_embeds` /
`negative_prompt_embeds` may be passed instead.  Missing `latents` are drawn like `prepare_latents` (:671-690).
"""
import copy
import types
from typing import Callable, Optional

import numpy as np
import torch

from . import ops
from .parallel import CandidateShards


class DDIMScheduler:
    """Host mirror of the modified DDIMScheduler (scheduling_ddim.py): alpha table, `set_timesteps` ('leading' spacing),
    `_get_variance`, and `step` routed to the fused HIP kernel.  Defaults = SD-1.5's scheduler_config.json."""

    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, steps_offset=1, set_alpha_to_one=False):
        betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
        self.final_alpha_cumprod = torch.tensor(1.0) if set_alpha_to_one else self.alphas_cumprod[0]
        self.num_train_timesteps, self.steps_offset = num_train_timesteps, steps_offset
        self.num_inference_steps = None
        self.init_noise_sigma = 1.0
        self.order = 1

    def set_timesteps(self, num_inference_steps, device=None):
        self.num_inference_steps = num_inference_steps
        ratio = self.num_train_timesteps // num_inference_steps
        ts = (np.arange(0, num_inference_steps) * ratio).round()[::-1].copy().astype(np.int64) + self.steps_offset
        self.timesteps = torch.from_numpy(ts)
        return self.timesteps

    def scale_model_input(self, sample, timestep=None):
        return sample

    def coefficients(self, timestep, eta=1.0):
        """(alpha_t, alpha_prev, sigma_t) as f32 scalars (scheduling_ddim.py:403-405, 253-261, 431)."""
        timestep = int(timestep)
        prev = timestep - self.num_train_timesteps // self.num_inference_steps
        a_t = self.alphas_cumprod[timestep]
        a_p = self.alphas_cumprod[prev] if prev >= 0 else self.final_alpha_cumprod
        var = ((1 - a_p) / (1 - a_t)) * (1 - a_t / a_p)
        return float(a_t), float(a_p), float(eta * var ** 0.5)

    def step(self, model_output, timestep, sample, eta=1.0, variance_noise=None, return_dict=False, generator=None):
        """(prev_sample, pred_original_sample); variance_noise may hold N candidates ([N, *sample.shape]) -> N prev samples."""
        a_t, a_p, sig = self.coefficients(timestep, eta)
        if variance_noise is None and eta > 0:
            variance_noise = torch.randn(model_output.shape, dtype=sample.dtype).to(sample.device)   # randn_tensor(..., dtype=model_output.dtype) on the CPU run (:457-460)
        z = variance_noise
        if z is not None and z.dim() == sample.dim():
            z = z.unsqueeze(0)
        prev, x0 = ops.ddim_candidates(sample.contiguous(), model_output.contiguous(), None if z is None else z.contiguous(), a_t, a_p, sig)
        if z is None or variance_noise.dim() == sample.dim():
            prev = prev[0]
        return prev, x0


class _MCTSNode:
    __slots__ = ('latents', 'children', 'parent', 'visits', 'total_reward', 'depth')

    def __init__(self, latents, parent=None):
        self.latents, self.children, self.parent = latents, [], parent
        self.visits, self.total_reward = 0, 0.0
        self.depth = 0 if parent is None else parent.depth + 1

    def ucb(self, c):                              # pipeline...:1185-1197
        if self.visits == 0:
            return float('inf')
        pv = self.parent.visits if self.parent else 1
        return self.total_reward / self.visits + c * np.sqrt(np.log(pv) / self.visits)


class SDSearchPipeline:
    def __init__(self, unet, vae, scheduler: Optional[DDIMScheduler] = None, device='cuda', mcts_dead_compute=False,
                 text_encoder=None, tokenizer=None, mcts_backprop=False, shards: Optional[CandidateShards] = None):
        """shards: candidate sharding over the ranks of the process group (default: CandidateShards(), a world of one outside
        torch.distributed); pass CandidateShards(enabled=False) to run the whole search on every rank."""
        self.shards = shards if shards is not None else CandidateShards()
        self.unet, self.vae = unet, vae
        self.scheduler = scheduler or DDIMScheduler()
        self.device = torch.device(device)
        if self.device.type != 'cuda':
            raise RuntimeError('SDSearchPipeline (HIP) needs a GPU device')
        self.mcts_dead_compute = mcts_dead_compute
        self.mcts_backprop = mcts_backprop
        self.text_encoder, self.tokenizer = text_encoder, tokenizer
        self.unet_rows = 0
        self.decoded = 0                              # images through the VAE decoder on THIS rank
        self.scorer_calls = 0

    @torch.no_grad()
    def encode_prompt(self, prompt, negative_prompt=None):
        """(prompt_embeds, negative_prompt_embeds) as `StableDiffusionPipeline.encode_prompt` computes them for one image per
        prompt with classifier-free guidance (pipeline...:382-460): prompt padded to the tokenizer's model_max_length, the
        negative prompt ("" by default) padded to the same length, attention mask only if the text encoder asks for one."""
        if self.text_encoder is None or self.tokenizer is None:
            raise ValueError('pass prompt_embeds / negative_prompt_embeds, or build the pipeline with text_encoder= and tokenizer=')
        tok, te = self.tokenizer, self.text_encoder
        dev = self.device
        prompts = [prompt] if isinstance(prompt, str) else list(prompt)

        def run(texts, max_length):
            enc = tok(texts, padding='max_length', max_length=max_length, truncation=True, return_tensors='pt')
            mask = enc.attention_mask.to(dev) if getattr(getattr(te, 'config', None), 'use_attention_mask', False) else None
            return te(enc.input_ids.to(dev), attention_mask=mask)[0]
        pe = run(prompts, tok.model_max_length)
        neg = [''] * len(prompts) if negative_prompt is None else ([negative_prompt] * len(prompts) if isinstance(negative_prompt, str) else list(negative_prompt))
        if len(neg) != len(prompts):
            raise ValueError(f'negative_prompt has batch size {len(neg)}, prompt {len(prompts)}')
        ne = run(neg, pe.shape[1])
        dt = getattr(te, 'dtype', pe.dtype)
        return pe.to(dt), ne.to(dt)

    # ------------------------------------------------------------------------------------------
    def _eps(self, x, t, embeds_u, embeds_c, guidance):
        """CFG noise prediction for a batch of latents: one 2n-row U-Net call (reference: n calls of 2 rows)."""
        n = x.shape[0]
        ehs = torch.cat([embeds_u.expand(n, -1, -1), embeds_c.expand(n, -1, -1)])
        out = self.unet(torch.cat([x, x]), t, encoder_hidden_states=ehs, return_dict=False)[0].contiguous()
        self.unet_rows += 2 * n
        return ops.cfg_combine(out[:n].contiguous(), out[n:].contiguous(), guidance)

    def _score(self, score_function, x0, prompt, scores=None):
        """decode + quantise the batch once, then score it: ONE scorer call for the whole batch when the scorer says it takes batches
        (`scorer.batched`: this package's CLIP / brightness scorers), else the reference's per-image calls (pipeline...:1114)."""
        image = self.vae.decode(x0 / self.vae.config.scaling_factor, return_dict=False)[0].float().contiguous()
        u8 = ops.quantize_u8(image, f32_math=True)
        n = u8.shape[0]
        self.decoded += n
        if n > 1 and getattr(score_function, 'batched', False):
            s = score_function(images=[u8[j:j + 1] for j in range(n)], prompts=[prompt], timesteps=None)
            self.scorer_calls += 1
            vals = [float(v) for v in (s.float().cpu().tolist() if torch.is_tensor(s) else s)]
        else:
            vals = []
            for j in range(n):
                s = score_function(images=[u8[j:j + 1]], prompts=[prompt], timesteps=None)
                self.scorer_calls += 1
                vals.append(s.item() if torch.is_tensor(s) else float(s))
        if scores is not None:
            scores += vals
        return vals

    def _gather(self, local_vals, n_total):
        """rewards of this rank's candidates [lo, hi) -> the rewards of all n_total candidates, identical on every rank (one all-gather)."""
        if self.shards.solo:
            return list(local_vals)
        t = torch.tensor(local_vals, dtype=torch.float64, device=self.device if self.shards.dev_direct else 'cpu')
        return self.shards.gather_rewards(t, n_total, 1).cpu().tolist()

    def _evaluate(self, noise_pred, t, x, cands_dev, eu, ec, g, eta, score_function, prompt, scores):
        """candidates [N,1,C,H,W] -> (latents_cand [N,C,H,W], rewards of all N): steps 1083-1114 of the reference for all N at once.
        Sharded: the first DDIM step runs for all N on every rank (replicated inputs, 32 KB per candidate: this is what makes every
        candidate latent available everywhere), the U-Net / decode / score only for this rank's candidates [lo, hi)."""
        a_t, a_p, sig = self.scheduler.coefficients(t, eta)
        prev, _ = ops.ddim_candidates(x.contiguous(), noise_pred.contiguous(), cands_dev.contiguous(), a_t, a_p, sig, want_x0=False)
        lat_c = prev.reshape(prev.shape[0], *x.shape[1:])
        n_total = lat_c.shape[0]
        lo, hi = self.shards.span(n_total)
        mine = lat_c[lo:hi].contiguous()
        vals = []
        if hi > lo:
            np2 = self._eps(mine, t, eu, ec, g)                                # same t, not t-1 (:1090)
            _, x0 = ops.ddim_candidates(mine, np2.contiguous(), None, a_t, a_p, sig)
            vals = self._score(score_function, x0, prompt)
        allv = self._gather(vals, n_total)
        scores += allv
        return lat_c, allv

    def _up(self, t, dtype):
        return t.to(self.device, dtype).contiguous()

    def _mcts_backprop(self, timesteps, latents, eu, ec, g, eta, score_function, prompt, scores, params, randn, dtype):
        """The tree search the reference's MCTS branch sets out to do but never completes (pipeline...:1201-1313 builds the nodes
        and runs the rollouts, then drops `pred_x0`: no decode, no score, `visits` / `total_reward` never updated, so `best_child`
        is always the first child).  Behind `mcts_backprop=True` only.  Differences from the reference code, all required for the
        statistics to mean anything: a node at depth d is expanded and rolled out from timestep index i+d (the reference uses
        the root's `t` and `range(i, ...)` for every node); the rollout's final latents are decoded and scored; the reward is
        added along the path to the root.  Host RNG order: per simulation, the expansion noise (if any), then one variance
        noise per rollout step."""
        sch, S, N, c = self.scheduler, params['S'], params['N'], params.get('c', 1.414)
        nT = len(timesteps)
        best_reward = None
        for i in range(nT):
            root = _MCTSNode(latents)
            root.visits = 1
            for _ in range(S):
                node = root
                while node.children and all(ch.visits > 0 for ch in node.children) and len(node.children) >= N:
                    node = max(node.children, key=lambda ch: ch.ucb(c))
                ti = i + node.depth
                if ti < nT and len(node.children) < N:                         # expansion
                    t = timesteps[ti]
                    child_lat, _ = sch.step(self._eps(node.latents, t, eu, ec, g), t, node.latents, eta,
                                            variance_noise=self._up(randn(), dtype))
                    node = _MCTSNode(child_lat, parent=node)
                    node.parent.children.append(node)
                tmp = node.latents
                for j in range(i + node.depth, nT):                            # stochastic DDIM rollout to the last timestep
                    tj = timesteps[j]
                    tmp, _ = sch.step(self._eps(tmp, tj, eu, ec, g), tj, tmp, eta, variance_noise=self._up(randn(), dtype))
                r = self._score(score_function, tmp, prompt, scores)[0]
                best_reward = r if best_reward is None else max(best_reward, r)
                while node is not None:                                        # back-propagation
                    node.visits += 1
                    node.total_reward += r
                    node = node.parent
            kids = [ch for ch in root.children if ch.visits > 0]
            if kids:
                latents = max(kids, key=lambda ch: ch.total_reward / ch.visits).latents
        self._mcts_latents = latents
        return best_reward

    # ------------------------------------------------------------------------------------------
    @torch.no_grad()
    def __call__(self, prompt=None, num_inference_steps=100, guidance_scale=7.5, negative_prompt=None, eta=1.0, latents=None,
                 prompt_embeds=None, negative_prompt_embeds=None, score_function: Optional[Callable] = None, method='eps_greedy',
                 params=None, output_type='pt', height=None, width=None):
        """Keyword surface of the reference call (pipeline...:785-817) for the arguments the search path uses:
        `pipe(prompt=..., num_inference_steps=..., score_function=..., method=..., params=...) -> (output, max_score)`."""
        if prompt is not None and prompt_embeds is not None:                  # check_inputs (:657-661)
            raise ValueError(f'Cannot forward both `prompt`: {prompt} and `prompt_embeds`. Please make sure to only forward one of the two.')
        if prompt_embeds is None or negative_prompt_embeds is None:
            if prompt is None:
                raise ValueError('pass `prompt` (with a text_encoder/tokenizer in the pipeline) or prompt_embeds + negative_prompt_embeds')
            pe, ne = self.encode_prompt(prompt, negative_prompt)
            prompt_embeds = pe if prompt_embeds is None else prompt_embeds
            negative_prompt_embeds = ne if negative_prompt_embeds is None else negative_prompt_embeds
        if isinstance(prompt, list):
            if len(prompt) != 1:
                raise ValueError('the search loop runs one prompt per call (as main.py:135 of the reference does)')
            prompt = prompt[0]
        sch, dev = self.scheduler, self.device
        dtype = getattr(self.unet, 'dtype', torch.float32)
        eu, ec = self._up(negative_prompt_embeds, dtype), self._up(prompt_embeds, dtype)
        timesteps = sch.set_timesteps(num_inference_steps)
        if latents is None:                                                   # prepare_latents (:671-690) on the reference's CPU run
            cfgu = self.unet.config
            vsf = 2 ** (len(getattr(self.vae.config, 'block_out_channels', [0] * 4)) - 1)
            hh = (height // vsf) if height else cfgu.sample_size
            ww = (width // vsf) if width else cfgu.sample_size
            latents = torch.randn((1, cfgu.in_channels, hh, ww), dtype=dtype)
        latents = self._up(latents * sch.init_noise_sigma, dtype)
        shape = tuple(latents.shape)
        scores, g = [], float(guidance_scale)
        self.unet_rows = self.decoded = self.scorer_calls = 0
        coll0 = self.shards.collectives
        # randn_like(latents) on the reference's CPU run draws in the LATENTS' dtype (fp16 latents -> fp16 normal sampler, a
        # different stream from f32 draws rounded to fp16); the candidate arithmetic (:1371-1379) runs in that dtype too
        randn = lambda: torch.randn(shape, dtype=dtype)
        max_score = None

        if method == 'beam':
            B, N = params['B'], params['N']
            self.shards.require_candidates(B * N, 'SD beam search')
            best = [copy.deepcopy(latents) for _ in range(B)]
            a_eta = eta
            for i, t in enumerate(timesteps):
                # the B beams' noise predictions in ONE 2B-row U-Net call (the reference makes B calls of 2 rows: rows are independent)
                noise_preds = self._eps(torch.cat(best), t, eu, ec, g)
                a_t, a_p, sig = sch.coefficients(t, a_eta)
                lat_all = []
                for b_, beam in enumerate(best):
                    cands = torch.stack([randn() for _ in range(N)])           # :1080, per beam, in the reference's order
                    for _ in range(N):
                        randn()                                                # dropped variance noise of the second step (:1109)
                    prev, _ = ops.ddim_candidates(beam.contiguous(), noise_preds[b_:b_ + 1].contiguous(), self._up(cands, dtype), a_t, a_p, sig,
                                                  want_x0=False)
                    lat_all.append(prev.reshape(N, *beam.shape[1:]))
                lat_all = torch.cat(lat_all)                                   # [B*N, C, H, W], beam-major like the reference's list
                lo, hi = self.shards.span(B * N)
                vals = []
                if hi > lo:
                    mine = lat_all[lo:hi].contiguous()
                    np2 = self._eps(mine, t, eu, ec, g)
                    _, x0 = ops.ddim_candidates(mine, np2.contiguous(), None, a_t, a_p, sig)
                    vals = self._score(score_function, x0, prompt)
                vals = self._gather(vals, B * N)                               # ONE all-gather per timestep
                scores += vals
                order = sorted(range(len(vals)), key=lambda k: vals[k], reverse=True)      # stable: first wins ties (:1132)
                best = [lat_all[k:k + 1].clone() for k in order[:B]]           # survivors: already on every rank
            max_score, latents = float('-inf'), best[0]
            finals = self._score(score_function, torch.cat(best), prompt, scores)          # :1157-1170 (B images: replicated)
            for lat_c, v in zip(best, finals):
                if v > max_score:
                    max_score, latents = v, lat_c
        elif method == 'mcts' and self.mcts_backprop:
            max_score = self._mcts_backprop(timesteps, latents, eu, ec, g, eta, score_function, prompt, scores, params, randn, dtype)
            latents = self._mcts_latents
        elif method == 'mcts':
            for i, t in enumerate(timesteps):
                children = []
                for _ in range(params['S']):
                    node = latents
                    if len(children) < params['N']:
                        noise_pred = self._eps(node, t, eu, ec, g)
                        child, _ = sch.step(noise_pred, t, node, eta, variance_noise=self._up(randn(), dtype))
                        children.append(child)
                        node = child
                    tmp = node
                    if self.mcts_dead_compute:
                        self._eps(node, t, eu, ec, g)
                    for j in range(i, len(timesteps)):                        # rollout: consumes RNG, result never used
                        z = randn()
                        if self.mcts_dead_compute:
                            tmp, _ = sch.step(self._eps(tmp, timesteps[j], eu, ec, g), timesteps[j], tmp, eta, variance_noise=self._up(z, dtype))
                if children:
                    latents = children[0]
        else:
            for i, t in enumerate(timesteps):
                noise_pred = self._eps(latents, t, eu, ec, g)
                pivot = randn()                                                # :1366 (host copy; f32)
                if method in ('eps_greedy', 'zero_order'):
                    N = params['N']
                    self.shards.require_candidates(N, 'SD eps-greedy / zero-order search')
                    thr = params['eps'] if method == 'eps_greedy' else 0.0
                    pivot_d = self._up(pivot, dtype)
                    for _ in range(params['K']):
                        # host RNG in the reference's call order (:1371-1379): the coin, the Gaussian, the step length; the candidate
                        # ARITHMETIC (normalise, scale, add the pivot) runs on the device in the latents' dtype (dts_candidate_noise_sd)
                        draws, mode, scale = [], [], []
                        for _ in range(N):
                            r = torch.rand(1).item()
                            if r < thr:
                                draws.append(randn())
                                mode.append(0)
                                scale.append((0.0, 0.0, 0.0))
                            else:
                                draws.append(randn())
                                mode.append(1)
                                scale.append((torch.rand(1).item(), float(params['lambda']), float(np.sqrt(shape[-1] * shape[-2] * shape[-3]))))
                        for _ in range(N):
                            randn()                                            # dropped variance noise (:1410)
                        cands = ops.candidate_noise_sd(pivot_d, self._up(torch.stack(draws), dtype), torch.tensor(mode, dtype=torch.int32, device=dev),
                                                       torch.tensor(scale, dtype=torch.float32, device=dev))
                        _, vals = self._evaluate(noise_pred, t, latents, cands, eu, ec, g, eta, score_function, prompt, scores)
                        max_score = max(vals)
                        pivot_d = cands[vals.index(max_score)].contiguous()    # first max (:1432-1433)
                    pivot = pivot_d
                latents, _ = sch.step(noise_pred, t, latents, eta, variance_noise=self._up(pivot, dtype))
        image = self.vae.decode(latents / self.vae.config.scaling_factor, return_dict=False)[0]
        if max_score is None:
            u8 = ops.quantize_u8(image.float().contiguous(), f32_math=True)
            max_score = score_function(images=[u8], prompts=[prompt], timesteps=None)
            scores.append(max_score.item() if torch.is_tensor(max_score) else float(max_score))
        if output_type == 'latent':
            images = latents
        elif output_type == 'pt':
            images = (image / 2 + 0.5).clamp(0, 1)
        else:
            import PIL.Image
            arr = ((image / 2 + 0.5).clamp(0, 1).permute(0, 2, 3, 1).float().cpu().numpy() * 255).round().astype('uint8')
            images = [PIL.Image.fromarray(a) for a in arr]
        out = types.SimpleNamespace(images=images, nsfw_content_detected=None, latents=latents, scores=scores, unet_rows=self.unet_rows,
                                    decoded=self.decoded, scorer_calls=self.scorer_calls, collectives=self.shards.collectives - coll0)
        return out, max_score
