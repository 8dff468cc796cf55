"""Oracle: the SD backend's search loop and DDIM step on CPU (torch CPU, fp32).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restates the modified parts of the vendored diffusers (sd/diffusers/src/diffusers/...):
  schedulers/scheduling_ddim.py:253-261   _get_variance
  schedulers/scheduling_ddim.py:297-340   set_timesteps ('leading' spacing, steps_offset)
  schedulers/scheduling_ddim.py:342-471   step (eta default 1.0; returns (prev_sample, pred_original_sample))
  pipelines/stable_diffusion/pipeline_stable_diffusion.py:1045-1170  beam
  pipelines/stable_diffusion/pipeline_stable_diffusion.py:1172-1333  mcts (never back-propagates: SURVEY.md 3.3)
  pipelines/stable_diffusion/pipeline_stable_diffusion.py:1335-1455  naive / zero_order / eps_greedy
  pipelines/stable_diffusion/pipeline_stable_diffusion.py:1457-1485  final decode + score
The U-Net and the VAE are opaque callables (diffusers modules in the reference).  RNG: the torch global CPU
generator in the reference's call order, including the `variance_noise` draws whose values are never used
(scheduling_ddim.py:457-460 when `step` is called without `variance_noise`).
"""
import copy
import math

import numpy as np
import torch


class DDIMOracle:
    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, steps_offset=1, set_alpha_to_one=False):
        betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2   # scaled_linear
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
        self.final_alpha_cumprod = torch.tensor(1.0) if set_alpha_to_one else self.alphas_cumprod[0]
        self.num_train_timesteps, self.steps_offset = num_train_timesteps, steps_offset
        self.num_inference_steps = None
        self.init_noise_sigma = 1.0

    def set_timesteps(self, n):
        self.num_inference_steps = n
        ratio = self.num_train_timesteps // n
        ts = (np.arange(0, n) * ratio).round()[::-1].copy().astype(np.int64) + self.steps_offset
        self.timesteps = torch.from_numpy(ts)
        return self.timesteps

    def coefficients(self, timestep, eta=1.0):
        prev = timestep - self.num_train_timesteps // self.num_inference_steps
        a_t = self.alphas_cumprod[timestep]
        a_p = self.alphas_cumprod[prev] if prev >= 0 else self.final_alpha_cumprod
        var = ((1 - a_p) / (1 - a_t)) * (1 - a_t / a_p)
        return a_t, a_p, eta * var ** 0.5

    def step(self, model_output, timestep, sample, eta=1.0, variance_noise=None):
        a_t, a_p, std = self.coefficients(int(timestep), eta)
        x0 = (sample - (1 - a_t) ** 0.5 * model_output) / a_t ** 0.5
        direction = (1 - a_p - std ** 2) ** 0.5 * model_output
        prev = a_p ** 0.5 * x0 + direction
        if eta > 0:
            if variance_noise is None:
                variance_noise = torch.randn(model_output.shape, dtype=model_output.dtype)   # drawn even if the caller drops `prev`
            prev = prev + std * variance_noise
        return prev, x0


def to_u8(image):
    return (image * 127.5 + 128).clip(0, 255).to(torch.uint8)


@torch.no_grad()
def sd_search(unet, vae, sched: DDIMOracle, prompt_embeds, negative_prompt_embeds, latents, *, num_inference_steps,
              score_function, method, params, guidance_scale=7.5, eta=1.0, prompt=None):
    """Counterpart of StableDiffusionPipeline.__call__ from step 4 on (prompt already embedded)."""
    embeds = torch.cat([negative_prompt_embeds, prompt_embeds])
    timesteps = sched.set_timesteps(num_inference_steps)
    latents = latents * sched.init_noise_sigma
    sf = vae.config.scaling_factor
    trace = dict(scores=[], unet_rows=0)

    def eps_theta(x, t):
        out = unet(torch.cat([x] * 2), t, encoder_hidden_states=embeds, return_dict=False)[0]
        trace['unet_rows'] += 2 * x.shape[0]
        u, c = out.chunk(2)
        return u + guidance_scale * (c - u)

    def score_latents(x0):
        image = vae.decode(x0 / sf, return_dict=False)[0]
        s = score_function(images=[to_u8(image)], prompts=[prompt], timesteps=None)
        v = s.item() if torch.is_tensor(s) else float(s)
        trace['scores'].append(v)
        return v

    def evaluate(noise_pred, t, x, cand):
        lat_c, _ = sched.step(noise_pred, t, x, eta, variance_noise=cand)
        np2 = eps_theta(lat_c, t)                                  # same t, not t-1 (pipeline...:1090)
        _, x0 = sched.step(np2, t, lat_c, eta)                     # draws an unused variance noise
        return lat_c, score_latents(x0)

    max_score = None
    if method == 'beam':
        best = [copy.deepcopy(latents) for _ in range(params['B'])]
        for i, t in enumerate(timesteps):
            scored, scores = [], []
            for beam in best:
                noise_pred = eps_theta(beam, t)
                cands = [torch.randn_like(beam) for _ in range(params['N'])]
                for cand in cands:
                    lat_c, v = evaluate(noise_pred, t, beam, cand)
                    scored.append(lat_c)
                    scores.append(v)
            order = sorted(range(len(scores)), key=lambda k: scores[k], reverse=True)
            best = [scored[k] for k in order[:params['B']]]
        max_score = float('-inf')
        latents = best[0]
        for lat_c in best:
            v = score_latents(lat_c)                               # decodes the latent itself (:1159)
            if v > max_score:
                max_score, latents = v, lat_c
    elif method == 'mcts':
        for i, t in enumerate(timesteps):
            children = []
            for _ in range(params['S']):
                node = latents                                      # visits are never updated: selection never descends
                if len(children) < params['N']:
                    noise_pred = eps_theta(node, t)
                    noise = torch.randn_like(node)
                    child, _ = sched.step(noise_pred, t, node, eta, variance_noise=noise)
                    children.append(child)
                    node = child
                eps_theta(node, t)                                  # "simulation" U-Net call, result unused
                tmp = node.clone()
                for j in range(i, len(timesteps)):
                    tmp, _ = sched.step(eps_theta(tmp, timesteps[j]), timesteps[j], tmp, eta)   # fresh noise each step
            if children:
                latents = children[0]                               # max() over all -inf keys returns the first child
    else:
        for i, t in enumerate(timesteps):
            noise_pred = eps_theta(latents, t)
            pivot = torch.randn_like(latents)
            if method in ('eps_greedy', 'zero_order'):
                for _ in range(params['K']):
                    cands = []
                    for _ in range(params['N']):
                        r = torch.rand(1).item()
                        if r < (params['eps'] if method == 'eps_greedy' else 0.0):
                            cands.append(torch.randn_like(latents))
                        else:
                            u = torch.randn_like(latents)
                            u = u / torch.norm(u)
                            cands.append(pivot + u * torch.rand(1).item() * params['lambda'] *
                                         np.sqrt(latents.shape[-1] * latents.shape[-2] * latents.shape[-3]))
                    vals = [evaluate(noise_pred, t, latents, c)[1] for c in cands]
                    max_score = max(vals)
                    pivot = cands[vals.index(max_score)]            # dict insertion order => first max
            latents, _ = sched.step(noise_pred, t, latents, eta, variance_noise=pivot)
    image = vae.decode(latents / sf, return_dict=False)[0]
    if max_score is None:
        max_score = score_function(images=[to_u8(image)], prompts=[prompt], timesteps=None)
        trace['scores'].append(max_score.item() if torch.is_tensor(max_score) else float(max_score))
    return dict(latents=latents, image=image, max_score=max_score, scores=trace['scores'], unet_rows=trace['unet_rows'])
