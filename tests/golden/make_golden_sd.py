#!/usr/bin/env python3
"""Golden vectors for the SD backend's search loop, produced by running the REFERENCE's modified
StableDiffusionPipeline.__call__ / DDIMScheduler (vendored diffusers, CPU, fp32) around the tiny stand-in U-Net / VAE
of tests/sd_standins.py.  Run: PYTHONHASHSEED=0 python tests/golden/make_golden_sd.py  (needs /root/reference).
Only arrays / scalars are written (tests/golden/sd_golden.npz + sd_manifest.json)."""
import importlib.util
import json
import os
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

import numpy as np
import torch

torch.set_num_threads(4)
import transformers
import transformers.utils
transformers.utils.FLAX_WEIGHTS_NAME = 'flax_model.msgpack'          # name removed in transformers 5.x
from transformers import CLIPModel, CLIPProcessor, ViTForImageClassification, ViTImageProcessor  # noqa: F401,E402

REF = os.environ.get('DTS_REFERENCE', '/root/reference')
spec = importlib.util.spec_from_file_location('diffusers', os.path.join(REF, 'sd/diffusers/src/diffusers/__init__.py'))
sys.modules['diffusers'] = importlib.util.module_from_spec(spec)
spec.loader.exec_module(sys.modules['diffusers'])                    # exactly as main.py:48-51 of the reference
from diffusers import StableDiffusionPipeline, DDIMScheduler         # noqa: E402

for _n in ('torchvision', 'torchvision.models', 'torchvision.transforms'):
    sys.modules[_n] = types.ModuleType(_n)
sys.modules['torchvision'].models = sys.modules['torchvision.models']
sys.modules['torchvision'].transforms = sys.modules['torchvision.transforms']
sys.path.insert(0, os.path.join(REF, 'sd'))
import scorers as ref_sd_scorers                                      # noqa: E402
from sd_standins import TinyUNet, TinyVAE                             # noqa: E402

CASES = {
    'naive': {}, 'eps_greedy': dict(N=3, K=2, eps=0.4), 'zero_order': {'N': 3, 'K': 2, 'eps': 0.4, 'lambda': 2.0},
    'beam': dict(B=2, N=2), 'mcts': dict(N=2, S=3),
}
STEPS = 4
MARGIN = 4e-5


class ScoreLogger:
    def __init__(self, scorer):
        self.scorer, self.calls = scorer, []

    def __call__(self, images, prompts, timesteps):
        s = self.scorer(images, prompts, timesteps)
        self.calls.append(float(s.item() if torch.is_tensor(s) else s))
        return s


class UNetCounter(torch.nn.Module):
    def __init__(self, unet):
        super().__init__()
        self.inner, self.config, self.rows = unet, unet.config, 0

    @property
    def dtype(self):
        return self.inner.dtype

    @property
    def device(self):
        return self.inner.device

    def forward(self, sample, *a, **k):
        self.rows += sample.shape[0]
        return self.inner(sample, *a, **k)


def main():
    out, manifest = {}, {'torch': torch.__version__, 'steps': STEPS, 'cases': {}}
    sched = DDIMScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule='scaled_linear', clip_sample=False,
                          set_alpha_to_one=False, steps_offset=1, num_train_timesteps=1000)     # SD-1.5 scheduler_config.json
    unet, vae = UNetCounter(TinyUNet()), TinyVAE()
    pipe = StableDiffusionPipeline(vae=vae, text_encoder=None, tokenizer=None, unet=unet, scheduler=sched, safety_checker=None,
                                   feature_extractor=None, requires_safety_checker=False)
    g = torch.Generator().manual_seed(3)
    pe, ne = torch.randn(1, 5, 8, generator=g), torch.randn(1, 5, 8, generator=g)
    lat = torch.randn(1, 4, 8, 8, generator=g)
    out['prompt_embeds'], out['negative_prompt_embeds'], out['latents'] = pe.numpy(), ne.numpy(), lat.numpy()
    # DDIM known answers (G7): alpha table samples, timesteps, one step with and without variance noise
    sched.set_timesteps(50)
    out['ddim_timesteps_50'] = sched.timesteps.numpy()
    out['ddim_alphas_cumprod'] = sched.alphas_cumprod.numpy()
    sched.set_timesteps(STEPS)
    x, e, z = torch.randn(1, 4, 8, 8, generator=g), torch.randn(1, 4, 8, 8, generator=g), torch.randn(1, 4, 8, 8, generator=g)
    for t in sched.timesteps.tolist():
        prev, x0 = sched.step(e, t, x, variance_noise=z, return_dict=False)
        out[f'ddim_step_{t}_prev'], out[f'ddim_step_{t}_x0'] = prev.numpy(), x0.numpy()
    out['ddim_x'], out['ddim_e'], out['ddim_z'] = x.numpy(), e.numpy(), z.numpy()
    for method, p in CASES.items():
        params = {'N': 4, 'lambda': 0.15, 'eps': 0.4, 'K': 20, 'B': 2, 'S': 8}
        params.update(p)
        for seed in range(64):
            torch.manual_seed(seed)
            sl = ScoreLogger(ref_sd_scorers.BrightnessScorer())
            unet.rows = 0
            res, score = pipe(prompt=None, prompt_embeds=pe, negative_prompt_embeds=ne, latents=lat.clone(), num_inference_steps=STEPS,
                              score_function=sl, method=method, params=params, output_type='pt')
            gaps = []
            if method in ('eps_greedy', 'zero_order'):
                v = np.array(sl.calls).reshape(-1, params['N'])
                s = np.sort(v, axis=1)[:, ::-1]
                gaps = (s[:, 0] - s[:, 1]).tolist()
            if method == 'beam':
                v = np.array(sl.calls[:-params['B']]).reshape(STEPS, -1)
                s = np.sort(v, axis=1)[:, ::-1]
                gaps = (s[:, params['B'] - 1] - s[:, params['B']]).tolist() + (s[:, 0] - s[:, 1]).tolist()
            if all(x_ == 0 or x_ >= MARGIN for x_ in gaps):
                break
        else:
            raise RuntimeError(f'{method}: no seed with safe margins')
        out[f'{method}_image'] = res.images.numpy()              # postprocessed: (decode/2 + 0.5).clamp(0, 1)
        out[f'{method}_scores'] = np.array(sl.calls)
        manifest['cases'][method] = dict(params=params, seed=seed, unet_rows=unet.rows, scorer_calls=len(sl.calls),
                                         max_score=float(score.item() if torch.is_tensor(score) else score),
                                         min_gap=min([x_ for x_ in gaps if x_ > 0], default=None))
        print(method, manifest['cases'][method], flush=True)
    np.savez_compressed(os.path.join(HERE, 'sd_golden.npz'), **out)
    json.dump(manifest, open(os.path.join(HERE, 'sd_manifest.json'), 'w'), indent=1)
    print('wrote', len(out), 'arrays')


if __name__ == '__main__':
    main()
