"""GPU parity of the SD backend's candidate-batched search loop (diffusion_tts_amd/sd_pipeline.py) against the traces
of the reference's modified StableDiffusionPipeline (tests/golden/make_golden_sd.py), f32 stand-in U-Net / VAE."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from conftest import ROOT                      # noqa: E402
from sd_standins import TinyUNet, TinyVAE      # noqa: E402

DEV = 'cuda'


@pytest.fixture(scope='module')
def sdg():
    return np.load(os.path.join(ROOT, 'tests', 'golden', 'sd_golden.npz')), \
        json.load(open(os.path.join(ROOT, 'tests', 'golden', 'sd_manifest.json')))


def test_ddim_step_kernel_matches_reference_scheduler(sdg):
    from diffusion_tts_amd.sd_pipeline import DDIMScheduler
    g, m = sdg
    s = DDIMScheduler()
    assert np.array_equal(s.set_timesteps(50).numpy(), g['ddim_timesteps_50'])
    s.set_timesteps(m['steps'])
    x, e, z = (torch.from_numpy(g[k]).to(DEV) for k in ('ddim_x', 'ddim_e', 'ddim_z'))
    for t in s.timesteps.tolist():
        prev, x0 = s.step(e, t, x, variance_noise=z)
        assert np.allclose(prev.cpu().numpy(), g[f'ddim_step_{t}_prev'], atol=2e-6)
        assert np.allclose(x0.cpu().numpy(), g[f'ddim_step_{t}_x0'], atol=2e-6)


@pytest.mark.parametrize('method', ['naive', 'eps_greedy', 'zero_order', 'beam', 'mcts'])
@pytest.mark.parametrize('dead', [False, True])
def test_sd_search_matches_reference(sdg, method, dead):
    if dead and method != 'mcts':
        pytest.skip('mcts_dead_compute only affects mcts')
    from diffusion_tts_amd.sd_pipeline import SDSearchPipeline
    from diffusion_tts_amd.scorers import BrightnessScorer
    g, m = sdg
    meta = m['cases'][method]
    unet, vae = TinyUNet().to(DEV), TinyVAE().to(DEV)          # constructors touch the global RNG: build before seeding
    pipe = SDSearchPipeline(unet, vae, device=DEV, mcts_dead_compute=dead)
    torch.manual_seed(meta['seed'])
    out, score = pipe(prompt=None, prompt_embeds=torch.from_numpy(g['prompt_embeds']),
                      negative_prompt_embeds=torch.from_numpy(g['negative_prompt_embeds']), latents=torch.from_numpy(g['latents']).clone(),
                      num_inference_steps=m['steps'], score_function=BrightnessScorer(), method=method, params=meta['params'], output_type='pt')
    if method != 'mcts' or dead:
        assert out.unet_rows == meta['unet_rows']              # same candidate rows through the U-Net, in fewer, larger calls
    assert len(out.scores) == meta['scorer_calls']
    assert np.allclose(np.array(out.scores), g[f'{method}_scores'], atol=2e-5)
    assert np.allclose(out.images.float().cpu().numpy(), g[f'{method}_image'], atol=2e-5)
    assert abs(float(score.item() if torch.is_tensor(score) else score) - meta['max_score']) < 2e-5
