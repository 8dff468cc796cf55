"""CPU: the SD-backend oracle (oracle/sd_loop.py) reproduces the reference's modified StableDiffusionPipeline /
DDIMScheduler on the golden traces (tests/golden/make_golden_sd.py): DDIM known answers, and for every search method
the full sequence of rewards, the U-Net row count and the final latents."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import ROOT
from sd_standins import TinyUNet, TinyVAE
from oracle.sd_loop import DDIMOracle, sd_search
from oracle.scorers import BrightnessOracle

torch.set_num_threads(4)


@pytest.fixture(scope='module')
def sdg():
    return np.load(os.path.join(ROOT, 'tests', 'golden', 'sd_golden.npz')), \
        json.load(open(os.path.join(ROOT, 'tests', 'golden', 'sd_manifest.json')))


class ListBrightness:
    """sd/scorers.py:30-74 BrightnessScorer on the SD calling convention (a list holding one [1,3,H,W] uint8 tensor)."""

    def __call__(self, images, prompts, timesteps):
        return BrightnessOracle()(torch.cat([im if im.dim() == 4 else im[None] for im in images]).cpu(), None, None)


def test_ddim_known_answers(sdg):
    g, m = sdg
    s = DDIMOracle()
    assert np.array_equal(s.set_timesteps(50).numpy(), g['ddim_timesteps_50'])
    assert np.allclose(s.alphas_cumprod.numpy(), g['ddim_alphas_cumprod'], rtol=1e-6)
    s.set_timesteps(m['steps'])
    x, e, z = (torch.from_numpy(g[k]) for k in ('ddim_x', 'ddim_e', 'ddim_z'))
    for t in s.timesteps.tolist():
        prev, x0 = s.step(e, t, x, variance_noise=z)
        assert np.allclose(prev.numpy(), g[f'ddim_step_{t}_prev'], atol=1e-6)
        assert np.allclose(x0.numpy(), g[f'ddim_step_{t}_x0'], atol=1e-6)


@pytest.mark.parametrize('method', ['naive', 'eps_greedy', 'zero_order', 'beam', 'mcts'])
def test_sd_search_matches_reference(sdg, method):
    g, m = sdg
    meta = m['cases'][method]
    unet, vae = TinyUNet(), TinyVAE()                 # constructors touch the global RNG: build before seeding
    torch.manual_seed(meta['seed'])
    res = sd_search(unet, vae, DDIMOracle(), torch.from_numpy(g['prompt_embeds']),
                    torch.from_numpy(g['negative_prompt_embeds']), torch.from_numpy(g['latents']).clone(),
                    num_inference_steps=m['steps'], score_function=ListBrightness(), method=method, params=meta['params'])
    assert res['unet_rows'] == meta['unet_rows']
    assert len(res['scores']) == meta['scorer_calls']
    assert np.allclose(np.array(res['scores']), g[f'{method}_scores'], atol=2e-6)
    img = (res['image'] / 2 + 0.5).clamp(0, 1)                    # VaeImageProcessor.postprocess(output_type='pt')
    assert np.allclose(img.numpy(), g[f'{method}_image'], atol=1e-5)
    ms = res['max_score']
    assert abs(float(ms.item() if torch.is_tensor(ms) else ms) - meta['max_score']) < 2e-6
