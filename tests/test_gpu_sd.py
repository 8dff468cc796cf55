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


@pytest.mark.parametrize('dtype', [torch.float16, torch.float32])
def test_candidate_noise_sd_rounds_like_the_reference_expression(dtype):
    """dts_candidate_noise_sd against the reference's own expression evaluated by torch on the CPU in the latents' dtype
    (pipeline_stable_diffusion.py:1377-1379): `to_add / torch.norm(to_add)`, then `pivot + to_add * rand * lambda * np.sqrt(numel)` --
    Python evaluates the products left to right, three tensor-by-scalar multiplies that each round in the tensor's type.  float16: bit for
    bit (the norm is rounded to f16, which absorbs the summation order of its f32 accumulation); float32: within one ulp of the norm."""
    from diffusion_tts_amd import ops
    g = torch.Generator().manual_seed(3)
    n, shape = 6, (1, 4, 64, 64)
    pivot = torch.randn(shape, generator=g).to(dtype)
    u = torch.randn((n,) + shape, generator=g).to(dtype)
    mode = torch.tensor([1, 0, 1, 1, 0, 1], dtype=torch.int32)
    rands = torch.rand(n, generator=g).tolist()
    lam, root = 0.15, float(np.sqrt(shape[-1] * shape[-2] * shape[-3]))
    want = []
    for c in range(n):
        if mode[c] == 0:
            want.append(u[c])
        else:
            to_add = u[c] / torch.norm(u[c])
            want.append(pivot + to_add * rands[c] * lam * root)
    want = torch.stack(want)
    scale = torch.tensor([[r, lam, root] if m_ else [0.0, 0.0, 0.0] for r, m_ in zip(rands, mode.tolist())], dtype=torch.float32)
    got = ops.candidate_noise_sd(pivot.to(DEV), u.to(DEV), mode.to(DEV), scale.to(DEV)).cpu()
    if dtype == torch.float16:
        if not torch.equal(got, want):              # say where and by how much before failing
            bad = (got != want)
            per = bad.reshape(n, -1).sum(dim=1).tolist()
            idx = bad.reshape(-1).nonzero().reshape(-1)[:6].tolist()
            uf = u.float().reshape(-1)
            detail = [(i_, float(got.reshape(-1)[i_]), float(want.reshape(-1)[i_]), float(uf[i_]), float(pivot.reshape(-1)[i_ % pivot.numel()])) for i_ in idx]
            raise AssertionError(f'{int(bad.sum())} of {bad.numel()} values differ, per candidate {per}; (index, got, want, u, pivot): {detail}; '
                                 f'norms {[float(torch.norm(u[c_])) for c_ in range(n)]}')
    else:
        assert float((got - want).abs().max()) <= 2 ** -22 * float(want.abs().max())
        assert torch.equal(got[1], want[1]) and torch.equal(got[4], want[4])
