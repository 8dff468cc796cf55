"""GPU: the SD backend at SD-1.5's real tensor shapes and dtype -- latents [*,4,64,64] fp16, prompt embeddings [1,77,768], decoded
images [*,3,512,512] -- against traces of the reference's own pipeline run in fp16 on CPU around shape-faithful stand-ins
(tests/golden/make_golden_sd_shapes.py); the CLIP scorer against the reference's CLIPScorer.__call__ with an injected random-init
CLIP; prompt encoding inside the pipeline; the MCTS back-propagation flag."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from conftest import ROOT                                                   # noqa: E402
from sd_standins import ShapeVAE, shape_unet, TinyTextEncoder, TinyTokenizer, tiny_clip   # noqa: E402

DEV = 'cuda'


@pytest.fixture(scope='module')
def sg():
    return np.load(os.path.join(ROOT, 'tests', 'golden', 'sd_shapes_golden.npz')), \
        json.load(open(os.path.join(ROOT, 'tests', 'golden', 'sd_shapes_manifest.json')))


@pytest.fixture(scope='module')
def clip_scorer():
    from diffusion_tts_amd.scorers import CLIPScorer, ByteTokenizer
    return CLIPScorer(model=tiny_clip(0), tokenizer=ByteTokenizer(1000, 998, 999), device=DEV)


def test_clip_scorer_matches_reference_call(sg, clip_scorer):
    """sd/scorers.py:166-213 with the same random-init CLIP: uint8 images in the SD loop's calling convention (a 1-element list of
    [1,3,512,512]) and a float [0,1] batch with one prompt per image."""
    g, _ = sg
    gen = torch.Generator().manual_seed(11)
    imgs = [torch.randint(0, 256, (1, 3, 512, 512), generator=gen, dtype=torch.uint8) for _ in range(3)]
    assert int(torch.cat(imgs).long().sum()) == int(g['clip_images_checksum'][0])
    got = [float(clip_scorer([im.to(DEV)], ['a photo of a cat'], None)) for im in imgs]
    assert np.allclose(got, g['clip_scores_u8'], atol=2e-5), (got, g['clip_scores_u8'])
    fimg = torch.rand(2, 3, 300, 260, generator=gen)
    got = clip_scorer(fimg.to(DEV), ['two dogs', 'a red car'], None).cpu().numpy()
    assert np.allclose(got, g['clip_scores_f32'], atol=2e-5)
    assert float(clip_scorer(fimg.to(DEV), None, None).abs().max()) == 0.0          # prompts=None -> zeros (:186-188)


def test_clip_scorer_without_local_files_says_what_to_pass():
    from diffusion_tts_amd.scorers import CLIPScorer
    with pytest.raises(RuntimeError, match='pass model='):
        CLIPScorer(model_id='openai/clip-vit-large-patch14', device=DEV)


def test_pipeline_encodes_the_prompt_and_draws_latents_like_the_reference(sg):
    """`pipe(prompt=..., num_inference_steps=..., score_function=..., method=..., params=...)` with nothing else: the pipeline runs the
    text encoder (encode_prompt, pipeline...:382-460) and prepare_latents itself.  f32, brightness."""
    from diffusion_tts_amd.sd_pipeline import SDSearchPipeline
    from diffusion_tts_amd.scorers import BrightnessScorer
    g, m = sg
    unet, vae, te = shape_unet().to(DEV), ShapeVAE().to(DEV), TinyTextEncoder().to(DEV)
    pipe = SDSearchPipeline(unet, vae, device=DEV, text_encoder=te, tokenizer=TinyTokenizer())
    pe, ne = pipe.encode_prompt('an astronaut riding a horse')
    assert np.allclose(pe.cpu().numpy(), g['enc_prompt_embeds'], atol=1e-5) and np.allclose(ne.cpu().numpy(), g['enc_negative_prompt_embeds'], atol=1e-5)
    torch.manual_seed(m['prompt_call']['seed'])
    out, score = pipe(prompt='an astronaut riding a horse', num_inference_steps=2, score_function=BrightnessScorer(), method='eps_greedy',
                      params={'N': 2, 'lambda': 0.15, 'eps': 0.4, 'K': 1, 'B': 2, 'S': 8}, output_type='pt')
    assert len(out.scores) == m['prompt_call']['scorer_calls']
    assert np.allclose(out.scores[:2], g['prompt_call_scores'][:2], atol=2e-5)          # first group: same inputs whatever is picked
    assert out.images.shape == (1, 3, 512, 512)
    with pytest.raises(ValueError, match='Cannot forward both'):
        pipe(prompt='x', prompt_embeds=pe, negative_prompt_embeds=ne, score_function=BrightnessScorer(), method='naive', params={})


@pytest.mark.parametrize('method', ['naive', 'eps_greedy', 'beam'])
def test_fp16_search_at_sd_shapes_matches_reference(sg, clip_scorer, method):
    """K13 (fused DDIM candidate step), cfg_combine and quantize_u8 at config-4 sizes in fp16: [N,4,64,64] latents, 2N-row U-Net
    calls, N-row decodes to [N,3,512,512].  Scores are compared up to and including the first survivor decision whose reward gap is
    below the fp16 noise margin recorded by the generator (later states may legitimately differ); the image too when every decision
    was safe."""
    from diffusion_tts_amd.sd_pipeline import SDSearchPipeline
    from diffusion_tts_amd.scorers import BrightnessScorer
    g, m = sg
    meta = m['cases'][method]
    unet, vae, te = shape_unet().half().to(DEV), ShapeVAE().half().to(DEV), TinyTextEncoder().half().to(DEV)
    pipe = SDSearchPipeline(unet, vae, device=DEV, text_encoder=te, tokenizer=TinyTokenizer())
    scorer = clip_scorer if meta['scorer'] == 'clip' else BrightnessScorer()
    torch.manual_seed(meta['seed'])
    out, score = pipe(prompt='a photo of a cat', latents=torch.from_numpy(g['latents']).clone(), num_inference_steps=m['steps'],
                      score_function=scorer, method=method, params=meta['params'], output_type='pt')
    assert out.unet_rows == meta['unet_rows'] and len(out.scores) == meta['scorer_calls']
    assert out.images.shape == (1, 3, 512, 512) and out.images.dtype == torch.float16
    dec, safe = meta['decisions'], meta['safe_decisions']
    ncmp = len(out.scores) if safe == len(dec) else dec[safe]['after_calls']
    tol = 3e-4 if meta['scorer'] == 'brightness' else 4e-4        # = the generator's decision margins (fp16 CPU vs fp16 GPU convolutions)
    err = np.abs(np.array(out.scores[:ncmp]) - g[f'{method}_scores'][:ncmp]).max()
    print(f'{method} fp16 @ SD shapes: {ncmp}/{len(out.scores)} scores compared ({safe}/{len(dec)} safe decisions), max err {err:.2e}')
    assert err < tol, err
    if safe == len(dec):
        img = out.images.float().cpu().numpy()[:, :, ::4, ::4]
        assert np.abs(img - g[f'{method}_image']).max() < 3e-2
        assert abs(float(score.item() if torch.is_tensor(score) else score) - meta['max_score']) < tol


def test_mcts_backprop_flag(sg):
    """Default: the reference's behaviour (no scoring inside the tree, first child).  mcts_backprop=True: every simulation is scored
    and back-propagated -- visits add up, the chosen child is the best-mean one, the result differs from the reference-faithful run."""
    from diffusion_tts_amd.sd_pipeline import SDSearchPipeline
    from diffusion_tts_amd.scorers import BrightnessScorer
    g, _ = sg
    unet, vae, te = shape_unet().to(DEV), ShapeVAE().to(DEV), TinyTextEncoder().to(DEV)
    lat = torch.from_numpy(g['latents']).float()
    params = {'N': 2, 'S': 5, 'lambda': 0.15, 'eps': 0.4, 'K': 1, 'B': 2}
    res = {}
    for flag in (False, True):
        pipe = SDSearchPipeline(unet, vae, device=DEV, text_encoder=te, tokenizer=TinyTokenizer(), mcts_backprop=flag)
        torch.manual_seed(0)
        out, score = pipe(prompt='a photo of a cat', latents=lat.clone(), num_inference_steps=3, score_function=BrightnessScorer(),
                          method='mcts', params=params, output_type='pt')
        res[flag] = (out, float(score.item() if torch.is_tensor(score) else score))
    assert len(res[False][0].scores) == 1                                  # reference: only the final image is ever scored
    assert len(res[True][0].scores) == 3 * params['S']                     # one decode + score per simulation
    assert res[True][1] == max(res[True][0].scores) and np.isfinite(res[True][0].scores).all()
    assert not torch.equal(res[True][0].images, res[False][0].images)
