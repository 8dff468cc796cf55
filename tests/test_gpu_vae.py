"""GPU: the SD VAE decoder on the HIP kernels (diffusion_tts_amd/vae.py) against the reference's AutoencoderKL.decode (goldens) and
the oracle, at the narrow width, at SD-1.5's width, and as the `vae` of the SD search loop at [N,4,64,64] -> [N,3,512,512]."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from conftest import ROOT                                        # noqa: E402
from diffusion_tts_amd import init as dinit                      # noqa: E402
from oracle import vae as ovae                                   # noqa: E402

DEV = 'cuda'
CASES = {'narrow': ((64, 64, 128, 128), 3), 'sd15_width': ((128, 256, 512, 512), 4)}


@pytest.fixture(scope='module')
def vg():
    return np.load(os.path.join(ROOT, 'tests', 'golden', 'vae_golden.npz'))


@pytest.mark.parametrize('name', ['narrow', 'sd15_width'])
@pytest.mark.parametrize('dtype,tol', [(torch.float16, 1.5e-2), (torch.bfloat16, 8e-2)])
def test_hip_vae_decoder_matches_reference(vg, name, dtype, tol):
    """AutoencoderKL.decode of the reference (fp32, CPU) vs the HIP decoder in the 16-bit activation types: relative to the image's
    own scale (|image| <= ~3).  Exercises the zero-padded 4-channel conv_in, the 1x1 shortcuts, the fused nearest-2x convs, the
    head-dim-512 attention (sd15_width) / head-dim-128 (narrow) and conv_out."""
    from diffusion_tts_amd.vae import VAEDecoder
    boc, seed = CASES[name]
    sd = dinit.vae_decoder_state_dict(boc, 2, 4, seed=seed)
    dec = VAEDecoder(sd, block_out_channels=boc, device=DEV, dtype=dtype)
    z = torch.from_numpy(vg[f'{name}_z'])
    got = dec.decode(z.to(DEV), return_dict=False)[0]
    want = torch.from_numpy(vg[f'{name}_image'])
    assert got.dtype == dtype and tuple(got.shape) == tuple(want.shape)
    err = (got.float().cpu() - want).abs().max().item() / want.abs().max().item()
    print(f'VAE decoder {name} {str(dtype).split(".")[-1]}: rel. max err {err:.2e}')
    assert err < tol, err
    assert dec.decodes == z.shape[0]


def test_hip_vae_decoder_batch_rows_are_independent_and_identical(vg):
    """N candidates decoded as one batch = N separate decodes; identical latents give bit-identical images (ties stay ties)."""
    from diffusion_tts_amd.vae import VAEDecoder
    sd = dinit.vae_decoder_state_dict((128, 256, 512, 512), 2, 4, seed=4)
    dec = VAEDecoder(sd, device=DEV, dtype=torch.float16)
    g = torch.Generator().manual_seed(1)
    z1 = torch.randn(1, 4, 16, 16, generator=g)
    z = torch.cat([z1, z1, torch.randn(2, 4, 16, 16, generator=g)]).to(DEV)
    out = dec.decode(z)[0]
    assert torch.equal(out[0], out[1]) and not torch.equal(out[0], out[2])
    single = dec.decode(z[2:3].contiguous())[0]
    assert (single[0].float() - out[2].float()).abs().max().item() < 2e-2


def test_sd_search_loop_with_the_hip_vae_at_sd_shapes():
    """The SD loop's decode step through the HIP decoder at config-4 sizes: [N,4,64,64] fp16 latents -> [N,3,512,512] images, N-row
    batches, same scores as decoding through the oracle's torch decoder on the GPU-produced latents (brightness, 1e-3)."""
    from diffusion_tts_amd.sd_pipeline import SDSearchPipeline
    from diffusion_tts_amd.scorers import BrightnessScorer
    from diffusion_tts_amd.vae import VAEDecoder
    from sd_standins import shape_unet, TinyTextEncoder, TinyTokenizer
    sd = dinit.vae_decoder_state_dict(seed=5)
    dec = VAEDecoder(sd, device=DEV, dtype=torch.float16)
    unet, te = shape_unet().half().to(DEV), TinyTextEncoder().half().to(DEV)
    pipe = SDSearchPipeline(unet, dec, device=DEV, text_encoder=te, tokenizer=TinyTokenizer())
    torch.manual_seed(0)
    lat = torch.randn(1, 4, 64, 64).half()
    out, score = pipe(prompt='a photo of a cat', latents=lat, num_inference_steps=2, score_function=BrightnessScorer(),
                      method='eps_greedy', params={'N': 3, 'K': 1, 'eps': 0.4, 'lambda': 2.0, 'B': 2, 'S': 4}, output_type='pt')
    assert out.images.shape == (1, 3, 512, 512) and out.images.dtype == torch.float16
    assert dec.decodes == 2 * 3 + 1 and len(out.scores) == 6                  # N candidates per step, decoded as one batch + final
    # the final image against the CPU oracle decoder on the final latents (fp32): fp16 activations through 30 convs
    with torch.no_grad():
        want = ovae.decode({k: v.float() for k, v in sd.items()}, out.latents.float().cpu() / 0.18215)
    got = out.images.float().cpu()
    want = (want / 2 + 0.5).clamp(0, 1)
    assert (got - want).abs().max().item() < 3e-2


def test_vae_decoder_from_a_diffusers_directory(tmp_path, vg):
    """Weight ingestion (SD side): a diffusers `vae/` directory -- config.json + diffusion_pytorch_model.safetensors,
    encoder tensors included -- loads into the same decoder as the state dict does (bit-identical images), and a directory without
    safetensors is refused rather than un-pickled."""
    import json
    from safetensors.torch import save_file
    from diffusion_tts_amd.vae import VAEDecoder
    boc = (64, 64, 128, 128)
    sd = dinit.vae_decoder_state_dict(boc, 2, 4, seed=3)
    d = tmp_path / 'vae'
    d.mkdir()
    full = {k: v.to(torch.float16).contiguous() for k, v in sd.items()}
    full['encoder.conv_in.weight'] = torch.zeros(8, 3, 3, 3, dtype=torch.float16)          # must be skipped
    save_file(full, str(d / 'diffusion_pytorch_model.safetensors'))
    (d / 'config.json').write_text(json.dumps({'_class_name': 'AutoencoderKL', 'act_fn': 'silu', 'block_out_channels': list(boc),
                                               'layers_per_block': 2, 'latent_channels': 4, 'norm_num_groups': 32, 'scaling_factor': 0.18215,
                                               'up_block_types': ['UpDecoderBlock2D'] * 4}))
    a = VAEDecoder.from_pretrained(str(d), device=DEV, dtype=torch.float16)
    # the CLI (`main.py --backend sd`, reference main.py:111-147) takes this decoder, not diffusers' AutoencoderKL, when the directory exists
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import main as cli
    os.environ['DTS_SD_VAE_DIR'] = str(d)
    try:
        c = cli.load_sd_vae('runwayml/stable-diffusion-v1-5', torch.device(DEV))
    finally:
        del os.environ['DTS_SD_VAE_DIR']
    assert isinstance(c, VAEDecoder)
    b = VAEDecoder({k: v.float() for k, v in full.items() if not k.startswith('encoder.')}, block_out_channels=boc, device=DEV, dtype=torch.float16)
    z = torch.from_numpy(vg['narrow_z']).to(DEV)
    assert torch.equal(a.decode(z, return_dict=False)[0], b.decode(z, return_dict=False)[0])
    assert torch.equal(c.decode(z, return_dict=False)[0], b.decode(z, return_dict=False)[0])
    assert a.config.block_out_channels == list(boc) and a.config.scaling_factor == 0.18215
    empty = tmp_path / 'empty'
    empty.mkdir()
    (empty / 'diffusion_pytorch_model.bin').write_bytes(b'x')
    with pytest.raises(FileNotFoundError):
        VAEDecoder.from_pretrained(str(empty))
    (d / 'config.json').write_text(json.dumps({'act_fn': 'relu'}))
    with pytest.raises(ValueError):
        VAEDecoder.from_pretrained(str(d))


def test_sd_beam_search_at_config4_size_with_the_hip_vae_and_clip():
    """BASELINE config 4 at its own size -- SD beam search B=4, N=16, CLIP scorer, [N,4,64,64] fp16 latents, [N,3,512,512] decodes through the
    HIP VAE decoder (SD-1.5 width) -- over 2 DDIM steps with the stand-in U-Net and a random-init CLIP: row / decode / scorer-call counts of
    the reference loop (pipeline_stable_diffusion.py:1074-1170: per step B beams x N candidates, 2N U-Net rows per beam), finite scores, the
    kept beams are the top-B of each step, and a second run reproduces the first bit for bit."""
    from diffusion_tts_amd.sd_pipeline import SDSearchPipeline
    from diffusion_tts_amd.scorers import CLIPScorer, ByteTokenizer
    from diffusion_tts_amd.vae import VAEDecoder
    from sd_standins import shape_unet, TinyTextEncoder, TinyTokenizer, tiny_clip
    B, N, steps = 4, 16, 2
    dec = VAEDecoder(dinit.vae_decoder_state_dict(seed=5), device=DEV, dtype=torch.float16)
    unet, te = shape_unet().half().to(DEV), TinyTextEncoder().half().to(DEV)
    pipe = SDSearchPipeline(unet, dec, device=DEV, text_encoder=te, tokenizer=TinyTokenizer())
    scorer = CLIPScorer(model=tiny_clip(0), tokenizer=ByteTokenizer(1000, 998, 999), device=DEV)
    runs = []
    for rep in range(2):
        torch.manual_seed(7)
        lat = torch.randn(1, 4, 64, 64).half()
        d0 = dec.decodes
        out, score = pipe(prompt='a photo of a cat', latents=lat, num_inference_steps=steps, score_function=scorer, method='beam',
                          params={'N': N, 'B': B, 'K': 20, 'lambda': 0.15, 'eps': 0.4, 'S': 8}, output_type='pt')
        runs.append((out, float(score), dec.decodes - d0))
    out, score, decodes = runs[0]
    assert out.images.shape == (1, 3, 512, 512) and out.images.dtype == torch.float16
    assert out.unet_rows == steps * B * (2 + 2 * N)                              # per beam and step: eps of the beam (cond + uncond) + 2N candidate rows
    assert len(out.scores) == steps * B * N + B                                  # every candidate scored, then the B finalists
    assert decodes == steps * B * N + B + 1                                      # ... each decoded once, + the returned image
    sc = np.array(out.scores, dtype=np.float64)
    assert np.isfinite(sc).all() and np.isfinite(out.images.float().cpu().numpy()).all()
    last = sc[(steps - 1) * B * N: steps * B * N]
    assert abs(score - np.sort(last)[::-1][:B].max()) < 5e-3                     # the winner is one of the last step's top-B (fp16 re-score)
    assert runs[1][1] == score and torch.equal(runs[1][0].images, out.images) and runs[1][0].scores == out.scores
