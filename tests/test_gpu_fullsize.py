"""GPU: the full-size BASELINE networks (ADM ImageNet-64, DDPM++ CIFAR-32, the ImageNet-64 classifier) against the CPU
oracle on the same seeded inputs -- every real layer shape (192-wide tiles, split-K levels, T=1024 attention, concat
decoders) -- plus size-independent properties at the benchmark's candidate batch (N = 64)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from helpers import oracle_net, oracle_cls_cfg                       # noqa: E402
from diffusion_tts_amd import init as dinit                          # noqa: E402
from diffusion_tts_amd.config import adm_imagenet64, ddpmpp_cifar10, ClassifierConfig   # noqa: E402

DEV = 'cuda'
torch.set_num_threads(8)


@pytest.fixture(scope='module')
def adm():
    cfg = adm_imagenet64()
    sd, _ = dinit.refill_degenerate(dinit.edm_state_dict(cfg, 0), 0)
    return cfg, sd


@pytest.mark.parametrize('dtype,tol', [(torch.float32, 3e-4), (torch.bfloat16, 8e-2)])
def test_adm64_forward_matches_oracle(adm, manifest, dtype, tol):
    from diffusion_tts_amd.networks import EDMPrecond
    cfg, sd = adm
    ck = dinit.checksum(dinit.edm_state_dict(cfg, 0))
    ref = manifest['adm_imagenet64']['checksum_raw']                 # weights == the reference constructor's
    assert ck['numel'] == ref['numel'] and abs(ck['abs_sum'] - ref['abs_sum']) < 1e-9 * ref['abs_sum']
    g = torch.Generator().manual_seed(11)
    x = torch.randn(2, 3, 64, 64, generator=g, dtype=torch.float64) * 3.0
    sigma = torch.tensor([2.5, 0.4], dtype=torch.float64)
    lab = torch.eye(1000)[torch.tensor([17, 923])]
    want = oracle_net(cfg, sd)(x, sigma, lab)
    got = EDMPrecond(cfg, sd, device=DEV, dtype=dtype)(x, sigma, lab).cpu()
    err = (got - want).abs().max().item() / max(1.0, want.abs().max().item())
    assert err < tol, err


def test_ddpmpp32_forward_matches_oracle():
    from diffusion_tts_amd.networks import EDMPrecond
    cfg = ddpmpp_cifar10()
    sd, _ = dinit.refill_degenerate(dinit.edm_state_dict(cfg, 0), 0)
    g = torch.Generator().manual_seed(12)
    x = torch.randn(3, 3, 32, 32, generator=g, dtype=torch.float64) * 2.0
    sigma = torch.tensor([1.7], dtype=torch.float64)
    lab = torch.eye(10)[torch.tensor([1, 4, 9])]
    want = oracle_net(cfg, sd)(x, sigma, lab)
    got = EDMPrecond(cfg, sd, device=DEV, dtype=torch.float32)(x, sigma, lab).cpu()
    assert (got - want).abs().max().item() < 3e-4 * max(1.0, want.abs().max().item())


def test_classifier64_matches_oracle():
    from diffusion_tts_amd.classifier import EncoderUNetModel
    from oracle.classifier import encoder_unet
    cfg = ClassifierConfig()
    sd, _ = dinit.refill_degenerate(dinit.classifier_state_dict(cfg, 1), 1)
    g = torch.Generator().manual_seed(13)
    img = torch.randint(0, 256, (2, 3, 64, 64), generator=g, dtype=torch.uint8).float() / 255.0
    want = encoder_unet(sd, oracle_cls_cfg(cfg), img, torch.zeros(2))
    got = EncoderUNetModel(cfg, sd, device=DEV, dtype=torch.float32)(img.to(DEV), torch.zeros(2, device=DEV)).cpu()
    assert (got - want).abs().max().item() < 5e-4 * max(1.0, want.abs().max().item())


def test_candidate_batch_64_properties(adm):
    """At the benchmark size (N = 64 candidates, bf16): (a) identical candidates give bit-identical outputs and equal
    rewards (the dead sigma-steps' ties); (b) a row's output depends on the batch it rides in only through the
    split-K factor of the low-resolution layers (a different but fixed summation order): equal to bf16 rounding."""
    from diffusion_tts_amd.networks import EDMPrecond
    cfg, sd = adm
    net = EDMPrecond(cfg, sd, device=DEV, dtype=torch.bfloat16)
    g = torch.Generator().manual_seed(14)
    x1 = torch.randn(1, 3, 64, 64, generator=g, dtype=torch.float64) * 5
    xs = torch.cat([x1.repeat(60, 1, 1, 1), torch.randn(4, 3, 64, 64, generator=g, dtype=torch.float64) * 5])
    lab = torch.eye(1000)[torch.tensor([7])].repeat(64, 1)
    sigma = torch.tensor([5.0], dtype=torch.float64)
    D = net(xs, sigma, lab)
    assert all(torch.equal(D[0], D[i]) for i in range(1, 60))
    assert not torch.equal(D[0], D[60])
    D8 = net(xs[56:64].contiguous(), sigma, lab[:8])
    assert (D8 - D[56:64]).abs().max().item() < 2e-2 * D.abs().max().item()


def _full_run(cfg, sd, method, params, latents, labels, seed):
    """Same search on the GPU (f32 = parity mode) and through the CPU oracle."""
    from diffusion_tts_amd import sampler as sm, scorers as S
    from diffusion_tts_amd.hashing import seed0_scale
    from diffusion_tts_amd.networks import EDMPrecond
    from oracle import sampler as osamp, scorers as oscore
    kw = dict(seed=seed, num_steps=18, S_churn=40, S_min=0.05, S_max=50, S_noise=1.003)
    onet = oracle_net(cfg, sd)
    o = osamp.search(onet, latents, labels, method=method, params=dict(scorer=oscore.BrightnessOracle(), **params),
                     scale_fn=seed0_scale, **kw)
    net = EDMPrecond(cfg, sd, device=DEV, dtype=torch.float32)
    h = sm.generate_image_grid(net, None, latents, labels, gridw=1, gridh=1, device=torch.device(DEV),
                               sampling_method={'naive': sm.SamplingMethod.NAIVE, 'rejection': sm.SamplingMethod.REJECTION_SAMPLING}[method],
                               sampling_params=dict(scorer=S.BrightnessScorer(), **params), scale_fn=seed0_scale,
                               compute_dtype=torch.float32, verbose=False, **kw)
    return o, h, onet.evals


def test_baseline_config1_ddpmpp_naive_full_trajectory():
    """BASELINE.json configs[0] at full size: CIFAR-10 DDPM++ (55.7M), NAIVE, 18 steps, S_churn=40 -- 35 denoiser evaluations.
    north_star tolerance: final image within 1e-3 abs of the CPU path (here: the oracle, pinned to the reference)."""
    cfg = ddpmpp_cifar10()
    sd, _ = dinit.refill_degenerate(dinit.edm_state_dict(cfg, 0), 0)
    g = torch.Generator().manual_seed(0)
    latents = torch.randn(1, 3, 32, 32, generator=g)
    labels = torch.eye(10)[torch.tensor([3])]
    o, h, evals = _full_run(cfg, sd, 'naive', {}, latents, labels, seed=0)
    assert evals == 35 and h['net_rows'] == 35
    assert (o['x'] - h['x'].cpu()).abs().max().item() < 1e-3
    assert (o['image'].int() - h['image'].int()).abs().max().item() <= 1


def test_baseline_config2_ddpmpp_rejection_full_trajectories():
    """BASELINE.json configs[1] (REJECTION, brightness scorer) at full network size with N=4 of the 16 trajectories (the oracle
    runs on the host): 4 x 35 rows, one scorer call of 4 and the final one; same survivor, same image."""
    cfg = ddpmpp_cifar10()
    sd, _ = dinit.refill_degenerate(dinit.edm_state_dict(cfg, 0), 0)
    g = torch.Generator().manual_seed(1)
    latents = torch.randn(1, 3, 32, 32, generator=g)
    labels = torch.eye(10)[torch.tensor([7])]
    o, h, evals = _full_run(cfg, sd, 'rejection', dict(N=4), latents, labels, seed=0)
    assert evals == 4 * 35 and h['net_rows'] == 4 * 35
    gap = torch.sort(o['rewards'][0].flatten(), descending=True).values
    if float(gap[0] - gap[1]) > 4e-5:                                    # decision margin above fp32 noise
        assert torch.equal(o['selected'][0], h['selected'][0])
        assert (o['x'] - h['x'].cpu()).abs().max().item() < 1e-3
    assert (torch.cat([r.flatten() for r in o['rewards']]) - torch.cat([r.flatten() for r in h['rewards']])).abs().max().item() < 5e-5


def test_baseline_config3_adm64_eps_greedy_imagenet_scorer_reduced(adm):
    """BASELINE.json configs[2] (the headline workload: ADM ImageNet-64 + eps-greedy + the 64x64 classifier as scorer) at full
    network size, cut to 3 sigma-steps, N=4, K=1 so that the host oracle finishes in seconds: rewards to 5e-5, and wherever
    the top-2 reward gap is above fp32 noise the same survivors and the same final state (1e-3)."""
    from diffusion_tts_amd import sampler as sm, scorers as S
    from diffusion_tts_amd.hashing import seed0_scale
    from diffusion_tts_amd.networks import EDMPrecond
    from oracle import sampler as osamp, scorers as oscore
    cfg, sd = adm
    ccfg = ClassifierConfig()
    csd, _ = dinit.refill_degenerate(dinit.classifier_state_dict(ccfg, 1), 1)
    g = torch.Generator().manual_seed(5)
    latents = torch.randn(1, 3, 64, 64, generator=g)
    labels = torch.eye(1000)[torch.tensor([207])]
    params = dict(N=4, K=1, lambda_param=0.15, eps=0.4)
    kw = dict(seed=0, num_steps=3, S_churn=40, S_min=0.05, S_max=50, S_noise=1.003)
    onet = oracle_net(cfg, sd)
    o = osamp.search(onet, latents, labels, method='eps_greedy', params=dict(scorer=oscore.ImageNetOracle(oracle_cls_cfg(ccfg), csd), **params),
                     scale_fn=seed0_scale, **kw)
    net = EDMPrecond(cfg, sd, device=DEV, dtype=torch.float32)
    scorer = S.ImageNetScorer(weights=csd, cfg=ccfg, device=DEV, compute_dtype=torch.float32)
    h = sm.generate_image_grid(net, None, latents, labels, gridw=1, gridh=1, device=torch.device(DEV),
                               sampling_method=sm.SamplingMethod.EPS_GREEDY, sampling_params=dict(scorer=scorer, **params),
                               scale_fn=seed0_scale, compute_dtype=torch.float32, verbose=False, **kw)
    assert h['net_rows'] == onet.evals
    same = True
    for ro, rh, so, sh in zip(o['rewards'], h['rewards'], o['selected'], h['selected']):
        assert (ro - rh).abs().max().item() < 5e-5
        top = torch.sort(ro.flatten(), descending=True).values
        if same and float(top[0] - top[1]) > 4e-5:
            assert torch.equal(so, sh)
        elif not torch.equal(so, sh):
            same = False                                   # a sub-noise decision went the other way: later states differ legitimately
    if same:
        assert (o['x'] - h['x'].cpu()).abs().max().item() < 1e-3
