#!/usr/bin/env python3
"""Full-size golden vectors: outputs of THE REFERENCE ITSELF (imported on CPU in the build container) at the shapes of BASELINE
configs 1-3 -- the 192-wide ADM ImageNet-64 denoiser, the 128-wide DDPM++ CIFAR-32 denoiser, the w=128 d=4 ImageNet-64 classifier --
and ONE N = 64 eps-greedy search through the reference's own generate_image_grid code path with those full-size networks.

Run:  PYTHONHASHSEED=0 python tests/golden/make_golden_fullsize.py      (needs /root/reference; ~4 min on 8 cores)

Writes tests/golden/fullsize_golden.npz + fullsize_manifest.json: inputs, outputs, scalars and weight checksums only.  No weights
and nothing of the reference's text is stored: the weights are re-created by diffusion_tts_amd.init (proved equal to the reference
constructors parameter for parameter by this script and by make_golden.py) under the documented weight rule.

What is captured
  fwd_adm64_*      EDMPrecond(DhariwalUNet mc=192, [1,2,3,4], 3 blocks, attention 32/16/8) : 2 rows, per-row sigma  (networks.py:372-461,632-671)
  fwd_ddpmpp32_*   EDMPrecond(SongUNet DDPM++ mc=128, [2,2,2], 4 blocks, augment_dim 9)     : 2 rows, per-row sigma  (networks.py:229-363)
  cls64_*          EncoderUNetModel(w=128, d=4, attention pool) logits + ImageNetScorer rewards: 2 images           (unet.py:701-912, scorers.py:143-174)
  eg64_*           generate_image_grid(EPS_GREEDY, N=64, K=1, num_steps=2, sigma_max=3): the rewards of both decisions (64 each), the
                   selected indices, the final fp64 state / uint8 image, the first candidate's denoiser input and output     (main.py:714-886)
"""
import json
import os
import sys
import time

assert os.environ.get('PYTHONHASHSEED') == '0', 'run with PYTHONHASHSEED=0 (edm/main.py:776 hashes strings)'

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg                       # the import recipe, ref_edm / ref_classifier / run_ref_search, the weight checks  # noqa: E402

import numpy as np                             # noqa: E402
import torch                                   # noqa: E402

from diffusion_tts_amd import init as dinit    # noqa: E402
from diffusion_tts_amd.config import ClassifierConfig, adm_imagenet64, ddpmpp_cifar10  # noqa: E402

torch.set_num_threads(max(1, min(8, os.cpu_count() or 1)))
EG = dict(N=64, K=1, lambda_param=0.15, eps=0.4)
EG_STEPS, EG_SIGMA_MAX, EG_SEED = 2, 3.0, 0


def ref_full(cfg, seed):
    """the reference module with the product's weights (checked equal to the constructor's draws) + the weight rule"""
    mod = mg.ref_edm(cfg, seed)
    sd = dinit.edm_state_dict(cfg, seed)
    mg.assert_same_params(mod, sd, cfg.arch)
    sd2, refilled = dinit.refill_degenerate(sd, seed)
    mg.load_refilled(mod, sd2)
    return mod, dict(checksum=dinit.checksum(sd2), checksum_raw=dinit.checksum(sd), refilled=len(refilled))


def main():
    t00 = time.time()
    out, man = {}, {'torch': torch.__version__, 'numpy': np.__version__, 'threads': torch.get_num_threads(),
                    'net_seed': mg.NET_SEED, 'cls_seed': mg.CLS_SEED}
    g = torch.Generator().manual_seed(4242)

    # ---- denoiser forwards (x fp64 -> D fp32), two rows with different sigmas and labels
    adm, man['adm_imagenet64'] = ref_full(adm_imagenet64(), mg.NET_SEED)
    for name, net, cfg in (('adm64', adm, adm_imagenet64()),):
        r = cfg.img_resolution
        sig = torch.tensor([2.5, 0.3], dtype=torch.float64)
        x = torch.randn(2, 3, r, r, generator=g, dtype=torch.float64) * (1 + sig.reshape(-1, 1, 1, 1))
        lab_idx = torch.tensor([7, 421])
        with torch.no_grad():
            D = net(x, sig, torch.eye(cfg.label_dim)[lab_idx])
        out[f'fwd_{name}_x'], out[f'fwd_{name}_sigma'], out[f'fwd_{name}_label_idx'], out[f'fwd_{name}_D'] = x.numpy(), sig.numpy(), lab_idx.numpy(), D.numpy()
    print(f'[{time.time() - t00:6.1f}s] adm64 forward', flush=True)
    dd, man['ddpmpp_cifar10'] = ref_full(ddpmpp_cifar10(), mg.NET_SEED)
    sig = torch.tensor([5.0, 0.2], dtype=torch.float64)
    x = torch.randn(2, 3, 32, 32, generator=g, dtype=torch.float64) * (1 + sig.reshape(-1, 1, 1, 1))
    lab_idx = torch.tensor([3, 8])
    with torch.no_grad():
        D = dd(x, sig, torch.eye(10)[lab_idx])
    out['fwd_ddpmpp32_x'], out['fwd_ddpmpp32_sigma'], out['fwd_ddpmpp32_label_idx'], out['fwd_ddpmpp32_D'] = x.numpy(), sig.numpy(), lab_idx.numpy(), D.numpy()
    del dd
    print(f'[{time.time() - t00:6.1f}s] ddpmpp32 forward', flush=True)

    # ---- classifier + ImageNetScorer arithmetic (edm/scorers.py:143-174: uint8 / 255, timesteps 0, softmax, target class)
    ccfg = ClassifierConfig()
    cls = mg.ref_classifier(ccfg, mg.CLS_SEED)
    csd = dinit.classifier_state_dict(ccfg, mg.CLS_SEED)
    mg.assert_same_params(cls, csd, 'cls_full')
    csd2, crefilled = dinit.refill_degenerate(csd, mg.CLS_SEED)
    cls.load_state_dict(csd2, strict=True)
    man['cls_imagenet64'] = dict(checksum=dinit.checksum(csd2), checksum_raw=dinit.checksum(csd), refilled=len(crefilled))
    inet = mg.ref_scorers.ImageNetScorer.__new__(mg.ref_scorers.ImageNetScorer)
    torch.nn.Module.__init__(inet)
    inet.model = cls
    img = torch.randint(0, 256, (2, 3, 64, 64), generator=g, dtype=torch.uint8)
    lab2 = torch.eye(1000)[torch.tensor([7, 421])]
    with torch.no_grad():
        out['cls64_logits'] = cls(img.float() / 255.0, torch.zeros(2)).numpy()
    out['cls64_images'], out['cls64_label_idx'] = img.numpy(), np.array([7, 421])
    out['cls64_rewards'] = inet(img, lab2, torch.zeros(2)).numpy()
    print(f'[{time.time() - t00:6.1f}s] classifier forward', flush=True)

    # ---- one N = 64 eps-greedy search through the reference's generate_image_grid (CPU): 195 denoiser rows, 129 classifier images
    lat = torch.randn(1, 3, 64, 64, generator=torch.Generator().manual_seed(3))
    lab = torch.eye(1000)[torch.tensor([5])]
    saved = mg.ref_main.generate_image_grid

    def gig(*a, **kw):                       # run_ref_search leaves sigma_max at the reference default: pass ours through
        kw['sigma_max'] = EG_SIGMA_MAX
        return saved(*a, **kw)
    mg.ref_main.generate_image_grid = gig
    try:
        # (edm/main.py's loop does not disable autograd; with 64 candidates of the full-size net on a CPU the recorded graph of ONE forward
        # exceeds this container's 64 GB.  no_grad changes no value.)
        with torch.no_grad():
            lg, sl, png, err = mg.run_ref_search(adm, inet, lat, lab, 'EPS_GREEDY', EG, EG_STEPS, seed=EG_SEED)
    finally:
        mg.ref_main.generate_image_grid = saved
    assert err is None, err
    out['eg64_latents'], out['eg64_label_idx'] = lat.numpy(), np.array([5])
    rewards = [sc.numpy() for _, sc in sl.calls]
    assert [r.shape[0] for r in rewards] == [64, 64, 1], [r.shape for r in rewards]
    out['eg64_rewards0'], out['eg64_rewards1'], out['eg64_final_score'] = rewards
    out['eg64_selected'] = np.array([int(rewards[0].argmax()), int(rewards[1].argmax())])
    out['eg64_image'] = png
    # the denoiser calls: [0] x_hat of the 64 candidates of decision 0, [1] their Euler point, [2],[3] the pivot step of timestep 0, ...
    rows = [c[0].shape[0] for c in lg.calls]
    assert rows == [64, 64, 1, 1, 64, 1], rows
    out['eg64_sigmas'] = np.array([float(c[1][0]) for c in lg.calls])
    out['eg64_c0_x_row0'], out['eg64_c0_D_row0'] = lg.calls[0][0][:1].numpy(), lg.calls[0][2][:1].numpy()
    out['eg64_pivot_step0_x'], out['eg64_pivot_step0_D'] = lg.calls[2][0].numpy(), lg.calls[2][2].numpy()
    out['eg64_x_after_step0'] = lg.calls[4][0][:1].numpy()      # x_hat of timestep 1's candidate 0 == the state after timestep 0 (sigma 0.002 < S_min: no churn noise)
    out['eg64_last_x'], out['eg64_last_D'] = lg.calls[5][0].numpy(), lg.calls[5][2].numpy()
    srt = [np.sort(r)[::-1] for r in rewards[:2]]
    man['eg64'] = dict(params=EG, num_steps=EG_STEPS, sigma_max=EG_SIGMA_MAX, seed=EG_SEED, net_rows=int(sum(rows)), scorer_calls=len(sl.calls),
                       top2_gaps=[float(s[0] - s[1]) for s in srt], selected=[int(v) for v in out['eg64_selected']],
                       S=dict(S_churn=40, S_min=0.05, S_max=50, S_noise=1.003))
    print(f'[{time.time() - t00:6.1f}s] eps-greedy N=64: selected {man["eg64"]["selected"]}, top-2 gaps {man["eg64"]["top2_gaps"]}', flush=True)

    # ---- a longer free-running search against the reference itself: N = 64, K = 2, three sigma steps from sigma_max = 3 (sigma = 3, 0.19,
    # 0.002): 645 denoiser rows, 6 decisions of which the first four carry state from one to the next (the last two, at sigma 0.002 < S_min,
    # are exact 64-way ties).  Seeds are tried until every non-tie decision has a top-2 gap >= 5e-8, i.e. well above fp32 reward noise (~1e-8).
    if os.environ.get('DTS_GOLDEN_LONG', '1') != '0':
        EG2 = dict(N=64, K=2, lambda_param=0.15, eps=0.4)

        def gig3(*a, **kw):
            kw['sigma_max'] = EG_SIGMA_MAX
            return saved(*a, **kw)
        for seed2 in range(1, 5):
            mg.ref_main.generate_image_grid = gig3
            try:
                with torch.no_grad():
                    lg2, sl2, png2, err2 = mg.run_ref_search(adm, inet, lat, lab, 'EPS_GREEDY', EG2, 3, seed=seed2)
            finally:
                mg.ref_main.generate_image_grid = saved
            assert err2 is None, err2
            rew2 = [sc.numpy() for _, sc in sl2.calls]
            assert [r.shape[0] for r in rew2] == [64] * 6 + [1], [r.shape for r in rew2]
            gaps2 = [float(np.sort(r)[::-1][0] - np.sort(r)[::-1][1]) for r in rew2[:6]]
            print(f'[{time.time() - t00:6.1f}s] long search seed {seed2}: top-2 gaps {gaps2}', flush=True)
            if all(g_ == 0.0 or g_ >= 5e-8 for g_ in gaps2):
                break
        else:
            raise RuntimeError('no seed with decidable margins')
        for j in range(6):
            out[f'eg64long_rewards{j}'] = rew2[j]
        out['eg64long_final_score'] = rew2[6]
        out['eg64long_selected'] = np.array([int(r.argmax()) for r in rew2[:6]])
        out['eg64long_image'] = png2
        out['eg64long_last_D'] = lg2.calls[-1][2].numpy()
        man['eg64long'] = dict(params=EG2, num_steps=3, sigma_max=EG_SIGMA_MAX, seed=seed2, net_rows=int(sum(c[0].shape[0] for c in lg2.calls)),
                               scorer_calls=len(sl2.calls), top2_gaps=gaps2, selected=[int(v) for v in out['eg64long_selected']],
                               sigmas=sorted({round(float(c[1][0]), 6) for c in lg2.calls}, reverse=True), S=dict(S_churn=40, S_min=0.05, S_max=50, S_noise=1.003))

    np.savez_compressed(os.path.join(HERE, 'fullsize_golden.npz'), **out)
    with open(os.path.join(HERE, 'fullsize_manifest.json'), 'w') as f:
        json.dump(man, f, indent=1)
    print('wrote', len(out), 'arrays;', os.path.getsize(os.path.join(HERE, 'fullsize_golden.npz')), 'bytes')


if __name__ == '__main__':
    main()
