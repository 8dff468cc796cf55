#!/usr/bin/env python3
"""Generates the golden vectors under tests/golden/ by IMPORTING THE REFERENCE (CPU) in the build container.

Run:  PYTHONHASHSEED=0 python tests/golden/make_golden.py        (needs /root/reference; ~2-4 min)

Nothing of the reference is written out: only input/output arrays, scalars and checksums.  Network
weights are NOT stored; they are re-created by diffusion_tts_amd.init (which this script proves equal to
the reference constructors, parameter for parameter) and pinned by checksum.

Import recipe: SURVEY.md section 8(c) (torchvision stub; local-path pickle replaced by an in-memory
hand-over so no pickle with embedded source is ever produced).
"""
import json
import os
import sys
import tempfile
import types

assert os.environ.get('PYTHONHASHSEED') == '0', 'run with PYTHONHASHSEED=0 (edm/main.py:776 hashes strings)'

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

import numpy as np
import torch

torch.set_num_threads(4)

import transformers  # noqa: F401  (must precede the torchvision stub)
from transformers import CLIPModel, CLIPProcessor, ViTForImageClassification, ViTImageProcessor  # noqa: F401

for _n in ('torchvision', 'torchvision.models', 'torchvision.transforms'):
    sys.modules[_n] = types.ModuleType(_n)
sys.modules['torchvision'].models = sys.modules['torchvision.models']
sys.modules['torchvision'].transforms = sys.modules['torchvision.transforms']
sys.modules['torchvision.transforms'].Compose = lambda x: x
sys.modules['torchvision.transforms'].ToTensor = lambda: None
REF = os.environ.get('DTS_REFERENCE', '/root/reference')
sys.path.insert(0, os.path.join(REF, 'edm'))
import main as ref_main                      # noqa: E402  edm/main.py
from training import networks as ref_networks  # noqa: E402
import unet as ref_unet                      # noqa: E402
import scorers as ref_scorers                # noqa: E402
import PIL.Image                             # noqa: E402

from diffusion_tts_amd import init as dinit  # noqa: E402
from diffusion_tts_amd.config import EDMConfig, ClassifierConfig, adm_imagenet64, ddpmpp_cifar10  # noqa: E402

# ------------------------------------------------------------------------------------------------
TINY = {
    'adm_tiny': EDMConfig('adm', 16, 3, 10, 64, [1, 2], 4, 1, [8], 0),
    'ddpmpp_tiny': EDMConfig('ddpmpp', 16, 3, 10, 64, [1, 2], 4, 1, [8], 9),
}
TINY_CLS = ClassifierConfig(image_size=16, in_channels=3, model_channels=64, out_channels=10, num_res_blocks=1,
                            attention_ds=(2,), channel_mult=(1, 2), num_head_channels=64)
NET_SEED, CLS_SEED = 0, 1


def ref_edm(cfg: EDMConfig, seed):
    torch.manual_seed(seed)
    if cfg.arch == 'adm':
        kw = dict(model_type='DhariwalUNet', model_channels=cfg.model_channels, channel_mult=cfg.channel_mult,
                  num_blocks=cfg.num_blocks, attn_resolutions=cfg.attn_resolutions, augment_dim=cfg.augment_dim)
    else:
        kw = dict(model_type='SongUNet', model_channels=cfg.model_channels, channel_mult=cfg.channel_mult,
                  num_blocks=cfg.num_blocks, attn_resolutions=cfg.attn_resolutions, augment_dim=cfg.augment_dim,
                  embedding_type='positional', encoder_type='standard', decoder_type='standard',
                  channel_mult_noise=1, resample_filter=[1, 1], dropout=0.13)
    return ref_networks.EDMPrecond(img_resolution=cfg.img_resolution, img_channels=cfg.img_channels,
                                   label_dim=cfg.label_dim, **kw).eval()


def ref_classifier(cfg: ClassifierConfig, seed):
    torch.manual_seed(seed)
    return ref_unet.EncoderUNetModel(
        image_size=cfg.image_size, in_channels=cfg.in_channels, model_channels=cfg.model_channels,
        out_channels=cfg.out_channels, num_res_blocks=cfg.num_res_blocks, attention_resolutions=cfg.attention_ds,
        channel_mult=cfg.channel_mult, use_fp16=False, num_head_channels=cfg.num_head_channels,
        use_scale_shift_norm=True, resblock_updown=True, pool='attention').eval()


def assert_same_params(mod, sd, what):
    ref = {k: v for k, v in mod.named_parameters()}
    assert list(ref.keys()) == list(sd.keys()), (what, [k for k in ref if k not in sd][:5], [k for k in sd if k not in ref][:5])
    for k in ref:
        assert ref[k].shape == sd[k].shape and torch.equal(ref[k].detach(), sd[k]), (what, k)


def load_refilled(mod, sd):
    missing, unexpected = mod.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all('resample_filter' in m for m in missing), missing


class NetLogger(torch.nn.Module):
    """Stands in for the unpickled network (edm/main.py:69-70) and records every denoiser call."""

    def __init__(self, net):
        super().__init__()
        self.net = net
        self.img_resolution, self.img_channels, self.label_dim = net.img_resolution, net.img_channels, net.label_dim
        self.calls = []

    def round_sigma(self, s):
        return self.net.round_sigma(s)

    def forward(self, x, sigma, class_labels=None):
        out = self.net(x, sigma, class_labels)
        self.calls.append((x.detach().clone(), torch.as_tensor(sigma).detach().clone().reshape(-1), out.detach().clone()))
        return out


class ScoreLogger:
    def __init__(self, scorer):
        self.scorer, self.calls = scorer, []

    def __call__(self, images, labels, timesteps):
        s = self.scorer(images, labels, timesteps)
        self.calls.append((images.detach().clone(), s.detach().clone()))
        return s


def run_ref_search(net, scorer, latents, labels, method, params, num_steps, seed=0):
    logger, slog = NetLogger(net), ScoreLogger(scorer)

    class _HandOver:                      # replaces `pickle.load(f)['ema']`
        @staticmethod
        def load(f):
            return {'ema': logger}

    saved = ref_main.pickle
    ref_main.pickle = _HandOver
    np.random.seed(0)
    err = None
    with tempfile.TemporaryDirectory() as td:
        png = os.path.join(td, 'out.png')
        try:
            ref_main.generate_image_grid(
                os.path.abspath(__file__), png, latents, labels, seed=seed, gridw=latents.shape[0], gridh=1,
                device=torch.device('cpu'), num_steps=num_steps, S_churn=40, S_min=0.05, S_max=50, S_noise=1.003,
                sampling_method=getattr(ref_main.SamplingMethod, method), sampling_params=dict(scorer=slog, **params))
            img = np.array(PIL.Image.open(png))
        except AttributeError as e:
            err, img = repr(e), None
        finally:
            ref_main.pickle = saved
    return logger, slog, img, err


def main():
    out = {}
    manifest = {'torch': torch.__version__, 'numpy': np.__version__, 'python': sys.version.split()[0],
                'hash_algorithm': sys.hash_info.algorithm, 'pillow': PIL.__version__,
                'net_seed': NET_SEED, 'cls_seed': CLS_SEED, 'weight_rule': dinit.refill_degenerate.__doc__}

    # ---- hash() scale table (edm/main.py:776) under PYTHONHASHSEED=0
    tab = np.array([[[hash(f"{i}_{k}_{n}") % 1000 / 1000.0 for n in range(64)] for k in range(4)] for i in range(18)])
    out['hash_scale_table'] = tab

    # ---- weights: product initialiser == reference constructor; then the refill rule on both
    nets, sds = {}, {}
    for name, cfg in TINY.items():
        mod = ref_edm(cfg, NET_SEED)
        sd = dinit.edm_state_dict(cfg, NET_SEED)
        assert_same_params(mod, sd, name)
        sd2, refilled = dinit.refill_degenerate(sd, NET_SEED)
        load_refilled(mod, sd2)
        nets[name], sds[name] = mod, sd2
        manifest[name] = dict(cfg=cfg.__dict__, checksum=dinit.checksum(sd2), checksum_raw=dinit.checksum(sd),
                              refilled=refilled)
    cls = ref_classifier(TINY_CLS, CLS_SEED)
    csd = dinit.classifier_state_dict(TINY_CLS, CLS_SEED)
    assert_same_params(cls, csd, 'cls_tiny')
    csd2, crefilled = dinit.refill_degenerate(csd, CLS_SEED)
    cls.load_state_dict(csd2, strict=True)
    manifest['cls_tiny'] = dict(cfg={k: (list(v) if isinstance(v, tuple) else v) for k, v in TINY_CLS.__dict__.items()},
                                checksum=dinit.checksum(csd2), checksum_raw=dinit.checksum(csd), refilled=crefilled)

    # ---- full-size: checksums + FLOPs only
    from torch.utils.flop_counter import FlopCounterMode
    for name, cfg in (('adm_imagenet64', adm_imagenet64()), ('ddpmpp_cifar10', ddpmpp_cifar10())):
        mod = ref_edm(cfg, NET_SEED)
        sd = dinit.edm_state_dict(cfg, NET_SEED)
        assert_same_params(mod, sd, name)
        with FlopCounterMode(display=False) as fc, torch.no_grad():
            mod(torch.randn(1, 3, cfg.img_resolution, cfg.img_resolution), torch.tensor([1.5]),
                torch.eye(cfg.label_dim)[:1])
        manifest[name] = dict(checksum_raw=dinit.checksum(sd), flops=float(fc.get_total_flops()),
                              params=sum(v.numel() for v in sd.values()))
        del mod, sd
    fcfg = ClassifierConfig()
    mod = ref_classifier(fcfg, CLS_SEED)
    sd = dinit.classifier_state_dict(fcfg, CLS_SEED)
    assert_same_params(mod, sd, 'cls_full')
    with FlopCounterMode(display=False) as fc, torch.no_grad():
        mod(torch.rand(1, 3, 64, 64), torch.zeros(1))
    manifest['cls_imagenet64'] = dict(checksum_raw=dinit.checksum(sd), flops=float(fc.get_total_flops()),
                                      params=sum(v.numel() for v in sd.values()))
    del mod, sd

    # ---- G3: denoiser forward KATs (x fp64 -> D fp32), per-row sigma included
    g = torch.Generator().manual_seed(1234)
    x = torch.randn(3, 3, 16, 16, generator=g, dtype=torch.float64)
    labels = torch.eye(10)[torch.tensor([3, 7, 1])]
    for name, net in nets.items():
        for tag, sig in (('hi', torch.tensor(12.5, dtype=torch.float64)), ('lo', torch.tensor(0.07, dtype=torch.float64))):
            xs = x * (1 + sig)
            with torch.no_grad():
                D = net(xs, sig, labels)
            out[f'fwd_{name}_{tag}_x'] = xs.numpy()
            out[f'fwd_{name}_{tag}_sigma'] = sig.numpy()
            out[f'fwd_{name}_{tag}_D'] = D.numpy()
        sig = torch.tensor([80.0, 1.3, 0.01], dtype=torch.float64)
        xs = x * (1 + sig.reshape(-1, 1, 1, 1))
        with torch.no_grad():
            D = net(xs, sig, labels)
        out[f'fwd_{name}_rows_x'], out[f'fwd_{name}_rows_sigma'], out[f'fwd_{name}_rows_D'] = xs.numpy(), sig.numpy(), D.numpy()
    out['fwd_labels'] = labels.numpy()

    # ---- G5: scorers
    img = torch.randint(0, 256, (5, 3, 16, 16), generator=g, dtype=torch.uint8)
    out['score_images'] = img.numpy()
    out['score_brightness'] = ref_scorers.BrightnessScorer()(img, None, torch.zeros(5)).numpy()
    inet = ref_scorers.ImageNetScorer.__new__(ref_scorers.ImageNetScorer)
    torch.nn.Module.__init__(inet)
    inet.model = cls
    lab5 = torch.eye(10)[torch.tensor([0, 9, 4, 4, 2])]
    out['score_labels'] = lab5.numpy()
    with torch.no_grad():
        out['score_cls_logits'] = cls(img.float() / 255.0, torch.zeros(5)).numpy()
    out['score_imagenet'] = inet(img, lab5, torch.zeros(5)).numpy()
    jp = ref_scorers.CompressibilityScorer()
    out['score_jpeg'] = jp(img, None, None).numpy()

    # ---- G1/G4/G6: search traces through the reference's generate_image_grid (CPU)
    bright = ref_scorers.BrightnessScorer()
    lat_g = torch.Generator().manual_seed(99)
    latents1 = torch.randn(1, 3, 16, 16, generator=lat_g)
    latents2 = torch.randn(2, 3, 16, 16, generator=lat_g)
    lab1 = torch.eye(10)[torch.tensor([4])]
    lab2 = torch.eye(10)[torch.tensor([4, 8])]
    out['search_latents1'], out['search_latents2'] = latents1.numpy(), latents2.numpy()
    out['search_lab1'], out['search_lab2'] = lab1.numpy(), lab2.numpy()
    cases = [
        # (case, net, scorer, method, params, steps, batch)
        ('naive_adm', 'adm_tiny', 'brightness', 'NAIVE', {}, 6, 2),
        ('naive_ddpmpp', 'ddpmpp_tiny', 'brightness', 'NAIVE', {}, 18, 1),
        ('rejection_adm', 'adm_tiny', 'brightness', 'REJECTION_SAMPLING', dict(N=4), 6, 2),
        ('rejection_ddpmpp', 'ddpmpp_tiny', 'brightness', 'REJECTION_SAMPLING', dict(N=4), 6, 1),
        ('epsgreedy_adm_bright', 'adm_tiny', 'brightness', 'EPS_GREEDY', dict(N=4, K=2, lambda_param=0.15, eps=0.4), 6, 2),
        ('epsgreedy_adm_imagenet', 'adm_tiny', 'imagenet', 'EPS_GREEDY', dict(N=4, K=2, lambda_param=0.15, eps=0.4), 6, 1),
        ('zeroorder_adm', 'adm_tiny', 'brightness', 'ZERO_ORDER', dict(N=4, K=2, lambda_param=0.15, eps=0.0), 6, 1),
        ('mcts_adm', 'adm_tiny', 'brightness', 'MCTS', dict(N=2, S=4), 6, 2),
        ('beam_adm', 'adm_tiny', 'brightness', 'BEAM_SEARCH', dict(B=2, N=2), 6, 1),
    ]
    case_meta = {}
    for case, netname, scname, method, params, steps, batch in cases:
        lat, lab = (latents1, lab1) if batch == 1 else (latents2, lab2)
        scorer = bright if scname == 'brightness' else inet
        # A decision whose top-2 reward gap is below fp32 noise (or below one truncation flip of the uint8 image,
        # ~1e-5 for brightness) is numerically undetermined even between two runs of the reference on different
        # BLAS builds.  Pick the first search seed for which every decision is an exact tie (first-max rule) or has a
        # gap >= MARGIN, so "same selected indices" is a meaningful parity statement.
        MARGIN = 4e-5
        for seed in range(64):
            lg, sl, png, err = run_ref_search(nets[netname], scorer, lat, lab, method, params, steps, seed=seed)
            if err is not None:
                break
            gaps = []
            if method in ('EPS_GREEDY', 'ZERO_ORDER'):
                for imgs, sc in sl.calls[:-1]:
                    v = sc.reshape(params['N'], batch)
                    t = v.sort(dim=0, descending=True).values
                    gaps += (t[0] - t[1]).tolist()
            elif method == 'REJECTION_SAMPLING':
                t = sl.calls[0][1].reshape(batch, -1).sort(dim=1, descending=True).values
                gaps += (t[:, 0] - t[:, 1]).tolist()
            if all(g == 0 or g >= MARGIN for g in gaps):
                break
        else:
            raise RuntimeError(f'{case}: no seed with safe decision margins')
        meta = dict(net=netname, scorer=scname, method=method, params=params, num_steps=steps, batch=batch, seed=seed,
                    min_gap=(min([g for g in gaps if g > 0]) if err is None and any(g > 0 for g in gaps) else None),
                    error=err, net_calls=len(lg.calls), net_rows=int(sum(c[0].shape[0] for c in lg.calls)),
                    scorer_calls=len(sl.calls))
        case_meta[case] = meta
        if err is not None:
            continue
        out[f'{case}_image'] = png
        out[f'{case}_last_D'] = lg.calls[-1][2].numpy()
        for j, (imgs, sc) in enumerate(sl.calls):
            out[f'{case}_score{j}'] = sc.numpy()
        out[f'{case}_score_img_first'] = sl.calls[0][0].numpy()
        out[f'{case}_score_img_last'] = sl.calls[-1][0].numpy()
        # G4: the first two and the last denoiser calls in full (x_hat/x_next fp64, sigma, D)
        for tag, idx in (('c0', 0), ('c1', 1), ('cl', len(lg.calls) - 1)):
            out[f'{case}_{tag}_x'] = lg.calls[idx][0].numpy()
            out[f'{case}_{tag}_sigma'] = lg.calls[idx][1].numpy()
            out[f'{case}_{tag}_D'] = lg.calls[idx][2].numpy()
        out[f'{case}_sigmas'] = np.array([float(c[1][0]) for c in lg.calls])
        print(case, meta, flush=True)
    manifest['cases'] = case_meta

    np.savez_compressed(os.path.join(HERE, 'edm_golden.npz'), **out)
    with open(os.path.join(HERE, 'manifest.json'), 'w') as f:
        json.dump(manifest, f, indent=1, default=lambda o: o if not isinstance(o, (np.floating, np.integer)) else o.item())
    print('wrote', len(out), 'arrays;', os.path.getsize(os.path.join(HERE, 'edm_golden.npz')), 'bytes')


if __name__ == '__main__':
    main()
