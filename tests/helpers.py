"""Shared test plumbing: tiny-net construction from the golden manifest (weights re-created by the
product initialiser and pinned by checksum), oracle wrappers."""
import numpy as np
import torch

from diffusion_tts_amd import init as dinit
from diffusion_tts_amd.config import EDMConfig, ClassifierConfig
from oracle.edm_nets import NetCfg, EDMPrecondOracle
from oracle.classifier import ClsCfg


def tiny_edm(manifest, name):
    c = manifest[name]['cfg']
    cfg = EDMConfig(**c)
    sd, _ = dinit.refill_degenerate(dinit.edm_state_dict(cfg, manifest['net_seed']), manifest['net_seed'])
    ck = dinit.checksum(sd)
    ref = manifest[name]['checksum']
    assert ck['numel'] == ref['numel'] and abs(ck['sum'] - ref['sum']) < 1e-9 * max(1, abs(ref['sum'])) \
        and abs(ck['abs_sum'] - ref['abs_sum']) < 1e-9 * ref['abs_sum'], (ck, ref)
    return cfg, sd


def tiny_cls(manifest):
    c = dict(manifest['cls_tiny']['cfg'])
    c['attention_ds'] = tuple(c['attention_ds'])
    c['channel_mult'] = tuple(c['channel_mult'])
    cfg = ClassifierConfig(**c)
    sd, _ = dinit.refill_degenerate(dinit.classifier_state_dict(cfg, manifest['cls_seed']), manifest['cls_seed'])
    ck = dinit.checksum(sd)
    ref = manifest['cls_tiny']['checksum']
    assert ck['numel'] == ref['numel'] and abs(ck['abs_sum'] - ref['abs_sum']) < 1e-9 * ref['abs_sum'], (ck, ref)
    return cfg, sd


def oracle_net(cfg: EDMConfig, sd):
    ncfg = NetCfg(arch=cfg.arch, img_resolution=cfg.img_resolution, img_channels=cfg.img_channels,
                  label_dim=cfg.label_dim, model_channels=cfg.model_channels, channel_mult=list(cfg.channel_mult),
                  channel_mult_emb=cfg.channel_mult_emb, num_blocks=cfg.num_blocks,
                  attn_resolutions=list(cfg.attn_resolutions), augment_dim=cfg.augment_dim)
    return EDMPrecondOracle(ncfg, sd)


def oracle_cls_cfg(cfg: ClassifierConfig):
    return ClsCfg(image_size=cfg.image_size, in_channels=cfg.in_channels, model_channels=cfg.model_channels,
                  out_channels=cfg.out_channels, num_res_blocks=cfg.num_res_blocks, attention_ds=tuple(cfg.attention_ds),
                  channel_mult=tuple(cfg.channel_mult), num_head_channels=cfg.num_head_channels)


def T(a):
    return torch.from_numpy(np.asarray(a))


MARGIN = 4e-5          # top-2 reward gap above which a selection is reproducible in fp32


def check_decisions(rewards_o, selected_o, selected_h, what, margin=MARGIN):
    """Compares the survivor choices of the oracle and of the GPU run, decision by decision, and SAYS which kind each was:
      safe      top-2 gap > margin                 -> indices must be equal
      tie       top-2 gap == 0 (identical rows)    -> first-max rule, indices must be equal
      sub-noise 0 < gap <= margin                  -> not reproducible between two fp32 summation orders; once one of these goes
                                                     the other way the later states legitimately differ and checking stops
    Fails if NO decision was checkable (the test would otherwise assert nothing).  Returns (all_equal_so_far, report)."""
    kinds = {'safe': 0, 'tie': 0, 'sub-noise': 0, 'diverged-after': 0}
    same = True
    for ro, so, sh in zip(rewards_o, selected_o, selected_h):
        r = ro.reshape(ro.shape[0], -1) if ro.dim() > 1 else ro.reshape(-1, 1)
        if not same:
            kinds['diverged-after'] += 1
            continue
        top = torch.sort(r, dim=0, descending=True).values
        gap = float((top[0] - top[1]).min()) if r.shape[0] > 1 else float('inf')
        kind = 'tie' if gap == 0 else ('safe' if gap > margin else 'sub-noise')
        kinds[kind] += 1
        if kind in ('safe', 'tie'):
            assert torch.equal(so.reshape(-1), sh.reshape(-1)), (what, kind, gap, so, sh)
        elif not torch.equal(so.reshape(-1), sh.reshape(-1)):
            same = False
    report = f'{what}: decisions {kinds}'
    print(report)
    assert kinds['safe'] + kinds['tie'] > 0, f'{report} -- no decision had a checkable margin: pick another seed'
    return same, report


def synthetic_edm_pickle(cfg, sd, precond='EDMPrecond'):
    """Bytes of a pickle with the layout of an NVIDIA EDM network pickle (edm/torch_utils/persistence.py:123-131): nested
    `torch_utils.persistence._reconstruct_persistent_obj(meta)` calls whose `state` is a torch.nn.Module `__dict__`.  The
    `module_src` field, which in a real file carries the reference's networks.py, holds a placeholder: this fixture contains
    no reference text.  tests/golden/check_pkl_loader.py checks the same reader against reference-written pickles."""
    import collections
    import pickle
    import sys
    import types

    def _reconstruct_persistent_obj(meta):          # never called: only its qualified name is pickled
        raise RuntimeError

    mod = types.ModuleType('torch_utils.persistence')
    _reconstruct_persistent_obj.__module__ = 'torch_utils.persistence'
    _reconstruct_persistent_obj.__qualname__ = '_reconstruct_persistent_obj'
    mod._reconstruct_persistent_obj = _reconstruct_persistent_obj
    pkg = types.ModuleType('torch_utils')
    pkg.persistence = mod

    class Obj:
        def __init__(self, class_name, **attrs):
            self.class_name = class_name
            self.state = dict(training=False, _parameters=collections.OrderedDict(), _buffers=collections.OrderedDict(),
                              _non_persistent_buffers_set=set(), _modules=collections.OrderedDict(), **attrs)

        def __reduce__(self):
            meta = dict(type='class', version=6, module_src='# (source text omitted in the synthetic fixture)',
                        class_name=self.class_name, state=self.state)
            return (_reconstruct_persistent_obj, (meta,))

    adm = cfg.arch == 'adm'
    init_kwargs = dict(img_resolution=cfg.img_resolution, in_channels=cfg.img_channels, out_channels=cfg.img_channels,
                       label_dim=cfg.label_dim, model_channels=cfg.model_channels, channel_mult=list(cfg.channel_mult),
                       num_blocks=cfg.num_blocks, attn_resolutions=list(cfg.attn_resolutions), augment_dim=cfg.augment_dim)
    if not adm:
        init_kwargs.update(embedding_type='positional', encoder_type='standard', decoder_type='standard', channel_mult_noise=1,
                           resample_filter=[1, 1], dropout=0.13)
    top = Obj(precond, img_resolution=cfg.img_resolution, img_channels=cfg.img_channels, label_dim=cfg.label_dim, use_fp16=False,
              sigma_min=cfg.sigma_min, sigma_max=cfg.sigma_max, sigma_data=cfg.sigma_data, _init_args=(), _init_kwargs=None)
    model = Obj('DhariwalUNet' if adm else 'SongUNet', _init_args=(), _init_kwargs=init_kwargs)
    top.state['_modules']['model'] = model
    for key, value in sd.items():
        parts = key.split('.')
        assert parts[0] == 'model'
        node = model
        for name in parts[1:-1]:
            node = node.state['_modules'].setdefault(name, Obj('Module'))
        node.state['_parameters'][parts[-1]] = torch.nn.Parameter(value.clone(), requires_grad=False)
    saved = {k: sys.modules.get(k) for k in ('torch_utils', 'torch_utils.persistence')}
    sys.modules['torch_utils'], sys.modules['torch_utils.persistence'] = pkg, mod
    try:
        return pickle.dumps(dict(ema=top))
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v


def full_weights(manifest_full, which):
    """(cfg, state dict) of a full-size BASELINE network: the product initialiser + the weight rule, pinned by the checksum the golden
    generator recorded for the weights it loaded into the REFERENCE module (tests/golden/make_golden_fullsize.py)."""
    from diffusion_tts_amd.config import adm_imagenet64, ddpmpp_cifar10
    if which == 'cls_imagenet64':
        cfg = ClassifierConfig()
        sd, _ = dinit.refill_degenerate(dinit.classifier_state_dict(cfg, manifest_full['cls_seed']), manifest_full['cls_seed'])
    else:
        cfg = {'adm_imagenet64': adm_imagenet64, 'ddpmpp_cifar10': ddpmpp_cifar10}[which]()
        sd, _ = dinit.refill_degenerate(dinit.edm_state_dict(cfg, manifest_full['net_seed']), manifest_full['net_seed'])
    ck, ref = dinit.checksum(sd), manifest_full[which]['checksum']
    assert ck['numel'] == ref['numel'] and abs(ck['abs_sum'] - ref['abs_sum']) < 1e-9 * ref['abs_sum'] \
        and abs(ck['sum'] - ref['sum']) < 1e-9 * max(1.0, abs(ref['sum'])), (which, ck, ref)
    return cfg, sd
