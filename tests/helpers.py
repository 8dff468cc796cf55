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
