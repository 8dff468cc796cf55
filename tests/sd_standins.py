"""Tiny stand-ins for the diffusers U-Net / VAE of the SD backend (pure torch modules, deterministic init).

The SD search loop treats the U-Net and the VAE as opaque callables ("diffusers U-Net path", BASELINE config 4);
these stand-ins have the call surface the reference pipeline uses (`unet(x, t, encoder_hidden_states=..., ...,
return_dict=False)[0]`, `vae.decode(z, return_dict=False, generator=None)[0]`, `.config.*`) so the SAME modules can
be driven by the reference's `StableDiffusionPipeline.__call__` (golden generation, CPU) and by this build's loop
(GPU tests)."""
import math
import types

import torch
import torch.nn as nn


class TinyUNet(nn.Module):
    def __init__(self, in_channels=4, width=16, ctx_dim=8, sample_size=8, seed=0):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        self.config = types.SimpleNamespace(in_channels=in_channels, sample_size=sample_size, time_cond_proj_dim=None,
                                            _diffusers_version='0.33.0.dev0', addition_embed_type=None)
        self.conv_in = nn.Conv2d(in_channels, width, 3, padding=1)
        self.time = nn.Linear(16, width)
        self.ctx = nn.Linear(ctx_dim, width)
        self.conv_mid = nn.Conv2d(width, width, 3, padding=1)
        self.conv_out = nn.Conv2d(width, in_channels, 3, padding=1)
        for p in self.parameters():
            with torch.no_grad():
                p.copy_(torch.randn(p.shape, generator=g) * (0.3 / math.sqrt(max(1, p[0].numel()))))

    @property
    def dtype(self):
        return self.conv_in.weight.dtype

    @property
    def device(self):
        return self.conv_in.weight.device

    def forward(self, sample, timestep, encoder_hidden_states=None, timestep_cond=None, cross_attention_kwargs=None,
                added_cond_kwargs=None, return_dict=False):
        t = torch.as_tensor(timestep, device=sample.device, dtype=torch.float32).reshape(-1)
        if t.numel() == 1:
            t = t.expand(sample.shape[0])
        freqs = torch.exp(-math.log(10000.0) * torch.arange(8, device=sample.device, dtype=torch.float32) / 8)
        emb = torch.cat([torch.sin(t[:, None] * freqs), torch.cos(t[:, None] * freqs)], dim=1).to(sample.dtype)
        h = self.conv_in(sample)
        h = h + self.time(emb)[:, :, None, None] + self.ctx(encoder_hidden_states.mean(dim=1))[:, :, None, None]
        h = self.conv_mid(torch.nn.functional.silu(h))
        return (self.conv_out(torch.nn.functional.silu(h)),)


class TinyVAE(nn.Module):
    def __init__(self, latent_channels=4, seed=1):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        self.config = types.SimpleNamespace(scaling_factor=0.18215, block_out_channels=[8, 8], latent_channels=latent_channels,
                                            force_upcast=False)
        self.up = nn.ConvTranspose2d(latent_channels, 8, 2, stride=2)
        self.out = nn.Conv2d(8, 3, 3, padding=1)
        for p in self.parameters():
            with torch.no_grad():
                p.copy_(torch.randn(p.shape, generator=g) * (0.5 / math.sqrt(max(1, p[0].numel()))))

    @property
    def dtype(self):
        return self.out.weight.dtype

    @property
    def device(self):
        return self.out.weight.device

    def decode(self, z, return_dict=False, generator=None):
        return (torch.tanh(self.out(torch.nn.functional.silu(self.up(z)))),)
