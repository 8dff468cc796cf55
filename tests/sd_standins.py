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


class ShapeVAE(nn.Module):
    """Decoder stand-in with SD-1.5's tensor shapes: latents [n,4,64,64] -> images [n,3,512,512] (three nearest-2x stages, the
    AutoencoderKL decoder's up path in miniature; `block_out_channels` has 4 entries so the pipeline's vae_scale_factor is 8)."""

    def __init__(self, latent_channels=4, width=8, seed=1):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        self.config = types.SimpleNamespace(scaling_factor=0.18215, block_out_channels=[width] * 4, latent_channels=latent_channels,
                                            force_upcast=False)
        self.c0 = nn.Conv2d(latent_channels, width, 3, padding=1)
        self.c1 = nn.Conv2d(width, width, 3, padding=1)
        self.c2 = nn.Conv2d(width, width, 3, padding=1)
        self.out = nn.Conv2d(width, 3, 3, padding=1)
        for p in self.parameters():
            with torch.no_grad():
                p.copy_(torch.randn(p.shape, generator=g) * (0.7 / math.sqrt(max(1, p[0].numel()))))

    @property
    def dtype(self):
        return self.out.weight.dtype

    @property
    def device(self):
        return self.out.weight.device

    def decode(self, z, return_dict=False, generator=None):
        f = torch.nn.functional
        h = f.silu(self.c0(z))
        h = f.silu(self.c1(f.interpolate(h, scale_factor=2.0, mode='nearest')))
        h = f.silu(self.c2(f.interpolate(h, scale_factor=2.0, mode='nearest')))
        return (torch.tanh(self.out(f.interpolate(h, scale_factor=2.0, mode='nearest'))),)


def shape_unet(seed=0):
    """U-Net stand-in at SD-1.5's I/O shapes: sample [n,4,64,64], encoder_hidden_states [n,77,768]."""
    return TinyUNet(in_channels=4, width=16, ctx_dim=768, sample_size=64, seed=seed)


class TinyTextEncoder(nn.Module):
    """Text-encoder stand-in with CLIPTextModel's call surface: te(input_ids, attention_mask=None)[0] -> [n, 77, 768]."""

    def __init__(self, vocab=512, dim=768, seed=2):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        self.config = types.SimpleNamespace(use_attention_mask=False)
        self.emb = nn.Embedding(vocab, dim)
        self.pos = nn.Parameter(torch.zeros(77, dim))
        self.mix = nn.Linear(dim, dim)
        for p in self.parameters():
            with torch.no_grad():
                p.copy_(torch.randn(p.shape, generator=g) * (0.5 / math.sqrt(max(1, p[0].numel() if p.dim() > 1 else 1))))

    @property
    def dtype(self):
        return self.mix.weight.dtype

    @property
    def device(self):
        return self.mix.weight.device

    def forward(self, input_ids, attention_mask=None):
        h = self.emb(input_ids) + self.pos[:input_ids.shape[1]]
        return (torch.tanh(self.mix(h)),)


class TinyTokenizer:
    """Tokenizer stand-in with the CLIPTokenizer calling convention `encode_prompt` uses (pipeline...:390-396)."""
    model_max_length = 77

    def __init__(self, vocab=512):
        self.vocab = vocab

    added_tokens_encoder = {}

    def tokenize(self, text):
        return text.split()

    def __call__(self, texts, padding='max_length', max_length=None, truncation=False, return_tensors='pt'):
        if isinstance(texts, str):
            texts = [texts]
        rows = [[self.vocab - 2] + [b % (self.vocab - 2) for b in t.encode('utf-8')] + [self.vocab - 1] for t in texts]
        if truncation and max_length:
            rows = [r[:max_length - 1] + [self.vocab - 1] if len(r) > max_length else r for r in rows]
        width = max_length if padding == 'max_length' else max(len(r) for r in rows)
        ids = torch.tensor([r + [self.vocab - 1] * (width - len(r)) for r in rows], dtype=torch.long)
        return types.SimpleNamespace(input_ids=ids, attention_mask=torch.ones_like(ids))

    def batch_decode(self, ids):
        return [''] * len(ids)


def tiny_clip(seed=0):
    """Random-init CLIP with the real model's structure at toy width (transformers.CLIPModel): 224-pixel vision tower with 32-pixel
    patches, 77-position text tower; BASELINE config 4 prescribes random-init weights for the CLIP scorer."""
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        from transformers import CLIPConfig, CLIPModel, CLIPTextConfig, CLIPVisionConfig
        tc = CLIPTextConfig(vocab_size=1000, hidden_size=32, intermediate_size=64, num_hidden_layers=2, num_attention_heads=2,
                            max_position_embeddings=77, projection_dim=16, bos_token_id=998, eos_token_id=999, pad_token_id=999)
        vc = CLIPVisionConfig(hidden_size=32, intermediate_size=64, num_hidden_layers=2, num_attention_heads=2, image_size=224,
                              patch_size=32, projection_dim=16)
        cfg = CLIPConfig(text_config=tc.to_dict(), vision_config=vc.to_dict(), projection_dim=16)
        torch.manual_seed(seed)
        return CLIPModel(cfg).eval()
