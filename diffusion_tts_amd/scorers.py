"""Reward functions ("scorers") of the EDM backend with the reference's plugin surface
(edm/scorers.py:14-23): any object callable as `scorer(images, labels_or_prompts, timesteps) -> Tensor[len(images)]`.

  BrightnessScorer     edm/scorers.py:25-54    -> dts_brightness  (HIP, on the images' device)
  ImageNetScorer       edm/scorers.py:56-174   -> classifier.EncoderUNetModel + dts_softmax_gather (HIP).
                       The reference keeps this scorer on the CPU (main.py:69); here it runs beside the denoiser.
  CompressibilityScorer edm/scorers.py:176-243 -> host PIL JPEG byte length, unchanged: entropy coding is a CPU
                       codec and an opaque callable to the search loop.
"""
import io
import os
import warnings

import numpy as np
import torch

from . import ops
from .config import ClassifierConfig


class Scorer:
    """Base class for all scorers (edm/scorers.py:14-23)."""

    def __init__(self, dtype=torch.float32):
        self.dtype = dtype

    def eval(self):
        return self

    def __call__(self, images, prompts, timesteps):
        raise NotImplementedError('Subclasses must implement __call__')


def _as_u8_batch(images, device):
    if isinstance(images, list):                      # list of PIL images (edm/scorers.py:31-34) or of tensors (sd/scorers.py:31-51)
        parts = []
        for im in images:
            if isinstance(im, torch.Tensor):
                parts.append(im if im.dim() == 4 else im.unsqueeze(0))
            else:
                parts.append(torch.from_numpy(np.array(im)).permute(2, 0, 1).unsqueeze(0))
        images = torch.cat([p.to(device) for p in parts], dim=0)
    if not isinstance(images, torch.Tensor):
        raise TypeError(f'unsupported image container {type(images)}')
    if images.dtype != torch.uint8:
        raise ValueError('the HIP scorers take uint8 images in [0,255] (what the search loop passes, edm/main.py:126)')
    return images.to(device).contiguous()


class BrightnessScorer(Scorer):
    @torch.no_grad()
    def __call__(self, images, prompts, timesteps):
        dev = images.device if isinstance(images, torch.Tensor) and images.is_cuda else torch.device('cuda')
        img = _as_u8_batch(images, dev)
        if img.dim() != 4 or img.size(1) != 3:
            raise ValueError('BrightnessScorer expects [n,3,h,w]')
        return ops.brightness(img)


class ImageNetScorer(Scorer):
    """softmax probability of the target class under the 64x64 noisy ImageNet classifier.

    The reference downloads OpenAI's `64x64_classifier.pt` (edm/scorers.py:61-74).  No network here: weights are
    read from `weights` (a state dict or a torch-saved file, e.g. ~/.cache/imagenet_classifier/64x64_classifier.pt
    if present) or, failing that, random-initialised with the documented weight rule (BASELINE: random-init)."""

    CACHE = os.path.expanduser('~/.cache/imagenet_classifier/64x64_classifier.pt')

    def __init__(self, dtype=torch.float32, weights=None, cfg: ClassifierConfig = None, seed=1, device='cuda',
                 compute_dtype=torch.bfloat16):
        super().__init__(dtype)
        from .classifier import EncoderUNetModel
        from . import init as dinit
        self.cfg = cfg or ClassifierConfig()
        if weights is None and os.path.exists(self.CACHE):
            weights = self.CACHE
        if isinstance(weights, str):
            weights = torch.load(weights, map_location='cpu')
        if weights is None:
            warnings.warn('ImageNetScorer: no classifier weights available offline; using random-init weights '
                          f'(seed {seed}, zero-init layers re-drawn)')
            weights, _ = dinit.refill_degenerate(dinit.classifier_state_dict(self.cfg, seed), seed)
        self.model = EncoderUNetModel(self.cfg, weights, device=device, dtype=compute_dtype)
        self.device = torch.device(device)
        self.images_scored = 0

    @torch.no_grad()
    def __call__(self, images, class_labels, timesteps):
        img = _as_u8_batch(images, self.device)
        x = ops.u8_to_unit_f32(img)                                      # [0,1] (edm/scorers.py:153)
        n = x.shape[0]
        ts = torch.zeros(n, device=self.device) if timesteps is None else timesteps
        logits = self.model(x, ts)
        class_labels = class_labels.to(self.device)
        target = torch.argmax(class_labels, dim=1) if class_labels.dim() > 1 else class_labels
        self.images_scored += n
        return ops.softmax_gather(logits, target.to(torch.int32).contiguous())


class CompressibilityScorer(Scorer):
    def __init__(self, quality=80, min_size=0, max_size=3000, dtype=torch.float32):
        super().__init__(dtype)
        self.quality, self.min_size, self.max_size = quality, min_size, max_size

    def _score(self, image):
        from PIL import Image
        if image.ndim == 3:
            if image.shape[0] in (1, 3):
                image = np.transpose(image, (1, 2, 0))
            if image.shape[2] == 1:
                image = image.squeeze(2)
        if not (image.ndim == 2 or (image.ndim == 3 and image.shape[2] in (1, 3, 4))):
            raise ValueError(f'Invalid image shape: {image.shape}')
        if image.dtype != np.uint8:
            image = (image * 255).astype(np.uint8) if image.max() <= 1.0 else image.astype(np.uint8)
        buf = io.BytesIO()
        Image.fromarray(image).save(buf, format='JPEG', quality=self.quality)
        size = len(buf.getvalue())
        return 1.0 - min(1.0, max(0.0, (size - self.min_size) / (self.max_size - self.min_size)))

    @torch.no_grad()
    def __call__(self, images, prompts, timesteps):
        if isinstance(images, torch.Tensor):
            if images.dim() == 4:
                return torch.tensor([self._score(im.cpu().numpy()) for im in images])
            return torch.tensor([self._score(images.cpu().numpy())])
        if isinstance(images, list):
            return torch.tensor([self._score(np.array(im)) for im in images])
        return torch.tensor([self._score(np.array(images))])
