"""Oracle: reward functions of the EDM backend (torch CPU).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

  edm/scorers.py:25-54    BrightnessScorer.__call__
  edm/scorers.py:142-174  ImageNetScorer.__call__   (classifier itself: oracle/classifier.py)
  edm/scorers.py:176-243  CompressibilityScorer (PIL JPEG byte length; host codec, parity unpinned:
                          depends on the Pillow/libjpeg build)
"""
import io

import numpy as np
import torch

from .classifier import ClsCfg, encoder_unet


class BrightnessOracle:
    @torch.no_grad()
    def __call__(self, images, prompts, timesteps):
        if images.dtype == torch.uint8:
            images = images.float() / 255.0
        if images.size(1) == 3:
            w = torch.tensor([0.2126, 0.7152, 0.0722]).view(1, 3, 1, 1)
            lum = (images * w).sum(dim=1).mean(dim=(1, 2))
        else:
            lum = images.mean(dim=(1, 2))
        return torch.clamp(lum, 0.0, 1.0)


class ImageNetOracle:
    """softmax probability of the target class under the (random-init or loaded) classifier."""

    def __init__(self, cfg: ClsCfg, state_dict):
        self.cfg = cfg
        self.sd = {k: v.detach().float() for k, v in state_dict.items()}
        self.images_scored = 0

    @torch.no_grad()
    def __call__(self, images, class_labels, timesteps):
        if images.dtype == torch.uint8:
            images = images.float() / 255.0           # [0,1], not [-1,1] (scorers.py:153)
        logits = encoder_unet(self.sd, self.cfg, images, timesteps)
        probs = torch.softmax(logits, dim=1)
        target = torch.argmax(class_labels, dim=1) if class_labels.dim() > 1 else class_labels
        self.images_scored += images.shape[0]
        return probs[torch.arange(probs.size(0)), target]


class CompressibilityOracle:
    def __init__(self, quality=80, min_size=0, max_size=3000):
        self.quality, self.min_size, self.max_size = quality, min_size, max_size

    def _one(self, chw):
        from PIL import Image
        img = np.transpose(chw, (1, 2, 0)) if chw.ndim == 3 and chw.shape[0] in (1, 3) else chw
        if img.ndim == 3 and img.shape[2] == 1:
            img = img.squeeze(2)
        if img.dtype != np.uint8:
            img = (img * 255).astype(np.uint8) if img.max() <= 1.0 else img.astype(np.uint8)
        buf = io.BytesIO()
        Image.fromarray(img).save(buf, format='JPEG', quality=self.quality)
        size = len(buf.getvalue())
        return 1.0 - min(1.0, max(0.0, (size - self.min_size) / (self.max_size - self.min_size)))

    @torch.no_grad()
    def __call__(self, images, prompts, timesteps):
        if images.dim() == 4:
            return torch.tensor([self._one(im.cpu().numpy()) for im in images])
        return torch.tensor([self._one(images.cpu().numpy())])
