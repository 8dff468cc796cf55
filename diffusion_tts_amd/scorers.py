"""Reward functions ("scorers") of the EDM backend with the reference's plugin surface
(edm/scorers.py:14-23): any object callable as `scorer(images, labels_or_prompts, timesteps) -> Tensor[len(images)]`.

  BrightnessScorer     edm/scorers.py:25-54    -> dts_brightness  (HIP, on the images' device)
  ImageNetScorer       edm/scorers.py:56-174   -> classifier.EncoderUNetModel + dts_softmax_gather (HIP).
                       The reference keeps this scorer on the CPU (main.py:69); here it runs beside the denoiser.
  CompressibilityScorer edm/scorers.py:176-243 -> host PIL JPEG byte length, unchanged: entropy coding is a CPU
                       codec and an opaque callable to the search loop.
  CLIPScorer           sd/scorers.py:149-213   -> HF CLIP image/text towers (third-party arithmetic, run as PyTorch-ROCm
                       modules on the GPU) + dts_cosine_rows (HIP) for the normalise-and-dot tail.  The reference fetches
                       `openai/clip-vit-large-patch14`; there is no network here, so the model / image processor / tokenizer
                       can be injected (random-init `CLIPModel(CLIPConfig)` for BASELINE config 4).
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
    batched = True

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
                 compute_dtype=ops.F16X3):
        """compute_dtype: the classifier's activation / weight type.  Default ops.F16X3 (split precision: the reference scores in fp32,
        main.py:69, and the search's argmax is taken on these rewards, edm/main.py:842).  Beside a bfloat16 denoiser use float16: the reward
        error of a 16-bit search iteration is the CLASSIFIER's (measured, tools/dtype_mix.py: bf16 scorer 2.0e-6 whatever the
        denoiser's type; f16 scorer 2.2e-7 ... 4.4e-7), the classifier is 8 % of an iteration's FLOPs, and float16 is what the
        reference's networks use on a GPU (edm/training/networks.py:658); bfloat16 is accepted, float32 is the parity mode."""
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


class ByteTokenizer:
    """Stand-in for CLIP's BPE tokenizer when its vocabulary files are not on disk (no network): UTF-8 bytes shifted past the
    special ids, `bos ... eos`, padded with eos to the longest prompt of the call, truncated to `max_length` -- the calling
    convention `CLIPScorer` uses (sd/scorers.py:194-200: padding=True, truncation=True, max_length=77, return_tensors='pt').
    Deterministic and fixed for a given prompt, which is all a random-init CLIP needs; NOT CLIP's vocabulary."""

    def __init__(self, vocab_size=49408, bos_token_id=None, eos_token_id=None):
        self.vocab_size = vocab_size
        self.bos = vocab_size - 2 if bos_token_id is None else bos_token_id
        self.eos = vocab_size - 1 if eos_token_id is None else eos_token_id

    def __call__(self, prompts, padding=True, truncation=True, max_length=77, return_tensors='pt'):
        if isinstance(prompts, str):
            prompts = [prompts]
        body = self.vocab_size - 2
        rows = [[self.bos] + [b % body for b in p.encode('utf-8')][:max_length - 2] + [self.eos] for p in prompts]
        width = max(len(r) for r in rows)
        ids = torch.tensor([r + [self.eos] * (width - len(r)) for r in rows], dtype=torch.long)
        mask = torch.tensor([[1] * len(r) + [0] * (width - len(r)) for r in rows], dtype=torch.long)
        return _Encoding(input_ids=ids, attention_mask=mask)


class _Encoding(dict):
    def to(self, device):
        return _Encoding({k: v.to(device) for k, v in self.items()})


def _features(out):
    """`CLIPModel.get_image_features` / `get_text_features` return the projected embedding as a tensor (transformers 4.x, what
    the reference was written against) or inside `BaseModelOutputWithPooling.pooler_output` (transformers 5.x)."""
    return out if isinstance(out, torch.Tensor) else out.pooler_output


class CLIPScorer(Scorer):
    """Cosine similarity between the CLIP embeddings of the prompt and of each image (sd/scorers.py:149-213).

    `CLIPScorer()` behaves like the reference's: it loads `model_id` through transformers -- from the local HF cache only,
    since nothing can be downloaded here, and raises with that message when the files are missing.  For runs without the
    pretrained files pass `model=` (a `transformers.CLIPModel`, e.g. random-init from a `CLIPConfig`), optionally
    `image_processor=` (default `CLIPImageProcessor()`: the stock 224 / CLIP-mean-std pipeline, no files needed) and
    `tokenizer=` (default `ByteTokenizer` sized to the model's vocabulary)."""

    batched = True        # one call scores a whole list / batch of images (the SD loop then calls it once per decoded batch)

    def __init__(self, model_id='openai/clip-vit-large-patch14', dtype=torch.float32, model=None, image_processor=None,
                 tokenizer=None, processor=None, device='cuda', device_preprocess=True):
        """device_preprocess: run the image processor's resize / rescale / normalise on the GPU (clip_preprocess.DevicePreprocessor:
        Pillow's integer bicubic + the processor's own value table, identical pixel_values) whenever the images are square uint8 GPU
        tensors and the processor has the stock configuration; False keeps the reference's host path (images.cpu() + PIL) always."""
        super().__init__(dtype)
        self.device = torch.device(device)
        if processor is not None:                                     # a full CLIPProcessor, as the reference holds
            image_processor = image_processor or getattr(processor, 'image_processor', processor)
            tokenizer = tokenizer or getattr(processor, 'tokenizer', None)
        if model is None:
            try:
                from transformers import CLIPModel, CLIPProcessor
                model = CLIPModel.from_pretrained(model_id, local_files_only=True)
                proc = CLIPProcessor.from_pretrained(model_id, local_files_only=True)
                image_processor, tokenizer = proc.image_processor, proc.tokenizer
            except Exception as e:
                raise RuntimeError(f'CLIPScorer: {model_id!r} is not in the local Hugging Face cache and cannot be downloaded here; '
                                   f'pass model= (a transformers.CLIPModel) [and image_processor=/tokenizer=]') from e
        if image_processor is None:
            import warnings
            with warnings.catch_warnings():
                warnings.simplefilter('ignore')
                from transformers import CLIPImageProcessor
                image_processor = CLIPImageProcessor()
        if tokenizer is None:
            tc = model.config.text_config
            tokenizer = ByteTokenizer(tc.vocab_size, getattr(tc, 'bos_token_id', None), getattr(tc, 'eos_token_id', None))
            if not (0 <= tokenizer.bos < tc.vocab_size and 0 <= tokenizer.eos < tc.vocab_size):
                tokenizer = ByteTokenizer(tc.vocab_size)
        self.clip = model.to(self.device).eval()
        self.image_processor, self.tokenizer = image_processor, tokenizer
        self._text_cache = {}
        self.device_preprocessed = 0                                   # images whose pixel_values were built on the GPU
        self._dev_pre = None
        if device_preprocess:
            from .clip_preprocess import DevicePreprocessor
            self._dev_pre = DevicePreprocessor(image_processor, self.device)

    @torch.no_grad()
    def __call__(self, images, prompts, timesteps=None):
        dev = self.device
        # sd/scorers.py:170-173: float images already in [0,1] are not rescaled, everything else (uint8) is
        if self._dev_pre is not None and self._dev_pre.supports(images):
            pix = self._dev_pre(images)                                # uint8 GPU images: the processor's arithmetic on the device
            self.device_preprocessed += pix.shape[0]
        else:
            do_rescale = not (isinstance(images, torch.Tensor) and images.dtype == torch.float32 and images.max() <= 1.0)
            if isinstance(images, list):                               # the SD loop passes [uint8 tensor [1,3,H,W]] (pipeline...:1114)
                images = [im[0] if (isinstance(im, torch.Tensor) and im.dim() == 4) else im for im in images]
                images = [im.cpu() if isinstance(im, torch.Tensor) else im for im in images]
            elif isinstance(images, torch.Tensor):
                images = images.cpu()
            pix = self.image_processor(images=images, return_tensors='pt', do_rescale=do_rescale)['pixel_values']
        img_emb = _features(self.clip.get_image_features(pixel_values=pix.to(self.dtype).to(dev))).float().contiguous()
        n = img_emb.shape[0]
        if prompts is None:
            return torch.zeros(n, device=dev)                          # :186-188
        if not isinstance(prompts, list):
            prompts = [prompts] * n
        elif len(prompts) == 1 and n > 1:
            prompts = prompts * n
        key = tuple(prompts)
        txt_emb = self._text_cache.get(key)                            # the prompt is fixed over a search: encode it once
        if txt_emb is None:
            enc = self.tokenizer(prompts, padding=True, truncation=True, max_length=77, return_tensors='pt').to(dev)
            txt_emb = _features(self.clip.get_text_features(**enc)).float().contiguous()
            if len(self._text_cache) < 64:
                self._text_cache[key] = txt_emb
        return ops.cosine_rows(img_emb, txt_emb).to(self.dtype)        # :182-183,205-211 (normalise both, row-wise dot)
