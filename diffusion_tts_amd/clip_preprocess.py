"""CLIP image pre-processing on the device for the SD search loop's scorer.

The reference scores every candidate image through `CLIPProcessor(images=...)` (sd/scorers.py:166-180): transformers'
`CLIPImageProcessor` = convert to a PIL image, resize the shorter edge to 224 with PIL's BICUBIC filter (antialiased: the filter support
scales with the reduction), centre-crop 224x224, rescale by 1/255 and normalise with CLIP's mean / std.  In the candidate-batched loop
that host path (images.cpu() + PIL per candidate) is the bottleneck of BASELINE config 4 (profiles/r02_sd_config4.txt: 126 decodes/s
with CLIP against 238 with a device-side scorer).  Here the same arithmetic runs on the GPU, bit for bit:

  * Pillow's 8-bit resampling is INTEGER arithmetic on coefficient tables derived from double-precision filter weights
    (Pillow 12.2 src/libImaging/Resample.c: precompute_coeffs, normalize_coeffs_8bpc, ImagingResampleHorizontal_8bpc /
    Vertical_8bpc).  `resample_tables` builds exactly those tables on the host (same IEEE double operations in the same order);
    `dts_resample_u8` applies them (horizontal pass, then vertical pass, uint8 in between, as Pillow does).
  * rescale + normalise map each of the 256 byte values of a channel to one float: `value_lut` obtains that table FROM THE IMAGE
    PROCESSOR ITSELF (a 224x224 probe image holding every byte value passes through it unresized), so whatever arithmetic and
    rounding order the installed transformers version uses is reproduced exactly.

`DevicePreprocessor.supports(...)` says when the device path is exact (square uint8 inputs, the stock resize -> centre-crop -> rescale
-> normalise configuration); everything else stays on the processor's own host path."""
import math

import numpy as np
import torch

from . import ops

PRECISION_BITS = 32 - 8 - 2          # Resample.c: 8-bit images keep 22 fractional bits


def _bicubic(x):
    """Resample.c bicubic_filter (a = -0.5)."""
    a = -0.5
    if x < 0.0:
        x = -x
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def resample_tables(in_size, out_size, support=2.0, filt=_bicubic):
    """(bounds int32 [out,2], coefs int32 [out,ksize]) of Pillow's precompute_coeffs + normalize_coeffs_8bpc for a full-image box."""
    in0, in1 = 0.0, float(in_size)
    scale = filterscale = (in1 - in0) / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = support * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    coefs = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = in0 + (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        w = [filt((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        if ww != 0.0:
            w = [v / ww for v in w]
        for x, v in enumerate(w):                         # normalize_coeffs_8bpc: round half away from zero, as the C casts do
            coefs[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, coefs


def value_lut(image_processor, channels=3):
    """lut f32 [channels, 256]: what the processor's rescale + normalise make of every byte value, read off the processor itself."""
    size = _target(image_processor)
    probe = np.zeros((size, size, channels), dtype=np.uint8)
    probe.reshape(-1, channels)[:256] = np.arange(256, dtype=np.uint8)[:, None]
    pix = image_processor(images=[torch.from_numpy(probe).permute(2, 0, 1).contiguous()], return_tensors='pt')['pixel_values'][0]
    return pix.reshape(channels, -1)[:, :256].to(torch.float32).contiguous()


def _edge(v):
    if v is None:
        return None
    if isinstance(v, int):
        return v
    g = v.get if hasattr(v, 'get') else (lambda k, d=None: getattr(v, k, d))
    for k in ('shortest_edge', 'height'):
        if g(k, None) is not None:
            return int(g(k))
    return None


def _target(image_processor):
    return _edge(getattr(image_processor, 'crop_size', None)) or _edge(getattr(image_processor, 'size', None))


class DevicePreprocessor:
    """pixel_values of a batch of square uint8 images [n,3,S,S] on the GPU, identical to `image_processor(images=...)`."""

    def __init__(self, image_processor, device):
        self.proc, self.device = image_processor, torch.device(device)
        p = image_processor
        self.size = _edge(getattr(p, 'size', None))
        crop = _edge(getattr(p, 'crop_size', None))
        res = getattr(p, 'resample', None)
        self.exact = bool(
            getattr(p, 'do_resize', False) and getattr(p, 'do_rescale', False) and getattr(p, 'do_normalize', False)
            and self.size is not None and int(res) == 3                                       # PIL.Image.BICUBIC
            and (not getattr(p, 'do_center_crop', False) or crop == self.size))
        self._tables = {}
        self._lut = None
        self._probed = False

    def _probe(self):
        """The attribute test above cannot tell Pillow's integer bicubic resampler from a torchvision-backed 'fast' processor with the same
        configuration (antialiased float bicubic): run ONE small non-trivial image through the processor itself and through the device path
        and keep the device path only if the pixel_values are identical."""
        self._probed = True
        g = torch.Generator().manual_seed(1234)
        img = torch.randint(0, 256, (1, 3, 64, 64), dtype=torch.uint8, generator=g)
        try:
            want = self.proc(images=[img[0].permute(1, 2, 0).numpy()], return_tensors='pt')['pixel_values'].to(torch.float32)
            got = self(img.to(self.device)).cpu()
            ok = tuple(want.shape) == tuple(got.shape) and torch.equal(want, got)
        except Exception:
            ok = False
        if not ok:
            self.exact = False

    def supports(self, images):
        if not self.exact:
            return False
        if isinstance(images, list):
            ok = len(images) > 0 and all(isinstance(im, torch.Tensor) and im.is_cuda and im.dtype == torch.uint8 and im.dim() in (3, 4)
                                         and im.shape[-3] == 3 and im.shape[-1] == im.shape[-2] and (im.dim() == 3 or im.shape[0] == 1)
                                         and im.shape[-1] == images[0].shape[-1] for im in images)
        else:
            ok = (isinstance(images, torch.Tensor) and images.is_cuda and images.dtype == torch.uint8 and images.dim() == 4
                  and images.shape[1] == 3 and images.shape[2] == images.shape[3])
        if ok and not self._probed:
            self._probe()
        return ok and self.exact

    def __call__(self, images):
        if isinstance(images, list):
            images = torch.cat([im if im.dim() == 4 else im.unsqueeze(0) for im in images], dim=0)
        images = images.to(self.device).contiguous()
        n, c, s, _ = images.shape
        if self._lut is None:
            self._lut = value_lut(self.proc, c).to(self.device)
        x = images.view(n * c, s, s)
        if s != self.size:                       # Image.resize returns a copy when the size already matches
            tab = self._tables.get(s)
            if tab is None:
                b, k = resample_tables(s, self.size)
                tab = self._tables[s] = (torch.from_numpy(b).to(self.device), torch.from_numpy(k).to(self.device))
            x = ops.resample_u8(x, self.size, 1, *tab)             # horizontal pass first, uint8 in between (ImagingResample)
            x = ops.resample_u8(x, self.size, 0, *tab)
        return ops.lut_u8_f32(x.view(n, c, self.size, self.size), self._lut)
