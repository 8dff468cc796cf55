"""Oracle: Pillow's 8-bit bicubic resize restated in numpy (TEST INFRASTRUCTURE ONLY, see oracle/__init__.py).

The SD backend's CLIP scorer (sd/scorers.py:166-180) hands every candidate image to `CLIPProcessor`, whose image half is Pillow's
`Image.resize(..., resample=BICUBIC)`.  Pillow is a third-party dependency of the reference (not vendored under /root/reference; the
version installed here is Pillow 12.2.0); the algorithm restated is its published src/libImaging/Resample.c:
  bicubic_filter, precompute_coeffs, normalize_coeffs_8bpc, ImagingResampleHorizontal_8bpc, ImagingResampleVertical_8bpc.
Pinned by tests/test_clip_preprocess.py against Pillow itself (`PIL.Image.resize`) and against transformers' CLIPImageProcessor on
random uint8 images, in this container."""
import math

import numpy as np


def bicubic(x, a=-0.5):
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def coefficients(in_size, out_size, support=2.0):
    """precompute_coeffs + normalize_coeffs_8bpc for the box (0, in_size): [(first, [int coefficient, ...]), ...] per output sample."""
    scale = filterscale = in_size / out_size
    filterscale = max(filterscale, 1.0)
    sup = support * filterscale
    inv = 1.0 / filterscale
    out = []
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - sup + 0.5), 0)
        xmax = min(int(center + sup + 0.5), in_size) - xmin
        w = [bicubic((x + xmin - center + 0.5) * inv) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        if ww != 0.0:
            w = [v / ww for v in w]
        out.append((xmin, [int(-0.5 + v * (1 << 22)) if v < 0 else int(0.5 + v * (1 << 22)) for v in w]))
    return out


def _pass(img, out_size):
    """one pass along the LAST axis of a uint8 array: clip8((2^21 + sum pixel * k) >> 22)."""
    tabs = coefficients(img.shape[-1], out_size)
    out = np.empty(img.shape[:-1] + (out_size,), dtype=np.uint8)
    src = img.astype(np.int64)
    for xx, (first, ks) in enumerate(tabs):
        acc = np.full(img.shape[:-1], 1 << 21, dtype=np.int64)
        for t, k in enumerate(ks):
            acc += src[..., first + t] * k
        out[..., xx] = np.clip(acc >> 22, 0, 255).astype(np.uint8)
    return out


def resize_bicubic_u8(img, out_h, out_w):
    """img uint8 [..., H, W] -> [..., out_h, out_w]: horizontal pass, then vertical pass (ImagingResample), uint8 in between."""
    x = img
    if img.shape[-1] != out_w:
        x = _pass(x, out_w)
    if img.shape[-2] != out_h:
        x = np.swapaxes(_pass(np.ascontiguousarray(np.swapaxes(x, -1, -2)), out_h), -1, -2)
    return np.ascontiguousarray(x)
