"""Thin tensor-level wrappers over the C ABI (include/dts.h).  PyTorch supplies device memory and the
current HIP stream; every computation happens in libdts_hip.so.  Inputs must live on the GPU: there is
no CPU path here (the CPU restatement lives in oracle/ and is test infrastructure only).

Activation tensors are NHWC: shape [n, h, w, c], contiguous, dtype float32 / bfloat16 / float16.
"""
import ctypes as C
import math
import os

import torch

from . import _lib as L

_DT = {torch.float32: L.DTS_F32, torch.bfloat16: L.DTS_BF16, torch.float16: L.DTS_F16}

# The split-precision compute mode (dts.h DTS_F16X3), the DEFAULT of every surface that takes a compute dtype: activations are float32
# tensors and every kernel but the convolutions (and the d = 64 attention) is the float32 (parity-mode) one; the convolutions -- 99.9 % of
# the FLOPs -- run on the 16-bit matrix cores over the f16 split image of their input (per 32 channels: hi | lo * 2^11) against weights
# packed per 32 input channels as hi | lo, three MFMAs per staged K step (hi.hi, lo.hi, hi.lo), f32 accumulate: ~2^-22 products instead
# of f16's 2^-11, at a third of the f16 rate instead of the f32 matrix instruction's sixteenth.
F16X3 = 'f16x3'


def act_dtype(dtype):
    """storage type of the activations of a network computing in `dtype`"""
    return torch.float32 if dtype == F16X3 else dtype


def dtype_name(dtype):
    return dtype if isinstance(dtype, str) else {torch.float32: 'f32', torch.bfloat16: 'bf16', torch.float16: 'f16'}[dtype]


class X3Weight:
    """A conv weight packed for the split-precision mode: `packed` [O][kh][kw][2*I] float16 = per 32 input channels hi(32) | lo(32) of
    w * 2^k (k chosen per layer so that 2^13 < max|w| * 2^k <= 2^14: the lo parts of all weights that matter are normal f16 numbers;
    the matrix cores flush subnormal inputs), `acc_scale` = 2^-k.  The kernels scale the hi fragment by 2^-11 in registers where it meets
    the activations' lo * 2^11 half."""

    def __init__(self, packed, acc_scale, shape):
        self.packed, self.acc_scale, self.shape = packed, float(acc_scale), tuple(shape)
        self.device, self.dtype = packed.device, F16X3

    def numel(self):
        return self.packed.numel()


class SplitQKV:
    """The qkv projection's result in the split-precision attention's operand form: `data` float16 [n,t,6*C] = hi(3C) | lo(3C) of the
    values * 2^6 (dts_split2_f16's image, written by conv2d(..., out_split2=True)); `shape` is the logical [n, h, w, 3*C]."""

    def __init__(self, data, shape):
        self.data, self.shape = data, tuple(shape)
        self.device, self.dtype = data.device, torch.float32

    def view(self, n, t, c3):
        return SplitQKV(self.data.view(n, t, 2 * c3), (n, t, c3))

    def numel(self):                     # logical elements; each is an f16 pair = 4 bytes, like the f32 value it stands for
        return self.data.numel() // 2

    def element_size(self):
        return 4


class SplitAct:
    """An activation already in the split-precision operand form: `data` float16 [n,h,w,2*C], per 32 channels hi(32) | lo * 2^11 (32)
    (dts_split3_f16 / dts_gn_apply_x3 / dts_attention_x3); `shape` is the logical NHWC shape.  Only convolutions with an X3Weight read it."""

    def __init__(self, data, c):
        self.data, self.shape = data, tuple(data.shape[:3]) + (c,)
        self.device, self.dtype = data.device, torch.float32

    def numel(self):
        return self.data.numel() // 2

    def planes(self):
        """(hi, lo * 2^11) as two float16 tensors of the logical shape (tests / debugging)"""
        return split_planes(self.data, self.shape[-1])


def split_planes(data, c):
    """the two halves of a split-precision operand image `data` [..., 2*c] as tensors [..., c]: (hi, lo * 2^11)"""
    v = data.reshape(*data.shape[:-1], c // 32, 2, 32)
    return v[..., 0, :].reshape(*data.shape[:-1], c), v[..., 1, :].reshape(*data.shape[:-1], c)


def dt_code(dtype):
    try:
        return _DT[dtype]
    except KeyError:
        raise ValueError(f'unsupported activation dtype {dtype}')


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _ptr(t, name='tensor', dtype=None):
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError(f'{name} must be a GPU tensor (the HIP path has no CPU fallback)')
    if not t.is_contiguous():
        raise ValueError(f'{name} must be contiguous')
    if dtype is not None and t.dtype != dtype:
        raise ValueError(f'{name}: expected {dtype}, got {t.dtype}')
    return t.data_ptr()


def _rows(t, name, dtype):
    """(pointer, leading dimension) of a 2-D row-major view whose rows may be strided (a column slice)."""
    if t is None:
        return None, 0
    if not t.is_cuda:
        raise RuntimeError(f'{name} must be a GPU tensor (the HIP path has no CPU fallback)')
    if t.dim() != 2 or t.stride(1) != 1 or t.dtype != dtype:
        raise ValueError(f'{name}: expected a 2-D {dtype} view with unit inner stride')
    return t.data_ptr(), t.stride(0)


def _call(name, *args):
    lib = L.load()
    L.check(getattr(lib, name)(*args, _stream()), name)


# ---- layout / packing ---------------------------------------------------------------------------
def nchw_to_nhwc(x, dtype):
    n, c, h, w = x.shape
    out = torch.empty((n, h, w, c), dtype=dtype, device=x.device)
    _call('dts_nchw_to_nhwc', _ptr(x, 'x', torch.float32), _ptr(out), dt_code(dtype), n, c, h, w)
    return out


def nchw_to_nhwc_pad(x, dtype, cpad):
    """f32 NCHW [n,c,h,w] -> NHWC [n,h,w,cpad] in `dtype`, channels c.. zero."""
    n, c, h, w = x.shape
    out = torch.empty((n, h, w, cpad), dtype=dtype, device=x.device)
    _call('dts_nchw_to_nhwc_pad', _ptr(x, 'x', torch.float32), _ptr(out), dt_code(dtype), n, c, h, w, cpad)
    return out


def nhwc_to_nchw(x):
    n, h, w, c = x.shape
    out = torch.empty((n, c, h, w), dtype=torch.float32, device=x.device)
    _call('dts_nhwc_to_nchw', _ptr(x), dt_code(x.dtype), _ptr(out), n, c, h, w)
    return out


def pack_conv_weight(w, dtype, out_perm=None):
    """OIHW float32 (device) -> [O][kh][kw][I] in `dtype`; out_perm: int32 device tensor of source rows."""
    if w.dim() == 3:                      # conv1d weight [O, I, 1]
        w = w.unsqueeze(-1)
    w = w.contiguous()
    O, I, kh, kw = w.shape
    if dtype == F16X3:                    # load-time preparation in float32 on the device: exact scaling, two roundings
        wp = w if out_perm is None else w[out_perm.long()]
        amax = float(wp.abs().max())
        k = 0 if amax == 0.0 else max(-24, min(24, int(math.floor(math.log2(16384.0 / amax)))))
        ws = (wp * (2.0 ** k)).permute(0, 2, 3, 1).contiguous()            # [O][kh][kw][I], exact (power of two)
        if I % 32:
            raise ValueError(f'pack_conv_weight(F16X3): {I} input channels are not a multiple of 32 (one K step of the split image)')
        hi = ws.to(torch.float16)
        lo = (ws - hi.to(torch.float32)).to(torch.float16)
        packed = torch.stack([hi.view(O, kh, kw, I // 32, 32), lo.view(O, kh, kw, I // 32, 32)], dim=-2)      # [.., I/32, 2, 32]: hi(32) | lo(32)
        return X3Weight(packed.reshape(O, kh, kw, 2 * I).contiguous(), 2.0 ** -k, (O, kh, kw, I))
    out = torch.empty((O, kh, kw, I), dtype=dtype, device=w.device)
    _call('dts_pack_conv_weight', _ptr(w, 'w', torch.float32), _ptr(out), dt_code(dtype), O, I, kh, kw,
          _ptr(out_perm, 'perm', torch.int32))
    return out


# ---- convolution --------------------------------------------------------------------------------
_CONV_WS = {}
CONV_WS_BYTES = 96 << 20        # split-K scratch per device (covers 8 splits of the 8x8 / 16x16 ADM levels at N=64)


def _conv_workspace(device):
    key = (device.type, device.index)
    ws = _CONV_WS.get(key)
    if ws is None:
        ws = _CONV_WS[key] = torch.empty(CONV_WS_BYTES, dtype=torch.uint8, device=device)
    return ws


def _conv_args(x1, w, x2, up):
    n, hin, win, c1 = x1.shape
    a = L.ConvArgs()
    a.c1, a.c2 = c1, 0 if x2 is None else x2.shape[-1]
    a.n, a.hin, a.win, a.cout, a.ksize = n, hin, win, w.shape[0], w.shape[1]
    a.up, a.dtype = int(up), dt_code(x1.dtype)
    return a


def conv_kernel(x1, w, x2=None, up=False, residual=None, gn_coef=None):
    """Which kernel ops.conv2d launches for these arguments: 0 = 4-wave implicit GEMM, 6 / 4 = ping-pong kernel with 192- / 128-cout
    blocks (dts_conv_kernel; measurement aid for bench.py)."""
    a = _conv_args(x1, w, x2, up)
    if isinstance(w, X3Weight):           # the launch sees one 16-bit source of 2*(c1+c2) channels
        a.c1, a.c2, a.dtype = 2 * (a.c1 + a.c2), 0, L.DTS_F16X3
    a.residual = _ptr(residual)
    a.gn_coef = _ptr(gn_coef, 'gn_coef', torch.float32)
    return int(L.load().dts_conv_kernel(C.byref(a)))


def conv_fuses_gn(x1, w, *, x2=None, up=False):
    """True if conv2d(x1, w, ..., gn_coef=...) applies the GroupNorm of its input inside the kernel for this shape / dtype
    (the ping-pong / halo kernel: 3x3, cout % 192 == 0, 16-bit, square power-of-two images >= 16, no fused upsample)."""
    return bool(L.load().dts_conv_fuses_gn(C.byref(_conv_args(x1, w, x2, up))))


def _skip_fields(a, skip, n, ho, wo, cout):
    """dts_conv_args.skip_*: skip = (SplitAct of the block input, X3Weight of the 1x1 layer, up)"""
    src, sw, sup = skip
    if not isinstance(src, SplitAct) or not isinstance(sw, X3Weight):
        raise ValueError('conv2d: skip = (SplitAct, X3Weight, up)')
    sh = (n, ho // 2, wo // 2) if sup else (n, ho, wo)
    if tuple(src.shape[:3]) != sh or tuple(sw.shape) != (cout, 1, 1, src.shape[3]):
        raise ValueError(f'conv2d: skip source {tuple(src.shape)} / weight {tuple(sw.shape)} do not match an output of {(n, ho, wo, cout)} (up={bool(sup)})')
    a.skip_c, a.skip_x, a.skip_w = src.data.shape[-1], _ptr(src.data, 'skip_x', torch.float16), _ptr(sw.packed, 'skip_w', torch.float16)
    a.skip_acc_scale, a.skip_up = sw.acc_scale, int(bool(sup))


def conv_folds_skip(x1, w, skip):
    """True if conv2d(x1, w, bias, skip=skip) accumulates the block's 1x1 skip convolution inside this 3x3 launch (dts_conv_folds_skip:
    split-precision mode, ping-pong kernel, a grid that needs no K split); skip = (SplitAct of the block input, its X3Weight, up)."""
    if not isinstance(w, X3Weight) or skip is None or not isinstance(skip[0], SplitAct):
        return False
    a = _conv_args(x1, w, None, False)
    a.c1, a.c2, a.dtype = 2 * a.c1, 0, L.DTS_F16X3
    _skip_fields(a, skip, a.n, a.hin, a.win, a.cout)
    return bool(L.load().dts_conv_folds_skip(C.byref(a)))


def conv2d(x1, w, bias=None, *, x2=None, bias_nc=None, residual=None, up=False, out_scale=1.0, out=None, gn_stats=False,
           timing_events=None, gn_coef=None, gn_silu=True, out_split2=False, skip=None):
    """skip=(SplitAct, X3Weight, up): the block's 1x1 skip convolution of that operand accumulated by this launch (only where conv_folds_skip()
    says so; `bias` is then the sum of the two layers' biases and there is no `residual`).
    timing_events=(start, stop): raw hipEvent_t handles attached to the conv kernel's own dispatch (measurement only).
    gn_stats=True: the epilogue also emits the GroupNorm moments of the output (per 64-pixel strip and channel); they
    ride on the returned tensor as `out._gn_stats` (None when the launch could not produce them) and are consumed by
    group_norm(), which then skips its own pass over the tensor."""
    n, hin, win, c1 = x1.shape
    c2 = 0 if x2 is None else x2.shape[-1]
    cout, kh, kw, cin = w.shape
    if cin != c1 + c2 or kh != kw:
        raise ValueError(f'conv2d: weight {tuple(w.shape)} does not match inputs ({c1}+{c2})')
    ho, wo = (2 * hin, 2 * win) if up else (hin, win)
    x3 = isinstance(w, X3Weight)
    if x3 and (x1.dtype != torch.float32 or gn_coef is not None):
        raise ValueError('conv2d: a split-precision weight takes float32 activations (and no fused input GroupNorm)')
    if isinstance(x1, SplitAct) and (not x3 or x2 is not None):
        raise ValueError('conv2d: a split activation feeds a split-precision weight, alone')
    out_split2 = bool(out_split2) and x3 and not gn_stats       # (split-precision mode only: the qkv projection feeding attention(x3=True))
    if out_split2:
        out = torch.empty((n, ho, wo, 2 * cout), dtype=torch.float16, device=x1.device)
    elif out is None:
        out = torch.empty((n, ho, wo, cout), dtype=x1.dtype, device=x1.device)
    a = L.ConvArgs()
    a.out_split2 = int(out_split2)
    dt_in = x1.dtype
    if x3:      # the conv reads the f16 split image (per 32 channels hi | lo) of concat(x1, x2); epilogue operands and output stay float32
        xs = x1.data if isinstance(x1, SplitAct) else split3_f16(x1, x2)
        a.x1, a.c1, a.x2, a.c2 = _ptr(xs, 'x1', torch.float16), 2 * (c1 + c2), None, 0
        a.w, a.acc_scale = _ptr(w.packed, 'w', torch.float16), w.acc_scale
    else:
        a.x1, a.c1 = _ptr(x1, 'x1'), c1
        a.x2, a.c2 = _ptr(x2, 'x2', x1.dtype), c2
        a.w = _ptr(w, 'w', x1.dtype)
    a.bias = _ptr(bias, 'bias', torch.float32)
    a.bias_nc, a.ld_bias_nc = _rows(bias_nc, 'bias_nc', x1.dtype)
    a.residual = _ptr(residual, 'residual', x1.dtype)
    if residual is not None and tuple(residual.shape) != (n, ho, wo, cout):
        raise ValueError(f'conv2d: residual shape {tuple(residual.shape)} != {(n, ho, wo, cout)}')
    a.out = _ptr(out, 'out', torch.float16 if out_split2 else x1.dtype)
    a.n, a.hin, a.win, a.cout, a.ksize = n, hin, win, cout, kh
    a.up, a.out_scale, a.dtype = int(up), float(out_scale), (L.DTS_F16X3 if x3 else dt_code(x1.dtype))
    ws = _conv_workspace(x1.device)
    a.workspace, a.workspace_bytes = ws.data_ptr(), ws.numel()
    st = None
    if gn_stats and (ho * wo) % 64 == 0:
        st = torch.empty(((n * ho * wo) // 64, cout, 2), dtype=torch.float32, device=x1.device)
        a.stats_out = st.data_ptr()
    if timing_events is not None:
        a.ev_start, a.ev_stop = timing_events
    if skip is not None:
        if not x3 or residual is not None or up:
            raise ValueError('conv2d: skip= is the split-precision mode\'s, without residual / upsample')
        _skip_fields(a, skip, n, ho, wo, cout)
    if gn_coef is not None:       # GroupNorm (+SiLU) of the input applied on the staged tile: only where conv_fuses_gn() says so
        if tuple(gn_coef.shape) != (n, c1 + c2, 2):
            raise ValueError(f'conv2d: gn_coef shape {tuple(gn_coef.shape)} != {(n, c1 + c2, 2)}')
        a.gn_coef, a.gn_silu = _ptr(gn_coef, 'gn_coef', torch.float32), int(gn_silu)
    _call('dts_conv2d', C.byref(a))
    if out_split2:
        return SplitQKV(out, (n, ho, wo, cout))
    out._gn_stats = st if (st is not None and a.stats_written) else None
    return out


def split3_f16(x1, x2=None):
    """float32 NHWC [n,h,w,c1] (+ [n,h,w,c2]) -> float16 [n,h,w,2*(c1+c2)]: per 32 channels of the channel concat hi(32) | lo * 2^11 (32)
    (dts_split3_f16; the historical name: round 4's image had three planes)."""
    n, h, w, c1 = x1.shape
    c2 = 0 if x2 is None else x2.shape[-1]
    out = torch.empty((n, h, w, 2 * (c1 + c2)), dtype=torch.float16, device=x1.device)
    _call('dts_split3_f16', _ptr(x1, 'x1', torch.float32), c1, _ptr(x2, 'x2', torch.float32), c2, _ptr(out), n * h * w)
    return out


def conv_in3(x, w, bias, cout, dtype):
    """x f32 NCHW [n,3,h,w]; w f32 OIHW [cout,3,3,3] -> NHWC [n,h,w,cout]."""
    n, c, h, wd = x.shape
    assert c == 3
    out = torch.empty((n, h, wd, cout), dtype=dtype, device=x.device)
    _call('dts_conv_in3', _ptr(x, 'x', torch.float32), _ptr(w, 'w', torch.float32), _ptr(bias, 'bias', torch.float32),
          _ptr(out), dt_code(dtype), n, h, wd, cout)
    return out


def conv_out3(x, w_ohwi, bias):
    """x NHWC [n,h,w,c]; w f32 [3,3,3,c] (O,kh,kw,I) -> f32 NCHW [n,3,h,w]."""
    n, h, wd, c = x.shape
    out = torch.empty((n, 3, h, wd), dtype=torch.float32, device=x.device)
    _call('dts_conv_out3', _ptr(x), dt_code(x.dtype), _ptr(w_ohwi, 'w', torch.float32), _ptr(bias, 'bias', torch.float32),
          _ptr(out), n, h, wd, c)
    return out


# ---- group norm ---------------------------------------------------------------------------------
def gn_coef(x1, groups, eps, gamma, beta, *, x2=None, scale_shift=None):
    n, h, w, c1 = x1.shape
    c2 = 0 if x2 is None else x2.shape[-1]
    C_ = c1 + c2
    lib = L.load()
    ws = torch.empty((lib.dts_gn_ws_floats(n, groups),), dtype=torch.float32, device=x1.device)
    coef = torch.empty((n, C_, 2), dtype=torch.float32, device=x1.device)
    ss_ptr, ss_ld = _rows(scale_shift, 'scale_shift', x1.dtype)
    _call('dts_gn_coef', _ptr(x1, 'x1'), c1, _ptr(x2, 'x2', x1.dtype), c2, dt_code(x1.dtype), n, h * w, groups, float(eps),
          _ptr(gamma, 'gamma', torch.float32), _ptr(beta, 'beta', torch.float32), ss_ptr, ss_ld, _ptr(coef), _ptr(ws))
    return coef


def gn_apply(x1, coef, *, x2=None, silu=True, pool=False, split_out=False, raw_split=False):
    """split_out (float32 inputs only): the result leaves as the f16 split image a split-precision convolution reads (SplitAct).
    raw_split (with split_out): returns (normalised image, image of the un-normalised [2x2-averaged if pool] input) from the one pass."""
    n, h, w, c1 = x1.shape
    c2 = 0 if x2 is None else x2.shape[-1]
    ho, wo = (h // 2, w // 2) if pool else (h, w)
    if split_out:
        out = torch.empty((n, ho, wo, 2 * (c1 + c2)), dtype=torch.float16, device=x1.device)
        raw = torch.empty_like(out) if raw_split else None
        _call('dts_gn_apply_x3', _ptr(x1, 'x1', torch.float32), c1, _ptr(x2, 'x2', torch.float32), c2, _ptr(coef, 'coef', torch.float32),
              _ptr(out), _ptr(raw), n, h, w, int(silu), int(pool))
        return (SplitAct(out, c1 + c2), SplitAct(raw, c1 + c2)) if raw_split else SplitAct(out, c1 + c2)
    out = torch.empty((n, ho, wo, c1 + c2), dtype=x1.dtype, device=x1.device)
    _call('dts_gn_apply', _ptr(x1, 'x1'), c1, _ptr(x2, 'x2', x1.dtype), c2, dt_code(x1.dtype),
          _ptr(coef, 'coef', torch.float32), _ptr(out), n, h, w, int(silu), int(pool))
    return out


GN_FUSED_MAX_HW = 64        # 8x8 levels take the single-launch kernel (measured: tools/gn_bench.py; larger levels are bandwidth-bound)


def _coef_from_strips(x1, x2, groups, eps, gamma, beta, scale_shift):
    """coef [n, C, 2] from the strip statistics the producing convolutions attached to x1 (and x2), or None."""
    n, h, w, c1 = x1.shape
    c2 = 0 if x2 is None else x2.shape[-1]
    st1 = getattr(x1, '_gn_stats', None)
    st2 = None if x2 is None else getattr(x2, '_gn_stats', None)
    if st1 is None or (x2 is not None and st2 is None) or (h * w) % 64 != 0:
        return None
    coef = torch.empty((n, c1 + c2, 2), dtype=torch.float32, device=x1.device)
    ss_ptr, ss_ld = _rows(scale_shift, 'scale_shift', x1.dtype)
    _call('dts_gn_coef_strips', _ptr(st1, 'st1', torch.float32), c1, _ptr(st2, 'st2', torch.float32), c2, dt_code(x1.dtype), n,
          h * w, groups, float(eps), _ptr(gamma, 'gamma', torch.float32), _ptr(beta, 'beta', torch.float32), ss_ptr, ss_ld,
          _ptr(coef))
    return coef


def gn_coefficients(x1, groups, eps, gamma, beta, *, x2=None, scale_shift=None):
    """The GroupNorm of concat(x1, x2) as per-(sample, channel) coefficients (a, b): norm(x)*gamma+beta [*(1+scale)+shift] == x*a+b.
    Taken from the producers' strip statistics when they are attached, else from a statistics pass over the tensor.  What
    conv2d(..., gn_coef=) consumes when the consuming convolution applies the norm itself (conv_fuses_gn)."""
    coef = _coef_from_strips(x1, x2, groups, eps, gamma, beta, scale_shift)
    return coef if coef is not None else gn_coef(x1, groups, eps, gamma, beta, x2=x2, scale_shift=scale_shift)


def group_norm(x1, groups, eps, gamma, beta, *, x2=None, scale_shift=None, silu=True, pool=False, path=None, split_out=False, raw_split=False):
    """GroupNorm [+ (1+scale), shift] [+ SiLU] [+ 2x2 average pool].  path: None = auto, 'fused' | 'split' (tests).
    split_out: float32 in, SplitAct out (the operand form of a split-precision convolution); raw_split (with split_out): a pair, see gn_apply."""
    n, h, w, c1 = x1.shape
    c2 = 0 if x2 is None else x2.shape[-1]
    cg = (c1 + c2) // groups
    fused_ok = (not pool) and cg % 2 == 0 and cg <= 64
    if path in (None, 'strips'):
        coef = _coef_from_strips(x1, x2, groups, eps, gamma, beta, scale_shift)
        if coef is not None:
            return gn_apply(x1, coef, x2=x2, silu=silu, pool=pool, split_out=split_out, raw_split=raw_split)
    if path == 'strips':
        raise ValueError('strip statistics are not attached to the input(s)')
    if path is None:
        path = 'fused' if (fused_ok and h * w <= GN_FUSED_MAX_HW and not split_out) else 'split'
    if path == 'fused':
        if not fused_ok:
            raise ValueError('fused GroupNorm needs pool=False and an even number (<= 64) of channels per group')
        out = torch.empty((n, h, w, c1 + c2), dtype=x1.dtype, device=x1.device)
        ss_ptr, ss_ld = _rows(scale_shift, 'scale_shift', x1.dtype)
        _call('dts_gn_fused', _ptr(x1, 'x1'), c1, _ptr(x2, 'x2', x1.dtype), c2, dt_code(x1.dtype), n, h * w, groups, float(eps),
              _ptr(gamma, 'gamma', torch.float32), _ptr(beta, 'beta', torch.float32), ss_ptr, ss_ld, _ptr(out), int(silu))
        return out
    return gn_apply(x1, gn_coef(x1, groups, eps, gamma, beta, x2=x2, scale_shift=scale_shift), x2=x2, silu=silu, pool=pool,
                    split_out=split_out, raw_split=raw_split)


def resample2x(x, up):
    n, h, w, c = x.shape
    ho, wo = (2 * h, 2 * w) if up else (h // 2, w // 2)
    out = torch.empty((n, ho, wo, c), dtype=x.dtype, device=x.device)
    _call('dts_resample2x', _ptr(x), _ptr(out), dt_code(x.dtype), n, h, w, c, int(up))
    return out


# ---- attention ----------------------------------------------------------------------------------
def attention(qkv, heads, scale, x3=False, split_out=False):
    """qkv [n, t, 3*heads*d] (q|k|v blocks) -> [n, t, heads*d].  x3 (split-precision mode, float32 qkv or a SplitQKV, head dim 64): Q.K^T and
    P.V on the 16-bit matrix cores with hi/lo operand pairs (dts_attention_x3) instead of the f32 matrix instruction; same accuracy.
    split_out (with x3): the result leaves as the operand image of the proj convolution (SplitAct [n, t, 1, C])."""
    n, t, c3 = qkv.shape
    c = c3 // 3
    d = c // heads
    pre = isinstance(qkv, SplitQKV)
    # (short sequences stay on the f32 kernel: at T = 64 the split pass alone costs what that launch does -- tools/att_bench.py --x3)
    if pre or (x3 and d == 64 and t >= 128 and qkv.dtype == torch.float32):
        if pre:
            sp = qkv.data
        else:
            sp = torch.empty((n, t, 2 * c3), dtype=torch.float16, device=qkv.device)
            _call('dts_split2_f16', _ptr(qkv, 'qkv', torch.float32), c3, _ptr(sp), n * t)
        if split_out:
            out = torch.empty((n, t, 2 * c), dtype=torch.float16, device=qkv.device)
            _call('dts_attention_x3', _ptr(sp), _ptr(out), 1, n, t, heads, d, float(scale))
            return SplitAct(out.view(n, t, 1, 2 * c), c)
        out = torch.empty((n, t, c), dtype=torch.float32, device=qkv.device)
        _call('dts_attention_x3', _ptr(sp), _ptr(out), 0, n, t, heads, d, float(scale))
        return out
    out = torch.empty((n, t, c), dtype=qkv.dtype, device=qkv.device)
    _call('dts_attention', _ptr(qkv), _ptr(out), dt_code(qkv.dtype), n, t, heads, d, float(scale))
    return out


def attention_x3_ok(t, d):
    """whether attention(..., x3=True) takes the split-precision kernel for this sequence length / head dim (so that the qkv projection may
    write its operand image directly and the proj convolution may read one)"""
    return d == 64 and t >= 128 and os.environ.get('DTS_X3_FUSE_IMAGES', '1') != '0'     # (0: A/B aid -- f32 tensors + split passes)


# ---- embedding / preconditioning ----------------------------------------------------------------
def linear(x, w, bias=None, *, act_in=False, act_out=False, out=None, accumulate=False):
    m, k = x.shape
    nn_ = w.shape[0]
    if out is None:
        out = torch.empty((m, nn_), dtype=torch.float32, device=x.device)
    _call('dts_linear', _ptr(x, 'x', torch.float32), x.stride(0) if x.dim() == 2 else k, _ptr(w, 'w', torch.float32),
          _ptr(bias, 'bias', torch.float32), _ptr(out, 'out', torch.float32), out.shape[-1], m, k, nn_,
          int(act_in), int(act_out), int(accumulate))
    return out


def pos_embedding(v, freqs, swap=False):
    n, half = v.shape[0], freqs.shape[0]
    out = torch.empty((n, 2 * half), dtype=torch.float32, device=v.device)
    _call('dts_pos_embedding', _ptr(v, 'v', torch.float32), _ptr(freqs, 'freqs', torch.float32), _ptr(out), n, half, int(swap))
    return out


def edm_precond_in(x, sigma, sigma_data):
    """x f64 NCHW, sigma f64 [1] or [n] -> (c_in*x as f32 NCHW, coef f32 [n,4] = c_skip,c_out,c_in,c_noise)."""
    n = x.shape[0]
    chw = x[0].numel()
    xin = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    coef = torch.empty((n, 4), dtype=torch.float32, device=x.device)
    _call('dts_edm_precond_in', _ptr(x, 'x', torch.float64), _ptr(sigma, 'sigma', torch.float64), sigma.numel(),
          float(sigma_data), _ptr(xin), _ptr(coef), n, chw)
    return xin, coef


def edm_precond_out(x, F, coef):
    n = x.shape[0]
    D = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    _call('dts_edm_precond_out', _ptr(x, 'x', torch.float64), _ptr(F, 'F', torch.float32), _ptr(coef, 'coef', torch.float32),
          _ptr(D), n, x[0].numel())
    return D


def cast_from_f32(x, dtype):
    if dtype == torch.float32:
        return x
    out = torch.empty(x.shape, dtype=dtype, device=x.device)
    _call('dts_cast_from_f32', _ptr(x, 'x', torch.float32), _ptr(out), dt_code(dtype), x.numel())
    return out


def cast_to_f32(x):
    if x.dtype == torch.float32:
        return x
    out = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    _call('dts_cast_to_f32', _ptr(x), dt_code(x.dtype), _ptr(out), x.numel())
    return out


# ---- Heun step ----------------------------------------------------------------------------------
def heun_xhat(x_cur, eps, noise_coef, nb, interleave=False):
    """x_hat[i] = x_cur[src(i)] + noise_coef*eps[i]; x_cur has xb rows, broadcast to nb rows
    (Tensor.repeat order unless interleave=True = repeat_interleave order)."""
    xb = x_cur.shape[0]
    chw = x_cur[0].numel()
    if eps.dtype not in (torch.float64, torch.float32):
        raise ValueError('eps must be float64 or float32')
    if eps.shape[0] != nb:
        raise ValueError('eps rows != nb')
    x_hat = torch.empty((nb,) + tuple(x_cur.shape[1:]), dtype=torch.float64, device=x_cur.device)
    _call('dts_heun_xhat', _ptr(x_cur, 'x_cur', torch.float64), xb, int(interleave), _ptr(eps, 'eps'),
          int(eps.dtype == torch.float32), float(noise_coef), _ptr(x_hat), nb, chw)
    return x_hat


def heun_euler(x_hat, D, t_hat, t_next):
    d_cur = torch.empty_like(x_hat)
    x_next = torch.empty_like(x_hat)
    _call('dts_heun_euler', _ptr(x_hat, 'x_hat', torch.float64), _ptr(D, 'D', torch.float32), float(t_hat), float(t_next),
          _ptr(d_cur), _ptr(x_next), x_hat.numel())
    return d_cur, x_next


def heun_correct(x_hat, D2, d_cur, t_hat, t_next, x_next):
    _call('dts_heun_correct', _ptr(x_hat, 'x_hat', torch.float64), _ptr(D2, 'D2', torch.float32),
          _ptr(d_cur, 'd_cur', torch.float64), float(t_hat), float(t_next), _ptr(x_next, 'x_next', torch.float64), x_hat.numel())
    return x_next


# ---- scorer plumbing ----------------------------------------------------------------------------
def quantize_u8(x, f32_math=False):
    """(x*127.5+128).clip(0,255) truncated to uint8; f64 arithmetic (EDM loop) unless f32_math (SD loop, f32 input)."""
    if x.dtype not in (torch.float64, torch.float32):
        raise ValueError('quantize_u8: float64/float32 only')
    if f32_math and x.dtype != torch.float32:
        raise ValueError('f32_math needs a float32 input')
    out = torch.empty(x.shape, dtype=torch.uint8, device=x.device)
    _call('dts_quantize_u8', _ptr(x), 2 if f32_math else int(x.dtype == torch.float32), _ptr(out), x.numel())
    return out


def cfg_combine(uncond, cond, guidance):
    out = torch.empty_like(uncond)
    _call('dts_cfg_combine', _ptr(uncond, 'uncond'), _ptr(cond, 'cond', uncond.dtype), float(guidance), _ptr(out),
          dt_code(uncond.dtype), uncond.numel())
    return out


def brightness(img_u8):
    n, c, h, w = img_u8.shape
    assert c == 3
    out = torch.empty((n,), dtype=torch.float32, device=img_u8.device)
    _call('dts_brightness', _ptr(img_u8, 'img', torch.uint8), _ptr(out), n, h * w)
    return out


def u8_to_unit_f32(img_u8):
    out = torch.empty(img_u8.shape, dtype=torch.float32, device=img_u8.device)
    _call('dts_u8_to_unit_f32', _ptr(img_u8, 'img', torch.uint8), _ptr(out), img_u8.numel())
    return out


def resample_u8(img, out_len, axis, bounds, coefs):
    """One pass of Pillow's 8-bit separable resampling: img uint8 [planes, h, w] -> [planes, h, out_len] (axis 1) or [planes, out_len, w]
    (axis 0); bounds int32 [out_len, 2], coefs int32 [out_len, ksize] (clip_preprocess.resample_tables)."""
    planes, h, w = img.shape
    out = torch.empty((planes, h, out_len) if axis else (planes, out_len, w), dtype=torch.uint8, device=img.device)
    _call('dts_resample_u8', _ptr(img, 'img', torch.uint8), _ptr(out), planes, h, w, out_len, int(axis), _ptr(bounds, 'bounds', torch.int32),
          _ptr(coefs, 'coefs', torch.int32), coefs.shape[1])
    return out


def lut_u8_f32(img, lut):
    """img uint8 [n, c, h, w], lut f32 [c, 256] -> f32 [n, c, h, w]."""
    n, c, h, w = img.shape
    out = torch.empty(img.shape, dtype=torch.float32, device=img.device)
    _call('dts_lut_u8_f32', _ptr(img, 'img', torch.uint8), _ptr(lut, 'lut', torch.float32), _ptr(out), n, c, h * w)
    return out


def cosine_rows(a, b):
    """a [n,d] f32, b [n,d] or [1,d] f32 -> [n] f32: cosine similarity of the L2-normalised rows (CLIP reward tail)."""
    n, d = a.shape
    out = torch.empty((n,), dtype=torch.float32, device=a.device)
    _call('dts_cosine_rows', _ptr(a, 'a', torch.float32), _ptr(b, 'b', torch.float32), b.shape[0], _ptr(out), n, d)
    return out


def attnpool_tokens(x, pos):
    n, h, w, c = x.shape
    out = torch.empty((n, h * w + 1, c), dtype=x.dtype, device=x.device)
    _call('dts_attnpool_tokens', _ptr(x), _ptr(pos, 'pos', torch.float32), _ptr(out), dt_code(x.dtype), n, h * w, c)
    return out


def take_token(x, token):
    n, t, c = x.shape
    out = torch.empty((n, c), dtype=torch.float32, device=x.device)
    _call('dts_take_token', _ptr(x), dt_code(x.dtype), _ptr(out), n, t, c, int(token))
    return out


def softmax_gather(logits, target):
    n, k = logits.shape
    out = torch.empty((n,), dtype=torch.float32, device=logits.device)
    _call('dts_softmax_gather', _ptr(logits, 'logits', torch.float32), _ptr(target, 'target', torch.int32), _ptr(out), n, k)
    return out


def candidate_noise(pivot, g, mode, scale):
    """pivot [b,...] f64, g [N*b,...] f64 (n-major), mode int32 [N], scale f32 [N] -> candidates [N*b,...] f64."""
    b = pivot.shape[0]
    nb = g.shape[0]
    out = torch.empty_like(g)
    _call('dts_candidate_noise', _ptr(pivot, 'pivot', torch.float64), _ptr(g, 'g', torch.float64),
          _ptr(mode, 'mode', torch.int32), _ptr(scale, 'scale', torch.float32), _ptr(out), nb, b, pivot[0].numel())
    return out


def candidate_noise_sd(pivot, u, mode, scale):
    """pivot [1,C,H,W], u [N,1,C,H,W] (same dtype), mode int32 [N], scale f32 [N,3] = (rand, lambda, sqrt(numel)) per candidate ->
    candidates [N,1,C,H,W]: the SD backend's eps-greedy / zero-order builder (pipeline_stable_diffusion.py:1371-1379) in the latents' dtype,
    the three tensor-by-scalar products rounded one by one like the reference's."""
    n = u.shape[0]
    if tuple(scale.shape) != (n, 3):
        raise ValueError(f'candidate_noise_sd: scale must be [{n}, 3] (rand, lambda, sqrt(numel)), got {tuple(scale.shape)}')
    out = torch.empty_like(u)
    _call('dts_candidate_noise_sd', _ptr(pivot, 'pivot'), _ptr(u, 'u', pivot.dtype), _ptr(mode, 'mode', torch.int32),
          _ptr(scale, 'scale', torch.float32), _ptr(out), dt_code(pivot.dtype), n, pivot.numel())
    return out


def ddim_candidates(x, e, z, alpha_t, alpha_prev, sigma_t, want_x0=True):
    """x, e: [count] tensors (any shape, same dtype); z: [ncand, *x.shape] or None -> (prev [ncand,*], x0)."""
    ncand = 1 if z is None else z.shape[0]
    prev = torch.empty((ncand,) + tuple(x.shape), dtype=x.dtype, device=x.device)
    x0 = torch.empty_like(x) if want_x0 else None
    _call('dts_ddim_candidates', _ptr(x), _ptr(e, 'e', x.dtype), _ptr(z, 'z', x.dtype), _ptr(prev), _ptr(x0), dt_code(x.dtype),
          float(alpha_t), float(alpha_prev), float(sigma_t), ncand, x.numel())
    return prev, x0
