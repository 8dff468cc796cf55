"""GPU parity of every C-ABI kernel against the oracle's primitives (torch CPU fp32/fp64) on seeded inputs.
f32 mode is the parity path (tight tolerances); bf16/f16 are the throughput modes (tolerance = storage
rounding of inputs/outputs, accumulation is f32 in both)."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import edm_nets as onet          # noqa: E402
from oracle import classifier as ocls        # noqa: E402
from oracle import sampler as osamp          # noqa: E402
from oracle import scorers as oscore         # noqa: E402

DEV = 'cuda'
DTYPES = [torch.float32, torch.bfloat16, torch.float16]
TOL = {torch.float32: 2e-5, torch.bfloat16: 2.5e-2, torch.float16: 4e-3}


@pytest.fixture(scope='module')
def ops():
    from diffusion_tts_amd import ops as o
    return o


def g(seed):
    return torch.Generator().manual_seed(seed)


def rel_err(got, ref):
    return float((got.double() - ref.double()).abs().max() / max(1e-6, float(ref.double().abs().max())))


def to_nhwc(ops, x_nchw, dtype):
    return ops.nchw_to_nhwc(x_nchw.to(DEV, torch.float32).contiguous(), dtype)


def from_nhwc(ops, x):
    return ops.nhwc_to_nchw(x).cpu()


def q(x, dtype):
    """round-trip through the storage dtype so the CPU reference sees the same inputs."""
    return x.to(dtype).to(torch.float32)


@pytest.mark.parametrize('dtype', DTYPES)
def test_layout_roundtrip(ops, dtype):
    x = torch.randn(2, 24, 5, 7, generator=g(0))
    y = from_nhwc(ops, to_nhwc(ops, x, dtype))
    assert torch.equal(y, q(x, dtype))


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('case', ['3x3', '1x1', 'up', 'cat', 'full_epilogue', 'ragged', 'wide', 'tile192', 'tile576', 'splitk', 'splitk_full'])
def test_conv2d(ops, dtype, case):
    gen = g(1)
    n, h, w, c1, c2, cout, k, up = 2, 8, 8, 64, 0, 64, 3, False
    if case == '1x1':
        k = 1
    if case == 'up':
        up = True
    if case == 'cat':
        c1, c2, k = 128, 64, 1
    if case == 'ragged':
        n, h, w = 3, 5, 7                 # 105 pixels: partial pixel tile
    if case == 'wide':
        n, h, w, c1, cout = 8, 32, 32, 128, 128          # 8192 pixels: exercises the XCD remap; below 2x2-tile threshold
    if case == 'tile192':
        cout = 192                         # 192x128 block tile
    if case == 'tile576':
        n, h, w, c1, cout = 1, 16, 16, 128, 576
    if case in ('splitk', 'splitk_full'):
        n, h, w, c1, cout = 1, 8, 8, 256, 128            # 1 tile, 36+ K-steps: split-K with f32 slabs + reduce kernel
    x1 = q(torch.randn(n, c1, h, w, generator=gen), dtype)
    x2 = q(torch.randn(n, c2, h, w, generator=gen), dtype) if c2 else None
    wt = q(torch.randn(cout, c1 + c2, k, k, generator=gen) / math.sqrt((c1 + c2) * k * k), dtype)
    bias = torch.randn(cout, generator=gen)
    xin = x1 if x2 is None else torch.cat([x1, x2], 1)
    ref = onet.conv2d(xin, wt, bias, up=up)
    kw = {}
    if case in ('full_epilogue', 'splitk_full'):
        bnc = q(torch.randn(n, cout, generator=gen), dtype)
        res = q(torch.randn(n, cout, h, w, generator=gen), dtype)
        ref = (ref + bnc[:, :, None, None] + res) * 0.70710678
        kw = dict(bias_nc=bnc.to(DEV, dtype), residual=to_nhwc(ops, res, dtype), out_scale=0.70710678)
    wp = ops.pack_conv_weight(wt.to(DEV), dtype)
    out = ops.conv2d(to_nhwc(ops, x1, dtype), wp, bias.to(DEV), x2=None if x2 is None else to_nhwc(ops, x2, dtype), up=up, **kw)
    got = from_nhwc(ops, out)
    assert got.shape == ref.shape
    assert rel_err(got, ref) < TOL[dtype], (case, rel_err(got, ref))


@pytest.mark.parametrize('case', ['3x3', '1x1', 'up', 'cat', 'full_epilogue', 'ragged', 'tile192', 'tile576', 'splitk_full', 'pp192', 'pp128',
                                  'pp_res', 'tiny_values', 'huge_weights'])
def test_conv2d_split_precision(ops, case):
    """dtype F16X3: float32 activations, convolution on the 16-bit matrix cores over the f16 split image (per 32 channels hi | lo * 2^11) against
    weights packed per 32 input channels as hi | lo, three MFMAs per staged K step (dts.h).  Products carry ~22 bits: the result must sit at the f32 parity kernel's distance from the f64 reference (not at
    f16's 2^-11), on every launch form: implicit GEMM (4 / 8 waves, split K + reduce), ping-pong with 192- and 128-cout blocks, concat,
    fused upsample, full epilogue (bias, per-sample bias, residual, scale), ragged tiles; operands whose lo parts are f16-subnormal."""
    gen = g(5)
    n, h, w, c1, c2, cout, k, up = 2, 8, 8, 64, 0, 64, 3, False
    xs, wsc = 1.0, 1.0
    if case == '1x1':
        k = 1
    if case == 'up':
        up = True
    if case == 'cat':
        c1, c2, k = 128, 64, 1
    if case == 'ragged':
        n, h, w = 3, 5, 7
    if case == 'tile192':
        cout = 192
    if case == 'tile576':
        n, h, w, c1, cout = 1, 16, 16, 128, 576
    if case == 'splitk_full':
        n, h, w, c1, cout = 1, 8, 8, 256, 128
    if case in ('pp192', 'pp_res'):
        n, h, w, c1, cout = 4, 32, 32, 128, 192          # 16 blocks: the launcher takes the ping-pong kernel when forced below
    if case == 'pp128':
        n, h, w, c1, cout = 4, 32, 32, 64, 128
    if case == 'tiny_values':
        xs = 2.0 ** -9                                    # |x| ~ 2e-3: every lo part is an f16 subnormal
    if case == 'huge_weights':
        wsc = 300.0                                       # max |w| ~ 100: the pack scales DOWN (k < 0)
    x1 = torch.randn(n, c1, h, w, generator=gen) * xs
    x2 = torch.randn(n, c2, h, w, generator=gen) * xs if c2 else None
    wt = torch.randn(cout, c1 + c2, k, k, generator=gen) / math.sqrt((c1 + c2) * k * k) * wsc
    bias = torch.randn(cout, generator=gen) * xs
    xin = x1 if x2 is None else torch.cat([x1, x2], 1)
    ref = F.conv2d(F.interpolate(xin.double(), scale_factor=2, mode='nearest') if up else xin.double(), wt.double(), bias.double(), padding=k // 2)
    kw, kw32 = {}, {}
    if case in ('full_epilogue', 'splitk_full', 'pp_res'):
        bnc = torch.randn(n, cout, generator=gen)
        res = torch.randn(n, cout, h, w, generator=gen)
        ref = (ref + bnc.double()[:, :, None, None] + res.double()) * 0.70710678
        kw = dict(bias_nc=bnc.to(DEV), residual=to_nhwc(ops, res, torch.float32), out_scale=0.70710678)
    from diffusion_tts_amd import _lib
    if case.startswith('pp'):
        _lib.set_tuning('conv_variant', 1)
    try:
        x1d, x2d = to_nhwc(ops, x1, torch.float32), (None if x2 is None else to_nhwc(ops, x2, torch.float32))
        w3 = ops.pack_conv_weight(wt.to(DEV), ops.F16X3)
        assert tuple(w3.shape) == (cout, k, k, c1 + c2) and w3.packed.shape[-1] == 2 * (c1 + c2) and w3.packed.dtype == torch.float16
        if case.startswith('pp'):
            assert ops.conv_kernel(x1d, w3) in (4, 6)
        out = ops.conv2d(x1d, w3, bias.to(DEV), x2=x2d, up=up, gn_stats=True, **kw)
        out32 = ops.conv2d(x1d, ops.pack_conv_weight(wt.to(DEV), torch.float32), bias.to(DEV), x2=x2d, up=up, gn_stats=True, **kw)
    finally:
        _lib.set_tuning('conv_variant', -1)
    assert out.dtype == torch.float32
    got, got32 = from_nhwc(ops, out).double(), from_nhwc(ops, out32).double()
    scale = float(ref.abs().max())
    e3, e32 = float((got - ref).abs().max()) / scale, float((got32 - ref).abs().max()) / scale
    print(f'split precision {case}: rel err {e3:.2e} (f32 kernel {e32:.2e})')
    assert e3 < 3e-6 and e3 < 16 * max(e32, 1e-7), (case, e3, e32)
    if out._gn_stats is not None and out32._gn_stats is not None:      # GroupNorm strip statistics of the f32 outputs
        assert torch.allclose(out._gn_stats, out32._gn_stats, rtol=1e-4, atol=1e-4 * max(1.0, scale) ** 2)


@pytest.mark.parametrize('case', ['192_full_grid', '384_up', '576_16x16', '128_blocks', 'forced_small', 'scales_apart'])
def test_conv2d_folds_the_skip_convolution(ops, case):
    """Split-precision mode: a UNetBlock's 1x1 skip convolution (networks.py:164,177: x = (conv1(h) + skip(orig)) * skip_scale) accumulated by
    conv1's OWN launch -- a second K loop of the ping-pong kernel over the block input's operand image (dts_conv_args.skip_*) -- against the two
    launches it replaces (1x1 conv, then the 3x3 with that output as `residual`) and against the f64 reference: both block sizes of the kernel,
    a skip operand at half the resolution (decoder up blocks), a full-chip grid and a forced small one, packed weights whose powers of two are
    ~2^9 apart (the accumulators change units between the loops).  Where the launcher would split K the query must answer no."""
    from diffusion_tts_amd import _lib
    gen = g(11)
    n, hw, c, cs, cout, up, forced, wsc = 12, 64, 192, 384, 192, False, False, 1.0
    if case == '384_up':
        n, hw, c, cs, cout, up = 24, 32, 384, 192, 384, True
    if case == '576_16x16':
        n, hw, c, cs, cout, up = 64, 16, 576, 768, 576, True
    if case == '128_blocks':
        n, hw, c, cs, cout = 24, 32, 256, 128, 256
    if case in ('forced_small', 'scales_apart'):
        n, hw, c, cs, cout, forced = 2, 32, 192, 64, 192, True
    if case == 'scales_apart':
        wsc = 2.0 ** -12 * 1.37
    hs = hw // 2 if up else hw
    h = torch.randn(n, c, hw, hw, generator=gen)
    src = torch.randn(n, cs, hs, hs, generator=gen)
    w1 = torch.randn(cout, c, 3, 3, generator=gen) / math.sqrt(c * 9)
    ws = torch.randn(cout, cs, 1, 1, generator=gen) / math.sqrt(cs) * wsc
    b1, bs = torch.randn(cout, generator=gen), torch.randn(cout, generator=gen)
    srcd = src.double()
    ref = (F.conv2d(h.double(), w1.double(), b1.double(), padding=1) +
           F.conv2d(F.interpolate(srcd, scale_factor=2, mode='nearest') if up else srcd, ws.double(), bs.double())) * 0.70710678
    hd = ops.SplitAct(ops.split3_f16(to_nhwc(ops, h, torch.float32)), c)
    sd_ = ops.SplitAct(ops.split3_f16(to_nhwc(ops, src, torch.float32)), cs)
    w1p, wsp = ops.pack_conv_weight(w1.to(DEV), ops.F16X3), ops.pack_conv_weight(ws.to(DEV), ops.F16X3)
    if case == 'scales_apart':
        assert w1p.acc_scale / wsp.acc_scale >= 2.0 ** 7
    skip = (sd_, wsp, up)
    if forced:
        assert not ops.conv_folds_skip(hd, w1p, skip), 'a grid that would split K must not fold'
        with pytest.raises(Exception, match='cannot fold'):
            ops.conv2d(hd, w1p, (b1 + bs).to(DEV), skip=skip)
        _lib.set_tuning('conv_variant', 1)
    try:
        assert ops.conv_folds_skip(hd, w1p, skip), case
        folded = ops.conv2d(hd, w1p, (b1 + bs).to(DEV), skip=skip, out_scale=0.70710678, gn_stats=True)
        again = ops.conv2d(hd, w1p, (b1 + bs).to(DEV), skip=skip, out_scale=0.70710678, gn_stats=True)
        _lib.set_tuning('conv_skip_fold', 0)
        assert not ops.conv_folds_skip(hd, w1p, skip)
        sk = ops.conv2d(sd_, wsp, bs.to(DEV), up=up)
        two = ops.conv2d(hd, w1p, b1.to(DEV), residual=sk, out_scale=0.70710678, gn_stats=True)
    finally:
        _lib.set_tuning('conv_variant', -1)
        _lib.set_tuning('conv_skip_fold', -1)
    assert torch.equal(folded, again)                          # fixed summation order
    got, got2 = from_nhwc(ops, folded).double(), from_nhwc(ops, two).double()
    scale = float(ref.abs().max())
    e1, e2 = float((got - ref).abs().max()) / scale, float((got2 - ref).abs().max()) / scale
    print(f'skip fold {case}: rel err folded {e1:.2e}, two launches {e2:.2e}')
    # (the forced small grids: the two-launch conv1 takes a K split there, whose shorter f32 sums sit ~3x closer to the f64 reference)
    assert e1 < 3e-6 and e1 < 4 * max(e2, 2e-7), (case, e1, e2)
    # GroupNorm moments of the output: per image (which 64 pixels make a strip differs between the kernels: patch rows / consecutive pixels)
    assert folded._gn_stats is not None and two._gn_stats is not None
    per_image = lambda st: st.view(n, -1, cout, 2).sum(1)
    assert torch.allclose(per_image(folded._gn_stats), per_image(two._gn_stats), rtol=1e-4, atol=1e-3 * max(1.0, scale) ** 2)


def test_split_images_saturate_instead_of_poisoning(ops):
    """an activation beyond the f16 range (|x| >= 65520; |qkv| >= 1024 for the attention image, which carries 2^6) used to become
    hi = inf, lo = x - inf = NaN in the operand image and from there NaN in every output the element touches.  The images saturate at the
    largest finite f16 value instead (a clamped operand, not a poisoned search); a NaN stays a NaN; everything in range is untouched."""
    x = torch.zeros(1, 1, 2, 64)
    x[0, 0, 0, :6] = torch.tensor([7.0e4, -3.0e5, 65504.0, 1.0e-3, float('inf'), -1.5])
    x[0, 0, 1, 0] = float('nan')
    xd = x.to(DEV)
    img = ops.SplitAct(ops.split3_f16(xd), 64)
    hi, lo = (t.float().cpu() for t in img.planes())
    assert torch.equal(hi[0, 0, 0, :3], torch.tensor([65504.0, -65504.0, 65504.0])) and torch.equal(lo[0, 0, 0, :3], torch.zeros(3))
    assert float(hi[0, 0, 0, 4]) == 65504.0 and float(lo[0, 0, 0, 4]) == 0.0
    assert abs(float(hi[0, 0, 0, 3].double() + lo[0, 0, 0, 3].double() / 2048.0) - float(x[0, 0, 0, 3])) < 4e-10 and float(hi[0, 0, 0, 5]) == -1.5
    assert torch.isnan(hi[0, 0, 1, 0]) and torch.isfinite(hi[0, 0, 0]).all() and torch.isfinite(lo[0, 0, 0]).all()
    wt = torch.randn(64, 64, 1, 1, generator=g(3)) / 8.0
    out = ops.conv2d(xd, ops.pack_conv_weight(wt.to(DEV), ops.F16X3)).cpu()
    want = F.conv2d(x[:, :, :1].clamp(-65504.0, 65504.0).permute(0, 3, 1, 2).double(), wt.double())[0, :, 0, 0]
    assert torch.isfinite(out[0, 0, 0]).all() and float((out[0, 0, 0].double() - want).abs().max()) < 1e-5 * float(want.abs().max())
    assert torch.isnan(out[0, 0, 1]).all()                          # the NaN pixel stays visible
    q_ = torch.zeros(1, 2, 192)
    q_[0, 0, :3] = torch.tensor([2000.0, -1.0e6, 3.0])
    sp = torch.empty((1, 2, 384), dtype=torch.float16, device=DEV)
    ops._call('dts_split2_f16', q_.to(DEV).data_ptr(), 192, sp.data_ptr(), 2)
    sp = sp.float().cpu()
    assert torch.equal(sp[0, 0, :3], torch.tensor([65504.0, -65504.0, 192.0])) and torch.equal(sp[0, 0, 192:195], torch.zeros(3)) and torch.isfinite(sp).all()


@pytest.mark.parametrize('variant', ['plain', 'concat_ss', 'pool', 'wide'])
def test_group_norm_split_precision_output(ops, variant):
    """group_norm(..., split_out=True) (dts_gn_apply_x3): the normalised tensor leaves as the f16 split image (per 32 channels hi | lo * 2^11)
    -- bit for bit what dts_split3_f16 makes of the f32 result of the plain pass -- and a split-precision conv reads it directly."""
    gen = g(91)
    n, h, w, c1, c2 = 3, 16, 16, 128, 0
    if variant == 'concat_ss':
        c1, c2 = 128, 64
    if variant == 'wide':
        n, h, w, c1 = 1, 8, 8, 1536                     # more than 256 chunks per row: the grid-stride kernel
    C_ = c1 + c2
    x1 = to_nhwc(ops, torch.randn(n, c1, h, w, generator=gen) * 2 + 0.5, torch.float32)
    x2 = to_nhwc(ops, torch.randn(n, c2, h, w, generator=gen), torch.float32) if c2 else None
    gamma, beta = torch.randn(C_, generator=gen).to(DEV), torch.randn(C_, generator=gen).to(DEV)
    ss = (torch.randn(n, 2 * C_, generator=gen) * 0.3).to(DEV) if variant == 'concat_ss' else None
    kw = dict(x2=x2, scale_shift=ss, silu=True, pool=variant == 'pool')
    plain = ops.group_norm(x1, 32, 1e-5, gamma, beta, **kw)
    sp = ops.group_norm(x1, 32, 1e-5, gamma, beta, split_out=True, **kw)
    assert isinstance(sp, ops.SplitAct) and tuple(sp.shape) == tuple(plain.shape) and sp.data.dtype == torch.float16
    ref3 = ops.split3_f16(plain)
    ndiff = int((sp.data != ref3).sum())
    hi16, lo16 = sp.planes()
    assert sp.data.shape[-1] == 2 * C_ and tuple(hi16.shape) == tuple(plain.shape)
    rec_a = hi16.float() + lo16.float() / 2048.0
    print(f'gn split {variant}: {ndiff} of {ref3.numel()} f16 values differ from split3(plain); max |reconstructed - plain| = {float((rec_a - plain).abs().max()):.3e}')
    if variant in ('plain', 'concat_ss'):                 # the row kernel: the same f32 values, so the same split image bit for bit
        assert ndiff == 0
    else:       # the grid-stride kernel: its two instantiations are compiled separately (FMA contraction, exp argument folding): the f32
        # values behind the two results agree to a few ulp, not bit for bit
        assert ndiff < 0.05 * ref3.numel() and float((rec_a - plain).abs().max()) <= 4e-6 * float(plain.abs().max())
    # hi + lo * 2^-11 reproduces the f32 value to ~2^-22, and a split-precision conv takes the image as is
    hi, lo = hi16.float(), lo16.float()
    # the layout: channel c's hi half at 64 * (c // 32) + c % 32 of the 2C-wide row, its lo half 32 further
    cc = torch.arange(C_, device=sp.data.device)
    assert torch.equal(sp.data[..., 64 * (cc // 32) + cc % 32], hi16) and torch.equal(sp.data[..., 64 * (cc // 32) + cc % 32 + 32], lo16)
    assert float((hi + lo / 2048.0 - plain).abs().max()) < 4e-7 * max(1.0, float(plain.abs().max()))
    wt = torch.randn(64, C_, 1, 1, generator=gen) / math.sqrt(C_)
    w3 = ops.pack_conv_weight(wt.to(DEV), ops.F16X3)
    assert float((ops.conv2d(sp, w3) - ops.conv2d(plain, w3)).abs().max()) < 1e-5
    # raw_split: the same pass also writes the operand image of the UN-normalised input (2x2-averaged in the pooled form, with
    # dts_resample2x's arithmetic) -- what the block's 1x1 skip convolution reads: bit for bit dts_split3_f16 of the (resampled) input
    sp2, raw = ops.group_norm(x1, 32, 1e-5, gamma, beta, split_out=True, raw_split=True, **kw)
    assert torch.equal(sp2.data, sp.data) and isinstance(raw, ops.SplitAct) and tuple(raw.shape) == tuple(plain.shape)
    if variant == 'pool':
        want = ops.split3_f16(ops.resample2x(x1, up=False))
    else:
        want = ops.split3_f16(x1, x2)
    assert torch.equal(raw.data, want), int((raw.data != want).sum())


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_conv2d_big_tiles(ops, dtype):
    """cout % 128 == 0 and >= 32768 pixels selects the 128x128 block tile."""
    gen = g(11)
    n, h, w, c, cout = 8, 64, 64, 64, 128
    x = q(torch.randn(n, c, h, w, generator=gen), dtype)
    wt = q(torch.randn(cout, c, 3, 3, generator=gen) / math.sqrt(c * 9), dtype)
    ref = onet.conv2d(x, wt, None)
    out = ops.conv2d(to_nhwc(ops, x, dtype), ops.pack_conv_weight(wt.to(DEV), dtype), None)
    assert rel_err(from_nhwc(ops, out), ref) < TOL[dtype]


@pytest.mark.parametrize('dtype', DTYPES)
def test_conv_epilogue_emits_group_norm_statistics(ops, dtype):
    """the GroupNorm moments fused into the conv epilogue give the same normalisation as the standalone pass,
    for single and concatenated sources; a split-K launch emits them from its reduce pass."""
    gen = g(21)
    n, h, w, c, cout = 2, 16, 16, 64, 128
    x = q(torch.randn(n, c, h, w, generator=gen), dtype)
    wt = q(torch.randn(cout, c, 1, 1, generator=gen) / 8, dtype)        # 1x1: few K-steps, so no split-K in any dtype
    wt2 = q(torch.randn(64, c, 1, 1, generator=gen) / 8, dtype)
    bias = torch.randn(cout, generator=gen)
    xd = to_nhwc(ops, x, dtype)
    y1 = ops.conv2d(xd, ops.pack_conv_weight(wt.to(DEV), dtype), bias.to(DEV), gn_stats=True)
    y2 = ops.conv2d(xd, ops.pack_conv_weight(wt2.to(DEV), dtype), None, gn_stats=True)
    assert y1._gn_stats is not None and y2._gn_stats is not None
    ctot = cout + 64
    gamma, beta = torch.randn(ctot, generator=gen), torch.randn(ctot, generator=gen)
    a = ops.group_norm(y1, 32, 1e-5, gamma.to(DEV), beta.to(DEV), x2=y2, path='strips')
    y1._gn_stats = y2._gn_stats = None
    b = ops.group_norm(y1, 32, 1e-5, gamma.to(DEV), beta.to(DEV), x2=y2, path='split')
    assert rel_err(a.float().cpu(), b.float().cpu()) < (1e-5 if dtype == torch.float32 else 2e-2)
    ref = onet.silu(onet.group_norm(torch.cat([from_nhwc(ops, y1), from_nhwc(ops, y2)], 1), gamma, beta, 1e-5))
    assert rel_err(from_nhwc(ops, a), ref) < (2e-5 if dtype == torch.float32 else TOL[dtype])
    # split-K launch (few tiles, long K): the statistics come from the reduce pass (residual + bias + scale included)
    xs = q(torch.randn(2, 256, 8, 8, generator=gen), dtype)
    ws = q(torch.randn(128, 256, 3, 3, generator=gen) / 48, dtype)
    rs = q(torch.randn(2, 128, 8, 8, generator=gen), dtype)
    bs = torch.randn(128, generator=gen)
    ys = ops.conv2d(to_nhwc(ops, xs, dtype), ops.pack_conv_weight(ws.to(DEV), dtype), bs.to(DEV), residual=to_nhwc(ops, rs, dtype),
                    out_scale=0.7, gn_stats=True)
    assert ys._gn_stats is not None and tuple(ys._gn_stats.shape) == (2, 128, 2)
    yf = ys.float().reshape(2, 64, 128)
    want = torch.stack([yf.sum(1), (yf * yf).sum(1)], dim=-1)              # per (sample = strip, channel): sum, sum of squares
    assert rel_err(ys._gn_stats.cpu(), want.cpu()) < 1e-5
    g2, b2 = torch.randn(128, generator=gen), torch.randn(128, generator=gen)
    a2 = ops.group_norm(ys, 32, 1e-5, g2.to(DEV), b2.to(DEV), path='strips')
    ys._gn_stats = None
    b2_ = ops.group_norm(ys, 32, 1e-5, g2.to(DEV), b2.to(DEV), path='split')
    assert rel_err(a2.float().cpu(), b2_.float().cpu()) < (1e-5 if dtype == torch.float32 else 2e-2)


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('res,c1,c2,cout', [(16, 128, 0, 192), (32, 64, 128, 384), (64, 192, 0, 192)])
def test_conv_applies_group_norm_of_its_input(ops, dtype, res, c1, c2, cout):
    """GroupNorm (+ adaptive scale/shift) + SiLU of the input applied inside the 3x3 conv on its staged halo tile == the separate
    apply pass followed by the conv, BIT FOR BIT (same arithmetic, same rounding); image borders stay zero-padded; concat input."""
    gen = g(31)
    n = 2 if res < 64 else 1
    C_ = c1 + c2
    x1 = q(torch.randn(n, c1, res, res, generator=gen) * 1.5 + 0.3, dtype)
    x2 = q(torch.randn(n, c2, res, res, generator=gen), dtype) if c2 else None
    wt = q(torch.randn(cout, C_, 3, 3, generator=gen) / math.sqrt(C_ * 9), dtype)
    bias = torch.randn(cout, generator=gen)
    gamma, beta = torch.randn(C_, generator=gen), torch.randn(C_, generator=gen)
    ss = q(torch.randn(n, 2 * C_, generator=gen) * 0.3, dtype).to(DEV, dtype).contiguous()
    x1d, x2d = to_nhwc(ops, x1, dtype), (to_nhwc(ops, x2, dtype) if c2 else None)
    wp = ops.pack_conv_weight(wt.to(DEV), dtype)
    from diffusion_tts_amd import _lib
    _lib.set_tuning('conv_variant', 1)            # the small test grids are below the launcher's block-count threshold for this kernel
    try:
        _check_fused_gn(ops, dtype, n, C_, x1, x2, x1d, x2d, wt, wp, bias, gamma, beta, ss, gen)
    finally:
        _lib.set_tuning('conv_variant', -1)


def _check_fused_gn(ops, dtype, n, C_, x1, x2, x1d, x2d, wt, wp, bias, gamma, beta, ss, gen):
    c2 = 0 if x2 is None else x2.shape[1]
    assert ops.conv_fuses_gn(x1d, wp, x2=x2d)
    coef = ops.gn_coefficients(x1d, 32, 1e-5, gamma.to(DEV), beta.to(DEV), x2=x2d, scale_shift=ss)
    fused = ops.conv2d(x1d, wp, bias.to(DEV), x2=x2d, gn_coef=coef, gn_silu=True, gn_stats=True)
    h = ops.gn_apply(x1d, coef, x2=x2d, silu=True)
    plain = ops.conv2d(h, wp, bias.to(DEV), gn_stats=True)
    assert torch.equal(fused, plain)
    assert torch.allclose(fused._gn_stats, plain._gn_stats)
    # and against the oracle's group norm + conv
    xin = torch.cat([x1, x2], 1) if c2 else x1
    hn = onet.group_norm(xin, gamma, beta, 1e-5)
    hn = onet.silu(hn * (1 + ss.float().cpu()[:, :C_, None, None]) + ss.float().cpu()[:, C_:, None, None])
    ref = onet.conv2d(hn, wt, bias)
    assert rel_err(from_nhwc(ops, fused), ref) < TOL[dtype]
    # shapes the kernel does not take are refused, not silently mis-computed
    small = to_nhwc(ops, q(torch.randn(1, 64, 8, 8, generator=gen), dtype), dtype)
    w8 = ops.pack_conv_weight(q(torch.randn(192, 64, 3, 3, generator=gen), dtype).to(DEV), dtype)
    assert not ops.conv_fuses_gn(small, w8)
    with pytest.raises(RuntimeError):
        ops.conv2d(small, w8, None, gn_coef=torch.zeros(1, 64, 2, device=DEV))


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('n,res,c1,c2,cout,ks,extras', [
    (2, 16, 192, 0, 192, 3, 'res+stats'),          # linear 256-pixel tiles (W = 16), 192-cout blocks
    (1, 32, 128, 64, 384, 3, 'bnc+stats'),         # 16x16 patches, concat input, two cout tiles
    (2, 32, 64, 0, 128, 3, 'res+stats'),           # 128-cout blocks (classifier / VAE widths)
    (1, 64, 128, 0, 256, 3, 'stats'),              # 128-cout blocks, two cout tiles, 64x64 image
    (2, 16, 256, 256, 512, 3, 'res'),              # 128-cout blocks x 4, concat, linear tiles
    (1, 16, 768, 0, 128, 3, 'res+stats'),          # one tile per cout block, long K: split-K (whole chunks per split) + reduce pass
    (2, 16, 192, 0, 384, 3, 'up+stats'),           # nearest-2x upsample fused into the halo gather: 16x16 input -> 32x32 patches
    (3, 8, 256, 0, 256, 3, 'up+stats'),            # 8x8 input -> 16x16 output (linear tiles), 128-cout blocks
])
def test_ping_pong_conv_kernel_on_every_kind_of_shape(ops, dtype, n, res, c1, c2, cout, ks, extras):
    """conv_pp_kernel forced (DTS_CONV_VARIANT=1; the test grids are below the launcher's block-count threshold) against the f32 parity
    kernel on the same 16-bit-rounded inputs: the only difference allowed is the output rounding (<= 1 ulp of the 16-bit type at
    the tensor's scale) -- residual, per-sample bias, scale, strip statistics, concat source, 192- and 128-cout blocks, split-K."""
    from diffusion_tts_amd import _lib
    gen = g(51)
    C_ = c1 + c2
    x1 = q(torch.randn(n, c1, res, res, generator=gen), dtype)
    x2 = q(torch.randn(n, c2, res, res, generator=gen), dtype) if c2 else None
    wt = q(torch.randn(cout, C_, ks, ks, generator=gen) / math.sqrt(C_ * ks * ks), dtype)
    bias = torch.randn(cout, generator=gen).to(DEV)
    resid = q(torch.randn(n, cout, res, res, generator=gen), dtype) if 'res' in extras else None
    bnc = q(torch.randn(n, cout, generator=gen), dtype) if 'bnc' in extras else None
    up = 'up' in extras
    ro = 2 * res if up else res
    kw = dict(out_scale=0.8, gn_stats='stats' in extras, up=up)

    def run(dt):
        a = dict(kw)
        if x2 is not None:
            a['x2'] = to_nhwc(ops, x2, dt)
        if resid is not None:
            a['residual'] = to_nhwc(ops, resid, dt)
        if bnc is not None:
            a['bias_nc'] = bnc.to(DEV, dt).contiguous()
        return ops.conv2d(to_nhwc(ops, x1, dt), ops.pack_conv_weight(wt.to(DEV), dt), bias, **a)

    x1d, wp = to_nhwc(ops, x1, dtype), ops.pack_conv_weight(wt.to(DEV), dtype)
    _lib.set_tuning('conv_variant', 1)
    try:
        assert ops.conv_kernel(x1d, wp, x2=None if x2 is None else to_nhwc(ops, x2, dtype), up=up,
                               residual=None if resid is None else to_nhwc(ops, resid, dtype)) == (6 if cout % 192 == 0 else 4)
        w1 = ops.pack_conv_weight(q(torch.randn(cout, C_, 1, 1, generator=gen), dtype).to(DEV), dtype)
        assert ops.conv_kernel(x1d, w1, x2=None if x2 is None else to_nhwc(ops, x2, dtype)) == 0      # 1x1 layers never take it
        got = run(dtype)
        again = run(dtype)
    finally:
        _lib.set_tuning('conv_variant', -1)
    ref = run(torch.float32)
    assert torch.equal(got, again)
    scale = float(ref.abs().max())
    ulp = scale * (2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11)
    assert float((got.float() - ref).abs().max()) <= 1.01 * ulp
    if 'stats' in extras:
        assert got._gn_stats is not None
        yf = got.float().reshape(n, ro * ro, cout)
        want = torch.stack([yf.sum(1), (yf * yf).sum(1)], dim=-1)          # per sample (the strips of a sample summed)
        assert rel_err(got._gn_stats.reshape(n, -1, cout, 2).sum(1).cpu(), want.cpu()) < 1e-5


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('variant', ['ping_pong', 'ring3', 'ring4'])
def test_conv_kernels_with_hand_counted_waits_repeat_bit_for_bit(ops, dtype, variant):
    """Race screen inside the suite (was tools/pp_stress.py): the kernels whose LDS-DMA loads are awaited by hand-counted `vmcnt` --
    conv_pp_kernel (forced on every eligible shape) and conv_igemm_kernel with its 3-deep ring -- launched 12 times per shape with the
    caches / clocks disturbed in between: every launch reproduces the first one bit for bit (outputs AND strip statistics) and sits
    within one output ulp of the f32 parity kernel.  Shapes: the classes of test_ping_pong_conv_kernel_on_every_kind_of_shape at the
    sizes of a sharded search (residual, concat, split-K, 192- and 128-cout blocks, 8x8 level, 1x1)."""
    from diffusion_tts_amd import _lib
    torch.manual_seed(1)
    shapes = [  # n, res, c1, c2, cout, ksize, residual
        (8, 64, 192, 0, 192, 3, True), (8, 32, 384, 384, 384, 3, False), (16, 16, 576, 0, 576, 3, True), (8, 16, 576, 768, 576, 3, False),
        (8, 32, 256, 0, 256, 3, False), (8, 64, 128, 0, 128, 3, False), (8, 16, 1152, 0, 576, 3, False), (16, 32, 192, 0, 384, 3, False),
    ]
    if variant in ('ring3', 'ring4'):   # the 4-wave kernel's small-grid launches (ring4: the 4-deep ring of the DTS_CONV_STAGES A/B knob, 4 waves forced): 8x8 level (split-K), 1x1 layers, 16x16 level
        shapes = [(8, 8, 768, 0, 768, 3, True), (8, 8, 768, 768, 768, 3, False), (8, 8, 768, 0, 2304, 1, False), (8, 16, 576, 0, 576, 1, True),
                  (8, 16, 576, 0, 1728, 1, False), (16, 8, 512, 0, 512, 3, True), (8, 16, 576, 0, 576, 3, True), (2, 32, 384, 0, 1152, 1, False)]
    knob, val = ('conv_variant', 1) if variant == 'ping_pong' else ('conv_stages', 3 if variant == 'ring3' else 4)
    trash = torch.empty(32 << 20, device=DEV, dtype=torch.float32)
    for (n, res, c1, c2, cout, ks, use_res) in shapes:
        x1 = torch.randn(n, res, res, c1, device=DEV).to(dtype)
        x2 = torch.randn(n, res, res, c2, device=DEV).to(dtype) if c2 else None
        C_ = c1 + c2
        w = (torch.randn(cout, ks, ks, C_, device=DEV) / math.sqrt(ks * ks * C_)).to(dtype)
        b = torch.randn(cout, device=DEV)
        r = torch.randn(n, res, res, cout, device=DEV).to(dtype) if use_res else None
        _lib.set_tuning(knob, val)
        if variant == 'ring4':
            _lib.set_tuning('conv_waves', 4)          # the 8-wave form has no 4-deep ring
        try:
            if variant == 'ping_pong':
                assert ops.conv_kernel(x1, w, x2=x2, residual=r) in (4, 6)
            first = first_st = None
            for rep in range(12):
                if rep % 3 == 1:
                    trash.normal_()                       # evict L2 / MALL, change what the memory system is doing
                if rep % 3 == 2:
                    torch.cuda.synchronize()
                    torch.cuda._sleep(2_000_000)
                o = ops.conv2d(x1, w, b, x2=x2, residual=r, out_scale=0.9, gn_stats=True)
                st = o._gn_stats
                if first is None:
                    first, first_st = o.clone(), (None if st is None else st.clone())
                else:
                    assert torch.equal(o, first), (variant, n, res, c1, c2, cout, ks, rep)
                    assert st is None or torch.equal(st, first_st), (variant, n, res, c1, c2, cout, ks, rep)
        finally:
            _lib.set_tuning(knob, -1)
            if variant == 'ring4':
                _lib.set_tuning('conv_waves', -1)
        ref = ops.conv2d(x1.float(), w.float(), b, x2=None if x2 is None else x2.float(), residual=None if r is None else r.float(),
                         out_scale=0.9)
        ulp = float(ref.abs().max()) * (2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11)
        assert float((first.float() - ref).abs().max()) <= 1.01 * ulp, (variant, n, res, c1, c2, cout, ks)


def test_conv2d_identical_rows_are_bit_identical(ops):
    """ties must stay ties: the same candidate at different batch positions gives the same bits."""
    gen = g(2)
    x = torch.randn(1, 64, 8, 8, generator=gen).repeat(5, 1, 1, 1)
    wt = torch.randn(64, 64, 3, 3, generator=gen) / 24
    for dtype in (torch.float32, torch.bfloat16):
        out = ops.conv2d(to_nhwc(ops, x, dtype), ops.pack_conv_weight(wt.to(DEV), dtype), None)
        for i in range(1, 5):
            assert torch.equal(out[0], out[i])


@pytest.mark.parametrize('dtype', DTYPES)
def test_conv_in3_out3(ops, dtype):
    gen = g(3)
    x = torch.randn(2, 3, 16, 16, generator=gen)
    w = torch.randn(64, 3, 3, 3, generator=gen) / 5
    b = torch.randn(64, generator=gen)
    out = ops.conv_in3(x.to(DEV), w.to(DEV), b.to(DEV), 64, dtype)
    assert rel_err(from_nhwc(ops, out), onet.conv2d(x, w, b)) < TOL[dtype]
    y = q(torch.randn(2, 64, 16, 16, generator=gen), dtype)
    w3 = torch.randn(3, 64, 3, 3, generator=gen) / 24
    b3 = torch.randn(3, generator=gen)
    got = ops.conv_out3(to_nhwc(ops, y, dtype), w3.permute(0, 2, 3, 1).contiguous().to(DEV), b3.to(DEV)).cpu()
    assert rel_err(got, onet.conv2d(y, w3, b3)) < 1e-5                 # 16 x 16: the tiled kernel (halo through LDS, scalar-cache weights)
    y2 = q(torch.randn(3, 64, 12, 20, generator=gen), dtype)             # not multiples of 16: the direct kernel
    got2 = ops.conv_out3(to_nhwc(ops, y2, dtype), w3.permute(0, 2, 3, 1).contiguous().to(DEV), b3.to(DEV)).cpu()
    assert rel_err(got2, onet.conv2d(y2, w3, b3)) < 1e-5
    y3 = q(torch.randn(2, 192, 32, 48, generator=gen), dtype)            # several patches per image, 192 channels (the ADM output conv's width)
    w4 = torch.randn(3, 192, 3, 3, generator=gen) / 40
    got3 = ops.conv_out3(to_nhwc(ops, y3, dtype), w4.permute(0, 2, 3, 1).contiguous().to(DEV), None).cpu()
    assert rel_err(got3, onet.conv2d(y3, w4, None)) < 1e-5


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('cout,res,n', [(192, 64, 3), (128, 32, 5), (64, 48, 2), (192, 24, 2)])
def test_conv_in3_matrix_core_path(ops, dtype, cout, res, n):
    """First conv at the U-Nets' shapes (networks.py:410 ADM 3->192 at 64x64, :284 DDPM++ 3->128 at 32x32): the f32-MFMA kernel keeps
    the f32 input and weights exact, so it must agree with the direct f32 kernel to the output rounding; res=24 (not a multiple
    of 16) takes the direct kernel."""
    gen = g(33)
    x = torch.randn(n, 3, res, res, generator=gen) * 3
    w = torch.randn(cout, 3, 3, 3, generator=gen) / 5
    b = torch.randn(cout, generator=gen)
    got = from_nhwc(ops, ops.conv_in3(x.to(DEV), w.to(DEV), b.to(DEV), cout, dtype))
    exact = from_nhwc(ops, ops.conv_in3(x.to(DEV), w.to(DEV), b.to(DEV), cout, torch.float32))
    assert rel_err(exact, onet.conv2d(x, w, b)) < 1e-5
    assert torch.equal(got, q(exact, dtype)) or (got - q(exact, dtype)).abs().max() <= 2 * (exact.abs().max() * 2.0 ** -8)
    assert rel_err(got, exact) < TOL[dtype]
    nb = from_nhwc(ops, ops.conv_in3(x.to(DEV), w.to(DEV), None, cout, dtype))
    assert rel_err(nb, onet.conv2d(x, w, None)) < TOL[dtype]


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('path', ['fused', 'split'])
@pytest.mark.parametrize('variant', ['plain', 'adaptive', 'pool', 'cat', 'nosilu'])
def test_group_norm(ops, dtype, variant, path):
    if variant == 'pool' and path == 'fused':
        pytest.skip('pooled GroupNorm always takes the split path')
    gen = g(4)
    n, c, h, w = 3, 192, 8, 8
    x = q(torch.randn(n, c, h, w, generator=gen) * 2 + 0.5, dtype)
    gamma, beta = torch.randn(c, generator=gen), torch.randn(c, generator=gen)
    groups = min(32, c // 4)
    ref = onet.group_norm(x, gamma, beta, 1e-5)
    ss = None
    if variant == 'adaptive':
        ssv = q(torch.randn(n, 2 * c, generator=gen) * 0.3, dtype)
        scale, shift = ssv[:, :c, None, None], ssv[:, c:, None, None]
        ref = torch.addcmul(shift, ref, scale + 1)
        ss = ssv.to(DEV, dtype)
    silu = variant != 'nosilu'
    if silu:
        ref = onet.silu(ref)
    if variant == 'pool':
        ref = onet.resample_down(ref)
    if variant == 'cat':
        x1, x2 = to_nhwc(ops, x[:, :128], dtype), to_nhwc(ops, x[:, 128:], dtype)
    else:
        x1, x2 = to_nhwc(ops, x, dtype), None
    out = ops.group_norm(x1, groups, 1e-5, gamma.to(DEV), beta.to(DEV), x2=x2, scale_shift=ss, silu=silu, pool=variant == 'pool',
                         path=path)
    got = from_nhwc(ops, out)
    assert rel_err(got, ref) < (1e-5 if dtype == torch.float32 else TOL[dtype])


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_group_norm_many_pixels_and_wide(ops, dtype):
    gen = g(41)
    for (n, c, h) in ((2, 64, 64), (1, 1536, 8)):
        x = q(torch.randn(n, c, h, h, generator=gen) + 1.0, dtype)
        gamma, beta = torch.randn(c, generator=gen), torch.randn(c, generator=gen)
        ref = onet.silu(onet.group_norm(x, gamma, beta, 1e-6))
        for path in ('split', 'fused'):
            out = ops.group_norm(to_nhwc(ops, x, dtype), min(32, c // 4), 1e-6, gamma.to(DEV), beta.to(DEV), path=path)
            assert rel_err(from_nhwc(ops, out), ref) < (2e-5 if dtype == torch.float32 else TOL[dtype]), path


@pytest.mark.parametrize('dtype', DTYPES)
def test_resample(ops, dtype):
    x = q(torch.randn(2, 64, 8, 8, generator=g(5)), dtype)
    xd = to_nhwc(ops, x, dtype)
    assert rel_err(from_nhwc(ops, ops.resample2x(xd, up=False)), onet.resample_down(x)) < TOL[dtype]
    assert torch.equal(from_nhwc(ops, ops.resample2x(xd, up=True)), onet.resample_up(x))


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('shape', [(2, 64, 2, 64), (1, 256, 3, 64), (2, 65, 2, 64), (1, 64, 1, 128), (1, 256, 1, 256), (1, 1024, 2, 64)])
def test_attention(ops, dtype, shape):
    n, t, heads, d = shape
    gen = g(6)
    c = heads * d
    qkv = q(torch.randn(n, t, 3 * c, generator=gen), dtype)
    qh = qkv[..., :c].reshape(n, t, heads, d).permute(0, 2, 3, 1).reshape(n * heads, d, t)
    kh = qkv[..., c:2 * c].reshape(n, t, heads, d).permute(0, 2, 3, 1).reshape(n * heads, d, t)
    vh = qkv[..., 2 * c:].reshape(n, t, heads, d).permute(0, 2, 3, 1).reshape(n * heads, d, t)
    w = onet.attention_weights(qh, kh)                       # [nh, tq, tk]
    a = torch.einsum('nqk,nck->ncq', w, vh)                  # [nh, d, t]
    ref = a.reshape(n, heads, d, t).permute(0, 3, 1, 2).reshape(n, t, c)
    out = ops.attention(qkv.to(DEV, dtype), heads, 1.0 / math.sqrt(d)).float().cpu()
    assert rel_err(out, ref) < (3e-5 if dtype == torch.float32 else TOL[dtype])


@pytest.mark.parametrize('shape,spread', [(sh, sp) for sp in (1.0, 4.0, 0.01) for sh in [(2, 64, 2), (1, 256, 3), (2, 65, 2), (3, 1, 1), (1, 1024, 2), (2, 130, 12)]] +
                         [((16, 1024, 4), 1.0), ((32, 300, 6), 1.0)])      # the last two: grids large enough for two query tiles per wave (full and ragged)
def test_attention_split_precision(ops, shape, spread):
    """ops.attention(..., x3=True) (dts_split2_f16 + dts_attention_x3, head dim 64): Q.K^T and P.V on the 16-bit matrix cores with hi/lo
    operand pairs.  Against an f64 reference it must be as close as the f32 kernel (not f16's 2^-11): full, ragged and one-token
    sequences; sharp softmaxes (spread 4: logits of +-30, most probabilities below the f16 normal range) and near-uniform ones with tiny
    operands (spread 0.01: every lo part is f16-subnormal before its 2^11 scaling)."""
    n, t, heads = shape
    d, gen = 64, g(61)
    c = heads * d
    qkv = torch.randn(n, t, 3 * c, generator=gen) * spread
    q64, k64, v64 = (qkv[..., i * c:(i + 1) * c].double().reshape(n, t, heads, d).permute(0, 2, 1, 3) for i in range(3))
    w = torch.softmax(q64 @ k64.transpose(-1, -2) / math.sqrt(d), dim=-1)
    ref = (w @ v64).permute(0, 2, 1, 3).reshape(n, t, c)
    x = qkv.to(DEV)
    if t >= 128:
        got3 = ops.attention(x, heads, 1.0 / math.sqrt(d), x3=True).double().cpu()
    else:       # ops.attention keeps short sequences on the f32 kernel: call the split path itself
        from diffusion_tts_amd import _lib
        sp = torch.empty((n, t, 6 * c), dtype=torch.float16, device=DEV)
        out = torch.empty((n, t, c), dtype=torch.float32, device=DEV)
        st = torch.cuda.current_stream().cuda_stream
        _lib.check(_lib.load().dts_split2_f16(x.data_ptr(), 3 * c, sp.data_ptr(), n * t, st), 'dts_split2_f16')
        _lib.check(_lib.load().dts_attention_x3(sp.data_ptr(), out.data_ptr(), 0, n, t, heads, d, 1.0 / math.sqrt(d), st), 'dts_attention_x3')
        got3 = out.double().cpu()
    got32 = ops.attention(x, heads, 1.0 / math.sqrt(d)).double().cpu()
    scale = float(ref.abs().max())
    e3, e32 = float((got3 - ref).abs().max()) / scale, float((got32 - ref).abs().max()) / scale
    print(f'split-precision attention n={n} t={t} heads={heads} spread={spread}: rel err {e3:.2e} (f32 kernel {e32:.2e})')
    assert e3 < max(1e-6, 1.25 * e32), (e3, e32)      # (measured: 0.4x ... 0.7x the f32 kernel's error everywhere; sharp softmaxes amplify both)


def test_split_precision_attention_block_without_f32_tensors(ops):
    """The attention block of the split-precision mode passes operand images, not f32 tensors: the qkv projection writes the attention's
    image (conv2d(out_split2=True) == dts_split2_f16 of its f32 result, bit for bit) and the attention writes the proj convolution's
    (attention(split_out=True) == dts_split3_f16 of its f32 result, bit for bit)."""
    gen = g(73)
    n, hh, ww, c, heads = 2, 16, 16, 128, 2
    x = to_nhwc(ops, torch.randn(n, c, hh, ww, generator=gen), torch.float32)
    wq = ops.pack_conv_weight((torch.randn(3 * c, c, 1, 1, generator=gen) / math.sqrt(c)).to(DEV), ops.F16X3)
    bq = torch.randn(3 * c, generator=gen).to(DEV)
    plain = ops.conv2d(x, wq, bq)
    fused = ops.conv2d(x, wq, bq, out_split2=True)
    assert isinstance(fused, ops.SplitQKV) and tuple(fused.shape) == tuple(plain.shape)
    sp = torch.empty((n, hh * ww, 6 * c), dtype=torch.float16, device=DEV)
    ops._call('dts_split2_f16', plain.data_ptr(), 3 * c, sp.data_ptr(), n * hh * ww)
    assert torch.equal(fused.data.view(n, hh * ww, 6 * c), sp)
    a32 = ops.attention(plain.view(n, hh * ww, 3 * c), heads, 0.125, x3=True)
    a3 = ops.attention(fused.view(n, hh * ww, 3 * c), heads, 0.125, x3=True, split_out=True)
    assert isinstance(a3, ops.SplitAct) and torch.equal(a3.data.view(n, hh * ww, 2 * c), ops.split3_f16(a32.view(n, hh * ww, 1, c)).view(n, hh * ww, 2 * c))


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('t', [64, 200, 1024])
def test_attention_head_dim_512(ops, dtype, t):
    """The SD VAE's mid-block attention: one head of dim 512 (two 256-wide value slices per query block, no register prefetch),
    full and ragged sequence lengths."""
    n, heads, d = 2, 1, 512
    gen = g(16)
    qkv = q(torch.randn(n, t, 3 * d, generator=gen), dtype)
    qh = qkv[..., :d].permute(0, 2, 1)
    kh = qkv[..., d:2 * d].permute(0, 2, 1)
    vh = qkv[..., 2 * d:].permute(0, 2, 1)
    w = onet.attention_weights(qh, kh)
    ref = torch.einsum('nqk,nck->ncq', w, vh).permute(0, 2, 1)
    out = ops.attention(qkv.to(DEV, dtype), heads, 1.0 / math.sqrt(d)).float().cpu()
    assert rel_err(out, ref) < TOL[dtype]
    with pytest.raises(RuntimeError):
        ops.attention(qkv.to(DEV, torch.float32), heads, 1.0 / math.sqrt(d))        # head dim 512 is 16-bit only


def test_linear_and_pos_embedding(ops):
    gen = g(7)
    x = torch.randn(5, 100, generator=gen)
    w, b = torch.randn(37, 100, generator=gen), torch.randn(37, generator=gen)
    got = ops.linear(x.to(DEV), w.to(DEV), b.to(DEV), act_in=True, act_out=True).cpu()
    ref = onet.silu(F.linear(onet.silu(x), w, b))
    assert rel_err(got, ref) < 1e-5
    acc = ops.linear(x.to(DEV), w.to(DEV), None, out=got.to(DEV).clone(), accumulate=True).cpu()
    assert rel_err(acc, ref + F.linear(x, w)) < 1e-5
    for kk in (102, 768, 1000):                 # 102: the scalar-load form (k % 4 != 0); 768 / 1000: 16-byte loads with a ragged last trip
        x2, w2 = torch.randn(9, kk, generator=gen), torch.randn(21, kk, generator=gen) / math.sqrt(kk)
        assert rel_err(ops.linear(x2.to(DEV), w2.to(DEV), None).cpu(), F.linear(x2, w2)) < 1e-5
    v = torch.tensor([-1.3, 0.02, 1.09])
    half = 32
    freqs = (1 / 10000) ** (torch.arange(half, dtype=torch.float32) / half)
    ref = onet.positional_embedding(v, 2 * half)
    assert rel_err(ops.pos_embedding(v.to(DEV), freqs.to(DEV)).cpu(), ref) < 1e-6
    sw = ref.reshape(3, 2, -1).flip(1).reshape(3, -1)
    assert rel_err(ops.pos_embedding(v.to(DEV), freqs.to(DEV), swap=True).cpu(), sw) < 1e-6


def test_precond_and_heun(ops):
    gen = g(8)
    x = torch.randn(4, 3, 8, 8, generator=gen, dtype=torch.float64) * 10
    for sig in (torch.tensor([3.7], dtype=torch.float64), torch.tensor([80.0, 1.0, 0.3, 0.002], dtype=torch.float64)):
        xin, coef = ops.edm_precond_in(x.to(DEV), sig.to(DEV), 0.5)
        s = sig.float().reshape(-1, 1, 1, 1)
        c_in = 1 / (0.25 + s ** 2).sqrt()
        assert rel_err(xin.cpu(), c_in * x.float()) < 1e-6
        Fx = torch.randn(4, 3, 8, 8, generator=gen)
        D = ops.edm_precond_out(x.to(DEV), Fx.to(DEV), coef).cpu()
        ref = 0.25 / (s ** 2 + 0.25) * x.float() + s * 0.5 / (s ** 2 + 0.25).sqrt() * Fx
        assert rel_err(D, ref) < 1e-6
        assert rel_err(coef[:, 3].cpu(), (s.flatten().log() / 4).expand(4)) < 1e-6
    # Heun step against the oracle's step with a fixed analytic "denoiser"
    class Net:
        def round_sigma(self, s):
            return torch.as_tensor(s)

        def __call__(self, xx, sigma, labels=None):
            return (xx.float() * 0.3 + 0.1)
    ctx = osamp._Ctx(Net(), 18, 40, 0.05, 50, 1.003)
    t = osamp.sigma_schedule(Net(), 18)
    xc = torch.randn(1, 3, 8, 8, generator=gen, dtype=torch.float64) * 5
    eps = torch.randn(6, 3, 8, 8, generator=gen, dtype=torch.float64)
    for i in (0, 5, 17):
        t_cur, t_next = t[i], t[i + 1]
        ref_next, _ = ctx.heun_step(xc.repeat(6, 1, 1, 1), t_cur, t_next, i, eps, None)
        gamma = min(40 / 18, math.sqrt(2) - 1) if 0.05 <= float(t_cur) <= 50 else 0
        t_hat = float(t_cur) + gamma * float(t_cur)
        coef = math.sqrt(t_hat ** 2 - float(t_cur) ** 2) * 1.003
        x_hat = ops.heun_xhat(xc.to(DEV), eps.to(DEV), coef, 6)
        D = (x_hat.float() * 0.3 + 0.1)
        d_cur, x_next = ops.heun_euler(x_hat, D.contiguous(), t_hat, float(t_next))
        if i < 17:
            D2 = (x_next.float() * 0.3 + 0.1)
            ops.heun_correct(x_hat, D2.contiguous(), d_cur, t_hat, float(t_next), x_next)
        assert float((x_next.cpu() - ref_next).abs().max()) < 1e-12 * max(1.0, float(ref_next.abs().max()))
    # repeat_interleave order and f32 noise
    xb = torch.randn(2, 3, 4, 4, generator=gen, dtype=torch.float64)
    e32 = torch.randn(6, 3, 4, 4, generator=gen)
    got = ops.heun_xhat(xb.to(DEV), e32.to(DEV), 0.5, 6, interleave=True).cpu()
    assert torch.allclose(got, xb.repeat_interleave(3, 0) + 0.5 * e32.double(), atol=1e-15)


def test_quantize_brightness_softmax(ops):
    gen = g(9)
    x = torch.randn(4, 3, 16, 16, generator=gen, dtype=torch.float64) * 1.2
    u = ops.quantize_u8(x.to(DEV))
    assert torch.equal(u.cpu(), osamp.to_uint8(x))
    assert torch.equal(ops.quantize_u8(x.float().to(DEV)).cpu(), osamp.to_uint8(x.float().double()))
    b = ops.brightness(u).cpu()
    assert torch.allclose(b, oscore.BrightnessOracle()(osamp.to_uint8(x), None, None), atol=2e-7)
    assert torch.equal(ops.u8_to_unit_f32(u).cpu(), osamp.to_uint8(x).float() / 255.0)
    logits = torch.randn(5, 1000, generator=gen) * 3
    tgt = torch.tensor([0, 999, 5, 77, 500], dtype=torch.int32)
    ref = torch.softmax(logits, 1)[torch.arange(5), tgt.long()]
    assert torch.allclose(ops.softmax_gather(logits.to(DEV), tgt.to(DEV)).cpu(), ref, rtol=2e-6, atol=1e-9)


def test_candidate_noise(ops):
    gen = g(10)
    B, N = 2, 5
    pivot = torch.randn(B, 3, 8, 8, generator=gen, dtype=torch.float64)
    gg = torch.randn(N * B, 3, 8, 8, generator=gen, dtype=torch.float64)
    mode = torch.tensor([1, 0, 1, 1, 0], dtype=torch.int32)
    scale = (torch.ones(N) * torch.tensor([0.1, 0.0, 0.731, 0.999, 0.5]) * (0.15 * np.sqrt(3 * 64 * 64))).float()
    ref = []
    for n in range(N):
        u = gg[n * B:(n + 1) * B]
        if mode[n] == 1:
            u = u / torch.norm(u, p=2, dim=(1, 2, 3), keepdim=True)
            ref.append(pivot + scale[n].reshape(1, 1, 1, 1) * u)
        else:
            ref.append(u)
    ref = torch.cat(ref, 0)
    got = ops.candidate_noise(pivot.to(DEV), gg.to(DEV), mode.to(DEV), scale.to(DEV)).cpu()
    assert float((got - ref).abs().max()) < 1e-14


@pytest.mark.parametrize('dtype', DTYPES)
def test_attnpool_tokens_and_take(ops, dtype):
    gen = g(12)
    n, c, r = 2, 64, 4
    x = q(torch.randn(n, c, r, r, generator=gen), dtype)
    pos = torch.randn(c, r * r + 1, generator=gen)
    xf = x.reshape(n, c, -1)
    ref = torch.cat([xf.mean(-1, keepdim=True), xf], -1) + pos[None]        # [n, c, t]
    tok = ops.attnpool_tokens(to_nhwc(ops, x, dtype), pos.to(DEV))
    assert rel_err(tok.float().cpu(), ref.permute(0, 2, 1)) < TOL[dtype]
    assert torch.equal(ops.take_token(tok, 0).cpu(), tok[:, 0].float().cpu())


@pytest.mark.parametrize('dtype', [torch.float32, torch.float16])
def test_ddim_candidates(ops, dtype):
    gen = g(13)
    x = q(torch.randn(1, 4, 8, 8, generator=gen), dtype)
    e = q(torch.randn(1, 4, 8, 8, generator=gen), dtype)
    z = q(torch.randn(3, 1, 4, 8, 8, generator=gen), dtype)
    a_t, a_p = 0.3, 0.45
    var = (1 - a_p) / (1 - a_t) * (1 - a_t / a_p)
    sig = 1.0 * math.sqrt(var)
    x0 = (x - math.sqrt(1 - a_t) * e) / math.sqrt(a_t)
    ref = math.sqrt(a_p) * x0 + math.sqrt(1 - a_p - sig ** 2) * e + sig * z
    prev, x0g = ops.ddim_candidates(x.to(DEV, dtype), e.to(DEV, dtype), z.to(DEV, dtype), a_t, a_p, sig)
    assert rel_err(prev.float().cpu(), ref) < TOL[dtype] and rel_err(x0g.float().cpu(), x0) < TOL[dtype]


def test_errors_are_loud(ops):
    x = torch.zeros(1, 4, 4, 48, device=DEV)
    w = torch.zeros(64, 3, 3, 48, device=DEV)
    with pytest.raises(RuntimeError, match='channels'):
        ops.conv2d(x, w)
    with pytest.raises(RuntimeError, match='GPU tensor'):
        ops.quantize_u8(torch.zeros(4, dtype=torch.float64))
