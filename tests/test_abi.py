"""CPU: the C-ABI library builds, loads, and exports every symbol include/dts.h declares (no compute calls)."""
import ctypes
import os
import re

from conftest import ROOT


def declared_symbols():
    src = open(os.path.join(ROOT, 'include', 'dts.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(dts_[a-z0-9_]+)\s*\(', src)))


def test_library_exports_every_declared_symbol():
    from diffusion_tts_amd import build, _lib
    build.build()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    syms = declared_symbols()
    assert len(syms) >= 30
    for s in syms:
        assert hasattr(lib, s), f'{s} declared in include/dts.h but not exported'
    bound = set(_lib.SIGNATURES) | set(_lib.OTHER)
    assert bound == set(syms), (bound ^ set(syms))
    assert _lib.load().dts_version() == _lib.ABI_VERSION


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, 'diffusion_tts_amd')
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith('.py'):
                txt = open(os.path.join(dp, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle\b', txt, flags=re.M), f
    for f in ('main.py',):
        p = os.path.join(ROOT, f)
        if os.path.exists(p):
            assert not re.search(r'^\s*(from|import)\s+oracle\b', open(p).read(), flags=re.M)


def test_only_our_own_writes_touch_m0_in_the_conv_kernels():
    """VERDICT r1 hygiene: the conv kernels clobber the reserved M0 register in inline asm; that is sound only while hipcc itself
    never uses M0 there.  The build checks the disassembly; here it is a test."""
    import os
    from diffusion_tts_amd import build
    obj = os.path.join(build.CSRC, 'conv_igemm.o')
    if not os.path.exists(obj) or not os.path.exists(build.OBJDUMP):
        import pytest
        pytest.skip('object file / llvm-objdump not present')
    assert build.check_m0(obj) > 100
