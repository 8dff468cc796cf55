"""Builds libdts_hip.so (gfx950) in-tree with hipcc.  `python -m diffusion_tts_amd.build [--force]`."""
import os
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'libdts_hip.so')
SOURCES = ['conv_igemm.hip', 'conv_small.hip', 'groupnorm.hip', 'attention.hip', 'elementwise.hip']
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wall', '-Wno-unused-function']
# per-source extras.  attention: keep the MFMA accumulators in VGPRs -- the online-softmax rescale reads and rewrites them
# every key tile, and with the default AGPR form hipcc emitted 80 v_accvgpr_read/write per tile (a quarter of the loop's VALU).
EXTRA = {'attention.hip': ['-mllvm', '-amdgpu-mfma-vgpr-form=1']}


def _newer(a, b):
    return (not os.path.exists(b)) or os.path.getmtime(a) > os.path.getmtime(b)


def build(force=False, verbose=False):
    hdrs = [os.path.join(CSRC, 'dts_common.h'), os.path.join(HERE, '..', 'include', 'dts.h')]
    objs, jobs = [], []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(CSRC, src.replace('.hip', '.o'))
        objs.append(o)
        if force or _newer(s, o) or any(_newer(h, o) for h in hdrs):
            jobs.append([HIPCC, *FLAGS, *EXTRA.get(src, []), '-c', s, '-o', o])

    def run(cmd):
        if verbose:
            print(' '.join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f'hipcc failed: {" ".join(cmd)}\n{r.stdout}\n{r.stderr}')
        return r.stderr

    with ThreadPoolExecutor(max_workers=4) as ex:
        for warn in ex.map(run, jobs):
            if verbose and warn.strip():
                print(warn)
    if jobs or force or not os.path.exists(LIB):
        check_m0(os.path.join(CSRC, 'conv_igemm.o'))
        run([HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB, *objs])
    return LIB


OBJDUMP = os.environ.get('LLVM_OBJDUMP', '/opt/rocm/lib/llvm/bin/llvm-objdump')


def check_m0(obj):
    """conv_igemm.hip writes M0 (the LDS-DMA destination base) in inline asm and declares it clobbered instead of saving and
    restoring it around every load; hipcc warns that clobbering a reserved register 'may lead to undefined behaviour'.  It is
    sound only while NO compiler-generated instruction in those kernels uses M0.  This pins that: in the gfx950 code object every
    instruction that names m0 must be one of our own `s_mov_b32 m0, s<N>` (or vcc_lo / vcc_hi as the scalar source) (the LDS-DMA loads read M0 implicitly).  Returns the
    number of such writes; raises if anything else touches M0."""
    import glob
    import tempfile
    if not os.path.exists(OBJDUMP):
        # the M0-clobber shortcut of glds16()/bdma16() is only sound while this check passes: say loudly that it did not run
        print(f'WARNING: {OBJDUMP} not found: the M0 safety check of conv_igemm.o was SKIPPED (set LLVM_OBJDUMP)', file=sys.stderr)
        return -1
    with tempfile.TemporaryDirectory() as tmp:
        local = os.path.join(tmp, 'conv_igemm.o')
        with open(obj, 'rb') as f, open(local, 'wb') as g:
            g.write(f.read())
        subprocess.run([OBJDUMP, '--offloading', local], capture_output=True, text=True, cwd=tmp)
        dev = glob.glob(local + '.*gfx950*')
        if not dev:
            raise RuntimeError('no gfx950 code object found in ' + obj)
        dis = subprocess.run([OBJDUMP, '-d', dev[0]], capture_output=True, text=True).stdout
    uses = [ln.split('//')[0].strip() for ln in dis.splitlines() if 'm0' in ln.split('//')[0]]
    # our statement is `s_mov_b32 m0, <scalar operand>`; under SGPR pressure hipcc hands it vcc_lo / vcc_hi as that operand
    ours = re.compile(r'^s_mov_b32 m0, (s\d+|vcc_lo|vcc_hi)$')
    bad = [u for u in uses if not ours.match(u)]
    if bad or not uses:
        raise RuntimeError(f'conv_igemm code object: {len(bad)} instruction(s) other than our `s_mov_b32 m0, sN` use M0 '
                           f'(first: {bad[:3]}); the M0-clobber shortcut in glds16() is no longer safe')
    return len(uses)


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True))
