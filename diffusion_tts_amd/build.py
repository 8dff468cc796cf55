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

    if jobs and os.path.exists(LIB):
        os.remove(LIB)                  # a failed compile / M0 check must not leave the previous library behind to be measured by mistake
    with ThreadPoolExecutor(max_workers=4) as ex:
        for warn in ex.map(run, jobs):
            if verbose and warn.strip():
                print(warn)
    if jobs or force or not os.path.exists(LIB):
        for o in M0_OBJECTS:
            check_m0(os.path.join(CSRC, o))
        run([HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB, *objs])
    return LIB


OBJDUMP = os.environ.get('LLVM_OBJDUMP', '/opt/rocm/lib/llvm/bin/llvm-objdump')
READELF = os.environ.get('LLVM_READELF', os.path.join(os.path.dirname(OBJDUMP), 'llvm-readelf'))
# objects whose kernels write M0 in inline asm (LDS-DMA destination) and await the loads with hand-counted s_waitcnt vmcnt(N)
M0_OBJECTS = ['conv_igemm.o', 'attention.o']


def _device_object(obj, tmp):
    import glob
    local = os.path.join(tmp, os.path.basename(obj))
    with open(obj, 'rb') as f, open(local, 'wb') as g:
        g.write(f.read())
    subprocess.run([OBJDUMP, '--offloading', local], capture_output=True, text=True, cwd=tmp)
    dev = glob.glob(local + '.*gfx950*')
    if not dev:
        raise RuntimeError('no gfx950 code object found in ' + obj)
    return dev[0]


def check_m0(obj):
    """conv_igemm.hip and attention.hip write M0 (the LDS-DMA destination base) in inline asm and declare it clobbered instead of saving
    and restoring it around every load; hipcc warns that clobbering a reserved register 'may lead to undefined behaviour'.  It is sound
    only while NO compiler-generated instruction in those kernels uses M0.  This pins that: in the gfx950 code object every instruction
    that names m0 must be one of our own `s_mov_b32 m0, s<N>` (or vcc_lo / vcc_hi as the scalar source) (the LDS-DMA loads read M0
    implicitly).  It also pins the second assumption of those kernels: the LDS-DMA loads are awaited with hand-counted `s_waitcnt vmcnt(N)`,
    and a register spill to scratch would put compiler-generated vmcnt-counted memory operations inside the counted window -- so every
    kernel of the object that issues LDS-DMA (names m0) must have no scratch (.private_segment_fixed_size == 0) and no VGPR spill.
    Returns the number of M0 writes; raises if anything else touches M0 or such a kernel spills."""
    import tempfile
    if not os.path.exists(OBJDUMP):
        # the M0-clobber shortcut of glds16()/bdma16()/att_glds16() is only sound while this check passes: say loudly that it did not run
        print(f'WARNING: {OBJDUMP} not found: the M0 safety check of {os.path.basename(obj)} was SKIPPED (set LLVM_OBJDUMP)', file=sys.stderr)
        return -1
    with tempfile.TemporaryDirectory() as tmp:
        dev = _device_object(obj, tmp)
        dis = subprocess.run([OBJDUMP, '-d', dev], capture_output=True, text=True).stdout
        notes = subprocess.run([READELF, '--notes', dev], capture_output=True, text=True).stdout if os.path.exists(READELF) else ''
    name = os.path.basename(obj)
    # our statement is `s_mov_b32 m0, <scalar operand>`; under SGPR pressure hipcc hands it vcc_lo / vcc_hi as that operand
    ours = re.compile(r'^s_mov_b32 m0, (s\d+|vcc_lo|vcc_hi)$')
    uses, bad, dma_kernels, cur = 0, [], set(), None
    for ln in dis.splitlines():
        m = re.match(r'^[0-9a-f]+ <([^>]+)>:', ln)
        if m:
            cur = m.group(1)
            continue
        code = ln.split('//')[0].strip()
        if 'm0' in code:
            uses += 1
            dma_kernels.add(cur)
            if not ours.match(code):
                bad.append(code)
    if bad or not uses:
        raise RuntimeError(f'{name} code object: {len(bad)} instruction(s) other than our `s_mov_b32 m0, sN` use M0 '
                           f'(first: {bad[:3]}); the M0-clobber shortcut of the LDS-DMA helpers is no longer safe')
    # (no exemption: since round 4 every LDS-DMA kernel of the library -- the 2-deep-ring conv_igemm_kernel forms with their uncounted
    # vmcnt(0) waits included -- compiles without scratch, and a spill in any of them should fail the build rather than cost time silently)
    if notes:
        spilled = []
        for blk in re.split(r'\n\s+- \.agpr_count', notes)[1:]:
            g = lambda k: re.search(r'\.%s:\s+(\S+)' % k, blk).group(1)
            if g('name') in dma_kernels and (int(g('private_segment_fixed_size')) != 0 or int(g('vgpr_spill_count')) != 0):
                spilled.append((g('name'), int(g('private_segment_fixed_size')), int(g('vgpr_spill_count'))))
        if spilled:
            raise RuntimeError(f'{name}: LDS-DMA kernels with scratch / VGPR spills (their hand-counted vmcnt waits would no longer hold): {spilled[:3]}')
    else:
        print(f'WARNING: {READELF} not found: the no-scratch check of the LDS-DMA kernels in {name} was SKIPPED', file=sys.stderr)
    return uses


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True))
