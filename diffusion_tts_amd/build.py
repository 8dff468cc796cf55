"""Builds libdts_hip.so (gfx950) in-tree with hipcc.  `python -m diffusion_tts_amd.build [--force]`."""
import os
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
        run([HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB, *objs])
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True))
