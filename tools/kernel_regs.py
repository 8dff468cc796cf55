#!/usr/bin/env python3
"""Register / spill / scratch report of the gfx950 kernels in a built object (reads the code object's metadata notes):
    python tools/kernel_regs.py [diffusion_tts_amd/csrc/conv_igemm.o] [name filter]"""
import glob
import os
import re
import subprocess
import sys
import tempfile

LLVM = os.environ.get('LLVM_BIN', '/opt/rocm/lib/llvm/bin')


def report(obj, flt=''):
    with tempfile.TemporaryDirectory() as tmp:
        local = os.path.join(tmp, 'k.o')
        with open(obj, 'rb') as f, open(local, 'wb') as g:
            g.write(f.read())
        subprocess.run([os.path.join(LLVM, 'llvm-objdump'), '--offloading', local], capture_output=True, cwd=tmp)
        dev = glob.glob(local + '.*gfx950*')
        txt = subprocess.run([os.path.join(LLVM, 'llvm-readelf'), '--notes', dev[0]], capture_output=True, text=True).stdout
    rows = []
    for blk in re.split(r'\n\s+- \.agpr_count', txt)[1:]:
        g = lambda k: re.search(r'\.%s:\s+(\S+)' % k, blk).group(1)
        name = g('name')
        cf = '/usr/bin/c++filt'
        if os.path.exists(cf):
            name = subprocess.run([cf, name], capture_output=True, text=True).stdout.strip() or name
        if flt in name:
            rows.append((name[:110], int(g('vgpr_count')), int(g('vgpr_spill_count')), int(g('sgpr_count')), int(g('sgpr_spill_count')),
                         int(g('private_segment_fixed_size')), int(g('group_segment_fixed_size'))))
    for r in sorted(rows):
        print(f'{r[0]:110s} vgpr {r[1]:3d} spill {r[2]:3d} | sgpr {r[3]:3d} spill {r[4]:3d} | scratch {r[5]:4d} B | static LDS {r[6]} B')


if __name__ == '__main__':
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    report(sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, 'diffusion_tts_amd', 'csrc', 'conv_igemm.o'), sys.argv[2] if len(sys.argv) > 2 else '')
