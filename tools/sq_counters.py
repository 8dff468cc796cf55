#!/usr/bin/env python3
"""Matrix-pipe utilisation per kernel from rocprofv3 --pmc passes (collected separately, kernel trace only: tools/sq_counters.sh).
Per kernel name: launches, SQ_INSTS_MFMA, SQ_INSTS_VALU, SQ_INSTS_LDS, SQ_WAVE_CYCLES, SQ_BUSY_CYCLES and the derived figure the north star
asks for -- MFMA busy = matrix-pipe cycles / wave residency cycles per SIMD: a 16x16x32 16-bit MFMA occupies its SIMD's matrix pipe for
16 cycles (8 passes x ... = 16384 FLOP at 1024 FLOP/cycle/SIMD), a 16x16x4 f32 MFMA for 32 cycles per 2048 FLOP... expressed here simply as
   mfma_busy = SQ_INSTS_MFMA x cycles_per_mfma / (SQ_WAVE_CYCLES x 4 / waves_per_simd_resident)
which needs the residency; without it the tool prints the two robust ratios: MFMA instructions per wave-cycle (x 16 = pipe cycles per wave
cycle of ONE wave; two co-resident waves per SIMD share the pipe, so the pipe's busy fraction is up to twice that) and VALU per MFMA.
    python tools/sq_counters.py <pmc dir A> <pmc dir B> [name filter ...]"""
import collections
import csv
import glob
import os
import sys


def load(directory):
    out = collections.defaultdict(lambda: collections.defaultdict(float))
    calls = collections.defaultdict(lambda: collections.defaultdict(int))
    for f in glob.glob(os.path.join(directory, '**', '*counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            out[r['Kernel_Name']][r['Counter_Name']] += float(r['Counter_Value'])
            calls[r['Kernel_Name']][r['Counter_Name']] += 1
    return out, calls


def main():
    dirs = [a for a in sys.argv[1:] if os.path.isdir(a)]
    flt = [a for a in sys.argv[1:] if not os.path.isdir(a)]
    tot, calls = collections.defaultdict(dict), {}
    for d in dirs:
        o, c = load(d)
        for k, v in o.items():
            tot[k].update(v)
            calls.setdefault(k, max(c[k].values()))
    rows = []
    for k, v in tot.items():
        if flt and not any(f in k for f in flt):
            continue
        if v.get('SQ_INSTS_MFMA', 0) <= 0 or 'SQ_WAVE_CYCLES' not in v:
            continue
        cyc = 32 if ('<float' in k and 'f16_t' not in k) or 'attention32' in k else 16
        busy1 = v['SQ_INSTS_MFMA'] * cyc / (v['SQ_WAVE_CYCLES'] * 4)          # SQ_WAVE_CYCLES counts quad-cycles of wave residency
        rows.append((v['SQ_WAVE_CYCLES'], k, calls[k], v, busy1))
    for _, k, n, v, busy1 in sorted(rows, reverse=True)[:16]:
        name = k.replace('(anonymous namespace)::', '').replace('void ', '')[:84]
        print(f'{name:84s} x{n:5d}  MFMA/launch {v["SQ_INSTS_MFMA"] / n:12.0f}  VALU/MFMA {v.get("SQ_INSTS_VALU", 0) / v["SQ_INSTS_MFMA"]:5.2f}  '
              f'LDS/MFMA {v.get("SQ_INSTS_LDS", 0) / v["SQ_INSTS_MFMA"]:5.2f}  matrix-pipe cycles per wave-cycle {busy1:5.3f}  '
              f'(x resident waves per SIMD = pipe busy)  LDS bank conflict {100 * v.get("SQ_LDS_BANK_CONFLICT", 0) / max(1.0, v.get("SQ_ACTIVE_INST_LDS", 1)):4.1f} %')


if __name__ == '__main__':
    main()
