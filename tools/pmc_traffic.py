#!/usr/bin/env python3
"""Aggregates two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; collected separately, with --kernel-trace only) into HBM bytes
per launch per kernel, with the corrections MI355X_MICROARCH.md prescribes: both counters are in KiB... (units stated in the
output), and gfx950 reports half the bytes of wide coalesced reads, hence fetch x2.

    python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/r01_hbm_traffic_pmc.json
"""
import collections
import csv
import glob
import json
import os
import sys


def per_kernel(directory, counter):
    files = glob.glob(os.path.join(directory, '**', '*counter_collection.csv'), recursive=True)
    if not files:
        raise SystemExit(f'no counter_collection.csv under {directory}')
    tot, cnt = collections.defaultdict(float), collections.defaultdict(int)
    for f in files:
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] != counter:
                continue
            tot[r['Kernel_Name']] += float(r['Counter_Value'])
            cnt[r['Kernel_Name']] += 1
    return tot, cnt


def short(name):
    return name.replace('void (anonymous namespace)::', '').replace('(anonymous namespace)::', '').split('(')[0]


def main():
    fetch_dir, write_dir, out = sys.argv[1:4]
    ft, fc = per_kernel(fetch_dir, 'FETCH_SIZE')
    wt, wc = per_kernel(write_dir, 'WRITE_SIZE')
    kernels = {}
    for k in sorted(ft, key=lambda k: -(ft[k] + wt.get(k, 0))):
        if fc[k] == 0 or wc.get(k, 0) == 0:
            continue
        fetch = ft[k] * 1024 / fc[k]                 # counters are in KiB
        write = wt[k] * 1024 / wc[k]
        kernels[short(k)] = {'launches': fc[k], 'fetch_size_bytes_per_launch': fetch, 'fetch_corrected_x2': 2 * fetch,
                             'write_size_bytes_per_launch': write, 'hbm_bytes_per_launch': 2 * fetch + write}
    # per conv kernel family (the name bench.py's roofline.kernel uses), launch-weighted over its template instantiations
    fam = {}
    for famname in ('conv_pp_kernel', 'conv_igemm_kernel'):
        ks = [v for k, v in kernels.items() if k.startswith(famname)]
        nl = sum(v['launches'] for v in ks)
        if nl:
            fam[famname] = {'launches': nl, 'hbm_bytes_per_launch': sum(v['hbm_bytes_per_launch'] * v['launches'] for v in ks) / nl}
    doc = {'kernels_by_family': fam, 'command': 'rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python3 bench.py --steps 2 --warmup 1 '
                      '--no-cpu-baseline --no-kernel-timing --no-e2e',
           'note': 'FETCH_SIZE/WRITE_SIZE are KiB; gfx950 reports half the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM), hence '
                   'fetch_corrected_x2; Infinity-Cache hits are counted too, so this is an upper bound on HBM reads',
           'kernels': dict(list(kernels.items())[:14])}
    with open(out, 'w') as f:
        json.dump(doc, f, indent=1)
    for k, v in list(kernels.items())[:8]:
        print(f'{k[:70]:70s} {v["launches"]:6d} launches  {v["hbm_bytes_per_launch"] / 1e6:9.1f} MB/launch')


if __name__ == '__main__':
    main()
