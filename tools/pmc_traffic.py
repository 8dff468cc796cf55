#!/usr/bin/env python3
"""Aggregates two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; collected separately, with --kernel-trace only) into HBM bytes
per launch per kernel, with the corrections MI355X_MICROARCH.md prescribes: both counters are in KiB... (units stated in the
output), and gfx950 reports half the bytes of wide coalesced reads, hence fetch x2.

    python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/r02_hbm_traffic_pmc.json [conv_sequence.json]
"""
import collections
import csv
import glob
import json
import os
import sys


def newest(directory):
    """the counter file of the LATEST run under `directory` (gpurun merges every call's files into the same local folder)"""
    files = glob.glob(os.path.join(directory, '**', '*counter_collection.csv'), recursive=True)
    return [max(files, key=os.path.getmtime)] if files else []


def per_kernel(directory, counter):
    files = newest(directory)
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


def conv_dispatches(directory, counter):
    """(kernel family, value) of every conv dispatch in dispatch order."""
    out = []
    for f in newest(directory):
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] != counter:
                continue
            n = short(r['Kernel_Name'])
            if n.startswith('conv_pp_kernel') or n.startswith('conv_igemm_kernel'):
                fam = n.split('<')[0]
                targs = [t.strip() for t in n[n.index('<') + 1:n.rindex('>')].split(',')]
                if fam == 'conv_igemm_kernel':           # <T, MT, NT, WM, WN, PF, STAGES>: PF (fragment prefetch) is set for the 3x3 launches
                    fam += '/3x3' if targs[5] == 'true' else '/1x1'
                elif targs[4] == '4':                    # conv_pp_kernel<T, TAPS, DBG, GN, MT>: MT = 4 -> 128-cout blocks
                    fam += '/128'
                out.append((int(r['Dispatch_Id']), fam, float(r['Counter_Value'])))
    out.sort()
    return [(k, v) for _, k, v in out]


def per_shape(fetch_dir, write_dir, seq_file):
    """Per conv shape: PMC bytes per launch next to the algorithmic bytes.  The conv launches of a step are a fixed sequence
    (bench.py --conv-sequence wrote it from the same command), so dispatch i of the trace is entry i mod len(sequence)."""
    seq = json.load(open(seq_file))['sequence']
    fd, wd = conv_dispatches(fetch_dir, 'FETCH_SIZE'), conv_dispatches(write_dir, 'WRITE_SIZE')
    L = len(seq)
    if not fd or len(fd) % L or len(wd) % L:
        return {'error': f'{len(fd)} / {len(wd)} conv dispatches are not a multiple of the {L}-launch step sequence'}
    bad = sum(1 for i, (k, _) in enumerate(fd) if k != seq[i % L]['kernel'])
    if bad:
        return {'error': f'{bad} dispatches run a different kernel than the logged sequence says'}
    agg = collections.OrderedDict()
    for which, d in (('fetch', fd), ('write', wd)):
        for i, (_, v) in enumerate(d):
            e = seq[i % L]
            t = agg.setdefault((e['shape'], e['kernel']), {'launches_per_step': 0, 'alg_bytes': e['alg_bytes'], 'flop': e['flop'], 'us': 0.0,
                                                           'fetch': 0.0, 'write': 0.0, 'nf': 0, 'nw': 0})
            t[which] += v * 1024
            t['nf' if which == 'fetch' else 'nw'] += 1
    for i, e in enumerate(seq):
        t = agg[(e['shape'], e['kernel'])]
        t['launches_per_step'] += 1
        t['us'] += e['us']
    rows = []
    for (shape, kern), t in agg.items():
        hbm = 2 * t['fetch'] / t['nf'] + t['write'] / t['nw']
        rows.append({'shape': shape, 'kernel': kern, 'launches_per_step': t['launches_per_step'], 'avg_us': round(t['us'] / t['launches_per_step'], 1),
                     'alg_mb': round(t['alg_bytes'] / 1e6, 1), 'pmc_mb': round(hbm / 1e6, 1), 'pmc_over_alg': round(hbm / t['alg_bytes'], 2),
                     'tflops': round(t['flop'] * t['launches_per_step'] / t['us'] / 1e6, 1)})
    rows.sort(key=lambda r: -r['avg_us'] * r['launches_per_step'])
    return {'rows': rows}


def short(name):
    return name.replace('void (anonymous namespace)::', '').replace('(anonymous namespace)::', '').split('(')[0]


def main():
    fetch_dir, write_dir, out = sys.argv[1:4]
    seq_file = sys.argv[4] if len(sys.argv) > 4 else None
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
    # families as bench.py names them: the ping-pong / halo kernel, and the 4-wave kernel's 3x3 (fragment-prefetch instantiation,
    # last template argument true) and 1x1 launches
    def targ(k, i):
        return [t.strip() for t in k[k.index('<') + 1:k.rindex('>')].split(',')][i] if '<' in k else ''
    pick = {'conv_pp_kernel': lambda k: k.startswith('conv_pp_kernel') and targ(k, 4) != '4',
            'conv_pp_kernel/128': lambda k: k.startswith('conv_pp_kernel') and targ(k, 4) == '4',
            'conv_igemm_kernel/3x3': lambda k: k.startswith('conv_igemm_kernel') and targ(k, 5) == 'true',
            'conv_igemm_kernel/1x1': lambda k: k.startswith('conv_igemm_kernel') and targ(k, 5) == 'false',
            'conv_igemm_kernel': lambda k: k.startswith('conv_igemm_kernel')}
    for famname, f_ in pick.items():
        ks = [v for k, v in kernels.items() if f_(k)]
        nl = sum(v['launches'] for v in ks)
        if nl:
            fam[famname] = {'launches': nl, 'hbm_bytes_per_launch': sum(v['hbm_bytes_per_launch'] * v['launches'] for v in ks) / nl}
    doc = {'kernels_by_family': fam, 'command': 'rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python3 bench.py --steps 2 --warmup 1 '
                      '--no-cpu-baseline --no-kernel-timing --no-e2e --no-parity --no-subrecords',
           'note': 'FETCH_SIZE/WRITE_SIZE are KiB; gfx950 reports half the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM), hence '
                   'fetch_corrected_x2; Infinity-Cache hits are counted too, so this is an upper bound on HBM reads',
           'kernels': dict(list(kernels.items())[:14])}
    if seq_file:
        doc['conv_per_shape'] = per_shape(fetch_dir, write_dir, seq_file)
        for r in doc['conv_per_shape'].get('rows', [])[:60]:
            print(f"{r['kernel']:18s} x{r['launches_per_step']:3d} {r['avg_us']:7.1f} us {r['tflops']:7.1f} TF/s  alg {r['alg_mb']:7.1f} MB  pmc {r['pmc_mb']:7.1f} MB  "
                  f"x{r['pmc_over_alg']:.2f}  {r['shape']}")
        if 'error' in doc['conv_per_shape']:
            print('per-shape:', doc['conv_per_shape']['error'])
    with open(out, 'w') as f:
        json.dump(doc, f, indent=1)
    for k, v in list(kernels.items())[:8]:
        print(f'{k[:70]:70s} {v["launches"]:6d} launches  {v["hbm_bytes_per_launch"] / 1e6:9.1f} MB/launch')


if __name__ == '__main__':
    main()
