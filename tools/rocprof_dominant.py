#!/usr/bin/env python3
"""The dominant conv kernel's average launch duration out of a `rocprofv3 --kernel-trace --stats` run of bench.py, written as the small
JSON bench.py quotes beside its in-process figure (roofline.rocprof_avg_launch_us / rocprof_frac):

    python tools/rocprof_dominant.py <kernel_stats.csv> <dtype: f16x3|bf16|f16|f32> <out.json> [commit]

The kernel family is the one bench.py names `conv_pp_kernel` (the 192-cout ping-pong / halo form of the dtype's instantiation); recomputing
avg_launch_gflop / avg_launch_us / peak from this file and the bench line gives the line's rocprof_frac."""
import csv
import os
import json
import sys

# (name prefixes: the family spans the instantiations with and without the folded skip convolution, conv_pp_kernel<.., SK = false | true>)
PAT = {'bf16': 'conv_pp_kernel<bf16_t, 9, 0, false, 6, bf16_t', 'f16': 'conv_pp_kernel<f16_t, 9, 0, false, 6, f16_t',
       'f16x3': 'conv_pp_kernel<f16_t, 9, 0, false, 6, float'}


def short(name):
    import re
    return re.sub(r'\(ConvP\)$|\(AttP\)$', '', name.replace('void (anonymous namespace)::', '').replace('(anonymous namespace)::', ''))


def _digest():
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    return bench.csrc_digest()


def main():
    stats, dtype, out = sys.argv[1:4]
    commit = sys.argv[4] if len(sys.argv) > 4 else 'unknown'
    rows = list(csv.DictReader(open(stats)))
    tot = sum(float(r['TotalDurationNs']) for r in rows)
    dom = [r for r in rows if PAT[dtype] in r['Name']]
    if not dom:          # older name form without the output-type argument
        dom = [r for r in rows if PAT[dtype].rsplit(',', 1)[0] in r['Name']]
    r = dom[0]
    calls, ns = sum(int(x['Calls']) for x in dom), sum(float(x['TotalDurationNs']) for x in dom)          # launch-weighted over the family's instantiations
    doc = {'kernel_family': 'conv_pp_kernel', 'kernel': short(r['Name']), 'dtype': dtype,
           'instantiations': [{'name': short(x['Name']), 'calls': int(x['Calls']), 'avg_us': round(float(x['AverageNs']) / 1e3, 3)} for x in dom],
           'calls': calls, 'avg_launch_us': round(ns / calls / 1e3, 3), 'share_of_kernel_time': round(ns / tot, 4),
           'collected_at_commit': commit, 'csrc_sha256': _digest(),
           'command': 'rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 3 --warmup 1 --dtype %s --no-cpu-baseline --no-kernel-timing --no-e2e --no-parity --no-subrecords' % dtype,
           'top_kernels': [{'name': short(x['Name'])[:90], 'calls': int(x['Calls']),
                            'avg_us': round(float(x['AverageNs']) / 1e3, 2), 'pct': float(x['Percentage'])} for x in rows[:12]]}
    json.dump(doc, open(out, 'w'), indent=1)
    print(doc['kernel'], doc['calls'], 'calls', doc['avg_launch_us'], 'us', f"{100 * doc['share_of_kernel_time']:.1f} % of kernel time")


if __name__ == '__main__':
    main()
