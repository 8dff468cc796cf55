#!/bin/bash
# scratch GPU job of the moment (one gpurun call): edit, run, read gpurun_out/job_*.  Committed form = the last job run.
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; rm -rf $O/job_prof_long
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/job_prof_long -- python3 $R/bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-e2e --no-parity --no-subrecords > $O/job_prof_long.log 2>&1 || exit 1
cd $R
python3 - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/job_prof_long/*/*kernel_trace.csv')[0]
rows = [r for r in csv.DictReader(open(f)) if 'conv_pp_kernel<bf16_t, 9, 0, false, 6>' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
d = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in rows]
per = 112
print('conv_pp_kernel<bf16,9,0,false,6> launches', len(d), 'iterations', len(d) / per)
for it in range(len(d) // per):
    seg = d[it * per:(it + 1) * per]
    print(f'  iteration {it:2d}: average {sum(seg) / len(seg):7.2f} us')
print(f'all launches {sum(d) / len(d):.2f} us; last 12 iterations {sum(d[-12 * per:]) / (12 * per):.2f} us')
PY
find $O -name "*kernel_trace.csv" -size +20M -delete; find $O -name "*.db" -delete
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-e2e --no-parity --no-subrecords 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('in-process:', d['value'], r['avg_launch_us'], r['achieved'], r['traffic'], r['traffic_source'][:60])"
