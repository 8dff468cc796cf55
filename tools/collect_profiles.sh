#!/bin/bash
# copies the summaries of tools/final_run.sh + tools/final_profiles.sh (gpurun_out/final_*) into profiles/ under the round's names:
#   tools/collect_profiles.sh r06
# and writes the two small JSON files per dtype that bench.py quotes in its line (roofline.traffic, roofline.rocprof_avg_launch_us):
#   profiles/<rd>_hbm_traffic_pmc_<dtype>.json, profiles/<rd>_rocprof_dominant_<dtype>.json   (stamped with the commit they were collected at)
set -e
cd "$(dirname "$0")/.."
RD=${1:-r06}; O=gpurun_out; HEAD=$(git rev-parse --short HEAD)
cp "$(ls -t $O/final_prof_x3/*/*kernel_stats.csv | head -1)" profiles/${RD}_bench_kernel_stats.csv           # the headline mode (f16x3)
cp "$(ls -t $O/final_prof_bf16/*/*kernel_stats.csv | head -1)" profiles/${RD}_bf16_kernel_stats.csv
cp "$(ls -t $O/final_prof_c8/*/*kernel_stats.csv | head -1)" profiles/${RD}_c8_kernel_stats.csv
python tools/rocprof_dominant.py profiles/${RD}_bench_kernel_stats.csv f16x3 profiles/${RD}_rocprof_dominant_f16x3.json $HEAD
python tools/rocprof_dominant.py profiles/${RD}_bf16_kernel_stats.csv bf16 profiles/${RD}_rocprof_dominant_bf16.json $HEAD
grep "^{" $O/final_bench.json > profiles/${RD}_bench.json
python tools/pmc_traffic.py $O/final_pmc_fetch_f16x3 $O/final_pmc_write_f16x3 $O/final_hbm_traffic_pmc_f16x3.json $O/final_conv_sequence.json > $O/final_pmc_per_shape_f16x3.txt 2>&1
python tools/pmc_traffic.py $O/final_pmc_fetch_bf16 $O/final_pmc_write_bf16 $O/final_hbm_traffic_pmc_bf16.json $O/final_conv_sequence_bf16.json > $O/final_pmc_per_shape_bf16.txt 2>&1
for dt in f16x3 bf16; do
python - $O/final_hbm_traffic_pmc_$dt.json profiles/${RD}_hbm_traffic_pmc_$dt.json $HEAD $dt <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); d['collected_at_commit'] = sys.argv[3]; d['dtype'] = sys.argv[4]      # the tree the counters were collected from (bench.py quotes it)
sys.path.insert(0, '.'); import bench; d['csrc_sha256'] = bench.csrc_digest()      # ... and the digest of its kernel sources: bench.py flags the figure as stale when they change
d['command'] = d['command'].replace('bench.py', 'bench.py --dtype ' + sys.argv[4])
json.dump(d, open(sys.argv[2], 'w'))
PY
done
cp $O/final_conv_sequence.json profiles/${RD}_conv_sequence.json
cp $O/final_conv_sequence_bf16.json profiles/${RD}_conv_sequence_bf16.json
for dt in f16x3 bf16; do
{ echo "# per conv layer shape of one search iteration in $dt (bench.py --conv-sequence): PMC HBM bytes per launch (FETCH_SIZE x2 + WRITE_SIZE, separate rocprofv3 passes) next to the algorithmic bytes"
  echo "# (every input / weight / residual element read once, every output element written once).  x 'launches per step'; time and TFLOP/s (algorithmic) from the in-process dispatch events."
  cat $O/final_pmc_per_shape_$dt.txt; } > profiles/${RD}_conv_per_shape_pmc_$dt.txt
done
grep -v amdgpu.ids $O/final_bench.err > profiles/${RD}_bench_stderr_conv_table.txt
grep -v amdgpu.ids $O/final_bench_bf16.err > profiles/${RD}_bench_bf16_stderr_conv_table.txt
for p in mcts:mcts_s256 2rank_gloo:2rank_gloo_one_gpu bf16:bf16 rccl1:rccl_one_rank f32:f32; do
  a=${p%%:*}; b=${p##*:}; grep "^{" $O/final_bench_$a.json > profiles/${RD}_bench_$b.json
done
for n in 1 2 4 8; do [ -f $O/final_scale_n$n.json ] && grep "^{" $O/final_scale_n$n.json > profiles/${RD}_scale_n$n.json; done
{ echo "# DTS_SHARD_ALWAYS_COLLECT=1 NCCL_DEBUG=VERSION python bench.py ... (one MI355X): a ONE-rank RCCL process group whose reward all-gather, barriers and"
  echo "# max-over-ranks all-reduce are all issued on device tensors"
  grep -v amdgpu $O/final_bench_rccl1.err | tail -8; } > profiles/${RD}_bench_rccl_one_rank.log
grep -v amdgpu.ids $O/final_vae.txt > profiles/${RD}_vae_decode.txt
grep -v amdgpu.ids $O/final_att.txt > profiles/${RD}_attention.txt
{ echo "# tools/sd_bench.py on one MI355X: BASELINE config 4 with this build's parts (DESIGN.md section 5)"
  grep "^SD beam" $O/final_sd.txt; } > profiles/${RD}_sd_config4_final.txt
# rocprof average of the dominant kernel next to the in-process one, so that roofline.frac can be recomputed from profiles/ alone
python - $RD <<'PY'
import csv, glob, json, sys
rd = sys.argv[1]
d = json.loads(open(f'profiles/{rd}_bench.json').read().strip().splitlines()[-1])
r = d['roofline']
rp = json.load(open(f'profiles/{rd}_rocprof_dominant_{d["dtype"]}.json'))
rp_us = rp['avg_launch_us']
out = {'dtype': d['dtype'], 'kernel': rp['kernel'], 'rocprofv3_avg_launch_us': rp_us, 'rocprofv3_calls': rp['calls'],
       'in_process_avg_launch_us': r['avg_launch_us'], 'avg_launch_gflop': r['avg_launch_gflop'],
       'frac_from_rocprofv3': round(r['avg_launch_gflop'] / rp_us * 1e3 / r['peak'], 4), 'frac_in_process': r['frac'],
       'ratio_rocprof_over_in_process': round(rp_us / r['avg_launch_us'], 4),
       'how': 'rocprofv3 --kernel-trace --stats of `bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-e2e --no-parity --no-subrecords` (same '
              'workload, same graphs; a profiled process runs a few per cent slower: MI355X_MICROARCH.md DVFS item 2); in-process = HIP events attached to the '
              'kernel\'s own dispatch in the default bench.py run.  FLOPs are ALGORITHMIC (f16x3: x3 for the matrix work)'}
try:      # the SAME profiled process measured both ways (final_profiles.sh: final_prof_both)
    b = json.loads([l for l in open('gpurun_out/final_prof_both.json') if l.startswith('{')][-1])['roofline']
    import os
    rows2 = list(csv.DictReader(open(max(glob.glob('gpurun_out/final_prof_both/*/*kernel_stats.csv'), key=os.path.getmtime))))
    names = [i_['name'] for i_ in rp.get('instantiations', [])] or [rp['kernel']]       # the family's instantiations (with / without the folded skip conv), launch-weighted
    dom2 = [x for x in rows2 if any(n_ in x['Name'] for n_ in names)]
    us2 = sum(float(x['TotalDurationNs']) for x in dom2) / sum(int(x['Calls']) for x in dom2) / 1e3
    out['same_process'] = {'in_process_avg_launch_us': b['avg_launch_us'], 'rocprofv3_avg_launch_us': round(us2, 2),
                           'ratio': round(us2 / b['avg_launch_us'], 4), 'frac_in_process': b['frac'],
                           'note': 'one process under rocprofv3: HIP events attached to the dispatches (eager launches of the instrumented steps) vs the trace average over ALL '
                                   'dispatches of that kernel in the process (graph replays + the instrumented eager steps)'}
except Exception as e:
    out['same_process'] = {'error': str(e)}
json.dump(out, open(f'profiles/{rd}_dominant_kernel_rocprof_vs_inprocess.json', 'w'), indent=1)
print(out)
PY
tail -2 $O/final_pytest.log
python - $RD <<'PY'
import json, sys
rd = sys.argv[1]
d=json.loads(open(f'profiles/{rd}_bench.json').read().strip().splitlines()[-1])
r=d['roofline']
print('dtype', d['dtype'], 'value',d['value'],'ms',d['ms_per_step'],'dom',r['kernel'],r['achieved'],r['frac'],'matrix',r.get('matrix_work_frac'),'all',r['all_conv']['achieved'],r['all_conv']['frac'],'step frac',r['whole_step_frac'])
print('e2e',d.get('e2e_evals_per_s'),'cpu',d['cpu_baseline']['value'], 'headline parity', d['parity'].get('headline'))
print('other', d.get('other_dtype'))
for k,v in d.get('sub_records',{}).items(): print(k, v.get('value'), v.get('ms_per_step'), (v.get('roofline') or {}).get('frac'), v.get('cpu_baseline'))
PY
