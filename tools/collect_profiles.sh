#!/bin/bash
# copies the summaries of tools/final_run.sh (gpurun_out/final_*) into profiles/ under the round's names:  tools/collect_profiles.sh r03
set -e
cd "$(dirname "$0")/.."
RD=${1:-r04}; O=gpurun_out
cp "$(ls -t $O/final_prof/*/*kernel_stats.csv | head -1)" profiles/${RD}_bench_kernel_stats.csv
cp "$(ls -t $O/final_prof_c8/*/*kernel_stats.csv | head -1)" profiles/${RD}_c8_kernel_stats.csv
cp "$(ls -t $O/final_prof_x3/*/*kernel_stats.csv | head -1)" profiles/${RD}_f16x3_kernel_stats.csv
grep "^{" $O/final_bench.json > profiles/${RD}_bench.json
python - $O/final_hbm_traffic_pmc.json profiles/${RD}_hbm_traffic_pmc.json "$(git rev-parse --short HEAD)" <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); d['collected_at_commit'] = sys.argv[3]      # the tree the counters were collected from (bench.py quotes it)
json.dump(d, open(sys.argv[2], 'w'))
PY
cp $O/final_conv_sequence.json profiles/${RD}_conv_sequence.json
{ echo "# per conv layer shape of one search iteration (bench.py --conv-sequence): PMC HBM bytes per launch (FETCH_SIZE x2 + WRITE_SIZE, separate rocprofv3 passes) next to the algorithmic bytes"
  echo "# (every input / weight / residual element read once, every output element written once).  x 'launches per step'; time and TFLOP/s from the in-process dispatch events."
  cat $O/final_pmc_per_shape.txt; } > profiles/${RD}_conv_per_shape_pmc.txt
grep -v amdgpu.ids $O/final_bench.err > profiles/${RD}_bench_stderr_conv_table.txt
for p in mcts:mcts_s256 2rank_gloo:2rank_gloo_one_gpu f16:f16 rccl1:rccl_one_rank f16x3:f16x3 f32:f32; do
  a=${p%%:*}; b=${p##*:}; grep "^{" $O/final_bench_$a.json > profiles/${RD}_bench_$b.json
done
for n in 1 2 4 8; do [ -f $O/final_scale_n$n.json ] && grep "^{" $O/final_scale_n$n.json > profiles/${RD}_scale_n$n.json; done
{ echo "# DTS_SHARD_ALWAYS_COLLECT=1 NCCL_DEBUG=VERSION python bench.py ... (one MI355X): a ONE-rank RCCL process group whose reward all-gather, barriers and"
  echo "# max-over-ranks all-reduce are all issued on device tensors"
  grep -v amdgpu $O/final_bench_rccl1.err | tail -8; } > profiles/${RD}_bench_rccl_one_rank.log
grep -v amdgpu.ids $O/final_vae.txt > profiles/${RD}_vae_decode.txt
grep -v amdgpu.ids $O/final_att.txt > profiles/${RD}_attention.txt
{ echo "# tools/sd_bench.py on one MI355X: BASELINE config 4 with this build's parts (see profiles/r03_sd_config4.txt header in git history / DESIGN.md section 5)"
  grep "^SD beam" $O/final_sd.txt; } > profiles/${RD}_sd_config4_final.txt
grep -v amdgpu.ids $O/final_bench_f16x3.err > profiles/${RD}_bench_f16x3_stderr_conv_table.txt
# rocprof average of the dominant kernel next to the in-process one, so that roofline.frac can be recomputed from profiles/ alone
python - $RD <<'PY'
import csv, json, sys
rd = sys.argv[1]
d = json.loads(open(f'profiles/{rd}_bench.json').read().strip().splitlines()[-1])
r = d['roofline']
rows = list(csv.DictReader(open(f'profiles/{rd}_bench_kernel_stats.csv')))
dom = [x for x in rows if 'conv_pp_kernel<bf16_t, 9, 0, false, 6' in x['Name']][0]
rp_us = float(dom['AverageNs']) / 1e3
out = {'kernel': dom['Name'], 'rocprofv3_avg_launch_us': round(rp_us, 2), 'rocprofv3_calls': int(dom['Calls']),
       'in_process_avg_launch_us': r['avg_launch_us'], 'avg_launch_gflop': r['avg_launch_gflop'],
       'frac_from_rocprofv3': round(r['avg_launch_gflop'] / rp_us * 1e3 / r['peak'], 4), 'frac_in_process': r['frac'],
       'ratio_rocprof_over_in_process': round(rp_us / r['avg_launch_us'], 4),
       'how': 'rocprofv3 --kernel-trace --stats of `bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-e2e --no-parity --no-subrecords` (same '
              'workload, same graphs; a profiled process runs a few per cent slower: MI355X_MICROARCH.md DVFS item 2); in-process = HIP events attached to the '
              'kernel\'s own dispatch in the default bench.py run'}
try:      # the SAME profiled process measured both ways (final_run.sh: final_prof_both)
    import glob
    b = json.loads([l for l in open('gpurun_out/final_prof_both.json') if l.startswith('{')][-1])['roofline']
    rows2 = list(csv.DictReader(open(sorted(glob.glob('gpurun_out/final_prof_both/*/*kernel_stats.csv'))[-1])))
    dom2 = [x for x in rows2 if 'conv_pp_kernel<bf16_t, 9, 0, false, 6' in x['Name']][0]
    out['same_process'] = {'in_process_avg_launch_us': b['avg_launch_us'], 'rocprofv3_avg_launch_us': round(float(dom2['AverageNs']) / 1e3, 2),
                           'ratio': round(float(dom2['AverageNs']) / 1e3 / b['avg_launch_us'], 4), 'frac_in_process': b['frac'],
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
print('value',d['value'],'ms',d['ms_per_step'],'dom',r['kernel'],r['achieved'],r['frac'],'all',r['all_conv']['achieved'],r['all_conv']['frac'],'step frac',r['whole_step_frac'])
print('e2e',d.get('e2e_evals_per_s'),'cpu',d['cpu_baseline']['value'], 'agreement', d['parity']['index_agreement'])
for k,v in d.get('sub_records',{}).items(): print(k, v.get('value'), v.get('ms_per_step'), (v.get('roofline') or {}).get('frac'), v.get('cpu_baseline'))
PY
