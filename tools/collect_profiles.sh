#!/bin/bash
# copies the summaries of tools/final_run.sh (gpurun_out/final_*) into profiles/ under their round-2 names
set -e
cd "$(dirname "$0")/.."
O=gpurun_out
new=$(ls -t $O/final_prof/*/*kernel_stats.csv | head -1)
cp "$new" profiles/r02_bench_kernel_stats.csv
cp $O/final_bench.json profiles/r02_bench.json
cp $O/final_hbm_traffic_pmc.json profiles/r02_hbm_traffic_pmc.json
cp $O/final_conv_sequence.json profiles/r02_conv_sequence.json
{ echo "# per conv layer shape of one search iteration (bench.py --conv-sequence): PMC HBM bytes per launch (FETCH_SIZE x2 + WRITE_SIZE, separate rocprofv3 passes) next to the algorithmic bytes"
  echo "# (every input / weight / residual element read once, every output element written once).  x 'launches per step'; time and TFLOP/s from the in-process dispatch events."
  echo "# Split-K layers (8x8 level, some 16x16) carry their f32 slab write + re-read: x4-6; the 64x64 and 32x32 layers sit at x1.0-1.6."
  cat $O/final_pmc_per_shape.txt; } > profiles/r02_conv_per_shape_pmc.txt
grep -v amdgpu.ids $O/final_bench.err > profiles/r02_bench_stderr_conv_table.txt
for p in rej32:ddpmpp32_rejection cand8:candidates8 mcts:mcts_s256 2rank_gloo:2rank_gloo_one_gpu f16:f16 rccl1:rccl_one_rank; do
  a=${p%%:*}; b=${p##*:}; grep "^{" $O/final_bench_$a.json > profiles/r02_bench_$b.json
done
{ echo "# DTS_SHARD_ALWAYS_COLLECT=1 NCCL_DEBUG=VERSION python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-e2e --no-kernel-timing  (one MI355X):"
  echo "# a ONE-rank RCCL process group (init with device_id) whose reward all-gather, barriers and max-over-ranks all-reduce are all issued on device tensors"
  grep -v amdgpu $O/final_bench_rccl1.err | tail -8; } > profiles/r02_bench_rccl_one_rank.log
grep -v amdgpu.ids $O/final_vae.txt > profiles/r02_vae_decode.txt
grep -v amdgpu.ids $O/final_att.txt > profiles/r02_attention_xcd_n64.txt
{ echo "# tools/pp_stress.py on an MI355X: conv_pp_kernel forced (DTS_CONV_VARIANT=1) on every production shape class, 12 launches each with L2/MALL evictions and idle"
  echo "# gaps in between: every launch bit-identical to the first (outputs and strip statistics) and within one output ulp of the f32 parity kernel."
  grep -v amdgpu $O/final_pp_stress.txt; } > profiles/r02_conv_pp_stress.txt
{ echo "# tools/sd_bench.py on one MI355X: BASELINE config 4 with this build's parts -- SD beam search B=4, N=16, [N,4,64,64] fp16 latents, candidate-batched U-Net calls"
  echo "# (stand-in U-Net), fused DDIM candidate step, N-row decodes through the HIP VAE decoder (SD-1.5 width, random init); 4 DDIM steps = 260 candidate decodes."
  echo "# The CLIP line pays the reference's host-side PIL image processor (transformers CLIPImageProcessor) for every candidate; brightness shows the loop + decoder."
  grep "^SD beam" $O/final_sd.txt; } > profiles/r02_sd_config4.txt
tail -2 $O/final_pytest.log
python - <<'PY'
import json
d=json.loads(open('profiles/r02_bench.json').read().strip().splitlines()[-1])
r=d['roofline']
print('value',d['value'],'ms',d['ms_per_step'],'dom',r['kernel'],r['achieved'],r['frac'],'all',r['all_conv']['achieved'],r['all_conv']['frac'],'step frac',r['whole_step_frac'])
print('e2e',d.get('e2e_evals_per_s'),d.get('e2e_seconds_per_image'),'cpu',d['cpu_baseline']['value'])
p=d['parity']; print({k:p[k] for k in ('f32','f16','bf16')}); print(p['index_agreement']['f16'],p['index_agreement']['bf16'],p['index_agreement']['f32_reward_given_up'])
for n in ('ddpmpp32_rejection','candidates8','mcts_s256','2rank_gloo_one_gpu','f16','rccl_one_rank'):
    e=json.loads(open(f'profiles/r02_bench_{n}.json').read().strip().splitlines()[-1]); print(n,e['value'],e['ms_per_step'],e.get('weak_value'),(e.get('roofline') or {}).get('frac'))
PY
