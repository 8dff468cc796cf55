#!/bin/bash
# SQ counters of attention_x3_kernel (T = 1024, 64 rows): three --pmc passes, kernel trace only
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*" | sort -u > $O/sq_counter_names.txt
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAVES"
P2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM"
P3="SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_WAVE_CYCLES"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1)); rm -rf $O/att_pmc_$i
  rocprofv3 --kernel-trace --pmc $P --output-format csv -d $O/att_pmc_$i -- python3 $R/tools/att_one.py > $O/att_pmc_$i.log 2>&1 || { tail -5 $O/att_pmc_$i.log; }
done
python3 - <<'PY'
import csv, glob, collections, os
O=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out'
tot=collections.defaultdict(float); n=collections.defaultdict(int)
for i in (1,2,3):
    for f in glob.glob(f'{O}/att_pmc_{i}/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if 'attention_x3' in r['Kernel_Name']:
                tot[r['Counter_Name']]+=float(r['Counter_Value']); n[r['Counter_Name']]+=1
with open(f'{O}/att_pmc_summary.txt','w') as g:
    for k in sorted(tot): g.write(f'{k:32s} {tot[k]/max(1,n[k]):16.0f} per launch ({n[k]} launches)\n')
print(open(f'{O}/att_pmc_summary.txt').read())
PY
find $O/att_pmc_* -name "*kernel_trace.csv" -delete; find $O/att_pmc_* -name "*.db" -delete
