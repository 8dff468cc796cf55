#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2z; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES --output-format csv -d $O/pmc_a -- python3 $R/tools/conv_bench.py --n 64 --iters 3 --no-res --stats --variants conv_variant=1 > $O/pmc_a.log 2>&1; echo "rc a=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD --output-format csv -d $O/pmc_b -- python3 $R/tools/conv_bench.py --n 64 --iters 3 --no-res --stats --variants conv_variant=1 > $O/pmc_b.log 2>&1; echo "rc b=$?"
find $O -name "*kernel_trace.csv" -size +20M -delete
python3 - <<PY
import csv,glob,collections
for d in ('pmc_a','pmc_b'):
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob('$O/'+d+'/**/*counter_collection.csv',recursive=True):
        for r in csv.DictReader(open(f)):
            if 'conv_pp' in r['Kernel_Name'] or 'conv_igemm' in r['Kernel_Name']:
                agg[(r['Kernel_Name'][29:75],r['Grid_Size'])][r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in agg.items():
        print(d,k,{c:round(sum(x)/len(x)) for c,x in v.items()})
PY
