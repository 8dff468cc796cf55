#!/bin/bash
# SQ counters per kernel over one N = 64 search iteration in the headline mode: three --pmc passes, kernel trace only (never combined with other trace domains)
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
Q="--steps 1 --warmup 0 --no-cpu-baseline --no-kernel-timing --no-e2e --no-parity --no-subrecords"
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAVES"
P2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM"
i=0
for P in "$P1" "$P2"; do
  i=$((i+1)); rm -rf $O/step_pmc_$i
  rocprofv3 --kernel-trace --pmc $P --output-format csv -d $O/step_pmc_$i -- python3 $R/bench.py $Q > $O/step_pmc_$i.log 2>&1 || { tail -5 $O/step_pmc_$i.log; exit 1; }
done
python3 - <<'PY'
import csv, glob, collections, os, re
O=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out'
tot=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(lambda: collections.defaultdict(int))
for i in (1,2):
    for f in glob.glob(f'{O}/step_pmc_{i}/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            k=r['Kernel_Name'].replace('(anonymous namespace)::','').replace('void ','')
            k=re.sub(r'\(.*$','',k)[:70]
            tot[k][r['Counter_Name']]+=float(r['Counter_Value']); n[k][r['Counter_Name']]+=1
rows=sorted(tot.items(), key=lambda kv:-kv[1].get('SQ_WAVE_CYCLES',0))[:12]
with open(f'{O}/step_pmc_summary.txt','w') as g:
    g.write('# per kernel over the process (graph capture + 1 timed step): fractions of SQ_WAVE_CYCLES (wave residency); pipe = SQ_VALU_MFMA_BUSY_CYCLES / (4 x SQ_WAVE_CYCLES) x resident waves per SIMD is NOT applied: "mfma/wave" = matrix-pipe cycles per wave-cycle of ONE wave\n')
    for k,v in rows:
        w=v.get('SQ_WAVE_CYCLES',1.0)
        g.write(f'{k:70s} launches {n[k].get("SQ_WAVE_CYCLES",0):5d}  wait_any {v.get("SQ_WAIT_ANY",0)/w:5.2f}  wait_inst {v.get("SQ_WAIT_INST_ANY",0)/w:5.2f} (lds {v.get("SQ_WAIT_INST_LDS",0)/w:4.2f})  active {v.get("SQ_ACTIVE_INST_ANY",0)/w:5.2f} (valu {v.get("SQ_ACTIVE_INST_VALU",0)/w:4.2f} lds {v.get("SQ_ACTIVE_INST_LDS",0)/w:4.2f} vmem {v.get("SQ_ACTIVE_INST_VMEM",0)/w:4.2f} sca {v.get("SQ_ACTIVE_INST_SCA",0)/w:4.2f})  mfma/wave {v.get("SQ_VALU_MFMA_BUSY_CYCLES",0)/(4*w):5.3f}  VALU/MFMA {v.get("SQ_INSTS_VALU",0)/max(1,v.get("SQ_INSTS_MFMA",1)):5.2f}  bankconf {v.get("SQ_LDS_BANK_CONFLICT",0)/max(1,v.get("SQ_ACTIVE_INST_LDS",1)):4.2f}\n')
print(open(f'{O}/step_pmc_summary.txt').read())
PY
find $O/step_pmc_* -name "*kernel_trace.csv" -delete; find $O/step_pmc_* -name "*.db" -delete; find $O/step_pmc_* -name "*counter_collection.csv" -size +20M -delete
