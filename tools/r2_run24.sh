#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2x; mkdir -p $O
( time python bench.py ) > $O/bench_default.json 2> $O/bench_default.err; echo "rc=$?"; tail -4 $O/bench_default.err; python - <<PY
import json
d=json.loads([l for l in open('$O/bench_default.json') if l.startswith('{')][-1])
print(d['value'], d['ms_per_step'], d.get('other_dtype'), d['parity']['index_agreement']['f16'], d['parity']['index_agreement']['bf16'], d['roofline']['frac'], d.get('e2e_evals_per_s'))
PY
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
