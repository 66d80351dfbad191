#!/bin/bash
# Round 5, GPU session A: the new reference-native comparison tests on the MI355X, then the default bench command — is the LAST stdout line the compact one (< 4 KB) with roofline + cpu_baseline?
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r5a; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_zz_ref_native_gpu.py -m gpu -q > $O/pytest_ref_native.log 2>&1; echo "pytest rc=$?" | tee -a $O/summary.txt; tail -5 $O/pytest_ref_native.log | tee -a $O/summary.txt
( time timeout 1500 python bench.py --gpus 1 > $O/bench_default.out 2> $O/bench_default.err ) 2>> $O/summary.txt; echo "bench rc=$?" | tee -a $O/summary.txt
tail -c 200 $O/bench_default.out | head -c 0
python - <<'PY' | tee -a $O/summary.txt
import json,os
d=os.path.join(os.environ.get("GRAFT_REPO_ROOT","."),"gpurun_out/r5a")
out=open(os.path.join(d,"bench_default.out")).read().strip().splitlines()
print("stdout lines", len(out), "last line bytes", len(out[-1]) if out else None)
l=json.loads(out[-1])
print("DEFAULT", l["value"], l["ms_per_step"], l.get("recall_at_10"), l["config"]["rerankK"], l["roofline"]["frac"], l["cpu_baseline"]["value"], l.get("full"))
print(json.dumps(l.get("workloads")))
PY
cp $R/bench_full.json $O/ 2>/dev/null
