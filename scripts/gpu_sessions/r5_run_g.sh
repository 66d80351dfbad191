#!/bin/bash
# Round 5, GPU session G: UBR on by default — table kernel without data-dependent loops (parity with the restatement), then the DEFAULT
# bench command (all sub-workloads through the new form) exactly as the driver runs it
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r5g; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_zz_ubr_gpu.py -m gpu -q -x > $O/pytest_ubr.log 2>&1; echo "pytest ubr rc=$?" | tee -a $O/summary.txt; tail -3 $O/pytest_ubr.log | tee -a $O/summary.txt
( time timeout 1500 python bench.py --gpus 1 > $O/bench_default.out 2> $O/bench_default.err ) 2>> $O/summary.txt; echo "bench rc=$?" | tee -a $O/summary.txt
grep -E "sub-run|batch sweep|evaluate" $O/bench_default.err | cut -c1-300 | tee -a $O/summary.txt
python - <<'PY' | tee -a $O/summary.txt
import json,os
d=os.path.join(os.environ.get("GRAFT_REPO_ROOT","."),"gpurun_out/r5g")
out=open(os.path.join(d,"bench_default.out")).read().strip().splitlines()
print("stdout lines", len(out), "last line bytes", len(out[-1]) if out else None)
l=json.loads(out[-1])
print("DEFAULT", l["value"], l["ms_per_step"], l.get("recall_at_10"), l["config"]["rerankK"], l["roofline"]["frac"], l["roofline"]["kernel"][:40], l["cpu_baseline"]["value"], l["cpu_baseline"]["matches_gpu_topk"], l.get("kernel_ms_per_step"))
print(json.dumps(l.get("workloads")))
PY
cp $R/bench_full.json $O/ 2>/dev/null
