#!/bin/bash
# Round 5, GPU session X: the round's last library (bound tables with the floored quotient + 2^-10 bucket): the whole -m gpu suite,
# smoke(), the driver's command, then the rocprofv3 passes of scripts/profile_r5.sh on the same library
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r5x; mkdir -p $O
cd $R
( time timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1 ) 2>> $O/summary.txt; echo "pytest rc=$?" | tee -a $O/summary.txt; tail -2 $O/pytest_gpu.log | tee -a $O/summary.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $O/summary.txt; tail -2 $O/smoke.log | cut -c1-200 | tee -a $O/summary.txt
( time timeout 1500 python bench.py --gpus 1 > $O/bench_default.out 2> $O/bench_default.err ) 2>> $O/summary.txt; echo "bench rc=$?" | tee -a $O/summary.txt
cp bench_full.json $O/bench_full.json 2>/dev/null
grep -E "evaluate|sub-run|batch sweep" $O/bench_default.err | cut -c1-200 | tee -a $O/summary.txt
python - $O/bench_default.out <<'PY' | tee -a $O/summary.txt
import json,sys
lines=open(sys.argv[1]).read().strip().splitlines()
l=json.loads(lines[-1])
print("stdout lines", len(lines), "last line bytes", len(lines[-1]))
print("DEFAULT", l["value"], l["unit"], "ms/step", l["ms_per_step"], "recall", l.get("recall_at_10"), "rerankK", l["config"].get("rerankK"))
print("roofline", {k: v for k, v in l["roofline"].items() if k not in ("kernel", "note")})
print("cpu_baseline", l.get("cpu_baseline"))
print("workloads", json.dumps(l.get("workloads"))[:1500])
PY
bash scripts/profile_r5.sh r5_10m > $O/profile.log 2>&1; echo "profile rc=$?" | tee -a $O/summary.txt
