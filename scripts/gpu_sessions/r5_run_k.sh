#!/bin/bash
# Round 5, GPU session K: checkpoint — the whole -m gpu suite on the default (non-experimental) library with UBR on, smoke(), then the
# headline with the lane-parallel entry-row scoring (setup clocks per query: was ~72 k)
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r5k; mkdir -p $O
cd $R
( time timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1 ) 2>> $O/summary.txt; echo "pytest rc=$?" | tee -a $O/summary.txt; tail -4 $O/pytest_gpu.log | tee -a $O/summary.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log | cut -c1-160 | tee -a $O/summary.txt
JVECTOR_BENCH_ENV_SWEEP="JVECTOR_HIP_GS_PROF=1" timeout 1200 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-flat --no-sub-workloads --cal-queries 4096 > $O/bench_sweep.json 2> $O/bench_sweep.err
echo "bench rc=$?" | tee -a $O/summary.txt
grep -E "sweep|prof\]|evaluate" $O/bench_sweep.err | cut -c1-400 | awk '!seen[$0]++' | tee -a $O/summary.txt
python - <<'PY' | tee -a $O/summary.txt
import json,os
d=os.path.join(os.environ.get("GRAFT_REPO_ROOT","."),"gpurun_out/r5k")
l=json.loads(open(os.path.join(d,"bench_sweep.json")).read().strip().splitlines()[-1])
print("DEFAULT", l["value"], l["ms_per_step"], l["kernel_ms_per_step"], l["roofline"]["frac"])
PY
