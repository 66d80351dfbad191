#!/bin/bash
# Round 4, GPU session I: improveConnections as a device pass (jv_hip_builder_improve_batch) — builder tests on hardware, then the
# headline index built with one improve pass over every node of every level: calibrated rerankK, expansions, QPS, build time
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r4i; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_builder.py -m gpu -q > $O/pytest_builder.log 2>&1; echo "pytest builder rc=$?" | tee -a $O/summary.txt; tail -3 $O/pytest_builder.log | tee -a $O/summary.txt
timeout 1500 python bench.py --build-improve 1 --steps 6 --warmup 2 --no-cpu-baseline --no-flat --no-sub-workloads > $O/bench_improve.json 2> $O/bench_improve.err
echo "bench rc=$?" | tee -a $O/summary.txt
grep -E "calibrate|evaluate|improve pass|\[build\] \{" $O/bench_improve.err | cut -c1-400 | tee -a $O/summary.txt
python - <<'PY' | tee -a $O/summary.txt
import json,os
d=os.path.join(os.environ.get("GRAFT_REPO_ROOT","."),"gpurun_out/r4i")
try:
    l=json.loads([x for x in open(os.path.join(d,"bench_improve.json")).read().splitlines() if x.startswith("{")][-1])
    print("IMPROVE", l["value"], l["ms_per_step"], l["recall_at_10"], l["config"]["rerankK"], l["avg_expanded"], l["avg_visited"], l["graph_build_s"], l["graph_build"])
except Exception as e:
    print("no line", e)
PY
