#!/bin/bash
# Round 4, GPU session L: does a wider construction beam (150) add to what the improveConnections pass gave?  (finer rerankK ladder)
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r4l; mkdir -p $O
cd $R
timeout 1500 python bench.py --build-beam 150 --steps 6 --warmup 2 --no-cpu-baseline --no-flat --no-sub-workloads > $O/bench_beam150.json 2> $O/bench_beam150.err
echo "bench rc=$?" | tee -a $O/summary.txt
grep -E "calibrate|evaluate|layered" $O/bench_beam150.err | cut -c1-300 | tail -8 | tee -a $O/summary.txt
python - <<'PY' | tee -a $O/summary.txt
import json,os
d=os.path.join(os.environ.get("GRAFT_REPO_ROOT","."),"gpurun_out/r4l")
try:
    l=json.loads([x for x in open(os.path.join(d,"bench_beam150.json")).read().splitlines() if x.startswith("{")][-1])
    print("BEAM150", l["value"], l["ms_per_step"], l["recall_at_10"], l["config"]["rerankK"], l["avg_expanded"], l["graph_build_s"])
except Exception as e:
    print("no line", e)
PY
