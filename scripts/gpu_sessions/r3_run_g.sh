#!/bin/bash
# Round 3, GPU session G (closing): the whole -m gpu suite, smoke, the default bench line, the rocprofv3 passes of that very
# configuration (scripts/profile_r3.sh), the GraphSearcher-object QPS
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r3g; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/summary.txt; tail -3 $O/pytest_gpu.log | tee -a $O/summary.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log | tee -a $O/summary.txt
timeout 1500 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?" | tee -a $O/summary.txt
python - <<'PY' | tee -a $O/summary.txt
import json,os
d=os.path.join(os.environ.get("GRAFT_REPO_ROOT","."),"gpurun_out/r3g")
try:
    l=json.loads([x for x in open(os.path.join(d,"bench_default.json")).read().splitlines() if x.startswith("{")][-1])
    print("DEFAULT", l["value"], l["ms_per_step"], l["recall_at_10"], l["recall_se"], l["config"]["rerankK"], l["kernel_ms_per_step"], l["roofline"]["frac"], l["roofline"]["traffic"], (l.get("cpu_baseline") or {}).get("value"), (l.get("cpu_baseline") or {}).get("matches_gpu_topk"), (l.get("flat_mode") or {}).get("value"), l.get("graph_build_s"), l.get("encode"))
except Exception as e:
    print("DEFAULT no line", e)
PY
timeout 900 python scripts/searcher_bench.py > $O/searcher_bench.json 2> $O/searcher_bench.err; tail -1 $O/searcher_bench.json | tee -a $O/summary.txt
bash scripts/profile_r3.sh r3_10m > $O/profile.log 2>&1; tail -5 $O/profile.log | cut -c1-300
