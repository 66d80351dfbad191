#!/bin/bash
# Round 4, GPU session J: two improveConnections passes; and one pass with a narrower search (beam 64) for the improve pass only
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r4j; mkdir -p $O
cd $R
for cfg in "2"; do
timeout 1500 python bench.py --build-improve $cfg --steps 6 --warmup 2 --no-cpu-baseline --no-flat --no-sub-workloads > $O/bench_improve$cfg.json 2> $O/bench_improve$cfg.err
echo "bench improve=$cfg rc=$?" | tee -a $O/summary.txt
grep -E "evaluate|\[build\] \{" $O/bench_improve$cfg.err | cut -c1-300 | tee -a $O/summary.txt
python - $cfg <<'PY' | tee -a $O/summary.txt
import json,os,sys
d=os.path.join(os.environ.get("GRAFT_REPO_ROOT","."),"gpurun_out/r4j")
try:
    l=json.loads([x for x in open(os.path.join(d,f"bench_improve{sys.argv[1]}.json")).read().splitlines() if x.startswith("{")][-1])
    print("IMPROVE", sys.argv[1], l["value"], l["ms_per_step"], l["recall_at_10"], l["config"]["rerankK"], l["avg_expanded"], l["avg_visited"], l["graph_build_s"])
except Exception as e:
    print("no line", e)
PY
done
