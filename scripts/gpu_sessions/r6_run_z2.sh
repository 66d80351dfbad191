#!/bin/bash
# Round 6, GPU session Z2: would ordering a batch's queries by locality pay?  The upper bound: every batch sorted by the mixture cluster
# nearest to the query (bench.py JVECTOR_BENCH_SORT_QUERIES), same index, same process.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r6z2; mkdir -p $O
cd $R
JVECTOR_BENCH_SORT_QUERIES=1 timeout 900 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-flat --no-sub-workloads --cal-queries 4096 --rerank 74 > $O/bench.json 2> $O/bench.err
echo "bench rc=$?" | tee $O/summary.txt
grep -E "sorted queries" $O/bench.err | cut -c1-300 | tee -a $O/summary.txt
python - <<PY | tee -a $O/summary.txt
import json
l=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print("line", round(l["value"]), round(l["ms_per_step"],2), l.get("kernel_ms_per_step"), l["recall_at_10"])
PY
