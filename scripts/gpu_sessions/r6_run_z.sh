#!/bin/bash
# Round 6, GPU session Z: three chunks of a rerank round in flight (GS_RR_DEPTH=3, an A/B library) against two, over one cached index.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r6z; mkdir -p $O
cd $R
: > $O/summary.txt
for v in default rr3 default rr3; do
  if [ $v = default ]; then unset JVECTOR_HIP_LIBRARY; else export JVECTOR_HIP_LIBRARY=$R/build/variants/libjvector_hip_$v.so; fi
  timeout 900 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-flat --no-sub-workloads --cal-queries 4096 --rerank 74 --index-cache /tmp/idx10m.npz > $O/bench_$v.json 2> $O/bench_$v.err
  echo "== $v rc=$?" | tee -a $O/summary.txt
  python - <<PY | tee -a $O/summary.txt
import json
l=json.loads(open("$O/bench_$v.json").read().strip().splitlines()[-1])
print("line $v", round(l["value"]), round(l["ms_per_step"],2), l.get("kernel_ms_per_step"), l["recall_at_10"])
PY
done
