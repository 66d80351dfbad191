#!/bin/bash
# Round 6, GPU session M: how much of the rerank's time is the UNDERFILLED second wavefront of a 76-candidate list?  The same index at
# rerankK 64 (one full wave per query), 76 (64 + 12), 128 (two full waves): exact-region ms per step.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r6m; mkdir -p $O
cd $R
C=/tmp/jv_index_m.npz
for rk in 76 64 128 12; do
  timeout 900 python bench.py --index-cache $C --rerank $rk --steps 6 --warmup 2 --no-cpu-baseline --no-flat --no-sub-workloads --cal-queries 1024 --eval-queries 1024 > $O/bench_$rk.json 2> $O/bench_$rk.err
  echo "rerank $rk rc=$?" | tee -a $O/summary.txt
done
python - <<'PY' | tee -a $O/summary.txt
import json,os
d=os.path.join(os.environ.get("GRAFT_REPO_ROOT","."),"gpurun_out/r6m")
for rk in (12,64,76,128):
    try:
        l=[json.loads(x) for x in open(os.path.join(d,f"bench_{rk}.json")).read().strip().splitlines() if x.startswith("{")][-1]
        print(rk, l["ms_per_step"], l.get("kernel_ms_per_step"), l.get("recall_at_10"))
    except Exception as e:
        print(rk, "failed", e)
PY
