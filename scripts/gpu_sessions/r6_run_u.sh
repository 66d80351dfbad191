#!/bin/bash
# Round 6, GPU session U: non-temporal loads for what the traversal uses once — the popped node's row / block / magnitudes (ntb), the
# fused rerank's vector rows (ntr), both (ntbr) — A/B libraries over one cached index; every run also re-times the steps with the
# rerank as a kernel of its own (sweep).
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r6u; mkdir -p $O
cd $R
: > $O/summary.txt
for v in default ntb ntr ntbr; do
  if [ $v = default ]; then unset JVECTOR_HIP_LIBRARY; else export JVECTOR_HIP_LIBRARY=$R/build/variants/libjvector_hip_$v.so; fi
  JVECTOR_BENCH_ENV_SWEEP="JVECTOR_HIP_GS_FUSED_RERANK=0;JVECTOR_HIP_GS_FUSED_RERANK=1" \
    timeout 900 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-flat --no-sub-workloads --cal-queries 4096 --rerank 74 --index-cache /tmp/idx10m.npz > $O/bench_$v.json 2> $O/bench_$v.err
  echo "== $v rc=$?" | tee -a $O/summary.txt
  grep -E "sweep" $O/bench_$v.err | cut -c1-300 | tee -a $O/summary.txt
  python - <<PY | tee -a $O/summary.txt
import json
l=json.loads(open("$O/bench_$v.json").read().strip().splitlines()[-1])
print("line $v", round(l["value"]), round(l["ms_per_step"],2), l.get("kernel_ms_per_step"), l["recall_at_10"])
PY
done
