#!/bin/bash
# Round 6, GPU session V: the whole -m gpu suite on the library with the fused rerank + the coalesced query-norm kernels, then the
# headline (no sub-runs) with its env sweep fused on / off.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r6v; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1
echo "pytest rc=$?" | tee $O/summary.txt
tail -3 $O/pytest.txt | tee -a $O/summary.txt
JVECTOR_BENCH_ENV_SWEEP="JVECTOR_HIP_GS_FUSED_RERANK=0;JVECTOR_HIP_GS_FUSED_RERANK=1" \
  timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-flat --no-sub-workloads > $O/bench.json 2> $O/bench.err
echo "bench rc=$?" | tee -a $O/summary.txt
grep -E "sweep|evaluate" $O/bench.err | cut -c1-300 | tee -a $O/summary.txt
python - <<PY | tee -a $O/summary.txt
import json
l=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print("line", round(l["value"]), round(l["ms_per_step"],2), l.get("kernel_ms_per_step"), l["recall_at_10"], l["roofline"].get("frac"), l["roofline"].get("frac_traversal_bytes_only"))
PY
