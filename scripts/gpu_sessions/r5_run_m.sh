#!/bin/bash
# Round 5, GPU session M (the exact stage bounded by the longest survivor list, no memset): the flat search's two-stage filter (k_adc_bq.hip) — parity with the single-stage filter and the oracle, the flat
# tests that take the filtered path, then C2, one C4 shard and the sharded C-ABI test with the form on / off
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r5m; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_zz_flat_bq_gpu.py tests/test_gpu_parity.py tests/test_sharded_cabi.py tests/test_sharded.py -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/summary.txt; tail -4 $O/pytest.log | tee -a $O/summary.txt
for BQ in 1 0; do
  for W in c2 c4; do
    JVECTOR_HIP_ADC_BQ=$BQ timeout 600 python bench.py --workload $W --no-cpu-baseline > $O/${W}_bq$BQ.out 2> $O/${W}_bq$BQ.err; echo "$W bq=$BQ rc=$?" | tee -a $O/summary.txt
    python - $O/${W}_bq$BQ.out <<'PY' | tee -a $O/summary.txt
import json,sys
l=[json.loads(x) for x in open(sys.argv[1]).read().strip().splitlines() if x.startswith("{")][-1]
print("  ", l["value"], l["unit"], "ms/step", l["ms_per_step"], "recall", l.get("recall_at_10"), l.get("kernel_ms_per_step"))
PY
  done
done
