#!/bin/bash
# Round 6, GPU session F: the scoring pass with 8 / 16 / 32 lanes per survivor by survivor count (two thirds of the headline's expansions
# keep <= 4 of their fresh neighbours: one round of codebook requests instead of two) + the exchange area on a 16-byte boundary.
# Parity of every traversal form, then the headline with the phase clocks.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r6f; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_zz_ubr_gpu.py tests/test_zz_device_traversal_gpu.py tests/test_graph_search.py tests/test_builder.py tests/test_zz_builder_reference_order_gpu.py -m gpu -x -q > $O/pytest.txt 2>&1
echo "pytest rc=$?" | tee -a $O/summary.txt
tail -3 $O/pytest.txt | tee -a $O/summary.txt
JVECTOR_BENCH_ENV_SWEEP="JVECTOR_HIP_GS_WAVES_PER_CU=4,JVECTOR_HIP_GS_PROF=1;JVECTOR_HIP_GS_PROF=1" \
  timeout 1500 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-flat --no-sub-workloads --cal-queries 4096 > $O/bench.json 2> $O/bench.err
echo "bench rc=$?" | tee -a $O/summary.txt
grep -E "sweep|prof\]|evaluate" $O/bench.err | cut -c1-400 | awk '!seen[$0]++' | tee -a $O/summary.txt
python - <<'PY' | tee -a $O/summary.txt
import json,os
d=os.path.join(os.environ.get("GRAFT_REPO_ROOT","."),"gpurun_out/r6f")
l=json.loads(open(os.path.join(d,"bench.json")).read().strip().splitlines()[-1])
print("DEFAULT", l["value"], l["ms_per_step"], l.get("kernel_ms_per_step"))
PY
