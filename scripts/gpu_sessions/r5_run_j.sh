#!/bin/bash
# Round 5, GPU session J: UBR with trims before a spill and entry products as natural packed multiplies + scalar addition chains (no operand moves); table kernel likewise

timeout 900 python -m pytest tests/test_zz_ubr_gpu.py -m gpu -q -x 2>&1 | tail -2
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r5j; mkdir -p $O
cd $R
JVECTOR_BENCH_ENV_SWEEP="JVECTOR_HIP_GS_PROF=1;JVECTOR_HIP_GS_UBR_TRIM=32" \
  timeout 1200 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-flat --no-sub-workloads --cal-queries 4096 > $O/bench_sweep.json 2> $O/bench_sweep.err
echo "bench rc=$?" | tee -a $O/summary.txt
grep -E "sweep|prof\]|evaluate" $O/bench_sweep.err | cut -c1-400 | awk '!seen[$0]++' | tee -a $O/summary.txt
python - <<'PY' | tee -a $O/summary.txt
import json,os
d=os.path.join(os.environ.get("GRAFT_REPO_ROOT","."),"gpurun_out/r5j")
l=json.loads(open(os.path.join(d,"bench_sweep.json")).read().strip().splitlines()[-1])
print("DEFAULT", l["value"], l["ms_per_step"], l["kernel_ms_per_step"])
PY
