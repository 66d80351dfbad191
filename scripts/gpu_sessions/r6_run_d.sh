#!/bin/bash
# Round 6, GPU session D: (1) the bound-table kernel with pass 1 rewritten (float v_min/v_max DPP chain, one vote per query for
# non-finite entries, results carried in lanes): bytes / meta still equal the restatement?  (2) rerank gather shapes: 64 rows x 64
# floats (default), 32 x 128, 16 x 256 — longer contiguous pieces per gathered row; (3) the headline.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r6d; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_zz_ubr_gpu.py tests/test_gpu_parity.py -m gpu -x -q > $O/pytest.txt 2>&1
echo "pytest rc=$?" | tee -a $O/summary.txt
tail -3 $O/pytest.txt | tee -a $O/summary.txt
for sh in 1 2; do
JVECTOR_HIP_EXACT_TR_SHAPE=$sh timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "exact" > $O/pytest_shape$sh.txt 2>&1
echo "pytest shape $sh rc=$?" | tee -a $O/summary.txt
tail -1 $O/pytest_shape$sh.txt | tee -a $O/summary.txt
done
JVECTOR_BENCH_ENV_SWEEP="JVECTOR_HIP_EXACT_TR_SHAPE=1;JVECTOR_HIP_EXACT_TR_SHAPE=2;JVECTOR_HIP_EXACT_TR_SHAPE=0" \
  timeout 1500 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-flat --no-sub-workloads --cal-queries 4096 > $O/bench.json 2> $O/bench.err
echo "bench rc=$?" | tee -a $O/summary.txt
grep -E "sweep|evaluate" $O/bench.err | cut -c1-300 | tee -a $O/summary.txt
python - <<'PY' | tee -a $O/summary.txt
import json,os
d=os.path.join(os.environ.get("GRAFT_REPO_ROOT","."),"gpurun_out/r6d")
l=json.loads(open(os.path.join(d,"bench.json")).read().strip().splitlines()[-1])
print("DEFAULT", l["value"], l["ms_per_step"], l.get("kernel_ms_per_step"))
PY
