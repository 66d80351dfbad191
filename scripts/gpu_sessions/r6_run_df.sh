#!/bin/bash
# Round 6, GPU session DF: DEFERRED exact scores above level 0 (gs_body.h DEFER, option gs_defer): parity of the bound form on the
# device, then the headline with the option off / from level 2 / from level 1, same index, same process.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r6df; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_zz_ubr_gpu.py tests/test_zz_device_traversal_gpu.py -m gpu -x -q > $O/pytest.txt 2>&1
echo "pytest rc=$?" | tee $O/summary.txt
tail -3 $O/pytest.txt | tee -a $O/summary.txt
JVECTOR_BENCH_ENV_SWEEP="JVECTOR_HIP_GS_DEFER=0;JVECTOR_HIP_GS_DEFER=1;JVECTOR_HIP_GS_DEFER=1,JVECTOR_HIP_GS_DEFER_MIN_LEVEL=1;JVECTOR_HIP_GS_DEFER=1,JVECTOR_HIP_GS_DEFER_MIN_LEVEL=3;JVECTOR_HIP_GS_DEFER=0;JVECTOR_HIP_GS_DEFER=1" \
  timeout 1500 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-flat --no-sub-workloads --cal-queries 4096 --rerank 74 > $O/bench.json 2> $O/bench.err
echo "bench rc=$?" | tee -a $O/summary.txt
grep -E "sweep|evaluate" $O/bench.err | cut -c1-420 | awk '!seen[$0]++' | tee -a $O/summary.txt
python - <<PY | tee -a $O/summary.txt
import json
l=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print("line", round(l["value"]), round(l["ms_per_step"],2), l.get("kernel_ms_per_step"), l["recall_at_10"])
PY
