#!/bin/bash
# Round 6, GPU session P: the builder's searches through gs_ubr_pass (batched reads, 32 / 16 / 8 lanes per survivor): the builder's
# identity tests, then the headline with its build seconds (round 5 / session G: 37.5 s, search 13.9 s).
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r6p; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_builder.py tests/test_zz_builder_reference_order_gpu.py tests/test_zz_ubr_gpu.py tests/test_builder_reference_goldens.py -m gpu -x -q > $O/pytest.txt 2>&1
echo "pytest rc=$?" | tee -a $O/summary.txt
tail -3 $O/pytest.txt | tee -a $O/summary.txt
timeout 1500 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-flat --no-sub-workloads --cal-queries 4096 > $O/bench.json 2> $O/bench.err
echo "bench rc=$?" | tee -a $O/summary.txt
grep -E "^\[build\]|evaluate" $O/bench.err | cut -c1-400 | tee -a $O/summary.txt
python - <<'PY' | tee -a $O/summary.txt
import json,os
d=os.path.join(os.environ.get("GRAFT_REPO_ROOT","."),"gpurun_out/r6p")
l=[json.loads(x) for x in open(os.path.join(d,"bench.json")).read().strip().splitlines() if x.startswith("{")][-1]
print("DEFAULT", l["value"], l["ms_per_step"], l.get("kernel_ms_per_step"), "build_s", l.get("graph_build_s"), "rerank frac", l.get("rerank_roofline_frac"))
PY
