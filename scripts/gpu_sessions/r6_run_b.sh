#!/bin/bash
# Round 6, GPU session B: the traversal's dependent chain shortened without touching its semantics — (1) the popped node's row + block
# requested before addTopCandidate, (2) the candidate-tier trim moved into that window and run from registers, (3) the bound's drop test in
# f32 with a margin instead of the f64 finish, (4) the scoring loop's LDS reads batched (query sub-vectors 3 entries at a time, the
# owner's column 7 words at a time).  Parity on the device, then the headline with the phase clocks.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r6b; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_zz_ubr_gpu.py tests/test_zz_device_traversal_gpu.py tests/test_graph_search.py -m gpu -x -q > $O/pytest.txt 2>&1
echo "pytest rc=$?" | tee -a $O/summary.txt
tail -3 $O/pytest.txt | tee -a $O/summary.txt
JVECTOR_BENCH_ENV_SWEEP="JVECTOR_HIP_GS_WAVES_PER_CU=4,JVECTOR_HIP_GS_PROF=1;JVECTOR_HIP_GS_PROF=1" \
  timeout 1500 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-flat --no-sub-workloads --cal-queries 4096 > $O/bench_sweep.json 2> $O/bench_sweep.err
echo "bench rc=$?" | tee -a $O/summary.txt
grep -E "sweep|prof\]|evaluate|graph_search device" $O/bench_sweep.err | cut -c1-600 | awk '!seen[$0]++' | tee -a $O/summary.txt
python - <<'PY' | tee -a $O/summary.txt
import json,os
d=os.path.join(os.environ.get("GRAFT_REPO_ROOT","."),"gpurun_out/r6b")
l=json.loads(open(os.path.join(d,"bench_sweep.json")).read().strip().splitlines()[-1])
print("DEFAULT", l["value"], l["ms_per_step"], l.get("kernel_ms_per_step"), l.get("recall"), l.get("config"))
PY
