#!/bin/bash
# Round 3, GPU session E: the register-resident ADC table (gs_lutr) on hardware — parity, then A/B at 1M and on the 10M index
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r3e; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_zz_device_traversal_gpu.py tests/test_graph_search.py -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/summary.txt; tail -3 $O/pytest_gpu.log | tee -a $O/summary.txt
SW="JVECTOR_HIP_GS_LUTR=1;JVECTOR_HIP_GS_LUTR=1,JVECTOR_HIP_GS_PROF=1;JVECTOR_HIP_GS_V1_LOG2=0"
JVECTOR_BENCH_ENV_SWEEP="$SW" JVECTOR_HIP_GRAPH_TIMING=1 timeout 600 python bench.py --n 1000000 --steps 5 --warmup 2 --no-flat --no-cpu-baseline --queries 16384 > $O/bench_1m.json 2> $O/bench_1m.err
grep -E "sweep|prof\] clocks|evaluate" $O/bench_1m.err | cut -c1-300 | tee -a $O/summary.txt
grep "graph_search device" $O/bench_1m.err | tail -3 | cut -c1-300 | tee -a $O/summary.txt
C=/tmp/jv_index_10m.npz
JVECTOR_BENCH_ENV_SWEEP="$SW" timeout 1200 python bench.py --index-cache $C --steps 10 --warmup 2 --no-cpu-baseline --no-flat > $O/bench_10m.json 2> $O/bench_10m.err
grep -E "sweep|prof\] clocks|evaluate|\[build\] \{" $O/bench_10m.err | cut -c1-300 | tee -a $O/summary.txt
python - <<'PY' | tee -a $O/summary.txt
import json,os
d=os.path.join(os.environ.get("GRAFT_REPO_ROOT","."),"gpurun_out/r3e")
for f in ("bench_1m","bench_10m"):
    try:
        l=json.loads([x for x in open(os.path.join(d,f+".json")).read().splitlines() if x.startswith("{")][-1])
        print(f, l["value"], l["ms_per_step"], l.get("recall_at_10"), l["config"].get("rerankK"), l.get("kernel_ms_per_step"))
    except Exception as e:
        print(f, "no line", e)
PY
