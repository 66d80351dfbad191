#!/bin/bash
# Round 3, GPU session B: the two-choice LDS tier + the C-ABI builder on hardware.
#   1. -m gpu suite   2. 1M bench with tier sweep   3. 10M default bench (index built by jv_hip_builder_*) with tier sweep   4. C5 at 1M
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r3b; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/summary.txt; tail -5 $O/pytest_gpu.log | tee -a $O/summary.txt
JVECTOR_BENCH_ENV_SWEEP="JVECTOR_HIP_GS_V1_LOG2=0;JVECTOR_HIP_GS_V1_LOG2=11;JVECTOR_HIP_GS_V1_LOG2=12" \
  timeout 600 python bench.py --n 1000000 --steps 5 --warmup 2 --no-flat --no-cpu-baseline --queries 16384 > $O/bench_1m.json 2> $O/bench_1m.err
grep -E "sweep|evaluate|\[build\] \{" $O/bench_1m.err | cut -c1-400 | tee -a $O/summary.txt
C=/tmp/jv_index_10m.npz
JVECTOR_BENCH_ENV_SWEEP="JVECTOR_HIP_GS_V1_LOG2=0;JVECTOR_HIP_GS_V1_LOG2=12;JVECTOR_HIP_GS_V1_LOG2=0;JVECTOR_HIP_GS_V1_LOG2=12,JVECTOR_HIP_GS_PROF=1;JVECTOR_HIP_GS_V1_LOG2=0,JVECTOR_HIP_GS_PROF=1" JVECTOR_HIP_GRAPH_TIMING=0 \
  timeout 1200 python bench.py --index-cache $C --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_10m.json 2> $O/bench_10m.err
grep -E "sweep|prof\] clocks|evaluate|\[build\] \{" $O/bench_10m.err | cut -c1-400 | tee -a $O/summary.txt
timeout 900 python bench.py --workload c5 --n 1000000 > $O/c5_1m.json 2> $O/c5_1m.err
tail -3 $O/c5_1m.err | cut -c1-300 | tee -a $O/summary.txt
python - <<'PY' | tee -a $O/summary.txt
import json,os
d=os.path.join(os.environ.get("GRAFT_REPO_ROOT","."),"gpurun_out/r3b")
for f in ("bench_1m","bench_10m","c5_1m"):
    try:
        l=json.loads([x for x in open(os.path.join(d,f+".json")).read().splitlines() if x.startswith("{")][-1])
        print(f, l["value"], l["ms_per_step"], l.get("recall_at_10"), l["config"].get("rerankK"), l.get("kernel_ms_per_step"), l["roofline"]["frac"] if l.get("roofline") else None, l.get("graph_build_s"), l.get("seconds"), l.get("recall_at_10_by_rerankK"), (l.get("cpu_baseline") or {}).get("value"))
    except Exception as e:
        print(f, "no line", e)
PY
