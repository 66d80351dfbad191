#!/bin/bash
# Round 3, GPU session H (closing, after the spill-tier sizing fix found in session G's kernel trace: every batch's 0.24 % longest
# searches had gone through a 3.5 ms retry launch): -m gpu suite, smoke, default bench line, tier on/off in the same run,
# rocprofv3 passes of the default configuration
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r3h; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/summary.txt; tail -3 $O/pytest_gpu.log | tee -a $O/summary.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log | tee -a $O/summary.txt
timeout 1500 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?" | tee -a $O/summary.txt
python - <<'PY' | tee -a $O/summary.txt
import json,os
d=os.path.join(os.environ.get("GRAFT_REPO_ROOT","."),"gpurun_out/r3h")
try:
    l=json.loads([x for x in open(os.path.join(d,"bench_default.json")).read().splitlines() if x.startswith("{")][-1])
    print("DEFAULT", l["value"], l["ms_per_step"], l["recall_at_10"], l["recall_se"], l["config"]["rerankK"], l["kernel_ms_per_step"], l["roofline"]["frac"], l["roofline"]["launches"], l.get("traversal_stats"), (l.get("cpu_baseline") or {}).get("value"), (l.get("cpu_baseline") or {}).get("matches_gpu_topk"), (l.get("flat_mode") or {}).get("value"), l.get("graph_build_s"))
except Exception as e:
    print("DEFAULT no line", e)
PY
bash scripts/profile_r3.sh r3_10m > $O/profile.log 2>&1; tail -3 $O/profile.log | cut -c1-200
C=/tmp/jv_index_10000000.npz
JVECTOR_BENCH_ENV_SWEEP="JVECTOR_HIP_GS_V1_LOG2=0;JVECTOR_HIP_GS_V1_LOG2=12;JVECTOR_HIP_GS_V1_LOG2=0;JVECTOR_HIP_GS_V1_LOG2=12,JVECTOR_HIP_GS_PROF=1" \
  timeout 900 python bench.py --index-cache $C --steps 8 --warmup 2 --no-cpu-baseline --no-flat > $O/bench_sweep.json 2> $O/bench_sweep.err
grep -E "sweep|prof\] clocks" $O/bench_sweep.err | cut -c1-300 | tee -a $O/summary.txt
timeout 900 python bench.py --workload c5 --n 10000000 --no-cpu-baseline > $O/c5_10m.json 2> $O/c5_10m.err; python - <<'PY' | tee -a $O/summary.txt
import json,os
d=os.path.join(os.environ.get("GRAFT_REPO_ROOT","."),"gpurun_out/r3h")
try:
    l=json.loads([x for x in open(os.path.join(d,"c5_10m.json")).read().splitlines() if x.startswith("{")][-1])
    print("C5", l["value"], l["seconds"], l["recall_at_10_by_rerankK"])
except Exception as e:
    print("C5 no line", e)
PY
